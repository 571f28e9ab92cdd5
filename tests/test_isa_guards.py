"""CPU (needs hipcc, which cross-compiles gfx950 without a GPU): two places of the kernels manage hardware wait counters by
hand, which is only correct if the compiler leaves the instructions around them where the source put them.  The build is
reproducible, so the ISA of the shipped sources is checked here:

  k_pyramid_rows  the two grey rows a step fetches are loaded by inline `buffer_load_dword` and waited for by an inline
                  `s_waitcnt vmcnt(<stores of this step>)` (csrc/pyramid.hip).  The loaded registers must not be read by
                  anything but the OR with the zero the wait block produces -- a copy inserted between the load and the wait
                  would copy a register whose data is still in flight (round 5 saw exactly that with a read-write operand).
  k_match_sweep   a wavefront's LDS-DMA pieces of the next tile must have landed before the barrier that releases the other
                  wavefronts onto them: an explicit `s_waitcnt vmcnt(0)` sits in front of the barriers of the tile loop."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "openpano_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _isa(src, tmp_path, kernel_prefix):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path / (src + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fopenmp", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
           "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = open(out).read().split("\n")
    bodies = {}
    cur = None
    for ln in lines:
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", ln)
        if m and kernel_prefix in m.group(1):
            cur = m.group(1); bodies[cur] = []
        elif cur is not None:
            bodies[cur].append(ln)
            if ln.strip().startswith("s_endpgm"):
                cur = None
    assert bodies, f"no kernel matching {kernel_prefix} in {src}"
    return bodies


def _asm_blocks(body):
    """-> list of (start index, [instructions]) of the inline-asm blocks"""
    blocks, i = [], 0
    while i < len(body):
        if "#ASMSTART" in body[i]:
            j = i + 1
            ins = []
            while "#ASMEND" not in body[j]:
                ins.append(body[j].strip()); j += 1
            blocks.append((i, ins)); i = j
        i += 1
    return blocks


def test_pyramid_rows_hand_waited_loads(tmp_path):
    (name, body), = _isa("pyramid.hip", tmp_path, "k_pyramid_rows").items()
    blocks = _asm_blocks(body)
    loads = [ins[0] for _, ins in blocks if len(ins) == 1 and ins[0].startswith("buffer_load_dword")]
    assert len(loads) == 2, loads
    regs = [re.match(r"buffer_load_dword (v\d+),", l).group(1) for l in loads]
    waits = [ins for _, ins in blocks if ins and ins[0].startswith("s_waitcnt vmcnt(")]
    assert sorted(w[0] for w in waits) == ["s_waitcnt vmcnt(0)", "s_waitcnt vmcnt(12)", "s_waitcnt vmcnt(6)"] or \
        {w[0] for w in waits} >= {"s_waitcnt vmcnt(0)", "s_waitcnt vmcnt(6)", "s_waitcnt vmcnt(12)"}, waits
    zregs = set()
    for w in waits:
        if w[0] in ("s_waitcnt vmcnt(6)", "s_waitcnt vmcnt(12)") or (w[0] == "s_waitcnt vmcnt(0)" and len(w) == 2):
            m = re.match(r"v_mov_b32 (v\d+), 0$", w[1])
            assert m, w
            zregs.add(m.group(1))
    assert len(zregs) == 1, zregs              # one register carries the zero out of every wait block
    z = zregs.pop()
    in_asm = set()
    for i, ins in blocks:
        in_asm.update(range(i, i + len(ins) + 2))
    for k, ln in enumerate(body):
        t = ln.strip()
        if not t or t.startswith((";", ".")) or k in in_asm:
            continue
        ops = re.split(r"[ ,]+", t)
        for r in regs:
            touched = any(re.fullmatch(re.escape(r), o) or re.fullmatch(r"v\[(\d+):(\d+)\]", o) and int(re.fullmatch(r"v\[(\d+):(\d+)\]", o).group(1)) <= int(r[1:]) <= int(re.fullmatch(r"v\[(\d+):(\d+)\]", o).group(2)) for o in ops[1:])
            if touched:
                # the only instruction allowed to name a loaded register: v_or_b32 dst, <zero of the wait block>, <loaded register>
                assert ops[0].startswith("v_or_b32") and z in ops[2:4] and ops[1] != r, (k, t)


def test_match_sweep_dma_wait_before_barrier(tmp_path):
    bodies = _isa("match.hip", tmp_path, "k_match_sweep")
    assert len(bodies) == 2
    for name, body in bodies.items():
        explicit = [i for i, ins in _asm_blocks(body) if ins == ["s_waitcnt vmcnt(0)"]]
        assert len(explicit) >= 3, (name, explicit)          # before the first tile, and in both halves of the unrolled tile loop
        for i in explicit:           # the next barrier follows with no VMEM instruction in between (register work may sit there)
            nxt = [l.strip() for l in body[i + 3: i + 60] if l.strip() and not l.strip().startswith(";")]
            assert any(l.startswith("s_barrier") for l in nxt), (name, nxt[:12])
            upto = next(k for k, l in enumerate(nxt) if l.startswith("s_barrier"))
            assert not any(l.startswith(("buffer_", "global_", "flat_", "scratch_")) for l in nxt[:upto]), (name, nxt[:upto])


def test_no_kernel_uses_scratch_memory():
    """Every kernel of the library keeps its state in registers and LDS: `ScratchSize [bytes/lane]` is 0 for all of them
    (round 5 shipped 16 / 40 B/lane in k_match_sweep and 12 B/lane in k_descriptor -- a scratch reload sat inside the
    candidate re-score loop and one in every keypoint of the descriptor kernel)."""
    import glob
    from concurrent.futures import ThreadPoolExecutor
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")

    def usage(src):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fopenmp", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
               "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", src, "-o", os.devnull]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        names = re.findall(r"Function Name: (\S+)", r.stderr)
        scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
        assert len(names) == len(scratch), src          # (a host-only TU has no kernels)
        return list(zip(names, scratch))

    with ThreadPoolExecutor(4) as ex:
        rows = [r for rs in ex.map(usage, sorted(glob.glob(os.path.join(CSRC, "*.hip")))) for r in rs]
    assert len(rows) >= 40
    assert not [r for r in rows if r[1] != 0], [r for r in rows if r[1] != 0]
