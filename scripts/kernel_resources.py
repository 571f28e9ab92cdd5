#!/usr/bin/env python3
"""Compile every csrc/*.hip with -Rpass-analysis=kernel-resource-usage and print one line per kernel:
VGPRs, AGPRs, SGPRs, scratch bytes/lane, occupancy (waves/SIMD), LDS bytes.  No GPU needed (hipcc cross-compiles)."""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "openpano_amd", "csrc")
files = sys.argv[1:] or sorted(glob.glob(os.path.join(src, "*.hip")))
for f in files:
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fopenmp",
           f"-I{ROOT}/include", f"-I{src}", "-Rpass-analysis=kernel-resource-usage", "-c", f, "-o", "/dev/null"] + [a for a in os.environ.get("EXTRA", "").split() if a]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = {}
    for line in err.splitlines():
        m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
        elif ":" in t:
            k, v = t.split(":", 1); cur[k.strip()] = v.strip()
            if k.strip().startswith("LDS Size"):
                name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
                name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
                print(f"{os.path.basename(f):16s} {name:40s} VGPR {cur.get('VGPRs'):>4s} AGPR {cur.get('AGPRs'):>3s} SGPR {cur.get('TotalSGPRs', cur.get('SGPRs')):>4s} "
                      f"scratch {cur.get('ScratchSize [bytes/lane]'):>4s} occ {cur.get('Occupancy [waves/SIMD]'):>2s} LDS {cur.get('LDS Size [bytes/block]'):>6s}")
