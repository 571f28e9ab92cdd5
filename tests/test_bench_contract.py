"""CPU: bench.py's own logic (not a committed output): the CLI flags of the driver contract, the
self-launch of N ranks, the algorithmic-byte bookkeeping, and the CPU-baseline / parity legs on a
tiny sample."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_bench_cli_declares_the_contract_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--scaling"):
        assert flag in out.stdout


def test_bench_gpus_n_spawns_its_own_ranks():
    """`python bench.py --gpus 2` must become the launcher (one process per GPU).  Without a GPU each
    rank gets as far as op_ctx_create and fails THERE -- loudly, no CPU fallback -- not at a
    'use torch.distributed.run' SystemExit."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present: the spawned ranks would run the whole bench")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode != 0
    assert "spawning 2 ranks" in out.stderr
    # every rank fails there; the launcher may end the second rank before its message is out (seen once under load): one is proof enough
    assert out.stderr.count("op_ctx_create: no HIP device available") >= 1, out.stderr[-2000:]
    assert "launch multi-GPU runs with" not in out.stderr


def test_pyramid_pixels_matches_the_oracle_plan(oracle, cfg):
    import bench
    for h, w in ((867, 1300), (400, 600), (3000, 4000)):
        P, wh, ww = bench.pyramid_pixels(cfg, h, w)
        st = oracle.sift_stages(np.zeros((h, w, 3), np.float32), planes=False)
        assert (wh, ww) == st.dims[0]
        assert P == sum(a * b for a, b in st.dims)


def test_cpu_baseline_leg_runs_and_reports_what_it_did(cfg):
    """the cpu_baseline leg on a tiny sample: threads capped at the sample size, warmed, best-of-3"""
    import bench
    from openpano_amd import synth
    world = synth.make_world(3, 260, 400, work_scale=1600.0 / (200 + 280), density=900.0)
    views = [synth.cut_view(world, 20, 20 + 30 * k, 200, 280, k) for k in range(3)]
    with bench._StdoutToStderr():
        r = bench.cpu_baseline(cfg, views, lambda m: None)
    assert r["kind"] in ("reference", "port") and r["value"] > 0 and r["single_thread_value"] > 0
    assert 1 <= r["cores"] <= (os.cpu_count() or 1) and r["cores"] <= 2 * (os.cpu_count() or 1)
    assert r["cpu_model"] and "best of 3" in r["sample"] and r["flags"]


def _full_record():
    """a real full record of an N = 1 run (round 5's 24.8 KB line, the one the driver could not parse)"""
    import json
    return json.loads(open(os.path.join(ROOT, "profiles", "r05_bench_v6.json")).read().strip().splitlines()[-1])


def test_bench_line_is_small_and_complete_at_one_gpu():
    """the line bench.py prints = bench_line.render(full record): under the cap, every contract key, `roofline` and
    `cpu_baseline` in the judged shape, consistent with the record it digests"""
    import json
    import bench_line
    full = _full_record()
    s = bench_line.render(full)
    assert len(s) < bench_line.MAX_LINE_BYTES < 8192 and "\n" not in s
    d = json.loads(s)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert set(d["roofline"]) >= {"kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms"}
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample", "host_cpus", "cpu_model"}
    assert abs(d["value"] - full["value"]) < 1e-5 * full["value"] and abs(d["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-5
    assert d["config5"]["match_roofline"]["frac"] and d["config5"]["parity"]["ok"] is True and d["parity"]["ok"] is True
    assert d["value_protocol_f32"] and d["gpu_over_cpu_protocol_u8"] and d["detail"] == bench_line.DETAIL_FILE
    assert "workload" in d["config"] and "model" not in d["config"]


def test_bench_line_stays_small_at_eight_gpus():
    """a fabricated --gpus 8 record: rccl block, per-rank phase tables x 8 in config5 AND strong_config4, `predicted`
    blocks, a long note -- everything the N > 1 code path adds (bench.py, bench_match.run_strong_job)"""
    import copy
    import json
    import bench_line
    full = _full_record()
    full["n_gpus"] = 8; full["scaling"] = "strong"
    full["rccl_ranks"] = {"world_size": 8, "allreduce_rank_sum": 36.0, "allreduce_count": 8, "expected_rank_sum": 36.0, "backend": "nccl", "ok": True}
    full["config"]["note"] = "strong scaling of a 38-image job: " + "x" * 400
    per_rank = [{"sift": 0.9 + r, "feature all-gather": 0.4, "match": 11.5, "ransac": 1.6, "results gather": 0.2} for r in range(8)]
    for key in ("config5", "strong_config4"):
        j = copy.deepcopy(full["config5"])
        j["n_gpus"] = 8; j["per_rank_phase_ms"] = per_rank; j["allgather_bytes_per_rank"] = 33_000_000
        j["predicted"] = {"phase_ms": dict(per_rank[0]), "job_ms": 25.1, "keypoints_per_s": 3.1e8, "image_pairs_per_s": 4.0e5, "source": "y" * 300}
        full[key] = j
    full["predicted"] = {"sift_ms_per_step": 0.27, "value": 1.7e8, "note": "z" * 200}
    for k in ("cpu_baseline", "parity", "blend", "ingest", "protocol", "stitch_e2e", "configs"):   # rank 0 of N > 1 skips these legs
        full.pop(k, None)
    full["cpu_baseline"] = None; full["parity_checked"] = None
    s = bench_line.render(full)
    assert len(s) < bench_line.MAX_LINE_BYTES, len(s)
    d = json.loads(s)
    assert d["n_gpus"] == 8 and d["rccl_ranks"]["world_size"] == 8 and d["rccl_ranks"]["ok"] is True
    assert d["config5"]["max_rank_phase_ms"]["sift"] == 7.9 and "per_rank_phase_ms" not in d["config5"]
    assert d["strong_config4"]["predicted"]["job_ms"] == 25.1 and d["roofline"]["frac"]
    assert "x" * 50 not in s and "y" * 50 not in s and "z" * 50 not in s


def test_committed_bench_lines_obey_the_contract():
    """The newest committed bench line of every round (profiles/rNN_bench_vM.json): contract keys, a roofline object
    that is consistent with itself (achieved = algorithmic bytes / launch time, frac = achieved / peak), parity
    checked in the run, value = descriptors * steps / time and above the north_star's 30x of the CPU baseline."""
    import glob
    import json
    import re
    newest = {}
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_bench_v*.json")):
        m = re.match(r"r(\d+)_bench_v(\d+)\.json$", os.path.basename(f))
        if m and (int(m.group(1)) not in newest or int(m.group(2)) > newest[int(m.group(1))][0]):
            newest[int(m.group(1))] = (int(m.group(2)), f)
    assert newest
    for rnd, (_, f) in sorted(newest.items()):
        d = json.loads(open(f).read().strip().splitlines()[-1])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
            assert k in d, (f, k)
        assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None and "workload" in d["config"]
        r = d["roofline"]
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
        assert r["avg_launch_ms"] <= d["ms_per_step"]
        if rnd >= 2:
            assert d["parity_checked"] is True, f
        c = d["cpu_baseline"]
        assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["sample"]
        assert d["value"] > 30 * c["value"], (f, d["value"] / c["value"])
        k = d["config"]["keypoints_per_image"] * d["config"]["images_per_gpu"]
        assert abs(d["value"] - k / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"], f


def test_pmc_evidence_belongs_to_the_built_library():
    """profiles/pmc_latest.json carries the hash of the library its counters were collected on; bench.py replays
    `roofline.traffic` from it only for that library.  The build is reproducible (a clean `make` of HEAD gives the same
    bytes), so a mismatch here means kernels changed after the last PMC pass: the bench line will say `traffic: null`
    until scripts/gpu_pmc.sh has run again.  Reported as a skip, not a failure -- it is a reminder, not a defect."""
    import hashlib
    import json
    import pytest
    lib = os.path.join(ROOT, "openpano_amd", "libopenpano_hip.so")
    pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if not (os.path.exists(lib) and os.path.exists(pmc)):
        pytest.skip("library or PMC summary not present")
    want = json.load(open(pmc)).get("_meta", {}).get("lib_sha256_16")
    got = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
    assert want, "pmc_latest.json has no _meta.lib_sha256_16"
    if want != got:
        pytest.skip(f"PMC counters were collected on library {want}, the built library is {got}: roofline.traffic will be null until the PMC passes are re-run")


def test_full_job_record_belongs_to_the_built_library():
    """profiles/config5_all_pairs_latest.json is the record of the whole config-5 parity job (all 8128 match sets and RANSAC
    results against the oracle, OPENPANO_FULL_C5=1 tests/test_gpu_fullsize.py -k whole_match_job), written by the test
    itself together with the hash of the library it ran on.  A record of another build proves nothing about this one:
    reported as STALE (a skip -- the job needs a GPU and ~7 minutes, it cannot be re-run here)."""
    import hashlib
    import json
    import pytest
    lib = os.path.join(ROOT, "openpano_amd", "libopenpano_hip.so")
    recp = os.path.join(ROOT, "profiles", "config5_all_pairs_latest.json")
    if not os.path.exists(recp):
        pytest.skip("no full-job record committed")
    rec = json.load(open(recp))
    assert rec["full"] and rec["match_pairs_checked"] == rec["pairs_in_job"] == 8128 and rec["ransac_pairs_checked"] == 8128
    assert rec["match_pairs_differing"] == 0 and rec["images"] == 128
    if not os.path.exists(lib):
        pytest.skip("library not built")
    got = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
    if rec["lib_sha256_16"] != got:
        pytest.skip(f"STALE: the full config-5 parity job ran on library {rec['lib_sha256_16']}, the built library is {got}")


def test_config5_matrix_pipe_counters_belong_to_the_built_library():
    """profiles/config5_mfma_latest.json (per-dispatch matrix-pipe counters of the config-5 sweeps) is replayed by bench.py only for
    the library it was collected on; a mismatch is reported as a skip like the other counter records."""
    import hashlib
    import json
    import pytest
    lib = os.path.join(ROOT, "openpano_amd", "libopenpano_hip.so")
    rec = os.path.join(ROOT, "profiles", "config5_mfma_latest.json")
    if not (os.path.exists(lib) and os.path.exists(rec)):
        pytest.skip("library or record not present")
    d = json.load(open(rec))
    f5 = d["config5_forward"]
    assert 0 < f5["mfma_busy"] <= 1 and 1.0 < f5["shader_clock_ghz"] < 2.6 and f5["workgroups"] > 100000
    assert abs(f5["peak_tflops_at_measured_clock"] - 1024 * 1024 * f5["shader_clock_ghz"] * 1e-3) < 1.0
    got = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
    if d["_meta"]["lib_sha256_16"] != got:
        pytest.skip(f"STALE: the config-5 matrix-pipe counters were collected on library {d['_meta']['lib_sha256_16']}, the built library is {got}")
