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
    assert out.stderr.count("op_ctx_create: no HIP device available") >= 2, out.stderr[-2000:]
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
