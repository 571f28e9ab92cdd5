"""CPU: the bench line committed under profiles/ obeys the driver's contract (one JSON object with
the agreed keys, a consistent roofline and CPU baseline) and bench.py parses / declares its flags."""
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest():
    fs = glob.glob(os.path.join(ROOT, "profiles", "r*_bench_v*.json"))
    assert fs, "no committed bench line under profiles/"
    return max(fs, key=lambda f: (int(re.search(r"r(\d+)_", os.path.basename(f)).group(1)), int(re.search(r"_v(\d+)", f).group(1))))


def test_committed_bench_line_obeys_contract():
    d = json.load(open(_latest()))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["unit"] and d["higher_is_better"] is True
    for k in ("value", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert r["traffic"] is None or r["traffic"] > 0.5 * r["algorithmic_bytes_per_launch"]
    # achieved = algorithmic bytes / live kernel time
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"] and c["unit"] == d["unit"]
    # value = descriptors of the batch / time per batch
    k = d["config"]["keypoints_per_image"] * d["config"]["images_per_gpu"]
    assert abs(d["value"] - k / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["value"] > 30 * c["value"]                      # >= 30x the reference CPU on config 4


def test_bench_cli_declares_the_contract_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout
