#!/usr/bin/env python3
"""GPU probe: the hooked literal CLI (oracle/_ref/image-stitching-hipfast) on natural BASELINE configs, wall time per run.

    [CLI_REPS=n] [CLI_TIMEOUT=s] python scripts/cli_probe.py [2] [3] [4]
"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import natural  # noqa: E402
from PIL import Image  # noqa: E402
from openpano_amd.config import DEFAULTS  # noqa: E402

binary = os.path.join(ROOT, "oracle", "_ref", "image-stitching-hipfast")
for key in (sys.argv[1:] or ["2", "4"]):
    cfgk = int(key[0])
    over = dict(ORDERED_INPUT=1) if cfgk in (2, 3) else dict()
    d = tempfile.mkdtemp()
    files = []
    for k, v in enumerate(natural.config_views(cfgk)):
        p = os.path.join(d, f"{k:02d}.png"); Image.fromarray(v).save(p, compress_level=1); files.append(p)
    vals = dict(DEFAULTS); vals.update(over); vals["LAZY_READ"] = 0
    with open(os.path.join(d, "config.cfg"), "w") as f:
        for k, v in vals.items():
            f.write(f"{k} {v}\n")
    env = dict(os.environ); env["OPENPANO_TEST_SEED"] = "38"; env["OMP_NUM_THREADS"] = os.environ.get("CLI_THREADS", "32")
    for rep in range(int(os.environ.get("CLI_REPS", "1"))):
        t0 = time.perf_counter()
        try:
            r = subprocess.run([binary] + files, capture_output=True, text=True, env=env, timeout=int(os.environ.get("CLI_TIMEOUT", "150")), cwd=d)
            out = r.stdout + r.stderr; rc = r.returncode
        except subprocess.TimeoutExpired as e:
            out = (e.stdout or b"").decode(errors="replace") + (e.stderr or b"").decode(errors="replace"); rc = "TIMEOUT"
        print(f"== config {key} rep {rep}: rc {rc} wall {time.perf_counter() - t0:.2f} s  " +
              "  ".join(ln.strip() for ln in out.splitlines() if "milliseconds" in ln), flush=True)
        if rc != 0:
            print("last lines:", out.splitlines()[-12:])
