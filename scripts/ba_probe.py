#!/usr/bin/env python3
"""Host probe: wall time of libpano_host.so's camera estimation / bundle adjustment on a config-4 sized
pairwise table for several OpenMP team sizes (PANO_BA_THREADS); run on the box whose cores matter."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from camera_util import host_impl, rotating_camera_scene
    h = host_impl()
    shapes, table, _ = rotating_camera_scene(5, n=38, rows=2, w=1300, h=867, focal=1235., step_deg=14.0, npts=160)
    best = 1e9
    for _ in range(3):
        t = time.time(); h.estimate(shapes, table); best = min(best, time.time() - t)
    print("threads", os.environ.get("PANO_BA_THREADS"), "pairs", len(table) // 2, "matches", sum(len(t[4]) for t in table) // 2, "best of 3: %.1f ms" % (best * 1e3))
else:
    for t in sys.argv[1:] or ["1", "4", "8", "16", "32", "64"]:
        env = dict(os.environ, PANO_BA_THREADS=t, PANO_BA_PROFILE="1")
        subprocess.run([sys.executable, __file__, "child"], env=env)
