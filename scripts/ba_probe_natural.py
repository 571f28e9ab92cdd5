"""Host probe like ba_probe.py on a match table of BASELINE config 4's natural-texture size (~49 k inlier matches): PANO_BA_THREADS=N python scripts/ba_probe_natural.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from camera_util import host_impl, rotating_camera_scene
h = host_impl()
shapes, table, _ = rotating_camera_scene(5, n=38, rows=2, w=1300, h=867, focal=1235., step_deg=14.0, npts=480)
best = 1e9
for _ in range(3):
    t = time.time(); h.estimate(shapes, table); best = min(best, time.time() - t)
print("threads", os.environ.get("PANO_BA_THREADS"), "pairs", len(table)//2, "matches", sum(len(t[4]) for t in table)//2, "best of 3: %.1f ms" % (best*1e3))
