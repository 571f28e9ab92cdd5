#!/usr/bin/env python3
"""GPU probe: config-4 SIFT step wall time vs loop length and with / without the per-stage HIP events (what bench.py's
timed region pays for measuring its roofline live)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from openpano_amd import hip, synth
from openpano_amd.config import PanoConfig
cfg = PanoConfig(); dev = torch.device("cuda", 0)
views = synth.image_set(38, 867, 1300, seed=38, overlap=0.45, rows=2, shuffle=True)
d = [torch.from_numpy(v).to(dev) for v in views]; torch.cuda.synchronize()
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
ctx = hip.Context(0, stream.cuda_stream)
call = hip.SiftCall(ctx, cfg, [(t.data_ptr(), 867, 1300) for t in d])
f = call()
def loop(steps, prof):
    global f
    ctx.set_profiling(prof); ctx.profile_reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        f.free(); f = call()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / steps * 1e3
    ctx.set_profiling(False)
    return t
for steps in (3, 10, 20, 50, 200, 1000):
    print("steps %4d  events on %.4f ms  events off %.4f ms" % (steps, loop(steps, True), loop(steps, False)), flush=True)
print("again: 20 steps events on %.4f  off %.4f ; 200 steps on %.4f off %.4f" % (loop(20, True), loop(20, False), loop(200, True), loop(200, False)))
