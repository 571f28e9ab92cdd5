#!/usr/bin/env python3
"""GPU probe: wall time of successive op_match_pairs calls on a config-5-shaped job (32 of the 4000x3000 images, 496
pairs) -- first calls vs steady state, without and with the per-stage HIP events, and with the result lists read back."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from openpano_amd import hip, synth
from openpano_amd.config import PanoConfig
cfg = PanoConfig(); dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
imgs = synth.config5_views(range(n), dev); torch.cuda.synchronize()
ctx = hip.Context(0)
f = hip.SiftCall(ctx, cfg, [(t.data_ptr(), 3000, 4000, "u8") for t in imgs])()
pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
def call(read):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mh = hip.match_pairs_handle(ctx, cfg, f, pairs)
    ctx.sync(); t1 = time.perf_counter()
    nm = sum(len(x) for x in mh.lists()) if read else -1
    t2 = time.perf_counter(); mh.free()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3, nm
for k in range(6):
    print("events off  call %d: match + sync %.3f ms, lists to host %.3f ms (%d matches)" % ((k,) + call(k % 2 == 1)), flush=True)
ctx.set_profiling(True)
for k in range(3):
    print("events on   call %d: match + sync %.3f ms, lists to host %.3f ms (%d matches)" % ((k,) + call(True)), flush=True)
print({k: round(v[0] / 3, 3) for k, v in ctx.profile().items() if k.startswith("matcher")})
ctx.set_profiling(False)
for k in range(3):
    print("events off  call %d: match + sync %.3f ms, lists to host %.3f ms (%d matches)" % ((k,) + call(False)), flush=True)
