#!/usr/bin/env python3
"""GPU probe: one pair of the config-5 job under the microscope (exact fp32 distances in the reference's order, float64
scores, margins).  usage: match_case.py <img_i> <img_j> <idx_i> <idx_j> [variant]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 5: os.environ["OPENPANO_HIP_LIB"] = os.path.join(ROOT, "openpano_amd", "variants", f"libopenpano_hip_{sys.argv[5]}.so")
import numpy as np, torch
from openpano_amd import hip, synth
from openpano_amd.config import PanoConfig
i, j, ai, bj = map(int, sys.argv[1:5])
dev = torch.device("cuda:0"); ctx = hip.Context(0); cfg = PanoConfig()
imgs = synth.config5_views([i, j], dev); torch.cuda.synchronize()
f = hip.SiftCall(ctx, cfg, [(t.data_ptr(), 3000, 4000, "u8") for t in imgs])()
got = hip.match_pairs(ctx, cfg, f, [(0, 1)])[0]
D = [f.get(0)[0], f.get(1)[0]]
print("K", len(D[0]), len(D[1]), "matches", len(got), "has", (ai, bj), any((g[0] == ai and g[1] == bj) for g in got))


def dist32(x, Y):     # feature/dist.cc: four stride-4 fp32 partial sums walked in order, (v0+v1)+(v2+v3)
    v = np.zeros((4, len(Y)), np.float32)
    for t in range(32):
        for k in range(4):
            d = (x[4 * t + k] - Y[:, 4 * t + k]).astype(np.float32)
            v[k] = (v[k] + (d * d).astype(np.float32)).astype(np.float32)
    return ((v[0] + v[1]).astype(np.float32) + (v[2] + v[3]).astype(np.float32)).astype(np.float32)


def look(tag, x, Y, skip=-1):
    d = dist32(x, Y); o = np.argsort(d, kind="stable")
    o = [c for c in o if c != skip][:6]
    s64 = Y.astype(np.float64) @ x.astype(np.float64) - 0.5 * (Y.astype(np.float64) ** 2).sum(1)
    E = 8e-5 * (float((x.astype(np.float64) ** 2).sum()) + float((Y.astype(np.float64) ** 2).sum(1).max()))
    print(tag, "E(8e-5) = %.2f" % E)
    for c in o: print("    col %5d  d2 %.3f  true score %.3f  (2nd best score - this) %.3f" % (c, d[c], s64[c], s64[o[1]] - s64[c]))
    return d, o


small, big = (0, 1) if len(D[0]) <= len(D[1]) else (1, 0)
qa, qb = (ai, bj) if small == 0 else (bj, ai)          # row in the smaller (query) set, column in the other
print("query set = image", (i, j)[small], "row", qa, " other column", qb)
d, o = look("forward: row vs all columns", D[small][qa], D[big])
rr = np.float32(cfg.MATCH_REJECT_NEXT_RATIO) ** 2 if hasattr(cfg, "MATCH_REJECT_NEXT_RATIO") else None
print("   ratio^2", rr, " mn", d[o[0]], "next", d[o[1]], " mn > rr*next ?", (d[o[0]] > np.float32(rr) * d[o[1]]) if rr is not None else "")
d2, o2 = look("reverse: best column vs all rows but the query", D[big][o[0]], D[small], skip=qa)
print("   fmn", d[o[0]], "rev next", d2[o2[0]], " fmn > rr*next ?", (d[o[0]] > np.float32(rr) * d2[o2[0]]) if rr is not None else "")
