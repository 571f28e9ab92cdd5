#!/usr/bin/env python3
"""GPU probe: all-pairs matches of the config-5 job with the product library vs a variant library (run in a subprocess);
every pair on which they differ is decided by the exact-matcher oracle.  usage: match_diff.py <variant-name> [n_images]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def run(out, n):
    import torch
    from openpano_amd import hip, synth
    from openpano_amd.config import PanoConfig
    dev = torch.device("cuda:0"); ctx = hip.Context(0); cfg = PanoConfig()
    imgs = synth.config5_views(list(range(n)), dev)
    torch.cuda.synchronize()          # the C-ABI works on its own stream
    f = hip.SiftCall(ctx, cfg, [(t.data_ptr(), 3000, 4000, "u8") for t in imgs])()
    pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
    got = hip.match_pairs(ctx, cfg, f, pairs)
    np.savez(out, pairs=np.array(pairs), lens=np.array([len(g) for g in got]), flat=np.concatenate([g.reshape(-1, 2) for g in got]) if got else np.zeros((0, 2), int))
    return f, pairs, got


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        run(sys.argv[2], int(sys.argv[3])); sys.exit(0)
    variant, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 128
    tmp = os.path.join(tempfile.gettempdir(), "match_diff_child.npz")
    env = dict(os.environ, OPENPANO_HIP_LIB=os.path.join(ROOT, "openpano_amd", "variants", f"libopenpano_hip_{variant}.so"))
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", tmp, str(n)], env=env, check=True)
    f, pairs, got = run(os.path.join(tempfile.gettempdir(), "match_diff_parent.npz"), n)
    z = np.load(tmp); off = np.concatenate([[0], np.cumsum(z["lens"])])
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from checkers import Oracle
    from openpano_amd.config import PanoConfig
    orc = Oracle(PanoConfig())
    ndiff = 0
    for k, (i, j) in enumerate(pairs):
        other = z["flat"][off[k]:off[k + 1]]
        if np.array_equal(other, got[k]): continue
        ndiff += 1
        if ndiff > 6: continue
        want = orc.match_exact(f.get(i)[0], f.get(j)[0])
        print(f"pair ({i},{j}): product {len(got[k])} matches ({'==' if np.array_equal(got[k], want) else '!='} oracle), {variant} {len(other)} ({'==' if np.array_equal(other, want) else '!='} oracle), oracle {len(want)}")
        a, b, c = set(map(tuple, got[k])), set(map(tuple, other)), set(map(tuple, want))
        print("   product - oracle", sorted(a - c)[:5], " oracle - product", sorted(c - a)[:5], f" {variant} - oracle", sorted(b - c)[:5], f" oracle - {variant}", sorted(c - b)[:5])
    print(f"{len(pairs)} pairs, {sum(len(g) for g in got)} matches (product), {int(z['lens'].sum())} ({variant}); differing pairs: {ndiff}")
