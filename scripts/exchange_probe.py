#!/usr/bin/env python3
"""GPU probe: where the time of the one-rank exchange (OPENPANO_FORCE_DIST path, nccl backend) goes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from openpano_amd.distributed import allgatherv_features
dev = torch.device("cuda", 0)
for n, k in ((38, 1220), (128, 4050)):
    counts = [k] * n
    desc = torch.rand((n * k, 128), device=dev); coor = torch.rand((n * k, 2), dtype=torch.float64, device=dev)
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g = allgatherv_features(desc, coor, counts, n)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f"n={n} K={k} rep {rep}: exchange {1e3 * (t1 - t0):.3f} ms ({desc.numel() * 4 / 1e6:.0f} MB)")
    # pieces
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hdr = torch.zeros(n, dtype=torch.int64, device=dev); out = torch.empty(n, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(out, hdr); h = out.cpu()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    a = torch.empty((n * k, 128), device=dev); b = torch.empty((n * k, 2), dtype=torch.float64, device=dev)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    a.copy_(desc); b.copy_(coor)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"   header collective + D2H {1e3 * (t1 - t0):.3f} ms, allocation {1e3 * (t2 - t1):.3f} ms, own-slice copy {1e3 * (t3 - t2):.3f} ms")
dist.destroy_process_group()
