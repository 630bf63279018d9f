"""Full sort of fp64 keys at several sizes: sample sort (second-level fan-out scaled to n) against the LSD passes, best of 4.
usage: sortsizes.py"""
import os, sys, time
sys.path.insert(0, ".")
import torch
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
g = torch.Generator(device="cuda"); g.manual_seed(2)
for n in (4_000_000, 8_400_000, 17_000_000, 40_000_000, 70_000_000, 130_000_000, 260_000_000, 520_000_000, 1_000_000_000):
    v = torch.randn(n, device="cuda", dtype=torch.float64, generator=g)
    col = DeviceColumn.from_torch(v)
    out = []
    for env in (None, "1"):
        if env: os.environ["VNM_SORT_NO_SAMPLE"] = env
        else: os.environ.pop("VNM_SORT_NO_SAMPLE", None)
        best = 1e9
        for rep in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            idx = ops.sort_indices([col], [L.ASC])
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e3)
            del idx
        out.append(best)
    print(f"n={n:>11d}  default {out[0]:7.2f} ms   lsd {out[1]:7.2f} ms", flush=True)
    del v, col
