"""Full sort of N fp64 keys (ORDER BY v DESC, int64 row ids + rebuilt keys out): clean N(11, 9) keys, the same with 0.1 % of one
value and 0.1 % NaN (heavy codes: side list), and with VNM_SORT_NO_SAMPLE=1 (the LSD passes).  python tools/sortbench.py [N]"""
import os, sys, time
sys.path.insert(0, ".")
import torch
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
g = torch.Generator(device="cuda"); g.manual_seed(2)
v = torch.randn(n, device="cuda", dtype=torch.float64, generator=g) * 3.0 + 11.0


def run(tag):
    col = DeviceColumn.from_torch(v)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        idx, key = ops.sort_indices_keyed([col], [L.DESC])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        del idx, key
    print(f"{tag:42s} {dt:7.2f} ms", flush=True)


run("clean keys, sample sort")
os.environ["VNM_SORT_NO_SAMPLE"] = "1"; run("clean keys, LSD sort"); del os.environ["VNM_SORT_NO_SAMPLE"]
v[::1000] = 12.5
v[7::1000] = float("nan")
run("0.1 % one value + 0.1 % NaN, sample sort")
v[::3] = 7.0
run("a third of the rows one value, sample sort")
os.environ["VNM_SORT_NO_SAMPLE"] = "1"; run("the same, LSD sort")
# ---- NULL rows: 0.1 % NULL (configs[4]'s variant) -- a class of the side list -- against the LSD passes
import pyarrow as pa
del os.environ["VNM_SORT_NO_SAMPLE"]
v2 = torch.randn(n, device="cuda", dtype=torch.float64, generator=g) * 3.0 + 11.0
bits = torch.full(((n + 7) // 8,), 255, dtype=torch.uint8, device="cuda")
bits[::125] = 254          # one NULL per 1000 rows
v = v2


def run_nulls(tag):
    col = DeviceColumn.from_torch(v, validity=bits)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        idx, key = ops.sort_indices_keyed([col], [L.DESC])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        del idx, key
    print(f"{tag:42s} {dt:7.2f} ms", flush=True)


run_nulls("0.1 % NULL, sample sort")
os.environ["VNM_SORT_NO_SAMPLE"] = "1"; run_nulls("0.1 % NULL, LSD sort")
