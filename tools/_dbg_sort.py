import ctypes, os, sys
sys.path.insert(0, ".")
import numpy as np, torch, pyarrow as pa
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
os.environ["VNM_SSORT_MIN_ROWS"] = "1000"
os.environ["VNM_SORT_TRACE"] = "1"
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000
torch.manual_seed(0)
v = torch.randn(n, device="cuda", dtype=torch.float64) * 3 + 11
col = DeviceColumn.from_torch(v)
out = torch.full((n,), -1, dtype=torch.int64, device="cuda")
key = torch.full((n,), -1, dtype=torch.int64, device="cuda")
wrote = ctypes.c_int(0)
od = (ctypes.c_int * 1)(L.ASC)
import time
for rep in range(3):
    out.fill_(-1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    L.check(L.lib().vnm_sort_indices_keyed(1, ops.dcol_array([col]), od, n, 0, out.data_ptr(), key.data_ptr(), ctypes.byref(wrote), None))
    torch.cuda.synchronize(); print("ms", (time.perf_counter() - t0) * 1e3)
print("unwritten", int((out == -1).sum()), "wrote key", wrote.value)
bad_range = (out < 0) | (out >= n)
print("out of range values", int(bad_range.sum()))
if int(bad_range.sum()):
    pos = torch.nonzero(bad_range).flatten()
    print("positions", pos[:8].tolist(), pos[-3:].tolist(), "values", out[pos[:4]].tolist(), "key there", key[pos[:4]].tolist())
    print("as double", out[pos[:4]].view(torch.float64).tolist())
    good = out[~bad_range]
    cnt = torch.bincount(good, minlength=n)
    missing = torch.nonzero(cnt == 0).flatten()
    dup = torch.nonzero(cnt > 1).flatten()
    print("missing rows", len(missing), missing[:10].tolist(), "dup rows", len(dup), dup[:10].tolist())
    print("missing rows mod 2048", (missing[:20] % 2048).tolist())
    print("dup rows mod 2048", (dup[:20] % 2048).tolist())
    ref = torch.argsort(v, stable=True)
    rank = torch.empty_like(ref); rank[ref] = torch.arange(n, device="cuda")
    print("ranks of missing", rank[missing[:10]].tolist())
    print("ranks of dup", rank[dup[:10]].tolist())
    sys.exit(0)
ref = torch.argsort(v, stable=True)
print("mismatch", int((out != ref).sum()))
bad = torch.nonzero(out != ref)[:10].flatten().tolist()
print("first bad positions", bad, out[bad].tolist() if bad else None, ref[bad].tolist() if bad else None)
ok = out >= 0
if int(ok.sum()):
    vv = v[out[ok]]
    print("sorted among written?", bool((vv[1:] >= vv[:-1]).all()))
u, c = torch.unique(out[ok], return_counts=True)
print("distinct written", len(u), "dups", int((c > 1).sum()))
