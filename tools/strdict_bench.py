"""Dictionary-encoding a string key column: the device route (vnm_strdict_encode through vinum_lib.KeyDictionary; host buffers in,
int32 codes out, PCIe included) against the host route it replaced (Arrow dictionary_encode + NumPy merge into the running
dictionary).  usage: strdict_bench.py [N]"""
import sys, time, ctypes
sys.path.insert(0, ".")
import numpy as np, pyarrow as pa
from vinum_amd import _lib as L
from vinum_amd.vinum_lib import KeyDictionary
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
rng = np.random.default_rng(0)
lib = L.lib()
for G in (100, 100_000, 5_000_000):
    vals = pa.array([f"city_{i:07d}" for i in range(G)])
    col = vals.take(pa.array(rng.integers(0, G, n)))
    out = []
    for route in ("device", "host"):
        d = KeyDictionary(pa.string())
        if route == "host":
            d._device = False
            sub = col.slice(0, n // 10)          # (the host route on a tenth of the rows: it takes seconds)
        else:
            sub = col
        ts = []
        for rep in range(2):                      # first call builds the dictionary, second finds every value in it
            lib.vnm_set_profiling(1)
            t0 = time.perf_counter(); c = d.encode(sub); ts.append(time.perf_counter() - t0)
            ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
            lib.vnm_profile_query(b"strdict_encode", ctypes.byref(ms), ctypes.byref(cnt))
            lib.vnm_set_profiling(0)
        out.append((route, len(sub), ts, ms.value))
    for route, m, ts, kms in out:
        print(f"G={G:>8d} {route:6s}: first batch {m / ts[0] / 1e6:8.1f} Mrows/s, known values {m / ts[1] / 1e6:8.1f} Mrows/s"
              + (f"  (kernel {kms:.1f} ms for {m} rows, {col.nbytes / 1e6:.0f} MB)" if route == "device" else ""), flush=True)
