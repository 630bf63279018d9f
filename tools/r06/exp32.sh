#!/bin/bash
timeout 1200 python -m pytest tests/test_gpu_sort_project.py tests/test_gpu_round6.py -q -x -m gpu -k "sort or order" 2>&1 | tail -3
python - <<'PY'
import time, torch, ctypes
from vinum_amd import _lib as L, ops
from vinum_amd.device import DeviceColumn
n = 1_000_000_000
g = torch.Generator(device="cuda"); g.manual_seed(3)
v = torch.randn(n, device="cuda", dtype=torch.float64, generator=g)
sv = torch.sort(v).values
for name, col, order in (("random desc", v, L.DESC), ("sorted, asked ascending", sv, L.ASC), ("sorted, asked descending (no ties)", sv, L.DESC)):
    c = DeviceColumn.from_torch(col)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        idx = ops.sort_indices([c], [order])
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    buf = ctypes.create_string_buffer(300); L.lib().vnm_route_last(buf, 300)
    print(f"{name}: {ms:.1f} ms  {buf.value.decode()[:60]}")
    del idx
PY
