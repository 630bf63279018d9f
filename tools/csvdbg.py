import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vinum_amd import _lib as L
lib = L.lib()
for s in ["7e4", "7E4", "70000", "7.0e4", "7e+4", "7e-4", "0.5", "12", "1.5e0", "7e"]:
    text = ("x\n" + s + "\n").encode()
    out = (L.DCol * 1)(); n = ctypes.c_int64(0); fb = (ctypes.c_int * 3)()
    L.check(lib.vnm_csv_parse_block(text, len(text), 1, ord(","), 1, 1, (ctypes.c_int * 1)(0), (ctypes.c_int * 1)(L.F64), out, ctypes.byref(n), fb, None))
    v = np.empty(1, np.float64); L.check(lib.vnm_memcpy_d2h(v.ctypes.data, out[0].values, 8))
    vb = np.empty(8, np.uint8); L.check(lib.vnm_memcpy_d2h(vb.ctypes.data, out[0].validity, 8))
    print(repr(s), "rows", n.value, "value", v[0], "valid", vb[0] & 1, "fb", list(fb))
