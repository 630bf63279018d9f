#!/usr/bin/env python3
"""Golden vectors for the two float questions VERDICT r01 raised, produced by the REAL reference
(oracle/_ref/libvinum_ref.so = the reference's own C++ aggregate compiled where it lies; build container only).

  fsum_ref.arrow  (inputs: float_cases.fsum_table(), regenerated from the seed; checksum in float_cases.json)
      NON-quantised float64 / float32 inputs (lognormal, normal, a wide-dynamic-range mix, a cancelling column) in
      groups of 1 ... ~1e5 rows, and what the reference's sequential `sum += x` (agg_funcs.h:294-305, AVG :455-467,
      519-522) returns for them.  The test measures the ULP distance of the HIP path to this AND to the exactly
      rounded sum (math.fsum).
  minmax_ref.arrow  (inputs: float_cases.minmax_table())
      float64 MIN / MAX inputs with NaNs and mixed +0.0 / -0.0, and what `if ((row < last) ^ is_max) last = row`
      (agg_funcs.h:198) returns for them in THIS row order -- the rule is row-order dependent on such inputs; the test
      pins the domain on which the HIP path's total order agrees with it.

Usage:  python tests/golden/gen_golden_float.py          (from the repo root)
"""
import os
import sys

import numpy as np
import pyarrow as pa

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref  # noqa: E402
from tests.golden import float_cases as C  # noqa: E402

SINGLE = 1


def write(name, table):
    with pa.OSFile(os.path.join(HERE, name), "wb") as f:
        with pa.ipc.new_file(f, table.schema) as w:
            w.write_table(table)


def run_ref(table, funcs, chunk):
    agg = ref.RefAggregate(SINGLE, ["k"], ["k"], funcs)
    t = table.combine_chunks()
    for start in range(0, t.num_rows, chunk):
        for b in t.slice(start, chunk).to_batches():
            agg.next(b)
    return pa.Table.from_batches([agg.result()])


def main():
    import json
    t = C.fsum_table()
    write("fsum_ref.arrow", run_ref(t, C.FSUM_FUNCS, C.FSUM_CHUNK))
    m = C.minmax_table()
    write("minmax_ref.arrow", run_ref(m, C.MINMAX_FUNCS, C.MINMAX_CHUNK))
    with open(os.path.join(HERE, "float_cases.json"), "w") as f:
        json.dump({"fsum_sha256": C.table_digest(t), "minmax_sha256": C.table_digest(m), "numpy": np.__version__,
                   "generator": "tests/golden/gen_golden_float.py over oracle/_ref (the reference's own aggregate)"}, f, indent=1)


if __name__ == "__main__":
    assert ref.available(), "build oracle/_ref first: make -C oracle/ref_build"
    main()
    print("wrote fsum_ref.arrow, minmax_ref.arrow, float_cases.json")
