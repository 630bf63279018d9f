"""Seeded inputs of the float-SUM / float-MIN-MAX golden cases (pure NumPy: no reference code).

Shared by tests/golden/gen_golden_float.py (which runs the REAL reference over them in the build container and commits
its outputs, fsum_ref.arrow / minmax_ref.arrow, plus a SHA-256 of the inputs in float_cases.json) and by the tests,
which regenerate the same inputs from the seeds and refuse to run if the checksum differs.
"""
import hashlib

import numpy as np
import pyarrow as pa

COUNT_STAR, COUNT, MIN, MAX, SUM, AVG = range(6)

FSUM_COLS = ["logn", "norm", "wide", "cancel", "f32"]
FSUM_FUNCS = [(COUNT_STAR, "", "n")] + [f for c in FSUM_COLS for f in ((SUM, c, f"sum_{c}"), (AVG, c, f"avg_{c}"))]
FSUM_CHUNK = 50_000
MINMAX_FUNCS = [(MIN, "v", "mn"), (MAX, "v", "mx"), (COUNT, "v", "c")]
MINMAX_CHUNK = 3000


def fsum_table() -> pa.Table:
    """NON-quantised inputs in groups of 1 ... 1e5 rows (rows shuffled): lognormal fares, mixed-sign normals, an
    80-binade dynamic range, a column whose terms cancel to ~1e-12 of their size, and a float32 column."""
    rng = np.random.default_rng(2026)
    sizes = [1] * 20 + [2] * 20 + [3] * 20 + [10] * 20 + [100] * 12 + [1000] * 8 + [10_000] * 4 + [100_000] * 2
    k = np.repeat(np.arange(len(sizes), dtype=np.int64) * 7919 - 50, sizes)
    rng.shuffle(k)
    n = len(k)
    logn = rng.lognormal(2.4, 1.3, n)
    norm = rng.normal(0.0, 1e3, n)
    wide = rng.lognormal(0.0, 1.0, n) * np.exp2(rng.integers(-40, 40, n).astype(np.float64))
    cancel = rng.normal(0.0, 1.0, n) * 1e12
    m = len(cancel[1::2])
    cancel[1::2] = -cancel[0::2][:m] + rng.normal(0.0, 1.0, m)
    f32 = rng.lognormal(1.0, 1.0, n).astype(np.float32)
    return pa.table({"k": pa.array(k),
                     "logn": pa.array(logn, mask=rng.random(n) < 0.02),
                     "norm": pa.array(norm),
                     "wide": pa.array(wide),
                     "cancel": pa.array(cancel),
                     "f32": pa.array(f32, mask=rng.random(n) < 0.02)})


def minmax_table() -> pa.Table:
    """float64 MIN / MAX inputs, 240 groups of 40 rows (shuffled, 3 % NULLs):
       groups   0..59  plain values;  60..119 with NaNs (every tenth group ALL NaN);
       groups 120..179 the extreme is zero with both +0.0 and -0.0 present;  180..239 NaNs and mixed zeros."""
    rng = np.random.default_rng(7)
    groups, per = 240, 40
    n = groups * per
    k = np.repeat(np.arange(groups, dtype=np.int64), per)
    v = rng.normal(0.0, 100.0, n)
    for grp in range(60, 120):
        rows = np.nonzero(k == grp)[0]
        if grp % 10 == 0:
            v[rows] = np.nan
        else:
            v[rng.choice(rows, size=int(rng.integers(1, 4)), replace=False)] = np.nan
    for grp in range(120, 240):
        rows = np.nonzero(k == grp)[0]
        v[rows] = np.abs(v[rows]) * (1.0 if grp % 2 == 0 else -1.0)     # zero is the MIN (even) / the MAX (odd)
        z = rng.choice(rows, size=4, replace=False)
        v[z[:2]] = 0.0
        v[z[2:]] = -0.0
        if grp >= 180:
            v[rng.choice(np.setdiff1d(rows, z), size=2, replace=False)] = np.nan
    perm = rng.permutation(n)
    return pa.table({"k": pa.array(k[perm]), "v": pa.array(v[perm], mask=rng.random(n) < 0.03)})


def table_digest(t: pa.Table) -> str:
    h = hashlib.sha256()
    for col in t.combine_chunks().columns:
        for chunk in col.chunks:
            for buf in chunk.buffers():
                if buf is not None:
                    h.update(buf.to_pybytes())
    return h.hexdigest()
