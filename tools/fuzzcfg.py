"""Rebuilds the configuration of one seed of tests/test_gpu_agg.py::test_random_plans_vs_oracle (debugging aid)."""
import sys
sys.path.insert(0, ".")
import numpy as np, pyarrow as pa
from tests import util


def make(seed):
    envs = {}
    """Seeded differential test over the whole dispatch space: random key columns (1-3, mixed widths, NULLs), random
    function lists over random typed inputs (NULLs, narrow types), random group counts, hints (right / absent),
    predicates on any column, one or two batches, skewed or uniform keys.  Whatever path the operator picks (scan
    kernels, narrow / wide partitioned entries, packed composite keys, spill, fallbacks) must equal the oracle."""
    from oracle import oracle as O
    rng = np.random.default_rng(1000 + seed)
    envs["VNM_AGG_ESTIMATE_MIN_ROWS"] = "1000"
    if rng.random() < 0.5:
        envs["VNM_AGG_PART_L1_MAX"] = "4"
    n = int(rng.integers(150_000, 420_000))
    groups = int(rng.choice([3, 40, 900, 5_000, 60_000, 250_000]))
    nkeys = int(rng.choice([1, 1, 1, 2, 3]))
    skew = rng.random() < 0.3

    def int_col(lo, hi, dtype, null_p):
        a = rng.integers(lo, hi, n).astype(dtype)
        return pa.array(a, mask=(rng.random(n) < null_p) if null_p else None)

    cols = {}
    per_key = max(2, int(round(groups ** (1.0 / nkeys))))
    for j in range(nkeys):
        u = rng.random(n)
        if skew:
            u = u ** 6
        vals = np.floor(u * per_key).astype(np.int64)
        kind = rng.choice(["i64", "i32", "f64", "u8"]) if (nkeys > 1 or rng.random() < 0.3) else "i64"
        null_p = 0.05 if rng.random() < 0.3 else 0.0
        mask = (rng.random(n) < null_p) if null_p else None
        if kind == "i64":
            arr = pa.array(vals * 7919 - 13, mask=mask)
        elif kind == "i32":
            arr = pa.array((vals - per_key // 2).astype(np.int32), mask=mask)
        elif kind == "u8":
            arr = pa.array((vals % 251).astype(np.uint8), mask=mask)
        else:
            arr = pa.array(vals.astype(np.float64) * 0.5 - 1.0, mask=mask)
        cols[f"k{j}"] = arr
    key_names = list(cols)
    makers = {
        "f64": lambda p: pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0, mask=(rng.random(n) < p) if p else None),
        "i64": lambda p: int_col(-2**45, 2**45, np.int64, p),
        "i32": lambda p: int_col(-2**31, 2**31 - 1, np.int32, p),
        "u16": lambda p: int_col(0, 2**16, np.uint16, p),
        "u64": lambda p: pa.array(rng.integers(0, 2**63, n).astype(np.uint64) * np.uint64(2), mask=(rng.random(n) < p) if p else None),
        "f32": lambda p: pa.array((rng.integers(0, 2**10, n) / 8.0).astype(np.float32), mask=(rng.random(n) < p) if p else None),
    }
    ninputs = int(rng.choice([0, 1, 1, 2, 3]))
    in_names = []
    for c in range(ninputs):
        t = str(rng.choice(list(makers)))
        cols[f"v{c}"] = makers[t](0.15 if rng.random() < 0.35 else 0.0)
        in_names.append(f"v{c}")
    funcs = [(O.COUNT_STAR, "", "n")] if (ninputs == 0 or rng.random() < 0.5) else []
    for name in in_names:
        picks = rng.choice([O.SUM, O.AVG, O.MIN, O.MAX, O.COUNT], size=int(rng.integers(1, 4)), replace=False)
        for f in picks:
            funcs.append((int(f), name, f"f{len(funcs)}"))
    pred = None
    r = rng.random()
    if r < 0.3 and in_names:
        pred = (in_names[0], ">", 5 if pa.types.is_integer(cols[in_names[0]].type) else 5.0)
    elif r < 0.5:
        cols["p"] = pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0)
        pred = ("p", "<=", 90.0)
    t = pa.table(cols)
    names = t.schema.names
    kind = O.SINGLE if nkeys == 1 else O.MULTI
    hint = groups if rng.random() < 0.5 else 0
    batches = t.to_batches() if rng.random() < 0.5 else util.sliced_batches(t, n // 2 + 1)
    return dict(envs=envs, t=t, kind=kind, key_names=key_names, funcs=funcs, pred=pred, hint=hint, batches=batches, in_names=in_names)


if __name__ == "__main__":
    c = make(int(sys.argv[1]))
    print("envs", c["envs"], "rows", c["t"].num_rows, "schema", [(f.name, str(f.type), c["t"].column(f.name).null_count) for f in c["t"].schema])
    print("funcs", c["funcs"], "pred", c["pred"], "hint", c["hint"], "batches", [b.num_rows for b in c["batches"]])
