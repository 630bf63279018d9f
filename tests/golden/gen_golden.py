#!/usr/bin/env python3
"""Generate tests/golden/*.arrow + manifest.json from the REAL reference.

Runs ONLY in the build container (needs /root/reference and oracle/_ref/libvinum_ref.so built
by `make -C oracle/ref_build`).  The committed outputs are data: seeded synthetic inputs and the
outputs the reference produced for them.

  * aggregate / sort cases  -> the reference's C++ operators themselves
        (vinum_cpp/src/operators/aggregate/*, operators/sort/sort.cpp) through oracle/ref.py
  * filter cases            -> the exact third-party call sequence of the reference's Python
        filter path.  The arithmetic there lives in NumPy (>=1.19 pinned, setup.py:34; 2.2 here)
        and Apache Arrow (3.0.0 pinned, setup.py:33; 25.0 here), neither under /root/reference:
            vinum/arrow/record_batch.py:101-125  column -> NumPy (nulls become NaN, float64)
            vinum/core/expressions.py:30-36      mask = x <op> literal        (NumPy ufunc)
            vinum/arrow/record_batch.py:85-90    pa.array(mask); batch.filter(mask, 'emit_null')
  * projection cases        -> NumPy ufuncs exactly as vinum/core/expressions.py:13-24 dispatches
        them, left-folded as vinum/core/base.py:145-151 does.

Usage:  python tests/golden/gen_golden.py          (from the repo root)
"""
import json
import os
import sys

import numpy as np
import pyarrow as pa

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref  # noqa: E402

COUNT_STAR, COUNT, MIN, MAX, SUM, AVG = range(6)
ONE_GROUP, SINGLE, MULTI = range(3)


def with_nulls(rng, arr, frac):
    if frac <= 0:
        return pa.array(arr)
    return pa.array(arr, mask=rng.random(len(arr)) < frac)


def wide_table(seed, n, card, null_frac=0.07):
    rng = np.random.default_rng(seed)
    k = rng.integers(0, card, n)
    cols = {}
    cols["k_i8"] = with_nulls(rng, (k % 200 - 100).astype(np.int8), null_frac)
    cols["k_i16"] = with_nulls(rng, (k % 3000 - 1500).astype(np.int16), null_frac)
    cols["k_i32"] = with_nulls(rng, (k - card // 2).astype(np.int32), null_frac)
    cols["k_i64"] = with_nulls(rng, (k.astype(np.int64) - card // 2) * 1000003, null_frac)
    cols["k_u8"] = with_nulls(rng, (k % 251).astype(np.uint8), null_frac)
    cols["k_u16"] = with_nulls(rng, (k % 60000).astype(np.uint16), null_frac)
    cols["k_u32"] = with_nulls(rng, (k.astype(np.uint32) * np.uint32(2654435761)), null_frac)
    cols["k_u64"] = with_nulls(rng, (k.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)), null_frac)
    kf = (k % 97).astype(np.float64) / 4.0 - 3.0
    kf[k % 97 == 12] = -0.0
    kf[k % 97 == 13] = 0.0
    cols["k_f64"] = with_nulls(rng, kf, null_frac)
    cols["k_f32"] = with_nulls(rng, kf.astype(np.float32), null_frac)
    cols["k_date32"] = with_nulls(rng, (18000 + k % 400).astype(np.int32), null_frac).view(pa.date32())
    cols["k_ts"] = with_nulls(rng, (1600000000000 + (k % 500) * 86400000).astype(np.int64),
                              null_frac).view(pa.timestamp("ms"))
    cols["k_time32"] = with_nulls(rng, (k % 86400).astype(np.int32), null_frac).view(pa.time32("s"))
    # values
    cols["v_i8"] = with_nulls(rng, rng.integers(-128, 128, n).astype(np.int8), 0.1)
    cols["v_i16"] = with_nulls(rng, rng.integers(-32768, 32768, n).astype(np.int16), 0.1)
    cols["v_i32"] = with_nulls(rng, rng.integers(-2**31, 2**31, n).astype(np.int32), 0.1)
    cols["v_i64"] = with_nulls(rng, rng.integers(-2**40, 2**40, n).astype(np.int64), 0.1)
    cols["v_u8"] = with_nulls(rng, rng.integers(0, 256, n).astype(np.uint8), 0.1)
    cols["v_u16"] = with_nulls(rng, rng.integers(0, 65536, n).astype(np.uint16), 0.1)
    cols["v_u32"] = with_nulls(rng, rng.integers(0, 2**32, n).astype(np.uint32), 0.1)
    cols["v_u64"] = with_nulls(rng, rng.integers(0, 2**50, n).astype(np.uint64), 0.1)
    cols["v_i64big"] = with_nulls(rng, (2**63 - 1 - rng.integers(0, 1000, n)).astype(np.int64)
                                  * rng.choice([-1, 1], n), 0.1)
    cols["v_u64big"] = with_nulls(rng, (np.uint64(2**64 - 1) - rng.integers(0, 1000, n).astype(np.uint64)), 0.1)
    cols["v_f64q"] = with_nulls(rng, rng.integers(0, 2**14, n).astype(np.float64) / 128.0, 0.1)
    cols["v_f64"] = with_nulls(rng, np.round(rng.lognormal(2.4, 0.6, n), 2), 0.1)
    cols["v_f32"] = with_nulls(rng, rng.normal(0, 100, n).astype(np.float32), 0.1)
    cols["v_time32"] = with_nulls(rng, rng.integers(0, 86400, n).astype(np.int32), 0.1).view(pa.time32("s"))
    cols["v_time64"] = with_nulls(rng, rng.integers(0, 86400 * 10**6, n).astype(np.int64), 0.1).view(pa.time64("us"))
    cols["v_date32"] = with_nulls(rng, rng.integers(0, 20000, n).astype(np.int32), 0.1).view(pa.date32())
    cols["v_ts"] = with_nulls(rng, rng.integers(0, 2**41, n).astype(np.int64), 0.1).view(pa.timestamp("us"))
    sparse = rng.normal(11, 9, n)
    cols["v_sparse"] = pa.array(sparse, mask=(k % 3 != 0))  # groups with k%3!=0 see only NULLs
    return pa.table(cols)


ALL_FUNCS = lambda col: [(COUNT, col, f"count_{col}"), (MIN, col, f"min_{col}"), (MAX, col, f"max_{col}"),
                         (SUM, col, f"sum_{col}"), (AVG, col, f"avg_{col}")]
MINMAX = lambda col: [(COUNT, col, f"count_{col}"), (MIN, col, f"min_{col}"), (MAX, col, f"max_{col}")]


def agg_cases():
    cases = []
    ints = ["v_i8", "v_i16", "v_i32", "v_i64", "v_u8", "v_u16", "v_u32", "v_u64"]
    f_int = [(COUNT_STAR, "", "count_star")] + [f for c in ints for f in ALL_FUNCS(c)]
    f_flt = [(COUNT_STAR, "", "count_star")] + [f for c in ["v_f64q", "v_f64", "v_f32", "v_sparse"]
                                                 for f in ALL_FUNCS(c)]
    f_tmp = ([(COUNT_STAR, "", "count_star")] + ALL_FUNCS("v_time32") + ALL_FUNCS("v_time64")
             + MINMAX("v_date32") + MINMAX("v_ts"))
    f_big = [(SUM, "v_i64big", "sum_big"), (AVG, "v_i64big", "avg_big"), (SUM, "v_u64big", "sum_ubig"),
             (AVG, "v_u64big", "avg_ubig"), (SUM, "v_i64", "sum_i64"), (COUNT_STAR, "", "n")]
    tabs = {"w300": dict(seed=11, n=4000, card=300), "w5k": dict(seed=12, n=6000, card=5000),
            "w7": dict(seed=13, n=3000, card=7)}
    for key in ["k_i8", "k_i16", "k_i32", "k_i64", "k_u8", "k_u16", "k_u32", "k_u64", "k_f64", "k_f32",
                "k_date32", "k_ts", "k_time32"]:
        cases.append(dict(name=f"single_{key}_ints", table="w300", kind=SINGLE, groupby=[key], agg_cols=[key],
                          funcs=f_int, chunk=1000))
    cases.append(dict(name="single_k_i64_floats", table="w300", kind=SINGLE, groupby=["k_i64"],
                      agg_cols=["k_i64"], funcs=f_flt, chunk=1500))
    cases.append(dict(name="single_k_i64_temporal", table="w300", kind=SINGLE, groupby=["k_i64"],
                      agg_cols=["k_i64"], funcs=f_tmp, chunk=777))
    cases.append(dict(name="single_k_i64_overflow", table="w7", kind=SINGLE, groupby=["k_i64"],
                      agg_cols=["k_i64"], funcs=f_big, chunk=512))
    cases.append(dict(name="single_k_i32_5k_floats", table="w5k", kind=SINGLE, groupby=["k_i32"],
                      agg_cols=["k_i32"], funcs=f_flt, chunk=2048))
    cases.append(dict(name="single_k_u64_5k_ints", table="w5k", kind=SINGLE, groupby=["k_u64"],
                      agg_cols=["k_u64"], funcs=f_int, chunk=4096))
    cases.append(dict(name="single_key_not_selected", table="w300", kind=SINGLE, groupby=["k_i32"],
                      agg_cols=[], funcs=[(COUNT_STAR, "", "n"), (SUM, "v_f64q", "s")], chunk=1000))
    cases.append(dict(name="multi_one_key", table="w300", kind=MULTI, groupby=["k_i64"], agg_cols=["k_i64"],
                      funcs=f_flt, chunk=1000))
    cases.append(dict(name="multi_i8_u16", table="w300", kind=MULTI, groupby=["k_i8", "k_u16"],
                      agg_cols=["k_i8", "k_u16"], funcs=f_flt, chunk=1000))
    cases.append(dict(name="multi_4keys", table="w7", kind=MULTI,
                      groupby=["k_i8", "k_date32", "k_f64", "k_ts"], agg_cols=["k_ts", "k_i8", "k_f64", "k_date32"],
                      funcs=f_int, chunk=700))
    cases.append(dict(name="multi_3keys_subset", table="w5k", kind=MULTI,
                      groupby=["k_i32", "k_u8", "k_f32"], agg_cols=["k_u8"], funcs=f_tmp, chunk=2500))
    cases.append(dict(name="one_group_ints", table="w300", kind=ONE_GROUP, groupby=[], agg_cols=[],
                      funcs=f_int, chunk=1000))
    cases.append(dict(name="one_group_floats", table="w300", kind=ONE_GROUP, groupby=[], agg_cols=[],
                      funcs=f_flt, chunk=999))
    cases.append(dict(name="one_group_temporal_big", table="w7", kind=ONE_GROUP, groupby=[], agg_cols=[],
                      funcs=f_tmp + f_big, chunk=1024))
    return tabs, cases


def sliced_batches(table: pa.Table, chunk):
    """Batches of `chunk` rows cut from a combined table -> non-zero Arrow offsets everywhere."""
    t = table.combine_chunks()
    out = []
    for start in range(0, t.num_rows, chunk):
        out.extend(t.slice(start, chunk).to_batches())
    return out


def write_ipc(path, batches, schema):
    with pa.OSFile(path, "wb") as f:
        with pa.ipc.new_file(f, schema) as w:
            for b in batches:
                w.write_batch(b)


def gen_agg(manifest):
    tabs, cases = agg_cases()
    for tname, spec in tabs.items():
        t = wide_table(**spec)
        write_ipc(os.path.join(HERE, f"agg_in_{tname}.arrow"), t.to_batches(), t.schema)
    cache = {}
    for c in cases:
        t = cache.setdefault(c["table"], wide_table(**tabs[c["table"]]))
        r = ref.RefAggregate(c["kind"], c["groupby"], c["agg_cols"], c["funcs"])
        for b in sliced_batches(t, c["chunk"]):
            r.next(b)
        res = r.result()
        out = f"agg_out_{c['name']}.arrow"
        write_ipc(os.path.join(HERE, out), [res], res.schema)
        manifest["agg"].append(dict(c, input=f"agg_in_{c['table']}.arrow", expected=out))
        print("agg", c["name"], res.num_rows, "groups")


def sort_table(seed, n):
    rng = np.random.default_rng(seed)
    a = rng.integers(-5, 5, n).astype(np.int64)
    f = np.round(rng.normal(11, 9, n), 1)
    f[rng.random(n) < 0.05] = np.nan
    f[rng.random(n) < 0.02] = -0.0
    g = rng.normal(0, 1, n).astype(np.float32)
    return pa.table({
        "rowid": pa.array(np.arange(n, dtype=np.int64)),
        "a": with_nulls(rng, a, 0.05),
        "f": with_nulls(rng, f, 0.05),
        "g": with_nulls(rng, g, 0.0),
        "u": with_nulls(rng, rng.integers(0, 2**64 - 1, n, dtype=np.uint64), 0.03),
        "d": with_nulls(rng, rng.integers(0, 30, n).astype(np.int32), 0.05).view(pa.date32()),
        "p": with_nulls(rng, rng.normal(0, 1, n), 0.1),
    })


def gen_sort(manifest):
    t = sort_table(21, 3000)
    write_ipc(os.path.join(HERE, "sort_in.arrow"), t.to_batches(), t.schema)
    cases = [
        ("f_asc", ["f"], [0]), ("f_desc", ["f"], [1]), ("a_asc_f_desc", ["a", "f"], [0, 1]),
        ("a_desc_d_asc_f_asc", ["a", "d", "f"], [1, 0, 0]), ("u_desc", ["u"], [1]), ("g_asc", ["g"], [0]),
        ("d_desc_a_asc", ["d", "a"], [1, 0]),
    ]
    for name, cols, orders in cases:
        s = ref.RefSort(cols, orders)
        for b in sliced_batches(t, 1000):
            s.next(b)
        res = s.sorted()
        out = f"sort_out_{name}.arrow"
        write_ipc(os.path.join(HERE, out), [res], res.schema)
        manifest["sort"].append(dict(name=name, cols=cols, orders=orders, chunk=1000, input="sort_in.arrow",
                                     expected=out))
        print("sort", name)


def ref_np_column(arr: pa.Array) -> np.ndarray:
    # record_batch.py:112-118
    try:
        return arr.to_numpy(zero_copy_only=True)
    except pa.ArrowInvalid:
        return arr.to_numpy(zero_copy_only=False)


REF_CMP = {  # expressions.py:30-36
    "eq": lambda x, y: x == y, "ne": lambda x, y: x != y, "gt": lambda x, y: x > y,
    "ge": lambda x, y: x >= y, "lt": lambda x, y: x < y, "le": lambda x, y: x <= y,
}


def filter_table(seed, n):
    rng = np.random.default_rng(seed)
    return pa.table({
        "fare": pa.array(rng.integers(0, 2**14, n).astype(np.float64) / 128.0),
        "fare_n": with_nulls(rng, rng.integers(0, 2**14, n).astype(np.float64) / 128.0, 0.05),
        "i": pa.array(rng.integers(-1000, 1000, n).astype(np.int64)),
        "i_n": with_nulls(rng, rng.integers(-1000, 1000, n).astype(np.int64), 0.05),
        "big_n": with_nulls(rng, (2**62 + rng.integers(0, 8, n)).astype(np.int64), 0.05),
        "f32": pa.array(rng.normal(0, 10, n).astype(np.float32)),
        "i32_n": with_nulls(rng, rng.integers(-50, 50, n).astype(np.int32), 0.1),
        "u8": pa.array(rng.integers(0, 255, n).astype(np.uint8)),
    })


def gen_filter(manifest):
    t = filter_table(31, 5000).combine_chunks()
    write_ipc(os.path.join(HERE, "filter_in.arrow"), t.to_batches(), t.schema)
    cases = [("fare", "gt", 64.0), ("fare", "le", 0.5), ("fare", "gt", 1e9), ("fare", "ge", -1.0),
             ("fare_n", "gt", 64.0), ("fare_n", "ne", 64.0), ("fare_n", "eq", 3.5),
             ("i", "gt", 5), ("i", "le", -3), ("i", "gt", 5.5), ("i_n", "gt", 5), ("i_n", "ne", 5),
             ("big_n", "gt", 2**62 + 3), ("f32", "lt", 0.1), ("i32_n", "ge", 0), ("u8", "lt", 100)]
    for col, op, lit in cases:
        outs = []
        for off, ln in [(0, 2000), (2000, 1777), (3777, 1223)]:
            batch = t.slice(off, ln).to_batches()[0]
            x = ref_np_column(batch.column(batch.schema.get_field_index(col)))
            with np.errstate(all="ignore"):
                mask = REF_CMP[op](x, lit)
            bitmask = pa.array(mask)                                             # record_batch.py:86-87
            outs.append(batch.filter(bitmask, null_selection_behavior="emit_null"))  # :88-90
        name = f"{col}_{op}_{str(lit).replace('.', 'p').replace('-', 'm').replace('+', '')}"
        out = f"filter_out_{name}.arrow"
        write_ipc(os.path.join(HERE, out), outs, t.schema)
        manifest["filter"].append(dict(name=name, column=col, op=op, literal=lit,
                                       literal_is_float=isinstance(lit, float),
                                       slices=[(0, 2000), (2000, 1777), (3777, 1223)],
                                       input="filter_in.arrow", expected=out))
        print("filter", name, sum(o.num_rows for o in outs))


def gen_project(manifest):
    rng = np.random.default_rng(41)
    n = 4000
    t = pa.table({
        "v": pa.array(rng.normal(11, 9, n)), "a": pa.array(rng.normal(0, 3, n)),
        "b": pa.array(rng.lognormal(0, 1, n)),
        "i": pa.array(rng.integers(-10**6, 10**6, n).astype(np.int64)),
        "j": pa.array(rng.integers(-50, 50, n).astype(np.int64)),
        "big": pa.array(rng.integers(2**62, 2**63 - 1, n).astype(np.int64)),
    })
    write_ipc(os.path.join(HERE, "project_in.arrow"), t.to_batches(), t.schema)
    N = {name: ref_np_column(t.column(name).combine_chunks()) for name in t.schema.names}
    # expression programs in a tiny prefix form our tests re-interpret; results via NumPy ufuncs
    # (expressions.py:13-24), n-ary chains left-folded (base.py:145-151)
    exprs = {
        "v*2+1": ("add", ("mul", "v", 2), 1), "v-a": ("sub", "v", "a"), "a*b": ("mul", "a", "b"),
        "v/b": ("div", "v", "b"), "i+j": ("add", "i", "j"), "i*j": ("mul", "i", "j"), "i/j": ("div", "i", "j"),
        "i%j": ("mod", "i", "j"), "v%b": ("mod", "v", "b"), "-v": ("neg", "v"), "-i": ("neg", "i"),
        "big+big": ("add", "big", "big"), "i+0.5": ("add", "i", 0.5), "i&j": ("band", "i", "j"),
        "i|j": ("bor", "i", "j"), "i^j": ("bxor", "i", "j"), "~i": ("bnot", "i"),
        "(1-v)*(2+a)": ("mul", ("sub", 1, "v"), ("add", 2, "a")), "i-j-j": ("sub", ("sub", "i", "j"), "j"),
    }
    UF = {"add": np.add, "sub": np.subtract, "mul": np.multiply, "div": np.divide, "mod": np.mod,
          "neg": np.negative, "band": np.bitwise_and, "bor": np.bitwise_or, "bxor": np.bitwise_xor,
          "bnot": lambda x: ~x}

    def ev(e):
        if isinstance(e, str):
            return N[e]
        if isinstance(e, (int, float)):
            return e
        return UF[e[0]](*[ev(x) for x in e[1:]])

    out_cols = {}
    with np.errstate(all="ignore"):
        for name, e in exprs.items():
            out_cols[name] = pa.array(ev(e))
    res = pa.table(out_cols)
    write_ipc(os.path.join(HERE, "project_out.arrow"), res.to_batches(), res.schema)
    manifest["project"].append(dict(input="project_in.arrow", expected="project_out.arrow",
                                    exprs={k: v for k, v in exprs.items()}))
    print("project", len(exprs), "expressions")


def main():
    if not ref.available():
        raise SystemExit("build oracle/_ref first:  make -C oracle/ref_build")
    manifest = {"agg": [], "sort": [], "filter": [], "project": [],
                "generator": "tests/golden/gen_golden.py", "numpy": np.__version__, "pyarrow": pa.__version__}
    gen_agg(manifest)
    gen_sort(manifest)
    gen_filter(manifest)
    gen_project(manifest)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)


if __name__ == "__main__":
    main()
