#!/usr/bin/env python3
"""Golden vectors for Sort over tables that carry NON-NUMERIC columns (VERDICT r03 missing #1): the shapes of the reference's
`orderby_queries` (vinum/tests/test_query_results.py:627-745: `select * ... order by total` over a table with string columns,
`order by city_from desc, total asc`) and its NULL / NaN ordering cases (:1252-1266), plus a seeded 6000-row table with string /
binary keys (NULLs, empty strings, multi-byte UTF-8, shared prefixes), boolean, decimal128 and date payloads.

Inputs: the two small tables are the fixture DATA of the reference's tests (vinum/tests/conftest.py:50-101, rows transcribed);
outputs: produced HERE by the reference's own Sort operator (vinum_cpp/src/operators/sort/sort.cpp through oracle/ref.py,
built by `make -C oracle/ref_build`).  Runs only in the build container.

Usage:  python tests/golden/gen_golden_sort_mixed.py        (from the repo root; rewrites the "sort_mixed" list of manifest.json)
"""
import decimal
import json
import os
import sys

import numpy as np
import pyarrow as pa

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref  # noqa: E402

NAN = float("nan")


def groupby_table():
    # vinum/tests/conftest.py:50-73 (create_test_groupby_data)
    names = ("id", "timestamp", "date", "vendor_id", "city_from", "city_to", "lat", "lng", "name", "tax", "tip", "total")
    rows = (
        (1, 1602127614, "2020-10-08T03:26:54", 1, "Berlin", "Munich", 52.51, 13.66, "Joe", 0.43, 1, 2.43),
        (2, 1602217613, "2020-10-09T04:26:53", 2, "Munich", "Riva", 48.51, 12.3, "Jonas", 2.0, 4.34, 143.15),
        (3, 1602304012, "2020-10-10T04:26:52", 1, "Riva", "Naples", 44.89, 14.23, "Joseph", 1.59, 11, 33.40),
        (4, 1602390411, "2020-10-11T04:26:51", 3, "San Francisco", "Naples", 42.89, 15.89, "Joseph", 1.69, 5.3, 53.1),
        (5, 1602476810, "2020-10-12T04:26:50", 1, "Berlin", "Riva", 44.89, 14.23, "Joseph", 1.59, 11, 33.40),
        (6, 1602563209, "2020-10-13T04:26:49", 2, "Munich", "Riva", 48.51, 12.3, "Jonas", 2.0, 5.34, 13.15),
        (7, 1602649608, "2020-10-14T04:26:48", 1, "Berlin", "Munich", 44.89, 14.23, "Joseph", 1.59, 11, 33.40),
        (8, 1602736007, "2020-10-15T04:26:47", 1, "Berlin", "Munich", 52.51, 13.66, "Joe", 0.43, 0.4, 2.43),
    )
    return pa.Table.from_pydict({n: [r[i] for r in rows] for i, n in enumerate(names)})


def null_table():
    # vinum/tests/conftest.py:76-101 (create_null_test_data)
    names = ("id", "timestamp", "date", "is_vendor", "city_from", "city_to", "lat", "lng", "name", "total")
    rows = (
        (1, 1602127614, None, True, None, "Munich", 52.51, 13.66, "Joe", None),
        (2, 1602217613, "2020-10-09T04:26:53", True, "Munich", "Riva", 48.51, 12.3, None, 143.15),
        (3, 1602304012, "2020-10-10T04:26:52", False, None, "Naples", 44.89, 14.23, "Joseph", 33.40),
        (4, 1602390411, "2020-10-11T04:26:51", None, "San Francisco", "Naples", 42.89, 15.89, "Joseph", 53.1),
        (5, None, "2020-10-12T04:26:50", True, "Berlin", "Riva", 44.89, 14.23, None, NAN),
        (6, 1602563209, "2020-10-13T04:26:49", None, "Munich", "Riva", 48.51, 12.3, "Jonas", None),
        (7, None, None, None, "Berlin", "Munich", 44.89, 14.23, "Joseph", 33.40),
        (8, 1602736007, "2020-10-15T04:26:47", None, "Berlin", "Munich", 52.51, 13.66, "Joe", NAN),
    )
    return pa.Table.from_pydict({n: [r[i] for r in rows] for i, n in enumerate(names)})


def random_table(seed=77, n=6000):
    rng = np.random.default_rng(seed)
    # (multi-byte UTF-8 spelled with escapes: Muenchen / Zuerich with umlauts, Tokyo in kanji)
    words = ["", "a", "ab", "abc", "abd", "b", "Berlin", "berlin", "München", "Munich", "Zürich", "東京", "zz", "a b", " lead", "trail "]
    words += [f"city_{i:04d}" for i in range(300)]
    s = pa.array([words[i] for i in rng.integers(0, len(words), n)], mask=rng.random(n) < 0.06)
    b = pa.array([bytes(rng.integers(0, 256, int(rng.integers(0, 4))).astype(np.uint8)) for _ in range(n)], type=pa.binary(), mask=rng.random(n) < 0.05)
    f = np.round(rng.normal(11, 9, n), 1)
    f[rng.random(n) < 0.04] = np.nan
    dec = pa.array([decimal.Decimal(int(x)).scaleb(-2) for x in rng.integers(-10**6, 10**6, n)], type=pa.decimal128(12, 2), mask=rng.random(n) < 0.03)
    return pa.table({
        "rowid": pa.array(np.arange(n, dtype=np.int64)),
        "s": s,
        "ls": pa.array([f"{int(x) % 17:02d}-tag" for x in rng.integers(0, 10**6, n)], type=pa.large_string()),
        "b": b,
        "f": pa.array(f, mask=rng.random(n) < 0.05),
        "i": pa.array(rng.integers(-4, 4, n).astype(np.int32), mask=rng.random(n) < 0.05),
        "flag": pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.1),
        "dec": dec,
        "day": pa.array(rng.integers(18000, 18030, n).astype(np.int32)).view(pa.date32()),
    })


def write_ipc(path, batches, schema):
    with pa.OSFile(path, "wb") as f:
        with pa.ipc.new_file(f, schema) as w:
            for b in batches:
                w.write_batch(b)


def sliced(t, chunk):
    t = t.combine_chunks()
    return [b for off in range(0, max(t.num_rows, 1), chunk) for b in t.slice(off, chunk).to_batches()]


def main():
    tables = {"groupby": (groupby_table(), 3), "null": (null_table(), 3), "random": (random_table(), 2500)}
    cases = [
        # test_query_results.py:627-745
        ("groupby", "star_order_by_total", ["total"], [0]),
        ("groupby", "star_order_by_total_tip", ["total", "tip"], [0, 0]),
        ("groupby", "city_desc_total_asc", ["city_from", "total"], [1, 0]),
        ("groupby", "city_desc_total_desc", ["city_from", "total"], [1, 1]),
        ("groupby", "date_string_desc", ["date"], [1]),
        # :1252-1266 (NULL and NaN ordering) and string keys with NULLs
        ("null", "total_id", ["total", "id"], [0, 0]),
        ("null", "total_desc_id", ["total", "id"], [1, 0]),
        ("null", "city_from_asc", ["city_from"], [0]),
        ("null", "name_desc_id_asc", ["name", "id"], [1, 0]),
        ("null", "timestamp_asc", ["timestamp"], [0]),
        ("random", "s_asc", ["s"], [0]),
        ("random", "s_desc_f_asc", ["s", "f"], [1, 0]),
        ("random", "i_asc_s_desc", ["i", "s"], [0, 1]),
        ("random", "b_asc_rowid_desc", ["b", "rowid"], [0, 1]),
        ("random", "ls_desc_s_asc_f_desc", ["ls", "s", "f"], [1, 0, 1]),
        ("random", "dec_desc", ["dec"], [1]),
        ("random", "f_desc_payload_only", ["f"], [1]),
        ("random", "day_asc_s_asc", ["day", "s"], [0, 0]),
    ]
    out = []
    for tname, (t, chunk) in tables.items():
        write_ipc(os.path.join(HERE, f"sortmix_in_{tname}.arrow"), sliced(t, chunk), t.schema)
    for tname, name, cols, orders in cases:
        t, chunk = tables[tname]
        s = ref.RefSort(cols, orders)
        for b in sliced(t, chunk):
            s.next(b)
        res = s.sorted()
        fn = f"sortmix_out_{name}.arrow"
        write_ipc(os.path.join(HERE, fn), [res], res.schema)
        out.append(dict(name=name, cols=cols, orders=orders, chunk=chunk, input=f"sortmix_in_{tname}.arrow", expected=fn))
        print("sort_mixed", name, res.num_rows)
    mp = os.path.join(HERE, "manifest.json")
    with open(mp) as f:
        man = json.load(f)
    man["sort_mixed"] = out
    with open(mp, "w") as f:
        json.dump(man, f, indent=1)


if __name__ == "__main__":
    main()
