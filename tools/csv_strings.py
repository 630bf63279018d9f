"""CSV ingest with string / date / timestamp columns: the device parser (vnm_csv_parse_block_ex) against the round-4 split (those
columns through pyarrow + dictionary_encode on the host) and against pyarrow's own streaming reader.
usage: python tools/csv_strings.py [rows] [distinct strings]"""
import io
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import pyarrow as pa
import pyarrow.csv as pacsv


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
    d = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000
    rng = np.random.default_rng(0)
    names = np.array([f"city{'%07d' % i}" for i in range(d)])
    t = pa.table({
        "id": pa.array(np.arange(n, dtype=np.int64)),
        "city": pa.array(names[rng.integers(0, d, n)]),
        "day": pa.array(rng.integers(0, 20000, n).astype(np.int32), type=pa.int32()).cast(pa.date32()),
        "ts": pa.array(rng.integers(0, 1_700_000_000, n).astype(np.int64)).cast(pa.timestamp("s")),
        "fare": pa.array(np.round(rng.lognormal(2.2, 0.6, n), 2)),
    })
    path = os.path.join(tempfile.mkdtemp(), "strings.csv")
    pacsv.write_csv(t, path, write_options=pacsv.WriteOptions(quoting_style="none"))
    size = os.path.getsize(path)
    print(f"{n} rows, {d} distinct strings, {size / 1e6:.0f} MB: {pacsv.open_csv(path).schema.types}")
    from vinum_amd import planner
    from vinum_amd.io import stream_csv

    def drain(reader):
        rows = 0
        while True:
            try:
                rows += reader.read_next_device_batch().num_rows
            except StopIteration:
                return rows

    q = dict(select=["city", ["fn", "count_star"], ["fn", "sum", "fare"], ["fn", "max", "ts"]], aliases=[None, "n", "s", "t"], group_by=["city"])
    for label, kw in (("device parser", {}), ("strings / dates through pyarrow (round 4)", {"numeric_only": True})):
        for rep in range(2):
            t0 = time.perf_counter()
            rows = drain(stream_csv(path, **kw))
            t1 = time.perf_counter()
            res = planner.execute(q, stream_csv(path, **kw))
            t2 = time.perf_counter()
        print(f"{label:45s} ingest {1e3 * (t1 - t0):8.1f} ms = {size / (t1 - t0) / 1e9:5.2f} GB/s, {rows / (t1 - t0) / 1e6:6.1f} M rows/s;"
              f"  GROUP BY city query {1e3 * (t2 - t1):8.1f} ms ({res.num_rows} groups)")
    t0 = time.perf_counter()
    rows = sum(b.num_rows for b in pacsv.open_csv(path, read_options=pacsv.ReadOptions(block_size=64 << 20)))
    t1 = time.perf_counter()
    print(f"{'pyarrow.csv.open_csv alone (host, threads)':45s} ingest {1e3 * (t1 - t0):8.1f} ms = {size / (t1 - t0) / 1e9:5.2f} GB/s, {rows / (t1 - t0) / 1e6:6.1f} M rows/s")
    os.remove(path)


if __name__ == "__main__":
    main()
