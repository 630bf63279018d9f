"""stream_csv group-by (BASELINE configs[0] / [3] shape) end to end from a CSV file in page cache:
pyarrow.csv streaming reader (what vinum.stream_csv uses; single-threaded) + staging  vs  the device tokeniser / parser.
python tools/csvbench.py [rows]"""
import io, os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pyarrow as pa, pyarrow.csv as pacsv
from vinum_amd import planner
from vinum_amd.io import stream_csv

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
rng = np.random.default_rng(0)
w = np.array([165, 34808, 7386, 2183, 1016, 3453, 989], float)
t = pa.table({"key": pa.array(np.arange(n)), "fare_amount": np.round(rng.lognormal(2.2, 0.6, n), 2),
              "pickup_longitude": rng.normal(-73.9, 0.1, n), "pickup_latitude": rng.normal(40.75, 0.1, n),
              "dropoff_longitude": rng.normal(-73.9, 0.1, n), "dropoff_latitude": rng.normal(40.75, 0.1, n),
              "passenger_count": rng.choice(7, n, p=w / w.sum()).astype(np.int64)})
path = os.path.join(tempfile.gettempdir(), "vnm_taxi.csv")
pacsv.write_csv(t, path, write_options=pacsv.WriteOptions(quoting_style="none"))
size = os.path.getsize(path)
q = dict(select=["passenger_count", ["fn", "count_star"], ["fn", "avg", "fare_amount"]], aliases=[None, "n", "m"], group_by=["passenger_count"])
def gpu():
    return planner.execute(q, stream_csv(path, block_size=256 << 20))
def cpu_reader():
    r = pacsv.open_csv(path, read_options=pacsv.ReadOptions(block_size=64 << 20),
                       convert_options=pacsv.ConvertOptions(include_columns=["passenger_count", "fare_amount"]))
    return planner.execute(q, r)
for name, fn in (("device tokeniser + parser (vinum_amd.io.stream_csv)", gpu), ("pyarrow.csv.open_csv (2 columns) + H2D staging", cpu_reader)):
    fn()
    t0 = time.perf_counter(); res = fn(); dt = time.perf_counter() - t0
    print(f"{name}: {dt * 1e3:.0f} ms for {n:.2e} rows / {size / 1e9:.2f} GB of CSV = {size / dt / 1e9:.2f} GB/s, {n / dt / 1e6:.1f} Mrows/s; groups {res.num_rows}")
os.remove(path)
