"""GPU CSV ingest (vnm_csv_parse_block, vinum_amd.io.GpuCsvReader) against pyarrow.csv -- the reader the reference's
stream_csv() / read_csv() delegate to (vinum/io/arrow.py:58-61,106) -- and against Python's float() for the exactness of
the decimal -> float64 conversion (both are correctly rounded, like Arrow's fast_float)."""
import io
import os

import numpy as np
import pyarrow as pa
import pyarrow.csv as pacsv
import pytest

from tests import util

pytestmark = pytest.mark.gpu


def _taxi(n, seed=0):
    rng = np.random.default_rng(seed)
    w = np.array([165, 34808, 7386, 2183, 1016, 3453, 989], float)
    t = pa.table({
        "key": pa.array([f"2009-06-15 17:26:{i % 60:02d}.{i:07d}" for i in range(n)]),
        "fare_amount": pa.array(np.round(rng.lognormal(2.2, 0.6, n), 2), mask=rng.random(n) < 0.01),
        "pickup_longitude": pa.array(rng.normal(-73.9, 0.1, n)),                      # 17 significant digits
        "pickup_latitude": pa.array(rng.normal(40.75, 0.1, n) * rng.choice([1.0, 1e-7, 1e9], n)),
        "passenger_count": pa.array(rng.choice(7, n, p=w / w.sum()).astype(np.int64), mask=rng.random(n) < 0.005),
        "big": pa.array(rng.integers(-2**62, 2**62, n).astype(np.int64)),
    })
    buf = io.BytesIO()
    pacsv.write_csv(t, buf, write_options=pacsv.WriteOptions(quoting_style="none"))
    return buf.getvalue()


def _read_all(reader):
    batches = []
    while True:
        try:
            batches.append(reader.read_next_batch())
        except StopIteration:
            break
    return pa.Table.from_batches(batches)


@pytest.mark.parametrize("block_size", [1 << 16, 1 << 20, 64 << 20])
def test_gpu_csv_reader_equals_pyarrow(block_size, tmp_path):
    from vinum_amd.io import stream_csv
    data = _taxi(120_000)
    path = os.path.join(tmp_path, "taxi.csv")
    with open(path, "wb") as f:
        f.write(data)
    exp = pacsv.read_csv(io.BytesIO(data), read_options=pacsv.ReadOptions(use_threads=False))
    got = _read_all(stream_csv(path, block_size=block_size))
    assert got.schema.names == exp.schema.names
    for name in exp.schema.names:
        a, e = got.column(name).combine_chunks(), exp.column(name).combine_chunks()
        assert a.type == e.type, name
        if pa.types.is_floating(e.type) or pa.types.is_integer(e.type):
            util.assert_col_equal(a, e, name)        # bit-exact, NULLs (empty fields) included
        else:
            assert a.to_pylist() == e.to_pylist(), name


def test_decimal_to_double_is_correctly_rounded():
    """Random decimal strings of every shape the parser accepts: the device result must be the bits of Python's float()
    (correctly rounded); shapes outside its domain must raise the fallback flag instead of producing a value."""
    import ctypes
    from vinum_amd import _lib as L
    rng = np.random.default_rng(4)
    strs = []
    for _ in range(60_000):
        nd = int(rng.integers(1, 20))
        digits = "".join(str(d) for d in rng.integers(0, 10, nd))
        dot = int(rng.integers(0, nd + 1))
        s = digits[:dot] + ("." + digits[dot:] if dot < nd or rng.random() < 0.2 else "")
        if s.startswith("."):
            s = ("0" if rng.random() < 0.5 else "") + s
        if rng.random() < 0.3:
            s += rng.choice(["e", "E"]) + rng.choice(["", "+", "-"]) + str(int(rng.integers(0, 12)))
        if rng.random() < 0.4:
            s = rng.choice(["-", "+"]) + s
        strs.append(s)
    strs += ["0", "-0", "0.0", "-0.0", "1", "9007199254740993", "9007199254740992.5", "0.1", "1e19", "1e-19", "123456789012345678.9",
             "4.35", "2.675", "1.0000000000000002", "8.41e-5", "5e-1", "179769313486231570e-17", "00012.5000", ".5", "5."]
    ok = [s for s in strs if sum(ch.isdigit() for ch in s.split("e")[0].split("E")[0].lstrip("+-").lstrip("0").replace(".", "")) <= 19]
    text = ("x\n" + "\n".join(ok) + "\n").encode()
    lib = L.lib()
    out = (L.DCol * 1)()
    n_rows = ctypes.c_int64(0)
    fb = (ctypes.c_int * 3)()
    L.check(lib.vnm_csv_parse_block(text, len(text), 1, ord(","), 1, 1, (ctypes.c_int * 1)(0), (ctypes.c_int * 1)(L.F64), out,
                                    ctypes.byref(n_rows), fb, None))
    assert n_rows.value == len(ok)
    vals = np.empty(len(ok), np.float64)
    L.check(lib.vnm_memcpy_d2h(vals.ctypes.data, out[0].values, vals.nbytes))
    lib.vnm_free_column(ctypes.byref(out[0]))
    exp = np.array([float(s) for s in ok])
    in_domain = np.array([abs(_dec_exp(s)) <= 19 for s in ok])
    bad = np.nonzero((vals.view(np.uint64) != exp.view(np.uint64)) & in_domain)[0]
    assert bad.size == 0, f"{bad.size} differ, e.g. " + "; ".join(f"{ok[i]!r}: device {vals[i]!r} vs float() {exp[i]!r}" for i in bad[:12])
    assert bool(fb[0]) == bool((~in_domain).any())


def _dec_exp(s):
    """decimal exponent q of the string's significand-as-integer form w * 10^q"""
    m = s.lstrip("+-")
    e = 0
    for sep in ("e", "E"):
        if sep in m:
            m, ex = m.split(sep)
            e = int(ex)
    frac = len(m.split(".")[1]) if "." in m else 0
    return e - frac


def test_fallbacks_go_to_pyarrow(tmp_path):
    """Quoted fields, 'nan' tokens, 20+ digit numbers, CRLF line ends: every block / column the device parser declines is read
    by pyarrow instead -- same table either way."""
    from vinum_amd.io import stream_csv
    rows = ["a,b,c,s"] + [f"{i},{i * 0.25},{i * 3 - 7},txt{i % 5}" for i in range(5000)]
    rows[100] = '100,nan,293,txt0'
    rows[200] = '200,0.123456789012345678901234,593,"quoted, text"'
    rows[300] = "300,,893,"
    data = ("\r\n".join(rows) + "\r\n").encode()
    path = os.path.join(tmp_path, "odd.csv")
    with open(path, "wb") as f:
        f.write(data)
    exp = pacsv.read_csv(io.BytesIO(data), read_options=pacsv.ReadOptions(use_threads=False))
    got = _read_all(stream_csv(path, block_size=1 << 14))
    assert got.schema == exp.schema
    for name in exp.schema.names:
        util.assert_col_equal(got.column(name).combine_chunks(), exp.column(name).combine_chunks(), name) \
            if not pa.types.is_string(exp.schema.field(name).type) else None
        if pa.types.is_string(exp.schema.field(name).type):
            assert got.column(name).to_pylist() == exp.column(name).to_pylist()


def test_config4_stream_csv_group_by_on_the_gpu(tmp_path):
    """BASELINE configs[0] / [3] shape: SELECT passenger_count, count(*), avg(fare_amount) FROM stream_csv(...) GROUP BY
    passenger_count -- only the two needed fields of each row are parsed, on the device."""
    from vinum_amd import planner
    from vinum_amd.io import stream_csv
    data = _taxi(200_000, seed=3)
    path = os.path.join(tmp_path, "taxi.csv")
    with open(path, "wb") as f:
        f.write(data)
    q = dict(select=["passenger_count", ["fn", "count_star"], ["fn", "avg", "fare_amount"]], aliases=[None, "n", "m"],
             group_by=["passenger_count"])
    got = planner.execute(q, stream_csv(path, block_size=4 << 20)).sort_by("passenger_count")
    t = pacsv.read_csv(io.BytesIO(data))
    exp = t.group_by("passenger_count", use_threads=False).aggregate([([], "count_all"), ("fare_amount", "mean")]).sort_by("passenger_count")
    assert got.column("passenger_count").to_pylist() == exp.column("passenger_count").to_pylist()
    assert got.column("n").cast(pa.int64()).to_pylist() == exp.column("count_all").to_pylist()
    # AVG = (exactly rounded sum) / count: at most 1 ULP from math.fsum / count (the division rounds once more).  pyarrow's mean
    # sums sequentially in float64, so it is only an order-dependent approximation of the same number.
    import math
    pcv = t.column("passenger_count").combine_chunks()
    pcn = pcv.fill_null(-1).to_numpy(zero_copy_only=False)          # the NULL key is a group of its own
    fv = t.column("fare_amount").combine_chunks()
    fare, fvalid = fv.fill_null(0.0).to_numpy(zero_copy_only=False), np.array(fv.is_valid())
    exact = []
    for k in exp.column("passenger_count").to_pylist():
        m = (pcn == (-1 if k is None else k)) & fvalid
        exact.append(math.fsum(fare[m].tolist()) / max(int(m.sum()), 1))
    exact = np.array(exact)
    a = np.array(got.column("m").to_pylist(), float)
    assert (util._ulp_diff(a, exact) <= 1).all()
    b = np.array(exp.column("fare_amount_mean").to_pylist(), float)
    assert (np.abs(a - exact) <= np.abs(b - exact) + np.spacing(np.abs(exact))).all()   # never further from the exact mean than pyarrow


def test_more_than_16_numeric_columns(tmp_path):
    """the library parses at most 16 columns per call; a wide file must be parsed in chunks, not rejected."""
    from vinum_amd.io import stream_csv
    rng = np.random.default_rng(5)
    ncol, n = 23, 3000
    cols = {f"c{j}": (rng.integers(-10**6, 10**6, n) if j % 2 else np.round(rng.normal(0, 100, n), 3)) for j in range(ncol)}
    lines = [",".join(cols)] + [",".join(repr(cols[c][i].item()) for c in cols) for i in range(n)]
    data = ("\n".join(lines) + "\n").encode()
    path = os.path.join(tmp_path, "wide.csv")
    with open(path, "wb") as f:
        f.write(data)
    exp = pacsv.read_csv(io.BytesIO(data), read_options=pacsv.ReadOptions(use_threads=False))
    got = _read_all(stream_csv(path, block_size=1 << 16))
    assert got.schema == exp.schema and got.num_rows == exp.num_rows
    for name in exp.schema.names:
        util.assert_col_equal(got.column(name).combine_chunks(), exp.column(name).combine_chunks(), name)


@pytest.mark.parametrize("ncol", [1, 3])
def test_empty_lines_are_not_rows(tmp_path, ncol):
    """pyarrow skips empty lines (ignore_empty_lines); a block holding one must come out with pyarrow's row count --
    in a one-column file too, where an empty line is not even ragged."""
    from vinum_amd.io import stream_csv
    names = ["a", "b", "c"][:ncol]
    rows = [",".join(str(i * (j + 2)) for j in range(ncol)) for i in range(4000)]
    rows.insert(1234, "")
    rows.insert(3000, "")
    data = (",".join(names) + "\n" + "\n".join(rows) + "\n\n").encode()     # ... and a trailing empty line
    path = os.path.join(tmp_path, "blank.csv")
    with open(path, "wb") as f:
        f.write(data)
    exp = pacsv.read_csv(io.BytesIO(data), read_options=pacsv.ReadOptions(use_threads=False))
    assert exp.num_rows == 4000
    reader = stream_csv(path, block_size=1 << 13)
    got = _read_all(reader)
    assert got.num_rows == exp.num_rows
    for name in exp.schema.names:
        util.assert_col_equal(got.column(name).combine_chunks(), exp.column(name).combine_chunks(), name)


def test_quoted_fields_stay_on_the_device(tmp_path):
    """Round 4 (the remainder of SURVEY 8 f2 named in VERDICT r03 missing #5, in part): rows with QUOTED fields -- string columns with
    delimiters inside the quotes, escaped quotes, quoted numbers, quoted empty fields -- are tokenised on the device: the numeric
    columns of such blocks are parsed there (no whole-block fallback), quoted numbers from between their quotes; equal to
    pyarrow.csv.read_csv (vinum/io/arrow.py:58-61,106 delegates to it).  A newline INSIDE a quoted value still sends the block to
    pyarrow."""
    import ctypes
    from vinum_amd import _lib as L
    from vinum_amd.io import stream_csv
    rng = np.random.default_rng(5)
    rows = ["id,city,fare,n,note"]
    for i in range(20_000):
        city = ['"New York, NY"', 'Berlin', '"say ""hi"", ok"', '""', '"a,b,c,,"', 'plain'][i % 6]
        fare = [f"{i * 0.125}", f'"{i * 0.5}"', '""', "", f'"{-i}.25"'][i % 5]
        n = [f"{i}", f'"{i * 7}"'][i % 2]
        rows.append(f'{i},{city},{fare},{n},"x,{i}"')
    data = ("\n".join(rows) + "\n").encode()
    path = os.path.join(tmp_path, "quoted.csv")
    with open(path, "wb") as f:
        f.write(data)
    exp = pacsv.read_csv(io.BytesIO(data), read_options=pacsv.ReadOptions(use_threads=False))
    assert exp.schema.field("fare").type == pa.float64() and exp.schema.field("n").type == pa.int64()
    L.lib().vnm_set_profiling(1)
    got = _read_all(stream_csv(path, block_size=1 << 16))
    ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
    L.lib().vnm_profile_query(b"csv_parse", ctypes.byref(ms), ctypes.byref(cnt))
    L.lib().vnm_set_profiling(0)
    assert cnt.value >= 1                                   # the block went through the device parser
    assert got.schema == exp.schema
    for name in exp.schema.names:
        if pa.types.is_string(exp.schema.field(name).type):
            assert got.column(name).to_pylist() == exp.column(name).to_pylist(), name
        else:
            util.assert_col_equal(got.column(name).combine_chunks(), exp.column(name).combine_chunks(), name)
    # only the numeric columns: nothing of these blocks is parsed by pyarrow at all
    got2 = _read_all(stream_csv(path, columns=["fare", "n", "id"], block_size=1 << 16))
    for name in ("fare", "n", "id"):
        util.assert_col_equal(got2.column(name).combine_chunks(), exp.column(name).combine_chunks(), name)
    # a newline inside a quoted value: the device parser declines the block (a row ends inside a quote), pyarrow reads it -- whatever
    # pyarrow makes of such a file (a table, or ArrowInvalid), this reader makes the same
    bad = b'a,b\n1,"two\nlines"\n2,x\n'
    p2 = os.path.join(tmp_path, "bad.csv")
    with open(p2, "wb") as f:
        f.write(bad)
    try:
        want = pacsv.read_csv(io.BytesIO(bad), read_options=pacsv.ReadOptions(use_threads=False))
    except pa.ArrowInvalid:
        want = None
    if want is None:
        with pytest.raises(pa.ArrowInvalid):
            _read_all(stream_csv(p2))
    else:
        assert _read_all(stream_csv(p2)).to_pydict() == want.to_pydict()


# ---- round 5: string / date32 / timestamp columns on the device (vnm_csv_parse_block_ex) ------------------------------------------
_CITIES = ["Berlin", "Zürich", "São Paulo", '"New York, NY"', "", '""', "東京", "plain text with spaces", '"a,b,c,,"', "x", "Ḽơᶉëᶆ ȋṕšᶙṁ"]


def _mixed(n, seed=0, odd=False):
    """id, city (utf8: non-ASCII, quoted with delimiters, empty), day (date32), ts (timestamp[s]: ' ' / 'T' / date only / hh:mm), tsn
    (timestamp[ns]), amount (float64), n (int64, NULLs).  odd: also rows only pyarrow parses (escaped quotes, zone-less oddities)."""
    rng = np.random.default_rng(seed)
    rows = ["id,city,day,ts,tsn,amount,n"]
    days = rng.integers(-25567, 47482, n)           # 1900-01-01 .. 2100-01-01
    secs = rng.integers(-2_208_988_800, 4_102_444_800, n)
    for i in range(n):
        city = _CITIES[int(rng.integers(0, len(_CITIES)))] if i % 7 else f"town{int(rng.integers(0, 5000))}"
        if odd and i % 1000 == 999:
            city = '"say ""hi"", ok"'
        day = "" if i % 53 == 0 else str(np.datetime64(int(days[i]), "D"))
        t = np.datetime64(int(secs[i]), "s")
        ts = str(t)                                                       # 'YYYY-MM-DDThh:mm:ss'
        k = i % 5
        if k == 1:
            ts = ts.replace("T", " ")
        elif k == 2:
            ts = ts[:10]
        elif k == 3:
            ts = ts[:16].replace("T", " ")
        elif k == 4 and i % 35 == 4:
            ts = ""
        nd = int(rng.integers(0, 10))
        tsn = str(np.datetime64(int(secs[i]) % 4_000_000_000, "s")).replace("T", " ")
        if nd:
            tsn += "." + "".join(str(d) for d in rng.integers(0, 10, nd))
        if i == 0:
            tsn = "2009-06-15 17:26:21.123456789"                         # (so that the probe infers nanoseconds)
        amount = f"{rng.normal(10, 3):.3f}"
        cnt = "" if i % 97 == 0 else str(int(rng.integers(0, 7)))
        rows.append(f"{i},{city},{day},{ts},{tsn},{amount},{cnt}")
    return ("\n".join(rows) + "\n").encode()


def _same_table(got, exp):
    assert got.schema == exp.schema
    assert got.num_rows == exp.num_rows
    for name in exp.schema.names:
        a, e = got.column(name).combine_chunks(), exp.column(name).combine_chunks()
        if pa.types.is_string(e.type):
            assert a.to_pylist() == e.to_pylist(), name
        elif pa.types.is_temporal(e.type):
            assert a.null_count == e.null_count, name
            st = pa.int32() if pa.types.is_date32(e.type) else pa.int64()
            assert a.cast(st).to_pylist() == e.cast(st).to_pylist(), name
        else:
            util.assert_col_equal(a, e, name)


@pytest.mark.parametrize("block_size", [1 << 15, 1 << 20, 64 << 20])
def test_string_date_and_timestamp_columns_stay_on_the_device(block_size, tmp_path, monkeypatch):
    """VERDICT r04 missing #4 (the remainder of SURVEY 8 f2): utf8, date32, timestamp[s] and timestamp[ns] columns are parsed by
    vnm_csv_parse_block_ex -- equal to pyarrow.csv.read_csv (what vinum/io/arrow.py:58-61,106 delegates to), and pyarrow is never
    asked for a column of these blocks."""
    from vinum_amd import io as vio
    data = _mixed(60_000, seed=1)
    path = os.path.join(tmp_path, "mixed.csv")
    with open(path, "wb") as f:
        f.write(data)
    exp = pacsv.read_csv(io.BytesIO(data), read_options=pacsv.ReadOptions(use_threads=False))
    assert [str(f.type) for f in exp.schema] == ["int64", "string", "date32[day]", "timestamp[s]", "timestamp[ns]", "double", "int64"]

    def no_host(self, text, names):
        raise AssertionError(f"pyarrow was asked to parse {names}")
    monkeypatch.setattr(vio.GpuCsvReader, "_host_parse", no_host)
    got = _read_all(vio.stream_csv(path, block_size=block_size))
    _same_table(got, exp)


def test_string_and_time_fields_the_device_declines_go_to_pyarrow(tmp_path):
    """Escaped quotes inside a quoted string, zone-less oddities (hour only), leading blanks: the column of THAT block is parsed by
    pyarrow, with the same running dictionary -- the table is pyarrow's either way."""
    from vinum_amd.io import stream_csv
    data = _mixed(30_000, seed=2, odd=True)
    lines = data.split(b"\n")
    for i in (1500, 9000, 20_000):
        f = lines[i].split(b",")
        if len(f) == 7:
            f[3] = b"2020-01-01 10"              # ISO 8601 hour only: Arrow reads it, the device parser declines
            f[2] = b" 2020-03-04"                # Arrow trims blanks of non-string fields
            lines[i] = b",".join(f)
    data = b"\n".join(lines)
    path = os.path.join(tmp_path, "odd.csv")
    with open(path, "wb") as f:
        f.write(data)
    exp = pacsv.read_csv(io.BytesIO(data), read_options=pacsv.ReadOptions(use_threads=False))
    assert exp.schema.field("ts").type == pa.timestamp("s") and exp.schema.field("day").type == pa.date32()
    reader = stream_csv(path, block_size=1 << 16)
    _same_table(_read_all(reader), exp)
    # the blocks whose string column went to pyarrow were encoded by the SAME dictionary, and no value entered its host copy twice
    # (a block that is not encoded on the device must not hand the previous block's new values over again)
    vals = pa.concat_arrays(reader._dicts["city"]._chunks).to_pylist()       # (the values as the encodes handed them over, in order)
    assert len(vals) == len(set(vals)) == len(set(exp.column("city").to_pylist()))


def test_calendar_arithmetic_against_numpy():
    """Every accepted date / timestamp spelling over the whole proleptic Gregorian range 0001 .. 9999 (leap years, century rules,
    negative epochs): bit-equal to numpy.datetime64; impossible dates and times raise the fallback flag, never a value."""
    import ctypes
    from vinum_amd import _lib as L
    rng = np.random.default_rng(11)
    n = 200_000
    days = rng.integers(-719162, 2932896, n)                       # 0001-01-01 .. 9999-12-31
    days[:6] = [-719162, 2932896, 0, -1, 11016, -25509]             # ... 2000-02-29, 1900-02-28
    sod = rng.integers(0, 86400, n)
    d = days.astype("datetime64[D]")
    text_d = [str(x) for x in d]
    ts = (days.astype(np.int64) * 86400 + sod)
    text_s = [f"{a}{'T' if i % 2 else ' '}{s // 3600:02d}:{s // 60 % 60:02d}:{s % 60:02d}" for i, (a, s) in enumerate(zip(text_d, sod.tolist()))]
    body = "\n".join(f"{a},{b}" for a, b in zip(text_d, text_s))
    text = ("d,t\n" + body + "\n").encode()
    lib = L.lib()

    def parse(text, kinds):
        k = len(kinds)
        out = (L.DCol * k)()
        n_rows = ctypes.c_int64(0)
        fb = (ctypes.c_int * (k + 2))()
        L.check(lib.vnm_csv_parse_block_ex(text, len(text), 1, ord(","), k, k, (ctypes.c_int * k)(*range(k)), (ctypes.c_int * k)(*kinds), None, out,
                                           ctypes.byref(n_rows), fb, None))
        cols = []
        for i, kind in enumerate(kinds):
            a = np.empty(n_rows.value, np.int32 if kind == L.CSV_DATE32 else np.int64)
            if a.nbytes:
                L.check(lib.vnm_memcpy_d2h(a.ctypes.data, out[i].values, a.nbytes))
            bits = np.empty((n_rows.value + 7) // 8, np.uint8)
            if bits.nbytes:
                L.check(lib.vnm_memcpy_d2h(bits.ctypes.data, out[i].validity, bits.nbytes))
            cols.append((a, np.unpackbits(bits, bitorder="little")[:n_rows.value].astype(bool)))
            lib.vnm_free_column(ctypes.byref(out[i]))
        return cols, list(fb)

    (dd, tt), fb = parse(text, [L.CSV_DATE32, L.CSV_TIMESTAMP_S])
    assert fb == [0, 0, 0, 0]
    assert dd[1].all() and tt[1].all()
    assert (dd[0] == days).all()
    assert (tt[0] == ts).all()
    # nanoseconds: 1678 .. 2261, one to nine fractional digits
    m = 50_000
    secs = rng.integers(-9_200_000_000, 9_200_000_000, m)          # 1678-06 .. 2261-07
    lines = []
    exp = np.empty(m, np.int64)
    for i in range(m):
        nd = int(rng.integers(0, 10))
        digits = "".join(str(x) for x in rng.integers(0, 10, nd))
        s = str(np.datetime64(int(secs[i]), "s")).replace("T", " ") + ("." + digits if nd else "")
        lines.append(s)
        exp[i] = int(secs[i]) * 10**9 + (int(digits) * 10**(9 - nd) if nd else 0)
    text = ("t\n" + "\n".join(lines) + "\n").encode()
    (nn,), fb = parse(text, [L.CSV_TIMESTAMP_NS])
    assert fb == [0, 0, 0] and nn[1].all()
    assert (nn[0] == exp).all()
    # never a guessed value
    for bad in ["2021-02-29", "1900-02-29", "2020-13-01", "2020-00-10", "2020-04-31", "2020-01-00", "2020-1-01", "20200101", "2020-01-01x", "abcd-01-01"]:
        ((_, v),), fb = parse(f"d\n2020-01-01\n{bad}\n".encode(), [L.CSV_DATE32])
        assert fb[0] == 1 and not v[1], bad
    for bad in ["2020-01-01 24:00:00", "2020-01-01 10:60:00", "2020-01-01 10:11:60", "2020-01-01 10", "2020-01-01 10:11:12Z", "2020-01-01 10:11:12+01:00",
                "2020-01-01 10:11:12.5", "2020-01-01_10:11:12", " 2020-01-01"]:
        ((_, v),), fb = parse(f"t\n2020-01-01 00:00:00\n{bad}\n".encode(), [L.CSV_TIMESTAMP_S])
        assert fb[0] == 1 and not v[1], bad
    for bad in ["2020-01-01 10:11:12.", "2020-01-01 10:11:12.1234567891", "1677-12-31 00:00:00", "2262-01-01 00:00:00", "2020-01-01 10:11:12.12a"]:
        ((_, v),), fb = parse(f"t\n2020-01-01 00:00:00\n{bad}\n".encode(), [L.CSV_TIMESTAMP_NS])
        assert fb[0] == 1 and not v[1], bad
    # the empty field is NULL
    ((_, v), (_, w)), fb = parse(b"i,d\n1,2020-01-01\n2,\n3,2020-01-03\n", [L.I64, L.CSV_DATE32])
    assert fb == [0, 0, 0, 0] and w.tolist() == [True, False, True]


def test_invalid_utf8_is_not_accepted(tmp_path):
    """pyarrow checks string columns (check_utf8) and raises; the device parser flags the column, so the reader raises the same way."""
    from vinum_amd.io import stream_csv
    rows = [b"id,name"] + [b"%d,ok%d" % (i, i) for i in range(200_000)]      # (beyond the 1 MB the schema is inferred from)
    for bad in (b"\xff", b"\xc3", b"\xe0\x80\x80", b"\xed\xa0\x80", b"\xf4\x90\x80\x80", b"\xc0\xaf"):
        r = list(rows)
        r[150_000] = b"150000,bad" + bad + b"x"
        data = b"\n".join(r) + b"\n"
        path = os.path.join(tmp_path, "bad_utf8.csv")
        with open(path, "wb") as f:
            f.write(data)
        with pytest.raises(pa.ArrowInvalid):      # the streaming reader the reference uses: types from the first block, then strict
            for _ in pacsv.open_csv(io.BytesIO(data), read_options=pacsv.ReadOptions(block_size=1 << 20)):
                pass
        reader = stream_csv(path, block_size=1 << 20)
        assert reader.schema.field("name").type == pa.string()
        with pytest.raises(pa.ArrowInvalid):
            _read_all(reader)
    # ... and every well-formed sequence length is taken
    good = "a,b\n1,é\n2,€\n3,😀\n4,߿ࠀ￿\U00010000\U0010ffff\n".encode()
    path = os.path.join(tmp_path, "good_utf8.csv")
    with open(path, "wb") as f:
        f.write(good)
    _same_table(_read_all(stream_csv(path)), pacsv.read_csv(io.BytesIO(good)))


def test_group_by_a_string_column_of_a_csv_stream(tmp_path, monkeypatch):
    """SELECT city, count(*), sum(n), min(day), max(ts) FROM stream_csv(...) GROUP BY city: the string key reaches the aggregate as
    dictionary codes that were never an Arrow string array on the host."""
    from vinum_amd import io as vio
    from vinum_amd import planner
    data = _mixed(80_000, seed=5)
    path = os.path.join(tmp_path, "mixed.csv")
    with open(path, "wb") as f:
        f.write(data)

    def no_host(self, text, names):
        raise AssertionError(f"pyarrow was asked to parse {names}")
    monkeypatch.setattr(vio.GpuCsvReader, "_host_parse", no_host)
    q = dict(select=["city", ["fn", "count_star"], ["fn", "sum", "n"], ["fn", "min", "day"], ["fn", "max", "ts"]], aliases=[None, "c", "s", "d", "t"],
             group_by=["city"])
    got = planner.execute(q, vio.stream_csv(path, block_size=1 << 20)).sort_by("city")
    t = pacsv.read_csv(io.BytesIO(data))
    exp = t.group_by("city", use_threads=False).aggregate([([], "count_all"), ("n", "sum"), ("day", "min"), ("ts", "max")]).sort_by("city")
    assert got.column("city").to_pylist() == exp.column("city").to_pylist()
    assert got.column("c").cast(pa.int64()).to_pylist() == exp.column("count_all").to_pylist()
    assert [None if x is None else int(x) for x in got.column("s").to_pylist()] == exp.column("n_sum").to_pylist()
    assert got.column("d").cast(pa.date32()).to_pylist() == exp.column("day_min").to_pylist()
    assert got.column("t").cast(pa.timestamp("s")).to_pylist() == exp.column("ts_max").to_pylist()


def test_boolean_and_time_columns_on_the_device(tmp_path, monkeypatch):
    """bool (pyarrow's true_values / false_values spellings, NULLs) and time32[s] (hh:mm[:ss]) columns are parsed on the device; a
    spelling the device declines (a blank-padded value) sends that column of that block to pyarrow and the codes still agree."""
    from vinum_amd import io as vio
    rng = np.random.default_rng(8)
    n = 40_000
    tv, fv = ["true", "True", "TRUE", "1"], ["false", "False", "FALSE", "0"]
    rows = ["id,flag,at,flag2"]
    for i in range(n):
        b = bool(rng.integers(0, 2))
        flag = "" if i % 41 == 0 else (tv if b else fv)[int(rng.integers(0, 4))]
        s = int(rng.integers(0, 86400))
        at = "" if i % 59 == 0 else (f"{s // 3600:02d}:{s // 60 % 60:02d}:{s % 60:02d}" if i % 3 else f"{s // 3600:02d}:{s // 60 % 60:02d}")
        rows.append(f"{i},{flag},{at},{'false' if i % 2 else 'true'}")
    data = ("\n".join(rows) + "\n").encode()
    path = os.path.join(tmp_path, "flags.csv")
    with open(path, "wb") as f:
        f.write(data)
    exp = pacsv.read_csv(io.BytesIO(data), read_options=pacsv.ReadOptions(use_threads=False))
    assert [str(f.type) for f in exp.schema] == ["int64", "bool", "time32[s]", "bool"]
    calls = []
    orig = vio.GpuCsvReader._host_parse
    monkeypatch.setattr(vio.GpuCsvReader, "_host_parse", lambda self, text, names: (calls.append(list(names)), orig(self, text, names))[1])
    got = _read_all(vio.stream_csv(path, block_size=1 << 16))
    assert not calls, calls
    assert got.schema == exp.schema
    for name in exp.schema.names:
        assert got.column(name).to_pylist() == exp.column(name).to_pylist(), name
    # a value pyarrow takes and the device declines (Arrow trims blanks around non-string fields)
    lines = data.split(b"\n")
    f = lines[30_000].split(b",")
    f[1] = b" true"
    f[2] = b" 01:02:03"
    lines[30_000] = b",".join(f)
    data2 = b"\n".join(lines)
    with open(path, "wb") as fh:
        fh.write(data2)
    try:
        exp2 = pacsv.read_csv(io.BytesIO(data2), read_options=pacsv.ReadOptions(use_threads=False))
    except pa.ArrowInvalid:
        exp2 = None
    if exp2 is not None and exp2.schema == exp.schema:
        got2 = _read_all(vio.stream_csv(path, block_size=1 << 16))
        assert calls, "the padded fields should have gone to pyarrow"
        for name in exp2.schema.names:
            assert got2.column(name).to_pylist() == exp2.column(name).to_pylist(), name


@pytest.mark.parametrize("seed", range(int(os.environ.get("VNM_FUZZ_SEEDS", "0")) or 32))
def test_csv_fuzz_vs_pyarrow(seed, tmp_path):
    """Random CSV files -- columns of every kind the device parses (int64, float64 in many spellings, utf8 with quotes / delimiters /
    non-ASCII, date32, timestamp[s] / [ns] in every accepted spelling, bool, time32) plus fields only pyarrow understands (NA tokens,
    blanks, escaped quotes, exponents, signs), random NULL shares and block sizes: the reader's table is pyarrow's, or both raise."""
    from vinum_amd.io import stream_csv
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.integers(200, 6000))
    kinds = [str(k) for k in rng.choice(["int", "float", "str", "date", "ts", "tsn", "bool", "time", "mixed"], int(rng.integers(2, 9)))]

    def field(kind, i):
        if rng.random() < 0.04:
            return ""
        odd = rng.random() < 0.01
        if kind == "int":
            v = int(rng.integers(-10**12, 10**12))
            return (f"+{v}" if v > 0 else f" {v}") if odd else str(v)
        if kind == "float":
            v = float(rng.normal(0, 1000))
            if odd:
                return str(rng.choice(["nan", "NA", "1e5", "-inf", " 2.5", "1_000.5", "0x10"]))
            return [f"{v:.3f}", f"{v:.0f}", repr(v), f"{v:.2e}", f"{v:.17g}"][int(rng.integers(0, 5))]
        if kind == "str":
            w = str(rng.choice(["a", "Berlin", "São Paulo", "x y", "東京", "", "N/A", "1", "true"]))
            if odd:
                return '"say ""hi"""'
            return f'"{w},{i % 7}"' if rng.random() < 0.2 else w
        if kind == "date":
            d = str(np.datetime64(int(rng.integers(-40000, 60000)), "D"))
            return (d.replace("-", "/") if rng.random() < 0.5 else " " + d) if odd else d
        if kind in ("ts", "tsn"):
            t = str(np.datetime64(int(rng.integers(-2_000_000_000, 4_000_000_000)), "s"))
            sp = int(rng.integers(0, 4))
            out = [t, t.replace("T", " "), t[:16].replace("T", " "), t[:10]][sp]
            if kind == "tsn" and sp < 2 and rng.random() < 0.7:
                out += "." + "".join(str(x) for x in rng.integers(0, 10, int(rng.integers(1, 10))))
            if odd:
                out = str(rng.choice([t + "Z", t[:13], t + "+01:00", "2021-02-30 00:00:00"]))
            return out
        if kind == "bool":
            return str(rng.choice(["yes", "T", " true"])) if odd else str(rng.choice(["true", "false", "True", "FALSE", "1", "0"]))
        if kind == "time":
            s = int(rng.integers(0, 86400))
            t = f"{s // 3600:02d}:{s // 60 % 60:02d}:{s % 60:02d}"
            return (t + ".5" if rng.random() < 0.5 else "24:00:00") if odd else (t if rng.random() < 0.8 else t[:5])
        return str(rng.choice(["12", "x", "2020-01-01", "1.5", "true", ""]))

    rows = [",".join(f"c{j}_{k}" for j, k in enumerate(kinds))]
    for i in range(n):
        rows.append(",".join(field(k, i) for k in kinds))
    data = ("\n".join(rows) + "\n").encode()
    path = os.path.join(tmp_path, "fuzz.csv")
    with open(path, "wb") as f:
        f.write(data)
    try:
        exp = pacsv.read_csv(io.BytesIO(data), read_options=pacsv.ReadOptions(use_threads=False))
    except pa.ArrowInvalid:
        exp = None
    block = int(rng.choice([1 << 12, 1 << 14, 1 << 20]))
    if exp is None:
        with pytest.raises(pa.ArrowInvalid):
            _read_all(stream_csv(path, block_size=block))
        return
    got = _read_all(stream_csv(path, block_size=block))
    assert got.schema == exp.schema, (got.schema, exp.schema)
    assert got.num_rows == exp.num_rows
    for name in exp.schema.names:
        a, e = got.column(name).combine_chunks(), exp.column(name).combine_chunks()
        if pa.types.is_floating(e.type):
            util.assert_col_equal(a, e, name)
        else:
            assert a.to_pylist() == e.to_pylist(), name
