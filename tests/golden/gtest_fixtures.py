"""Known-answer fixtures of the reference's own C++ unit test, transcribed as DATA.

Source of the numbers: /root/reference/vinum_cpp/test/hash_agg_test.cpp
  * test table            CreateTestTable          :155-249
  * overflow table        CreateOverflowTestTable  :251-273
  * empty batch           CreateEmptyTestRecordBatch :275-283
  * expected result batches  Create*ArgFuncs / CreateNoGrp_AggFuncs / CreateEmptyTable_AggFuncs :286-777
  * which operator class runs which case     TEST_F(...) :816-1003
  * protocol: the table is fed as TWO batches (chunksize = rows/2, :111-120), the result is
    sorted ascending by the listed key column(s) (nulls last) and compared exactly (:75-136).

Only values, validity flags, types and the (groupby_cols, agg_cols, agg_funcs) triples are
recorded here.  Round 3: the string / boolean columns of the test table, the GenericHashAggregate
runs of every case (TEST_F Generic_* :816-824, :846-854, :886-894, :916-924, :936-944, :976-984),
the string-keyed case (CreateStringGrp_DoubleArgFuncs :286-338), the boolean-keyed case
(CreateBooleanGrp_DateArgFuncs :601-650, TEST_F :946-954) and the string MIN / MAX case
(CreateInt64Grp_StringArgFuncs :439-477, TEST_F Single_/Multi_/Generic_ :866-894) are transcribed too.

Expected output column NAMES are ours (out_col_name of each AggFuncDef); the gtest's expected
schemas carry different labels, so comparisons are positional.
"""
import numpy as np
import pyarrow as pa

COUNT_STAR, COUNT, MIN, MAX, SUM, AVG = range(6)
ONE_GROUP, SINGLE, MULTI = range(3)
GENERIC = 3      # GenericHashAggregate (vinum/core/vinum_lib.cpp:92-109): not a kind of the C ABI, a class of vinum_amd.vinum_lib


def _arr(vals, valid, t):
    vals = list(vals)
    mask = np.array([not v for v in valid], dtype=bool)
    storage = {
        "int8": np.int8, "int64": np.int64, "uint64": np.uint64, "float64": np.float64, "float32": np.float32,
        "date64": np.int64, "time32ms": np.int32, "timestamp_ms": np.int64,
    }[t]
    a = pa.array(np.array(vals, dtype=storage), mask=mask if len(vals) else None)
    target = {"date64": pa.date64(), "time32ms": pa.time32("ms"), "timestamp_ms": pa.timestamp("ms")}.get(t)
    return a.view(target) if target is not None else a


T, F = True, False
ALL8 = [T] * 8


def _str(vals, valid):
    return pa.array([v if ok else None for v, ok in zip(vals, valid)], type=pa.string())


def test_table() -> pa.Table:
    """hash_agg_test.cpp:155-249"""
    cols = {
        "id": _arr([1, 2, 3, 4, 5, 6, 7, 8], ALL8, "int64"),
        "timestamp_int64": _arr([1602127614, 1602217613, 1602304012, 1602390411, 0, 1602563209, 0, 1602736007],
                                [T, T, T, T, F, T, F, T], "int64"),
        "date": _str(["", "2020-10-09T04:26:53", "2020-10-10T04:26:52", "2020-10-11T04:26:51", "2020-10-12T04:26:50",
                      "2020-10-13T04:26:49", "0", "2020-10-15T04:26:47"], [F, T, T, T, T, T, F, T]),
        "is_vendor": pa.array([v if ok else None for v, ok in zip([True, True, False, False, True, False, False, False],
                                                                  [T, T, T, F, T, F, F, F])], type=pa.bool_()),
        "city_from": _str(["", "Munich", "", "San Francisco", "Berlin", "Munich", "Berlin", "Berlin"], [F, T, F, T, T, T, T, T]),
        "city_to": _str(["Munich", "Riva", "Naples", "Naples", "Riva", "Riva", "Munich", "Munich"], ALL8),
        "name": _str(["Joe", "", "Joseph", "Joseph", "", "Jonas", "Joseph", "Joe"], [T, F, T, T, F, T, T, T]),
        "lat": _arr([52.51, 48.51, 44.89, 42.89, 44.89, 48.51, 44.89, 52.51], ALL8, "float64"),
        "lng": _arr([13.66, 12.3, 14.23, 15.89, 14.23, 12.3, 14.23, 13.66], ALL8, "float64"),
        "total": _arr([0, 143.15, 33.4, 53.1, 0, 0, 33.4, 0], [F, T, T, T, F, F, T, F], "float64"),
        "grp_int8": _arr([0, 2, 7, 3, 1, 2, 1, 1], [F, T, F, T, T, T, T, T], "int8"),
        "grp_neg_int8": _arr([0, -1, -1, 3, 1, -1, 1, 1], [F, T, F, T, T, T, T, T], "int8"),
        "date64": _arr([1611664426519, 1611664426386, 1611664426519, 1611664416382,
                        1611664416382, 1611664426519, 1611664416382, 1611664426386],
                       [F, T, T, T, F, T, T, T], "date64"),
        "time32": _arr([130, 7, 41, 7, 41, 130, 7, 130], [F, T, F, T, T, T, F, T], "time32ms"),
        "timestamp": _arr([1611664420588, 1611663913570, 1611663913570, 1611664414385,
                           1611664420588, 130, 1611664420588, 1611664414385],
                          [T, T, F, T, T, F, F, T], "timestamp_ms"),
        "grp_neg_int64": _arr([-9223372036854775807, -9223372036854775806, 9223372036854775807,
                               -9223372036854775807, 9223372036854775806, 9223372036854775806,
                               9223372036854775807, -9223372036854775806], ALL8, "int64"),
    }
    return pa.table(cols)


def overflow_table() -> pa.Table:
    """hash_agg_test.cpp:251-273"""
    v = [T, T, T, T, F, T, F, T]
    return pa.table({
        "id": _arr([1, 2, 1, 1, 2, 2, 1, 1], ALL8, "int64"),
        "int_64": _arr([9223372036854775807, 9223372036854775806, 9223372036854775805, 9223372036854775804,
                        9223372036854775803, 9223372036854775802, 9223372036854775801, 9223372036854775799],
                       v, "int64"),
        "uint_64": _arr([18446744073709551615, 18446744073709551614, 18446744073709551613,
                         18446744073709551612, 18446744073709551611, 18446744073709551610,
                         18446744073709551609, 18446744073709551608], v, "uint64"),
    })


def empty_batch() -> pa.RecordBatch:
    """hash_agg_test.cpp:275-283"""
    return pa.RecordBatch.from_arrays([pa.array([], pa.int64())], names=["id"])


def _dec(vals):
    import decimal
    return pa.array([decimal.Decimal(v) for v in vals], type=pa.decimal128(38, 0))


# Each case: table, operator kinds the gtest runs it with, groupby/agg cols, funcs, sort columns,
# expected columns (positional, sorted by key ascending with nulls last).
CASES = {
    # CreateDoubleGrp_IntArgFuncs :344-390; TEST_F Single_/Multi_DoubleGrp_IntArgFuncs :826-844
    "double_grp__int_arg_funcs": dict(
        table="test", kinds=[SINGLE, MULTI, GENERIC], groupby=["lat"], agg_cols=["lat"],
        funcs=[(COUNT_STAR, "", "count"), (MIN, "id", "min_0"), (MAX, "id", "max_0"),
               (SUM, "id", "sum_0"), (AVG, "id", "avg_0")],
        sort_cols=[0],
        expected=[
            _arr([42.89, 44.89, 48.51, 52.51], [T] * 4, "float64"),
            _arr([1, 3, 2, 2], [T] * 4, "uint64"),
            _arr([4, 3, 2, 1], [T] * 4, "int64"),
            _arr([4, 7, 6, 8], [T] * 4, "int64"),
            _arr([4, 15, 8, 9], [T] * 4, "int64"),
            _arr([4.0, 5.0, 4.0, 4.5], [T] * 4, "float64"),
        ]),
    # CreateInt64Grp_IntOverflowArgFuncs :392-444; TEST_F Single_Int64Grp_IntOverflowArgFuncs :856-864
    "int64_grp__int_overflow_arg_funcs": dict(
        table="overflow", kinds=[SINGLE, MULTI], groupby=["id"], agg_cols=["id"],
        funcs=[(SUM, "int_64", "sum_1"), (SUM, "uint_64", "sum_2"), (AVG, "int_64", "avg_1"),
               (AVG, "uint_64", "avg_2")],
        sort_cols=[0],
        expected=[
            _arr([1, 2], [T, T], "int64"),
            _dec(["36893488147419103215", "18446744073709551608"]),
            _dec(["73786976294838206448", "36893488147419103224"]),
            _arr([9.223372036854776e+18, 9.223372036854776e+18], [T, T], "float64"),
            _arr([1.8446744073709552e+19, 1.8446744073709552e+19], [T, T], "float64"),
        ]),
    # CreateInt8Grp_DoubleArgFuncs :492-546; TEST_F Single_/Multi_Int8Grp_DoubleArgFuncs :896-914
    "int8_grp__double_arg_funcs": dict(
        table="test", kinds=[SINGLE, MULTI, GENERIC], groupby=["grp_int8"], agg_cols=["grp_int8"],
        funcs=[(COUNT_STAR, "", "count"), (COUNT, "total", "count_9"), (MIN, "lat", "min_6"),
               (MAX, "lat", "max_6"), (SUM, "lat", "sum_6"), (AVG, "lat", "avg_6")],
        sort_cols=[0],
        expected=[
            _arr([1, 2, 3, 0], [T, T, T, F], "int8"),
            _arr([3, 2, 1, 2], [T] * 4, "uint64"),
            _arr([1, 1, 1, 1], [T] * 4, "uint64"),
            _arr([44.89, 48.51, 42.89, 44.89], [T] * 4, "float64"),
            _arr([52.51, 48.51, 42.89, 52.51], [T] * 4, "float64"),
            _arr([142.29, 97.02, 42.89, 97.4], [T] * 4, "float64"),
            _arr([47.43, 48.51, 42.89, 48.7], [T] * 4, "float64"),
        ]),
    # CreateMultiIntGrp_DateArgFuncs :548-611; TEST_F Multi_MultiIntGrp_DateArgFuncs :926-934
    "multi_int_grp__date_arg_funcs": dict(
        table="test", kinds=[MULTI, GENERIC], groupby=["grp_neg_int8", "date64", "time32", "timestamp"],
        agg_cols=["grp_neg_int8", "date64", "time32", "timestamp"],
        funcs=[(COUNT_STAR, "", "count"), (MIN, "date64", "min_12"), (MAX, "timestamp", "max_14"),
               (SUM, "time32", "sum_13")],
        sort_cols=[0, 1, 2, 3],
        expected=[
            _arr([-1, -1, 1, 1, 1, 3, 0, 0], [T, T, T, T, T, T, F, F], "int8"),
            _arr([1611664426386, 1611664426519, 1611664416382, 1611664426386, 0, 1611664416382,
                  1611664426519, 0], [T, T, T, T, F, T, T, F], "date64"),
            _arr([7, 130, 0, 130, 41, 7, 0, 0], [T, T, F, T, T, T, F, F], "time32ms"),
            _arr([1611663913570, 0, 0, 1611664414385, 1611664420588, 1611664414385, 0, 1611664420588],
                 [T, F, F, T, T, T, F, T], "timestamp_ms"),
            _arr([1] * 8, [T] * 8, "uint64"),
            _arr([1611664426386, 1611664426519, 1611664416382, 1611664426386, 0, 1611664416382,
                  1611664426519, 0], [T, T, T, T, F, T, T, F], "date64"),
            _arr([1611663913570, 0, 0, 1611664414385, 1611664420588, 1611664414385, 0, 1611664420588],
                 [T, F, F, T, T, T, F, T], "timestamp_ms"),
            _arr([7, 130, 0, 130, 41, 7, 0, 0], [T, T, F, T, T, T, F, F], "time32ms"),
        ]),
    # CreateNegInt64Grp_TimestampArgFuncs :652-707; TEST_F Single_/Multi_NegInt64Grp_... :956-974
    "neg_int64_grp__timestamp_arg_funcs": dict(
        table="test", kinds=[SINGLE, MULTI, GENERIC], groupby=["grp_neg_int64"], agg_cols=["grp_neg_int64"],
        funcs=[(COUNT_STAR, "", "count"), (COUNT, "timestamp", "count_ts"), (MIN, "timestamp", "min_14"),
               (MAX, "timestamp", "max_14"), (AVG, "grp_int8", "avg_10"), (AVG, "grp_neg_int8", "avg_11")],
        sort_cols=[0],
        expected=[
            _arr([-9223372036854775807, -9223372036854775806, 9223372036854775806, 9223372036854775807],
                 [T] * 4, "int64"),
            _arr([2, 2, 2, 2], [T] * 4, "uint64"),
            _arr([2, 2, 1, 0], [T] * 4, "uint64"),
            _arr([1611664414385, 1611663913570, 1611664420588, 0], [T, T, T, F], "timestamp_ms"),
            _arr([1611664420588, 1611664414385, 1611664420588, 0], [T, T, T, F], "timestamp_ms"),
            _arr([3.0, 1.5, 1.5, 1.0], [T] * 4, "float32"),
            _arr([3.0, 0, 0, 1.0], [T] * 4, "float32"),
        ]),
    # CreateStringGrp_DoubleArgFuncs :286-338; TEST_F Generic_StringGrp_DoubleArgFuncs :816-824
    "string_grp__double_arg_funcs": dict(
        table="test", kinds=[GENERIC], groupby=["city_from"], agg_cols=["city_from"],
        funcs=[(COUNT_STAR, "", "count"), (COUNT, "total", "count_9"), (MIN, "lat", "min_6"),
               (MAX, "lat", "max_6"), (SUM, "lat", "sum_6"), (AVG, "lat", "avg_6")],
        sort_cols=[0],
        expected=[
            _str(["Berlin", "Munich", "San Francisco", ""], [T, T, T, F]),
            _arr([3, 2, 1, 2], [T] * 4, "uint64"),
            _arr([1, 1, 1, 1], [T] * 4, "uint64"),
            _arr([44.89, 48.51, 42.89, 44.89], [T] * 4, "float64"),
            _arr([52.51, 48.51, 42.89, 52.51], [T] * 4, "float64"),
            _arr([142.29, 97.02, 42.89, 97.4], [T] * 4, "float64"),
            _arr([47.43, 48.51, 42.89, 48.7], [T] * 4, "float64"),
        ]),
    # CreateInt64Grp_StringArgFuncs :439-477; TEST_F Single_/Multi_/Generic_Int64Grp_StringArgFuncs :866-894
    # (COUNT, MIN and MAX of a STRING column: StringMinMaxFunc agg_funcs.h:219-261)
    "int64_grp__string_arg_funcs": dict(
        table="test", kinds=[SINGLE, MULTI, GENERIC], groupby=["id"], agg_cols=["id"],
        funcs=[(COUNT, "date", "count_2"), (MIN, "date", "min_2"), (MAX, "date", "max_2")],
        sort_cols=[0],
        expected=[
            _arr([1, 2, 3, 4, 5, 6, 7, 8], ALL8, "int64"),
            _arr([0, 1, 1, 1, 1, 1, 0, 1], ALL8, "uint64"),
            _str(["", "2020-10-09T04:26:53", "2020-10-10T04:26:52", "2020-10-11T04:26:51", "2020-10-12T04:26:50",
                  "2020-10-13T04:26:49", "", "2020-10-15T04:26:47"], [F, T, T, T, T, T, F, T]),
            _str(["", "2020-10-09T04:26:53", "2020-10-10T04:26:52", "2020-10-11T04:26:51", "2020-10-12T04:26:50",
                  "2020-10-13T04:26:49", "", "2020-10-15T04:26:47"], [F, T, T, T, T, T, F, T]),
        ]),
    # CreateBooleanGrp_DateArgFuncs :601-650; TEST_F BooleanGrp_DateArgFuncs :946-954 (sorted by column 1, the count: Arrow
    # cannot sort booleans).  SUM(time32) keeps time32 (agg_func_factory.cpp:132-137), AVG is float64.
    "boolean_grp__date_arg_funcs": dict(
        table="test", kinds=[GENERIC], groupby=["is_vendor"], agg_cols=["is_vendor"],
        funcs=[(COUNT_STAR, "", "count"), (MIN, "time32", "min_12"), (MAX, "time32", "max_14"),
               (SUM, "time32", "sum_13"), (AVG, "time32", "avg_13")],
        sort_cols=[1],
        expected=[
            pa.array([False, True, None], type=pa.bool_()),
            _arr([1, 3, 4], [T] * 3, "uint64"),
            _arr([0, 7, 7], [F, T, T], "time32ms"),
            _arr([0, 41, 130], [F, T, T], "time32ms"),
            _arr([0, 48, 267], [F, T, T], "time32ms"),
            _arr([0, 24.0, 89.0], [F, T, T], "float64"),
        ]),
    # CreateNoGrp_AggFuncs :709-757; TEST_F NoGrp_AggFuncs :986-993
    "no_grp__agg_funcs": dict(
        table="test", kinds=[ONE_GROUP], groupby=[], agg_cols=[],
        funcs=[(COUNT_STAR, "", "count_star"), (COUNT, "timestamp_int64", "count_int64"),
               (MIN, "timestamp_int64", "min_int64"), (MAX, "timestamp_int64", "max_int64"),
               (SUM, "timestamp_int64", "sum_int64"), (AVG, "timestamp_int64", "avg_int64")],
        sort_cols=[],
        expected=[
            _arr([8], [T], "uint64"), _arr([6], [T], "uint64"), _arr([1602127614], [T], "int64"),
            _arr([1602736007], [T], "int64"), _arr([9614338866], [T], "int64"),
            _arr([1602389811.0], [T], "float64"),
        ]),
    # CreateEmptyTable_AggFuncs :759-777; TEST_F EmptyTable_AggFuncs :995-1003
    "empty_table__agg_funcs": dict(
        table="empty", kinds=[ONE_GROUP], groupby=[], agg_cols=[],
        funcs=[(COUNT_STAR, "", "count_star")], sort_cols=[],
        expected=[_arr([0], [T], "uint64")]),
}


def table_for(case) -> pa.Table:
    return {"test": test_table, "overflow": overflow_table,
            "empty": lambda: pa.Table.from_batches([empty_batch()])}[case["table"]]()


def feed_batches(table: pa.Table):
    """aggregate_and_sort :108-120 -- two halves (chunksize = rows >> 1)."""
    mid = table.num_rows >> 1
    if mid <= 0:
        return table.to_batches() or [pa.RecordBatch.from_arrays(
            [pa.array([], f.type) for f in table.schema], names=table.schema.names)]
    return table.to_batches(max_chunksize=mid)


def sort_result(batch: pa.RecordBatch, sort_cols) -> pa.RecordBatch:
    """sort_table :75-106 -- ascending on the given column indices, nulls at end."""
    if batch.num_rows == 0 or not sort_cols:
        return batch
    import pyarrow.compute as pc
    keys = [(batch.schema.names[i], "ascending") for i in sort_cols]
    idx = pc.sort_indices(batch, sort_keys=keys)
    return batch.take(idx)
