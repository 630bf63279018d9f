"""ctypes binding of libvinum_hip.so (include/vinum_hip.h).

There is NO CPU fallback: if the shared library is missing, or no MI355X/HIP device is usable,
loading / initialisation raises.  Build the library with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C vinum_amd/csrc``.
"""
import ctypes
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libvinum_hip.so")

c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_void = ctypes.c_void_p
c_dbl = ctypes.c_double


class DCol(ctypes.Structure):
    """struct vnm_dcol"""
    _fields_ = [("values", c_void), ("validity", c_void), ("offset", c_i64), ("length", c_i64),
                ("type", ctypes.c_int32), ("flags", ctypes.c_int32)]


class ExprIns(ctypes.Structure):
    """struct vnm_expr_ins"""
    _fields_ = [("op", ctypes.c_int32), ("arg", ctypes.c_int32), ("imm_f", c_dbl), ("imm_i", c_i64)]


# enums (include/vinum_hip.h)
I8, I16, I32, I64, U8, U16, U32, U64, F32, F64 = range(10)
CSV_STRING, CSV_DATE32, CSV_TIMESTAMP_S, CSV_TIMESTAMP_NS, CSV_BOOL, CSV_TIME32_S = 200, 201, 202, 203, 204, 205   # vnm_csv_parse_block_ex column kinds
COUNT_STAR, COUNT, MIN, MAX, SUM, AVG = range(6)
ONE_GROUP, SINGLE_NUMERICAL, MULTI_NUMERICAL = range(3)
ASC, DESC = 0, 1
EQ, NE, GT, GE, LT, LE = range(6)
(EX_COL, EX_CONST_F, EX_CONST_I, EX_ADD, EX_SUB, EX_MUL, EX_DIV, EX_MOD, EX_NEG, EX_BAND, EX_BOR, EX_BXOR,
 EX_BNOT, EX_EQ, EX_NE, EX_GT, EX_GE, EX_LT, EX_LE, EX_AND, EX_OR, EX_NOT, EX_IS_NULL, EX_IS_NOT_NULL,
 EX_STORE) = range(25)
MASK_U8 = 100
OUT_U64, OUT_I64, OUT_F64, OUT_F32, OUT_DEC128, OUT_I32 = range(6)
FLAG_SUM32 = 1
# accumulator-word merge kinds (vnm_agg_plan_host)
M_ADD_U64, M_ADD_F64, M_MIN_U64, M_MAX_U64, M_ADD_F64C = range(5)

# name -> (restype, argtypes); mirrors include/vinum_hip.h one to one
PROTOTYPES = {
    "vnm_init": (c_int, [c_int]),
    "vnm_last_error": (ctypes.c_char_p, []),
    "vnm_device_count": (c_int, []),
    "vnm_device_synchronize": (c_int, []),
    "vnm_set_profiling": (c_int, [c_int]),
    "vnm_profile_query": (c_int, [ctypes.c_char_p, c_void, c_void]),
    "vnm_filter_cmp": (c_int, [c_void, c_int, c_int, c_dbl, c_i64, c_int, c_void, c_void, c_void, c_void, c_void]),
    "vnm_filter_mask": (c_int, [c_void, c_void, c_i64, c_int, c_void, c_void, c_void, c_void, c_void]),
    "vnm_pack_validity": (c_int, [c_void, c_i64, c_void, c_void]),
    "vnm_filter_scratch_bytes": (c_i64, [c_i64]),
    "vnm_agg_create": (c_void, [c_int, c_int, c_void, c_int, c_void, c_void, c_void, c_void]),
    "vnm_agg_destroy": (None, [c_void]),
    "vnm_agg_set_predicate": (c_int, [c_void, c_int, c_int, c_int, c_dbl, c_i64]),
    "vnm_agg_set_hint": (c_int, [c_void, c_i64]),
    "vnm_agg_set_exchange_mode": (c_int, [c_void, c_int]),
    "vnm_agg_next_device": (c_int, [c_void, c_i64, c_void, c_void, c_void, c_void]),
    "vnm_agg_set_async": (c_int, [c_void, c_int]),
    "vnm_agg_sync": (c_int, [c_void, c_void]),
    "vnm_agg_waiting": (c_int, [c_void, c_void, c_void, c_void, c_void]),
    "vnm_agg_set_input_expr": (c_int, [c_void, c_int, c_int, c_void, c_int]),
    "vnm_agg_next_device_expr": (c_int, [c_void, c_i64, c_void, c_void, c_void, c_int, c_void, c_void]),
    "vnm_agg_finish": (c_int, [c_void, c_void, c_void]),
    "vnm_agg_layout": (c_int, [c_void, c_void, c_void]),
    "vnm_agg_dense_ptrs": (c_int, [c_void, c_void, c_void]),
    "vnm_agg_bucket_by_owner": (c_int, [c_void, c_int, c_void, c_void, c_void]),
    "vnm_agg_merge_device": (c_int, [c_void, c_i64, c_void, c_void, c_void]),
    "vnm_agg_run_partitions": (c_i64, [c_void]),
    "vnm_agg_run_reorder": (c_int, [c_void, c_int, c_void, c_void, c_void, c_void]),
    "vnm_agg_merge_partitioned": (c_int, [c_void, c_int, c_i64, c_void, c_void, c_void, c_void]),
    "vnm_agg_merge_rows": (c_int, [c_void, c_i64, c_void, c_void]),
    "vnm_agg_merge_row_blocks": (c_int, [c_void, c_int, c_i64, c_void, c_void]),
    "vnm_agg_result_key": (c_int, [c_void, c_int, c_void, c_void]),
    "vnm_agg_result_func": (c_int, [c_void, c_int, c_void, c_void, c_void]),
    "vnm_agg_result_key_device": (c_int, [c_void, c_int, c_void, c_void, c_void, c_void]),
    "vnm_agg_result_device": (c_int, [c_void, c_int, c_void, c_void, c_void, c_void, c_void, c_void]),
    "vnm_agg_result_device_alloc": (c_int, [c_void, c_int, c_void, c_void, c_void, c_void, c_void, c_void, c_void]),
    "vnm_agg_dense_range": (c_int, [c_void, c_i64, c_void, c_void, c_void, c_void]),
    "vnm_agg_set_dense_range": (c_int, [c_void, c_int, ctypes.c_uint64, ctypes.c_uint64]),
    "vnm_agg_dense_table": (c_int, [c_void, c_void, c_void, c_void, c_void]),
    "vnm_agg_merge_dense_tables": (c_int, [c_void, c_void, c_int, c_void, ctypes.c_uint64, c_i64, c_void]),
    "vnm_agg_estimate_groups": (c_int, [c_void, c_i64, c_void, c_void, c_void]),
    "vnm_agg_result_func_device": (c_int, [c_void, c_int, c_void, c_void, c_void, c_void, c_void]),
    "vnm_agg_plan_host": (c_int, [c_int, c_int, c_void, c_int, c_void, c_void, c_void, c_void, c_void, c_void, c_void,
                                  c_void, c_void]),
    "vnm_agg_finalize_host": (c_int, [c_int, c_int, c_void, c_int, c_void, c_void, c_void, c_void, c_int, c_i64,
                                      c_void, c_void, c_void, c_void]),
    "vnm_agg_op_create": (c_void, [c_int, c_int, c_void, c_int, c_void, c_int, c_void, c_void, c_void]),
    "vnm_agg_op_next": (c_int, [c_void, c_void, c_void]),
    "vnm_agg_op_result": (c_int, [c_void, c_void, c_void]),
    "vnm_agg_op_destroy": (None, [c_void]),
    "vnm_sort_indices": (c_int, [c_int, c_void, c_void, c_i64, c_i64, c_void, c_void]),
    "vnm_sort_indices_keyed": (c_int, [c_int, c_void, c_void, c_i64, c_i64, c_void, c_void, c_void, c_void]),
    "vnm_take": (c_int, [c_void, c_void, c_i64, c_void, c_void, c_void]),
    "vnm_sort_op_create": (c_void, [c_int, c_void, c_void]),
    "vnm_sort_op_next": (c_int, [c_void, c_void, c_void]),
    "vnm_sort_op_sorted": (c_int, [c_void, c_i64, c_void, c_void]),
    "vnm_sort_op_destroy": (None, [c_void]),
    "vnm_project": (c_int, [c_int, c_void, c_int, c_void, c_i64, c_void, c_void, c_void]),
    "vnm_project_multi": (c_int, [c_int, c_void, c_int, c_void, c_i64, c_int, c_void, c_void, c_void]),
    "vnm_stage_column": (c_int, [c_void, c_void, c_i64, c_i64, ctypes.c_int32, c_void, c_void]),
    "vnm_free_column": (c_int, [c_void]),
    "vnm_csv_parse_block": (c_int, [c_void, c_i64, c_int, c_int, c_int, c_int, c_void, c_void, c_void, c_void, c_void, c_void]),
    "vnm_csv_parse_block_ex": (c_int, [c_void, c_i64, c_int, c_int, c_int, c_int, c_void, c_void, c_void, c_void, c_void, c_void, c_void]),
    "vnm_agg_op_next_stream": (c_int, [c_void, c_void]),
    "vnm_sort_op_next_stream": (c_int, [c_void, c_void]),
    "vnm_strdict_create": (c_void, []),
    "vnm_strdict_destroy": (None, [c_void]),
    "vnm_strdict_ids": (c_i64, [c_void]),
    "vnm_strdict_encode": (c_int, [c_void, c_void, c_int, c_void, c_void, c_i64, c_i64, c_void, c_void, c_void, c_void]),
    "vnm_strdict_encode_device": (c_int, [c_void, c_void, c_void, c_i64, c_void, c_i64, c_void, c_void, c_void, c_void]),
    "vnm_strdict_fetch_new": (c_int, [c_void, c_void, c_void, c_void]),
    "vnm_strdict_ranks_device": (c_int, [c_void, c_void, c_void]),
    "vnm_strdict_codes_to_ranks": (c_int, [c_void, c_void, c_i64, c_void, c_void]),
    "vnm_strdict_encode_spans": (c_int, [c_void, c_void, c_void, c_i64, c_void, c_void, c_void, c_void, c_void]),
    "vnm_strdict_last_new": (c_int, [c_void, c_void, c_void]),
    "vnm_take_varwidth": (c_int, [c_void, c_void, c_void, c_void, c_i64, c_void, c_void, c_void, c_void, c_void]),
    "vnm_take_bits": (c_int, [c_void, c_i64, c_void, c_i64, c_void, c_void]),
    "vnm_take_fixed16": (c_int, [c_void, c_void, c_i64, c_void, c_void]),
    "vnm_decimal128_sort_keys": (c_int, [c_void, c_i64, c_void, c_void, c_void]),
    "vnm_partition_by_owner": (c_int, [c_void, c_i64, c_void, c_int, c_void, c_void, c_void]),
    "vnm_malloc": (c_void, [c_i64]),
    "vnm_free": (c_int, [c_void]),
    "vnm_pool_trim": (c_i64, []),
    "vnm_pool_set_idle_trim": (c_int, [c_i64, c_i64]),
    "vnm_pool_cached_bytes": (c_i64, []),
    "vnm_route_counts": (c_i64, [c_void, c_i64]),
    "vnm_route_last": (c_i64, [c_void, c_i64]),
    "vnm_route_reset": (None, []),
    "vnm_memcpy_h2d": (c_int, [c_void, c_void, c_i64]),
    "vnm_memcpy_d2h": (c_int, [c_void, c_void, c_i64]),
    "vnm_memset": (c_int, [c_void, c_int, c_i64]),
}

_lib = None
_inited = False


class VinumHipError(RuntimeError):
    pass


def load():
    """dlopen libvinum_hip.so and attach prototypes.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VinumHipError(
            f"{LIB_PATH} is missing: build it (make -C vinum_amd/csrc). vinum_amd has no CPU fallback.")
    # If torch is (or will be) in the process, import it first so both share ONE HIP runtime
    # (torch wheels bundle libamdhip64.so.7; ours binds by the same soname).
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def lib():
    """Loaded AND initialised library (selects the current HIP device).  Raises without a GPU."""
    global _inited
    L = load()
    if not _inited:
        dev = -1
        if "torch" in sys.modules:
            import torch
            if torch.cuda.is_available():
                dev = torch.cuda.current_device()
        if L.vnm_init(dev) != 0:
            raise VinumHipError(L.vnm_last_error().decode())
        _inited = True
    return L


def check(rc):
    if rc != 0:
        raise VinumHipError(load().vnm_last_error().decode())


def last_error():
    return load().vnm_last_error().decode()
