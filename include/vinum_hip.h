/*
 * vinum_hip.h -- C ABI of libvinum_hip.so, the MI355X (gfx950) operator library behind Vinum's
 * native operator boundary.
 *
 * Every entry point is `extern "C"`, takes plain pointers / sizes / Arrow C Data Interface structs
 * and returns a status code (0 = ok; vnm_last_error() holds the message).  Nothing throws or aborts
 * across the ABI (the reference aborts on a bad handle, vinum/core/vinum_lib.cpp:62-63).
 * One handle = one HIP stream; handles are not thread-safe, distinct handles may be used from distinct
 * threads.  Device pointers are plain `void*` (hipMalloc / torch .data_ptr()).
 *
 * Each block cites the reference interface it replaces (paths relative to the reference root).
 */
#ifndef VINUM_HIP_H
#define VINUM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Arrow C Data Interface (https://arrow.apache.org/docs/format/CDataInterface.html) ---------- */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
struct ArrowSchema {
    const char* format;
    const char* name;
    const char* metadata;
    int64_t flags;
    int64_t n_children;
    struct ArrowSchema** children;
    struct ArrowSchema* dictionary;
    void (*release)(struct ArrowSchema*);
    void* private_data;
};
struct ArrowArray {
    int64_t length;
    int64_t null_count;
    int64_t offset;
    int64_t n_buffers;
    int64_t n_children;
    const void** buffers;
    struct ArrowArray** children;
    struct ArrowArray* dictionary;
    void (*release)(struct ArrowArray*);
    void* private_data;
};
#endif

/* ---- Arrow C Stream Interface (https://arrow.apache.org/docs/format/CStreamInterface.html) ------ */
#ifndef ARROW_C_STREAM_INTERFACE
#define ARROW_C_STREAM_INTERFACE
struct ArrowArrayStream {
    int (*get_schema)(struct ArrowArrayStream*, struct ArrowSchema* out);
    int (*get_next)(struct ArrowArrayStream*, struct ArrowArray* out);
    const char* (*get_last_error)(struct ArrowArrayStream*);
    void (*release)(struct ArrowArrayStream*);
    void* private_data;
};
#endif

/* ---- enums ---------------------------------------------------------------------------------- */
/* physical column types; temporal Arrow types map to their storage integers */
enum vnm_type { VNM_I8 = 0, VNM_I16, VNM_I32, VNM_I64, VNM_U8, VNM_U16, VNM_U32, VNM_U64, VNM_F32, VNM_F64 };
/* replaces py::enum_<AggFuncType>  vinum/core/vinum_lib.cpp:25-32 (same order) */
enum vnm_agg_func { VNM_COUNT_STAR = 0, VNM_COUNT, VNM_MIN, VNM_MAX, VNM_SUM, VNM_AVG };
/* which operator class: vinum/core/aggregate.py:96-104 */
enum vnm_agg_kind { VNM_ONE_GROUP = 0, VNM_SINGLE_NUMERICAL = 1, VNM_MULTI_NUMERICAL = 2 };
/* replaces py::enum_<SortOrder>  vinum/core/vinum_lib.cpp:34-37 */
enum vnm_sort_order { VNM_ASC = 0, VNM_DESC = 1 };
/* comparison predicates: vinum/core/expressions.py:30-36 */
enum vnm_cmp_op { VNM_EQ = 0, VNM_NE, VNM_GT, VNM_GE, VNM_LT, VNM_LE };
/* arithmetic / bitwise expression opcodes: vinum/core/expressions.py:13-24 */
enum vnm_expr_op {
    VNM_EX_COL = 0,   /* push input column  (arg = column index)            */
    VNM_EX_CONST_F,   /* push float literal (imm_f)                          */
    VNM_EX_CONST_I,   /* push int literal   (imm_i)                          */
    VNM_EX_ADD, VNM_EX_SUB, VNM_EX_MUL, VNM_EX_DIV, VNM_EX_MOD,   /* np.add .. np.mod         */
    VNM_EX_NEG,                                                    /* np.negative              */
    VNM_EX_BAND, VNM_EX_BOR, VNM_EX_BXOR, VNM_EX_BNOT,             /* np.bitwise_* and ~x      */
    /* predicates (vinum/core/expressions.py:27-48): the result is a boolean mask */
    VNM_EX_EQ, VNM_EX_NE, VNM_EX_GT, VNM_EX_GE, VNM_EX_LT, VNM_EX_LE, /* NumPy comparison lambdas :30-36 */
    VNM_EX_AND, VNM_EX_OR, VNM_EX_NOT,                             /* pc.and_ / pc.or_ / pc.invert :27-29 */
    VNM_EX_IS_NULL, VNM_EX_IS_NOT_NULL,                            /* pc.is_null / pc.is_valid :37-38 (arg = column) */
    VNM_EX_STORE      /* pop the top of the stack into output `arg` (vnm_project_multi: one SELECT list, one pass) */
};
/* out_type of vnm_project when the expression is a predicate: out_values is a byte mask (1 byte per row) */
#define VNM_MASK_U8 100
/* column flags */
#define VNM_FLAG_SUM32 1 /* time32: SUM accumulates and wraps in int32 (agg_func_factory.cpp:132-137) */

/* ---- device column view ------------------------------------------------------------------------
 * The GPU counterpart of the reference's ArrayIter family (vinum_cpp/src/common/array_iterators.h:
 * 13-255): raw values buffer + optional validity bitmap + Arrow offset.  `values`/`validity` are
 * DEVICE pointers to the start of the Arrow buffers (not offset-adjusted). */
typedef struct vnm_dcol {
    const void* values;
    const uint8_t* validity; /* NULL = no nulls */
    int64_t offset;
    int64_t length;
    int32_t type;  /* enum vnm_type */
    int32_t flags; /* VNM_FLAG_* */
} vnm_dcol;

/* ---- library ------------------------------------------------------------------------------------ */
/* replaces vinum_lib.import_pyarrow() (vinum/core/vinum_lib.cpp:22-23): one-time init; selects the
 * HIP device.  Fails (non-zero) when no gfx950 device / HIP runtime is usable -- there is no CPU
 * fallback anywhere in this library. */
int vnm_init(int device_id);
const char* vnm_last_error(void);
int vnm_device_count(void);
/* host-side statistics of the last operator call (kernel launches, bytes staged, ...); debugging aid */
int vnm_device_synchronize(void);

/* kernel timing with HIP events on the launch stream (bench.py roofline leg): spans are named after the
 * kernel they bracket ("filter_kernel", "agg_scan", ...) */
int vnm_set_profiling(int on);
int vnm_profile_query(const char* name, double* total_ms, int64_t* count);
/* Which route of DESIGN.md section 4 the operators took, and why (round 5).  Every operator call that commits a batch to a path leaves
 * a note: a route name and the reason in numbers (estimates, ranges, thresholds).  vnm_route_counts: "route=count" lines for every
 * route taken since the library was loaded (or vnm_route_reset) into buf (cap bytes, zero terminated); returns the bytes the full
 * text needs.  vnm_route_last: the calling thread's last note, "route: reason".  VNM_AGG_TRACE=1 prints every note to stderr. */
int64_t vnm_route_counts(char* buf, int64_t cap);
int64_t vnm_route_last(char* buf, int64_t cap);
void vnm_route_reset(void);

/* ---- filter: FilterOperator._kernel + RecordBatch.filter -----------------------------------------
 * replaces vinum/core/algebra.py:119-123, vinum/arrow/record_batch.py:85-90 and the NumPy comparison
 * lambdas vinum/core/expressions.py:30-36 for the `column <op> literal` predicate shape, fused:
 * compare -> wave-ballot rank -> single-pass (decoupled look-back) compaction of every payload column.
 * NULL predicate inputs compare False (they reach NumPy as NaN, record_batch.py:112-118), `!=` True.
 *
 * pred:      predicate column;  scalar_is_float selects dval (Python float literal) or ival (int).
 * n_payload: columns to compact (the predicate column itself may be listed).
 * out_values[i]: device buffer of >= length * width(type_i) bytes.
 * out_valid[i]:  device byte-per-row validity (>= length bytes) or NULL when payload i has no validity.
 * out_count: host int64, rows kept.  stream: hipStream_t or NULL (default stream). */
int vnm_filter_cmp(const vnm_dcol* pred, int op, int scalar_is_float, double dval, int64_t ival,
                   int n_payload, const vnm_dcol* payload, void** out_values, uint8_t** out_valid,
                   int64_t* out_count, void* stream);
/* generic boolean mask (1 byte / row, optional byte validity -> emit_null) -> compaction */
int vnm_filter_mask(const uint8_t* mask, const uint8_t* mask_valid, int64_t length, int n_payload,
                    const vnm_dcol* payload, void** out_values, uint8_t** out_valid, int64_t* out_count,
                    void* stream);
/* byte-per-row validity (as written by the filter) -> Arrow validity bitmap of (n + 7) / 8 bytes */
int vnm_pack_validity(const uint8_t* valid_bytes, int64_t n, uint8_t* bitmap, void* stream);
/* bytes needed for the look-back scratch of a filter over `length` rows */
int64_t vnm_filter_scratch_bytes(int64_t length);

/* ---- hash aggregate ------------------------------------------------------------------------------
 * replaces  SingleNumericalHashAggregate / MultiNumericalHashAggregate / OneGroupAggregate
 *           (vinum/core/vinum_lib.cpp:54-124; vinum_cpp/src/operators/aggregate/).
 * Device level: keys / inputs are vnm_dcol views of HBM-resident columns. */
typedef struct vnm_agg vnm_agg;
struct vnm_expr_ins;   /* defined with the projection below */

/* in_col_ids (may be NULL): functions with equal NON-NEGATIVE ids read the same input column (they then share loads
 * and accumulators, e.g. SUM(v) and AVG(v)); a negative id means "a column of its own" (never shared); ignored for
 * COUNT(*).
 * Narrow columns (round 5): a SINGLE narrow integer key (int8 .. uint32) and float32 input columns whose functions are SUM / AVG / COUNT
 * only are widened to 64 bits inside vnm_agg_next_device (one extra pass, 12 B/row) and take the int64 / float64 kernels; the caller
 * declares and passes the columns as they are, result columns keep the declared key type (sums / averages of float32 are float64 in the
 * reference too: agg_func_factory.cpp:126-131, 206-211), and a predicate over a widened float32 column compares in float32 as NumPy does. */
vnm_agg* vnm_agg_create(int kind, int n_keys, const int* key_types, int n_funcs, const int* funcs,
                        const int* in_types, const int* in_flags, const int* in_col_ids);
void vnm_agg_destroy(vnm_agg* h);
/* optional fused WHERE `pred <op> literal` evaluated inside the aggregate scan (no materialised
 * filtered batch): Filter -> Aggregate of vinum/planner/planner.py:373-378,463-469 in one pass. */
int vnm_agg_set_predicate(vnm_agg* h, int enabled, int op, int scalar_is_float, double dval, int64_t ival);
/* Multi-GPU: the finished result of this handle will be exchanged with other ranks.  rank_aligned = 1 restricts the
 * operator to run layouts that every rank derives identically from the keys alone (hash partitions: owner(f) = f * P / F,
 * vnm_agg_run_partitions / _reorder / vnm_agg_merge_partitioned); the dense-key paths, whose code range comes from a
 * per-rank sample, are only used while the group count stays small enough for the all-gather exchange, where nothing
 * needs aligning. */
int vnm_agg_set_exchange_mode(vnm_agg* h, int rank_aligned);
/* The operator's own group-count estimate for a batch (the sample it would take itself: exact for up to ~30 k groups,
 * HyperLogLog beyond), WITHOUT aggregating anything; 0 = not applicable (the key is not a plain 8-byte column).
 * Multi-GPU hosts call it on every rank, agree on the maximum and pass it to vnm_agg_set_hint, so that all ranks cut their
 * results into the same partitions (the partition count follows the hint). */
int vnm_agg_estimate_groups(vnm_agg* h, int64_t nrows, const vnm_dcol* key, int64_t* estimate, void* stream);
/* expected number of groups (0 = unknown): sizes the table and picks the kernel strategy */
int vnm_agg_set_hint(vnm_agg* h, int64_t expected_groups);
/* BaseAggregate::Next (base_aggregate.cpp:23-45).  inputs[i] is the input column of func i (ignored for
 * COUNT_STAR).  pred may be NULL when no predicate is set.  Asynchronous on `stream`. */
int vnm_agg_next_device(vnm_agg* h, int64_t nrows, const vnm_dcol* keys, const vnm_dcol* inputs,
                        const vnm_dcol* pred, void* stream);
/* Asynchronous streams (round 4).  The reference feeds an aggregate one record batch per Next() call (base_aggregate.cpp:23-45;
 * TableReaderOperator / stream_reader.py:32-94 produce them) and its state does not depend on where the batches are cut.  With
 * vnm_agg_set_async(h, 1), vnm_agg_next_device only RECORDS a batch of the hot shape (one plain 8-byte key; COUNT(*) / COUNT /
 * SUM / AVG of one plain float64 column; a plain float64 predicate column or none): no launch, allocation or host read-back per
 * call.  The waiting batches go to the device together, as the segments of ONE logical batch (one launch of the path's kernels
 * over all of them), when vnm_agg_sync / vnm_agg_finish / any result call is made, when 2^30 rows or 256 batches are waiting, or
 * when a batch of another shape arrives (that one is processed as usual, after the waiting ones).  Group-count estimates and the
 * dense path's code range are sampled from the FIRST waiting batch; errors of a waiting batch surface in the call that processes it.
 * CONTRACT: the buffers of every batch passed while async is on must stay alive and unchanged until the next vnm_agg_sync /
 * vnm_agg_finish / vnm_agg_result_* / vnm_agg_dense_table call on the handle returns.  Results are identical to the synchronous
 * mode (same kernels, same merges).  Default: off.
 * Batches over SEVERAL plain 8-byte input columns (int64 / uint64 key) are recorded too: when they go to the device the path is chosen
 * from the stream's total row count, the program is cut into one part per input column where a rule applies (the parts record the
 * batches in turn and launch once each), and otherwise the batches are processed one by one as in the synchronous mode. */
int vnm_agg_set_async(vnm_agg* h, int enabled);
/* processes every waiting batch and waits for `stream` */
int vnm_agg_sync(vnm_agg* h, void* stream);
/* Which batches of an asynchronous stream the library still only holds RECORDED.  Every vnm_agg_next_device call on the handle
 * has a sequence number (0, 1, 2, ...; *last_seq = the latest call's); *oldest_seq = the number of the oldest call whose batch is
 * still waiting somewhere in the operator (-1: none is).  The caller may release the buffers of every batch with a smaller number
 * -- what keeps the memory of a long stream bounded (the reference streams inputs larger than memory batch by batch:
 * vinum/api/stream_reader.py:32-94, README.rst:43-45).  *batches / *rows (optional): how much is waiting (at most 256 batches /
 * 2^30 rows per queue; a batch the parts of a split program hold is counted once per part).  Any pointer may be NULL. */
int vnm_agg_waiting(vnm_agg* h, int64_t* batches, int64_t* rows, int64_t* oldest_seq, int64_t* last_seq);
/* Expressions inside aggregates -- `sum((1 - total) * (2 + tax) * (1 - tip))`, vinum/tests/test_query_results.py:436-443;
 * the reference's planner projects the expression into a temporary column first (vinum/planner/planner.py:384-417).
 * vnm_agg_set_input_expr: the input column of function `func_idx` (and of every function sharing its in_col_id; declare
 * their input type as VNM_F64) is the value of `program` (postfix, as vnm_project; result float64) over the n_cols
 * columns passed to vnm_agg_next_device_expr as expr_cols.  In the hot shape ({COUNT(*), COUNT, SUM, AVG} of the
 * expression, plain 8-byte key or no GROUP BY, float64 columns without NULLs, program of + - * / negation within 16
 * instructions / 4 columns / stack depth 4) the expression is evaluated IN REGISTERS inside the scan / partition
 * kernels -- no materialised column; otherwise one fused vnm_project pass materialises it first.  inputs[i] of the
 * functions reading the expression are ignored. */
int vnm_agg_set_input_expr(vnm_agg* h, int func_idx, int n_ins, const struct vnm_expr_ins* program, int n_cols);
int vnm_agg_next_device_expr(vnm_agg* h, int64_t nrows, const vnm_dcol* keys, const vnm_dcol* inputs, const vnm_dcol* pred,
                             int n_expr_cols, const vnm_dcol* expr_cols, void* stream);
/* BaseAggregate::Result part 1: compact the table into dense device arrays; returns group count. */
int vnm_agg_finish(vnm_agg* h, int64_t* n_groups, void* stream);
/* dense partial state (after finish), for the multi-GPU exchange: key words then accumulator words,
 * each a device array of n_groups uint64.  n_key_words / n_acc_words describe the layout. */
int vnm_agg_layout(vnm_agg* h, int* n_key_words, int* n_acc_words);
int vnm_agg_dense_ptrs(vnm_agg* h, uint64_t** key_words, uint64_t** acc_words);
/* multi-GPU exchange helper: the finished dense run as rows [n_groups][n_key_words + n_acc_words] (device,
 * row-major) grouped by owner rank, owner = mix(key words) mod world exactly as vinum_amd/distributed.py::owner_of;
 * counts_host[world] receives the rows per owner. */
int vnm_agg_bucket_by_owner(vnm_agg* h, int world, uint64_t* out_rows, int64_t* counts_host, void* stream);
/* merge dense partial states produced by another handle with the same spec (device pointers) */
int vnm_agg_merge_device(vnm_agg* h, int64_t n, uint64_t* const* key_words, uint64_t* const* acc_words,
                         void* stream);
/* Partition-aligned exchange for large G (single 8-byte key, add-merge words): the partitioned path leaves the
 * groups of hash partition f contiguous and every rank uses the same hash bits, so owner(f) = f * world / F and
 * the owner merges partition by partition in LDS (no HBM atomics).
 *   vnm_agg_run_partitions: F when the finished result is such a run, else 0 (all ranks must agree).
 *   vnm_agg_run_reorder:    rows in partition order [n][2 + n_acc_words], device uint32 row counts per partition,
 *                           host row counts per owner.
 *   vnm_agg_merge_partitioned: on an EMPTY handle: received rows grouped by source rank, row offsets of the
 *                           source blocks (host, world + 1), device uint32 counts [world][nlocal]. */
int64_t vnm_agg_run_partitions(vnm_agg* h);
int vnm_agg_run_reorder(vnm_agg* h, int world, uint64_t* out_rows, uint32_t* out_part_counts,
                        int64_t* owner_counts_host, void* stream);
int vnm_agg_merge_partitioned(vnm_agg* h, int world, int64_t nlocal, const uint64_t* rows,
                              const int64_t* src_row_offsets_host, const uint32_t* part_counts, void* stream);
/* merge row-major partial groups [n][n_key_words + n_acc_words] (layout of vnm_agg_bucket_by_owner) */
int vnm_agg_merge_rows(vnm_agg* h, int64_t n, const uint64_t* rows, void* stream);
/* The receive buffer of the one-collective small-result exchange (vinum_amd/distributed.py::exchange_small_fixed): `nblocks` blocks
 * of (block_rows + 1) such rows, row 0 of every block being its header (word 0 = how many of the following rows hold partial
 * groups).  Merged like vnm_agg_merge_rows, the counts are read on the device. */
int vnm_agg_merge_row_blocks(vnm_agg* h, int nblocks, int64_t block_rows, const uint64_t* blocks, void* stream);
/* BaseAggregate::Result part 2 (+ agg funcs' Summarize, agg_funcs.h:72-80,358-397,482-491,519-540):
 * D2H + host finalisation.  key j -> vals[n] raw 64-bit patterns + valid[n] bytes.
 * func i -> cells of 16 bytes (decimal128 uses all 16), valid bytes; returns the output kind. */
enum vnm_out_kind { VNM_OUT_U64 = 0, VNM_OUT_I64, VNM_OUT_F64, VNM_OUT_F32, VNM_OUT_DEC128, VNM_OUT_I32 };
int vnm_agg_result_key(vnm_agg* h, int key_idx, uint64_t* vals, uint8_t* valid);
int vnm_agg_result_func(vnm_agg* h, int func_idx, void* cells16, uint8_t* valid, int* out_kind);

/* BaseAggregate::Result part 2 ON THE DEVICE: the result column of group key `key_idx` / function `func_idx` as Arrow
 * buffers in HBM (what base_aggregate.cpp:47-68 assembles from the functions' Summarize methods, agg_funcs.h:72-80,
 * 482-491, 519-540 incl. the 128-bit AVG): out_values = n_groups values of the column's OUTPUT type (keys and MIN / MAX:
 * the input type; COUNT: uint64; SUM: int64 / uint64 / float64, int32 for time32; AVG: float64, float32 for 8 / 16-bit
 * integers -- *out_kind as in vnm_agg_result_func), out_bitmap = Arrow validity bitmap of ((n_groups + 63) / 64) * 8
 * bytes, *null_count = NULL results.  Returns 2 (and leaves the buffers undefined) when an int64 / uint64 SUM overflowed
 * 64 bits in some group: the reference then promotes the whole column to decimal128 (agg_funcs.h:366-389), which
 * vnm_agg_result_func does on the host. */
int vnm_agg_result_key_device(vnm_agg* h, int key_idx, void* out_values, uint8_t* out_bitmap, int64_t* null_count, void* stream);
/* The same for n_cols result columns in ONE kernel launch (the accumulator words of a group are read once and
 * the launch cost is paid once): which[c] >= 0 selects aggregate function which[c], which[c] < 0 selects key
 * column ~which[c].  out_kinds[c] = VNM_OUT_* of a function column, -1 for a key column (it keeps its input
 * type).  Returns 2 when any selected int64 / uint64 SUM overflowed (all other columns are still valid). */
int vnm_agg_result_device(vnm_agg* h, int n_cols, const int* which, void* const* out_values,
                          uint8_t* const* out_bitmaps, int* out_kinds, int64_t* null_counts, void* stream);
int vnm_agg_result_func_device(vnm_agg* h, int func_idx, void* out_values, uint8_t* out_bitmap, int* out_kind,
                               int64_t* null_count, void* stream);
/* BaseAggregate::Result (base_aggregate.cpp:47-68: Reserve(G) on every builder, Summarize every group, one RecordBatch)
 * with the output columns ALLOCATED BY THE LIBRARY -- the group count is only known once the last pass has run, exactly as
 * the reference's builders only learn it in SummarizeGroups (single_numerical_hash_aggregate.cpp:48-68).  When the last
 * batch went through the dense-key path, its direct-addressed final pass is still pending at this point and writes the
 * result columns ITSELF (key, SUM, AVG, COUNT of a float64 column: the Summarize expressions of agg_funcs.h:139-142,
 * 286-292, 519-522 evaluated where the accumulators live) -- no dense partial state, no second kernel re-reading it.
 * Every other state: vnm_agg_finish + vnm_agg_result_device.  *n_groups = rows; out_values[c] = vnm_malloc block of
 * >= *n_groups cells of the column's output type (as vnm_agg_result_device); out_bitmaps[c] = validity bitmap block or
 * NULL when the column has no NULL result.  The CALLER frees both with vnm_free.  Returns 2 like vnm_agg_result_device. */
int vnm_agg_result_device_alloc(vnm_agg* h, int n_cols, const int* which, void** out_values, uint8_t** out_bitmaps,
                                int* out_kinds, int64_t* null_counts, int64_t* n_groups, void* stream);

/* Multi-GPU, large results (legal because base_aggregate.cpp:23-45 only ever Init()s or Update()s a group's
 * state: partial states of disjoint row sets merge commutatively).  The dense-key path replaces a key by a code inside a
 * key RANGE; when all ranks use the SAME range, their direct-addressed final tables are slot-compatible: they add up
 * element by element, a key's owner is a range of codes, and nothing has to be bucketed, counted or re-ordered.
 *   vnm_agg_dense_range      this rank's sampled key range of a batch, as order-preserving unsigned images (*lo > *hi: the
 *                            key type has no dense path).  Ranks reduce lo by MIN, hi by MAX ...
 *   vnm_agg_set_dense_range  ... and give every operator the agreed range BEFORE its first batch (lo > hi: none).
 *   vnm_agg_dense_table      after ONE batch: *table = 2^*bits slots {double sum; float lo; uint32 count} (count 0 = no
 *                            group), slot = scrambled key code; NULL when the batch did not take that path (spilled keys,
 *                            another aggregate shape, several batches): the caller falls back to the owner-bucketed
 *                            exchange.  geometry[4] = {range start, bits, multiplier, sign}: must agree on all ranks.
 *                            The table belongs to the handle.
 *   vnm_agg_merge_dense_tables   owner side: slices [code0, code0 + n) of the nsrc ranks' tables (rank order) -> the
 *                            result of the EMPTY handle h (`like`: any handle that produced one of the tables).  Sums merge
 *                            with the library's compensated add, in rank order: the result does not depend on the owner. */
int vnm_agg_dense_range(vnm_agg* h, int64_t nrows, const vnm_dcol* key, uint64_t* lo, uint64_t* hi, void* stream);
int vnm_agg_set_dense_range(vnm_agg* h, int key_type, uint64_t lo, uint64_t hi);
int vnm_agg_dense_table(vnm_agg* h, void** table, int* bits, uint64_t* geometry, void* stream);
int vnm_agg_merge_dense_tables(vnm_agg* h, const vnm_agg* like, int nsrc, const void* const* slices, uint64_t code0,
                               int64_t n, void* stream);

/* Host-only helpers (no GPU touched): how (functions, input types) lower onto 64-bit accumulator words
 * with commutative merge kinds (0 add-u64, 1 add-f64, 2 min-u64, 3 max-u64, 4 compensated add-f64: the high
 * word of a (hi, lo) pair -- every add into it is a returning atomic whose exact rounding error (TwoSum) is
 * added to the NEXT word, an ordinary add-f64 word; the float64 SUM / AVG of a group is hi + lo, i.e. the
 * correctly rounded exact sum up to second-order terms, whatever order the rows arrived in), and the
 * finalisation of one result column from dense accumulator words in HOST memory.  Used by the multi-GPU merge
 * and unit-tested on CPU. */
int vnm_agg_plan_host(int kind, int n_keys, const int* key_types, int n_funcs, const int* funcs,
                      const int* in_types, const int* in_flags, const int* in_col_ids, int* n_key_words,
                      int* n_acc_words, int* merge_kinds /* >= 40 ints */, int* n_ops,
                      int* op_kind_col_word /* >= 3 * 48 ints: per-row update kind, distinct input column, word;
                                               kinds: 0 count rows, 1 count valid, 2 sum f64, 3 sum i64 (wrap),
                                               4 sum low 32 bits, 5 sum high 32 (signed), 6 sum high 32 (unsigned),
                                               7 min, 8 max (on the order-preserving 64-bit encoding) */);
int vnm_agg_finalize_host(int kind, int n_keys, const int* key_types, int n_funcs, const int* funcs,
                          const int* in_types, const int* in_flags, const int* in_col_ids, int func_idx,
                          int64_t n, const uint64_t* const* acc_words, void* cells16, uint8_t* valid,
                          int* out_kind);

/* Arrow level (host RecordBatches through the C Data Interface; columns are staged to HBM with
 * pinned double-buffered DMA).  Same constructor arguments as the pybind classes. */
typedef struct vnm_agg_op vnm_agg_op;
vnm_agg_op* vnm_agg_op_create(int kind, int n_groupby, const char** groupby_cols, int n_aggcols,
                              const char** agg_cols, int n_funcs, const int* func_types,
                              const char** in_cols, const char** out_cols);
int vnm_agg_op_next(vnm_agg_op* h, struct ArrowArray* batch, struct ArrowSchema* schema); /* consumes both */
/* Every batch of an Arrow C stream, as one vnm_agg_op_next each (consumes and releases the stream).  The reference's pipeline
 * hands over 10 000-row batches (vinum/__init__.py:52, table_batch_reader.cpp:5-16); crossing the language boundary once per
 * few hundred of them instead of once each is what the caller saves -- the operator keeps small batches as they are (no
 * concatenation on the host) and stages them to the device together. */
int vnm_agg_op_next_stream(vnm_agg_op* h, struct ArrowArrayStream* stream);
int vnm_agg_op_result(vnm_agg_op* h, struct ArrowArray* out, struct ArrowSchema* out_schema);
void vnm_agg_op_destroy(vnm_agg_op* h);

/* ---- sort: Sort.next / Sort.sorted ------------------------------------------------------------------
 * replaces vinum/core/vinum_lib.cpp:126-142, vinum_cpp/src/operators/sort/sort.cpp:11-63
 * (arrow::compute::SortIndices + Take).  The order is total -- order-preserving key encodings, then the row id (stable) --
 * with NaN after all numbers and NULL after NaN for both ASC and DESC, whichever of the library's sorts produces it: the sample
 * sort over 8-byte entry words (one 8-byte key, order only, distinct keys: vnm_sort_apx.inc), the splitter sample sort (keyed
 * results, duplicated keys), the stable LSD radix sort (several keys, heavily duplicated keys, rows that arrive in key order),
 * top-K selection for limit > 0. */
int vnm_sort_indices(int n_keys, const vnm_dcol* keys, const int* orders, int64_t length,
                     int64_t limit /* <=0: full sort; >0: only the first `limit` rows are needed */,
                     int64_t* out_indices /* device, length entries (first `limit` valid) */, void* stream);
/* The same (Sort::Sorted = SortIndices + Take of EVERY column, sort.cpp:22-40), and -- for a full sort whose first key is
 * int64 / uint64 / float64 without NULL, NaN or -0.0 -- the sorted
 * values of that key straight from the sort (out_sorted_key0: device, length x 8 bytes; *wrote_key0 = 1), which saves
 * the caller the gather (vnm_take: 25 ms per 1e9 rows) for that column.  *wrote_key0 = 0: gather as usual. */
int vnm_sort_indices_keyed(int n_keys, const vnm_dcol* keys, const int* orders, int64_t length, int64_t limit,
                           int64_t* out_indices, void* out_sorted_key0, int* wrote_key0, void* stream);
int vnm_take(const vnm_dcol* col, const int64_t* indices, int64_t n, void* out_values, uint8_t* out_valid,
             void* stream);
/* Take for the column types vnm_take does not cover (Sort::Sorted takes EVERY column of the table, any Arrow type: sort.cpp:38-40).
 * All pointers are DEVICE pointers unless said otherwise.
 * vnm_take_varwidth: utf8 / binary values by row ids.  offsets: n_rows + 1 int64 offsets into `data` (int32 Arrow offsets are widened
 *   by the caller), validity: bitmap (bit i = row i) or NULL.  out_offsets: n + 1 int64s; *out_data: the gathered bytes, a block the
 *   CALLER frees with vnm_free (*out_bytes of them); out_valid: n bytes (1 = valid) or NULL.  A NULL row contributes no bytes.
 * vnm_take_bits: bits (bit_offset + indices[i]) of a bitmap -> n bytes of 0 / 1 (boolean values, validity bitmaps).
 * vnm_take_fixed16: 16-byte values (decimal128).
 * vnm_decimal128_sort_keys: decimal128 values -> (high word int64, low word uint64): two numeric sort keys, most significant first. */
int vnm_take_varwidth(const int64_t* offsets, const uint8_t* data, const uint8_t* validity, const int64_t* indices, int64_t n,
                      int64_t* out_offsets, uint8_t** out_data, int64_t* out_bytes, uint8_t* out_valid, void* stream);
int vnm_take_bits(const uint8_t* bits, int64_t bit_offset, const int64_t* indices, int64_t n, uint8_t* out_bytes, void* stream);
int vnm_take_fixed16(const void* values, const int64_t* indices, int64_t n, void* out_values, void* stream);
int vnm_decimal128_sort_keys(const void* values, int64_t n, int64_t* out_hi, uint64_t* out_lo, void* stream);
/* Distributed sample sort (Sort::Sorted over rows sharded by batch, sort.cpp:22-44): the per-rank partition step.
 * Stable partition of the rows 0 .. n-1 by OWNER = number of splitters <= codes[row] (ascending int64 splitters, at most 63 of them; a
 * code equal to a splitter goes to the upper owner).  out_order: n row numbers, the rows of owner 0 first, source order kept inside an
 * owner; out_counts: n_splitters + 1 rows-per-owner counts.  All DEVICE pointers; returns after the stream has been synchronised. */
int vnm_partition_by_owner(const int64_t* codes, int64_t n, const int64_t* splitters, int n_splitters, int64_t* out_order,
                           int64_t* out_counts, void* stream);
typedef struct vnm_sort_op vnm_sort_op;
vnm_sort_op* vnm_sort_op_create(int n, const char** cols, const int* orders);
int vnm_sort_op_next(vnm_sort_op* h, struct ArrowArray* batch, struct ArrowSchema* schema);
int vnm_sort_op_next_stream(vnm_sort_op* h, struct ArrowArrayStream* stream);   /* as vnm_agg_op_next_stream */
int vnm_sort_op_sorted(vnm_sort_op* h, int64_t limit, struct ArrowArray* out, struct ArrowSchema* out_schema);
void vnm_sort_op_destroy(vnm_sort_op* h);

/* ---- projection: arithmetic expression evaluation ----------------------------------------------------
 * replaces the NumPy ufunc dispatch vinum/core/expressions.py:13-24 as evaluated by
 * VectorizedExpression.evaluate (vinum/core/base.py:105-125,145-151) and ProjectOperator._kernel
 * (vinum/core/algebra.py:52-64): one fused kernel per output expression, no temporaries.
 * Program = postfix opcode stream over columns of ANY numeric width (int8..uint64, float32, float64).  NumPy
 * promotion (2.x / NEP 50): floats win (float32 only survives 8 / 16-bit integers), same-signedness integers widen,
 * unsigned + signed -> the next wider signed type (uint64 + signed -> float64), integer results wrap in the result
 * width, `/` -> float64 (float32 for float32 with narrow integers), `%` = floor-mod (sign of the divisor); a literal
 * (VNM_EX_CONST_*) is "weak" and takes the column's type (arg = 1 makes it a strong int64 / float64 value, what an IN
 * list is); a column with NULLs is evaluated as float64 with NaN (float32 stays float32). */
typedef struct vnm_expr_ins {
    int32_t op;  /* enum vnm_expr_op */
    int32_t arg; /* column index for VNM_EX_COL */
    double imm_f;
    int64_t imm_i;
} vnm_expr_ins;
/* out_type (returned): the vnm_type of the result, values stored with that type's width (a buffer of length*8 bytes
 * always suffices), or VNM_MASK_U8 (predicate) with out_values = length bytes (BETWEEN / IN are compiled to AND / OR
 * chains by the caller). */
int vnm_project(int n_ins, const vnm_expr_ins* program, int n_cols, const vnm_dcol* cols, int64_t length,
                void* out_values, int* out_type, void* stream);
/* A whole SELECT list in one pass (ProjectOperator._kernel evaluates every expression of the list over the same
 * batch, vinum/core/algebra.py:52-64): the program holds n_out expressions, each terminated by
 * VNM_EX_STORE(arg = output index); every input column is read from HBM once.  out_values[k] must hold
 * length*8 bytes (length bytes suffice for a predicate output); out_types[k] is returned per output. */
int vnm_project_multi(int n_ins, const vnm_expr_ins* program, int n_cols, const vnm_dcol* cols, int64_t length,
                      int n_out, void** out_values, int* out_types, void* stream);

/* ---- TableBatchReader -----------------------------------------------------------------------------
 * replaces vinum/core/vinum_lib.cpp:144-165 / vinum_cpp/src/operators/table_batch_reader.cpp:5-16.
 * Host-side zero-copy slicing lives in the Python shim (pyarrow Table.slice); this entry stages one
 * host column into HBM through the library's pinned double buffer and returns the device view. */
int vnm_stage_column(const void* host_values, const uint8_t* host_validity, int64_t offset, int64_t length,
                     int32_t type, vnm_dcol* out, void* stream);
int vnm_free_column(vnm_dcol* col);

/* ---- string dictionary: non-numeric GROUP BY keys -----------------------------------------------------------------------
 * replaces the scalar-vector keyed map of GenericHashAggregate (vinum_cpp/src/operators/aggregate/generic_hash_aggregate.h:10-45:
 * one hash + Equals of arrow::Scalar vectors per row) for utf8 / binary key columns: every row's value -> a running int32
 * dictionary code (equal bytes <=> equal code, across all batches given to one handle); the codes then are an ordinary
 * numeric key column of the aggregate operators.  NULL rows get code -1.  Codes are NOT dense (ids are handed out in chunks;
 * vnm_strdict_ids = their upper bound).
 * vnm_strdict_encode: one Arrow utf8 / large_utf8 / binary array as HOST buffers (offsets int32 or int64, data, validity bitmap
 * or NULL, logical offset / length); offsets + bytes cross PCIe once, the codes come back to out_codes_host (length int32s).
 * *n_new / *new_bytes: the values this call added to the dictionary; vnm_strdict_fetch_new copies their (id, length) pairs and
 * concatenated bytes (in that order) to host buffers of n_new / n_new / new_bytes entries -- the caller's copy of the
 * dictionary is the concatenation of these (how vinum_amd/vinum_lib.py decodes the result's key column).
 * vnm_strdict_encode_device: the same over DEVICE buffers (offsets->values / ->type (VNM_I32 | VNM_I64) / ->offset /
 * ->length = rows + 1; validity bitmap + bit offset or NULL; data = byte data_base of the Arrow data buffer). */
typedef struct vnm_strdict vnm_strdict;
vnm_strdict* vnm_strdict_create(void);
void vnm_strdict_destroy(vnm_strdict* h);
int64_t vnm_strdict_ids(vnm_strdict* h);
int vnm_strdict_encode(vnm_strdict* h, const void* offsets_host, int offsets_are_64, const uint8_t* data_host,
                       const uint8_t* validity_host, int64_t offset, int64_t length, int32_t* out_codes_host,
                       int64_t* n_new, int64_t* new_bytes, void* stream);
int vnm_strdict_encode_device(vnm_strdict* h, const vnm_dcol* offsets, const uint8_t* validity, int64_t validity_offset,
                              const uint8_t* data, int64_t data_base, int32_t* out_codes, int64_t* n_new, int64_t* new_bytes,
                              void* stream);
int vnm_strdict_fetch_new(vnm_strdict* h, int32_t* ids_host, int32_t* lens_host, uint8_t* bytes_host);
/* Order-preserving RANKS of the dictionary's values (round 5: utf8 / binary ORDER BY keys on the device -- Arrow's SortIndices compares
 * such values byte-wise, sort.cpp:22-37).  out_rank_of_id: DEVICE array of vnm_strdict_ids(h) int32s; entry id = the position of the
 * value with that id among all values in ascending byte order (ids never handed out are left untouched).  vnm_strdict_codes_to_ranks maps
 * a column of codes (device; -1 = NULL -> rank 0, the row's validity bit says NULL) to its ranks: an ordinary int32 sort key column. */
int vnm_strdict_ranks_device(vnm_strdict* h, int32_t* out_rank_of_id, void* stream);
int vnm_strdict_codes_to_ranks(const int32_t* codes, const int32_t* rank_of_id, int64_t n, int32_t* out_ranks, void* stream);

/* ---- CSV ingest ---------------------------------------------------------------------------------------
 * replaces, for numeric columns, the pyarrow.csv reader behind stream_csv() / read_csv() (vinum/io/arrow.py:58-61,106;
 * FileReaderOperator vinum/core/algebra.py:268-279): one block of CSV text (must end with '\n'; < 2 GiB) is staged once
 * and tokenised + parsed on the device.  n_fields = fields per row (from the header); field_idx[c] (ascending) / types[c]
 * (VNM_I64 | VNM_F64) select the columns; out_cols[c] receives an HBM column (values + validity bitmap; an empty field is
 * NULL; free with vnm_free_column).  Decimal -> float64 is exact integer arithmetic (correctly rounded, as strtod /
 * fast_float).  fallback[c] = 1: column c holds a field outside the device parser's domain (> 19 significant digits,
 * |decimal exponent| > 19, "nan" / "inf", stray characters): parse that column of this block on the host;
 * fallback[n_cols] = 1: a row ends inside a quoted field (a newline in a value; quoted fields themselves are tokenised: delimiters
 * inside quotes do not split, a quoted number is parsed from between its quotes); fallback[n_cols + 1] = 1: a row whose
 * field count differs from n_fields. */
int vnm_csv_parse_block(const char* host_text, int64_t nbytes, int skip_header, int delimiter, int n_fields, int n_cols,
                        const int* field_idx, const int* types, vnm_dcol* out_cols, int64_t* n_rows, int* fallback,
                        void* stream);
/* Round 5: the non-numeric columns pyarrow.csv infers for such files stay on the device too (the same reader, its `string`,
 * `date32[day]`, `timestamp[s]` and `timestamp[ns]` conversions: vinum/io/arrow.py:58-61,106).  types[c] may also be
 *   VNM_CSV_STRING        utf8: out_cols[c] = int32 codes of dicts[c] (a vnm_strdict the caller keeps for the column across blocks; never
 *                         NULL -- pyarrow reads an empty field of a string column as the empty string); the field bytes are encoded where
 *                         they lie in the staged text (vnm_strdict_encode_spans) and checked to be well-formed UTF-8 (fallback otherwise:
 *                         pyarrow raises on them); after the call vnm_strdict_last_new / vnm_strdict_fetch_new hand over the values
 *                         the block added, as after vnm_strdict_encode.  A field with an escaped quote ("" inside quotes): fallback.
 *   VNM_CSV_DATE32        YYYY-MM-DD -> int32 days since 1970-01-01
 *   VNM_CSV_TIMESTAMP_S   YYYY-MM-DD | YYYY-MM-DD[ T]hh:mm | ...:ss -> int64 seconds
 *   VNM_CSV_TIMESTAMP_NS  the same plus .f{1,9} -> int64 nanoseconds (years 1678 .. 2261)
 * Calendar-checked (month, day of month, leap years, hh < 24, mm / ss < 60); any other spelling Arrow's ISO 8601 parser knows (zone
 * offsets, hour-only times, "Z") raises the column's fallback flag.  dicts: n_cols entries (NULL where the column is no string) or NULL. */
#define VNM_CSV_STRING 200
#define VNM_CSV_DATE32 201
#define VNM_CSV_TIMESTAMP_S 202
#define VNM_CSV_TIMESTAMP_NS 203
/*   VNM_CSV_BOOL          true / True / TRUE / 1 and false / False / FALSE / 0 (pyarrow's true_values / false_values) -> int32 1 / 0: the
 *                         codes of a dictionary [false, true] (booleans travel dictionary-coded like every non-numeric column)
 *   VNM_CSV_TIME32_S      hh:mm | hh:mm:ss (hh < 24) -> int32 seconds since midnight */
#define VNM_CSV_BOOL 204
#define VNM_CSV_TIME32_S 205
int vnm_csv_parse_block_ex(const char* host_text, int64_t nbytes, int skip_header, int delimiter, int n_fields, int n_cols,
                           const int* field_idx, const int* types, vnm_strdict* const* dicts, vnm_dcol* out_cols, int64_t* n_rows,
                           int* fallback, void* stream);
/* vnm_strdict_encode_device over SPANS of one device buffer: row r = the lens[r] bytes at data[starts[r]] (no NULLs). */
int vnm_strdict_encode_spans(vnm_strdict* h, const int64_t* starts, const int32_t* lens, int64_t n, const uint8_t* data,
                             int32_t* out_codes, int64_t* n_new, int64_t* new_bytes, void* stream);
/* how many values (and bytes) the last encode of this handle added and vnm_strdict_fetch_new has not handed over yet: the sizes it
 * fills (a fetch hands them over ONCE) */
int vnm_strdict_last_new(vnm_strdict* h, int64_t* n_new, int64_t* new_bytes);

/* device memory helpers for hosts without a GPU allocator of their own (ctypes / cgo bindings).
 * vnm_malloc / vnm_free go through the library's caching allocator: a freed block is handed to the NEXT
 * request of a similar size at once, without waiting for the device.  That is safe while everything that
 * touches the block is enqueued on ONE stream (work is ordered on it; the entry points that return counts or
 * flags also synchronise it).  A host that spreads calls over several streams must synchronise the stream
 * that last used a block before vnm_free (as with hipFree, minus its implicit device synchronisation). */
void* vnm_malloc(int64_t bytes);
int vnm_free(void* p);
/* hand every block the caching allocator holds but nobody uses back to the device (hipFree); returns the
 * bytes released.  For long-lived hosts that want the HBM back between queries; live blocks are untouched. */
int64_t vnm_pool_trim(void);
/* The cache also gives itself back: once no allocation or release has gone through the allocator for `idle_ms` (default 2000;
 * VNM_POOL_IDLE_MS) a background thread releases the cached blocks beyond `keep_bytes` (default 256 MiB; VNM_POOL_KEEP_BYTES),
 * so a host that embeds the library (an unmodified Vinum process) does not sit on tens of GB of HBM after a large query, while a
 * query in flight is never trimmed under its feet.  idle_ms < 0: never; keep_bytes < 0: leave it unchanged.
 * vnm_pool_cached_bytes: what the cache holds right now. */
int vnm_pool_set_idle_trim(int64_t idle_ms, int64_t keep_bytes);
int64_t vnm_pool_cached_bytes(void);
int vnm_memcpy_h2d(void* dst, const void* src, int64_t bytes);
int vnm_memcpy_d2h(void* dst, const void* src, int64_t bytes);
int vnm_memset(void* dst, int value, int64_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* VINUM_HIP_H */
