/*
 * vinum_oracle.c -- CPU restatement of the reference hot path (TEST INFRASTRUCTURE).
 *
 * This file is the parity oracle and the "port" CPU baseline.  It is NOT part of the
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * it.  The product (vinum_amd/) never links, imports or falls back to it.
 *
 * Parity pinning: checked (tests/test_oracle_*.py, -m "not gpu") against
 *   (1) the known answers of the reference's own gtest, transcribed as data in
 *       tests/golden/gtest_fixtures.py  (vinum_cpp/test/hash_agg_test.cpp:155-777), and
 *   (2) tests/golden/*.npz produced by the REAL reference operators built from
 *       /root/reference by oracle/ref_build (generator: tests/golden/gen_golden.py).
 *
 * Every function cites the reference file:line it restates.  Scalar, row-at-a-time, in the
 * reference's own evaluation order (so float SUMs round identically).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* physical value types (temporal Arrow types map onto their storage ints) */
enum { ORC_I8 = 0, ORC_I16, ORC_I32, ORC_I64, ORC_U8, ORC_U16, ORC_U32, ORC_U64, ORC_F32, ORC_F64 };
/* same order as the pybind enum, vinum/core/vinum_lib.cpp:25-32 */
enum { ORC_COUNT_STAR = 0, ORC_COUNT, ORC_MIN, ORC_MAX, ORC_SUM, ORC_AVG };
/* operator kinds: vinum/core/aggregate.py:96-104 */
enum { ORC_ONE_GROUP = 0, ORC_SINGLE, ORC_MULTI };
/* output kinds of orc_agg_func */
enum { ORC_OUT_U64 = 0, ORC_OUT_I64, ORC_OUT_F64, ORC_OUT_F32, ORC_OUT_DEC128, ORC_OUT_I32 };
/* sum flavours for time32 (int32 accumulator, agg_func_factory.cpp:132-137) */
#define ORC_FLAG_SUM32 1

typedef struct {
    const void *values;      /* Arrow data buffer base (NOT offset-adjusted) */
    const uint8_t *validity; /* Arrow validity bitmap or NULL */
    int64_t offset;          /* Arrow array offset (elements / bits) */
    int64_t length;
    int32_t type;
    int32_t flags;
} orc_col;

typedef unsigned __int128 u128;
typedef __int128 i128;

/* array_iterators.h:27-29 : IsNull = nulls_ptr && !GetBit(nulls_ptr, offset + i) */
static inline int col_is_null(const orc_col *c, int64_t i) {
    if (!c->validity) return 0;
    int64_t b = c->offset + i;
    return !((c->validity[b >> 3] >> (b & 7)) & 1);
}

static inline int is_float_type(int t) { return t == ORC_F32 || t == ORC_F64; }
static inline int is_unsigned_type(int t) { return t >= ORC_U8 && t <= ORC_U64; }

/* array_iterators.h:215-217 (ints: static_cast<uint64_t>(native)), :239-248 (floats: memcpy bits
 * into a zeroed uint64) -- the group-key encoding. */
static inline uint64_t col_key_bits(const orc_col *c, int64_t i) {
    int64_t k = c->offset + i;
    switch (c->type) {
        case ORC_I8: return (uint64_t)(int64_t)((const int8_t *)c->values)[k];
        case ORC_I16: return (uint64_t)(int64_t)((const int16_t *)c->values)[k];
        case ORC_I32: return (uint64_t)(int64_t)((const int32_t *)c->values)[k];
        case ORC_I64: return (uint64_t)((const int64_t *)c->values)[k];
        case ORC_U8: return ((const uint8_t *)c->values)[k];
        case ORC_U16: return ((const uint16_t *)c->values)[k];
        case ORC_U32: return ((const uint32_t *)c->values)[k];
        case ORC_U64: return ((const uint64_t *)c->values)[k];
        case ORC_F32: { uint64_t r = 0; memcpy(&r, &((const float *)c->values)[k], 4); return r; }
        case ORC_F64: { uint64_t r = 0; memcpy(&r, &((const double *)c->values)[k], 8); return r; }
    }
    return 0;
}

static inline int64_t col_i64(const orc_col *c, int64_t i) {
    int64_t k = c->offset + i;
    switch (c->type) {
        case ORC_I8: return ((const int8_t *)c->values)[k];
        case ORC_I16: return ((const int16_t *)c->values)[k];
        case ORC_I32: return ((const int32_t *)c->values)[k];
        case ORC_I64: return ((const int64_t *)c->values)[k];
        case ORC_U8: return ((const uint8_t *)c->values)[k];
        case ORC_U16: return ((const uint16_t *)c->values)[k];
        case ORC_U32: return ((const uint32_t *)c->values)[k];
        case ORC_U64: return (int64_t)((const uint64_t *)c->values)[k];
    }
    return 0;
}

static inline double col_f64(const orc_col *c, int64_t i) {
    int64_t k = c->offset + i;
    if (c->type == ORC_F32) return (double)((const float *)c->values)[k];
    if (c->type == ORC_F64) return ((const double *)c->values)[k];
    if (c->type == ORC_U64) return (double)((const uint64_t *)c->values)[k];
    return (double)col_i64(c, i);
}

/* ------------------------------------------------------------------------------------------
 * Hash aggregate  (base_aggregate.cpp:23-45 row loop; single_numerical_hash_aggregate.cpp:15-46;
 * multi_numerical_hash_aggregate.cpp:17-43; one_group_aggregate.cpp:9-26)
 * ---------------------------------------------------------------------------------------- */
#define ORC_MAX_KEYS 8
#define ORC_MAX_FUNCS 64

typedef struct {
    uint8_t has; /* shared_ptr != nullptr in the reference */
    union {
        uint64_t u;
        int64_t i;
        double d;
        i128 h;
    } v;
    uint64_t cnt; /* AVG pair.second; COUNT value */
} orc_acc;

typedef struct orc_agg {
    int kind, n_keys, n_funcs;
    int key_types[ORC_MAX_KEYS];
    int funcs[ORC_MAX_FUNCS];
    int in_types[ORC_MAX_FUNCS];
    int in_flags[ORC_MAX_FUNCS];
    /* groups, insertion ordered */
    int64_t n_groups, cap_groups;
    uint64_t *gkeys;   /* n_groups * n_keys */
    uint8_t *gnull;    /* n_groups * n_keys */
    orc_acc *gacc;     /* n_groups * n_funcs */
    int64_t null_group; /* Single: index of the NULL-key group or -1 */
    /* open addressing index */
    int64_t tcap;      /* power of two */
    int64_t *tslot;    /* group index or -1 */
} orc_agg;

static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

orc_agg *orc_agg_create(int kind, int n_keys, const int *key_types, int n_funcs, const int *funcs,
                        const int *in_types, const int *in_flags) {
    if (n_keys > ORC_MAX_KEYS || n_funcs > ORC_MAX_FUNCS) return NULL;
    orc_agg *a = (orc_agg *)calloc(1, sizeof(orc_agg));
    a->kind = kind; a->n_keys = n_keys; a->n_funcs = n_funcs;
    for (int i = 0; i < n_keys; i++) a->key_types[i] = key_types[i];
    for (int i = 0; i < n_funcs; i++) {
        a->funcs[i] = funcs[i]; a->in_types[i] = in_types[i]; a->in_flags[i] = in_flags ? in_flags[i] : 0;
    }
    a->null_group = -1;
    a->tcap = 1024;
    a->tslot = (int64_t *)malloc(sizeof(int64_t) * a->tcap);
    for (int64_t i = 0; i < a->tcap; i++) a->tslot[i] = -1;
    return a;
}

void orc_agg_destroy(orc_agg *a) {
    if (!a) return;
    free(a->gkeys); free(a->gnull); free(a->gacc); free(a->tslot); free(a);
}

static int64_t agg_new_group(orc_agg *a) {
    if (a->n_groups == a->cap_groups) {
        int64_t nc = a->cap_groups ? a->cap_groups * 2 : 1024;
        int nk = a->n_keys ? a->n_keys : 1, nf = a->n_funcs ? a->n_funcs : 1;
        a->gkeys = (uint64_t *)realloc(a->gkeys, sizeof(uint64_t) * nc * nk);
        a->gnull = (uint8_t *)realloc(a->gnull, nc * nk);
        a->gacc = (orc_acc *)realloc(a->gacc, sizeof(orc_acc) * nc * nf);
        a->cap_groups = nc;
    }
    int64_t g = a->n_groups++;
    memset(&a->gacc[g * (a->n_funcs ? a->n_funcs : 1)], 0, sizeof(orc_acc) * (a->n_funcs ? a->n_funcs : 1));
    return g;
}

static uint64_t agg_hash(const orc_agg *a, const uint64_t *k, const uint8_t *nl) {
    /* multi_numerical_hash_aggregate.h:20-34 combines per-column hashes boost-style; the hash
     * VALUE is not observable (iteration order is unspecified), only equality is. */
    uint64_t seed = (uint64_t)a->n_keys;
    for (int i = 0; i < a->n_keys; i++) {
        uint64_t h = nl[i] ? 0 : mix64(k[i]);
        seed ^= h + 0x9e3779b9ULL + (seed << 6) + (seed >> 2);
    }
    return mix64(seed);
}

static inline int agg_key_eq(const orc_agg *a, int64_t g, const uint64_t *k, const uint8_t *nl) {
    /* IntKeyValue::operator== multi_numerical_hash_aggregate.h:15-17: null == null regardless of value */
    const uint64_t *gk = &a->gkeys[g * a->n_keys];
    const uint8_t *gn = &a->gnull[g * a->n_keys];
    for (int i = 0; i < a->n_keys; i++) {
        if (gn[i] != nl[i]) return 0;
        if (!nl[i] && gk[i] != k[i]) return 0;
    }
    return 1;
}

static void agg_rehash(orc_agg *a) {
    int64_t nc = a->tcap * 2;
    int64_t *ns = (int64_t *)malloc(sizeof(int64_t) * nc);
    for (int64_t i = 0; i < nc; i++) ns[i] = -1;
    for (int64_t g = 0; g < a->n_groups; g++) {
        if (g == a->null_group) continue;
        uint64_t h = agg_hash(a, &a->gkeys[g * a->n_keys], &a->gnull[g * a->n_keys]) & (nc - 1);
        while (ns[h] >= 0) h = (h + 1) & (nc - 1);
        ns[h] = g;
    }
    free(a->tslot); a->tslot = ns; a->tcap = nc;
}

/* GetOrCreateEntry */
static int64_t agg_lookup(orc_agg *a, const uint64_t *k, const uint8_t *nl, int *is_new) {
    if (a->kind == ORC_SINGLE && nl[0]) { /* single_numerical_hash_aggregate.cpp:24-32 */
        *is_new = 0;
        if (a->null_group < 0) {
            a->null_group = agg_new_group(a);
            a->gkeys[a->null_group] = k[0]; a->gnull[a->null_group] = 1;
            *is_new = 1;
        }
        return a->null_group;
    }
    uint64_t h = agg_hash(a, k, nl) & (a->tcap - 1);
    for (;;) {
        int64_t g = a->tslot[h];
        if (g < 0) break;
        if (agg_key_eq(a, g, k, nl)) { *is_new = 0; return g; }
        h = (h + 1) & (a->tcap - 1);
    }
    int64_t g = agg_new_group(a);
    memcpy(&a->gkeys[g * a->n_keys], k, sizeof(uint64_t) * a->n_keys);
    memcpy(&a->gnull[g * a->n_keys], nl, a->n_keys);
    a->tslot[h] = g;
    *is_new = 1;
    if (a->n_groups * 10 > a->tcap * 7) agg_rehash(a);
    return g;
}

/* MinMaxFunc::Update agg_funcs.h:187-201 -- `if ((row_val < *last) ^ is_max) *last = row_val;`
 * evaluated in the NATIVE type (signed / unsigned / float compare). */
static inline void minmax_update(orc_acc *s, const orc_col *c, int64_t i, int is_max) {
    if (col_is_null(c, i)) return;
    if (is_float_type(c->type)) {
        double row = col_f64(c, i); /* f32 -> f64 widening is exact and order preserving */
        if (!s->has) { s->has = 1; s->v.d = row; return; }
        if ((row < s->v.d) ^ is_max) s->v.d = row;
    } else if (is_unsigned_type(c->type)) {
        uint64_t row = (uint64_t)col_i64(c, i);
        if (!s->has) { s->has = 1; s->v.u = row; return; }
        if ((row < s->v.u) ^ is_max) s->v.u = row;
    } else {
        int64_t row = col_i64(c, i);
        if (!s->has) { s->has = 1; s->v.i = row; return; }
        if ((row < s->v.i) ^ is_max) s->v.i = row;
    }
}

/* SumFunc::Update agg_funcs.h:294-305 / SumOverflowFunc::Update :336-347.
 * Accumulator type by input type: agg_func_factory.cpp:108-149. */
static inline void sum_update(orc_acc *s, const orc_col *c, int64_t i) {
    if (col_is_null(c, i)) return;
    s->has = 1;
    switch (c->type) {
        case ORC_F32: case ORC_F64: s->v.d = s->cnt ? s->v.d + col_f64(c, i) : col_f64(c, i); break;
        case ORC_I64: s->v.h += (i128)col_i64(c, i); break;                   /* hugeint, Convert<int64> */
        case ORC_U64: s->v.h += (i128)(u128)(uint64_t)col_i64(c, i); break;   /* hugeint, Convert<uint64> */
        case ORC_I32:
            if (c->flags & ORC_FLAG_SUM32) { /* time32: SumFunc<Time32Type,int32_t> wraps in int32 */
                s->v.i = (int64_t)(int32_t)((uint32_t)s->v.i + (uint32_t)col_i64(c, i)); break;
            } /* fallthrough */
        default: s->v.u += (uint64_t)col_i64(c, i); break; /* int64/uint64 wraparound accumulate */
    }
    s->cnt++;
}

/* AvgFunc::Update agg_funcs.h:455-467: T_SUM by input type, agg_func_factory.cpp:177-247 */
static inline void avg_update(orc_acc *s, const orc_col *c, int64_t i) {
    if (col_is_null(c, i)) return;
    s->has = 1;
    switch (c->type) {
        case ORC_F32: case ORC_F64: s->v.d = s->cnt ? s->v.d + col_f64(c, i) : col_f64(c, i); break;
        case ORC_I64: s->v.h += (i128)col_i64(c, i); break;
        case ORC_U64: s->v.h += (i128)(u128)(uint64_t)col_i64(c, i); break;
        default: s->v.u += (uint64_t)col_i64(c, i); break;
    }
    s->cnt++;
}

static inline void acc_update(orc_agg *a, orc_acc *s, int f, const orc_col *c, int64_t i) {
    switch (a->funcs[f]) {
        case ORC_COUNT_STAR: s->has = 1; s->cnt++; break;                    /* agg_funcs.h:106-114 */
        case ORC_COUNT: s->has = 1; s->cnt += col_is_null(c, i) ? 0 : 1; break; /* :139-149 */
        case ORC_MIN: minmax_update(s, c, i, 0); break;
        case ORC_MAX: minmax_update(s, c, i, 1); break;
        case ORC_SUM: sum_update(s, c, i); break;
        case ORC_AVG: avg_update(s, c, i); break;
    }
}

int orc_agg_next(orc_agg *a, int64_t nrows, const orc_col *keys, const orc_col *inputs) {
    if (a->kind == ORC_ONE_GROUP) {
        /* one_group_aggregate.cpp:9-26: the single group exists from the first Next() on */
        if (a->n_groups == 0) { agg_new_group(a); }
        orc_acc *row = &a->gacc[0];
        for (int f = 0; f < a->n_funcs; f++) {
            if (a->funcs[f] == ORC_COUNT_STAR) { row[f].has = 1; row[f].cnt += (uint64_t)nrows; continue; } /* :119-122 */
            if (a->funcs[f] == ORC_COUNT) row[f].has = 1;
            for (int64_t i = 0; i < nrows; i++) acc_update(a, &row[f], f, &inputs[f], i);
        }
        return 0;
    }
    uint64_t k[ORC_MAX_KEYS]; uint8_t nl[ORC_MAX_KEYS];
    for (int64_t i = 0; i < nrows; i++) {
        for (int j = 0; j < a->n_keys; j++) {
            nl[j] = (uint8_t)col_is_null(&keys[j], i);
            k[j] = col_key_bits(&keys[j], i);
        }
        int is_new;
        int64_t g = agg_lookup(a, k, nl, &is_new);
        orc_acc *row = &a->gacc[g * a->n_funcs];
        /* Init(row) and Update() have the same effect on a fresh accumulator */
        for (int f = 0; f < a->n_funcs; f++) acc_update(a, &row[f], f, &inputs[f], i);
    }
    return 0;
}

int64_t orc_agg_ngroups(const orc_agg *a) { return a->n_groups; }

/* Output order: insertion order, Single's NULL group LAST (single_numerical_hash_aggregate.cpp:58-60).
 * (The reference's order among non-null groups is robin_hood iteration order = unspecified.) */
static int64_t out_group(const orc_agg *a, int64_t r) {
    if (a->null_group < 0) return r;
    if (r == a->n_groups - 1) return a->null_group;
    return r < a->null_group ? r : r + 1;
}

/* GroupBuilder agg_funcs.h:544-578: key value at the creating row, NULL key -> NULL */
void orc_agg_keys(const orc_agg *a, int j, uint64_t *vals, uint8_t *valid) {
    for (int64_t r = 0; r < a->n_groups; r++) {
        int64_t g = out_group(a, r);
        vals[r] = a->gkeys[g * a->n_keys + j];
        valid[r] = !a->gnull[g * a->n_keys + j];
    }
}

/* Hugeint::TryCast<double> huge_int.cpp:395-406 */
static double hugeint_to_double(i128 x) {
    uint64_t lower = (uint64_t)(u128)x;
    int64_t upper = (int64_t)(x >> 64);
    if (upper == -1) return -(double)(UINT64_MAX - lower) - 1;
    return (double)lower + (double)upper * (double)UINT64_MAX;
}

/* hugeint_try_cast_integer huge_int.cpp:334-355 (note: INT64_MIN does NOT fit -- strict '>') */
static int hugeint_fits_i64(i128 x, int64_t *out) {
    uint64_t lower = (uint64_t)(u128)x;
    int64_t upper = (int64_t)(x >> 64);
    if (upper == 0) { if (lower <= (uint64_t)INT64_MAX) { *out = (int64_t)lower; return 1; } return 0; }
    if (upper == -1) {
        if (lower > UINT64_MAX - (uint64_t)INT64_MAX) { *out = -(int64_t)(UINT64_MAX - lower + 1); return 1; }
    }
    return 0;
}
static int hugeint_fits_u64(i128 x, uint64_t *out) {
    uint64_t lower = (uint64_t)(u128)x;
    int64_t upper = (int64_t)(x >> 64);
    if (upper == 0) { *out = lower; return 1; }
    return 0; /* a sum of uint64 values is never negative */
}

/* Writes the result column of function f.  vals must hold 16 bytes per group (DEC128 uses all 16,
 * everything else the first 8 -- or 4 for F32/I32 -- of each 16-byte cell).  Returns the output kind. */
int orc_agg_func(const orc_agg *a, int f, void *vals_, uint8_t *valid) {
    uint8_t *vals = (uint8_t *)vals_;
    int func = a->funcs[f], t = a->in_types[f];
    int64_t n = a->n_groups;
    memset(vals, 0, (size_t)n * 16);
    if (func == ORC_COUNT_STAR || func == ORC_COUNT) {
        for (int64_t r = 0; r < n; r++) {
            const orc_acc *s = &a->gacc[out_group(a, r) * a->n_funcs + f];
            memcpy(vals + r * 16, &s->cnt, 8); valid[r] = 1;
        }
        return ORC_OUT_U64;
    }
    if (func == ORC_MIN || func == ORC_MAX) {
        for (int64_t r = 0; r < n; r++) {
            const orc_acc *s = &a->gacc[out_group(a, r) * a->n_funcs + f];
            valid[r] = s->has; if (s->has) memcpy(vals + r * 16, &s->v, 8);
        }
        return is_float_type(t) ? ORC_OUT_F64 : (is_unsigned_type(t) ? ORC_OUT_U64 : ORC_OUT_I64);
    }
    if (func == ORC_SUM) {
        if (t == ORC_I64 || t == ORC_U64) {
            /* SumOverflowFunc::Summarize agg_funcs.h:358-397: the first group that does not fit flips the
             * WHOLE column to decimal128(38,0).  (CopyBuilder's null test, :425-434, consults the input
             * iterator instead of the builder -- a reference bug; we keep the builder's validity.) */
            int overflow = 0;
            for (int64_t r = 0; r < n && !overflow; r++) {
                const orc_acc *s = &a->gacc[out_group(a, r) * a->n_funcs + f];
                if (!s->has) continue;
                int64_t i; uint64_t u;
                if (!(t == ORC_I64 ? hugeint_fits_i64(s->v.h, &i) : hugeint_fits_u64(s->v.h, &u))) overflow = 1;
            }
            for (int64_t r = 0; r < n; r++) {
                const orc_acc *s = &a->gacc[out_group(a, r) * a->n_funcs + f];
                valid[r] = s->has; if (!s->has) continue;
                if (overflow) memcpy(vals + r * 16, &s->v.h, 16);
                else { uint64_t lo = (uint64_t)(u128)s->v.h; memcpy(vals + r * 16, &lo, 8); }
            }
            return overflow ? ORC_OUT_DEC128 : (t == ORC_I64 ? ORC_OUT_I64 : ORC_OUT_U64);
        }
        for (int64_t r = 0; r < n; r++) {
            const orc_acc *s = &a->gacc[out_group(a, r) * a->n_funcs + f];
            valid[r] = s->has; if (s->has) memcpy(vals + r * 16, &s->v, 8);
        }
        if (is_float_type(t)) return ORC_OUT_F64;
        if (t == ORC_I32 && (a->in_flags[f] & ORC_FLAG_SUM32)) return ORC_OUT_I32;
        return is_unsigned_type(t) ? ORC_OUT_U64 : ORC_OUT_I64;
    }
    /* AVG: AvgFunc::Summarize agg_funcs.h:482-491, ComputeAvg :519-540 */
    int out_f32 = (t == ORC_I8 || t == ORC_I16 || t == ORC_U8 || t == ORC_U16); /* factory :179-196 */
    for (int64_t r = 0; r < n; r++) {
        const orc_acc *s = &a->gacc[out_group(a, r) * a->n_funcs + f];
        valid[r] = s->has; if (!s->has) continue;
        double avg;
        if (t == ORC_I64 || t == ORC_U64) {
            i128 cnt = (i128)(int64_t)s->cnt;           /* hugeint_t(int64_t) implicit ctor */
            i128 q = s->v.h / cnt, rem = s->v.h % cnt;  /* DivMod huge_int.cpp:218-265: trunc, rem has lhs sign */
            avg = hugeint_to_double(q);
            avg += hugeint_to_double(rem) / (double)s->cnt; /* `rem_double / count`, count is uint64 */
        } else if (is_float_type(t)) {
            avg = s->v.d / (double)s->cnt;
        } else if (is_unsigned_type(t)) {
            avg = (double)s->v.u / (double)s->cnt;      /* uint64 sum / double */
        } else {
            avg = (double)s->v.i / (double)s->cnt;      /* int64 sum / double */
        }
        if (out_f32) { float f32 = (float)avg; memcpy(vals + r * 16, &f32, 4); }
        else memcpy(vals + r * 16, &avg, 8);
    }
    return out_f32 ? ORC_OUT_F32 : ORC_OUT_F64;
}

/* ------------------------------------------------------------------------------------------
 * Filter: comparison predicate -> byte mask -> per-column compaction.
 * vinum/core/expressions.py:30-36 (NumPy lambdas), vinum/arrow/record_batch.py:85-90,101-125,
 * vinum/core/algebra.py:119-123.
 * NumPy semantics restated: a column WITH nulls reaches NumPy as float64 with NaN (record_batch.py:
 * 112-118) so every comparison on a null row is False (!= is True); an int column without nulls
 * compares as integers against an int literal and as float64 against a float literal; a float32
 * column (with or without nulls) compares in float32 against a (weak) Python scalar.
 * ---------------------------------------------------------------------------------------- */
enum { ORC_EQ = 0, ORC_NE, ORC_GT, ORC_GE, ORC_LT, ORC_LE };

#define CMP_BODY(a, b)                     \
    switch (op) {                          \
        case ORC_EQ: r = (a) == (b); break; \
        case ORC_NE: r = (a) != (b); break; \
        case ORC_GT: r = (a) > (b); break;  \
        case ORC_GE: r = (a) >= (b); break; \
        case ORC_LT: r = (a) < (b); break;  \
        default: r = (a) <= (b); break;     \
    }

/* scalar_is_float: literal is a Python float (dval) else a Python int (ival).
 * has_nulls: arr.null_count > 0 (decides the NaN-converted float64 path). */
void orc_cmp_mask(const orc_col *c, int op, int scalar_is_float, double dval, int64_t ival, int has_nulls,
                  uint8_t *mask) {
    int64_t n = c->length;
    for (int64_t i = 0; i < n; i++) {
        int r;
        if (c->type == ORC_F32) { /* float32 stays float32 in NumPy, NULL -> NaN */
            float a = (has_nulls && col_is_null(c, i)) ? NAN : ((const float *)c->values)[c->offset + i];
            float b = scalar_is_float ? (float)dval : (float)ival;
            CMP_BODY(a, b)
        } else if (is_float_type(c->type) || has_nulls || scalar_is_float) {
            double a = (has_nulls && col_is_null(c, i)) ? NAN : col_f64(c, i);
            double b = scalar_is_float ? dval : (double)ival;
            CMP_BODY(a, b)
        } else if (c->type == ORC_U64) {
            uint64_t a = (uint64_t)col_i64(c, i);
            if (ival < 0) { uint64_t big = 1, zero = 0; CMP_BODY(big, zero) } /* any uint64 > negative int */
            else { uint64_t b = (uint64_t)ival; CMP_BODY(a, b) }
        } else {
            int64_t a = col_i64(c, i), b = ival;
            CMP_BODY(a, b)
        }
        mask[i] = (uint8_t)r;
    }
}

static inline int type_width(int t) {
    switch (t) {
        case ORC_I8: case ORC_U8: return 1;
        case ORC_I16: case ORC_U16: return 2;
        case ORC_I32: case ORC_U32: case ORC_F32: return 4;
        default: return 8;
    }
}

/* pa.RecordBatch.filter(mask, null_selection_behavior='emit_null') for one fixed-width column.
 * mask_valid may be NULL (NumPy-born masks never carry nulls).  A NULL mask entry emits a NULL row.
 * out_values: n * width bytes, out_valid: n bytes (0/1).  Returns the number of rows written. */
int64_t orc_filter_col(const orc_col *c, const uint8_t *mask, const uint8_t *mask_valid, void *out_values,
                       uint8_t *out_valid) {
    int w = type_width(c->type);
    int64_t o = 0;
    const uint8_t *src = (const uint8_t *)c->values;
    uint8_t *dst = (uint8_t *)out_values;
    for (int64_t i = 0; i < c->length; i++) {
        if (mask_valid && !mask_valid[i]) {
            memset(dst + o * w, 0, (size_t)w); out_valid[o++] = 0;
        } else if (mask[i]) {
            memcpy(dst + o * w, src + (c->offset + i) * w, (size_t)w);
            out_valid[o++] = (uint8_t)!col_is_null(c, i);
        }
    }
    return o;
}

/* ------------------------------------------------------------------------------------------
 * Sort: arrow::compute::SortIndices(table, SortOptions(keys)) as called by sort.cpp:22-37.
 * Arrow (third-party, pinned 3.0.0 in setup.py:33; container 25.0.0) documents: stable; per key
 * nulls are placed at the end and NaNs just before nulls, for BOTH ascending and descending order.
 * Pinned by tests/golden/sort_*.npz generated through the real reference Sort (oracle/_ref).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int nkeys;
    const orc_col *cols;
    const int *desc;
} sort_ctx;

/* rank class: 0 = value, 1 = NaN, 2 = NULL */
static inline int sort_class(const orc_col *c, int64_t i) {
    if (col_is_null(c, i)) return 2;
    if (is_float_type(c->type)) { double v = col_f64(c, i); if (v != v) return 1; }
    return 0;
}

static int sort_cmp_rows(const sort_ctx *s, int64_t a, int64_t b) {
    for (int k = 0; k < s->nkeys; k++) {
        const orc_col *c = &s->cols[k];
        int ca = sort_class(c, a), cb = sort_class(c, b);
        if (ca != cb) return ca < cb ? -1 : 1;
        if (ca != 0) continue;
        int r;
        if (is_float_type(c->type)) { double x = col_f64(c, a), y = col_f64(c, b); r = (x > y) - (x < y); }
        else if (is_unsigned_type(c->type)) { uint64_t x = (uint64_t)col_i64(c, a), y = (uint64_t)col_i64(c, b); r = (x > y) - (x < y); }
        else { int64_t x = col_i64(c, a), y = col_i64(c, b); r = (x > y) - (x < y); }
        if (r) return s->desc[k] ? -r : r;
    }
    return 0;
}

static void merge_sort(const sort_ctx *s, int64_t *idx, int64_t *tmp, int64_t n) {
    if (n < 2) return;
    int64_t h = n / 2;
    merge_sort(s, idx, tmp, h);
    merge_sort(s, idx + h, tmp, n - h);
    int64_t i = 0, j = h, o = 0;
    while (i < h && j < n) tmp[o++] = (sort_cmp_rows(s, idx[j], idx[i]) < 0) ? idx[j++] : idx[i++];
    while (i < h) tmp[o++] = idx[i++];
    while (j < n) tmp[o++] = idx[j++];
    memcpy(idx, tmp, sizeof(int64_t) * n);
}

int orc_sort_indices(int nkeys, const orc_col *cols, const int *desc, int64_t n, int64_t *out_idx) {
    sort_ctx s = {nkeys, cols, desc};
    int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (n ? n : 1));
    for (int64_t i = 0; i < n; i++) out_idx[i] = i;
    merge_sort(&s, out_idx, tmp, n);
    free(tmp);
    return 0;
}

/* compute::Take for one fixed-width column (sort.cpp:40) */
void orc_take_col(const orc_col *c, const int64_t *idx, int64_t n, void *out_values, uint8_t *out_valid) {
    int w = type_width(c->type);
    const uint8_t *src = (const uint8_t *)c->values;
    uint8_t *dst = (uint8_t *)out_values;
    for (int64_t i = 0; i < n; i++) {
        memcpy(dst + i * w, src + (c->offset + idx[i]) * w, (size_t)w);
        out_valid[i] = (uint8_t)!col_is_null(c, idx[i]);
    }
}
