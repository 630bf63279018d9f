// Arrow-level operator handles: the entry points a binding of the reference's native boundary calls
// (vinum/core/vinum_lib.cpp:54-142: next(batch) per RecordBatch, one result()/sorted()).
// Batches cross as Arrow C Data Interface structs (no libarrow dependency); columns are staged into HBM,
// the device-level operators run there, and results come back as freshly allocated Arrow arrays.
#include <algorithm>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "vnm_agg.hpp"

using namespace vnm;

namespace {

// ---- Arrow C Data Interface helpers ------------------------------------------------------------------
struct ColType {
    int type = -1;   // vnm_type or -1 (unsupported)
    int flags = 0;
    std::string format;
};

ColType parse_format(const char* f) {
    ColType c;
    c.format = f ? f : "";
    const std::string& s = c.format;
    if (s == "c") c.type = VNM_I8;
    else if (s == "C") c.type = VNM_U8;
    else if (s == "s") c.type = VNM_I16;
    else if (s == "S") c.type = VNM_U16;
    else if (s == "i") c.type = VNM_I32;
    else if (s == "I") c.type = VNM_U32;
    else if (s == "l") c.type = VNM_I64;
    else if (s == "L") c.type = VNM_U64;
    else if (s == "f") c.type = VNM_F32;
    else if (s == "g") c.type = VNM_F64;
    else if (s == "tdD") c.type = VNM_I32;                              // date32
    else if (s == "tdm") c.type = VNM_I64;                              // date64
    else if (s == "tts" || s == "ttm") { c.type = VNM_I32; c.flags = VNM_FLAG_SUM32; }  // time32
    else if (s == "ttu" || s == "ttn") c.type = VNM_I64;                // time64
    else if (s.rfind("ts", 0) == 0 && s.size() >= 4 && s[3] == ':') c.type = VNM_I64;  // timestamp
    else if (s == "tDs" || s == "tDm" || s == "tDu" || s == "tDn") c.type = VNM_I64;   // duration
    return c;
}

void release_schema(struct ArrowSchema* s) {
    if (!s || !s->release) return;
    for (int64_t i = 0; i < s->n_children; i++) {
        if (s->children[i]) {
            if (s->children[i]->release) s->children[i]->release(s->children[i]);
            free(s->children[i]);
        }
    }
    free(s->children);
    free((void*)s->format);
    free((void*)s->name);
    s->release = nullptr;
}

void release_array(struct ArrowArray* a) {
    if (!a || !a->release) return;
    for (int64_t i = 0; i < a->n_children; i++) {
        if (a->children[i]) {
            if (a->children[i]->release) a->children[i]->release(a->children[i]);
            free(a->children[i]);
        }
    }
    free(a->children);
    for (int64_t i = 0; i < a->n_buffers; i++) free((void*)a->buffers[i]);
    free(a->buffers);
    a->release = nullptr;
}

void make_schema(struct ArrowSchema* s, const std::string& format, const std::string& name, int64_t n_children) {
    memset(s, 0, sizeof(*s));
    s->format = strdup(format.c_str());
    s->name = strdup(name.c_str());
    s->flags = 2;  // ARROW_FLAG_NULLABLE
    s->n_children = n_children;
    s->children = n_children ? (struct ArrowSchema**)calloc((size_t)n_children, sizeof(void*)) : nullptr;
    for (int64_t i = 0; i < n_children; i++) s->children[i] = (struct ArrowSchema*)calloc(1, sizeof(struct ArrowSchema));
    s->release = release_schema;
}

// primitive array with malloc'ed buffers; valid may be NULL (no nulls)
void make_primitive(struct ArrowArray* a, int64_t n, int width, const void* values, const uint8_t* valid_bytes) {
    memset(a, 0, sizeof(*a));
    a->length = n;
    a->n_buffers = 2;
    a->buffers = (const void**)calloc(2, sizeof(void*));
    int64_t nulls = 0;
    if (valid_bytes) for (int64_t i = 0; i < n; i++) nulls += !valid_bytes[i];
    a->null_count = nulls;
    if (nulls) {
        uint8_t* bm = (uint8_t*)calloc((size_t)((n + 7) / 8 + 1), 1);
        for (int64_t i = 0; i < n; i++) if (valid_bytes[i]) bm[i >> 3] |= (uint8_t)(1u << (i & 7));
        a->buffers[0] = bm;
    }
    void* data = malloc((size_t)(n ? n : 1) * width);
    if (n) memcpy(data, values, (size_t)n * width);
    a->buffers[1] = data;
    a->release = release_array;
}

// primitive array whose buffers are filled by D2H copies of device-finalised Arrow buffers (values of `width` bytes + bitmap)
int make_primitive_from_device(struct ArrowArray* a, int64_t n, int width, const void* dev_values, const uint8_t* dev_bitmap, int64_t null_count) {
    memset(a, 0, sizeof(*a));
    a->length = n;
    a->n_buffers = 2;
    a->buffers = (const void**)calloc(2, sizeof(void*));
    a->null_count = null_count;
    a->release = release_array;
    void* data = malloc((size_t)(n ? n : 1) * width);
    a->buffers[1] = data;
    if (n && hipMemcpy(data, dev_values, (size_t)n * width, hipMemcpyDeviceToHost) != hipSuccess) return set_error("result: device to host copy failed");
    if (null_count) {
        uint8_t* bm = (uint8_t*)calloc((size_t)((n + 63) / 64) * 8, 1);
        a->buffers[0] = bm;
        if (hipMemcpy(bm, dev_bitmap, (size_t)((n + 7) / 8), hipMemcpyDeviceToHost) != hipSuccess) return set_error("result: device to host copy failed");
    }
    return 0;
}

void make_struct(struct ArrowArray* a, int64_t n, int64_t n_children) {
    memset(a, 0, sizeof(*a));
    a->length = n;
    a->n_buffers = 1;
    a->buffers = (const void**)calloc(1, sizeof(void*));
    a->n_children = n_children;
    a->children = n_children ? (struct ArrowArray**)calloc((size_t)n_children, sizeof(void*)) : nullptr;
    for (int64_t i = 0; i < n_children; i++) a->children[i] = (struct ArrowArray*)calloc(1, sizeof(struct ArrowArray));
    a->release = release_array;
}

struct ImportedBatch {
    struct ArrowArray arr;
    struct ArrowSchema sch;
    bool live = false;
    void drop() {
        if (!live) return;
        if (arr.release) arr.release(&arr);
        if (sch.release) sch.release(&sch);
        live = false;
    }
};

int find_child(const struct ArrowSchema* s, const std::string& name) {
    for (int64_t i = 0; i < s->n_children; i++)
        if (s->children[i]->name && name == s->children[i]->name) return (int)i;
    return -1;
}

// stage child column `ci` of an imported struct batch into HBM
int stage_child(const struct ArrowArray* batch, int ci, const ColType& t, vnm_dcol* out, hipStream_t s) {
    const struct ArrowArray* c = batch->children[ci];
    const uint8_t* validity = (c->n_buffers > 0 && c->null_count != 0) ? (const uint8_t*)c->buffers[0] : nullptr;
    const void* values = c->n_buffers > 1 ? c->buffers[1] : nullptr;
    int64_t off = c->offset + batch->offset;
    VNM_TRY(vnm_stage_column(values, validity, off, batch->length, t.type, out, (void*)s));
    out->flags = t.flags;
    return 0;
}

}  // namespace

// =========================================================================================================
// aggregate operator
// =========================================================================================================
// n bits of src (from bit src_off) to dst (from bit dst_off); dst's bits in that range are zero on entry.  Byte-wise with a shift:
// ~1 ns per byte instead of a branch per bit (5e7 nullable rows per column: 100+ ms of the small-batch route).
static void copy_bits(uint8_t* dst, int64_t dst_off, const uint8_t* src, int64_t src_off, int64_t n) {
    int64_t i = 0;
    // head: up to the next byte boundary of dst
    for (; i < n && ((dst_off + i) & 7); i++)
        if ((src[(src_off + i) >> 3] >> ((src_off + i) & 7)) & 1) dst[(dst_off + i) >> 3] |= (uint8_t)(1u << ((dst_off + i) & 7));
    const int64_t body = (n - i) / 8;
    if (body > 0) {
        uint8_t* d = dst + ((dst_off + i) >> 3);
        const int64_t so = src_off + i;
        const uint8_t* sp = src + (so >> 3);
        const int sh = (int)(so & 7);
        if (sh == 0) memcpy(d, sp, (size_t)body);
        else for (int64_t k = 0; k < body; k++) d[k] = (uint8_t)((sp[k] >> sh) | (sp[k + 1] << (8 - sh)));
        i += body * 8;
    }
    for (; i < n; i++)
        if ((src[(src_off + i) >> 3] >> ((src_off + i) & 7)) & 1) dst[(dst_off + i) >> 3] |= (uint8_t)(1u << ((dst_off + i) & 7));
}
static void set_bits(uint8_t* dst, int64_t dst_off, int64_t n) {
    int64_t i = 0;
    for (; i < n && ((dst_off + i) & 7); i++) dst[(dst_off + i) >> 3] |= (uint8_t)(1u << ((dst_off + i) & 7));
    const int64_t body = (n - i) / 8;
    if (body > 0) { memset(dst + ((dst_off + i) >> 3), 0xFF, (size_t)body); i += body * 8; }
    for (; i < n; i++) dst[(dst_off + i) >> 3] |= (uint8_t)(1u << ((dst_off + i) & 7));
}

// Child column `ci` of several imported batches -> ONE column in HBM (Table::FromRecordBatches, as Sort does, sort.cpp:16): the
// chunks' values go through the pinned ring as they are -- no joined copy on the host -- and the validity bits, where a chunk has
// NULLs, are laid end to end by byte-wise shifts and staged as one small buffer.
int stage_children(const std::vector<std::unique_ptr<ImportedBatch>>& batches, int ci, const ColType& t, int64_t total, vnm_dcol* out) {
    const int w = type_width(t.type);
    bool any_null = false;
    for (auto& b : batches) if (b->arr.children[ci]->null_count != 0 && b->arr.children[ci]->buffers[0]) any_null = true;
    std::vector<const void*> srcs;
    std::vector<size_t> sizes;
    for (auto& b : batches) {
        const struct ArrowArray* ch = b->arr.children[ci];
        if (!b->arr.length) continue;
        srcs.push_back((const uint8_t*)ch->buffers[1] + (size_t)(ch->offset + b->arr.offset) * w);
        sizes.push_back((size_t)b->arr.length * w);
    }
    memset(out, 0, sizeof(*out));
    void* dv = pool_alloc((size_t)(total ? total : 1) * w);
    if (!dv) return 1;
    out->values = dv; out->type = t.type; out->length = total; out->flags = t.flags;
    int rc = stage_chunks(dv, srcs.data(), sizes.data(), srcs.size(), nullptr);
    if (!rc && any_null) {
        std::vector<uint8_t> bits((size_t)(total + 7) / 8 + 8, 0);
        int64_t pos = 0;
        for (auto& b : batches) {
            const struct ArrowArray* ch = b->arr.children[ci];
            const int64_t off = ch->offset + b->arr.offset, len = b->arr.length;
            const uint8_t* bm = (ch->null_count != 0) ? (const uint8_t*)ch->buffers[0] : nullptr;
            if (bm) copy_bits(bits.data(), pos, bm, off, len); else set_bits(bits.data(), pos, len);
            pos += len;
        }
        const size_t nb = (size_t)(total + 7) / 8;
        uint8_t* db = (uint8_t*)pool_alloc(nb ? nb : 1);
        if (!db) rc = 1;
        if (!rc && hipMemcpyAsync(db, bits.data(), nb, hipMemcpyHostToDevice, nullptr) != hipSuccess) rc = set_error("staging the validity bits failed");
        if (!rc && hipStreamSynchronize(nullptr) != hipSuccess) rc = set_error("staging failed");   // (bits is a local)
        if (rc) pool_free(db); else out->validity = db;
    }
    if (!rc && hipStreamSynchronize(nullptr) != hipSuccess) rc = set_error("staging failed");
    if (rc) { pool_free(dv); out->values = nullptr; }
    return rc;
}

// ---- non-numeric GROUP BY keys and COUNT inputs below the C ABI (round 6) ------------------------------------------------------------
// GenericHashAggregate (vinum_cpp/src/operators/aggregate/generic_hash_aggregate.h:10-45, bound at vinum/core/vinum_lib.cpp:92-109) keys
// its map on arrow::Scalar vectors of ANY type.  Here a utf8 / large_utf8 / binary / large_binary key column becomes the int32 codes of a
// per-operator device string dictionary (vnm_strdict: the bytes cross PCIe once, equal bytes <=> equal code across all batches), a boolean
// key its 0 / 1, a decimal128 key the index of its 16-byte value in a host map; the codes are an ordinary int32 key column of the numeric
// operators (NULL stays NULL through the column's own validity bitmap), and result() turns the groups' codes back into values of the
// column's type.  COUNT over a non-numeric column counts through an int8 stand-in with the column's validity (CountFunc, agg_funcs.h:
// 129-161, reads nothing but the validity).  MIN / MAX of strings (StringMinMaxFunc, agg_funcs.h:219-261) are not here: the Python shim
// runs them as MIN / MAX of order-preserving ranks (vinum_lib._StringMinMax).
enum { GK_NUM = 0, GK_STR, GK_BOOL, GK_DEC };
struct GenKey {
    int kind = GK_NUM;
    bool wide = false;             // 64-bit offsets (large_utf8 / large_binary)
    vnm_strdict* dict = nullptr;   // GK_STR
    std::vector<std::string> by_id;       // id -> bytes (ids are handed out in chunks: not dense)
    std::map<std::pair<uint64_t, uint64_t>, int32_t> dec_id;      // GK_DEC: (low, high) words -> code
    std::vector<std::pair<uint64_t, uint64_t>> dec_val;
    ~GenKey() { if (dict) vnm_strdict_destroy(dict); }
};
static int generic_kind(const std::string& f, bool* wide) {
    *wide = f == "U" || f == "Z";
    if (f == "u" || f == "U" || f == "z" || f == "Z") return GK_STR;
    if (f == "b") return GK_BOOL;
    if (f.rfind("d:", 0) == 0 && (f.find(",128") != std::string::npos || std::count(f.begin(), f.end(), ',') == 1)) return GK_DEC;
    return GK_NUM;
}
struct ColChunk { const struct ArrowArray* col; int64_t off, len; };   // rows [off, off + len) of a child array
static const uint8_t* chunk_validity(const ColChunk& c) { return (c.col->null_count != 0 && c.col->n_buffers > 0) ? (const uint8_t*)c.col->buffers[0] : nullptr; }
// the chunks' validity laid end to end; empty = no NULL anywhere
static std::vector<uint8_t> joined_validity(const std::vector<ColChunk>& chunks, int64_t total) {
    bool any = false;
    for (auto& c : chunks) any = any || chunk_validity(c) != nullptr;
    std::vector<uint8_t> bits;
    if (!any) return bits;
    bits.assign((size_t)(total + 7) / 8 + 8, 0);
    int64_t pos = 0;
    for (auto& c : chunks) {
        const uint8_t* bm = chunk_validity(c);
        if (bm) copy_bits(bits.data(), pos, bm, c.off, c.len); else set_bits(bits.data(), pos, c.len);
        pos += c.len;
    }
    return bits;
}
// a non-numeric key column -> int32 codes in HBM
static int stage_generic_key(GenKey& g, const std::vector<ColChunk>& chunks, int64_t total, vnm_dcol* out) {
    std::vector<int32_t> codes((size_t)(total ? total : 1), -1);
    int64_t pos = 0;
    for (auto& c : chunks) {
        if (!c.len) continue;
        const uint8_t* valid = chunk_validity(c);
        auto is_valid = [&](int64_t i) { return !valid || ((valid[(c.off + i) >> 3] >> ((c.off + i) & 7)) & 1); };
        if (g.kind == GK_STR) {
            int64_t n_new = 0, new_bytes = 0;
            VNM_TRY(vnm_strdict_encode(g.dict, c.col->buffers[1], g.wide ? 1 : 0, (const uint8_t*)c.col->buffers[2], valid, c.off, c.len, codes.data() + pos, &n_new, &new_bytes, nullptr));
            if (n_new > 0) {
                std::vector<int32_t> ids((size_t)n_new), lens((size_t)n_new);
                std::vector<uint8_t> bytes((size_t)(new_bytes ? new_bytes : 1));
                VNM_TRY(vnm_strdict_fetch_new(g.dict, ids.data(), lens.data(), bytes.data()));
                size_t o = 0;
                for (int64_t i = 0; i < n_new; i++) {
                    if ((size_t)ids[(size_t)i] >= g.by_id.size()) g.by_id.resize((size_t)ids[(size_t)i] + 1);
                    g.by_id[(size_t)ids[(size_t)i]].assign((const char*)bytes.data() + o, (size_t)lens[(size_t)i]);
                    o += (size_t)lens[(size_t)i];
                }
            }
        } else if (g.kind == GK_BOOL) {
            const uint8_t* v = (const uint8_t*)c.col->buffers[1];
            for (int64_t i = 0; i < c.len; i++) if (is_valid(i)) codes[(size_t)(pos + i)] = (v[(c.off + i) >> 3] >> ((c.off + i) & 7)) & 1;
        } else {
            const uint64_t* v = (const uint64_t*)c.col->buffers[1];
            for (int64_t i = 0; i < c.len; i++) {
                if (!is_valid(i)) continue;
                const std::pair<uint64_t, uint64_t> key(v[2 * (c.off + i)], v[2 * (c.off + i) + 1]);
                auto it = g.dec_id.find(key);
                if (it == g.dec_id.end()) { it = g.dec_id.emplace(key, (int32_t)g.dec_val.size()).first; g.dec_val.push_back(key); }
                codes[(size_t)(pos + i)] = it->second;
            }
        }
        pos += c.len;
    }
    const std::vector<uint8_t> bits = joined_validity(chunks, total);
    VNM_TRY(vnm_stage_column(codes.data(), bits.empty() ? nullptr : bits.data(), 0, total, VNM_I32, out, nullptr));
    if (hipStreamSynchronize(nullptr) != hipSuccess) return set_error("staging a key column's codes failed");   // (codes / bits are locals)
    return 0;
}
// COUNT over a non-numeric column: zeros with the column's validity
static int stage_count_standin(const std::vector<ColChunk>& chunks, int64_t total, vnm_dcol* out) {
    const std::vector<uint8_t> zeros((size_t)(total ? total : 1), 0);
    const std::vector<uint8_t> bits = joined_validity(chunks, total);
    VNM_TRY(vnm_stage_column(zeros.data(), bits.empty() ? nullptr : bits.data(), 0, total, VNM_I8, out, nullptr));
    if (hipStreamSynchronize(nullptr) != hipSuccess) return set_error("staging a COUNT stand-in failed");
    return 0;
}
// the groups' codes (host: int32 values + validity bytes) -> an Arrow array of the key column's type
static void make_generic_key_array(struct ArrowArray* a, const GenKey& g, int64_t n, const int32_t* codes, const uint8_t* valid_bytes) {
    memset(a, 0, sizeof(*a));
    a->length = n;
    a->release = release_array;
    int64_t nulls = 0;
    for (int64_t i = 0; i < n; i++) nulls += !valid_bytes[i];
    a->null_count = nulls;
    uint8_t* bm = nullptr;
    if (nulls) {
        bm = (uint8_t*)calloc((size_t)((n + 7) / 8 + 1), 1);
        for (int64_t i = 0; i < n; i++) if (valid_bytes[i]) bm[i >> 3] |= (uint8_t)(1u << (i & 7));
    }
    if (g.kind == GK_STR) {
        a->n_buffers = 3;
        a->buffers = (const void**)calloc(3, sizeof(void*));
        a->buffers[0] = bm;
        size_t bytes = 0;
        for (int64_t i = 0; i < n; i++) if (valid_bytes[i] && codes[i] >= 0 && (size_t)codes[i] < g.by_id.size()) bytes += g.by_id[(size_t)codes[i]].size();
        uint8_t* data = (uint8_t*)malloc(bytes ? bytes : 1);
        int32_t* o32 = g.wide ? nullptr : (int32_t*)malloc((size_t)(n + 1) * 4);
        int64_t* o64 = g.wide ? (int64_t*)malloc((size_t)(n + 1) * 8) : nullptr;
        size_t at = 0;
        for (int64_t i = 0; i < n; i++) {
            if (g.wide) o64[i] = (int64_t)at; else o32[i] = (int32_t)at;
            if (valid_bytes[i] && codes[i] >= 0 && (size_t)codes[i] < g.by_id.size()) {
                const std::string& v = g.by_id[(size_t)codes[i]];
                memcpy(data + at, v.data(), v.size());
                at += v.size();
            }
        }
        if (g.wide) o64[n] = (int64_t)at; else o32[n] = (int32_t)at;
        a->buffers[1] = g.wide ? (void*)o64 : (void*)o32;
        a->buffers[2] = data;
        return;
    }
    a->n_buffers = 2;
    a->buffers = (const void**)calloc(2, sizeof(void*));
    a->buffers[0] = bm;
    if (g.kind == GK_BOOL) {
        uint8_t* v = (uint8_t*)calloc((size_t)((n + 7) / 8 + 1), 1);
        for (int64_t i = 0; i < n; i++) if (valid_bytes[i] && codes[i] == 1) v[i >> 3] |= (uint8_t)(1u << (i & 7));
        a->buffers[1] = v;
        return;
    }
    uint64_t* v = (uint64_t*)calloc((size_t)(n ? n : 1) * 2, 8);
    for (int64_t i = 0; i < n; i++)
        if (valid_bytes[i] && codes[i] >= 0 && (size_t)codes[i] < g.dec_val.size()) { v[2 * i] = g.dec_val[(size_t)codes[i]].first; v[2 * i + 1] = g.dec_val[(size_t)codes[i]].second; }
    a->buffers[1] = v;
}

// ---- MIN / MAX over utf8 / large_utf8 / binary / large_binary columns below the C ABI -------------------------------------------------
// StringMinMaxFunc (agg_funcs.h:219-261): NULLs skipped, a group of NULLs only gives NULL, values compared byte-wise.  The device
// aggregates order-preserving RANKS: a column's values go through ONE dictionary that grows with the operator (vnm_strdict: the bytes of
// a value cross PCIe once), and for every device-sized batch a throw-away operator of the caller's kind takes MIN / MAX of
// rank(value) per group, ranks taken over the dictionary as it stands (vnm_strdict_ranks_device).  What it leaves is one CANDIDATE per
// group and function -- the id of the winning value, which does not depend on later growth -- kept in HBM as a chunk (group keys +
// ids).  Chunks are folded (the same operator over the chunks, ids ranked against the dictionary of that moment) whenever they
// hold more than twice the groups of the last fold: nothing accumulates per batch.  result() folds once more and lines the
// candidates up with the numeric operator's groups by key.
struct StrCol { int child = -1; GenKey dict; };
struct StrFn { int func = 0; int op_idx = 0; int col = 0; };          // VNM_MIN / VNM_MAX, index among the operator's functions, index into strcols
struct CandChunk {
    int64_t n = 0;
    std::vector<void*> kv; std::vector<uint8_t*> kb; std::vector<int64_t> knull;      // per key column: values of the key's width, bitmap, NULL count
    std::vector<int32_t*> ids; std::vector<uint8_t*> idb; std::vector<int64_t> idnull;  // per string function: candidate ids (-1 = NULL), bitmap
    void drop() {
        for (void* p : kv) pool_free(p);
        for (uint8_t* p : kb) pool_free(p);
        for (int32_t* p : ids) pool_free(p);
        for (uint8_t* p : idb) pool_free(p);
        kv.clear(); kb.clear(); ids.clear(); idb.clear(); knull.clear(); idnull.clear(); n = 0;
    }
};
__global__ void strmm_invert_kernel(const int32_t* rank_of_id, int64_t top, int32_t* id_of_rank) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id < top && rank_of_id[id] >= 0) id_of_rank[rank_of_id[id]] = (int32_t)id;
}
__global__ void strmm_rank_to_id_kernel(const int32_t* ranks, const uint8_t* bitmap, int64_t n, const int32_t* id_of_rank, int32_t* ids) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool valid = !bitmap || ((bitmap[i >> 3] >> (i & 7)) & 1);
    ids[i] = valid ? id_of_rank[ranks[i]] : -1;
}

struct vnm_agg_op {
    int kind;
    std::vector<std::unique_ptr<GenKey>> gkeys;   // per group-by column: how a non-numeric key travels (GK_NUM: as it is)
    std::vector<char> standin;                    // per function: COUNT over a non-numeric column (int8 stand-in with its validity)
    std::vector<std::string> groupby, agg_cols, in_cols, out_cols;
    std::vector<int> funcs;
    vnm_agg* dev = nullptr;
    bool inited = false;
    std::vector<int> key_idx, in_idx, aggcol_key;  // child indices; agg_col -> position in groupby
    std::vector<ColType> key_t, in_t;
    // small batches wait here until they add up to a device-sized one (vnm_agg_op_next)
    std::vector<std::unique_ptr<ImportedBatch>> pending;
    int64_t pending_rows = 0;
    std::vector<std::string> formats;   // child formats of the first batch: later batches must match on the columns used
    // MIN / MAX over utf8 / binary columns (StringMinMaxFunc, agg_funcs.h:219-261), see strmm_* below
    std::vector<std::unique_ptr<StrCol>> strcols;   // one device dictionary per such column
    std::vector<StrFn> strfns;
    std::vector<int> dev_of;                        // operator function -> function of the device operator (-1: a string function)
    std::vector<CandChunk> cands;
    int64_t cand_rows = 0, cand_floor = 0;
    int n_dev = 0;                                  // functions of the device operator
};

static int agg_op_init(vnm_agg_op* h, const struct ArrowSchema* sch) {
    // a failed earlier attempt (unknown column, unsupported type) must not leave half-filled index vectors behind
    h->key_idx.clear(); h->key_t.clear(); h->in_idx.clear(); h->in_t.clear(); h->aggcol_key.clear(); h->gkeys.clear(); h->standin.clear();
    h->strcols.clear(); h->strfns.clear(); h->dev_of.clear();
    for (auto& c : h->cands) c.drop();
    h->cands.clear(); h->cand_rows = 0; h->cand_floor = 0;
    if (h->dev) { vnm_agg_destroy(h->dev); h->dev = nullptr; }
    // lookup_col_indices base_aggregate.cpp:121-131
    for (auto& c : h->groupby) {
        int i = find_child(sch, c);
        if (i < 0) return set_error("Column not found: %s", c.c_str());
        h->key_idx.push_back(i);
        h->key_t.push_back(parse_format(sch->children[i]->format));
        std::unique_ptr<GenKey> g(new GenKey());
        if (h->key_t.back().type < 0) {   // a non-numeric key: its codes travel (the column keeps its format for the result)
            g->kind = generic_kind(h->key_t.back().format, &g->wide);
            if (g->kind == GK_NUM || h->kind == VNM_ONE_GROUP) return set_error("Unsupported data type for aggregation column.");
            if (g->kind == GK_STR && !(g->dict = vnm_strdict_create())) return 1;
            h->key_t.back().type = VNM_I32;
        }
        h->gkeys.push_back(std::move(g));
    }
    for (auto& c : h->agg_cols) {
        int pos = -1;
        for (size_t j = 0; j < h->groupby.size(); j++) if (h->groupby[j] == c) pos = (int)j;
        if (find_child(sch, c) < 0) return set_error("Column not found: %s", c.c_str());
        if (pos < 0) return set_error("aggregate column %s is not a group-by column", c.c_str());
        h->aggcol_key.push_back(pos);
    }
    std::vector<int> ktypes, itypes, iflags, ids, dfuncs;
    for (auto& t : h->key_t) ktypes.push_back(t.type);
    for (size_t i = 0; i < h->funcs.size(); i++) {
        h->standin.push_back(0);
        h->dev_of.push_back((int)dfuncs.size());
        if (h->funcs[i] == VNM_COUNT_STAR || h->in_cols[i].empty()) {
            h->in_idx.push_back(-1);
            h->in_t.push_back(ColType());
            dfuncs.push_back(h->funcs[i]); itypes.push_back(VNM_U64); iflags.push_back(0); ids.push_back(-1);
            continue;
        }
        int ci = find_child(sch, h->in_cols[i]);
        if (ci < 0) return set_error("Column not found: %s", h->in_cols[i].c_str());
        ColType t = parse_format(sch->children[ci]->format);
        h->in_idx.push_back(ci);
        h->in_t.push_back(t);
        bool wide_str = false;
        if (t.type < 0 && (h->funcs[i] == VNM_MIN || h->funcs[i] == VNM_MAX) && generic_kind(t.format, &wide_str) == GK_STR) {
            // MIN / MAX of strings / binaries: not a function of the device operator (strmm_* below)
            int col = -1;
            for (size_t q = 0; q < h->strcols.size(); q++) if (h->strcols[q]->child == ci) col = (int)q;
            if (col < 0) {
                std::unique_ptr<StrCol> sc(new StrCol());
                sc->child = ci; sc->dict.kind = GK_STR; sc->dict.wide = wide_str;
                if (!(sc->dict.dict = vnm_strdict_create())) return 1;
                col = (int)h->strcols.size();
                h->strcols.push_back(std::move(sc));
            }
            StrFn sf; sf.func = h->funcs[i]; sf.op_idx = (int)i; sf.col = col;
            h->strfns.push_back(sf);
            h->dev_of.back() = -1;
            continue;
        }
        if (t.type < 0 && h->funcs[i] != VNM_COUNT) {
            switch (h->funcs[i]) {
                case VNM_MIN: case VNM_MAX: return set_error("Column data type is not supported by min()/max().");
                case VNM_SUM: return set_error("Column data type is not supported by sum().");
                default: return set_error("Column data type is not supported by avg().");
            }
        }
        if (t.type < 0) {   // COUNT over a non-numeric column: an int8 stand-in with the column's validity (its own column id: the same
                            // column may also be a key, which travels as codes)
            h->standin.back() = 1;
            h->in_t.back().type = VNM_I8;
            dfuncs.push_back(h->funcs[i]); itypes.push_back(VNM_I8); iflags.push_back(0); ids.push_back(1000000 + ci);
            continue;
        }
        dfuncs.push_back(h->funcs[i]); itypes.push_back(t.type); iflags.push_back(t.flags); ids.push_back(ci);
    }
    if (dfuncs.empty() && !h->strfns.empty()) {   // string functions only: the device operator still lists the groups (a COUNT(*) nobody reads)
        dfuncs.push_back(VNM_COUNT_STAR); itypes.push_back(VNM_U64); iflags.push_back(0); ids.push_back(-1);
    }
    h->n_dev = (int)dfuncs.size();
    h->dev = vnm_agg_create(h->kind, (int)ktypes.size(), ktypes.data(), (int)dfuncs.size(), dfuncs.data(),
                            itypes.data(), iflags.data(), ids.data());
    if (!h->dev) return 1;
    h->inited = true;
    return 0;
}

// pulls every batch of an Arrow C stream through `take` (which consumes array and schema); releases the stream
template <class F>
static int drain_stream(struct ArrowArrayStream* stream, const char* who, F take) {
    int rc = 0;
    for (;;) {
        struct ArrowArray arr;
        memset(&arr, 0, sizeof(arr));
        if (stream->get_next(stream, &arr) != 0) {
            const char* msg = stream->get_last_error ? stream->get_last_error(stream) : nullptr;
            rc = set_error("%s: the stream failed: %s", who, msg ? msg : "(no message)");
            break;
        }
        if (!arr.release) break;   // end of the stream
        struct ArrowSchema sch;
        memset(&sch, 0, sizeof(sch));
        if (stream->get_schema(stream, &sch) != 0) {
            arr.release(&arr);
            rc = set_error("%s: the stream has no schema", who);
            break;
        }
        rc = take(&arr, &sch);
        if (arr.release) arr.release(&arr);     // (not taken: an error on the way)
        if (sch.release) sch.release(&sch);
        if (rc) break;
    }
    if (stream->release) stream->release(stream);
    return rc;
}

extern "C" {

vnm_agg_op* vnm_agg_op_create(int kind, int n_groupby, const char** groupby_cols, int n_aggcols, const char** agg_cols,
                              int n_funcs, const int* func_types, const char** in_cols, const char** out_cols) {
    if (ensure_init()) return nullptr;
    if (kind < VNM_ONE_GROUP || kind > VNM_MULTI_NUMERICAL) { set_error("vnm_agg_op_create: bad operator kind %d", kind); return nullptr; }
    vnm_agg_op* h = new vnm_agg_op();
    h->kind = kind;
    for (int i = 0; i < n_groupby; i++) h->groupby.push_back(groupby_cols[i]);
    for (int i = 0; i < n_aggcols; i++) h->agg_cols.push_back(agg_cols[i]);
    for (int i = 0; i < n_funcs; i++) {
        if (func_types[i] < VNM_COUNT_STAR || func_types[i] > VNM_AVG) { set_error("Unrecognized Aggregate function type."); delete h; return nullptr; }
        h->funcs.push_back(func_types[i]);
        h->in_cols.push_back(in_cols[i] ? in_cols[i] : "");
        h->out_cols.push_back(out_cols[i] ? out_cols[i] : "");
    }
    return h;
}

void vnm_agg_op_destroy(vnm_agg_op* h) {
    if (!h) return;
    for (auto& b : h->pending) b->drop();
    if (h->dev) vnm_agg_destroy(h->dev);
    for (auto& c : h->cands) c.drop();
    delete h;
}

}  // extern "C"

// ---- string MIN / MAX: one throw-away operator over (keys, ranks) -> a chunk of candidates --------------------------------------------
struct StrmmBatch { int64_t n; std::vector<vnm_dcol> keys; std::vector<vnm_dcol> ids; };   // ids: per string FUNCTION, int32 dictionary ids with validity
static int strmm_run(vnm_agg_op* h, const std::vector<StrmmBatch>& batches, CandChunk* out) {
    const int nk = (int)h->key_idx.size(), nf = (int)h->strfns.size(), nc = (int)h->strcols.size();
    std::vector<int> ktypes, funcs, itypes, iflags, ids;
    for (auto& t : h->key_t) ktypes.push_back(t.type);
    for (int f = 0; f < nf; f++) { funcs.push_back(h->strfns[f].func); itypes.push_back(VNM_I32); iflags.push_back(0); ids.push_back(f); }
    vnm_agg* tmp = vnm_agg_create(h->kind, nk, ktypes.data(), nf, funcs.data(), itypes.data(), iflags.data(), ids.data());
    if (!tmp) return 1;
    PoolScope pool;
    int rc = 0;
    // ranks of every dictionary as it stands now, and their inverse
    std::vector<int32_t*> rank_of(nc, nullptr), id_of(nc, nullptr);
    for (int c = 0; !rc && c < nc; c++) {
        const int64_t top = std::max<int64_t>(1, vnm_strdict_ids(h->strcols[c]->dict.dict));
        rank_of[c] = (int32_t*)pool.take((size_t)top * 4);
        id_of[c] = (int32_t*)pool.take((size_t)top * 4);
        if (!rank_of[c] || !id_of[c]) { rc = 1; break; }
        if (hipMemsetAsync(rank_of[c], 0xFF, (size_t)top * 4, nullptr) != hipSuccess || hipMemsetAsync(id_of[c], 0xFF, (size_t)top * 4, nullptr) != hipSuccess) { rc = set_error("string min / max: memset failed"); break; }
        rc = vnm_strdict_ranks_device(h->strcols[c]->dict.dict, rank_of[c], nullptr);
        if (!rc) {
            strmm_invert_kernel<<<(int)((top + 255) / 256), 256, 0, nullptr>>>(rank_of[c], top, id_of[c]);
            if (hipGetLastError() != hipSuccess) rc = set_error("string min / max: kernel launch failed");
        }
    }
    for (size_t b = 0; !rc && b < batches.size(); b++) {
        const StrmmBatch& sb = batches[b];
        if (sb.n <= 0) continue;
        PoolScope bp;
        std::vector<vnm_dcol> inputs((size_t)nf);
        for (int f = 0; !rc && f < nf; f++) {
            int same = -1;   // (two functions over one column of a data batch share the rank column)
            for (int g = 0; g < f; g++) if (sb.ids[g].values == sb.ids[f].values && sb.ids[g].offset == sb.ids[f].offset) same = g;
            if (same >= 0) { inputs[f] = inputs[same]; continue; }
            int32_t* r = (int32_t*)bp.take((size_t)sb.n * 4);
            if (!r) { rc = 1; break; }
            rc = vnm_strdict_codes_to_ranks((const int32_t*)sb.ids[f].values + sb.ids[f].offset, rank_of[h->strfns[f].col], sb.n, r, nullptr);
            inputs[f] = sb.ids[f];
            inputs[f].values = r - sb.ids[f].offset;      // (the validity bitmap keeps the column's own offset)
        }
        if (!rc) rc = vnm_agg_next_device(tmp, sb.n, sb.keys.data(), inputs.data(), nullptr, nullptr);
        if (hipStreamSynchronize(nullptr) != hipSuccess && !rc) rc = set_error("string min / max: stream synchronisation failed");
    }
    int64_t n = 0;
    if (!rc) rc = vnm_agg_finish(tmp, &n, nullptr);
    if (!rc) {
        out->n = n;
        const size_t bm_bytes = (size_t)((n + 63) / 64 + 1) * 8;
        for (int j = 0; !rc && j < nk; j++) {
            void* v = pool_alloc((size_t)(n ? n : 1) * 8);
            uint8_t* bm = (uint8_t*)pool_alloc(bm_bytes);
            int64_t nulls = 0;
            out->kv.push_back(v); out->kb.push_back(bm); out->knull.push_back(0);
            if (!v || !bm) { rc = 1; break; }
            rc = vnm_agg_result_key_device(tmp, j, v, bm, &nulls, nullptr);
            out->knull.back() = nulls;
        }
        for (int f = 0; !rc && f < nf; f++) {
            int32_t* rv = (int32_t*)pool.take((size_t)(n ? n : 1) * 8);
            int32_t* idv = (int32_t*)pool_alloc((size_t)(n ? n : 1) * 4);
            uint8_t* bm = (uint8_t*)pool_alloc(bm_bytes);
            int kind = 0;
            int64_t nulls = 0;
            out->ids.push_back(idv); out->idb.push_back(bm); out->idnull.push_back(0);
            if (!rv || !idv || !bm) { rc = 1; break; }
            rc = vnm_agg_result_func_device(tmp, f, rv, bm, &kind, &nulls, nullptr);
            out->idnull.back() = nulls;
            if (!rc && n) {
                strmm_rank_to_id_kernel<<<(int)((n + 255) / 256), 256, 0, nullptr>>>(rv, nulls ? bm : nullptr, n, id_of[h->strfns[f].col], idv);
                if (hipGetLastError() != hipSuccess) rc = set_error("string min / max: kernel launch failed");
            }
        }
        if (hipStreamSynchronize(nullptr) != hipSuccess && !rc) rc = set_error("string min / max: stream synchronisation failed");
    }
    vnm_agg_destroy(tmp);
    if (rc) out->drop();
    return rc;
}
// all chunks -> one (the candidates of every chunk ranked against the dictionary of this moment)
static int strmm_fold(vnm_agg_op* h) {
    if (h->cands.size() <= 1) return 0;
    const int nk = (int)h->key_idx.size(), nf = (int)h->strfns.size();
    std::vector<StrmmBatch> bs;
    for (auto& ch : h->cands) {
        StrmmBatch sb; sb.n = ch.n;
        for (int j = 0; j < nk; j++) {
            vnm_dcol d; memset(&d, 0, sizeof(d));
            d.values = ch.kv[j]; d.validity = ch.knull[j] ? ch.kb[j] : nullptr; d.length = ch.n; d.type = h->key_t[j].type; d.flags = h->key_t[j].flags;
            sb.keys.push_back(d);
        }
        for (int f = 0; f < nf; f++) {
            vnm_dcol d; memset(&d, 0, sizeof(d));
            d.values = ch.ids[f]; d.validity = ch.idnull[f] ? ch.idb[f] : nullptr; d.length = ch.n; d.type = VNM_I32;
            sb.ids.push_back(d);
        }
        bs.push_back(std::move(sb));
    }
    CandChunk folded;
    VNM_TRY(strmm_run(h, bs, &folded));
    for (auto& ch : h->cands) ch.drop();
    h->cands.clear();
    h->cand_rows = folded.n;
    h->cand_floor = folded.n;
    h->cands.push_back(std::move(folded));
    return 0;
}
// the string columns of one device batch (chunks_of(child) = where its rows lie) -> one more chunk of candidates
template <class ChunksOf>
static int strmm_batch(vnm_agg_op* h, int64_t total, const std::vector<vnm_dcol>& keys, ChunksOf chunks_of) {
    if (h->strfns.empty() || total <= 0) return 0;
    const int nc = (int)h->strcols.size();
    std::vector<vnm_dcol> idcols((size_t)nc);
    int rc = 0, staged = 0;
    for (int c = 0; !rc && c < nc; c++, staged += !rc) rc = stage_generic_key(h->strcols[c]->dict, chunks_of(h->strcols[c]->child), total, &idcols[c]);
    if (!rc) {
        StrmmBatch sb; sb.n = total; sb.keys = keys;
        for (auto& f : h->strfns) sb.ids.push_back(idcols[f.col]);
        CandChunk ch;
        rc = strmm_run(h, std::vector<StrmmBatch>{sb}, &ch);
        if (!rc) { h->cand_rows += ch.n; h->cands.push_back(std::move(ch)); }
    }
    for (int c = 0; c < staged; c++) vnm_free_column(&idcols[c]);
    if (!rc && h->cands.size() > 1 && h->cand_rows > 2 * std::max<int64_t>(h->cand_floor, (int64_t)1 << 16)) rc = strmm_fold(h);
    return rc;
}

// result(): the folded candidates lined up with the numeric operator's groups by key -> per string function, the winning id of every group
static int strmm_join(vnm_agg_op* h, int64_t n, std::vector<std::vector<int32_t>>* ids_of_fn) {
    const int nk = (int)h->key_idx.size(), nf = (int)h->strfns.size();
    ids_of_fn->assign((size_t)nf, std::vector<int32_t>((size_t)(n ? n : 1), -1));
    if (h->cands.empty() || n == 0) return 0;
    const CandChunk& ch = h->cands[0];
    const int64_t m = ch.n;
    std::vector<int> width((size_t)nk);
    size_t klen = 0;
    for (int j = 0; j < nk; j++) { width[j] = type_width(h->key_t[j].type); klen += 1 + (size_t)width[j]; }
    // candidates: key bytes -> row
    std::vector<std::vector<uint8_t>> cv((size_t)nk), cb((size_t)nk);
    for (int j = 0; j < nk; j++) {
        cv[j].resize((size_t)(m ? m : 1) * width[j]);
        if (m && hipMemcpy(cv[j].data(), ch.kv[j], (size_t)m * width[j], hipMemcpyDeviceToHost) != hipSuccess) return set_error("string min / max: device to host copy failed");
        if (ch.knull[j]) {
            cb[j].resize((size_t)(m + 7) / 8);
            if (hipMemcpy(cb[j].data(), ch.kb[j], cb[j].size(), hipMemcpyDeviceToHost) != hipSuccess) return set_error("string min / max: device to host copy failed");
        }
    }
    auto key_of = [&](std::string& k, int64_t r, const std::vector<const uint8_t*>& vals, const std::vector<int>& stride, const std::vector<std::function<bool(int64_t)>>& valid) {
        k.assign(klen, '\0');
        size_t at = 0;
        for (int j = 0; j < nk; j++) {
            const bool v = valid[j](r);
            k[at++] = v ? 1 : 0;
            if (v) memcpy(&k[at], vals[j] + (size_t)r * stride[j], (size_t)width[j]);
            at += (size_t)width[j];
        }
    };
    std::unordered_map<std::string, int64_t> row_of;
    row_of.reserve((size_t)m * 2 + 16);
    {
        std::vector<const uint8_t*> vals; std::vector<int> stride; std::vector<std::function<bool(int64_t)>> valid;
        for (int j = 0; j < nk; j++) {
            vals.push_back(cv[j].data()); stride.push_back(width[j]);
            const uint8_t* bm = ch.knull[j] ? cb[j].data() : nullptr;
            valid.push_back([bm](int64_t r) { return !bm || ((bm[r >> 3] >> (r & 7)) & 1); });
        }
        std::string k;
        for (int64_t r = 0; r < m; r++) { key_of(k, r, vals, stride, valid); row_of.emplace(k, r); }
    }
    std::vector<std::vector<int32_t>> cid((size_t)nf);
    for (int f = 0; f < nf; f++) {
        cid[f].resize((size_t)(m ? m : 1));
        if (m && hipMemcpy(cid[f].data(), ch.ids[f], (size_t)m * 4, hipMemcpyDeviceToHost) != hipSuccess) return set_error("string min / max: device to host copy failed");
    }
    // the numeric operator's groups
    std::vector<std::vector<uint64_t>> mv((size_t)nk);
    std::vector<std::vector<uint8_t>> mb((size_t)nk);
    for (int j = 0; j < nk; j++) {
        mv[j].resize((size_t)n); mb[j].resize((size_t)n);
        VNM_TRY(vnm_agg_result_key(h->dev, j, mv[j].data(), mb[j].data()));
    }
    std::vector<const uint8_t*> vals; std::vector<int> stride; std::vector<std::function<bool(int64_t)>> valid;
    for (int j = 0; j < nk; j++) {
        vals.push_back((const uint8_t*)mv[j].data()); stride.push_back(8);
        const uint8_t* vb = mb[j].data();
        valid.push_back([vb](int64_t r) { return vb[r] != 0; });
    }
    std::string k;
    for (int64_t r = 0; r < n; r++) {
        key_of(k, r, vals, stride, valid);
        auto it = row_of.find(k);
        if (it == row_of.end()) continue;
        for (int f = 0; f < nf; f++) (*ids_of_fn)[f][(size_t)r] = cid[f][(size_t)it->second];
    }
    return 0;
}
static void strmm_column(vnm_agg_op* h, int f, int64_t n, const std::vector<int32_t>& ids, struct ArrowArray* out) {
    std::vector<uint8_t> vb((size_t)(n ? n : 1));
    for (int64_t r = 0; r < n; r++) vb[(size_t)r] = ids[(size_t)r] >= 0;
    make_generic_key_array(out, h->strcols[(size_t)h->strfns[(size_t)f].col]->dict, n, ids.data(), vb.data());
}

extern "C" {

// one device batch out of everything that is pending
static int agg_op_flush(vnm_agg_op* h) {
    if (h->pending.empty()) return 0;
    const int64_t total = h->pending_rows;
    std::vector<vnm_dcol> keys(h->key_idx.size()), inputs((size_t)std::max(h->n_dev, 1));
    for (auto& d : inputs) { memset(&d, 0, sizeof(vnm_dcol)); d.length = total; }
    std::map<std::pair<int, int>, vnm_dcol> staged;   // (child, form: 0 as it is / 1 key codes / 2 COUNT stand-in)
    int rc = 0;
    auto chunks_of = [&](int ci) {
        std::vector<ColChunk> v;
        for (auto& b : h->pending) v.push_back(ColChunk{b->arr.children[ci], b->arr.children[ci]->offset + b->arr.offset, b->arr.length});
        return v;
    };
    auto get = [&](int ci, const ColType& t, vnm_dcol* out, GenKey* g = nullptr, bool standin = false) -> int {
        const int form = g && g->kind != GK_NUM ? 1 : (standin ? 2 : 0);
        auto it = staged.find({ci, form});
        if (it == staged.end()) {
            vnm_dcol d;
            if (form == 1) VNM_TRY(stage_generic_key(*g, chunks_of(ci), total, &d));
            else if (form == 2) VNM_TRY(stage_count_standin(chunks_of(ci), total, &d));
            else VNM_TRY(stage_children(h->pending, ci, t, total, &d));
            it = staged.emplace(std::make_pair(ci, form), d).first;
        }
        *out = it->second;
        return 0;
    };
    for (size_t j = 0; !rc && j < h->key_idx.size(); j++) rc = get(h->key_idx[j], h->key_t[j], &keys[j], h->gkeys[j].get());
    for (size_t i = 0; !rc && i < h->funcs.size(); i++)
        if (h->dev_of[i] >= 0 && h->in_idx[i] >= 0) rc = get(h->in_idx[i], h->in_t[i], &inputs[(size_t)h->dev_of[i]], nullptr, h->standin[i] != 0);
    if (!rc && total > 0) rc = vnm_agg_next_device(h->dev, total, keys.data(), inputs.data(), nullptr, nullptr);
    if (hipStreamSynchronize(nullptr) != hipSuccess && !rc) rc = set_error("vnm_agg_op_next: stream synchronisation failed");
    if (!rc) rc = strmm_batch(h, total, keys, chunks_of);
    for (auto& kv : staged) vnm_free_column(&kv.second);
    for (auto& b : h->pending) b->drop();
    h->pending.clear();
    h->pending_rows = 0;
    return rc;
}

// The reference streams 10 000-row batches by default (vinum/__init__.py:52, table_batch_reader.cpp:5-16); a launch per such
// batch would cost more than its rows.  Batches below 2^20 rows are therefore kept (the operator owns them: Arrow C Data
// move semantics) until 2^22 rows are waiting -- or result() is called -- and then go to the device as ONE batch; a large batch
// is staged straight from its own buffers as before.  Aggregates do not depend on where batches are cut (every accumulator
// merges commutatively, float sums are compensated), so the result is the same.
int vnm_agg_op_next(vnm_agg_op* h, struct ArrowArray* batch, struct ArrowSchema* schema) {
    if (!h || !batch || !schema) return set_error("vnm_agg_op_next: null argument");
    std::unique_ptr<ImportedBatch> ibp(new ImportedBatch());
    ImportedBatch& ib = *ibp;
    ib.arr = *batch; ib.sch = *schema; ib.live = true;
    batch->release = nullptr; schema->release = nullptr;  // ownership moved (Arrow C Data Interface move semantics)
    int rc = 0;
    if (!h->inited) {
        rc = agg_op_init(h, &ib.sch);
        if (!rc) {
            h->formats.clear();
            for (int64_t c = 0; c < ib.sch.n_children; c++) h->formats.push_back(ib.sch.children[c]->format ? ib.sch.children[c]->format : "");
        }
    } else {
        // the columns the operator reads must be where -- and what -- they were in the first batch
        auto same = [&](int ci) { return ci < (int)ib.sch.n_children && ci < (int)h->formats.size() && ib.sch.children[ci]->format &&
                                         h->formats[(size_t)ci] == ib.sch.children[ci]->format; };
        for (size_t j = 0; !rc && j < h->key_idx.size(); j++) if (!same(h->key_idx[j])) rc = set_error("vnm_agg_op_next: the batch schema changed");
        for (size_t i = 0; !rc && i < h->funcs.size(); i++) if (h->in_idx[i] >= 0 && !same(h->in_idx[i])) rc = set_error("vnm_agg_op_next: the batch schema changed");
    }
    if (rc) { ib.drop(); return rc; }
    static const int64_t small_rows = getenv("VNM_AGG_COALESCE_BELOW") ? atoll(getenv("VNM_AGG_COALESCE_BELOW")) : (1 << 20);
    static const int64_t flush_rows = getenv("VNM_AGG_COALESCE_ROWS") ? atoll(getenv("VNM_AGG_COALESCE_ROWS")) : (1 << 22);
    if (ib.arr.length < small_rows) {
        h->pending_rows += ib.arr.length;
        h->pending.push_back(std::move(ibp));
        return h->pending_rows >= flush_rows ? agg_op_flush(h) : 0;
    }
    rc = agg_op_flush(h);   // (rows that arrived earlier go first; the order does not matter to the result)
    if (rc) { ib.drop(); return rc; }
    std::vector<vnm_dcol> keys(h->key_idx.size()), inputs((size_t)std::max(h->n_dev, 1));
    for (auto& d : inputs) { memset(&d, 0, sizeof(vnm_dcol)); d.length = ib.arr.length; }
    std::map<std::pair<int, int>, vnm_dcol> staged;
    hipStream_t s = nullptr;
    auto get = [&](int ci, const ColType& t, vnm_dcol* out, GenKey* g = nullptr, bool standin = false) -> int {
        const int form = g && g->kind != GK_NUM ? 1 : (standin ? 2 : 0);
        auto it = staged.find({ci, form});
        if (it == staged.end()) {
            vnm_dcol d;
            const std::vector<ColChunk> one{ColChunk{ib.arr.children[ci], ib.arr.children[ci]->offset + ib.arr.offset, ib.arr.length}};
            if (form == 1) VNM_TRY(stage_generic_key(*g, one, ib.arr.length, &d));
            else if (form == 2) VNM_TRY(stage_count_standin(one, ib.arr.length, &d));
            else VNM_TRY(stage_child(&ib.arr, ci, t, &d, s));
            it = staged.emplace(std::make_pair(ci, form), d).first;
        }
        *out = it->second;
        return 0;
    };
    for (size_t j = 0; !rc && j < h->key_idx.size(); j++) rc = get(h->key_idx[j], h->key_t[j], &keys[j], h->gkeys[j].get());
    for (size_t i = 0; !rc && i < h->funcs.size(); i++)
        if (h->dev_of[i] >= 0 && h->in_idx[i] >= 0) rc = get(h->in_idx[i], h->in_t[i], &inputs[(size_t)h->dev_of[i]], nullptr, h->standin[i] != 0);
    if (!rc) rc = vnm_agg_next_device(h->dev, ib.arr.length, keys.data(), inputs.data(), nullptr, (void*)s);
    if (hipStreamSynchronize(s) != hipSuccess && !rc) rc = set_error("vnm_agg_op_next: stream synchronisation failed");
    if (!rc) rc = strmm_batch(h, ib.arr.length, keys, [&](int ci) { return std::vector<ColChunk>{ColChunk{ib.arr.children[ci], ib.arr.children[ci]->offset + ib.arr.offset, ib.arr.length}}; });
    for (auto& kv : staged) vnm_free_column(&kv.second);
    ib.drop();
    return rc;
}

int vnm_agg_op_next_stream(vnm_agg_op* h, struct ArrowArrayStream* stream) {
    if (!h || !stream || !stream->get_next) return set_error("vnm_agg_op_next_stream: null argument");
    return drain_stream(stream, "vnm_agg_op_next_stream", [&](struct ArrowArray* a, struct ArrowSchema* sc) { return vnm_agg_op_next(h, a, sc); });
}

int vnm_agg_op_result(vnm_agg_op* h, struct ArrowArray* out, struct ArrowSchema* out_schema) {
    if (!h || !out || !out_schema) return set_error("vnm_agg_op_result: null argument");
    if (!h->inited) return set_error("vnm_agg_op_result: no batch was ever passed to next()");
    VNM_TRY(agg_op_flush(h));
    int64_t n = 0;
    VNM_TRY(vnm_agg_finish(h->dev, &n, nullptr));
    std::vector<std::vector<int32_t>> str_ids;      // per string function: the winning dictionary id of every group (-1 = NULL)
    std::vector<int> strfn_of(h->funcs.size(), -1);
    if (!h->strfns.empty()) {
        VNM_TRY(strmm_fold(h));
        VNM_TRY(strmm_join(h, n, &str_ids));
        for (size_t f = 0; f < h->strfns.size(); f++) strfn_of[(size_t)h->strfns[f].op_idx] = (int)f;
    }
    const int64_t ncols = (int64_t)h->agg_cols.size() + (int64_t)h->funcs.size();
    make_struct(out, n, ncols);
    make_schema(out_schema, "+s", "", ncols);
    // BaseAggregate::Result on the DEVICE: every column is finalised by a kernel into Arrow-layout buffers and only those
    // cross PCIe (8 bytes per group and column instead of all accumulator words + a serial host loop).  The one case that
    // changes a column's TYPE -- an int64 / uint64 SUM overflowing 64 bits (-> decimal128, agg_funcs.h:366-389) -- makes
    // the kernel return 2 and the whole result falls back to the host finaliser below.
    if (getenv("VNM_AGG_HOST_FINALIZE") == nullptr) {
        void* dv = pool_alloc((size_t)(n ? n : 1) * 8);
        uint8_t* db = (uint8_t*)pool_alloc((size_t)((n + 63) / 64 + 1) * 8);
        if (!dv || !db) return 1;
        int rc = 0, col = 0;
        for (size_t a = 0; !rc && a < h->agg_cols.size(); a++, col++) {
            const int j = h->aggcol_key[a];
            const ColType& t = h->key_t[j];
            int64_t nulls = 0;
            rc = vnm_agg_result_key_device(h->dev, j, dv, db, &nulls, nullptr);
            if (!rc && h->gkeys[j]->kind != GK_NUM) {   // the groups' codes -> values of the key column's type
                std::vector<int32_t> codes((size_t)(n ? n : 1));
                std::vector<uint8_t> bm((size_t)((n + 7) / 8 + 8), 0xFF), vb((size_t)(n ? n : 1), 1);
                if (n && hipMemcpy(codes.data(), dv, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = set_error("result: device to host copy failed");
                if (!rc && nulls && hipMemcpy(bm.data(), db, (size_t)((n + 7) / 8), hipMemcpyDeviceToHost) != hipSuccess) rc = set_error("result: device to host copy failed");
                if (!rc) {
                    if (nulls) for (int64_t r = 0; r < n; r++) vb[(size_t)r] = (bm[(size_t)(r >> 3)] >> (r & 7)) & 1;
                    make_generic_key_array(out->children[col], *h->gkeys[j], n, codes.data(), vb.data());
                }
            } else if (!rc) rc = make_primitive_from_device(out->children[col], n, type_width(t.type), dv, db, nulls);
            make_schema(out_schema->children[col], t.format, h->agg_cols[a], 0);
        }
        for (size_t i = 0; !rc && i < h->funcs.size(); i++, col++) {
            int kind = 0;
            int64_t nulls = 0;
            if (strfn_of[i] >= 0) {
                strmm_column(h, strfn_of[i], n, str_ids[(size_t)strfn_of[i]], out->children[col]);
                make_schema(out_schema->children[col], h->in_t[i].format, h->out_cols[i], 0);
                continue;
            }
            rc = vnm_agg_result_func_device(h->dev, h->dev_of[i], dv, db, &kind, &nulls, nullptr);
            if (rc) break;
            const ColType& t = h->in_t[i];
            const int f = h->funcs[i];
            std::string fmt;
            int w = 8;
            if (f == VNM_MIN || f == VNM_MAX) { fmt = t.format; w = type_width(t.type); }
            else if (kind == VNM_OUT_U64) fmt = "L";
            else if (kind == VNM_OUT_I64) {
                fmt = "l";
                if (f == VNM_SUM && (t.format == "ttu" || t.format == "ttn" || t.format.rfind("tD", 0) == 0)) fmt = t.format;
            } else if (kind == VNM_OUT_I32) { fmt = t.format; w = 4; }
            else if (kind == VNM_OUT_F64) fmt = "g";
            else { fmt = "f"; w = 4; }
            rc = make_primitive_from_device(out->children[col], n, w, dv, db, nulls);
            make_schema(out_schema->children[col], fmt, h->out_cols[i], 0);
        }
        pool_free(dv); pool_free(db);
        if (rc == 0) return 0;
        release_array(out); release_schema(out_schema);       // rebuilt below
        if (rc != 2) return rc;
        make_struct(out, n, ncols);
        make_schema(out_schema, "+s", "", ncols);
    }
    std::vector<uint64_t> vals((size_t)(n ? n : 1));
    std::vector<uint8_t> valid((size_t)(n ? n : 1));
    std::vector<uint8_t> cells((size_t)(n ? n : 1) * 16);
    int col = 0;
    // group keys that are selected, in agg_cols order (base_aggregate.cpp:99-118, GroupBuilder agg_funcs.h:544-578)
    for (size_t a = 0; a < h->agg_cols.size(); a++, col++) {
        int j = h->aggcol_key[a];
        VNM_TRY(vnm_agg_result_key(h->dev, j, vals.data(), valid.data()));
        const ColType& t = h->key_t[j];
        int w = type_width(t.type);
        if (h->gkeys[j]->kind != GK_NUM) {
            std::vector<int32_t> codes((size_t)(n ? n : 1));
            for (int64_t r = 0; r < n; r++) codes[(size_t)r] = (int32_t)vals[(size_t)r];
            make_generic_key_array(out->children[col], *h->gkeys[j], n, codes.data(), valid.data());
            make_schema(out_schema->children[col], t.format, h->agg_cols[a], 0);
            continue;
        }
        std::vector<uint8_t> narrow((size_t)(n ? n : 1) * w);
        for (int64_t r = 0; r < n; r++) memcpy(&narrow[(size_t)r * w], &vals[r], (size_t)w);  // little endian truncation
        make_primitive(out->children[col], n, w, narrow.data(), valid.data());
        make_schema(out_schema->children[col], t.format, h->agg_cols[a], 0);
    }
    for (size_t i = 0; i < h->funcs.size(); i++, col++) {
        int kind = 0;
        if (strfn_of[i] >= 0) {
            strmm_column(h, strfn_of[i], n, str_ids[(size_t)strfn_of[i]], out->children[col]);
            make_schema(out_schema->children[col], h->in_t[i].format, h->out_cols[i], 0);
            continue;
        }
        VNM_TRY(vnm_agg_result_func(h->dev, h->dev_of[i], cells.data(), valid.data(), &kind));
        const ColType& t = h->in_t[i];
        std::string fmt;
        int w = 8;
        std::vector<uint8_t> buf((size_t)(n ? n : 1) * 16);
        auto pack = [&](int width) {
            w = width;
            for (int64_t r = 0; r < n; r++) memcpy(&buf[(size_t)r * width], &cells[(size_t)r * 16], (size_t)width);
        };
        const int f = h->funcs[i];
        if (f == VNM_MIN || f == VNM_MAX) {
            // type preserving (agg_func_factory.cpp:35-107): cells hold the value widened to 64 bits
            fmt = t.format;
            w = type_width(t.type);
            for (int64_t r = 0; r < n; r++) {
                if (t.type == VNM_F32) { double d; memcpy(&d, &cells[(size_t)r * 16], 8); float fl = (float)d; memcpy(&buf[(size_t)r * 4], &fl, 4); }
                else memcpy(&buf[(size_t)r * w], &cells[(size_t)r * 16], (size_t)w);
            }
        } else if (kind == VNM_OUT_U64) { fmt = "L"; pack(8); }
        else if (kind == VNM_OUT_I64) {
            fmt = "l";
            if (f == VNM_SUM && (t.format == "ttu" || t.format == "ttn" || t.format.rfind("tD", 0) == 0)) fmt = t.format;
            pack(8);
        } else if (kind == VNM_OUT_I32) { fmt = t.format; pack(4); }
        else if (kind == VNM_OUT_F64) { fmt = "g"; pack(8); }
        else if (kind == VNM_OUT_F32) { fmt = "f"; pack(4); }
        else { fmt = "d:38,0"; pack(16); }
        make_primitive(out->children[col], n, w, buf.data(), valid.data());
        make_schema(out_schema->children[col], fmt, h->out_cols[i], 0);
    }
    return 0;
}

}  // extern "C"

// =========================================================================================================
// sort operator
// =========================================================================================================
// ---- column kinds of a table under Sort (vnm_sort_op_sorted) ---------------------------------------------------------------------
enum { SC_NUM = 0, SC_VAR = 1, SC_BOOL = 2, SC_DEC = 3 };
struct SortColInfo {
    int kind = -1;
    ColType t;
    bool wide = false;      // large_utf8 / large_binary: 64-bit offsets
    std::string format;
};

static SortColInfo classify_sort_col(const char* f) {
    SortColInfo c;
    c.format = f ? f : "";
    c.t = parse_format(f);
    if (c.t.type >= 0) { c.kind = SC_NUM; return c; }
    if (c.format == "u" || c.format == "z") { c.kind = SC_VAR; return c; }
    if (c.format == "U" || c.format == "Z") { c.kind = SC_VAR; c.wide = true; return c; }
    if (c.format == "b") { c.kind = SC_BOOL; return c; }
    if (c.format.rfind("d:", 0) == 0) {   // d:precision,scale[,bit width]: 128 bits unless said otherwise
        const size_t c1 = c.format.find(','), c2 = c1 == std::string::npos ? std::string::npos : c.format.find(',', c1 + 1);
        if (c1 != std::string::npos && (c2 == std::string::npos || c.format.substr(c2 + 1) == "128")) c.kind = SC_DEC;
    }
    return c;
}

struct DevSortCol {
    vnm_dcol num{};              // SC_NUM
    int64_t* offs = nullptr;     // SC_VAR: total + 1 offsets (rebased to one data buffer)
    uint8_t* data = nullptr;
    uint8_t* bits = nullptr;     // SC_BOOL: the values, bit i = row i
    void* fixed = nullptr;       // SC_DEC: 16-byte values
    uint8_t* validity = nullptr; // bitmap, bit i = row i (null: no NULLs); SC_NUM keeps its own in `num`
    void free_all() {
        vnm_free_column(&num);
        pool_free(offs); pool_free(data); pool_free(bits); pool_free(fixed); pool_free(validity);
        offs = nullptr; data = nullptr; bits = nullptr; fixed = nullptr; validity = nullptr;
    }
};

// the validity bits of child `ci` of every batch laid end to end -> one device bitmap (null when the column has no NULLs)
static int stage_validity(const std::vector<std::unique_ptr<ImportedBatch>>& batches, int ci, int64_t total, uint8_t** out) {
    *out = nullptr;
    bool any_null = false;
    for (auto& b : batches) if (b->arr.children[ci]->null_count != 0 && b->arr.children[ci]->buffers[0]) any_null = true;
    if (!any_null || total == 0) return 0;
    std::vector<uint8_t> bits((size_t)(total + 7) / 8 + 8, 0);
    int64_t pos = 0;
    for (auto& b : batches) {
        const struct ArrowArray* ch = b->arr.children[ci];
        const int64_t off = ch->offset + b->arr.offset, len = b->arr.length;
        const uint8_t* bm = (ch->null_count != 0) ? (const uint8_t*)ch->buffers[0] : nullptr;
        if (bm) copy_bits(bits.data(), pos, bm, off, len); else set_bits(bits.data(), pos, len);
        pos += len;
    }
    const size_t nb = (size_t)(total + 7) / 8;
    uint8_t* db = (uint8_t*)pool_alloc(nb + 8);
    if (!db) return 1;
    if (hipMemcpy(db, bits.data(), nb, hipMemcpyHostToDevice) != hipSuccess) { pool_free(db); return set_error("staging the validity bits failed"); }
    *out = db;
    return 0;
}

static int stage_sort_col(const std::vector<std::unique_ptr<ImportedBatch>>& batches, int ci, const SortColInfo& info, int64_t total, DevSortCol* d) {
    if (info.kind == SC_NUM) return stage_children(batches, ci, info.t, total, &d->num);
    VNM_TRY(stage_validity(batches, ci, total, &d->validity));
    if (info.kind == SC_VAR) {
        std::vector<int64_t> offs((size_t)total + 1, 0);
        std::vector<const void*> srcs;
        std::vector<size_t> sizes;
        int64_t row = 0, base = 0;
        for (auto& b : batches) {
            const struct ArrowArray* ch = b->arr.children[ci];
            const int64_t off = ch->offset + b->arr.offset, len = b->arr.length;
            if (!len) continue;
            if (ch->n_buffers < 3 || !ch->buffers[1]) return set_error("Sort: a string / binary column without an offsets buffer");
            int64_t first, last;
            if (info.wide) {
                const int64_t* o = (const int64_t*)ch->buffers[1] + off;
                first = o[0]; last = o[len];
                for (int64_t i = 0; i <= len; i++) offs[(size_t)(row + i)] = base + (o[i] - first);
            } else {
                const int32_t* o = (const int32_t*)ch->buffers[1] + off;
                first = o[0]; last = o[len];
                for (int64_t i = 0; i <= len; i++) offs[(size_t)(row + i)] = base + ((int64_t)o[i] - first);
            }
            if (last > first) {
                if (!ch->buffers[2]) return set_error("Sort: a string / binary column without a data buffer");
                srcs.push_back((const uint8_t*)ch->buffers[2] + first);
                sizes.push_back((size_t)(last - first));
            }
            row += len; base += last - first;
        }
        d->offs = (int64_t*)pool_alloc(((size_t)total + 1) * 8);
        d->data = (uint8_t*)pool_alloc((size_t)(base > 0 ? base : 1));
        if (!d->offs || !d->data) return 1;
        if (hipMemcpy(d->offs, offs.data(), ((size_t)total + 1) * 8, hipMemcpyHostToDevice) != hipSuccess) return set_error("Sort: staging offsets failed");
        if (!srcs.empty()) VNM_TRY(stage_chunks(d->data, srcs.data(), sizes.data(), srcs.size(), nullptr));
        if (hipStreamSynchronize(nullptr) != hipSuccess) return set_error("staging failed");
        return 0;
    }
    if (info.kind == SC_BOOL) {
        std::vector<uint8_t> bits((size_t)(total + 7) / 8 + 8, 0);
        int64_t pos = 0;
        for (auto& b : batches) {
            const struct ArrowArray* ch = b->arr.children[ci];
            const int64_t off = ch->offset + b->arr.offset, len = b->arr.length;
            if (len && ch->buffers[1]) copy_bits(bits.data(), pos, (const uint8_t*)ch->buffers[1], off, len);
            pos += len;
        }
        d->bits = (uint8_t*)pool_alloc(bits.size());
        if (!d->bits) return 1;
        if (hipMemcpy(d->bits, bits.data(), bits.size(), hipMemcpyHostToDevice) != hipSuccess) return set_error("Sort: staging a boolean column failed");
        return 0;
    }
    // SC_DEC: 16-byte values
    std::vector<const void*> srcs;
    std::vector<size_t> sizes;
    for (auto& b : batches) {
        const struct ArrowArray* ch = b->arr.children[ci];
        if (!b->arr.length) continue;
        srcs.push_back((const uint8_t*)ch->buffers[1] + (size_t)(ch->offset + b->arr.offset) * 16);
        sizes.push_back((size_t)b->arr.length * 16);
    }
    d->fixed = pool_alloc((size_t)(total ? total : 1) * 16);
    if (!d->fixed) return 1;
    if (!srcs.empty()) VNM_TRY(stage_chunks(d->fixed, srcs.data(), sizes.data(), srcs.size(), nullptr));
    if (hipStreamSynchronize(nullptr) != hipSuccess) return set_error("staging failed");
    return 0;
}

static uint8_t* pack_bits(const uint8_t* bytes, int64_t n, int64_t* zeros) {
    uint8_t* bm = (uint8_t*)calloc((size_t)((n + 7) / 8 + 1), 1);
    int64_t z = 0;
    for (int64_t i = 0; i < n; i++) { if (bytes[i]) bm[i >> 3] |= (uint8_t)(1u << (i & 7)); else z++; }
    if (zeros) *zeros = z;
    return bm;
}

// rows `idx` of a staged column -> a freshly allocated Arrow array (host buffers)
static int take_sort_col(const DevSortCol& d, const SortColInfo& info, const int64_t* idx, int64_t n, struct ArrowArray* out) {
    const size_t n1 = (size_t)(n ? n : 1);
    if (info.kind == SC_NUM) {
        const int w = type_width(info.t.type);
        std::vector<uint8_t> hv(n1 * w), hb(n1);
        if (n > 0) {
            PoolScope pool;
            void* dv = pool.take((size_t)n * w);
            uint8_t* db = d.num.validity ? (uint8_t*)pool.take((size_t)n) : nullptr;
            if (!dv || (d.num.validity && !db)) return 1;
            VNM_TRY(vnm_take(&d.num, idx, n, dv, db, nullptr));
            VNM_HIP(hipMemcpy(hv.data(), dv, (size_t)n * w, hipMemcpyDeviceToHost));
            if (db) VNM_HIP(hipMemcpy(hb.data(), db, (size_t)n, hipMemcpyDeviceToHost));
        }
        make_primitive(out, n, w, hv.data(), d.num.validity ? hb.data() : nullptr);
        return 0;
    }
    PoolScope pool;
    std::vector<uint8_t> hvalid;
    uint8_t* dvalid = nullptr;
    if (d.validity && n > 0) {
        dvalid = (uint8_t*)pool.take((size_t)n);
        if (!dvalid) return 1;
        hvalid.resize((size_t)n);
    }
    memset(out, 0, sizeof(*out));
    out->length = n;
    out->release = release_array;
    if (info.kind == SC_VAR) {
        std::vector<int64_t> ho(n1 + 1, 0);
        std::vector<uint8_t> hd;
        if (n > 0) {
            int64_t* doffs = (int64_t*)pool.take(((size_t)n + 1) * 8);
            if (!doffs) return 1;
            uint8_t* ddata = nullptr;
            int64_t nbytes = 0;
            VNM_TRY(vnm_take_varwidth(d.offs, d.data, d.validity, idx, n, doffs, &ddata, &nbytes, dvalid, nullptr));
            PoolSlotGuard<uint8_t> g(&ddata);
            hd.resize((size_t)(nbytes > 0 ? nbytes : 1));
            VNM_HIP(hipMemcpy(ho.data(), doffs, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost));
            if (nbytes > 0) VNM_HIP(hipMemcpy(hd.data(), ddata, (size_t)nbytes, hipMemcpyDeviceToHost));
            if (dvalid) VNM_HIP(hipMemcpy(hvalid.data(), dvalid, (size_t)n, hipMemcpyDeviceToHost));
            if (!info.wide && nbytes > 0x7FFFFFFFLL) return set_error("Sort: the sorted %s column holds more than 2 GiB of bytes (use the large type)", info.format.c_str());
        }
        out->n_buffers = 3;
        out->buffers = (const void**)calloc(3, sizeof(void*));
        if (dvalid) { int64_t z = 0; out->buffers[0] = pack_bits(hvalid.data(), n, &z); out->null_count = z; }
        if (info.wide) {
            int64_t* o = (int64_t*)malloc((n1 + 1) * 8);
            memcpy(o, ho.data(), ((size_t)n + 1) * 8);
            out->buffers[1] = o;
        } else {
            int32_t* o = (int32_t*)malloc((n1 + 1) * 4);
            for (int64_t i = 0; i <= n; i++) o[i] = (int32_t)ho[(size_t)i];
            out->buffers[1] = o;
        }
        const size_t nb = hd.empty() ? 1 : hd.size();
        uint8_t* dd = (uint8_t*)malloc(nb);
        if (!hd.empty()) memcpy(dd, hd.data(), hd.size());
        out->buffers[2] = dd;
        return 0;
    }
    if (dvalid) {
        VNM_TRY(vnm_take_bits(d.validity, 0, idx, n, dvalid, nullptr));
        VNM_HIP(hipMemcpy(hvalid.data(), dvalid, (size_t)n, hipMemcpyDeviceToHost));
    }
    out->n_buffers = 2;
    out->buffers = (const void**)calloc(2, sizeof(void*));
    if (dvalid) { int64_t z = 0; out->buffers[0] = pack_bits(hvalid.data(), n, &z); out->null_count = z; }
    if (info.kind == SC_BOOL) {
        std::vector<uint8_t> hv(n1, 0);
        if (n > 0) {
            uint8_t* dv = (uint8_t*)pool.take((size_t)n);
            if (!dv) return 1;
            VNM_TRY(vnm_take_bits(d.bits, 0, idx, n, dv, nullptr));
            VNM_HIP(hipMemcpy(hv.data(), dv, (size_t)n, hipMemcpyDeviceToHost));
        }
        out->buffers[1] = pack_bits(hv.data(), n, nullptr);
        return 0;
    }
    // SC_DEC
    void* hv = malloc(n1 * 16);
    out->buffers[1] = hv;
    if (n > 0) {
        void* dv = pool.take((size_t)n * 16);
        if (!dv) return 1;
        VNM_TRY(vnm_take_fixed16(d.fixed, idx, n, dv, nullptr));
        VNM_HIP(hipMemcpy(hv, dv, (size_t)n * 16, hipMemcpyDeviceToHost));
    }
    return 0;
}

struct vnm_sort_op {
    std::vector<std::string> cols;
    std::vector<int> orders;
    std::vector<std::unique_ptr<ImportedBatch>> batches;
};

extern "C" {

vnm_sort_op* vnm_sort_op_create(int n, const char** cols, const int* orders) {
    if (ensure_init()) return nullptr;
    if (n < 1) { set_error("Sort needs at least one column"); return nullptr; }
    vnm_sort_op* h = new vnm_sort_op();
    for (int i = 0; i < n; i++) { h->cols.push_back(cols[i]); h->orders.push_back(orders[i] ? VNM_DESC : VNM_ASC); }
    return h;
}

void vnm_sort_op_destroy(vnm_sort_op* h) {
    if (!h) return;
    for (auto& b : h->batches) b->drop();
    delete h;
}

// Sort::Next (sort.cpp:11-13): only buffers the batch
int vnm_sort_op_next(vnm_sort_op* h, struct ArrowArray* batch, struct ArrowSchema* schema) {
    if (!h || !batch || !schema) return set_error("vnm_sort_op_next: null argument");
    auto ib = std::make_unique<ImportedBatch>();
    ib->arr = *batch; ib->sch = *schema; ib->live = true;
    batch->release = nullptr; schema->release = nullptr;
    h->batches.push_back(std::move(ib));
    return 0;
}

int vnm_sort_op_next_stream(vnm_sort_op* h, struct ArrowArrayStream* stream) {
    if (!h || !stream || !stream->get_next) return set_error("vnm_sort_op_next_stream: null argument");
    return drain_stream(stream, "vnm_sort_op_next_stream", [&](struct ArrowArray* a, struct ArrowSchema* sc) { return vnm_sort_op_next(h, a, sc); });
}

// Sort::Sorted (sort.cpp:15-63).  limit > 0: only the first `limit` rows are produced (LIMIT pushed into the
// sort; the reference sorts everything and slices later, same rows).
// Column types (round 5: every type the reference's own tests sort or carry along crosses the ABI and is handled on the device):
//   numeric / temporal     keys through the radix / sample sort, payload through vnm_take
//   utf8 / large_utf8 / binary / large_binary
//                          KEY: the values' order-preserving ranks (string dictionary on the device + a sort of the distinct
//                          values, vnm_strdict_ranks_device) as an int32 key column -- NULL stays NULL (last in both directions),
//                          equal values share a rank, so the stable sort keeps them in row order like SortIndices;
//                          PAYLOAD: gathered on the device (lengths -> prefix sums -> bytes, vnm_take_varwidth)
//   decimal128             KEY: (high int64, low uint64) as two keys; PAYLOAD: 16-byte gather
//   boolean                PAYLOAD: bit gather; as a KEY it raises like the reference (vinum/core/algebra.py:191-201)
int vnm_sort_op_sorted(vnm_sort_op* h, int64_t limit, struct ArrowArray* out, struct ArrowSchema* out_schema) {
    if (!h || !out || !out_schema) return set_error("vnm_sort_op_sorted: null argument");
    if (h->batches.empty()) return set_error("Failed to create table from record batches.");
    const struct ArrowSchema* sch = &h->batches[0]->sch;
    const int64_t ncols = sch->n_children;
    int64_t total = 0;
    for (auto& b : h->batches) {
        if (b->sch.n_children != ncols) return set_error("Failed to create table from record batches.");
        for (int64_t c = 0; c < ncols; c++)
            if (strcmp(b->sch.children[c]->format, sch->children[c]->format) != 0) return set_error("Failed to create table from record batches.");
        total += b->arr.length;
    }
    std::vector<SortColInfo> info((size_t)ncols);
    for (int64_t c = 0; c < ncols; c++) {
        info[c] = classify_sort_col(sch->children[c]->format);
        if (info[c].kind < 0) return set_error("Sort: column '%s' has a type the GPU path does not handle (format %s)",
                                               sch->children[c]->name, sch->children[c]->format);
    }
    std::vector<int> key_col;
    for (auto& name : h->cols) {
        int i = find_child(sch, name);
        if (i < 0) return set_error("Failed to sort table.");
        if (info[i].kind == SC_BOOL) return set_error("Failed to sort table.");   // (Arrow 3.0 has no boolean sort; algebra.py:191-201 rejects it first)
        key_col.push_back(i);
    }
    // Table::FromRecordBatches: every column of all batches as ONE column in HBM
    std::vector<DevSortCol> dev((size_t)ncols);
    PoolScope scratch;              // key helper buffers (ranks, decimal words)
    struct DictGuard { std::vector<vnm_strdict*> d; ~DictGuard() { for (auto* x : d) vnm_strdict_destroy(x); } } dicts;
    int rc = 0;
    for (int64_t c = 0; c < ncols && !rc; c++) rc = stage_sort_col(h->batches, (int)c, info[c], total, &dev[c]);
    const int64_t n_out = (limit > 0 && limit < total) ? limit : total;
    int64_t* idx = nullptr;
    if (!rc && total > 0) {
        std::vector<vnm_dcol> keys;
        std::vector<int> orders;
        for (size_t k = 0; k < key_col.size() && !rc; k++) {
            const int c = key_col[k];
            DevSortCol& d = dev[c];
            vnm_dcol kc{};
            kc.length = total; kc.validity = d.validity;
            if (info[c].kind == SC_NUM) { keys.push_back(d.num); orders.push_back(h->orders[k]); continue; }
            if (info[c].kind == SC_VAR) {
                vnm_strdict* sd = vnm_strdict_create();
                if (!sd) { rc = 1; break; }
                dicts.d.push_back(sd);
                int32_t* codes = (int32_t*)scratch.take((size_t)total * 4);
                int32_t* ranks = (int32_t*)scratch.take((size_t)total * 4);
                if (!codes || !ranks) { rc = 1; break; }
                vnm_dcol oc{};
                oc.values = d.offs; oc.type = VNM_I64; oc.length = total + 1;
                rc = vnm_strdict_encode_device(sd, &oc, d.validity, 0, d.data, 0, codes, nullptr, nullptr, nullptr);
                int32_t* rank_of = nullptr;
                if (!rc) { rank_of = (int32_t*)scratch.take((size_t)std::max<int64_t>(vnm_strdict_ids(sd), 1) * 4); if (!rank_of) rc = 1; }
                if (!rc) rc = vnm_strdict_ranks_device(sd, rank_of, nullptr);
                if (!rc) rc = vnm_strdict_codes_to_ranks(codes, rank_of, total, ranks, nullptr);
                kc.values = ranks; kc.type = VNM_I32;
                keys.push_back(kc); orders.push_back(h->orders[k]);
            } else {   // SC_DEC
                int64_t* hi = (int64_t*)scratch.take((size_t)total * 8);
                uint64_t* lo = (uint64_t*)scratch.take((size_t)total * 8);
                if (!hi || !lo) { rc = 1; break; }
                rc = vnm_decimal128_sort_keys(d.fixed, total, hi, lo, nullptr);
                kc.values = hi; kc.type = VNM_I64;
                keys.push_back(kc); orders.push_back(h->orders[k]);
                kc.values = lo; kc.type = VNM_U64;
                keys.push_back(kc); orders.push_back(h->orders[k]);
            }
        }
        if (!rc && keys.size() > 16) rc = set_error("Sort: more than 16 sort key words");
        if (!rc) { idx = (int64_t*)pool_alloc((size_t)total * 8); if (!idx) rc = 1; }
        if (!rc) rc = vnm_sort_indices((int)keys.size(), keys.data(), orders.data(), total, n_out < total ? n_out : 0, idx, nullptr);
    }
    if (!rc) {
        make_struct(out, n_out, ncols);
        make_schema(out_schema, "+s", "", ncols);
        for (int64_t c = 0; c < ncols && !rc; c++) {
            rc = take_sort_col(dev[c], info[c], idx, n_out, out->children[c]);
            if (!rc) make_schema(out_schema->children[c], info[c].format, sch->children[c]->name ? sch->children[c]->name : "", 0);
        }
        if (rc) { release_array(out); release_schema(out_schema); }
    }
    pool_free(idx);
    for (auto& d : dev) d.free_all();
    for (auto& b : h->batches) b->drop();
    h->batches.clear();
    return rc;
}

}  // extern "C"
