// CSV ingest for gfx950: tokenise one block of CSV text and parse its numeric columns on the device.
//
// Replaces, for numeric columns, what stream_csv() / read_csv() delegate to pyarrow.csv (vinum/io/arrow.py:58-61,106 and
// FileReaderOperator, vinum/core/algebra.py:268-279): the block's bytes cross PCIe ONCE as text (pinned double buffer) and
// the Arrow columns are born in HBM -- instead of being parsed on one CPU core, materialised as Arrow arrays in host memory
// and staged column by column.  Parity target: pyarrow.csv's defaults (third-party; Arrow's converters are fast_float /
// from_chars: correctly rounded) -- comma delimiter, first line = header, empty field = NULL, int64 / float64 columns.
//
//   csv_count_kernel    newlines per 16 KB slab
//   csv_scan_kernel     exclusive scan of the slab counts (one workgroup)
//   csv_offsets_kernel  row start offsets (ballot ranks inside the slab)
//   csv_parse_kernel    one lane per row: walk the fields (quoted fields: delimiters inside quotes do not split, round 4), parse
//                       the selected ones
//
// Decimal -> float64 is EXACT integer arithmetic, not floating point: a field is (sign, w, q) with w < 2^64 the significant
// digits (at most 19) and q the decimal exponent; q >= 0: the 128-bit product w * 10^q rounded to 53 bits (half to even);
// q < 0: the 128-by-64-bit quotient (w << s) / 10^-q with its remainder as the sticky bit, rounded the same way.  Both are
// the correctly rounded value of the decimal string -- what strtod / fast_float / Python's float() return.  Fields outside
// that domain (more than 19 significant digits, |q| > 19, "inf", "nan", hex floats, thousands separators, whitespace) are
// not guessed at: the lane raises the column's fallback flag and the caller has pyarrow parse that column of that block.
//
// Round 5: string, date and timestamp columns stay on the device too (vnm_csv_parse_block_ex).  A utf8 field is a SPAN of the staged
// text (quotes stripped, UTF-8 checked as pyarrow's check_utf8 does); the spans of a column go through the column's string dictionary
// where they lie (vnm_strdict_encode_spans) and the column is born as int32 dictionary codes -- what the host route
// (pyarrow -> dictionary_encode -> stage) produced, without the Arrow string array in between.  "YYYY-MM-DD" and
// "YYYY-MM-DD[ T]hh:mm[:ss[.fffffffff]]" become date32 / timestamp[s] / timestamp[ns] by integer arithmetic (days_from_civil); other
// ISO 8601 spellings (zone offsets, week dates, hour-only times) raise the fallback flag like the numeric corner cases.
#include <algorithm>

#include "vnm_common.hpp"

namespace vnm {

constexpr int CSV_SLAB = 16384;      // bytes per workgroup in the newline passes
constexpr int CSV_MAX_COLS = 16;

struct CsvArgs {
    const uint8_t* text;
    int64_t nbytes;
    int64_t first;                    // offset of the first data byte (after the header line, 0 without header)
    uint32_t* slab_counts;            // newlines per slab, then their exclusive prefix
    int64_t nslabs;
    int64_t* row_start;               // [nrows + 1]
    unsigned long long* flags;        // [0] a row ends inside a quoted field  [1 + c] column c needs the host parser  [20] malformed row (field count)  [21] a quote was seen
    uint8_t delim;
    int n_fields;                     // fields per row (from the header)
    int n_cols;
    int field_of[CSV_MAX_COLS];       // ascending field indices of the parsed columns
    int type_of[CSV_MAX_COLS];        // VNM_I64 / VNM_F64 / VNM_CSV_*
    void* out_values[CSV_MAX_COLS];   // 8 bytes per row (VNM_CSV_STRING: the span's start in the text; VNM_CSV_DATE32: 4 bytes per row)
    int32_t* span_len[CSV_MAX_COLS];  // VNM_CSV_STRING: the span's length
    uint8_t* out_valid[CSV_MAX_COLS]; // byte per row (packed into bitmaps afterwards)
    int64_t nrows;
};

__host__ __device__ inline bool csv_is32(int type) { return type == VNM_CSV_DATE32 || type == VNM_CSV_BOOL || type == VNM_CSV_TIME32_S; }   // four bytes per row

__global__ __launch_bounds__(256) void csv_count_kernel(CsvArgs a) {
    const int64_t base = (int64_t)blockIdx.x * CSV_SLAB;   // slabs are aligned in the buffer (16-byte loads); bytes before `first` are skipped
    uint32_t c = 0;
    bool quote = false;
    for (int i = threadIdx.x * 16; i < CSV_SLAB; i += 256 * 16) {
        const int64_t p = base + i;
        if (p >= a.first && p + 16 <= a.nbytes) {
            const uint4 v = *(const uint4*)(a.text + p);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const uint32_t ch = (w[k] >> (8 * b)) & 255u;
                    c += ch == '\n';
                    quote |= ch == '"';
                }
        } else {
            for (int k = 0; k < 16 && p + k < a.nbytes; k++) {
                if (p + k < a.first) continue;
                const uint8_t ch = a.text[p + k];
                c += ch == '\n';
                quote |= ch == '"';
            }
        }
    }
    for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
    __shared__ uint32_t sc[4];
    if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = c;
    if (__ballot(quote) && (threadIdx.x & 63) == 0) a.flags[21] = 1;   // (statistics only: quoted fields are handled by csv_parse_kernel)
    __syncthreads();
    if (threadIdx.x == 0) a.slab_counts[blockIdx.x] = sc[0] + sc[1] + sc[2] + sc[3];
}

// in place: counts -> exclusive prefix; total to slab_counts[nslabs]
__global__ __launch_bounds__(1024) void csv_scan_kernel(uint32_t* counts, int64_t n) {
    __shared__ uint32_t part[1024];
    const int tid = threadIdx.x;
    const int64_t per = (n + 1023) / 1024;
    const int64_t lo = tid * per, hi = lo + per < n ? lo + per : n;
    uint32_t s = 0;
    for (int64_t i = lo; i < hi; i++) s += counts[i];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int k = 0; k < 1024; k++) { const uint32_t t = part[k]; part[k] = run; run += t; }
        counts[n] = run;
    }
    __syncthreads();
    uint32_t run = part[tid];
    for (int64_t i = lo; i < hi; i++) { const uint32_t t = counts[i]; counts[i] = run; run += t; }
}

// row r starts right after the r-th newline of the data (row 0 at a.first); one lane per byte of the slab, 64 bytes per
// wave round, ranks from ballots
__global__ __launch_bounds__(256) void csv_offsets_kernel(CsvArgs a) {
    const int64_t base = (int64_t)blockIdx.x * CSV_SLAB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ uint32_t wcount[4];
    uint32_t seen = a.slab_counts[blockIdx.x];   // newlines before this slab
    if (blockIdx.x == 0 && threadIdx.x == 0) a.row_start[0] = a.first;
    for (int i0 = 0; i0 < CSV_SLAB; i0 += 256) {
        const int64_t p = base + i0 + threadIdx.x;
        const bool nl = p >= a.first && p < a.nbytes && a.text[p] == '\n';
        const unsigned long long m = __ballot(nl);
        if (lane == 0) wcount[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0;
        for (int w = 0; w < wave; w++) before += wcount[w];
        const uint32_t total = wcount[0] + wcount[1] + wcount[2] + wcount[3];
        if (nl) {
            const unsigned long long lt = lane == 0 ? 0ULL : (~0ULL >> (64 - lane));
            const uint32_t r = seen + before + (uint32_t)__popcll(m & lt);
            a.row_start[(int64_t)r + 1] = p + 1;
        }
        seen += total;
        __syncthreads();
    }
}

// ---- exact decimal -> binary64 ------------------------------------------------------------------------------------------
__device__ __forceinline__ int clz64(uint64_t x) { return __clzll((long long)x); }

// (hi:lo) as an integer, plus a sticky flag for discarded lower-order information, scaled by 2^e2 -> nearest double, ties to even
__device__ __forceinline__ double round_u128(uint64_t hi, uint64_t lo, bool sticky, int e2) {
    if (hi == 0 && lo == 0) return 0.0;
    int msb;   // index of the top set bit of (hi:lo)
    if (hi) msb = 127 - clz64(hi); else msb = 63 - clz64(lo);
    uint64_t mant;  // top 53 bits
    bool half = false, rest = sticky;
    if (msb <= 52) {
        mant = lo;                      // exact
        return ldexp((double)mant, e2);
    }
    const int sh = msb - 52;            // bits to drop
    if (sh < 64) {
        mant = (sh == 0) ? lo : ((lo >> sh) | (hi << (64 - sh)));
        if (hi >> sh) {}                // cannot happen: msb bounds mant to 53 bits
        const uint64_t dropped = lo & ((1ULL << sh) - 1ULL);
        half = (dropped >> (sh - 1)) & 1ULL;
        rest = rest || (dropped & ((1ULL << (sh - 1)) - 1ULL)) != 0;
        mant &= (1ULL << 53) - 1ULL;
    } else {
        const int s2 = sh - 64;
        mant = s2 == 0 ? hi : (hi >> s2);
        mant &= (1ULL << 53) - 1ULL;
        if (s2 == 0) { half = lo >> 63; rest = rest || (lo << 1) != 0; }
        else {
            const uint64_t dropped = hi & ((1ULL << s2) - 1ULL);
            half = (dropped >> (s2 - 1)) & 1ULL;
            rest = rest || (dropped & ((1ULL << (s2 - 1)) - 1ULL)) != 0 || lo != 0;
        }
    }
    if (half && (rest || (mant & 1ULL))) mant++;      // may carry to 2^53: still exactly representable
    return ldexp((double)mant, e2 + sh);
}

// quotient of (hi:lo) / d for hi < d (so the quotient fits 64 bits), remainder in *rem
__device__ __forceinline__ uint64_t udiv128_64(uint64_t hi, uint64_t lo, uint64_t d, uint64_t* rem) {
    uint64_t r = hi, q = 0;
#pragma unroll 4
    for (int i = 63; i >= 0; i--) {
        const bool carry = r >> 63;
        r = (r << 1) | ((lo >> i) & 1ULL);
        q <<= 1;
        if (carry || r >= d) { r -= d; q |= 1ULL; }
    }
    *rem = r;
    return q;
}

__device__ __constant__ uint64_t CSV_POW10[20] = {
    1ULL, 10ULL, 100ULL, 1000ULL, 10000ULL, 100000ULL, 1000000ULL, 10000000ULL, 100000000ULL, 1000000000ULL, 10000000000ULL,
    100000000000ULL, 1000000000000ULL, 10000000000000ULL, 100000000000000ULL, 1000000000000000ULL, 10000000000000000ULL,
    100000000000000000ULL, 1000000000000000000ULL, 10000000000000000000ULL};

// w * 10^q, correctly rounded; false when outside the exact domain (|q| > 19)
__device__ __forceinline__ bool decimal_to_double(uint64_t w, int q, double* out) {
    if (w == 0) { *out = 0.0; return true; }
    if (q >= 0) {
        if (q > 19) return false;
        const uint64_t p = CSV_POW10[q];
        const uint64_t lo = w * p, hi = __umul64hi(w, p);
        *out = round_u128(hi, lo, false, 0);
        return true;
    }
    const int k = -q;
    if (k > 19) return false;
    const uint64_t d = CSV_POW10[k];
    // numerator w << s with s chosen so that the quotient has 63 or 64 significant bits and hi < d
    const int lw = 64 - clz64(w), ld = 64 - clz64(d);
    const int s = 63 + ld - lw;                       // 0 < s <= 126
    uint64_t hi, lo;
    if (s >= 64) { hi = w << (s - 64); lo = 0; }
    else { hi = s == 0 ? 0 : (w >> (64 - s)); lo = w << s; }
    if (hi >= d) return false;                        // cannot happen by construction; never divide wrongly
    uint64_t rem;
    const uint64_t quo = udiv128_64(hi, lo, d, &rem);
    *out = round_u128(0, quo, rem != 0, -s);
    return true;
}

// one numeric field [p, p + len): 0 = ok, 1 = NULL (empty), 2 = needs the host parser.
// A single pass with an explicit state (0 mantissa, 1 right after e / E, 2 exponent digits): [sign] digits [. digits] [e|E [sign] digits]
__device__ __noinline__ int csv_parse_field(const uint8_t* p, int len, int type, uint64_t* bits) {
    if (len == 0) return 1;
    bool neg = false, eneg = false, seen_dot = false;
    uint64_t w = 0;
    int digits = 0, q = 0, any = 0, state = 0, ex = 0, exdigits = 0;
    const bool is_f = type == VNM_F64;
    for (int i = 0; i < len; i++) {
        const uint32_t ch = p[i];
        const bool dig = ch >= 48u && ch <= 57u;
        if (state == 0) {
            if (dig) {
                any = 1;
                if (digits == 0 && ch == 48u) { if (seen_dot) q--; }          // leading zeros carry no digits
                else {
                    if (digits >= 19) return 2;                                // beyond 64-bit significands: host parser
                    w = w * 10 + (ch - 48u);
                    digits++;
                    if (seen_dot) q--;
                }
            } else if (i == 0 && (ch == 45u || ch == 43u)) {                   // '-' '+'
                neg = ch == 45u;
            } else if (ch == 46u && !seen_dot && is_f) {                       // '.'
                seen_dot = true;
            } else if ((ch == 101u || ch == 69u) && is_f && any) {             // 'e' 'E'
                state = 1;
            } else return 2;
        } else if (state == 1) {
            if (ch == 45u || ch == 43u) { eneg = ch == 45u; state = 2; }
            else if (dig) { ex = (int)(ch - 48u); exdigits = 1; state = 2; }
            else return 2;
        } else {
            if (!dig || ex > 9999) return 2;
            ex = ex * 10 + (int)(ch - 48u);
            exdigits++;
        }
    }
    if (!any || (state != 0 && exdigits == 0)) return 2;
    q += eneg ? -ex : ex;
    if (!is_f) {
        if (neg ? w > 0x8000000000000000ULL : w > 0x7FFFFFFFFFFFFFFFULL) return 2;
        *bits = neg ? (uint64_t)0 - w : w;
        return 0;
    }
    double d;
    if (!decimal_to_double(w, q, &d)) return 2;
    *bits = (uint64_t)__double_as_longlong(neg ? -d : d);
    return 0;
}


// days since 1970-01-01 of a proleptic Gregorian date (the civil-calendar identity: eras of 400 years = 146097 days)
__device__ __forceinline__ int64_t csv_days_from_civil(int y, int m, int d) {
    y -= m <= 2;
    const int era = (y >= 0 ? y : y - 399) / 400;
    const int yoe = y - era * 400;
    const int doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1;
    const int doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
    return (int64_t)era * 146097 + doe - 719468;
}
__device__ __forceinline__ bool csv_two(const uint8_t* p, int* v) {
    const uint32_t a = p[0] - 48u, b = p[1] - 48u;
    *v = (int)(a * 10 + b);
    return a <= 9u && b <= 9u;
}

// one date / timestamp field: 0 = ok, 1 = NULL (empty), 2 = needs the host parser.  Accepted: YYYY-MM-DD, and for timestamps
// YYYY-MM-DD[ T]hh:mm, ...:ss, ...:ss.f{1,9} (nanoseconds only).  Anything else Arrow's ISO 8601 parser may or may not take: host.
__device__ __noinline__ int csv_parse_time_field(const uint8_t* p, int len, int type, uint64_t* bits) {
    if (len == 0) return 1;
    if (len < 10) return 2;
    int c1, c2, mo, dd;
    if (!csv_two(p, &c1) || !csv_two(p + 2, &c2) || p[4] != '-' || !csv_two(p + 5, &mo) || p[7] != '-' || !csv_two(p + 8, &dd)) return 2;
    const int y = c1 * 100 + c2;
    if (mo < 1 || mo > 12 || dd < 1) return 2;
    const bool leap = (y % 4 == 0 && y % 100 != 0) || y % 400 == 0;
    const int mdays = mo == 2 ? (leap ? 29 : 28) : ((mo == 4 || mo == 6 || mo == 9 || mo == 11) ? 30 : 31);
    if (dd > mdays) return 2;
    const int64_t days = csv_days_from_civil(y, mo, dd);
    if (type == VNM_CSV_DATE32) {
        if (len != 10) return 2;
        *bits = (uint64_t)days;
        return 0;
    }
    int hh = 0, mi = 0, ss = 0;
    int64_t frac = 0;
    if (len > 10) {
        if (len < 16 || (p[10] != ' ' && p[10] != 'T') || !csv_two(p + 11, &hh) || p[13] != ':' || !csv_two(p + 14, &mi)) return 2;
        if (len > 16) {
            if (len < 19 || p[16] != ':' || !csv_two(p + 17, &ss)) return 2;
            if (len > 19) {
                if (type != VNM_CSV_TIMESTAMP_NS || p[19] != '.' || len == 20 || len > 29) return 2;
                int64_t scale = 1000000000;
                for (int i = 20; i < len; i++) {
                    const uint32_t dg = p[i] - 48u;
                    if (dg > 9u) return 2;
                    scale /= 10;
                    frac += (int64_t)dg * scale;
                }
            }
        }
        if (hh > 23 || mi > 59 || ss > 59) return 2;
    }
    const int64_t secs = days * 86400 + hh * 3600 + mi * 60 + ss;
    if (type == VNM_CSV_TIMESTAMP_S) { *bits = (uint64_t)secs; return 0; }
    if (y < 1678 || y > 2261) return 2;          // int64 nanoseconds cover 1677-09-21 .. 2262-04-11: the edges go to the host parser
    *bits = (uint64_t)(secs * 1000000000LL + frac);
    return 0;
}

// a boolean field, pyarrow's spellings (ConvertOptions.true_values / false_values defaults) and nothing else; a time of day hh:mm[:ss]
__device__ __noinline__ int csv_parse_small_field(const uint8_t* p, int len, int type, uint64_t* bits) {
    if (len == 0) return 1;
    if (type == VNM_CSV_TIME32_S) {
        int hh, mi, ss = 0;
        if ((len != 5 && len != 8) || !csv_two(p, &hh) || p[2] != ':' || !csv_two(p + 3, &mi)) return 2;
        if (len == 8 && (p[5] != ':' || !csv_two(p + 6, &ss))) return 2;
        if (hh > 23 || mi > 59 || ss > 59) return 2;
        *bits = (uint64_t)(hh * 3600 + mi * 60 + ss);
        return 0;
    }
    auto is = [&](const char* w, int n) { if (len != n) return false; for (int i = 0; i < n; i++) if (p[i] != (uint8_t)w[i]) return false; return true; };
    if (is("1", 1) || is("true", 4) || is("True", 4) || is("TRUE", 4)) { *bits = 1; return 0; }
    if (is("0", 1) || is("false", 5) || is("False", 5) || is("FALSE", 5)) { *bits = 0; return 0; }
    return 2;
}

// well-formed UTF-8 (shortest forms, no surrogates, <= U+10FFFF): what pyarrow's check_utf8 demands of a string column
__device__ __noinline__ bool csv_utf8_ok(const uint8_t* p, int len) {
    int i = 0;
    while (i < len) {
        const uint32_t b = p[i];
        if (b < 0x80u) { i++; continue; }
        int need;
        uint32_t lo = 0x80u, hi = 0xBFu;
        if (b >= 0xC2u && b <= 0xDFu) need = 1;
        else if (b == 0xE0u) { need = 2; lo = 0xA0u; }
        else if (b == 0xEDu) { need = 2; hi = 0x9Fu; }
        else if (b >= 0xE1u && b <= 0xEFu) need = 2;
        else if (b == 0xF0u) { need = 3; lo = 0x90u; }
        else if (b >= 0xF1u && b <= 0xF3u) need = 3;
        else if (b == 0xF4u) { need = 3; hi = 0x8Fu; }
        else return false;
        if (i + need >= len) return false;      // the sequence is cut off by the end of the field
        const uint32_t c1 = p[i + 1];
        if (c1 < lo || c1 > hi) return false;
        for (int k = 2; k <= need; k++) { const uint32_t c = p[i + k]; if (c < 0x80u || c > 0xBFu) return false; }
        i += need + 1;
    }
    return true;
}

__global__ __launch_bounds__(256) void csv_parse_kernel(CsvArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.nrows; r += stride) {
        const int64_t lo = a.row_start[r];
        int64_t hi = a.row_start[r + 1] - 1;                  // the '\n'
        if (hi > lo && a.text[hi - 1] == '\r') hi--;
        const uint8_t* p = a.text + lo;
        const int len = (int)(hi - lo);
        int field = 0, fstart = 0, c = 0;
        // QUOTED fields (round 4; pyarrow.csv defaults: quote_char '"', double_quote, newlines_in_values = False): a field that STARTS
        // with a quote runs to its closing quote -- delimiters inside it do not split the row, "" is a literal quote -- so rows with
        // quoted string columns ("New York, NY") keep their numeric columns on the device.  A selected numeric field that is quoted
        // is parsed from between its quotes; one with an escaped quote, or with text behind the closing quote, goes to the host
        // parser; a row that ends inside a quote (a newline in a value) sends the block there.
        bool inq = false, quoted = false, weird = false;   // inside quotes; this field began with a quote; escapes / trailing text
        int qend = -1;                                     // index of the closing quote of this field
        for (int i = 0; i <= len; i++) {
            const uint32_t ch = i < len ? p[i] : 0u;
            if (i < len && inq) {
                if (ch == '"') {
                    if (i + 1 < len && p[i + 1] == '"') { weird = true; i++; }      // "" inside quotes
                    else { inq = false; qend = i; }
                }
                continue;
            }
            if (i < len && i == fstart && ch == '"') { inq = true; quoted = true; continue; }
            if (i == len || ch == a.delim) {
                if (c < a.n_cols && field == a.field_of[c]) {
                    uint64_t bits = 0;
                    int rc;
                    const int type = a.type_of[c];
                    const uint8_t* f = quoted ? p + fstart + 1 : p + fstart;             // the field between its quotes
                    const int flen = quoted ? qend - fstart - 1 : i - fstart;
                    if (quoted && (weird || qend != i - 1)) rc = 2;                      // an escaped quote, or text behind the closing quote
                    else if (type == VNM_CSV_STRING) {
                        // a string field is never NULL (pyarrow: strings_can_be_null = False; an empty field is the empty string)
                        rc = csv_utf8_ok(f, flen) ? 0 : 2;
                        bits = (uint64_t)(lo + (f - p));
                        a.span_len[c][r] = rc == 0 ? flen : 0;
                    } else if (type == VNM_I64 || type == VNM_F64) rc = csv_parse_field(f, flen, type, &bits);   // (quoted "" -> NULL, as pyarrow's quoted_strings_can_be_null)
                    else if (type == VNM_CSV_BOOL || type == VNM_CSV_TIME32_S) rc = csv_parse_small_field(f, flen, type, &bits);
                    else rc = csv_parse_time_field(f, flen, type, &bits);
                    if (rc == 2) a.flags[1 + c] = 1;
                    if (csv_is32(type)) ((int32_t*)a.out_values[c])[r] = (int32_t)bits;
                    else ((uint64_t*)a.out_values[c])[r] = bits;
                    a.out_valid[c][r] = rc == 0;
                    c++;
                }
                field++;
                fstart = i + 1;
                quoted = false; weird = false; qend = -1;
            }
        }
        if (inq) a.flags[0] = 1;      // the row ended inside a quoted field
        // ragged row: pyarrow raises on it, so does the caller.  An EMPTY line is not a row at all for pyarrow (ignore_empty_lines):
        // it is flagged the same way even in a one-column file, so that the block takes pyarrow's row count.
        if (field != a.n_fields || len == 0) a.flags[20] = 1;
        for (; c < a.n_cols; c++) {
            if (csv_is32(a.type_of[c])) ((int32_t*)a.out_values[c])[r] = 0; else ((uint64_t*)a.out_values[c])[r] = 0;
            if (a.span_len[c]) a.span_len[c][r] = 0;
            a.out_valid[c][r] = 0;
        }
    }
}

}  // namespace vnm

using namespace vnm;

extern "C" {

int vnm_pack_validity(const uint8_t* valid_bytes, int64_t n, uint8_t* bitmap, void* stream);

int vnm_csv_parse_block(const char* host_text, int64_t nbytes, int skip_header, int delimiter, int n_fields, int n_cols,
                        const int* field_idx, const int* types, vnm_dcol* out_cols, int64_t* n_rows, int* fallback /* [n_cols + 2] */,
                        void* stream) {
    return vnm_csv_parse_block_ex(host_text, nbytes, skip_header, delimiter, n_fields, n_cols, field_idx, types, nullptr, out_cols, n_rows, fallback, stream);
}

int vnm_csv_parse_block_ex(const char* host_text, int64_t nbytes, int skip_header, int delimiter, int n_fields, int n_cols,
                           const int* field_idx, const int* types, vnm_strdict* const* dicts, vnm_dcol* out_cols, int64_t* n_rows,
                           int* fallback /* [n_cols + 2] */, void* stream) {
    VNM_TRY(ensure_init());
    if (!host_text || !out_cols || !n_rows || !fallback) return set_error("vnm_csv_parse_block: null argument");
    if (n_cols < 1 || n_cols > CSV_MAX_COLS) return set_error("vnm_csv_parse_block: 1..%d columns per call", CSV_MAX_COLS);
    if (nbytes <= 0 || host_text[nbytes - 1] != '\n') return set_error("vnm_csv_parse_block: a block must end with a newline");
    if (nbytes >= (1LL << 31)) return set_error("vnm_csv_parse_block: blocks must be < 2 GiB");
    for (int c = 0; c < n_cols; c++) {
        const int t = types[c];
        if (t != VNM_I64 && t != VNM_F64 && (t < VNM_CSV_STRING || t > VNM_CSV_TIME32_S))
            return set_error("vnm_csv_parse_block: column %d: int64 / float64 / VNM_CSV_* only", c);
        if (t == VNM_CSV_STRING && (!dicts || !dicts[c])) return set_error("vnm_csv_parse_block: column %d: a string column needs its dictionary", c);
        if (c && field_idx[c] <= field_idx[c - 1]) return set_error("vnm_csv_parse_block: field indices must ascend");
        if (field_idx[c] < 0 || field_idx[c] >= n_fields) return set_error("vnm_csv_parse_block: field index out of range");
    }
    hipStream_t s = as_stream(stream);
    int64_t first = 0;
    if (skip_header) {
        const char* nl = (const char*)memchr(host_text, '\n', (size_t)nbytes);
        first = nl ? (nl - host_text) + 1 : nbytes;
    }
    for (int c = 0; c < n_cols + 2; c++) fallback[c] = 0;
    for (int c = 0; c < n_cols; c++) memset(&out_cols[c], 0, sizeof(vnm_dcol));
    *n_rows = 0;
    if (first >= nbytes) return 0;
    vnm_dcol text{};
    VNM_TRY(vnm_stage_column(host_text, nullptr, 0, nbytes, VNM_U8, &text, stream));   // the text crosses PCIe once
    // every way out frees the staged text, the scratch blocks and -- unless the block is handed over -- the output columns
    struct Cleanup {
        vnm_dcol* text; vnm_dcol* out; int n; bool keep_out = false;
        ~Cleanup() {
            vnm_free_column(text);
            if (!keep_out) for (int c = 0; c < n; c++) { vnm_free_column(&out[c]); memset(&out[c], 0, sizeof(vnm_dcol)); }
        }
    } cleanup{&text, out_cols, n_cols};
    PoolScope pool;
    // (declared last, so it runs first on the way out: an error return must not hand scratch blocks, the staged text or the output
    // columns back to the pool while kernels that use them may still be in flight -- ADVICE r03)
    struct SyncOnError { hipStream_t s; bool ok = false; ~SyncOnError() { if (!ok) (void)hipStreamSynchronize(s); } } sync_guard{s};
    CsvArgs a{};
    a.text = (const uint8_t*)text.values;
    a.nbytes = nbytes;
    a.first = first;
    a.nslabs = (nbytes + CSV_SLAB - 1) / CSV_SLAB;
    a.delim = (uint8_t)delimiter;
    a.n_fields = n_fields;
    a.n_cols = n_cols;
    for (int c = 0; c < n_cols; c++) { a.field_of[c] = field_idx[c]; a.type_of[c] = types[c]; }
    a.slab_counts = (uint32_t*)pool.take((size_t)(a.nslabs + 1) * 4);
    a.flags = (unsigned long long*)pool.take(256);
    if (!a.slab_counts || !a.flags) return 1;
    VNM_HIP(hipMemsetAsync(a.flags, 0, 256, s));
    {
        KernelTimer timer("csv_tokenize", s);
        csv_count_kernel<<<(int)a.nslabs, 256, 0, s>>>(a);
        csv_scan_kernel<<<1, 1024, 0, s>>>(a.slab_counts, a.nslabs);
    }
    uint32_t total = 0;
    VNM_HIP(hipMemcpyAsync(&total, a.slab_counts + a.nslabs, 4, hipMemcpyDeviceToHost, s));
    VNM_HIP(hipStreamSynchronize(s));
    a.nrows = (int64_t)total;
    *n_rows = a.nrows;
    if (a.nrows > 0) {
        a.row_start = (int64_t*)pool.take((size_t)(a.nrows + 1) * 8);
        if (!a.row_start) return 1;
        for (int c = 0; c < n_cols; c++) {
            if (types[c] == VNM_CSV_STRING) {       // spans now (scratch), codes behind the parse kernel
                a.out_values[c] = pool.take((size_t)a.nrows * 8);
                a.span_len[c] = (int32_t*)pool.take((size_t)a.nrows * 4);
                out_cols[c].values = pool_alloc((size_t)a.nrows * 4);
                if (!a.span_len[c] || !out_cols[c].values) return 1;
            } else
                out_cols[c].values = a.out_values[c] = pool_alloc((size_t)a.nrows * (csv_is32(types[c]) ? 4 : 8));      // (owned by out_cols from here: Cleanup)
            a.out_valid[c] = (uint8_t*)pool.take((size_t)a.nrows);
            if (!a.out_values[c] || !a.out_valid[c]) return 1;
        }
        {
            KernelTimer timer("csv_parse", s);
            csv_offsets_kernel<<<(int)a.nslabs, 256, 0, s>>>(a);
            const int grid = (int)std::min<int64_t>((a.nrows + 255) / 256, (int64_t)device_info().num_cus * 16);
            csv_parse_kernel<<<grid, 256, 0, s>>>(a);
        }
        if (hipGetLastError() != hipSuccess) return set_error("vnm_csv_parse_block: kernel launch failed");
        // Arrow validity bitmaps
        for (int c = 0; c < n_cols; c++) {
            out_cols[c].length = a.nrows;
            out_cols[c].type = types[c] == VNM_CSV_STRING || csv_is32(types[c]) ? VNM_I32 : (types[c] == VNM_F64 ? VNM_F64 : VNM_I64);
            if (types[c] == VNM_CSV_STRING) continue;       // never NULL
            uint8_t* bm = (uint8_t*)pool_alloc((size_t)((a.nrows + 63) / 64) * 8);
            if (!bm) return 1;
            out_cols[c].validity = bm;
            VNM_TRY(vnm_pack_validity(a.out_valid[c], a.nrows, bm, stream));
        }
    }
    unsigned long long fl[21] = {};
    VNM_HIP(hipMemcpyAsync(fl, a.flags, sizeof(fl), hipMemcpyDeviceToHost, s));
    VNM_HIP(hipStreamSynchronize(s));
    // string columns: the spans -> dictionary codes, while the text is still staged.  A column (or block) that goes to the host parser
    // anyway is not encoded: its values would enter the dictionary twice over otherwise harmlessly, but an invalid UTF-8 field must not
    if (a.nrows > 0 && !fl[0] && !fl[20])
        for (int c = 0; c < n_cols; c++)
            if (types[c] == VNM_CSV_STRING && !fl[1 + c])
                VNM_TRY(vnm_strdict_encode_spans(dicts[c], (const int64_t*)a.out_values[c], a.span_len[c], a.nrows, a.text, (int32_t*)out_cols[c].values,
                                                 nullptr, nullptr, stream));
    for (int c = 0; c < n_cols; c++) fallback[c] = fl[1 + c] != 0;
    fallback[n_cols] = fl[0] != 0;        // a row ends inside a quoted field (a newline in a value): the whole block needs the host reader
    fallback[n_cols + 1] = fl[20] != 0;   // a row with a different number of fields
    cleanup.keep_out = true;
    sync_guard.ok = true;     // (the flags read-back above synchronised the stream)
    return 0;
}

}  // extern "C"
