// K1: libsvm text -> (ids i32 [n,F], vals f32 [n,F], labels f32 [n]).  Host code, re-entrant.
// Replaces decode_libsvm (DeepFM.py:65-81; identical in DCN.py:68-84, PNN.py:67-83, NFM.py:65-76, AFM.py:64-80):
//   string_split([line], ' ')          -> tokens, empty tokens dropped (skip_empty=True) [TF-1.4]
//   string_to_number(token0, float32)  -> label
//   string_split(tokens[1:], ':')      -> [F,2] strings
//   string_to_number(col0, int32) / string_to_number(col1, float32)
// string_to_number(float32) is a correctly rounded decimal->binary32 conversion.  Fast path: a plain decimal
// with mantissa n < 2^29 and k <= 22 fractional digits converts as (float)((double)n / 10^k), which is
// correctly rounded (the double quotient's 2^-53 relative error cannot cross a binary32 rounding boundary
// for n < 2^29 -- see DESIGN.md "parser"); everything else goes through glibc strtof.
#include <algorithm>
#include <cerrno>
#include <cstdlib>
#include <functional>
#include <thread>
#include <vector>

#include "common.h"

namespace {

const double kPow10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11,
                           1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// parses [p,e) completely as a float; returns false on malformed input
bool parse_f32(const char* p, const char* e, float* out) {
    if (p >= e) return false;
    const char* q = p;
    bool neg = false;
    if (*q == '-' || *q == '+') { neg = (*q == '-'); ++q; }
    uint64_t n = 0;
    int digits = 0, frac = 0;
    bool fast = true, any = false;
    const char* r = q;
    for (; r < e && *r >= '0' && *r <= '9'; ++r) { n = n * 10 + (uint64_t)(*r - '0'); ++digits; any = true; if (n >= (1ull << 29)) fast = false; if (digits > 18) break; }
    if (r < e && *r == '.') {
        ++r;
        for (; r < e && *r >= '0' && *r <= '9'; ++r) { n = n * 10 + (uint64_t)(*r - '0'); ++digits; ++frac; any = true; if (n >= (1ull << 29)) fast = false; if (digits > 18) break; }
    }
    if (fast && any && r == e && frac <= 22) {
        const double v = (double)n / kPow10[frac];
        *out = (float)(neg ? -v : v);
        return true;
    }
    // general path (exponents, inf/nan, long mantissas): glibc strtof on a NUL-terminated copy
    char buf[128];
    const size_t len = (size_t)(e - p);
    if (len >= sizeof(buf)) return false;
    memcpy(buf, p, len);
    buf[len] = 0;
    char* end = nullptr;
    errno = 0;
    const float v = strtof(buf, &end);
    if (end != buf + len || end == buf) return false;
    // TF's StringToNumber (strings::safe_strtof) rejects leading whitespace-only / trailing junk; accepts inf/nan
    *out = v;
    return true;
}

bool parse_i32(const char* p, const char* e, int32_t* out) {
    if (p >= e) return false;
    bool neg = false;
    if (*p == '-' || *p == '+') { neg = (*p == '-'); ++p; }
    if (p >= e) return false;
    int64_t v = 0;
    for (; p < e; ++p) {
        if (*p < '0' || *p > '9') return false;
        v = v * 10 + (*p - '0');
        if (v > (int64_t)1 << 32) return false;
    }
    if (neg) v = -v;
    if (v < INT32_MIN || v > INT32_MAX) return false;
    *out = (int32_t)v;
    return true;
}

}  // namespace

extern "C" int dctr_parse_libsvm(const char* h_text, size_t nbytes, int field_size, int64_t max_rows, int32_t* h_ids,
                                 float* h_vals, float* h_labels, int64_t* n_rows, size_t* n_consumed) {
    using namespace dctr;
    DCTR_REQUIRE(h_text != nullptr && h_ids != nullptr && h_vals != nullptr && h_labels != nullptr && n_rows != nullptr,
                 "null argument");
    DCTR_REQUIRE(field_size > 0, "field_size must be > 0");
    const char* p = h_text;
    const char* end = h_text + nbytes;
    int64_t row = 0;
    size_t consumed = 0;
    int64_t line_no = 0;
    while (p < end && row < max_rows) {
        const char* eol = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
        const char* le = eol ? eol : end;
        const char* next = eol ? eol + 1 : end;
        ++line_no;
        const char* q = p;
        if (le > q && le[-1] == '\r') --le;
        // token 0: label
        while (q < le && *q == ' ') ++q;
        if (q == le) { p = next; consumed = (size_t)(p - h_text); continue; }   // empty line: TextLineDataset yields "", skipped here
        const char* t = q;
        while (q < le && *q != ' ') ++q;
        float label;
        if (!parse_f32(t, q, &label)) {
            set_error("StringToNumberOp could not correctly convert string: %.*s (line %lld, label)", (int)(q - t), t, (long long)line_no);
            return DCTR_ERR_PARSE;
        }
        int f = 0;
        int32_t* ir = h_ids + (size_t)row * field_size;
        float* vr = h_vals + (size_t)row * field_size;
        while (true) {
            while (q < le && *q == ' ') ++q;
            if (q == le) break;
            t = q;
            // ---- one-pass fast path for the token shape every Criteo line is made of: digits ':' digits [ '.' digits ] then a
            // space or the end of the line.  Anything else (signs, exponents, empty pieces, >9-digit ids, mantissas >= 2^29, more
            // tokens than fields) rewinds to `t` and takes the general path below, which owns all the error messages.
            if (f < field_size) {
                uint32_t id = 0;
                int nd = 0;
                while (q < le && (unsigned)(*q - '0') <= 9u) { id = id * 10u + (unsigned)(*q - '0'); ++q; ++nd; }
                if (nd >= 1 && nd <= 9 && q < le && *q == ':') {
                    ++q;
                    uint64_t n = 0;
                    int digits = 0, frac = 0;
                    while (q < le && (unsigned)(*q - '0') <= 9u && digits < 19) { n = n * 10 + (uint64_t)(*q - '0'); ++q; ++digits; }
                    if (q < le && *q == '.') {
                        ++q;
                        while (q < le && (unsigned)(*q - '0') <= 9u && digits < 19) { n = n * 10 + (uint64_t)(*q - '0'); ++q; ++digits; ++frac; }
                    }
                    if (digits >= 1 && digits <= 18 && (q == le || *q == ' ') && n < (1ull << 29) && frac <= 22) {
                        ir[f] = (int32_t)id;
                        vr[f] = frac == 0 ? (float)n : (float)((double)n / kPow10[frac]);       // (same conversion as parse_f32's fast path)
                        ++f;
                        continue;
                    }
                }
                q = t;
            }
            while (q < le && *q != ' ') ++q;
            // id:val with empty pieces dropped -> exactly two pieces
            const char* a0 = t;
            while (a0 < q && *a0 == ':') ++a0;
            const char* a1 = a0;
            while (a1 < q && *a1 != ':') ++a1;
            const char* b0 = a1;
            while (b0 < q && *b0 == ':') ++b0;
            const char* b1 = b0;
            while (b1 < q && *b1 != ':') ++b1;
            const char* c0 = b1;
            while (c0 < q && *c0 == ':') ++c0;
            if (a0 == a1 || b0 == b1 || c0 != q) {
                set_error("line %lld: token '%.*s' is not id:val (reshape of string_split(':') to [F,2] fails)", (long long)line_no,
                          (int)(q - t), t);
                return DCTR_ERR_PARSE;
            }
            if (f >= field_size) {
                set_error("line %lld: more than field_size=%d id:val tokens (batch/reshape to [-1,%d] fails)", (long long)line_no,
                          field_size, field_size);
                return DCTR_ERR_PARSE;
            }
            if (!parse_i32(a0, a1, &ir[f])) {
                set_error("StringToNumberOp could not correctly convert string: %.*s (line %lld, id)", (int)(a1 - a0), a0, (long long)line_no);
                return DCTR_ERR_PARSE;
            }
            if (!parse_f32(b0, b1, &vr[f])) {
                set_error("StringToNumberOp could not correctly convert string: %.*s (line %lld, value)", (int)(b1 - b0), b0, (long long)line_no);
                return DCTR_ERR_PARSE;
            }
            ++f;
        }
        if (f != field_size) {
            set_error("line %lld: %d id:val tokens, expected field_size=%d (batch/reshape to [-1,%d] fails)", (long long)line_no, f,
                      field_size, field_size);
            return DCTR_ERR_PARSE;
        }
        h_labels[row] = label;
        ++row;
        p = next;
        consumed = (size_t)(p - h_text);
    }
    // swallow trailing blank lines so callers see end-of-input
    *n_rows = row;
    if (n_consumed) *n_consumed = consumed;
    return DCTR_OK;
}

// The same decode over a whole file with a thread team inside the library (tf.data's map(decode_libsvm, num_parallel_calls),
// DeepFM.py:84): the buffer is cut at line boundaries into `threads` chunks; pass 1 counts the row-producing lines of every chunk
// (a line is skipped iff it holds nothing but spaces / a '\r', exactly as above), pass 2 parses every chunk straight into its rows
// of the caller's arrays.  h_ids == NULL: count only.  On a malformed line the buffer is re-parsed serially so that the error
// (message, line number) is the single-threaded one.
extern "C" int dctr_parse_libsvm_mt(const char* h_text, size_t nbytes, int field_size, int threads, int32_t* h_ids, float* h_vals,
                                    float* h_labels, int64_t capacity_rows, int64_t* n_rows) {
    using namespace dctr;
    DCTR_REQUIRE(h_text != nullptr || nbytes == 0, "null buffer");
    DCTR_REQUIRE(n_rows != nullptr && field_size > 0, "bad argument");
    threads = std::max(1, std::min(threads, 1024));
    if (nbytes < ((size_t)1 << 20)) threads = 1;
    std::vector<size_t> cut((size_t)threads + 1, nbytes);
    cut[0] = 0;
    for (int t = 1; t < threads; ++t) {
        size_t pos = std::max(cut[(size_t)t - 1], nbytes / (size_t)threads * (size_t)t);
        const char* nl = pos < nbytes ? static_cast<const char*>(memchr(h_text + pos, '\n', nbytes - pos)) : nullptr;
        cut[(size_t)t] = nl ? (size_t)(nl - h_text) + 1 : nbytes;
    }
    std::vector<int64_t> rows((size_t)threads, 0);
    auto count = [&](int t) {
        const char* p = h_text + cut[(size_t)t];
        const char* end = h_text + cut[(size_t)t + 1];
        int64_t n = 0;
        while (p < end) {
            const char* eol = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
            const char* le = eol ? eol : end;
            if (le > p && le[-1] == '\r') --le;
            const char* q = p;
            while (q < le && *q == ' ') ++q;
            if (q < le) ++n;
            p = eol ? eol + 1 : end;
        }
        rows[(size_t)t] = n;
    };
    auto run = [&](const std::function<void(int)>& fn) {
        std::vector<std::thread> team;
        for (int t = 1; t < threads; ++t) team.emplace_back(fn, t);
        fn(0);
        for (auto& th : team) th.join();
    };
    run(count);
    std::vector<int64_t> first((size_t)threads + 1, 0);
    for (int t = 0; t < threads; ++t) first[(size_t)t + 1] = first[(size_t)t] + rows[(size_t)t];
    *n_rows = first[(size_t)threads];
    if (h_ids == nullptr) return DCTR_OK;
    DCTR_REQUIRE(h_vals != nullptr && h_labels != nullptr, "null output");
    DCTR_REQUIRE(capacity_rows >= *n_rows, "capacity_rows=%lld < %lld lines", (long long)capacity_rows, (long long)*n_rows);
    std::vector<int> rc((size_t)threads, DCTR_OK);
    auto parse = [&](int t) {
        int64_t n = 0;
        const size_t r0 = (size_t)first[(size_t)t];
        rc[(size_t)t] = dctr_parse_libsvm(h_text + cut[(size_t)t], cut[(size_t)t + 1] - cut[(size_t)t], field_size, rows[(size_t)t] + 1,
                                          h_ids + r0 * (size_t)field_size, h_vals + r0 * (size_t)field_size, h_labels + r0, &n, nullptr);
        if (rc[(size_t)t] == DCTR_OK && n != rows[(size_t)t]) rc[(size_t)t] = DCTR_ERR_PARSE;
    };
    run(parse);
    for (int t = 0; t < threads; ++t)
        if (rc[(size_t)t] != DCTR_OK) {
            // the canonical error: parse serially up to the failure (outputs into scratch-free positions are harmless: same rows)
            int64_t n = 0;
            const int r = dctr_parse_libsvm(h_text, nbytes, field_size, capacity_rows, h_ids, h_vals, h_labels, &n, nullptr);
            if (r != DCTR_OK) return r;
            set_error("multi-threaded libsvm parse failed in chunk %d although the serial parse succeeded", t);
            return DCTR_ERR_PARSE;
        }
    return DCTR_OK;
}

// CSV text -> column tensors.  Replaces tf.decode_csv(line, record_defaults) of wide_n_deep.py:67-73 (record_defaults
// wide_n_deep.py:59-64: one float label, 13 float columns, 26 int columns): fields split on ',', an EMPTY field takes its
// column's default, float columns go through the same correctly rounded conversion as string_to_number, int columns must be
// plain base-10 int32 [TF-1.4 DecodeCSVOp: "Field i is not a valid int32/float"].  Quoted fields are not used by the
// reference's data and are rejected.  kinds[c]: 0 = float column, 1 = int32 column; float / int columns are written in order
// of appearance into h_f [rows, n_float] / h_i [rows, n_int].  Host code, re-entrant.
extern "C" int dctr_parse_csv(const char* h_text, size_t nbytes, int n_cols, const int8_t* kinds, const float* f_defaults,
                              const int32_t* i_defaults, int64_t max_rows, float* h_f, int32_t* h_i, int64_t* n_rows,
                              size_t* n_consumed) {
    using namespace dctr;
    DCTR_REQUIRE(h_text != nullptr && kinds != nullptr && n_rows != nullptr && n_cols > 0, "bad argument");
    int nf = 0, ni = 0;
    for (int c = 0; c < n_cols; ++c) (kinds[c] == 0 ? nf : ni) += 1;
    DCTR_REQUIRE((nf == 0 || (h_f != nullptr && f_defaults != nullptr)) && (ni == 0 || (h_i != nullptr && i_defaults != nullptr)), "null output");
    const char* p = h_text;
    const char* end = h_text + nbytes;
    int64_t row = 0, line_no = 0;
    size_t consumed = 0;
    while (p < end && row < max_rows) {
        const char* eol = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
        const char* le = eol ? eol : end;
        const char* next = eol ? eol + 1 : end;
        ++line_no;
        if (le > p && le[-1] == '\r') --le;
        if (le == p) { p = next; consumed = (size_t)(p - h_text); continue; }
        float* fr = h_f + (size_t)row * nf;
        int32_t* ir = h_i + (size_t)row * ni;
        int c = 0, fc = 0, ic = 0;
        const char* q = p;
        while (true) {
            const char* t = q;
            while (q < le && *q != ',') ++q;
            if (c >= n_cols) { set_error("Expect %d fields but have more in record (line %lld)", n_cols, (long long)line_no); return DCTR_ERR_PARSE; }
            if (t < q && *t == '"') { set_error("line %lld: quoted CSV fields are not supported", (long long)line_no); return DCTR_ERR_PARSE; }
            if (kinds[c] == 0) {
                if (t == q) fr[fc] = f_defaults[fc];
                else if (!parse_f32(t, q, &fr[fc])) {
                    set_error("Field %d in record is not a valid float: %.*s (line %lld)", c, (int)(q - t), t, (long long)line_no);
                    return DCTR_ERR_PARSE;
                }
                ++fc;
            } else {
                if (t == q) ir[ic] = i_defaults[ic];
                else if (!parse_i32(t, q, &ir[ic])) {
                    set_error("Field %d in record is not a valid int32: %.*s (line %lld)", c, (int)(q - t), t, (long long)line_no);
                    return DCTR_ERR_PARSE;
                }
                ++ic;
            }
            ++c;
            if (q == le) break;
            ++q;                    // skip the comma; a trailing comma yields one more (empty) field
        }
        if (c != n_cols) { set_error("Expect %d fields but have %d in record (line %lld)", n_cols, c, (long long)line_no); return DCTR_ERR_PARSE; }
        ++row;
        p = next;
        consumed = (size_t)(p - h_text);
    }
    *n_rows = row;
    if (n_consumed) *n_consumed = consumed;
    return DCTR_OK;
}

// decode_csv over a whole buffer with a thread team inside the library (wide_n_deep.py:84 map(parse_csv, num_parallel_calls=10));
// same scheme as dctr_parse_libsvm_mt: count the non-empty lines of every chunk, then parse every chunk into its rows.
// h_f == NULL && h_i == NULL: count only.
extern "C" int dctr_parse_csv_mt(const char* h_text, size_t nbytes, int n_cols, const int8_t* kinds, const float* f_defaults,
                                 const int32_t* i_defaults, int threads, float* h_f, int32_t* h_i, int64_t capacity_rows,
                                 int64_t* n_rows) {
    using namespace dctr;
    DCTR_REQUIRE((h_text != nullptr || nbytes == 0) && kinds != nullptr && n_rows != nullptr && n_cols > 0, "bad argument");
    int nf = 0, ni = 0;
    for (int c = 0; c < n_cols; ++c) (kinds[c] == 0 ? nf : ni) += 1;
    threads = std::max(1, std::min(threads, 1024));
    if (nbytes < ((size_t)1 << 20)) threads = 1;
    std::vector<size_t> cut((size_t)threads + 1, nbytes);
    cut[0] = 0;
    for (int t = 1; t < threads; ++t) {
        size_t pos = std::max(cut[(size_t)t - 1], nbytes / (size_t)threads * (size_t)t);
        const char* nl = pos < nbytes ? static_cast<const char*>(memchr(h_text + pos, '\n', nbytes - pos)) : nullptr;
        cut[(size_t)t] = nl ? (size_t)(nl - h_text) + 1 : nbytes;
    }
    std::vector<int64_t> rows((size_t)threads, 0);
    auto run = [&](const std::function<void(int)>& fn) {
        std::vector<std::thread> team;
        for (int t = 1; t < threads; ++t) team.emplace_back(fn, t);
        fn(0);
        for (auto& th : team) th.join();
    };
    run([&](int t) {
        const char* p = h_text + cut[(size_t)t];
        const char* end = h_text + cut[(size_t)t + 1];
        int64_t n = 0;
        while (p < end) {
            const char* eol = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
            const char* le = eol ? eol : end;
            if (le > p && le[-1] == '\r') --le;
            if (le > p) ++n;                                // (an empty line yields no record, as in dctr_parse_csv)
            p = eol ? eol + 1 : end;
        }
        rows[(size_t)t] = n;
    });
    std::vector<int64_t> first((size_t)threads + 1, 0);
    for (int t = 0; t < threads; ++t) first[(size_t)t + 1] = first[(size_t)t] + rows[(size_t)t];
    *n_rows = first[(size_t)threads];
    if (h_f == nullptr && h_i == nullptr) return DCTR_OK;
    DCTR_REQUIRE(capacity_rows >= *n_rows, "capacity_rows=%lld < %lld records", (long long)capacity_rows, (long long)*n_rows);
    std::vector<int> rc((size_t)threads, DCTR_OK);
    run([&](int t) {
        int64_t n = 0;
        const size_t r0 = (size_t)first[(size_t)t];
        rc[(size_t)t] = dctr_parse_csv(h_text + cut[(size_t)t], cut[(size_t)t + 1] - cut[(size_t)t], n_cols, kinds, f_defaults, i_defaults,
                                       rows[(size_t)t] + 1, h_f ? h_f + r0 * (size_t)nf : nullptr, h_i ? h_i + r0 * (size_t)ni : nullptr, &n,
                                       nullptr);
        if (rc[(size_t)t] == DCTR_OK && n != rows[(size_t)t]) rc[(size_t)t] = DCTR_ERR_PARSE;
    });
    for (int t = 0; t < threads; ++t)
        if (rc[(size_t)t] != DCTR_OK) {
            int64_t n = 0;                                   // the canonical (serial) error message and line number
            const int r = dctr_parse_csv(h_text, nbytes, n_cols, kinds, f_defaults, i_defaults, capacity_rows, h_f, h_i, &n, nullptr);
            if (r != DCTR_OK) return r;
            set_error("multi-threaded CSV parse failed in chunk %d although the serial parse succeeded", t);
            return DCTR_ERR_PARSE;
        }
    return DCTR_OK;
}

