// TFRecord files of tf.train.Example -> the slot-ordered CSR batches of the DIN / ESMM models.  Host code, re-entrant.
// Replaces tf.data.TFRecordDataset + tf.parse_single_example with the feature spec of DIN.py:59-76 / DeepCvrMTL.py:63-80
// (FixedLenFeature scalars and vectors, VarLenFeature id / value lists) for files written by the reference's
// Feature_pipeline/get_tfrecord.py:44-98.
//   record framing [TFRecord format]: uint64 length | uint32 masked crc32c(length) | payload | uint32 masked crc32c(payload),
//   masked(c) = ((c >> 15) | (c << 17)) + 0xa282ead8, little endian.
//   payload = Example{1: Features{1: map<string, Feature>}}; Feature{1: BytesList | 2: FloatList{1: packed float} |
//   3: Int64List{1: packed varint}}  (unpacked repeated encodings are accepted as well, as protobuf parsers must).
#include <vector>

#include "ops.h"

namespace dctr {
namespace {

uint32_t crc_table[8][256];
bool crc_ready = false;

void crc_init() {
    if (crc_ready) return;
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;      // CRC-32C (Castagnoli), reflected
        crc_table[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int t = 1; t < 8; ++t) crc_table[t][i] = (crc_table[t - 1][i] >> 8) ^ crc_table[0][crc_table[t - 1][i] & 0xFF];
    crc_ready = true;
}

uint32_t crc32c(const uint8_t* p, size_t n) {
    uint32_t c = 0xFFFFFFFFu;
    while (n >= 8) {                                        // slicing-by-8
        uint32_t lo, hi;
        memcpy(&lo, p, 4); memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = crc_table[7][lo & 0xFF] ^ crc_table[6][(lo >> 8) & 0xFF] ^ crc_table[5][(lo >> 16) & 0xFF] ^ crc_table[4][lo >> 24] ^
            crc_table[3][hi & 0xFF] ^ crc_table[2][(hi >> 8) & 0xFF] ^ crc_table[1][(hi >> 16) & 0xFF] ^ crc_table[0][hi >> 24];
        p += 8; n -= 8;
    }
    while (n--) c = (c >> 8) ^ crc_table[0][(c ^ *p++) & 0xFF];
    return c ^ 0xFFFFFFFFu;
}

inline uint32_t masked_crc(const uint8_t* p, size_t n) {
    const uint32_t c = crc32c(p, n);
    return ((c >> 15) | (c << 17)) + 0xa282ead8u;
}

// ---- protobuf wire format ------------------------------------------------------------------------------------------------------
struct Cursor {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    bool done() const { return p >= end; }
    uint64_t varint() {
        uint64_t v = 0;
        for (int shift = 0; shift < 64 && p < end; shift += 7) {
            const uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7F) << shift;
            if (!(b & 0x80)) return v;
        }
        ok = false;
        return 0;
    }
    Cursor sub() {                                          // length-delimited field
        const uint64_t n = varint();
        if (!ok || n > (uint64_t)(end - p)) { ok = false; return Cursor{p, p, false}; }
        Cursor c{p, p + n};
        p += n;
        return c;
    }
    void skip(int wire) {
        switch (wire) {
            case 0: varint(); break;
            case 1: if (end - p < 8) ok = false; else p += 8; break;
            case 2: sub(); break;
            case 5: if (end - p < 4) ok = false; else p += 4; break;
            default: ok = false;
        }
    }
};

struct FeatureRef {           // one map entry of Features: key + the kind's list payload
    const uint8_t* key = nullptr;
    size_t key_len = 0;
    int kind = 0;             // 1 bytes, 2 float, 3 int64, 0 = empty Feature
    Cursor list{nullptr, nullptr};
};

// Example -> its map entries
bool example_features(const uint8_t* buf, size_t n, std::vector<FeatureRef>& out) {
    out.clear();
    Cursor ex{buf, buf + n};
    while (ex.ok && !ex.done()) {
        const uint64_t tag = ex.varint();
        if (!ex.ok) return false;
        if ((tag >> 3) == 1 && (tag & 7) == 2) {            // Example.features
            Cursor fs = ex.sub();
            if (!ex.ok) return false;
            while (fs.ok && !fs.done()) {
                const uint64_t t2 = fs.varint();
                if ((t2 >> 3) == 1 && (t2 & 7) == 2) {      // Features.feature map entry
                    Cursor ent = fs.sub();
                    if (!fs.ok) return false;
                    FeatureRef r;
                    while (ent.ok && !ent.done()) {
                        const uint64_t t3 = ent.varint();
                        if ((t3 >> 3) == 1 && (t3 & 7) == 2) { Cursor k = ent.sub(); r.key = k.p; r.key_len = (size_t)(k.end - k.p); }
                        else if ((t3 >> 3) == 2 && (t3 & 7) == 2) {        // Feature
                            Cursor f = ent.sub();
                            while (f.ok && !f.done()) {
                                const uint64_t t4 = f.varint();
                                const int fld = (int)(t4 >> 3);
                                if (fld >= 1 && fld <= 3 && (t4 & 7) == 2) { r.kind = fld; r.list = f.sub(); }
                                else f.skip((int)(t4 & 7));
                            }
                            if (!f.ok) return false;
                        } else ent.skip((int)(t3 & 7));
                    }
                    if (!ent.ok) return false;
                    out.push_back(r);
                } else fs.skip((int)(t2 & 7));
            }
            if (!fs.ok) return false;
        } else ex.skip((int)(tag & 7));
    }
    return ex.ok;
}

// values of an Int64List / FloatList payload (field 1, packed or not); returns false on a malformed list
template <typename F>
bool for_each_int64(Cursor c, F&& f) {
    while (c.ok && !c.done()) {
        const uint64_t t = c.varint();
        if ((t >> 3) == 1 && (t & 7) == 2) { Cursor pk = c.sub(); while (pk.ok && !pk.done()) { const uint64_t v = pk.varint(); if (pk.ok) f((int64_t)v); } if (!pk.ok) return false; }
        else if ((t >> 3) == 1 && (t & 7) == 0) { const uint64_t v = c.varint(); if (c.ok) f((int64_t)v); }
        else c.skip((int)(t & 7));
    }
    return c.ok;
}
template <typename F>
bool for_each_float(Cursor c, F&& f) {
    while (c.ok && !c.done()) {
        const uint64_t t = c.varint();
        if ((t >> 3) == 1 && (t & 7) == 2) {
            Cursor pk = c.sub();
            if (!c.ok || ((pk.end - pk.p) & 3)) return false;
            for (; pk.p < pk.end; pk.p += 4) { float v; memcpy(&v, pk.p, 4); f(v); }
        } else if ((t >> 3) == 1 && (t & 7) == 5) {
            if (c.end - c.p < 4) return false;
            float v; memcpy(&v, c.p, 4); c.p += 4; f(v);
        } else c.skip((int)(t & 7));
    }
    return c.ok;
}

const FeatureRef* find(const std::vector<FeatureRef>& fs, const char* name) {
    if (name == nullptr) return nullptr;
    const size_t n = strlen(name);
    const FeatureRef* hit = nullptr;
    for (const auto& f : fs) if (f.key_len == n && memcmp(f.key, name, n) == 0) hit = &f;      // (a repeated key: the last one wins, as in protobuf maps)
    return hit;
}

}  // namespace
}  // namespace dctr

using namespace dctr;

extern "C" {

int dctr_crc32c(const uint8_t* h_buf, size_t nbytes, int masked, uint32_t* h_crc) {
    DCTR_REQUIRE(h_crc != nullptr && (h_buf != nullptr || nbytes == 0), "null argument");
    crc_init();
    *h_crc = masked ? masked_crc(h_buf, nbytes) : crc32c(h_buf, nbytes);
    return DCTR_OK;
}

int dctr_tfrecord_scan(const uint8_t* h_buf, size_t nbytes, int64_t max_records, int verify_crc, int64_t* rec_off, int64_t* rec_len,
                       int64_t* n_records, size_t* n_consumed) {
    DCTR_REQUIRE(h_buf != nullptr || nbytes == 0, "null buffer");
    DCTR_REQUIRE(n_records != nullptr, "null argument");
    crc_init();
    size_t pos = 0;
    int64_t n = 0;
    while (pos + 12 <= nbytes && (max_records < 0 || n < max_records)) {
        uint64_t len;
        uint32_t crc;
        memcpy(&len, h_buf + pos, 8);
        memcpy(&crc, h_buf + pos + 8, 4);
        if (verify_crc && crc != masked_crc(h_buf + pos, 8)) {
            set_error("corrupted record at %zu: length checksum mismatch (DataLossError)", pos);
            return DCTR_ERR_PARSE;
        }
        if (len > nbytes || pos + 12 + len + 4 > nbytes) break;             // a truncated tail: stop at the last whole record
        if (verify_crc) {
            uint32_t dcrc;
            memcpy(&dcrc, h_buf + pos + 12 + len, 4);
            if (dcrc != masked_crc(h_buf + pos + 12, (size_t)len)) {
                set_error("corrupted record at %zu: data checksum mismatch (DataLossError)", pos);
                return DCTR_ERR_PARSE;
            }
        }
        if (rec_off) rec_off[n] = (int64_t)(pos + 12);
        if (rec_len) rec_len[n] = (int64_t)len;
        ++n;
        pos += 12 + (size_t)len + 4;
    }
    *n_records = n;
    if (n_consumed) *n_consumed = pos;
    return DCTR_OK;
}

int dctr_tfrecord_frame(const uint8_t* h_payload, size_t nbytes, uint8_t* h_out) {
    DCTR_REQUIRE(h_out != nullptr && (h_payload != nullptr || nbytes == 0), "null argument");
    crc_init();
    const uint64_t len = nbytes;
    memcpy(h_out, &len, 8);
    const uint32_t c1 = masked_crc(h_out, 8);
    memcpy(h_out + 8, &c1, 4);
    if (nbytes) memcpy(h_out + 12, h_payload, nbytes);
    const uint32_t c2 = masked_crc(h_out + 12, nbytes);
    memcpy(h_out + 12 + nbytes, &c2, 4);
    return DCTR_OK;
}

int dctr_examples_to_slot_csr(const uint8_t* h_buf, const int64_t* rec_off, const int64_t* rec_len, int64_t n_records,
                              const dctr_slot_spec* slots, int n_specs, const char* const* label_names, int n_labels,
                              int64_t feature_size, int64_t cap_entries, int32_t* h_offsets, int32_t* h_ids, float* h_weights,
                              float* h_labels, int64_t* n_entries) {
    DCTR_REQUIRE(h_buf && rec_off && rec_len && slots && n_entries, "null argument");
    DCTR_REQUIRE(n_specs > 0 && n_labels >= 0 && (n_labels == 0 || label_names != nullptr), "bad spec");
    const bool fill = h_ids != nullptr;                    // counting pass: h_offsets / h_ids / h_weights / h_labels may all be NULL
    DCTR_REQUIRE(!fill || (h_offsets && h_weights && (n_labels == 0 || h_labels)), "null output");
    std::vector<FeatureRef> fs;
    std::vector<float> wtmp;
    int64_t e = 0, seg = 0;
    if (fill) h_offsets[0] = 0;
    auto put = [&](int64_t id, float w, int64_t rec, const char* name) -> int {
        if (id < 0 || (feature_size > 0 && id >= feature_size)) {
            set_error("record %lld, feature '%s': id %lld outside [0, %lld) (InvalidArgumentError)", (long long)rec, name, (long long)id,
                      (long long)feature_size);
            return DCTR_ERR_INVALID_ARG;
        }
        if (fill) {
            if (e >= cap_entries) { set_error("slot CSR: more than cap_entries=%lld entries", (long long)cap_entries); return DCTR_ERR_INVALID_ARG; }
            h_ids[e] = (int32_t)id; h_weights[e] = w;
        }
        ++e;
        return DCTR_OK;
    };
    for (int64_t r = 0; r < n_records; ++r) {
        if (!example_features(h_buf + rec_off[r], (size_t)rec_len[r], fs)) {
            set_error("record %lld: could not parse the tf.train.Example (InvalidArgumentError)", (long long)r);
            return DCTR_ERR_PARSE;
        }
        for (int l = 0; l < n_labels; ++l) {               // FixedLenFeature([], tf.float32): exactly one value, no default
            const FeatureRef* f = find(fs, label_names[l]);
            int cnt = 0;
            float v = 0.f;
            if (f != nullptr && f->kind == 2 && !for_each_float(f->list, [&](float x) { v = x; ++cnt; })) cnt = -1;
            if (cnt != 1) {
                set_error("record %lld: Feature: %s (data type: float) is required but could not be found or has %d values", (long long)r,
                          label_names[l], cnt);
                return DCTR_ERR_PARSE;
            }
            if (fill) h_labels[(size_t)l * n_records + r] = v;
        }
        for (int s = 0; s < n_specs; ++s) {
            const dctr_slot_spec& sp = slots[s];
            const FeatureRef* fi = find(fs, sp.ids_feature);
            if (fi != nullptr && fi->kind != 3 && fi->kind != 0) {
                set_error("record %lld: feature '%s' is not an Int64List", (long long)r, sp.ids_feature);
                return DCTR_ERR_PARSE;
            }
            if (sp.fixed_len >= 0) {
                // FixedLenFeature([n] or []): n (or 1) values, each a slot of its own with one entry of weight 1
                const int want = sp.fixed_len == 0 ? 1 : sp.fixed_len;
                int cnt = 0, rc = DCTR_OK;
                if (fi != nullptr && fi->kind == 3) {
                    const bool ok = for_each_int64(fi->list, [&](int64_t id) {
                        if (rc != DCTR_OK || cnt >= want) { ++cnt; return; }
                        rc = put(id, 1.0f, r, sp.ids_feature);
                        ++cnt; ++seg;
                        if (fill && rc == DCTR_OK) h_offsets[seg] = (int32_t)e;
                    });
                    if (!ok) cnt = -1;
                }
                DCTR_TRY(rc);
                if (cnt != want) {
                    set_error("record %lld: Feature: %s (data type: int64) is required with %d values, found %d", (long long)r, sp.ids_feature,
                              want, cnt);
                    return DCTR_ERR_PARSE;
                }
            } else {
                // VarLenFeature ids (+ VarLenFeature weights of the same length): one slot, any number of entries
                wtmp.clear();
                if (sp.vals_feature != nullptr) {
                    const FeatureRef* fv = find(fs, sp.vals_feature);
                    if (fv != nullptr && fv->kind == 2 && !for_each_float(fv->list, [&](float x) { wtmp.push_back(x); })) {
                        set_error("record %lld: malformed FloatList '%s'", (long long)r, sp.vals_feature);
                        return DCTR_ERR_PARSE;
                    }
                }
                int64_t j = 0;
                int rc = DCTR_OK;
                bool short_w = false;
                if (fi != nullptr && fi->kind == 3) {
                    const bool ok = for_each_int64(fi->list, [&](int64_t id) {
                        if (rc != DCTR_OK) return;
                        float w = 1.0f;
                        if (sp.vals_feature != nullptr) { if (j < (int64_t)wtmp.size()) w = wtmp[(size_t)j]; else short_w = true; }
                        rc = put(id, w, r, sp.ids_feature);
                        ++j;
                    });
                    if (!ok) { set_error("record %lld: malformed Int64List '%s'", (long long)r, sp.ids_feature); return DCTR_ERR_PARSE; }
                }
                DCTR_TRY(rc);
                if (sp.vals_feature != nullptr && (short_w || j != (int64_t)wtmp.size())) {
                    // embedding_lookup_sparse requires sp_ids and sp_weights with identical indices
                    set_error("record %lld: '%s' has %lld ids but '%s' has %zu weights (InvalidArgumentError)", (long long)r, sp.ids_feature,
                              (long long)j, sp.vals_feature, wtmp.size());
                    return DCTR_ERR_PARSE;
                }
                ++seg;
                if (fill) h_offsets[seg] = (int32_t)e;
            }
        }
        if (e > 0x7FFFFFFFll) { set_error("slot CSR: more than 2^31 entries"); return DCTR_ERR_INVALID_ARG; }
    }
    *n_entries = e;
    return DCTR_OK;
}

}  // extern "C"
