// Variable-length (multi-hot) lookups: tf.nn.embedding_lookup_sparse(params, sp_ids, sp_weights, combiner="sum") as the
// DIN / ESMM scripts use it (DIN.py:148,180-183; DeepCvrMTL.py:155-159) = gather + weight + segment sum over a CSR batch
// (K2/K8 generalised from fixed-F to CSR).  Op-level entry points, plus the slot-addressed forms the DIN / ESMM
// engine models use (engine.hip, dctr_train_step_csr).
//   forward : out[b, :] = sum_{j in [offsets[b], offsets[b+1])} weights[j] * emb[ids[j], :]        (empty row -> zeros)
//   backward: the IndexedSlices gradient, segment-summed per distinct id into a dctr_group's compact rows (group.hip),
//             ready for dctr_opt_table -- the same machinery as the fixed-F path, with a per-entry example index
#include "ops.h"
#include "lag.h"

namespace dctr {

// EL lanes walk a segment's entries side by side (KQ float4 pieces each), then fold their partial sums with shuffles: a
// multi-hot slot of ~60 ids keeps 16 row loads in flight instead of one dependent chain.  EL = 1: one lane per (segment, piece).
// LAG: table rows may lag (lag.h): an entry's row is advanced to step t-1 in registers before it is summed (nothing written back)
template <int KQ, int EL, bool LAG>
__global__ __launch_bounds__(256) void lookup_sparse_fwd_kernel(const float4* __restrict__ emb, int64_t rows, const int32_t* __restrict__ offsets,
                                                               const int32_t* __restrict__ ids, const float* __restrict__ weights, int B,
                                                               int S, float4* __restrict__ out, int out_ld4, int32_t* __restrict__ status,
                                                               LagView L) {
    // B segments; segment b is slot b % S of output row b / S (S == 1: one K-wide output per row)
    static_assert(KQ * EL <= 64, "a segment's lanes must sit in one wave");
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = (int)(t / (KQ * EL)), r = (int)(t % (KQ * EL));
    const int el = r / KQ, kq = r % KQ;
    if (b >= B) return;                                  // (whole groups leave together: KQ*EL divides the wave size)
    const int j0 = offsets[b], j1 = offsets[b + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = j0 + el; j < j1; j += EL) {
        const int id = ids[j];
        if (id < 0 || (int64_t)id >= rows) {            // TF CPU gather: InvalidArgumentError [TF-1.4]
            if (kq == 0) { atomicExch(&status[1], id); atomicExch(&status[0], 1); }
            continue;
        }
        const float w = weights != nullptr ? weights[j] : 1.0f;
        float4 v = emb[(size_t)id * KQ + kq];
        if constexpr (LAG) {
            const int64_t Tm1 = L.state->t - 1;
            const int nl = lag_behind(Tm1, L.ts[id]);
            if (nl > 0) {
                float4 m = L.s0[(size_t)id * KQ + kq], vv = L.s1[(size_t)id * KQ + kq];
                lag_catch_up4(L.state, L.state->hyper, L.l2, Tm1 - nl + 1, nl, v, m, vv);
            }
        }
        acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
    }
#pragma unroll
    for (int o = KQ; o < KQ * EL; o <<= 1) {
        acc.x += __shfl_xor(acc.x, o); acc.y += __shfl_xor(acc.y, o); acc.z += __shfl_xor(acc.z, o); acc.w += __shfl_xor(acc.w, o);
    }
    if (el == 0) out[(size_t)(b / S) * out_ld4 + (size_t)(b % S) * KQ + kq] = acc;
}

// entry_row[j] = b for offsets[b] <= j < offsets[b+1]   (binary search per entry); with S slots per output row the value is
// the float4 offset of the segment's K-wide piece, (b / S) * ld4 + (b % S) * KQ  (S == 1, ld4 == 1, KQ == 0: the row b itself)
__global__ __launch_bounds__(256) void entry_row_kernel(const int32_t* __restrict__ offsets, int B, int nnz, int S, int ld4, int KQ,
                                                       int32_t* __restrict__ entry_row) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nnz) return;
    int lo = 0, hi = B;                                  // invariant: offsets[lo] <= j < offsets[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offsets[mid] <= j) lo = mid; else hi = mid;
    }
    entry_row[j] = S == 1 && KQ == 0 ? lo : (lo / S) * ld4 + (lo % S) * KQ;
}

int lookup_sparse_slots_fwd(const float* emb, int64_t rows, int K, const int32_t* offsets, const int32_t* ids, const float* weights,
                            int n_seg, int S, float* out, int out_ld, int32_t* status, hipStream_t st, int64_t nnz_hint, const LagView* lag) {
    if (n_seg <= 0) return DCTR_OK;
    const int KQ = K / 4;
    // entry lanes per segment from the average segment length (nnz_hint < 0: unknown -> one lane)
    const int64_t avg = nnz_hint > 0 ? nnz_hint / n_seg : 0;
    int EL = avg >= 8 ? 16 : (avg >= 2 ? 4 : 1);
    while (EL > 1 && KQ * EL > 64) EL >>= 2;
    const float4* e4 = reinterpret_cast<const float4*>(emb);
    float4* o4 = reinterpret_cast<float4*>(out);
    const LagView LV = lag ? *lag : LagView{};
#define DCTR_S2(Q, L)                                                                                                            \
    do {                                                                                                                         \
        if (lag) lookup_sparse_fwd_kernel<Q, L, true><<<ceil_div((int64_t)n_seg * Q * L, 256), 256, 0, st>>>(e4, rows, offsets, ids, weights, n_seg, S, o4, out_ld / 4, status, LV); \
        else lookup_sparse_fwd_kernel<Q, L, false><<<ceil_div((int64_t)n_seg * Q * L, 256), 256, 0, st>>>(e4, rows, offsets, ids, weights, n_seg, S, o4, out_ld / 4, status, LV); \
    } while (0)
#define DCTR_S(Q)                                                                        \
    case Q:                                                                              \
        if (EL == 16 && Q * 16 <= 64) DCTR_S2(Q, (Q * 16 <= 64 ? 16 : 1));               \
        else if (EL >= 4 && Q * 4 <= 64) DCTR_S2(Q, (Q * 4 <= 64 ? 4 : 1));              \
        else DCTR_S2(Q, 1);                                                              \
        break
    switch (KQ) {
        DCTR_S(1); DCTR_S(2); DCTR_S(4); DCTR_S(8); DCTR_S(16); DCTR_S(32); DCTR_S(64);
        default: set_error("lookup_sparse: K=%d unsupported (K/4 must be a power of two)", K); return DCTR_ERR_UNSUPPORTED;
    }
#undef DCTR_S
#undef DCTR_S2
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

// the same table written segment by segment (4 lanes per segment): no search, entries of a segment are contiguous
__global__ __launch_bounds__(256) void entry_fill_kernel(const int32_t* __restrict__ offsets, int n_seg, int S, int ld4, int KQ,
                                                        int32_t* __restrict__ entry_row) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int seg = (int)(t >> 2), l = (int)(t & 3);
    if (seg >= n_seg) return;
    const int v = S == 1 && KQ == 0 ? seg : (seg / S) * ld4 + (seg % S) * KQ;
    for (int j = offsets[seg] + l; j < offsets[seg + 1]; j += 4) entry_row[j] = v;
}

int csr_entry_offsets(const int32_t* offsets, int n_seg, int nnz, int S, int ld, int K, int32_t* entry_off, hipStream_t st) {
    if (nnz <= 0) return DCTR_OK;
    if (nnz >= n_seg / 4) {           // (very sparse segment lists: the per-entry search touches less)
        entry_fill_kernel<<<ceil_div((int64_t)n_seg * 4, 256), 256, 0, st>>>(offsets, n_seg, S, ld / 4, K / 4, entry_off);
        DCTR_LAUNCH_CHECK();
        return DCTR_OK;
    }
    entry_row_kernel<<<ceil_div(nnz, 256), 256, 0, st>>>(offsets, n_seg, nnz, S, ld / 4, K / 4, entry_off);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

}  // namespace dctr

using namespace dctr;

extern "C" {

int dctr_embed_lookup_sparse_fwd(const float* d_emb, int64_t rows, int K, const int32_t* d_offsets, const int32_t* d_ids,
                                 const float* d_weights, int B, float* d_out, int out_ld, int32_t* d_status, void* stream) {
    DCTR_REQUIRE(d_emb && d_offsets && d_ids && d_out && d_status, "null argument");
    DCTR_REQUIRE(K % 4 == 0 && K >= 4 && K <= 256 && out_ld % 4 == 0 && out_ld >= K, "lookup_sparse: K=%d / out_ld=%d unsupported", K, out_ld);
    return lookup_sparse_slots_fwd(d_emb, rows, K, d_offsets, d_ids, d_weights, B, 1, d_out, out_ld, d_status, as_stream(stream), -1);
}

int dctr_embed_lookup_sparse_bwd(dctr_group_t g, const float* d_dout, int dout_ld, const int32_t* d_offsets, const int32_t* d_ids,
                                 const float* d_weights, int B, int nnz, int K, int32_t* d_entry_row, void* stream) {
    DCTR_REQUIRE(g && d_dout && d_offsets && d_ids && d_entry_row, "null argument");
    Group* G = reinterpret_cast<Group*>(g);
    hipStream_t st = as_stream(stream);
    DCTR_REQUIRE(nnz >= 0 && (int64_t)nnz <= G->max_entries, "lookup_sparse_bwd: nnz=%d exceeds the group's capacity", nnz);
    DCTR_TRY(group_ids(G, d_ids, nnz, 1, st));
    if (nnz == 0 || B <= 0) return DCTR_OK;
    DCTR_TRY(csr_entry_offsets(d_offsets, B, nnz, 1, 4, 0, d_entry_row, st));
    // nnz "examples" of one field: entry j reads the gradient row of example entry_row[j], scaled by weights[j]
    return embed_scatter_bwd(G, d_dout, dout_ld, nullptr, 0, nullptr, nullptr, nullptr, d_weights, nnz, 1, K, DCTR_GATHER_RAW, G->gemb,
                             nullptr, st, 1, d_entry_row);
}

}  // extern "C"
