// DIN attention pooling (attention_unit, DIN.py:152-172) over CSR batches.  For every entry j of a user multi-hot slot (b, s)
// paired with the ad slot a(s) of the same example:
//     ub_j = w_j * E[id_j]          ax = x[b, a(s)]  (the ad's embedding, already in the MLP input)
//     x_j  = [ub_j | ub_j - ax | ax]                 -> attention MLP (GEMM kernels) -> score sc_j -> att_j = sigmoid(sc_j)
//     x[b, s] = sum_j ub_j * att_j * [id_j > 0]      (replaces the slot's plain weighted sum)
// The dense [B, P] form of the script pads every row to the batch's longest list; padded positions carry dense_emb = 0 and so
// add nothing to the output or to any gradient -- the kernels here walk the real entries only.  The attention MLP runs over ALL
// nnz rows of the batch (rows of the other slots are zero-filled, their score unused): the GEMM sizes are then known on the host
// without a device round trip.
//   entry_off[j] = float4 offset of entry j's slot in x / dx  (b * ld4 + s * KQ);  pair_ad[s] = a(s) or -1
#include "ops.h"
#include "lag.h"

namespace dctr {

template <int KQ>
__global__ __launch_bounds__(256) void att_build_x_kernel(const float4* __restrict__ emb, int64_t rows, const int32_t* __restrict__ ids,
                                                         const float* __restrict__ weights, const int32_t* __restrict__ entry_off,
                                                         const int32_t* __restrict__ pair_ad, int nnz, const float4* __restrict__ x,
                                                         int ld4, float4* __restrict__ X, LagView L) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int j = (int)(t / KQ), kq = (int)(t % KQ);
    if (j >= nnz) return;
    const int off = entry_off[j], b = off / ld4, s = (off - b * ld4) / KQ;
    const int ad = pair_ad[s];
    float4 ub = make_float4(0.f, 0.f, 0.f, 0.f), ax = ub, df = ub;
    if (ad >= 0) {
        const int id = ids[j];
        if (id >= 0 && (int64_t)id < rows) {
            const float w = weights != nullptr ? weights[j] : 1.0f;
            float4 r = emb[(size_t)id * KQ + kq];
            if (L.ts != nullptr) {              // lagging rows (lag.h): as of step t-1, in registers
                const int64_t Tm1 = L.state->t - 1;
                const int nl = lag_behind(Tm1, L.ts[id]);
                if (nl > 0) {
                    float4 m = L.s0[(size_t)id * KQ + kq], vv = L.s1[(size_t)id * KQ + kq];
                    lag_catch_up4(L.state, L.state->hyper, L.l2, Tm1 - nl + 1, nl, r, m, vv);
                }
            }
            ub = make_float4(w * r.x, w * r.y, w * r.z, w * r.w);
        }
        ax = x[(size_t)b * ld4 + (size_t)ad * KQ + kq];
        df = make_float4(ub.x - ax.x, ub.y - ax.y, ub.z - ax.z, ub.w - ax.w);
    }
    float4* row = X + (size_t)j * 3 * KQ;
    row[kq] = ub; row[KQ + kq] = df; row[2 * KQ + kq] = ax;
}

// x[b, s] = sum_j ub_j att_j [id_j > 0] for the attention slots; att[j] = sigmoid(sc[j]) kept for the backward
template <int KQ, int EL>
__global__ __launch_bounds__(256) void att_pool_fwd_kernel(const int32_t* __restrict__ offsets, const int32_t* __restrict__ ids,
                                                          const int32_t* __restrict__ pair_ad, const float* __restrict__ sc, int n_seg,
                                                          int S, const float4* __restrict__ X, float* __restrict__ att,
                                                          float4* __restrict__ x, int ld4) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int seg = (int)(t / (KQ * EL)), r = (int)(t % (KQ * EL));
    const int el = r / KQ, kq = r % KQ;
    if (seg >= n_seg) return;
    const int s = seg % S;
    if (pair_ad[s] < 0) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = offsets[seg] + el; j < offsets[seg + 1]; j += EL) {
        const float a = 1.0f / (1.0f + expf(-sc[j]));
        if (kq == 0) att[j] = a;
        const float m = ids[j] > 0 ? a : 0.f;                                   // DIN.py:157: id 0 is the padding id
        const float4 ub = X[(size_t)j * 3 * KQ + kq];
        acc.x += m * ub.x; acc.y += m * ub.y; acc.z += m * ub.z; acc.w += m * ub.w;
    }
#pragma unroll
    for (int o = KQ; o < KQ * EL; o <<= 1) {
        acc.x += __shfl_xor(acc.x, o); acc.y += __shfl_xor(acc.y, o); acc.z += __shfl_xor(acc.z, o); acc.w += __shfl_xor(acc.w, o);
    }
    if (el == 0) x[(size_t)(seg / S) * ld4 + (size_t)s * KQ + kq] = acc;
}

// d sc_j = <dx[b,s], ub_j> [id_j > 0] att_j (1 - att_j)   (0 for the rows of the other slots)
template <int KQ>
__global__ __launch_bounds__(256) void att_bwd_scores_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ entry_off,
                                                            const int32_t* __restrict__ pair_ad, int nnz, const float4* __restrict__ dx,
                                                            int ld4, const float4* __restrict__ X, const float* __restrict__ att,
                                                            float* __restrict__ dsc) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int j = (int)(t / KQ), kq = (int)(t % KQ);
    if (j >= nnz) return;                                    // (KQ divides the wave: the lanes of an entry leave together)
    const int off = entry_off[j], b = off / ld4, s = (off - b * ld4) / KQ;
    float d = 0.f;
    const bool on = pair_ad[s] >= 0 && ids[j] > 0;
    if (on) {
        const float4 g = dx[(size_t)off + kq];
        const float4 ub = X[(size_t)j * 3 * KQ + kq];
        d = g.x * ub.x + g.y * ub.y + g.z * ub.z + g.w * ub.w;
    }
#pragma unroll
    for (int o = 1; o < KQ; o <<= 1) d += __shfl_xor(d, o);
    if (kq == 0) { const float a = att[j]; dsc[j] = on ? d * a * (1.0f - a) : 0.f; }
}

// per entry: dL/d ub_j = dX[j, 0:K] + dX[j, K:2K] + dx[b,s] att_j [id_j > 0]  -> dub[j, :], and where the table backward finds the
// entry's gradient row: goff[j] = the dub row for attention entries, the slot of dx otherwise (float4 offsets from dx's base)
template <int KQ>
__global__ __launch_bounds__(256) void att_bwd_combine_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ entry_off,
                                                             const int32_t* __restrict__ pair_ad, int nnz, const float4* __restrict__ dx,
                                                             int ld4, const float4* __restrict__ dX, const float* __restrict__ att,
                                                             float4* __restrict__ dub, int64_t dub_base4, int32_t* __restrict__ goff) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int j = (int)(t / KQ), kq = (int)(t % KQ);
    if (j >= nnz) return;
    const int off = entry_off[j], b = off / ld4, s = (off - b * ld4) / KQ;
    if (pair_ad[s] < 0) { if (kq == 0) goff[j] = off; return; }
    const float m = ids[j] > 0 ? att[j] : 0.f;
    const float4 g = dx[(size_t)off + kq];
    const float4 a = dX[(size_t)j * 3 * KQ + kq], c = dX[(size_t)j * 3 * KQ + KQ + kq];
    dub[(size_t)j * KQ + kq] = make_float4(a.x + c.x + m * g.x, a.y + c.y + m * g.y, a.z + c.z + m * g.z, a.w + c.w + m * g.w);
    if (kq == 0) goff[j] = (int32_t)(dub_base4 + (int64_t)j * KQ);
}

// dx[b, a(s)] += sum_j (dX[j, 2K:3K] - dX[j, K:2K]) over the entries of (b, s): the ad embedding's share of the attention inputs
template <int KQ, int EL>
__global__ __launch_bounds__(256) void att_bwd_dax_kernel(const int32_t* __restrict__ offsets, const int32_t* __restrict__ pair_ad,
                                                         int n_seg, int S, const float4* __restrict__ dX, float4* __restrict__ dx,
                                                         int ld4) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int seg = (int)(t / (KQ * EL)), r = (int)(t % (KQ * EL));
    const int el = r / KQ, kq = r % KQ;
    if (seg >= n_seg) return;
    const int ad = pair_ad[seg % S];
    if (ad < 0) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = offsets[seg] + el; j < offsets[seg + 1]; j += EL) {
        const float4 c = dX[(size_t)j * 3 * KQ + KQ + kq], a = dX[(size_t)j * 3 * KQ + 2 * KQ + kq];
        acc.x += a.x - c.x; acc.y += a.y - c.y; acc.z += a.z - c.z; acc.w += a.w - c.w;
    }
#pragma unroll
    for (int o = KQ; o < KQ * EL; o <<= 1) {
        acc.x += __shfl_xor(acc.x, o); acc.y += __shfl_xor(acc.y, o); acc.z += __shfl_xor(acc.z, o); acc.w += __shfl_xor(acc.w, o);
    }
    if (el == 0) {
        float4* p = dx + (size_t)(seg / S) * ld4 + (size_t)ad * KQ + kq;
        float4 v = *p;
        v.x += acc.x; v.y += acc.y; v.z += acc.z; v.w += acc.w;
        *p = v;
    }
}

#define DCTR_KQ_SWITCH(KQ, CALL)                                                       \
    switch (KQ) {                                                                      \
        case 1: { constexpr int Q = 1; constexpr int L = 16; CALL; } break;            \
        case 2: { constexpr int Q = 2; constexpr int L = 16; CALL; } break;            \
        case 4: { constexpr int Q = 4; constexpr int L = 16; CALL; } break;            \
        case 8: { constexpr int Q = 8; constexpr int L = 8; CALL; } break;             \
        case 16: { constexpr int Q = 16; constexpr int L = 4; CALL; } break;           \
        default: set_error("DIN attention pooling: embedding_size %d unsupported (4..64, K/4 a power of two)", 4 * (KQ)); \
                 return DCTR_ERR_UNSUPPORTED;                                          \
    }

int att_build_x(const float* emb, int64_t rows, int K, const int32_t* ids, const float* weights, const int32_t* entry_off,
                const int32_t* pair_ad, int nnz, const float* x, int ld, float* X, hipStream_t st, const LagView* lag) {
    if (nnz <= 0) return DCTR_OK;
    const int KQ = K / 4;
    const LagView LV = lag ? *lag : LagView{};
    DCTR_KQ_SWITCH(KQ, (void)L; (att_build_x_kernel<Q><<<ceil_div((int64_t)nnz * Q, 256), 256, 0, st>>>(
                                   reinterpret_cast<const float4*>(emb), rows, ids, weights, entry_off, pair_ad, nnz,
                                   reinterpret_cast<const float4*>(x), ld / 4, reinterpret_cast<float4*>(X), LV)));
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

int att_pool_fwd(const int32_t* offsets, const int32_t* ids, const int32_t* pair_ad, const float* sc, int n_seg, int S, int K,
                 const float* X, float* att, float* x, int ld, hipStream_t st) {
    if (n_seg <= 0) return DCTR_OK;
    const int KQ = K / 4;
    DCTR_KQ_SWITCH(KQ, (att_pool_fwd_kernel<Q, L><<<ceil_div((int64_t)n_seg * Q * L, 256), 256, 0, st>>>(
                           offsets, ids, pair_ad, sc, n_seg, S, reinterpret_cast<const float4*>(X), att, reinterpret_cast<float4*>(x), ld / 4)));
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

int att_bwd_scores(const int32_t* ids, const int32_t* entry_off, const int32_t* pair_ad, int nnz, int K, const float* dx, int ld,
                   const float* X, const float* att, float* dsc, hipStream_t st) {
    if (nnz <= 0) return DCTR_OK;
    const int KQ = K / 4;
    DCTR_KQ_SWITCH(KQ, (void)L; (att_bwd_scores_kernel<Q><<<ceil_div((int64_t)nnz * Q, 256), 256, 0, st>>>(
                                   ids, entry_off, pair_ad, nnz, reinterpret_cast<const float4*>(dx), ld / 4,
                                   reinterpret_cast<const float4*>(X), att, dsc)));
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

int att_bwd_combine(const int32_t* offsets, const int32_t* ids, const int32_t* entry_off, const int32_t* pair_ad, int nnz, int n_seg,
                    int S, int K, float* dx, int ld, const float* dX, const float* att, float* dub, int32_t* goff, hipStream_t st) {
    if (nnz <= 0) return DCTR_OK;
    const int KQ = K / 4;
    const int64_t dub_base4 = (dub - dx) / 4;            // dub lives behind dx in one allocation
    DCTR_REQUIRE(dub > dx && (dub - dx) % 4 == 0 && dub_base4 + (int64_t)nnz * KQ < 0x7FFFFFFFll, "attention gradient rows out of int32 reach");
    DCTR_KQ_SWITCH(KQ, (void)L; (att_bwd_combine_kernel<Q><<<ceil_div((int64_t)nnz * Q, 256), 256, 0, st>>>(
                                   ids, entry_off, pair_ad, nnz, reinterpret_cast<const float4*>(dx), ld / 4,
                                   reinterpret_cast<const float4*>(dX), att, reinterpret_cast<float4*>(dub), dub_base4, goff)));
    DCTR_LAUNCH_CHECK();
    DCTR_KQ_SWITCH(KQ, (att_bwd_dax_kernel<Q, L><<<ceil_div((int64_t)n_seg * Q * L, 256), 256, 0, st>>>(
                           offsets, pair_ad, n_seg, S, reinterpret_cast<const float4*>(dX), reinterpret_cast<float4*>(dx), ld / 4)));
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

}  // namespace dctr
