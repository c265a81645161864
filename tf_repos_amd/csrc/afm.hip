// K3/K4 for AFM (AFM.py:127-162): pairwise element-wise products of the F scaled embeddings, a one-hidden-layer
// attention MLP over the P = F(F-1)/2 pairs, softmax over the pairs, attention-weighted pooling, fc K -> 1.
//
//   pp[b,p,:]  = e[b,i_p,:] * e[b,j_p,:]                       AFM.py:134-138 (pairs lexicographic i<j)
//   ah         = relu(pp W_a + b_a)   [B*P, A]                 AFM.py:142-145   -> fp32 MFMA GEMM (gemm.hip)
//   s          = ah w_o + b_o         [B*P]                    AFM.py:147
//   att        = softmax_p(s)         [B,P]                    AFM.py:151  (+ dropout[0] in TRAIN, :152-153)
//   y_emb[b,:] = sum_p att[b,p] pp[b,p,:]   (+ dropout[1])     AFM.py:156-158
//   y_deep     = y_emb w_d + b_d                               AFM.py:160-162
// This first version materialises pp and ah in HBM (B*P*(K+A) floats: 3.1 GB each at B=4096, K=A=256 -- fits the
// 288 GB part with room to spare) so that the attention layer is one large GEMM; the flash-style fusion that keeps
// pp/ah on chip is listed as next work in DESIGN.md.
#include "common.h"
#include "engine.h"
#include "ops.h"

namespace dctr {

__device__ __forceinline__ float wsum64(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float wmax64(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

// pp4[(b*P + p)*KQ + kq] = e4[b,i_p,kq] * e4[b,j_p,kq]: blockIdx.y = example, a thread writes PAIR_FWD_U float4 pieces of its
// example's slice (32-bit index arithmetic, the loads of all pieces issued before the first store)
constexpr int PAIR_FWD_U = 4;
template <bool NT>
__global__ __launch_bounds__(256) void afm_pair_fwd_kernel(const float4* __restrict__ e, int e_ld4, const int16_t* __restrict__ pi,
                                                          const int16_t* __restrict__ pj, int B, int P, int KQ,
                                                          float4* __restrict__ pp) {
    const int b = blockIdx.y, n = P * KQ;
    const float4* eb = e + (size_t)b * e_ld4;
    float4* out = pp + (size_t)b * n;
    const int x0 = blockIdx.x * (256 * PAIR_FWD_U) + threadIdx.x;
    float4 a[PAIR_FWD_U], c[PAIR_FWD_U];
#pragma unroll
    for (int u = 0; u < PAIR_FWD_U; ++u) {
        const int x = min(x0 + u * 256, n - 1);
        const int p = x / KQ, kq = x - p * KQ;
        a[u] = eb[pi[p] * KQ + kq];
        c[u] = eb[pj[p] * KQ + kq];
    }
#pragma unroll
    for (int u = 0; u < PAIR_FWD_U; ++u) {
        const int x = x0 + u * 256;
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f r = {a[u].x * c[u].x, a[u].y * c[u].y, a[u].z * c[u].z, a[u].w * c[u].w};
        if (x < n) {
            if (NT) __builtin_nontemporal_store(r, reinterpret_cast<v4f*>(out + x));
            else *reinterpret_cast<v4f*>(out + x) = r;
        }
    }
}

// one block per example: softmax over the P scores, attention dropout, pooling over the pairs, y_emb dropout
// (ee != nullptr: the pair products are rebuilt from the example's embeddings, staged in LDS behind the weights -- 40 KB read per
//  example instead of its 759 KB slice of the [B P, K] pair tensor, K = 256)
template <int NT>
__global__ __launch_bounds__(NT) void afm_pool_fwd_kernel(const float* __restrict__ sc, const float* __restrict__ sc1, const float* __restrict__ sc_bias,
                                                         const float* __restrict__ pp, int P, int K,
                                                          float keep_att, float keep_emb, const uint64_t* __restrict__ seed_ptr,
                                                          int train, float* __restrict__ att, float* __restrict__ yemb,
                                                          const float4* __restrict__ ee, int e_ld4, int F, const int16_t* __restrict__ pi,
                                                          const int16_t* __restrict__ pj, int b0) {
    extern __shared__ __attribute__((aligned(16))) float sm[];        // [P] attention weights (after dropout) | [F][K] embeddings (ee)
    __shared__ float red[NT / 64];
    const int b = b0 + blockIdx.x, t = threadIdx.x;
    float4* es = reinterpret_cast<float4*>(sm + ((P + 3) & ~3));
    if (ee != nullptr)
        for (int x = t; x < F * (K >> 2); x += NT) es[x] = ee[(size_t)b * e_ld4 + x];
    // the scores: whole (sc), or the two column-slab parts of the product's epilogue + the bias (parts first: the order is fixed)
    const float sbias = sc_bias != nullptr ? sc_bias[0] : 0.f;
    auto score = [&](int p) { const size_t i = (size_t)b * P + p; return sc1 != nullptr ? (sc[i] + sc1[i]) + sbias : sc[i] + sbias; };
    float m = -3.0e38f;
    for (int p = t; p < P; p += NT) m = fmaxf(m, score(p));
    m = wmax64(m);
    if ((t & 63) == 0) red[t >> 6] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) m = fmaxf(m, red[w]);
    __syncthreads();
    float z = 0.f;
    for (int p = t; p < P; p += NT) { const float ex = expf(score(p) - m); sm[p] = ex; z += ex; }
    z = wsum64(z);
    if ((t & 63) == 0) red[t >> 6] = z;
    __syncthreads();
    z = 0.f;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) z += red[w];
    const float inv = 1.0f / z;
    const uint64_t seed = (train && (keep_att < 1.f || keep_emb < 1.f)) ? *seed_ptr : 0ull;
    const uint64_t row0 = dropout_row0(seed_ptr);        // (global row of this rank's first example, common.h)
    for (int p = t; p < P; p += NT) {
        const float a = sm[p] * inv;
        att[(size_t)b * P + p] = a;                                            // softmax output (before dropout): kept for the backward
        sm[p] = (train && keep_att < 1.f) ? a * dropout_scale(seed ^ DCTR_DROPOUT_SITE_AFM_ATT, (row0 + (uint64_t)b) * P + p, keep_att) : a;
    }
    __syncthreads();
    // y_emb[k] = sum_p a'[p] pp[p,k]: thread = (pair slice, float4 piece), slices summed through LDS
    __shared__ float4 acc4[NT];
    const int KQ = K >> 2, q = t % KQ, slice = t / KQ, n_slices = NT / KQ;
    const float4* pp4 = reinterpret_cast<const float4*>(pp) + (size_t)b * P * KQ;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = slice; p < P; p += n_slices) {
        const float a = sm[p];
        float4 v;
        if (ee != nullptr) {
            const float4 x = es[pi[p] * KQ + q], y = es[pj[p] * KQ + q];
            v = make_float4(x.x * y.x, x.y * y.y, x.z * y.z, x.w * y.w);
        } else {
            v = pp4[(size_t)p * KQ + q];
        }
        acc.x += a * v.x; acc.y += a * v.y; acc.z += a * v.z; acc.w += a * v.w;
    }
    acc4[t] = acc;
    __syncthreads();
    if (t < K) {
        const float* accf = reinterpret_cast<const float*>(acc4);
        float y = 0.f;
        for (int sl = 0; sl < n_slices; ++sl) y += accf[(size_t)sl * K + t];
        if (train && keep_emb < 1.f) y *= dropout_scale(seed ^ DCTR_DROPOUT_SITE_AFM_YEMB, (row0 + (uint64_t)b) * K + t, keep_emb);
        yemb[(size_t)b * K + t] = y;
    }
}

typedef float afm_f32x4 __attribute__((ext_vector_type(4)));
// The pooling as a product on the matrix cores, one WAVE per example: y_emb[k] = sum_{i<j} a'[pair(i,j)] e_i[k] e_j[k]
// = sum_i e_i[k] Z[i][k] with Z = U E, U the [F, F] upper-triangular matrix of the (dropped-out) attention weights.  The wave
// builds U in LDS while it normalises the softmax (12 KB per wave at F = 39), then per 64 columns of K: 4 F/16 x F/4 MFMAs of
// 16x16x4 with A fragments from U and B fragments float4 loads of e (element e of lane c is column 64 tq + 4 c + e: the product's
// column order is free), the row-wise products with e and the sum over the rows on the accumulators.  576 MFMAs per example at
// F = 39, K = 256, 40 KB of embeddings read once; the per-pair loop above walks 741 x 64 LDS float4 pairs.
template <int K, int TF>
__global__ __launch_bounds__(256) void afm_pool_fwd_mfma_kernel(const float* __restrict__ sc, const float* __restrict__ sc1, const float* __restrict__ sc_bias,
                                                               int P, int F, float keep_att, float keep_emb, const uint64_t* __restrict__ seed_ptr,
                                                               int train, float* __restrict__ att, float* __restrict__ yemb,
                                                               const float4* __restrict__ ee, int e_ld4, const int16_t* __restrict__ pi,
                                                               const int16_t* __restrict__ pj, int b0, int n) {
    constexpr int KQ = K / 4, NF = 16 * TF, LDU = NF + 1;
    extern __shared__ __attribute__((aligned(16))) float pf_sm[];        // per wave: U [NF][LDU]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;
    const int bi = blockIdx.x * 4 + wave;
    if (bi >= n) return;
    const int b = b0 + bi;
    float* U = pf_sm + (size_t)wave * NF * LDU;
    for (int x = lane; x < NF * LDU; x += 64) U[x] = 0.f;
    const float sbias = sc_bias != nullptr ? sc_bias[0] : 0.f;
    auto score = [&](int p) { const size_t i = (size_t)b * P + p; return sc1 != nullptr ? (sc[i] + sc1[i]) + sbias : sc[i] + sbias; };
    float m = -3.0e38f;
    for (int p = lane; p < P; p += 64) m = fmaxf(m, score(p));
    m = wmax64(m);
    float z = 0.f;
    for (int p = lane; p < P; p += 64) z += expf(score(p) - m);
    z = wsum64(z);
    const float inv = 1.0f / z;
    const uint64_t seed = (train && (keep_att < 1.f || keep_emb < 1.f)) ? *seed_ptr : 0ull;
    const uint64_t row0 = dropout_row0(seed_ptr);        // (global row of this rank's first example, common.h)
    for (int p = lane; p < P; p += 64) {
        const float a = expf(score(p) - m) * inv;
        att[(size_t)b * P + p] = a;                                            // softmax output (before dropout): kept for the backward
        U[pi[p] * LDU + pj[p]] = (train && keep_att < 1.f) ? a * dropout_scale(seed ^ DCTR_DROPOUT_SITE_AFM_ATT, (row0 + (uint64_t)b) * P + p, keep_att) : a;
    }
    const float4* eb = ee + (size_t)b * e_ld4;
    for (int tq = 0; tq < K / 64; ++tq) {
        afm_f32x4 acc[TF][4];
#pragma unroll
        for (int i = 0; i < TF; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][e] = afm_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int step = 0; step < 4 * TF; ++step) {
            const int f = 4 * step + q;
            const float4 bv = f < F ? eb[f * KQ + 16 * tq + c] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float bs[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int i = 0; i < TF; ++i) {
                const float av = U[(16 * i + c) * LDU + f];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bs[e], acc[i][e], 0, 0, 0);
            }
        }
        // register r of lane (c, q) in (i, e): Z[row 16 i + 4 q + r][column 64 tq + 4 c + e]
        float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < TF; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * i + 4 * q + r;
                if (row < F) {
                    const float4 ev = eb[row * KQ + 16 * tq + c];
                    y.x += ev.x * acc[i][0][r]; y.y += ev.y * acc[i][1][r]; y.z += ev.z * acc[i][2][r]; y.w += ev.w * acc[i][3][r];
                }
            }
        y.x += __shfl_xor(y.x, 16); y.y += __shfl_xor(y.y, 16); y.z += __shfl_xor(y.z, 16); y.w += __shfl_xor(y.w, 16);
        y.x += __shfl_xor(y.x, 32); y.y += __shfl_xor(y.y, 32); y.z += __shfl_xor(y.z, 32); y.w += __shfl_xor(y.w, 32);
        if (q == 0) {
            const int k = 64 * tq + 4 * c;
            if (train && keep_emb < 1.f) {
                y.x *= dropout_scale(seed ^ DCTR_DROPOUT_SITE_AFM_YEMB, (row0 + (uint64_t)b) * K + k + 0, keep_emb);
                y.y *= dropout_scale(seed ^ DCTR_DROPOUT_SITE_AFM_YEMB, (row0 + (uint64_t)b) * K + k + 1, keep_emb);
                y.z *= dropout_scale(seed ^ DCTR_DROPOUT_SITE_AFM_YEMB, (row0 + (uint64_t)b) * K + k + 2, keep_emb);
                y.w *= dropout_scale(seed ^ DCTR_DROPOUT_SITE_AFM_YEMB, (row0 + (uint64_t)b) * K + k + 3, keep_emb);
            }
            *reinterpret_cast<float4*>(yemb + (size_t)b * K + k) = y;
        }
    }
}

// backward of dropout[1] -> pooling -> dropout[0] -> softmax.  in: dy [B, dy_ld] = d y_emb (post-dropout); out: dsc [B,P], the
// post-dropout attention a' [B,P] and, in place of dy, the pre-dropout d y_emb.  d pp = a' (x) d y_emb is NOT materialised: the
// pair backward below forms it from these two.
template <int NT>
__global__ __launch_bounds__(NT) void afm_pool_bwd_kernel(float* __restrict__ dy, int dy_ld, const float* __restrict__ pp,
                                                          const float* __restrict__ att, int P, int K, float keep_att, float keep_emb,
                                                          const uint64_t* __restrict__ seed_ptr, float* __restrict__ dsc,
                                                          float* __restrict__ att_drop, const float4* __restrict__ ee, int e_ld4, int F,
                                                          const int16_t* __restrict__ pi, const int16_t* __restrict__ pj, int b0,
                                                          float* __restrict__ dsum) {
    extern __shared__ __attribute__((aligned(16))) float sm[];        // [K] dyemb (pre-dropout gradient) | [P] da | [F][K] embeddings (ee)
    __shared__ float red[NT / 64];
    float* dye = sm;
    float* da = sm + K;
    const int b = b0 + blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float4* es = reinterpret_cast<float4*>(sm + ((K + P + 3) & ~3));
    if (ee != nullptr)
        for (int x = t; x < F * (K >> 2); x += NT) es[x] = ee[(size_t)b * e_ld4 + x];
    const uint64_t seed = (keep_att < 1.f || keep_emb < 1.f) ? *seed_ptr : 0ull;
    const uint64_t row0 = dropout_row0(seed_ptr);        // (global row of this rank's first example, common.h)
    for (int k = t; k < K; k += NT) {
        float g = dy[(size_t)b * dy_ld + k];
        if (keep_emb < 1.f) g *= dropout_scale(seed ^ DCTR_DROPOUT_SITE_AFM_YEMB, (row0 + (uint64_t)b) * K + k, keep_emb);
        dye[k] = g;
        dy[(size_t)b * dy_ld + k] = g;
    }
    __syncthreads();
    // thread = (pair, float4 piece): d a'[p] = <d y_emb, pp[p,:]>, reduced over the KQ neighbouring lanes of the pair
    const int KQ = K >> 2, q = t % KQ;
    const float4* pp4 = reinterpret_cast<const float4*>(pp) + (size_t)b * P * KQ;
    const float4 d4 = reinterpret_cast<const float4*>(dye)[q];
    const float* ab = att + (size_t)b * P;
    float part = 0.f;                                          // sum_q att[q] * da[q]
    // (PB_U rows in flight per thread: one load per trip leaves a lone block per CU -- a small batch -- waiting out every latency)
    constexpr int PB_U = 4;
    for (int i0 = t; i0 < P * KQ; i0 += NT * PB_U) {
        float4 v[PB_U];
#pragma unroll
        for (int u = 0; u < PB_U; ++u) {
            const int i = min(i0 + u * NT, P * KQ - 1), p = i / KQ;
            if (ee != nullptr) {
                const float4 x = es[pi[p] * KQ + q], y = es[pj[p] * KQ + q];
                v[u] = make_float4(x.x * y.x, x.y * y.y, x.z * y.z, x.w * y.w);
            } else {
                v[u] = pp4[i];
            }
        }
#pragma unroll
        for (int u = 0; u < PB_U; ++u) {
            const int i = i0 + u * NT, p = min(i, P * KQ - 1) / KQ;
            float s = d4.x * v[u].x + d4.y * v[u].y + d4.z * v[u].z + d4.w * v[u].w;
            for (int o = 1; o < KQ; o <<= 1) s += __shfl_xor(s, o);
            if (q == 0 && i < P * KQ) {
                const float msk = keep_att < 1.f ? dropout_scale(seed ^ DCTR_DROPOUT_SITE_AFM_ATT, (row0 + (uint64_t)b) * P + p, keep_att) : 1.f;
                const float d = s * msk, a = ab[p];                // d att[p]
                da[p] = d;
                att_drop[(size_t)b * P + p] = a * msk;
                part += a * d;
            }
        }
    }
    part = wsum64(part);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) tot += red[w];
    float ds = 0.f;
    for (int p = t; p < P; p += NT) {
        const float d = ab[p] * (da[p] - tot);      // softmax backward
        dsc[(size_t)b * P + p] = d;
        ds += d;
    }
    if (dsum != nullptr) {                          // the example's share of attention_out's bias gradient (= sum of d score; ~0 by construction)
        __syncthreads();
        ds = wsum64(ds);
        if (lane == 0) red[wave] = ds;
        __syncthreads();
        if (t == 0) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) a += red[w];
            dsum[b] = a;
        }
    }
}

// The same from the example's EMBEDDINGS on the matrix cores: d a'[pair(i,j)] = <d y_emb, e_i . e_j> is entry (i, j) of
// G = (E diag(d y_emb)) E^T, an [F, F] product over K per example -- 384 MFMAs of 16x16x4 at F = 39, K = 256, against reading the
// example's 759 KB slice of the pair tensor (0.66 ms over B = 4096; this one 164 MB of embeddings in all).  One WAVE per example:
// lane (c, q) loads a float4 of field 16 t + c at k = 16 g + 4 q .. + 3 for each field tile t -- the same registers are the B
// fragments (E^T) and, scaled by d y_emb, the A fragments of the group's four MFMA steps; only tiles on or above the diagonal
// are formed.  Softmax / dropout backward on the accumulators, the example-wide sum by wave shuffles.
template <int K, int TF>
__global__ __launch_bounds__(256) void afm_pool_bwd_mfma_kernel(float* __restrict__ dy, int dy_ld, const float* __restrict__ att, int P, int F,
                                                               float keep_att, float keep_emb, const uint64_t* __restrict__ seed_ptr,
                                                               float* __restrict__ dsc, float* __restrict__ att_drop,
                                                               const float4* __restrict__ ee, int e_ld4, int b0, int n, float* __restrict__ dsum) {
    constexpr int KG = K / 16, KQ = K / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;
    const int bi = blockIdx.x * 4 + wave;
    if (bi >= n) return;
    const int b = b0 + bi;
    const uint64_t seed = (keep_att < 1.f || keep_emb < 1.f) ? *seed_ptr : 0ull;
    const uint64_t row0 = dropout_row0(seed_ptr);        // (global row of this rank's first example, common.h)
    // d y_emb before its dropout, this lane's 4 k of every group; written back in place (the pair backward reads it) by the c == 0 lanes
    float4 dk[KG];
    float* dyb = dy + (size_t)b * dy_ld;
#pragma unroll
    for (int g = 0; g < KG; ++g) {
        const int k = 16 * g + 4 * q;
        float4 v = *reinterpret_cast<const float4*>(dyb + k);
        if (keep_emb < 1.f) {
            v.x *= dropout_scale(seed ^ DCTR_DROPOUT_SITE_AFM_YEMB, (row0 + (uint64_t)b) * K + k + 0, keep_emb);
            v.y *= dropout_scale(seed ^ DCTR_DROPOUT_SITE_AFM_YEMB, (row0 + (uint64_t)b) * K + k + 1, keep_emb);
            v.z *= dropout_scale(seed ^ DCTR_DROPOUT_SITE_AFM_YEMB, (row0 + (uint64_t)b) * K + k + 2, keep_emb);
            v.w *= dropout_scale(seed ^ DCTR_DROPOUT_SITE_AFM_YEMB, (row0 + (uint64_t)b) * K + k + 3, keep_emb);
        }
        dk[g] = v;
    }
    afm_f32x4 acc[TF][TF];
#pragma unroll
    for (int i = 0; i < TF; ++i)
#pragma unroll
        for (int j = 0; j < TF; ++j) acc[i][j] = afm_f32x4{0.f, 0.f, 0.f, 0.f};
    const float4* eb = ee + (size_t)b * e_ld4;
    float4 ev[2][TF];
    auto load = [&](float4* d, int g) {
#pragma unroll
        for (int t = 0; t < TF; ++t) {
            const int f = 16 * t + c;
            d[t] = f < F ? eb[f * KQ + 4 * g + q] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    load(ev[0], 0);
#pragma unroll
    for (int g = 0; g < KG; ++g) {
        if (g + 1 < KG) load(ev[(g + 1) & 1], g + 1);
        const float4* e4 = ev[g & 1];
        const float dks[4] = {dk[g].x, dk[g].y, dk[g].z, dk[g].w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float bs[TF];
#pragma unroll
            for (int t = 0; t < TF; ++t) bs[t] = s == 0 ? e4[t].x : s == 1 ? e4[t].y : s == 2 ? e4[t].z : e4[t].w;
#pragma unroll
            for (int i = 0; i < TF; ++i)
#pragma unroll
                for (int j = i; j < TF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bs[i] * dks[s], bs[j], acc[i][j], 0, 0, 0);
        }
    }
    if (c == 0) {
#pragma unroll
        for (int g = 0; g < KG; ++g) *reinterpret_cast<float4*>(dyb + 16 * g + 4 * q) = dk[g];
    }
    // register r of lane (c, q) in tile (i, j): row 16 i + 4 q + r, column 16 j + c
    const float* ab = att + (size_t)b * P;
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < TF; ++i)
#pragma unroll
        for (int j = i; j < TF; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * i + 4 * q + r, col = 16 * j + c;
                float d = 0.f;
                if (row < col && col < F) {
                    const int p = row * F - (row * (row + 1)) / 2 + (col - row - 1);
                    const float msk = keep_att < 1.f ? dropout_scale(seed ^ DCTR_DROPOUT_SITE_AFM_ATT, (row0 + (uint64_t)b) * P + p, keep_att) : 1.f;
                    const float a = ab[p];
                    d = acc[i][j][r] * msk;                         // d att[p]
                    att_drop[(size_t)b * P + p] = a * msk;
                    part += a * d;
                }
                acc[i][j][r] = d;
            }
    const float tot = wsum64(part);
    float ds = 0.f;
#pragma unroll
    for (int i = 0; i < TF; ++i)
#pragma unroll
        for (int j = i; j < TF; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * i + 4 * q + r, col = 16 * j + c;
                if (row < col && col < F) {
                    const int p = row * F - (row * (row + 1)) / 2 + (col - row - 1);
                    const float v = ab[p] * (acc[i][j][r] - tot);      // softmax backward
                    dsc[(size_t)b * P + p] = v;
                    ds += v;
                }
            }
    if (dsum != nullptr) {
        ds = wsum64(ds);
        if (lane == 0) dsum[b] = ds;
    }
}

// dE[b,i,:] = sum_{j != i} (a'[b,pair(i,j)] * dyemb[b,:] + g2[b,pair(i,j),:]) * e[b,j,:]      (pooling path + attention path)
__global__ __launch_bounds__(256) void afm_pair_bwd_kernel(const float* __restrict__ e, int e_ld, const float* __restrict__ att_drop,
                                                          const float* __restrict__ dye, int dye_ld, const float* __restrict__ g2, int F,
                                                          int K, int P, float* __restrict__ dE, int de_ld) {
    const int b = blockIdx.y;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= F * K) return;
    const int i = x / K, k = x - i * K;
    const float* eb = e + (size_t)b * e_ld;
    const float* ad = att_drop + (size_t)b * P;
    const float* c = g2 + (size_t)b * P * K;
    const float dk = dye[(size_t)b * dye_ld + k];
    float s = 0.f;
    for (int j = 0; j < i; ++j) {
        const int p = j * F - (j * (j + 1)) / 2 + (i - j - 1);
        s += (ad[p] * dk + c[(size_t)p * K + k]) * eb[j * K + k];
    }
    for (int j = i + 1; j < F; ++j) {
        const int p = i * F - (i * (i + 1)) / 2 + (j - i - 1);
        s += (ad[p] * dk + c[(size_t)p * K + k]) * eb[j * K + k];
    }
    dE[(size_t)b * de_ld + x] = s;
}

// The same with float4 pieces and eight pairs in flight per thread (the loop above issues one 4-byte load per pair and waits for
// it: 1.1 TB/s on 6 GB at the reference's K = 256, B = 4096).  Thread = (field i, piece kq) of one example; every row of g2 is
// still read twice (once from each of its fields' side) -- a one-pass variant that folded both contributions into an LDS
// accumulator with ds_add_f32 (removed) measured 2x SLOWER than the loop above: LDS float atomics retire about one LANE per clock per CU
// (1.5 G lane-atomics at K = 256, B = 4096 = ~6 ms of LDS time however the banks are laid out).
template <int KQ>
__global__ __launch_bounds__(256) void afm_pair_bwd_v4_kernel(const float4* __restrict__ e, int e_ld4, const float* __restrict__ att_drop,
                                                             const float* __restrict__ dye, int dye_ld, const float4* __restrict__ g2,
                                                             int F, int P, float4* __restrict__ dE, int de_ld4) {
    const int b = blockIdx.y;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= F * KQ) return;
    const int i = x / KQ, kq = x - i * KQ;
    const float4* eb = e + (size_t)b * e_ld4;
    const float* ad = att_drop + (size_t)b * P;
    const float4* c = g2 + (size_t)b * P * KQ;
    const float4 dk = reinterpret_cast<const float4*>(dye + (size_t)b * dye_ld)[kq];
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    // the F-1 partner fields jj -> j = jj + (jj >= i), eight at a time; the tail is clamped and weighted 0 (no branch between the
    // loads: a predicated load in the middle of the batch makes the compiler wait for everything before it)
    constexpr int UN = 8;
    for (int j0 = 0; j0 < F - 1; j0 += UN) {
        float4 v[UN], ej[UN]; float a[UN], on[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int jj = j0 + u;
            on[u] = jj < F - 1 ? 1.f : 0.f;
            const int jc = jj < F - 1 ? jj : F - 2;
            const int j = jc + (jc >= i ? 1 : 0);
            const int lo = j < i ? j : i, hi = j < i ? i : j;
            const int p = lo * F - (lo * (lo + 1)) / 2 + (hi - lo - 1);
            v[u] = c[(size_t)p * KQ + kq];
            a[u] = ad[p];
            ej[u] = eb[j * KQ + kq];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            s.x += on[u] * (a[u] * dk.x + v[u].x) * ej[u].x; s.y += on[u] * (a[u] * dk.y + v[u].y) * ej[u].y;
            s.z += on[u] * (a[u] * dk.z + v[u].z) * ej[u].z; s.w += on[u] * (a[u] * dk.w + v[u].w) * ej[u].w;
        }
    }
    dE[(size_t)b * de_ld4 + x] = s;
}

// Every row of g2 read ONCE, without atomics: the pairs of one example are walked in round-robin-tournament order (the circle
// method: n = F rounded up to even, n - 1 rounds of n/2 pairs, no field twice in a round), so within a round every pair updates
// two accumulator rows nobody else touches -- plain LDS read-modify-write, a barrier between rounds.  Block = one example (4
// waves); a wave carries 64/KQ pairs at a time (lane = piece kq of its pair's rows), UN of them per round, and the rows of the
// NEXT round are in flight while this one is folded (the barrier waits for LDS only: a __syncthreads() would drain the loads).
// LDS: the accumulator and the example's embeddings, 2 * F * K floats (78 KB at F = 39, K = 256: two blocks per CU).
__device__ __forceinline__ bool rr_pair(int n, int F, int r, int m, int half, int& lo, int& hi) {
    int a = n - 1, c = r;
    if (m > 0) {
        a = r + m; if (a >= n - 1) a -= n - 1;
        c = r - m; if (c < 0) c += n - 1;
    }
    lo = a < c ? a : c;
    hi = a < c ? c : a;
    return m < half && hi < F;                     // (hi == F: the bye of an odd field count)
}

template <int KQ, int UN>
__global__ __launch_bounds__(256) void afm_pair_bwd_rr_kernel(const float4* __restrict__ e, int e_ld4, const float* __restrict__ att_drop,
                                                             const float* __restrict__ dye, int dye_ld, const float4* __restrict__ g2,
                                                             int F, int P, float4* __restrict__ dE, int de_ld4) {
    extern __shared__ float4 rr_sm[];
    float4* acc = rr_sm;
    float4* es = rr_sm + F * KQ;
    constexpr int S = 64 / KQ, G = 4 * S;
    const int b = blockIdx.x, t = threadIdx.x;
    const int g = (t >> 6) * S + (t & 63) / KQ, kq = (t & 63) % KQ;
    for (int x = t; x < F * KQ; x += 256) {
        es[x] = e[(size_t)b * e_ld4 + x];
        acc[x] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float4 dk = reinterpret_cast<const float4*>(dye + (size_t)b * dye_ld)[kq];
    const float* ad = att_drop + (size_t)b * P;
    const float4* c = g2 + (size_t)b * P * KQ;
    const int n = (F + 1) & ~1, half = n / 2;
    const int chunks = (half + G * UN - 1) / (G * UN), n_steps = (n - 1) * chunks;
    float4 v[UN], vn[UN];
    float a[UN], an[UN];
    auto fetch = [&](int s, float4* vv, float* aa) {
        const int r = s / chunks, m0 = (s - r * chunks) * G * UN + g;
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            int lo, hi;
            const bool ok = rr_pair(n, F, r, m0 + u * G, half, lo, hi);
            const int p = ok ? lo * F - (lo * (lo + 1)) / 2 + (hi - lo - 1) : 0;
            vv[u] = c[(size_t)p * KQ + kq];
            aa[u] = ad[p];
        }
    };
    fetch(0, v, a);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int s = 0; s < n_steps; ++s) {
        if (s + 1 < n_steps) fetch(s + 1, vn, an);
        const int r = s / chunks, m0 = (s - r * chunks) * G * UN + g;
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            int lo, hi;
            if (rr_pair(n, F, r, m0 + u * G, half, lo, hi)) {
                const float4 w = make_float4(v[u].x + a[u] * dk.x, v[u].y + a[u] * dk.y, v[u].z + a[u] * dk.z, v[u].w + a[u] * dk.w);
                const float4 el = es[lo * KQ + kq], eh = es[hi * KQ + kq];
                float4 al = acc[lo * KQ + kq], ah = acc[hi * KQ + kq];
                al.x += w.x * eh.x; al.y += w.y * eh.y; al.z += w.z * eh.z; al.w += w.w * eh.w;
                ah.x += w.x * el.x; ah.y += w.y * el.y; ah.z += w.z * el.z; ah.w += w.w * el.w;
                acc[lo * KQ + kq] = al;
                acc[hi * KQ + kq] = ah;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int u = 0; u < UN; ++u) { v[u] = vn[u]; a[u] = an[u]; }
    }
    for (int x = t; x < F * KQ; x += 256) dE[(size_t)b * de_ld4 + x] = acc[x];
}

template <int KQ, int UN>
int launch_pair_bwd_rr(const float* e, int e_ld, const float* att_drop, const float* dye, int dye_ld, const float* g2, int B, int F, int P,
                       float* dE, int de_ld, size_t lds, hipStream_t st) {
    auto kern = afm_pair_bwd_rr_kernel<KQ, UN>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    DCTR_HIP_CHECK(attr);
    kern<<<B, 256, lds, st>>>(reinterpret_cast<const float4*>(e), e_ld / 4, att_drop, dye, dye_ld, reinterpret_cast<const float4*>(g2), F, P,
                              reinterpret_cast<float4*>(dE), de_ld / 4);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

int afm_pair_bwd(const float* e, int e_ld, const float* att_drop, const float* dye, int dye_ld, const float* g2, const int16_t* pair_i,
                 const int16_t* pair_j, int B, int F, int K, int P, float* dE, int de_ld, hipStream_t st) {
    static const bool old = getenv("DCTR_AFM_PAIR_BWD_OLD") != nullptr;          // A/B knob: the one-load-per-pair kernel
    static const bool no_rr = getenv("DCTR_AFM_PAIR_BWD_TWICE") != nullptr;      // A/B knob: the two-reads kernel at every K
    const size_t lds_rr = (size_t)2 * F * K * sizeof(float);
    // (a block per example: below two blocks per CU the two-reads kernel, ten blocks per example at K = 256, is the faster one)
    if (!old && !no_rr && B >= 512 && F >= 2 && K >= 64 && lds_rr <= 160 * 1024 && e_ld % 4 == 0 && de_ld % 4 == 0 && dye_ld % 4 == 0) {
        switch (K / 4) {
            case 16: return launch_pair_bwd_rr<16, 2>(e, e_ld, att_drop, dye, dye_ld, g2, B, F, P, dE, de_ld, lds_rr, st);
            case 32: return launch_pair_bwd_rr<32, 3>(e, e_ld, att_drop, dye, dye_ld, g2, B, F, P, dE, de_ld, lds_rr, st);
            case 64: return launch_pair_bwd_rr<64, 5>(e, e_ld, att_drop, dye, dye_ld, g2, B, F, P, dE, de_ld, lds_rr, st);
            default: break;
        }
    }
    if (!old && F >= 2 && e_ld % 4 == 0 && de_ld % 4 == 0 && dye_ld % 4 == 0) {
        dim3 grid(ceil_div(F * (K / 4), 256), B);
#define DCTR_PB(Q) case Q: afm_pair_bwd_v4_kernel<Q><<<grid, 256, 0, st>>>(reinterpret_cast<const float4*>(e), e_ld / 4, att_drop, dye, dye_ld, \
                                     reinterpret_cast<const float4*>(g2), F, P, reinterpret_cast<float4*>(dE), de_ld / 4); DCTR_LAUNCH_CHECK(); return DCTR_OK;
        switch (K / 4) { DCTR_PB(1) DCTR_PB(2) DCTR_PB(4) DCTR_PB(8) DCTR_PB(16) DCTR_PB(32) DCTR_PB(64) default: break; }
#undef DCTR_PB
    }
    dim3 grid(ceil_div(F * K, 256), B);
    afm_pair_bwd_kernel<<<grid, 256, 0, st>>>(e, e_ld, att_drop, dye, dye_ld, g2, F, K, P, dE, de_ld);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

}  // namespace dctr

using namespace dctr;

// A/B knob DCTR_AFM_IN_PRODUCTS=0: the score dot, the ReLU-masked rank-one gradient of the last attention layer and attention_out's
// weight gradient as passes of their own (rowdot / out_layer_bwd) instead of inside the three products
// From this many pair rows on, a handle in split mode runs the attention layer's products on the tall split-precision kernels (gemm_ts.hip).
// Their blocks own a whole CU (512 registers per lane of a SIMD, 72-144 KB of LDS), so nothing of the step's other streams shares a CU with
// them: with only the forward and the input gradient on them (four waves each) the reference's B = 128 (95 k rows, 371 row tiles for
// 256 CUs) LOST 0.17 ms to that; with all three products, the eight-wave forward and no pair tensor it gains 0.3 of 0.9 ms, so the bound
// is the kernels' own (ts_takes: 65536 rows).
static int64_t afm_ts_min_rows() {
    static const int64_t v = getenv("DCTR_AFM_TS_MIN_ROWS") ? atoll(getenv("DCTR_AFM_TS_MIN_ROWS")) : 65536;     // A/B knob
    return v;
}
// ... and from this many multiply-adds per product (rows x K x A): the reference's own run.sh:18 point (B = 128, K = 256, A = 128: 3.1 G) is the
// smallest shape measured -- 0.58 -> 0.41 ms/step once the tall kernels run one block per row tile there (gemm_ts.hip: with looping blocks that
// hold every CU until the kernel ends it LOST, 0.73).  Below it the 256 partial slabs and the plane split are not known to be repaid.
static bool afm_ts_worth(int64_t rows, int K, int A) {
    static const double v = getenv("DCTR_AFM_TS_MIN_MACS") ? atof(getenv("DCTR_AFM_TS_MIN_MACS")) : 2.5e9;          // A/B knob
    return rows >= afm_ts_min_rows() && (double)rows * K * A >= v;
}
static bool afm_in_products() {
    static const bool off = [] { const char* v = getenv("DCTR_AFM_IN_PRODUCTS"); return v != nullptr && v[0] == '0'; }();
    return !off;
}
constexpr int AFM_MAX_CHUNKS = 8;
static int afm_chunks_wanted(int B, int P) {
    // OFF by default: measured at the reference point (B = 4096, K = A = 256) 2 / 4 / 8 chunks = 14.4 / 14.9 / 16.6 ms against 14.0
    // whole-batch -- the products stream their tall operand from HBM and every one of them slows down by about what the pass
    // running beside it takes (forward product 3.3 -> 5.6 ms per step under the other lane's elementwise passes).
    static const int forced = getenv("DCTR_AFM_CHUNKS") ? atoi(getenv("DCTR_AFM_CHUNKS")) : 1;     // A/B knob
    (void)P;
    return std::max(1, std::min(forced, std::min(B, AFM_MAX_CHUNKS)));
}

int afm_declare_params(dctr_engine* E) {
    auto add = engine_add_param;
    const dctr_config& c = E->cfg;
    const int nl = c.n_attention_layers;
    DCTR_REQUIRE(nl >= 1 && nl <= DCTR_MAX_LAYERS, "AFM: 1..%d attention layers (got %d)", DCTR_MAX_LAYERS, nl);
    for (int l = 0; l < nl; ++l) DCTR_REQUIRE(c.attention_layers[l] > 0, "AFM: attention layer widths must be > 0");
    E->A = c.attention_layers[nl - 1];          // what attention_out projects (AFM.py:147)
    E->keep_att = c.keep_prob[0] > 0.f ? c.keep_prob[0] : 1.f;
    E->keep_emb = c.keep_prob[1] > 0.f ? c.keep_prob[1] : 1.f;
    const int K = E->K, A = E->A;
    // the attention network runs fused over the pair rows when its shape allows (afm_fused.hip): the hidden layer [B*P, A] is
    // then never materialised, and the four attention parameters take their gradient from AFM_SLABS atomically filled slabs
    // (one hidden layer, the reference's default; the AFM.py:143-145 loop with more widths runs layer by layer)
    E->afm_fused = nl == 1 && getenv("DCTR_AFM_UNFUSED") == nullptr && afm_fused_supported(K, A);        // (the env knob is the A/B switch)
    int d = K;
    for (int l = 0; l < nl; ++l) {
        Fc fc;
        fc.in = d; fc.out = c.attention_layers[l];
        // (the unfused path runs in up to AFM_MAX_CHUNKS chunks of examples, each with a full set of weight-gradient slabs)
        const int nc_max = afm_chunks_wanted(E->MB, E->P);
        fc.splits = E->afm_fused ? AFM_SLABS : nc_max * choose_wgrad_splits(ceil_div(E->MB, nc_max) * E->P, fc.in, fc.out);
        // split-precision mode at a batch the tall products pay for: the gated weight gradient is gemm_ts.hip's, one slab per CU
        E->afm_ts_wgrad = nl == 1 && !E->afm_fused && nc_max == 1 && E->gemm_mode == 1 && afm_in_products() && ts_takes((int64_t)E->MB * E->P, fc.in, fc.out) &&
                          ws_takes((int64_t)E->MB * E->P, fc.out, fc.in) && afm_ts_worth((int64_t)E->MB * E->P, fc.in, fc.out);
        if (E->afm_ts_wgrad) fc.splits = TS_WGRAD_SLABS;
        char nm[64];
        snprintf(nm, sizeof(nm), "att_mlp%d/weights", l);
        fc.w = add(E, nm, {fc.in, fc.out}, false, fc.splits, 0.f);
        snprintf(nm, sizeof(nm), "att_mlp%d/biases", l);
        fc.b = add(E, nm, {fc.out}, false, fc.splits, 0.f);
        fc.last = fc.b;
        E->att_fc.push_back(fc);
        d = fc.out;
    }
    // attention_out's gradient slabs: one per ~512 pair rows (1024 at the reference's B = 4096; at B = 128 that many slabs made the
    // optimizer's slab sum, 1024 dependent strided reads per element, a 98 us kernel) -- or, when the last attention layer's weight
    // gradient also produces them (DR_BGATE_WGRAD, afm_backward), one per batch split of that product
    int ao_rows = 128;
    while (ao_rows < 1024 && (int64_t)ao_rows * 512 < (int64_t)E->MB * E->P) ao_rows *= 2;
    E->ao_splits = ao_rows;
    E->afm_gate_slabs = nl == 1 && !E->afm_fused && afm_in_products() && ws_takes((int64_t)E->MB * E->P, A, K);
    const int ao = E->afm_fused ? AFM_SLABS : E->afm_gate_slabs ? E->att_fc[0].splits : E->ao_splits;
    E->att_splits = E->att_fc[0].splits;
    E->p_att_w = E->att_fc[0].w;
    E->p_att_b = E->att_fc[0].b;
    E->p_ao_w = add(E, "attention_out/weights", {A, 1}, false, ao, 0.f);
    E->p_ao_b = add(E, "attention_out/biases", {1}, false, ao, 0.f);
    E->p_out_w = add(E, "deep_out/weights", {K, 1}, false, E->out_splits, 0.f);
    E->p_out_b = add(E, "deep_out/biases", {1}, false, E->out_splits, 0.f);
    return DCTR_OK;
}

template <typename T>
static int dm(T** p, size_t n) {
    DCTR_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(n, 4) * sizeof(T)));
    DCTR_HIP_CHECK(hipMemset(*p, 0, std::max<size_t>(n, 4) * sizeof(T)));
    return DCTR_OK;
}

int afm_alloc(dctr_engine* E) {
    const size_t MB = E->MB, P = E->P, K = E->K, A = E->A;
    DCTR_TRY(dm(&E->pairp, MB * P * K));
    DCTR_TRY(dm(&E->dpairp2, MB * P * K));
    if (!E->afm_fused) {
        for (const Fc& fc : E->att_fc) {
            float *a = nullptr, *da = nullptr;
            DCTR_TRY(dm(&a, MB * P * (size_t)fc.out));
            DCTR_TRY(dm(&da, MB * P * (size_t)fc.out));
            E->ahs.push_back(a);
            E->dahs.push_back(da);
        }
        E->ah = E->ahs.back();
        E->dah = E->dahs.back();
        DCTR_HIP_CHECK(hipStreamCreateWithFlags(&E->s_afm, hipStreamNonBlocking));
        DCTR_TRY(dm(&E->sc_parts, 2 * MB * P));
        // split-precision mode: the attention weight's bf16 planes for the tall products (gemm_ts.hip), [forward | input gradient]
        DCTR_HIP_CHECK(hipMalloc(&E->ts_planes, 2 * ts_plane_bytes(256, 256)));
        // ... and the sign words of the attention layer's output (32 bytes per pair row), what its input gradient reads instead of the rows
        DCTR_HIP_CHECK(hipMalloc(&E->ts_sign, ts_sign_bytes((int64_t)MB * P)));
    }
    (void)A;
    DCTR_TRY(dm(&E->sc, MB * P));
    DCTR_TRY(dm(&E->dsc, MB * P));
    DCTR_TRY(dm(&E->att, MB * P));
    DCTR_TRY(dm(&E->dE_buf, MB * E->D));
    std::vector<int16_t> pi, pj;
    for (int i = 0; i < E->F - 1; ++i)
        for (int j = i + 1; j < E->F; ++j) { pi.push_back((int16_t)i); pj.push_back((int16_t)j); }
    DCTR_TRY(dm(&E->pair_i, pi.size()));
    DCTR_TRY(dm(&E->pair_j, pj.size()));
    DCTR_HIP_CHECK(hipMemcpy(E->pair_i, pi.data(), pi.size() * 2, hipMemcpyHostToDevice));
    DCTR_HIP_CHECK(hipMemcpy(E->pair_j, pj.data(), pj.size() * 2, hipMemcpyHostToDevice));
    return DCTR_OK;
}

void afm_free(dctr_engine* E) {
    float* fl[] = {E->pairp, E->dpairp2, E->sc, E->dsc, E->att, E->dE_buf, E->sc_parts};
    for (float* p : fl) if (p) hipFree(p);
    for (float* p : E->ahs) hipFree(p);
    for (float* p : E->dahs) hipFree(p);
    if (E->ts_planes) hipFree(E->ts_planes);
    if (E->ts_sign) hipFree(E->ts_sign);
    if (E->pair_i) hipFree(E->pair_i);
    if (E->pair_j) hipFree(E->pair_j);
    if (E->s_afm) hipStreamDestroy(E->s_afm);
}

// out[s * stride] = sum of x over the s-th of `splits` equal ranges (attention_out's bias gradient = sum of d score, AFM.py:147-148)
__global__ __launch_bounds__(256) void vec_sum_partials_kernel(const float* __restrict__ x, int64_t n, int64_t per, float* __restrict__ out, int64_t stride) {
    __shared__ float red[4];
    const int64_t i0 = (int64_t)blockIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
    float s = 0.f;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = i + u * 256 < i1 ? x[i + u * 256] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    s = wsum64(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[(size_t)blockIdx.x * stride] = red[0] + red[1] + red[2] + red[3];
}

// The unfused path (K > 32: the reference's K = 256) CAN run the step's passes over the pair rows in chunks of examples on two
// lanes (streams): every pass is per example, so chunk c+1's elementwise passes (HBM-bound: pair products, score dot, pooling,
// ReLU mask, pair backward) would run under chunk c's products instead of between them (DCTR_AFM_CHUNKS=n; see afm_chunks_wanted).  Weight-gradient slabs are split
// between the chunks (n_part / chunks each), dropout masks are keyed by the example's global index.
static int afm_chunks(const dctr_engine* E, int B) {
    int nc = afm_chunks_wanted(B, E->P);
    if (E->afm_fused || E->s_afm == nullptr) return 1;
    auto divides = [&](int n) {
        if (E->params[E->p_ao_w].n_part % n || E->params[E->p_ao_b].n_part % n) return false;
        for (const Fc& fc : E->att_fc) if (fc.splits % n || E->params[fc.w].n_part % n || E->params[fc.b].n_part % n) return false;
        return true;
    };
    while (nc > 1 && (!divides(nc) || (nc - 1) * ceil_div(B, nc) >= B)) --nc;
    return nc;
}

// can this step run without the pair tensor?  Everything the forward product, the weight gradient and the two pooling kernels check, in one place
static bool afm_pairs_in_registers(dctr_engine* E, int n, const TsPairs* tp) {
    static const bool off = getenv("DCTR_AFM_PP") != nullptr && atoi(getenv("DCTR_AFM_PP")) == 1;                    // A/B knob
    static const bool pool_knobs = getenv("DCTR_AFM_POOL_FWD_LOOP") != nullptr || getenv("DCTR_AFM_POOL_BWD_PP") != nullptr;
    const int F = E->F, K = E->K, P = E->P;
    if (off || pool_knobs || !afm_in_products() || !E->afm_ts_wgrad || E->att_fc.size() != 1 || E->gemm_mode != 1 || E->ts_planes == nullptr || afm_chunks(E, n) != 1) return false;
    const Fc& fc = E->att_fc[0];
    const Param& w = E->params[fc.w];
    if (!afm_ts_worth((int64_t)n * P, fc.in, fc.out) || !ts_takes((int64_t)n * P, fc.in, fc.out) || !ts_pairs_ok(tp, (int64_t)n * P, K) || fc.in != K) return false;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!al16(E->ah) || !al16(E->dsc) || !al16(E->parts + w.part_off) || (w.padded & 3) || !al16(E->pp(fc.w)) || !al16(E->pp(fc.b)) || !al16(E->pp(E->p_ao_w))) return false;
    // forward pooling: the MFMA kernel (n >= 2048) or the LDS rebuild, both from e; backward pooling: the MFMA kernel
    const bool kf = (K == 64 || K == 128 || K == 256) && F <= 48 && E->e_ld % 4 == 0;
    const bool fwd_e = (n >= 2048 && kf) || ((size_t)(((P + 3) & ~3) + F * K) * sizeof(float) <= 128 * 1024 && E->e_ld % 4 == 0);
    return fwd_e && kf && E->Din_ld % 4 == 0;
}

static int afm_pool_fwd(dctr_engine* E, int b0, int n, bool train, hipStream_t st, int score_parts = 0) {
    const int F = E->F, K = E->K, P = E->P;
    static const bool no_mfma = getenv("DCTR_AFM_POOL_FWD_LOOP") != nullptr;       // A/B knob: the per-pair loop
    // (one wave per example is a chain of latencies: below two waves per SIMD of them -- B = 128: 0.585 -> 0.68 ms -- the 1024-thread loop)
    if (!no_mfma && n >= 2048 && F <= 48 && (K == 64 || K == 128 || K == 256) && E->e_ld % 4 == 0) {
        const float* s0 = score_parts > 0 ? E->sc_parts : E->sc;
        const float* s1 = score_parts > 1 ? E->sc_parts + (size_t)E->MB * P : nullptr;
        const float* sb = score_parts > 0 ? E->pp(E->p_ao_b) : nullptr;
        const int tf = ceil_div(F, 16);
        const size_t lds = (size_t)4 * (16 * tf) * (16 * tf + 1) * sizeof(float);
#define DCTR_PFM(K_, T_) afm_pool_fwd_mfma_kernel<K_, T_><<<ceil_div(n, 4), 256, lds, st>>>(s0, s1, sb, P, F, E->keep_att, E->keep_emb, &E->state->seed_t, \
            train ? 1 : 0, E->att, E->x_in, reinterpret_cast<const float4*>(E->e), E->e_ld / 4, E->pair_i, E->pair_j, b0, n)
#define DCTR_PFK(K_) if (tf == 1) DCTR_PFM(K_, 1); else if (tf == 2) DCTR_PFM(K_, 2); else DCTR_PFM(K_, 3)
        if (K == 64) { DCTR_PFK(64); } else if (K == 128) { DCTR_PFK(128); } else { DCTR_PFK(256); }
#undef DCTR_PFK
#undef DCTR_PFM
        DCTR_LAUNCH_CHECK();
        return DCTR_OK;
    }
    // the pooling rebuilds the pair products from the example's embeddings when they fit LDS beside its working set
    const size_t lds_pp = (size_t)P * sizeof(float), lds_e = (size_t)(((P + 3) & ~3) + F * K) * sizeof(float);
    const bool from_e = lds_e <= 128 * 1024 && E->e_ld % 4 == 0;
    // (a small batch is fewer blocks than CUs: 16 waves per example then)
    const bool wide = n < 512 && K >= 64;
    auto kern = wide ? afm_pool_fwd_kernel<1024> : afm_pool_fwd_kernel<256>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(afm_pool_fwd_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    static const hipError_t attr_w = hipFuncSetAttribute(reinterpret_cast<const void*>(afm_pool_fwd_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    DCTR_HIP_CHECK(attr);
    DCTR_HIP_CHECK(attr_w);
    const float* s0 = score_parts > 0 ? E->sc_parts : E->sc;
    const float* s1 = score_parts > 1 ? E->sc_parts + (size_t)E->MB * P : nullptr;
    const float* sb = score_parts > 0 ? E->pp(E->p_ao_b) : nullptr;
    kern<<<n, wide ? 1024 : 256, from_e ? lds_e : lds_pp, st>>>(s0, s1, sb, E->pairp, P, K, E->keep_att, E->keep_emb, &E->state->seed_t,
                                                                 train ? 1 : 0, E->att, E->x_in,
                                                                 from_e ? reinterpret_cast<const float4*>(E->e) : nullptr, E->e_ld / 4, F, E->pair_i, E->pair_j, b0);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

// after the gather (mode RAW + linear) has filled E->e / E->yw
int afm_forward(dctr_engine* E, int B, bool train, hipStream_t st) {
    const int K = E->K, P = E->P, A = E->A, KQ = K / 4;
    if (E->afm_fused) {
        afm_pair_fwd_kernel<false><<<dim3(ceil_div(P * KQ, 256 * PAIR_FWD_U), B), 256, 0, st>>>(reinterpret_cast<const float4*>(E->e), E->e_ld / 4, E->pair_i, E->pair_j, B,
                                                                P, KQ, reinterpret_cast<float4*>(E->pairp));
        DCTR_LAUNCH_CHECK();
        // scores straight from the pair products (the hidden layer never leaves the registers; the backward recomputes it)
        DCTR_TRY(afm_att_fwd(E->pairp, E->pp(E->p_att_w), E->pp(E->p_att_b), E->pp(E->p_ao_w), E->pp(E->p_ao_b), (int64_t)B * P, K, A, E->sc, st));
        return afm_pool_fwd(E, 0, B, train, st);
    }
    const int nc = afm_chunks(E, B), bc = ceil_div(B, nc);
    if (nc > 1) DCTR_TRY(fork(E, st, E->s_afm));
    for (int c = 0; c < nc; ++c) {
        hipStream_t s = (c & 1) ? E->s_afm : st;
        const int b0 = c * bc, n = std::min(bc, B - b0);
        const size_t r0 = (size_t)b0 * P;
        // nontemporal stores once the pair tensor cannot stay in the caches anyway (3.1 GB at the reference point: 13.52 -> 13.42 ms/step)
        static const bool nt_off = getenv("DCTR_AFM_PAIR_FWD_NT") != nullptr && atoi(getenv("DCTR_AFM_PAIR_FWD_NT")) == 0;       // A/B knob
        const bool nt = !nt_off && (size_t)n * P * K * sizeof(float) > ((size_t)512 << 20);
        auto materialise = [&]() -> int {
            (nt ? afm_pair_fwd_kernel<true> : afm_pair_fwd_kernel<false>)<<<dim3(ceil_div(P * KQ, 256 * PAIR_FWD_U), n), 256, 0, s>>>(reinterpret_cast<const float4*>(E->e + (size_t)b0 * E->e_ld), E->e_ld / 4, E->pair_i,
                                                                   E->pair_j, n, P, KQ, reinterpret_cast<float4*>(E->pairp + r0 * K));
            DCTR_LAUNCH_CHECK();
            return DCTR_OK;
        };
        // The pair tensor [B P, K] (3.1 GB at the reference point) is NOT written when everything that reads it can form its rows from the
        // gathered embeddings instead: the attention product and its weight gradient (gemm_ts.hip with TsPairs: e_i . e_j in the registers,
        // bit-identical) and the two pooling kernels (which already work from the embeddings at this shape).  DCTR_AFM_PP=1: written as before.
        const TsPairs tp{E->e + (size_t)b0 * E->e_ld, E->e_ld, n, E->pair_i, E->pair_j, P};
        bool gen = afm_pairs_in_registers(E, n, &tp);
        E->afm_pp_skipped = false;
        E->ts_sign_ready = false;
        E->afm_ah_skipped = false;
        if (!gen) DCTR_TRY(materialise());
        const float* x = E->pairp + r0 * K;
        int score_parts = 0;
        for (size_t l = 0; l < E->att_fc.size(); ++l) {     // relu(x W_l + b_l) over the chunk's pair rows
            const Fc& fc = E->att_fc[l];
            float* y = E->ahs[l] + r0 * fc.out;
            bool done = false;
            // the last layer's product also takes the score dot <ah[row, :], w_o> (AFM.py:147) from its accumulators, one part per
            // 128-column slab, when it is the tall-operand kernel with at most two slabs (a two-term sum is order-free)
            const bool in_products = l + 1 == E->att_fc.size() && fc.out <= 256 && afm_in_products();
            // split-precision mode: the tall product on the bf16 matrix pipe (gemm_ts.hip), the whole score from one wave's accumulators
            if (in_products && nc == 1 && E->gemm_mode == 1 && E->ts_planes != nullptr && afm_ts_worth((int64_t)n * P, fc.in, fc.out)) {
                // (one launch writes the weight's planes for this product and for the backward's gated input gradient)
                const bool both = E->afm_gate_slabs && ts_takes((int64_t)n * P, fc.out, fc.in);
                if (both) DCTR_TRY(ts_prepare(E->pp(fc.w), fc.in, fc.out, E->pp(E->p_ao_w), E->ts_planes, static_cast<char*>(E->ts_planes) + ts_plane_bytes(256, 256), s));
                // With the rows generated and the sign words written, NOBODY reads the layer's output ah [B P, A] (3.1 GB): the input gradient takes
                // its gate from the sign words, the weight gradient too (and the score-weight gradient from its own product) -- it is not stored.
                // DCTR_AFM_AH=1: stored as before (and read by the weight gradient).
                static const bool keep_ah = getenv("DCTR_AFM_AH") != nullptr && atoi(getenv("DCTR_AFM_AH")) == 1;           // A/B knob
                const bool no_ah = gen && both && !keep_ah && E->ts_sign != nullptr && ts_bits_enabled();
                DCTR_TRY(ts_fc_fwd_dot(x, fc.in, E->pp(fc.w), E->pp(fc.b), no_ah ? nullptr : y, fc.out, (int64_t)n * P, fc.in, fc.out, E->pp(E->p_ao_w), E->sc_parts + r0,
                                       E->ts_planes, !both, s, &done, gen ? &tp : nullptr, E->ts_sign ? static_cast<char*>(E->ts_sign) + ts_sign_bytes(r0) : nullptr));
                E->afm_ah_skipped = done && no_ah;
                E->ts_sign_ready = done && E->ts_sign != nullptr;
                if (gen && !done) {                     // (not taken after all: the rows are written and the product below reads them)
                    gen = false;
                    DCTR_TRY(materialise());
                }
                E->afm_pp_skipped = gen;
                E->ts_dgr_ready = both && done;
                if (done) score_parts = 1;
            }
            if (!done && in_products)
                DCTR_TRY(ws_fc_fwd_dot(x, fc.in, E->pp(fc.w), E->pp(fc.b), y, fc.out, n * P, fc.in, fc.out, 1, E->pp(E->p_ao_w), E->sc_parts + r0,
                                       (int64_t)E->MB * P, &score_parts, s, &done));
            if (!done) DCTR_TRY(fc_fwd(x, fc.in, E->pp(fc.w), E->pp(fc.b), y, fc.out, n * P, fc.in, fc.out, 1, 1.f, nullptr, 0, s));
            x = y;
        }
        if (score_parts == 0) DCTR_TRY(rowdot(E->ah + r0 * A, A, E->pp(E->p_ao_w), E->pp(E->p_ao_b), n * P, A, E->sc + r0, 0, s));
        DCTR_TRY(afm_pool_fwd(E, b0, n, train, s, score_parts));
    }
    if (nc > 1) DCTR_TRY(fork(E, E->s_afm, st));
    return DCTR_OK;      // the fc(K -> 1) output layer is fused into the head kernel
}

static int afm_pool_bwd(dctr_engine* E, int b0, int n, hipStream_t st) {
    const int F = E->F, K = E->K, P = E->P;
    static const bool no_mfma = getenv("DCTR_AFM_POOL_BWD_PP") != nullptr;       // A/B knob: the pass over the pair tensor
    if (!no_mfma && F <= 48 && (K == 64 || K == 128 || K == 256) && E->e_ld % 4 == 0 && E->Din_ld % 4 == 0) {
        const int tf = ceil_div(F, 16);
#define DCTR_PBM(K_, T_) afm_pool_bwd_mfma_kernel<K_, T_><<<ceil_div(n, 4), 256, 0, st>>>(E->dx_in, E->Din_ld, E->att, P, F, E->keep_att, E->keep_emb, \
            &E->state->seed_t, E->dsc, E->sc, reinterpret_cast<const float4*>(E->e), E->e_ld / 4, b0, n, E->sc_parts)
#define DCTR_PBK(K_) if (tf == 1) DCTR_PBM(K_, 1); else if (tf == 2) DCTR_PBM(K_, 2); else DCTR_PBM(K_, 3)
        if (K == 64) { DCTR_PBK(64); } else if (K == 128) { DCTR_PBK(128); } else { DCTR_PBK(256); }
#undef DCTR_PBK
#undef DCTR_PBM
        DCTR_LAUNCH_CHECK();
        return DCTR_OK;
    }
    const size_t lds_pp = (size_t)(K + P) * sizeof(float), lds_e = (size_t)(((K + P + 3) & ~3) + F * K) * sizeof(float);
    // (rebuilding the pair products from LDS-staged embeddings pays in the forward pooling, 0.51 -> 0.24 ms at K = 256, but not
    //  here: 0.66 -> 0.85 ms, the 40 KB of LDS per block cost more occupancy than the 759 KB read saves; DCTR_AFM_POOL_BWD_E=1)
    static const bool want_e = getenv("DCTR_AFM_POOL_BWD_E") != nullptr;
    const bool from_e = want_e && lds_e <= 150 * 1024 && E->e_ld % 4 == 0;
    const bool wide = n < 512 && K >= 64;
    auto kern = wide ? afm_pool_bwd_kernel<1024> : afm_pool_bwd_kernel<256>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(afm_pool_bwd_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    static const hipError_t attr_w = hipFuncSetAttribute(reinterpret_cast<const void*>(afm_pool_bwd_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    DCTR_HIP_CHECK(attr);
    DCTR_HIP_CHECK(attr_w);
    kern<<<n, wide ? 1024 : 256, from_e ? lds_e : lds_pp, st>>>(E->dx_in, E->Din_ld, E->pairp, E->att, P, K, E->keep_att, E->keep_emb,
                                                                 &E->state->seed_t, E->dsc, E->sc,
                                                                 from_e ? reinterpret_cast<const float4*>(E->e) : nullptr, E->e_ld / 4, F, E->pair_i, E->pair_j, b0,
                                                                 E->sc_parts);        // (the forward's score parts are free: [0, B) takes the per-example sums of d score)
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

// leaves dL/de in E->dE_buf; dense-gradient partial slabs in E->parts
int afm_backward(dctr_engine* E, int B, hipStream_t st, hipStream_t sw) {
    const int K = E->K;
    const Param& pw = E->params[E->p_out_w];
    const Param& pb = E->params[E->p_out_b];
    // deep_out (K -> 1): d y_emb(post-dropout) = dy (x) w_d into dx_in; dW/db partial slabs
    DCTR_TRY(out_layer_bwd(E->x_in, E->Din_ld, E->dy, E->pp(E->p_out_w), B, K, pw.n_part, 0, 1.f, E->dx_in, E->Din_ld,
                           E->part(E->p_out_w), pw.padded, E->part(E->p_out_b), pb.padded, st));
    return afm_interaction_backward(E, B, st, sw);
}

// the interaction layer alone (AFM.py:127-158 backward): dx_in [B, K] holds dL/d y_emb (after its dropout) on entry; leaves dL/de in
// dE_buf and the attention variables' partial slabs
int afm_interaction_backward(dctr_engine* E, int B, hipStream_t st, hipStream_t sw) {
    const int F = E->F, K = E->K, P = E->P, A = E->A;
    // (E->sc, the forward's scores, is free by now: it takes the post-dropout attention)
    const Param& aw = E->params[E->p_ao_w];
    const Param& ab = E->params[E->p_ao_b];
    if (E->afm_fused) {
        DCTR_TRY(afm_pool_bwd(E, 0, B, st));
        const Param& w = E->params[E->p_att_w];
        const Param& b = E->params[E->p_att_b];
        DCTR_TRY(afm_att_bwd(E->pairp, E->pp(E->p_att_w), E->pp(E->p_att_b), E->pp(E->p_ao_w), E->dsc, (int64_t)B * P, K, A, E->dpairp2,
                             E->part(E->p_att_w), w.padded, E->part(E->p_att_b), b.padded, E->part(E->p_ao_w), aw.padded,
                             E->part(E->p_ao_b), ab.padded, AFM_SLABS, st));
        return afm_pair_bwd(E->e, E->e_ld, E->sc, E->dx_in, E->Din_ld, E->dpairp2, E->pair_i, E->pair_j, B, F, K, P, E->dE_buf, E->D, st);
    }
    const int nc = afm_chunks(E, B), bc = ceil_div(B, nc);
    if (nc > 1) DCTR_TRY(fork(E, st, E->s_afm));
    for (int c = 0; c < nc; ++c) {
        hipStream_t s = (c & 1) ? E->s_afm : st;
        const int b0 = c * bc, n = std::min(bc, B - b0);
        const size_t r0 = (size_t)b0 * P;
        DCTR_TRY(afm_pool_bwd(E, b0, n, s));
        // One attention layer, tall enough for the weights-stationary kernel: d ah = dsc (x) w_o . 1[ah > 0] is never written -- the
        // layer's two backward products form it from ah on their operand loads (gemm_ws WS_GATE, gemm_dr DR_BGATE_WGRAD), and the
        // weight gradient's second column sums are attention_out's dW_o.  out_layer_bwd's 6.2 GB pass (1.05 ms at B = 4096) is gone.
        if (nc == 1 && E->afm_gate_slabs && ws_takes((int64_t)n * P, A, K) && aw.n_part == E->att_fc[0].splits && ab.n_part == aw.n_part) {
            const Fc& fc = E->att_fc[0];
            const Param& w = E->params[fc.w];
            const Param& bb = E->params[fc.b];
            // The two products cannot share a CU (128 KB + 98 KB of LDS): enqueued together they interleave block by block and end
            // together, with the pair backward exposed behind both.  dgrad first, then the weight gradient beside the pair backward
            // and the table step (12.42 -> 12.28 ms at B = 4096; at B = 128 the side-by-side form cost 0.81 against 0.67 ms).
            static const bool beside = getenv("DCTR_AFM_WGRAD_BESIDE") != nullptr;                       // A/B knob
            bool wdone = false, ddone = false;
            auto wgrad = [&]() -> int {
                DCTR_TRY(fork(E, st, sw));
                // db_o = sum of d score, from the per-example sums the backward pooling left (B values, not B P)
                vec_sum_partials_kernel<<<ab.n_part, 256, 0, sw>>>(E->sc_parts, n, ceil_div(n, ab.n_part), E->part(E->p_ao_b), ab.padded);
                DCTR_LAUNCH_CHECK();
                if (E->afm_ts_wgrad) {      // (declared with TS_WGRAD_SLABS slabs: every batch of this handle, whatever its size)
                    const TsPairs tp{E->e, E->e_ld, n, E->pair_i, E->pair_j, P};
                    const bool hb = E->afm_pp_skipped && E->ts_sign_ready && ts_bits_enabled();
                    DCTR_TRY(ts_fc_bwd_weights_gate(E->pairp, K, E->ah, A, E->dsc, E->pp(E->p_ao_w), E->part(fc.w), w.padded, E->part(fc.b), bb.padded,
                                                    E->part(E->p_ao_w), aw.padded, (int64_t)n * P, K, A, fc.splits, sw, &wdone, E->afm_pp_skipped ? &tp : nullptr,
                                                    hb ? E->ts_sign : nullptr, E->pp(fc.w), E->pp(fc.b)));
                    DCTR_REQUIRE(!E->afm_ah_skipped || (wdone && hb), "AFM: the forward did not store the attention layer's output and the weight gradient that works without it was not taken");
                }
                DCTR_REQUIRE(wdone || !E->afm_pp_skipped, "AFM: the forward left the pair tensor unwritten and the weight gradient that forms it in registers was not taken");
                if (!wdone)
                    DCTR_TRY(dr_fc_bwd_weights_partials_gate(E->pairp, K, E->ah, A, E->dsc, E->pp(E->p_ao_w), E->part(fc.w), w.padded, E->part(fc.b),
                                                             bb.padded, E->part(E->p_ao_w), aw.padded, n * P, K, A, fc.splits, sw, &wdone));
                return DCTR_OK;
            };
            if (beside) DCTR_TRY(wgrad());
            if (!beside || wdone) {
                if (E->gemm_mode == 1 && E->ts_planes != nullptr && afm_ts_worth((int64_t)n * P, K, A))     // split-precision mode: the gate is ONE exact bf16 plane (gemm_ts.hip)
                    DCTR_TRY(ts_fc_bwd_data_gate(E->ah, A, E->dsc, E->pp(E->p_ao_w), E->pp(fc.w), E->dpairp2, K, (int64_t)n * P, K, A,
                                                 static_cast<char*>(E->ts_planes) + ts_plane_bytes(256, 256), !E->ts_dgr_ready, st, &ddone,
                                                 E->ts_sign_ready ? E->ts_sign : nullptr));
                DCTR_REQUIRE(ddone || !E->afm_ah_skipped, "AFM: the forward did not store the attention layer's output and the input gradient that reads its sign words was not taken");
                if (!ddone) DCTR_TRY(ws_fc_bwd_data_gate(E->ah, A, E->dsc, E->pp(E->p_ao_w), E->pp(fc.w), E->dpairp2, K, n * P, K, A, st, &ddone));
                DCTR_REQUIRE(ddone, "AFM: the gated input gradient was not taken for a shape ws_takes() accepts");
                if (!beside) DCTR_TRY(wgrad());
                if (wdone) {
                    return afm_pair_bwd(E->e, E->e_ld, E->sc, E->dx_in, E->Din_ld, E->dpairp2, E->pair_i, E->pair_j, n, F, K, P, E->dE_buf, E->D, st);
                }
                // (the weight gradient's shape was not the direct kernel's: the materialising path below redoes both products)
            }
        }
        // attention_out (A -> 1) over the chunk's rows: d ah = dsc (x) w_o masked by relu, dW_o / db_o partial slabs
        const int ao_n = aw.n_part / nc, ab_n = ab.n_part / nc;
        DCTR_REQUIRE(ao_n == ab_n, "AFM: attention_out weight / bias slab counts differ");
        DCTR_TRY(out_layer_bwd(E->ah + r0 * A, A, E->dsc + r0, E->pp(E->p_ao_w), n * P, A, ao_n, 1, 1.f, E->dah + r0 * A, A,
                               E->part(E->p_ao_w) + (size_t)c * ao_n * aw.padded, aw.padded, E->part(E->p_ao_b) + (size_t)c * ab_n * ab.padded,
                               ab.padded, s));
        // attention layers, last to first: wgrad on the side stream (one chunk: beside its dgrad; several: the other lane's passes
        // are what runs beside it), dgrad (x the ReLU mask of the layer below) on the lane
        for (int l = (int)E->att_fc.size() - 1; l >= 0; --l) {
            const Fc& fc = E->att_fc[l];
            const Param& w = E->params[fc.w];
            const Param& b = E->params[fc.b];
            const float* xin = l > 0 ? E->ahs[l - 1] + r0 * fc.in : E->pairp + r0 * K;
            const float* dy = E->dahs[l] + r0 * fc.out;
            const int sp = fc.splits / nc;
            hipStream_t swl = nc > 1 ? s : sw;
            // Below ~1 M pair rows the two products of a layer must NOT run side by side: at the reference's B = 128 (95 k rows) the
            // weight gradient took 384 us beside the 131 us input gradient and 106 us alone -- each of them fills the chip, and the
            // pair backward / table step behind the dgrad ran 5x slower under it (step 0.81 -> 0.69 ms serial).  There the weight
            // gradient starts after the dgrad, beside the HBM-bound passes that follow it.
            const bool beside_dgrad = nc == 1 && (int64_t)n * P >= (1 << 20);
            float* dwp = E->part(fc.w) + (size_t)c * sp * w.padded;
            float* dbp = E->part(fc.b) + (size_t)c * sp * b.padded;
            if (beside_dgrad) {
                DCTR_TRY(fork(E, st, sw));              // dahs[l] is complete on st
                DCTR_TRY(fc_bwd_weights_partials(xin, fc.in, dy, fc.out, dwp, w.padded, dbp, b.padded, n * P, fc.in, fc.out, sp, swl));
            }
            if (l > 0) DCTR_TRY(fc_bwd_data(dy, fc.out, E->pp(fc.w), E->dahs[l - 1] + r0 * fc.in, fc.in, n * P, fc.in, fc.out, E->ahs[l - 1] + r0 * fc.in, fc.in, 1.f, s));
            else DCTR_TRY(fc_bwd_data(dy, fc.out, E->pp(fc.w), E->dpairp2 + r0 * K, K, n * P, K, fc.out, nullptr, 0, 1.f, s));
            if (nc == 1 && !beside_dgrad) {
                DCTR_TRY(fork(E, st, sw));              // the dgrad is enqueued: the weight gradient runs behind it, beside what follows
                DCTR_TRY(fc_bwd_weights_partials(xin, fc.in, dy, fc.out, dwp, w.padded, dbp, b.padded, n * P, fc.in, fc.out, sp, swl));
            }
            if (l == 0)
                DCTR_TRY(afm_pair_bwd(E->e + (size_t)b0 * E->e_ld, E->e_ld, E->sc + r0, E->dx_in + (size_t)b0 * E->Din_ld, E->Din_ld, E->dpairp2 + r0 * K,
                                      E->pair_i, E->pair_j, n, F, K, P, E->dE_buf + (size_t)b0 * E->D, E->D, s));
            if (nc > 1) DCTR_TRY(fc_bwd_weights_partials(xin, fc.in, dy, fc.out, dwp, w.padded, dbp, b.padded, n * P, fc.in, fc.out, sp, swl));
        }
    }
    if (nc > 1) DCTR_TRY(fork(E, E->s_afm, st));
    return DCTR_OK;
}
