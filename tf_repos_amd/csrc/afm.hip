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

// pp4[(b*P + p)*KQ + kq] = e4[b,i_p,kq] * e4[b,j_p,kq]
__global__ __launch_bounds__(256) void afm_pair_fwd_kernel(const float4* __restrict__ e, int e_ld4, const int16_t* __restrict__ pi,
                                                          const int16_t* __restrict__ pj, int B, int P, int KQ,
                                                          float4* __restrict__ pp) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)B * P * KQ;
    if (t >= n) return;
    const int kq = (int)(t % KQ);
    const int64_t bp = t / KQ;
    const int p = (int)(bp % P), b = (int)(bp / P);
    const float4 a = e[(size_t)b * e_ld4 + (size_t)pi[p] * KQ + kq];
    const float4 c = e[(size_t)b * e_ld4 + (size_t)pj[p] * KQ + kq];
    pp[t] = make_float4(a.x * c.x, a.y * c.y, a.z * c.z, a.w * c.w);
}

// one block per example: softmax over the P scores, attention dropout, pooling over the pairs, y_emb dropout
// (ee != nullptr: the pair products are rebuilt from the example's embeddings, staged in LDS behind the weights -- 40 KB read per
//  example instead of its 759 KB slice of the [B P, K] pair tensor, K = 256)
__global__ __launch_bounds__(256) void afm_pool_fwd_kernel(const float* __restrict__ sc, const float* __restrict__ pp, int P, int K,
                                                          float keep_att, float keep_emb, const uint64_t* __restrict__ seed_ptr,
                                                          int train, float* __restrict__ att, float* __restrict__ yemb,
                                                          const float4* __restrict__ ee, int e_ld4, int F, const int16_t* __restrict__ pi,
                                                          const int16_t* __restrict__ pj) {
    extern __shared__ __attribute__((aligned(16))) float sm[];        // [P] attention weights (after dropout) | [F][K] embeddings (ee)
    __shared__ float red[4];
    const int b = blockIdx.x, t = threadIdx.x;
    float4* es = reinterpret_cast<float4*>(sm + ((P + 3) & ~3));
    if (ee != nullptr)
        for (int x = t; x < F * (K >> 2); x += 256) es[x] = ee[(size_t)b * e_ld4 + x];
    const float* s = sc + (size_t)b * P;
    float m = -3.0e38f;
    for (int p = t; p < P; p += 256) m = fmaxf(m, s[p]);
    m = wmax64(m);
    if ((t & 63) == 0) red[t >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float z = 0.f;
    for (int p = t; p < P; p += 256) { const float ex = expf(s[p] - m); sm[p] = ex; z += ex; }
    z = wsum64(z);
    if ((t & 63) == 0) red[t >> 6] = z;
    __syncthreads();
    z = red[0] + red[1] + red[2] + red[3];
    const float inv = 1.0f / z;
    const uint64_t seed = (train && (keep_att < 1.f || keep_emb < 1.f)) ? *seed_ptr : 0ull;
    for (int p = t; p < P; p += 256) {
        const float a = sm[p] * inv;
        att[(size_t)b * P + p] = a;                                            // softmax output (before dropout): kept for the backward
        sm[p] = (train && keep_att < 1.f) ? a * dropout_scale(seed ^ DCTR_DROPOUT_SITE_AFM_ATT, (uint64_t)b * P + p, keep_att) : a;
    }
    __syncthreads();
    // y_emb[k] = sum_p a'[p] pp[p,k]: thread = (pair slice, float4 piece), slices summed through LDS
    __shared__ float4 acc4[256];
    const int KQ = K >> 2, q = t % KQ, slice = t / KQ, n_slices = 256 / KQ;
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
        if (train && keep_emb < 1.f) y *= dropout_scale(seed ^ DCTR_DROPOUT_SITE_AFM_YEMB, (uint64_t)b * K + t, keep_emb);
        yemb[(size_t)b * K + t] = y;
    }
}

// backward of dropout[1] -> pooling -> dropout[0] -> softmax.  in: dy [B, dy_ld] = d y_emb (post-dropout); out: dsc [B,P], the
// post-dropout attention a' [B,P] and, in place of dy, the pre-dropout d y_emb.  d pp = a' (x) d y_emb is NOT materialised: the
// pair backward below forms it from these two.
__global__ __launch_bounds__(256) void afm_pool_bwd_kernel(float* __restrict__ dy, int dy_ld, const float* __restrict__ pp,
                                                          const float* __restrict__ att, int P, int K, float keep_att, float keep_emb,
                                                          const uint64_t* __restrict__ seed_ptr, float* __restrict__ dsc,
                                                          float* __restrict__ att_drop, const float4* __restrict__ ee, int e_ld4, int F,
                                                          const int16_t* __restrict__ pi, const int16_t* __restrict__ pj) {
    extern __shared__ __attribute__((aligned(16))) float sm[];        // [K] dyemb (pre-dropout gradient) | [P] da | [F][K] embeddings (ee)
    __shared__ float red[4];
    float* dye = sm;
    float* da = sm + K;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float4* es = reinterpret_cast<float4*>(sm + ((K + P + 3) & ~3));
    if (ee != nullptr)
        for (int x = t; x < F * (K >> 2); x += 256) es[x] = ee[(size_t)b * e_ld4 + x];
    const uint64_t seed = (keep_att < 1.f || keep_emb < 1.f) ? *seed_ptr : 0ull;
    for (int k = t; k < K; k += 256) {
        float g = dy[(size_t)b * dy_ld + k];
        if (keep_emb < 1.f) g *= dropout_scale(seed ^ DCTR_DROPOUT_SITE_AFM_YEMB, (uint64_t)b * K + k, keep_emb);
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
    for (int i = t; i < P * KQ; i += 256) {
        const int p = i / KQ;
        float4 v;
        if (ee != nullptr) {
            const float4 x = es[pi[p] * KQ + q], y = es[pj[p] * KQ + q];
            v = make_float4(x.x * y.x, x.y * y.y, x.z * y.z, x.w * y.w);
        } else {
            v = pp4[i];
        }
        float s = d4.x * v.x + d4.y * v.y + d4.z * v.z + d4.w * v.w;
        for (int o = 1; o < KQ; o <<= 1) s += __shfl_xor(s, o);
        if (q == 0) {
            const float msk = keep_att < 1.f ? dropout_scale(seed ^ DCTR_DROPOUT_SITE_AFM_ATT, (uint64_t)b * P + p, keep_att) : 1.f;
            const float d = s * msk, a = ab[p];                // d att[p]
            da[p] = d;
            att_drop[(size_t)b * P + p] = a * msk;
            part += a * d;
        }
    }
    part = wsum64(part);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    for (int p = t; p < P; p += 256) dsc[(size_t)b * P + p] = ab[p] * (da[p] - tot);      // softmax backward
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

// The same with float4 pieces and four pairs in flight per thread (the loop above issues one 4-byte load per pair and waits for
// it: 1.1 TB/s on 6 GB at the reference's K = 256, B = 4096).  Thread = (field i, piece kq) of one example; every row of g2 is
// still read twice (once from each of its fields' side) -- a one-pass variant that folded both contributions into an LDS
// accumulator with ds_add_f32 measured 2x SLOWER than the loop above (the float atomics serialise).
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
    // partner fields j = 0 .. F-1 except i, four at a time
    for (int j0 = 0; j0 < F; j0 += 4) {
        float4 v[4], ej[4]; float a[4]; bool on[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u;
            on[u] = j < F && j != i;
            const int jc = on[u] ? j : (i == 0 ? 1 : 0);                 // (clamped to a valid partner: the loads carry no branch)
            const int lo = jc < i ? jc : i, hi = jc < i ? i : jc;
            const int p = lo * F - (lo * (lo + 1)) / 2 + (hi - lo - 1);
            v[u] = c[(size_t)p * KQ + kq];
            a[u] = ad[p];
            ej[u] = eb[jc * KQ + kq];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (on[u]) {
                s.x += (a[u] * dk.x + v[u].x) * ej[u].x; s.y += (a[u] * dk.y + v[u].y) * ej[u].y;
                s.z += (a[u] * dk.z + v[u].z) * ej[u].z; s.w += (a[u] * dk.w + v[u].w) * ej[u].w;
            }
    }
    dE[(size_t)b * de_ld4 + x] = s;
}

int afm_pair_bwd(const float* e, int e_ld, const float* att_drop, const float* dye, int dye_ld, const float* g2, const int16_t* pair_i,
                 const int16_t* pair_j, int B, int F, int K, int P, float* dE, int de_ld, hipStream_t st) {
    (void)pair_i; (void)pair_j;
    static const bool old = getenv("DCTR_AFM_PAIR_BWD_OLD") != nullptr;          // A/B knob: the one-load-per-pair kernel
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
    const int ao = E->afm_fused ? AFM_SLABS : E->ao_splits;
    int d = K;
    for (int l = 0; l < nl; ++l) {
        Fc fc;
        fc.in = d; fc.out = c.attention_layers[l];
        fc.splits = E->afm_fused ? AFM_SLABS : choose_wgrad_splits(E->MB * E->P, fc.in, fc.out);
        char nm[64];
        snprintf(nm, sizeof(nm), "att_mlp%d/weights", l);
        fc.w = add(E, nm, {fc.in, fc.out}, false, fc.splits, 0.f);
        snprintf(nm, sizeof(nm), "att_mlp%d/biases", l);
        fc.b = add(E, nm, {fc.out}, false, fc.splits, 0.f);
        fc.last = fc.b;
        E->att_fc.push_back(fc);
        d = fc.out;
    }
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
    float* fl[] = {E->pairp, E->dpairp2, E->sc, E->dsc, E->att, E->dE_buf};
    for (float* p : fl) if (p) hipFree(p);
    for (float* p : E->ahs) hipFree(p);
    for (float* p : E->dahs) hipFree(p);
    if (E->pair_i) hipFree(E->pair_i);
    if (E->pair_j) hipFree(E->pair_j);
}

// after the gather (mode RAW + linear) has filled E->e / E->yw
int afm_forward(dctr_engine* E, int B, bool train, hipStream_t st) {
    const int F = E->F, K = E->K, P = E->P, A = E->A, KQ = K / 4;
    const int64_t n4 = (int64_t)B * P * KQ;
    afm_pair_fwd_kernel<<<ceil_div(n4, 256), 256, 0, st>>>(reinterpret_cast<const float4*>(E->e), E->e_ld / 4, E->pair_i, E->pair_j, B,
                                                            P, KQ, reinterpret_cast<float4*>(E->pairp));
    DCTR_LAUNCH_CHECK();
    if (E->afm_fused) {
        // scores straight from the pair products (the hidden layer never leaves the registers; the backward recomputes it)
        DCTR_TRY(afm_att_fwd(E->pairp, E->pp(E->p_att_w), E->pp(E->p_att_b), E->pp(E->p_ao_w), E->pp(E->p_ao_b), (int64_t)B * P, K, A, E->sc, st));
    } else {
        const float* x = E->pairp;
        for (size_t l = 0; l < E->att_fc.size(); ++l) {     // relu(x W_l + b_l) over the B*P pair rows
            const Fc& fc = E->att_fc[l];
            DCTR_TRY(fc_fwd(x, fc.in, E->pp(fc.w), E->pp(fc.b), E->ahs[l], fc.out, B * P, fc.in, fc.out, 1, 1.f, nullptr, 0, st));
            x = E->ahs[l];
        }
        DCTR_TRY(rowdot(E->ah, A, E->pp(E->p_ao_w), E->pp(E->p_ao_b), B * P, A, E->sc, 0, st));
    }
    {
        // the pooling rebuilds the pair products from the example's embeddings when they fit LDS beside its working set
        const size_t lds_pp = (size_t)P * sizeof(float), lds_e = (size_t)(((P + 3) & ~3) + F * K) * sizeof(float);
        const bool from_e = lds_e <= 150 * 1024 && E->e_ld % 4 == 0;
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(afm_pool_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        DCTR_HIP_CHECK(attr);
        afm_pool_fwd_kernel<<<B, 256, from_e ? lds_e : lds_pp, st>>>(E->sc, E->pairp, P, K, E->keep_att, E->keep_emb, &E->state->seed_t,
                                                                     train ? 1 : 0, E->att, E->x_in,
                                                                     from_e ? reinterpret_cast<const float4*>(E->e) : nullptr, E->e_ld / 4, F, E->pair_i, E->pair_j);
        DCTR_LAUNCH_CHECK();
    }
    return DCTR_OK;      // the fc(K -> 1) output layer is fused into the head kernel
}

// leaves dL/de in E->dE_buf; dense-gradient partial slabs in E->parts
int afm_backward(dctr_engine* E, int B, hipStream_t st, hipStream_t sw) {
    const int F = E->F, K = E->K, P = E->P, A = E->A;
    const Param& pw = E->params[E->p_out_w];
    const Param& pb = E->params[E->p_out_b];
    // deep_out (K -> 1): d y_emb(post-dropout) = dy (x) w_d into dx_in; dW/db partial slabs
    DCTR_TRY(out_layer_bwd(E->x_in, E->Din_ld, E->dy, E->pp(E->p_out_w), B, K, pw.n_part, 0, 1.f, E->dx_in, E->Din_ld,
                           E->part(E->p_out_w), pw.padded, E->part(E->p_out_b), pb.padded, st));
    // (E->sc, the forward's scores, is free by now: it takes the post-dropout attention)
    {
        const size_t lds_pp = (size_t)(K + P) * sizeof(float), lds_e = (size_t)(((K + P + 3) & ~3) + F * K) * sizeof(float);
        // (rebuilding the pair products from LDS-staged embeddings pays in the forward pooling, 0.51 -> 0.24 ms at K = 256, but not
        //  here: 0.66 -> 0.85 ms, the 40 KB of LDS per block cost more occupancy than the 759 KB read saves; DCTR_AFM_POOL_BWD_E=1)
        static const bool want_e = getenv("DCTR_AFM_POOL_BWD_E") != nullptr;
        const bool from_e = want_e && lds_e <= 150 * 1024 && E->e_ld % 4 == 0;
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(afm_pool_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        DCTR_HIP_CHECK(attr);
        afm_pool_bwd_kernel<<<B, 256, from_e ? lds_e : lds_pp, st>>>(E->dx_in, E->Din_ld, E->pairp, E->att, P, K, E->keep_att, E->keep_emb,
                                                                     &E->state->seed_t, E->dsc, E->sc,
                                                                     from_e ? reinterpret_cast<const float4*>(E->e) : nullptr, E->e_ld / 4, F, E->pair_i, E->pair_j);
        DCTR_LAUNCH_CHECK();
    }
    const Param& aw = E->params[E->p_ao_w];
    const Param& ab = E->params[E->p_ao_b];
    if (E->afm_fused) {
        const Param& w = E->params[E->p_att_w];
        const Param& b = E->params[E->p_att_b];
        DCTR_TRY(afm_att_bwd(E->pairp, E->pp(E->p_att_w), E->pp(E->p_att_b), E->pp(E->p_ao_w), E->dsc, (int64_t)B * P, K, A, E->dpairp2,
                             E->part(E->p_att_w), w.padded, E->part(E->p_att_b), b.padded, E->part(E->p_ao_w), aw.padded,
                             E->part(E->p_ao_b), ab.padded, AFM_SLABS, st));
        return afm_pair_bwd(E->e, E->e_ld, E->sc, E->dx_in, E->Din_ld, E->dpairp2, E->pair_i, E->pair_j, B, F, K, P, E->dE_buf, E->D, st);
    }
    // attention_out (A -> 1) over the B*P rows: d ah = dsc (x) w_o masked by relu, dW_o / db_o partial slabs
    DCTR_TRY(out_layer_bwd(E->ah, A, E->dsc, E->pp(E->p_ao_w), B * P, A, aw.n_part, 1, 1.f, E->dah, A, E->part(E->p_ao_w), aw.padded,
                           E->part(E->p_ao_b), ab.padded, st));
    // attention layers, last to first: wgrad on the side stream, dgrad (x the ReLU mask of the layer below) on the critical path
    for (int l = (int)E->att_fc.size() - 1; l >= 0; --l) {
        const Fc& fc = E->att_fc[l];
        const Param& w = E->params[fc.w];
        const Param& b = E->params[fc.b];
        const float* xin = l > 0 ? E->ahs[l - 1] : E->pairp;
        DCTR_TRY(fork(E, st, sw));              // dahs[l] is complete on st
        DCTR_TRY(fc_bwd_weights_partials(xin, fc.in, E->dahs[l], fc.out, E->part(fc.w), w.padded, E->part(fc.b), b.padded, B * P, fc.in, fc.out,
                                         fc.splits, sw));
        if (l > 0) DCTR_TRY(fc_bwd_data(E->dahs[l], fc.out, E->pp(fc.w), E->dahs[l - 1], fc.in, B * P, fc.in, fc.out, E->ahs[l - 1], fc.in, 1.f, st));
        else DCTR_TRY(fc_bwd_data(E->dahs[0], fc.out, E->pp(fc.w), E->dpairp2, K, B * P, K, fc.out, nullptr, 0, 1.f, st));
    }
    return afm_pair_bwd(E->e, E->e_ld, E->sc, E->dx_in, E->Din_ld, E->dpairp2, E->pair_i, E->pair_j, B, F, K, P, E->dE_buf, E->D, st);
}
