// K3 / K5: interaction layers between the gather and the MLP.
//   PNN inner product  (PNN.py:141-153)  ip[b,p] = <e_i, e_j>, pairs (i<j) lexicographic
//   PNN outer product  (PNN.py:154-167)  op[b,p,a,c] = e_i[a] e_j[c]  (materialised; reference marks it "NOT ready yet")
//   DCN cross network  (DCN.py:140-145)  x_{l+1} = x0 (x_l . w_l) + x_l + b_l
// All are <= 1 flop/byte per HBM byte: one example's F*K scaled-embedding tile is staged once in LDS
// (or registers for DCN) and every pair / layer is computed from there.
#include "common.h"
#include "ops.h"

namespace dctr {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ __forceinline__ int pair_index(int i, int j, int F) {   // i < j
    return i * F - (i * (i + 1)) / 2 + (j - i - 1);
}

// ---- inner product forward: one wave per example, e tile in LDS with row stride K+1 (bank-conflict-free
// when lanes read different rows at the same k)
__global__ __launch_bounds__(256) void pnn_inner_fwd_kernel(const float* __restrict__ e, int e_ld, int B, int F, int K,
                                                           float* __restrict__ ip, int ip_ld) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    const int KS = K + 1;
    float* t = smem + (size_t)wave * F * KS;
    if (b < B) {
        const float* er = e + (size_t)b * e_ld;
        for (int x = lane; x < F * K; x += 64) t[(x / K) * KS + (x % K)] = er[x];
    }
    __syncthreads();
    if (b >= B) return;
    const int P = F * (F - 1) / 2;
    // decode pair p -> (i,j) incrementally per lane
    for (int p = lane; p < P; p += 64) {
        // find i: largest i with i*F - i(i+1)/2 <= p
        int i = 0, base = 0;
        while (true) {
            const int next = base + (F - 1 - i);
            if (p < next) break;
            base = next;
            ++i;
        }
        const int j = i + 1 + (p - base);
        const float* a = t + i * KS;
        const float* c = t + j * KS;
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += a[k] * c[k];
        ip[(size_t)b * ip_ld + p] = s;
    }
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
// C/D layout of v_mfma_f32_32x32x2_f32: register r of lane l holds row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31
__device__ __forceinline__ int mfma_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// ---- inner products on the matrix cores: per example G = E E^T ([F,K] x [K,F], F <= 64 padded to 64) as three 32x32 blocks
// (the lower-left one is never needed), K/2 MFMA steps each; a wave per example, E staged once in LDS (row stride K+1: both
// operands read conflict-free).  For a fixed i the pairs (i, j > i) are consecutive p: the accumulator rows store coalesced.
// The scalar form moved 2 LDS words per MAC (LDS-bound: 55 us at B=8192, F=39, K=32); MFMA operands are 1/32 of that.
template <int F, int K>
__global__ __launch_bounds__(256) void pnn_inner_fwd_mfma_kernel(const float* __restrict__ e, int e_ld, int B,
                                                                float* __restrict__ ip, int ip_ld) {
    constexpr int KS = K + 1, KQ = K / 4;
    constexpr int NLD = (F * KQ + 63) / 64;               // float4 loads per lane to stage one example
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, col = lane & 31;
    float* t = smem + (size_t)wave * 64 * KS;
    for (int x = lane; x < (64 - F) * KS; x += 64) t[F * KS + x] = 0.f;          // rows F..63 stay zero
    for (int b = blockIdx.x * 4 + wave; b < B; b += gridDim.x * 4) {
        // the example's [F, K] tile: every lane's loads are issued together (a load -> LDS store loop would expose the memory
        // latency NLD times per example: that, not LDS or FLOPs, was what the scalar kernel spent its 55 us on)
        const float4* er = reinterpret_cast<const float4*>(e + (size_t)b * e_ld);
        float4 v[NLD];
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int x = u * 64 + lane;
            v[u] = x < F * KQ ? er[x] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int x = u * 64 + lane;
            if (x < F * KQ) {
                float* d = t + (x / KQ) * KS + 4 * (x % KQ);
                d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
            }
        }
        __builtin_amdgcn_wave_barrier();
        f32x16 acc[3];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
#pragma unroll
        for (int k = 0; k < K; k += 2) {
            const float lo = t[col * KS + k + half], hi = t[(32 + col) * KS + k + half];      // rows 0..31 / 32..63 at column k+half
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(lo, lo, acc[0], 0, 0, 0);           // i in [0,32),  j in [0,32)
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(lo, hi, acc[1], 0, 0, 0);           // i in [0,32),  j in [32,64)
            if (F > 33) acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(hi, hi, acc[2], 0, 0, 0);   // i in [32,64), j in [32,64)
        }
        float* out = ip + (size_t)b * ip_ld;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int i0 = q == 2 ? 32 : 0, j0 = q == 0 ? 0 : 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + mfma_row(r, half), j = j0 + col;
                if (i < j && j < F) out[pair_index(i, j, F)] = acc[q][r];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

int pnn_inner_fwd(const float* e, int e_ld, int B, int F, int K, float* ip, int ip_ld, hipStream_t st) {
    if (B <= 0) return DCTR_OK;
    static const bool generic = getenv("DCTR_PNN_GENERIC") != nullptr;          // A/B knob
    if (F == 39 && e_ld % 4 == 0 && !generic) {         // the Criteo field count (compile-time sizes for the register arrays)
        const size_t ldsm = (size_t)4 * 64 * (K + 1) * sizeof(float);
        const int grid = std::min(ceil_div(B, 4), 256 * 4);
        switch (K) {
#define DCTR_PF(KK) case KK: pnn_inner_fwd_mfma_kernel<39, KK><<<grid, 256, ldsm, st>>>(e, e_ld, B, ip, ip_ld); DCTR_LAUNCH_CHECK(); return DCTR_OK
            DCTR_PF(4); DCTR_PF(8); DCTR_PF(16); DCTR_PF(32); DCTR_PF(64);
#undef DCTR_PF
            default: break;
        }
    }
    const size_t lds = (size_t)4 * F * (K + 1) * sizeof(float);
    DCTR_REQUIRE(lds <= 160 * 1024, "pnn_inner: F*K tile too large for LDS (F=%d K=%d)", F, K);
    pnn_inner_fwd_kernel<<<ceil_div(B, 4), 256, lds, st>>>(e, e_ld, B, F, K, ip, ip_ld);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

// ---- inner product backward: dE[b,i,:] += sum_{j != i} dip[b,pair(i,j)] e[b,j,:]
__global__ __launch_bounds__(256) void pnn_inner_bwd_kernel(const float* __restrict__ e, int e_ld, const float* __restrict__ dip,
                                                           int dip_ld, int B, int F, int K, float* __restrict__ dE, int de_ld) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    const int P = F * (F - 1) / 2;
    float* t = smem + (size_t)wave * (F * K + P);
    float* g = t + F * K;
    if (b < B) {
        const float* er = e + (size_t)b * e_ld;
        for (int x = lane; x < F * K; x += 64) t[x] = er[x];
        const float* dr = dip + (size_t)b * dip_ld;
        for (int x = lane; x < P; x += 64) g[x] = dr[x];
    }
    __syncthreads();
    if (b >= B) return;
    for (int x = lane; x < F * K; x += 64) {
        const int i = x / K, k = x - i * K;
        float s = 0.f;
        for (int j = 0; j < i; ++j) s += g[pair_index(j, i, F)] * t[j * K + k];
        for (int j = i + 1; j < F; ++j) s += g[pair_index(i, j, F)] * t[j * K + k];
        dE[(size_t)b * de_ld + x] += s;
    }
}

// On the matrix cores: dE_b += G_b E_b with G_b the dense symmetric [F, F] matrix of the example's pair gradients (zero
// diagonal), F <= 64 padded to 64 rows x FP columns.  A wave per example expands dip[b, :] into G in LDS (row stride odd: the A
// operand reads are conflict-free), the B operand e[b, k, n] comes straight from global memory (coalesced, each element once), two
// 32-row blocks x FP/2 MFMA steps per 32 embedding columns, and the accumulator rows are added into dE coalesced.
// (scalar form: 192 us at B=8192, F=39, K=32 -- two LDS words and the pair-index arithmetic per MAC.)
template <int F>
__global__ __launch_bounds__(256) void pnn_inner_bwd_mfma_kernel(const float* __restrict__ e, int e_ld, const float* __restrict__ dip,
                                                                int dip_ld, int B, int K, float* __restrict__ dE, int de_ld) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int P = F * (F - 1) / 2;
    constexpr int FP = (F + 1) & ~1;                      // contraction length (even)
    constexpr int GS = FP | 1;                            // row stride of G (odd)
    constexpr int NP = (P + 63) / 64;                     // pair gradients per lane
    int16_t* pi = reinterpret_cast<int16_t*>(smem);
    int16_t* pj = pi + P;
    float* Gall = smem + (((size_t)2 * P * sizeof(int16_t) + 15) / 16) * 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, col = lane & 31;
    for (int p = threadIdx.x; p < P; p += 256) {
        int i = 0, base = 0;
        while (p >= base + (F - 1 - i)) { base += F - 1 - i; ++i; }
        pi[p] = (int16_t)i; pj[p] = (int16_t)(i + 1 + (p - base));
    }
    float* G = Gall + (size_t)wave * 64 * GS;
    for (int x = lane; x < 64 * GS; x += 64) G[x] = 0.f;  // diagonal, rows >= F and the padding column stay zero
    __syncthreads();
    for (int b = blockIdx.x * 4 + wave; b < B; b += gridDim.x * 4) {
        // (every batch of global loads below is issued together and consumed afterwards: one dependent load per loop trip
        //  exposes the memory latency dozens of times per example -- that was the 120-190 us of the earlier forms)
        const float* dr = dip + (size_t)b * dip_ld;
        float gv[NP];
#pragma unroll
        for (int u = 0; u < NP; ++u) { const int p = u * 64 + lane; gv[u] = p < P ? dr[p] : 0.f; }
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int p = u * 64 + lane;
            if (p < P) { G[pi[p] * GS + pj[p]] = gv[u]; G[pj[p] * GS + pi[p]] = gv[u]; }
        }
        __builtin_amdgcn_wave_barrier();                  // (wave-private tile: the LDS queue of a wave is in order)
        const float* er = e + (size_t)b * e_ld;
        float* dr_out = dE + (size_t)b * de_ld;
        for (int n0 = 0; n0 < K; n0 += 32) {
            const int n = n0 + col;
            float bop[FP / 2], o0[16], o1[16];
#pragma unroll
            for (int s2 = 0; s2 < FP / 2; ++s2) { const int kk = 2 * s2 + half; bop[s2] = (kk < F && n < K) ? er[kk * K + n] : 0.f; }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = mfma_row(r, half);
                o0[r] = (i < F && n < K) ? dr_out[i * K + n] : 0.f;
                o1[r] = (32 + i < F && n < K) ? dr_out[(32 + i) * K + n] : 0.f;
            }
            f32x16 a0, a1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
#pragma unroll
            for (int s2 = 0; s2 < FP / 2; ++s2) {
                const int kk = 2 * s2 + half;
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(G[col * GS + kk], bop[s2], a0, 0, 0, 0);
                if (F > 32) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(G[(32 + col) * GS + kk], bop[s2], a1, 0, 0, 0);
            }
            if (n < K) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = mfma_row(r, half);
                    if (i < F) dr_out[i * K + n] = o0[r] + a0[r];
                    if (32 + i < F) dr_out[(32 + i) * K + n] = o1[r] + a1[r];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();                  // G is rewritten for the next example
    }
}

int pnn_inner_bwd(const float* e, int e_ld, const float* dip, int dip_ld, int B, int F, int K, float* dE, int de_ld,
                  hipStream_t st) {
    if (B <= 0) return DCTR_OK;
    static const bool generic = getenv("DCTR_PNN_GENERIC") != nullptr;          // A/B knob
    if (F == 39 && !generic) {                          // the Criteo field count (compile-time sizes for the register arrays)
        constexpr int P2 = 39 * 38 / 2, GS = 41;
        const size_t ldsm = (((size_t)2 * P2 * sizeof(int16_t) + 15) / 16) * 16 + (size_t)4 * 64 * GS * sizeof(float);
        DCTR_LAUNCH_RIDE(pnn_inner_bwd_mfma_kernel<39>, dim3(std::min(ceil_div(B, 4), 256 * 3)), dim3(256), (uint32_t)ldsm, st, e, e_ld, dip, dip_ld, B, K, dE, de_ld);
        DCTR_LAUNCH_CHECK();
        return DCTR_OK;
    }
    const int P = F * (F - 1) / 2;
    const size_t lds = (size_t)4 * (F * K + P) * sizeof(float);
    DCTR_REQUIRE(lds <= 160 * 1024, "pnn_inner_bwd: tile too large for LDS (F=%d K=%d)", F, K);
    pnn_inner_bwd_kernel<<<ceil_div(B, 4), 256, lds, st>>>(e, e_ld, dip, dip_ld, B, F, K, dE, de_ld);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

// ---- outer product, materialised exactly as PNN.py:166 writes it: [B, P*K*K]
__global__ __launch_bounds__(256) void pnn_outer_fwd_kernel(const float* __restrict__ e, int e_ld, int B, int F, int K,
                                                           float* __restrict__ op, int64_t op_ld) {
    const int b = blockIdx.y;
    const int P = F * (F - 1) / 2;
    const int64_t n = (int64_t)P * K * K;
    const float* er = e + (size_t)b * e_ld;
    for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < n; x += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(x / (K * K));
        const int r = (int)(x - (int64_t)p * K * K);
        const int a = r / K, c = r - a * K;
        int i = 0, base = 0;
        while (true) {
            const int next = base + (F - 1 - i);
            if (p < next) break;
            base = next;
            ++i;
        }
        const int j = i + 1 + (p - base);
        op[(size_t)b * op_ld + x] = er[i * K + a] * er[j * K + c];
    }
}

int pnn_outer_fwd(const float* e, int e_ld, int B, int F, int K, float* op, int64_t op_ld, hipStream_t st) {
    if (B <= 0) return DCTR_OK;
    const int64_t n = (int64_t)F * (F - 1) / 2 * K * K;
    dim3 grid((unsigned)std::min<int64_t>(ceil_div(n, 256), 1024), B);
    pnn_outer_fwd_kernel<<<grid, 256, 0, st>>>(e, e_ld, B, F, K, op, op_ld);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

// dE[b,f,k] += sum_{j>f} sum_c dop[b,p(f,j),k,c] e[b,j,c]  +  sum_{i<f} sum_a dop[b,p(i,f),a,k] e[b,i,a]
__global__ __launch_bounds__(256) void pnn_outer_bwd_kernel(const float* __restrict__ e, int e_ld, const float* __restrict__ dop,
                                                           int64_t dop_ld, int B, int F, int K, float* __restrict__ dE, int de_ld) {
    const int b = blockIdx.y;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= F * K) return;
    const int f = x / K, k = x - f * K;
    const float* er = e + (size_t)b * e_ld;
    const float* dr = dop + (size_t)b * dop_ld;
    float s = 0.f;
    for (int j = f + 1; j < F; ++j) {
        const float* d = dr + ((size_t)pair_index(f, j, F) * K + k) * K;
        for (int c = 0; c < K; ++c) s += d[c] * er[j * K + c];
    }
    for (int i = 0; i < f; ++i) {
        const float* d = dr + (size_t)pair_index(i, f, F) * K * K + k;
        for (int a = 0; a < K; ++a) s += d[(size_t)a * K] * er[i * K + a];
    }
    dE[(size_t)b * de_ld + x] += s;
}

int pnn_outer_bwd(const float* e, int e_ld, const float* dop, int64_t dop_ld, int B, int F, int K, float* dE, int de_ld,
                  hipStream_t st) {
    if (B <= 0) return DCTR_OK;
    dim3 grid(ceil_div(F * K, 256), B);
    pnn_outer_bwd_kernel<<<grid, 256, 0, st>>>(e, e_ld, dop, dop_ld, B, F, K, dE, de_ld);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

// sum over the G lanes that share an example: a wave (G = 64) or the whole 256-thread block (G = 256, F*K > 2560: --embedding_size
// 128 / 256 with 39 fields)
template <int G>
__device__ __forceinline__ float dcn_group_sum(float v) {
    v = wsum(v);
    if (G == 64) return v;
    __shared__ float part[4];
    __syncthreads();                         // (the previous round's readers are done with part[])
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    return part[0] + part[1] + part[2] + part[3];
}

// ---- DCN cross network.  One wave (or, for wide inputs, one block) per example; x0 and x_l live in registers (NR = ceil(D/G)
// floats per lane); all L layers are applied without leaving the CU.  xs[l] (l = 0..L) and s_l = x_l.w_l are kept for the backward.
template <int NR, int G = 64>
__global__ __launch_bounds__(256) void dcn_cross_fwd_kernel(const float* __restrict__ x0, int x0_ld, const float* __restrict__ w,
                                                           const float* __restrict__ bias, int B, int D, int L,
                                                           float* __restrict__ xs, float* __restrict__ xlw) {
    const int lane = threadIdx.x % G;
    const int b = blockIdx.x * (256 / G) + threadIdx.x / G;
    if (b >= B) return;
    float a0[NR], xl[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int d = lane + G * r;
        a0[r] = (d < D) ? x0[(size_t)b * x0_ld + d] : 0.f;
        xl[r] = a0[r];
        if (d < D) xs[(size_t)b * D + d] = a0[r];
    }
    for (int l = 0; l < L; ++l) {
        const float* wl = w + (size_t)l * D;
        const float* bl = bias + (size_t)l * D;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int d = lane + G * r;
            if (d < D) s += xl[r] * wl[d];
        }
        s = dcn_group_sum<G>(s);
        if (lane == 0) xlw[(size_t)l * B + b] = s;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int d = lane + G * r;
            if (d < D) {
                xl[r] = a0[r] * s + xl[r] + bl[d];
                xs[((size_t)(l + 1) * B + b) * D + d] = xl[r];
            }
        }
    }
}

int dcn_cross_fwd(const float* x0, int x0_ld, const float* w, const float* b, int B, int D, int L, float* xs,
                  float* xlw, hipStream_t st) {
    if (B <= 0) return DCTR_OK;
    const int nr = ceil_div(D, 64);
    dim3 grid(ceil_div(B, 4));
    if (nr <= 10) dcn_cross_fwd_kernel<10><<<grid, 256, 0, st>>>(x0, x0_ld, w, b, B, D, L, xs, xlw);
    else if (nr <= 20) dcn_cross_fwd_kernel<20><<<grid, 256, 0, st>>>(x0, x0_ld, w, b, B, D, L, xs, xlw);
    else if (nr <= 40) dcn_cross_fwd_kernel<40><<<grid, 256, 0, st>>>(x0, x0_ld, w, b, B, D, L, xs, xlw);
    else if (D <= 256 * 20) dcn_cross_fwd_kernel<20, 256><<<B, 256, 0, st>>>(x0, x0_ld, w, b, B, D, L, xs, xlw);       // one block per example
    else if (D <= 256 * 40) dcn_cross_fwd_kernel<40, 256><<<B, 256, 0, st>>>(x0, x0_ld, w, b, B, D, L, xs, xlw);
    else { set_error("dcn_cross: F*K=%d > 10240 unsupported", D); return DCTR_ERR_UNSUPPORTED; }
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

// backward per example (g = dL/dx_{l+1}):  t_l = g.x0 ; dx0 += g s_l ; g <- g + t_l w_l ; finally dx0 += g.
// G[l] = dL/dx_{l+1} and T[l] = t_l are written to scratch; db_l = colsum(G[l]), dw_l = colsum(T[l] * xs[l]).
template <int NR, int GS = 64>
__global__ __launch_bounds__(256) void dcn_cross_bwd_kernel(const float* __restrict__ xs, const float* __restrict__ xlw,
                                                           const float* __restrict__ w, const float* __restrict__ dxL, int dxl_ld,
                                                           int B, int D, int L, float* __restrict__ dx0, int dx0_ld,
                                                           float* __restrict__ G, float* __restrict__ T) {
    const int lane = threadIdx.x % GS;
    const int b = blockIdx.x * (256 / GS) + threadIdx.x / GS;
    if (b >= B) return;
    float a0[NR], g[NR], acc0[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int d = lane + GS * r;
        a0[r] = (d < D) ? xs[(size_t)b * D + d] : 0.f;
        g[r] = (d < D) ? dxL[(size_t)b * dxl_ld + d] : 0.f;
        acc0[r] = 0.f;
    }
    for (int l = L - 1; l >= 0; --l) {
        const float s = xlw[(size_t)l * B + b];
        const float* wl = w + (size_t)l * D;
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < NR; ++r) t += g[r] * a0[r];
        t = dcn_group_sum<GS>(t);
        if (lane == 0) T[(size_t)l * B + b] = t;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int d = lane + GS * r;
            if (d < D) {
                G[((size_t)l * B + b) * D + d] = g[r];
                acc0[r] += g[r] * s;
                g[r] += t * wl[d];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int d = lane + GS * r;
        if (d < D) dx0[(size_t)b * dx0_ld + d] += acc0[r] + g[r];
    }
}

// the cross parameters' gradient slabs out of what dcn_cross_bwd_kernel left in the scratch: column sums over the batch, 2 L small
// launches -- nothing on the way to dL/de needs them, so the training step runs them on its side stream
int dcn_cross_param_grads(const float* xs, int B, int D, int L, float* dw_part, float* db_part, int splits, int64_t part_stride,
                          const float* scratch, hipStream_t st) {
    if (B <= 0) return DCTR_OK;
    const float* G = scratch;                          // [L,B,D]
    const float* T = scratch + (size_t)L * B * D;      // [L,B]
    ColsumJobs J{};
    for (int l = 0; l < L; ++l) {
        // partial slabs: slab s of layer l at part + s*part_stride + l*D
        J.Y[J.n] = G + (size_t)l * B * D; J.rs[J.n] = nullptr; J.out[J.n] = db_part + (size_t)l * D; ++J.n;
        J.Y[J.n] = xs + (size_t)l * B * D; J.rs[J.n] = T + (size_t)l * B; J.out[J.n] = dw_part + (size_t)l * D; ++J.n;
        if (J.n + 2 > COLSUM_MAX_JOBS || l == L - 1) {
            DCTR_TRY(colsum_partials_batch(J, D, B, D, splits, part_stride, st));
            J.n = 0;
        }
    }
    return DCTR_OK;
}

// (dw_part == nullptr: the caller runs dcn_cross_param_grads itself, on a stream of its choice, behind this one)
int dcn_cross_bwd(const float* xs, const float* xlw, const float* w, const float* dxL, int dxl_ld, int B, int D, int L,
                  float* dx0, int dx0_ld, float* dw_part, float* db_part, int splits, int64_t part_stride,
                  float* scratch, hipStream_t st) {
    if (B <= 0) return DCTR_OK;
    float* G = scratch;                          // [L,B,D]
    float* T = scratch + (size_t)L * B * D;      // [L,B]
    const int nr = ceil_div(D, 64);
    dim3 grid(ceil_div(B, 4));
    if (nr <= 10) dcn_cross_bwd_kernel<10><<<grid, 256, 0, st>>>(xs, xlw, w, dxL, dxl_ld, B, D, L, dx0, dx0_ld, G, T);
    else if (nr <= 20) dcn_cross_bwd_kernel<20><<<grid, 256, 0, st>>>(xs, xlw, w, dxL, dxl_ld, B, D, L, dx0, dx0_ld, G, T);
    else if (nr <= 40) dcn_cross_bwd_kernel<40><<<grid, 256, 0, st>>>(xs, xlw, w, dxL, dxl_ld, B, D, L, dx0, dx0_ld, G, T);
    else if (D <= 256 * 20) dcn_cross_bwd_kernel<20, 256><<<B, 256, 0, st>>>(xs, xlw, w, dxL, dxl_ld, B, D, L, dx0, dx0_ld, G, T);
    else if (D <= 256 * 40) dcn_cross_bwd_kernel<40, 256><<<B, 256, 0, st>>>(xs, xlw, w, dxL, dxl_ld, B, D, L, dx0, dx0_ld, G, T);
    else { set_error("dcn_cross: F*K=%d > 10240 unsupported", D); return DCTR_ERR_UNSUPPORTED; }
    DCTR_LAUNCH_CHECK();
    if (dw_part == nullptr) return DCTR_OK;
    return dcn_cross_param_grads(xs, B, D, L, dw_part, db_part, splits, part_stride, scratch, st);
}

// ---- DCN cross network, the TRAINING STEP's pair of kernels (DCN.py:150-158 and its gradient).  The op-level kernels above keep every
// x_l ([L+1, B, D]) and hand dL/dx_{l+1} ([L, B, D]) to column-sum launches; at c3 that is 40 MB written by the forward and 30 MB
// written + 60 MB re-read behind the backward, for values that cost two FMAs to form.  Here
//   * the forward writes x_L and s_l = x_l . w_l only;
//   * the backward re-forms x_1 .. x_{L-1} from x_0, s_l, b_l in registers, walks the layers down, and accumulates
//     db_l = sum_b g_l and dw_l = sum_b t_l x_l in registers over the examples of its block; every block leaves ONE row of sums
//     ([blocks, L D] per parameter, <= 512 rows), which a column-sum launch on the side stream folds into the parameter's n_part
//     partial slabs.  (Blocks meeting in the slabs by float atomics was tried first: 1.9 M device-scope atomics from 512 blocks took
//     the kernel from ~20 to 60 us -- across XCDs they are performed at the memory side.)
// One group of GS lanes per example, float4 per lane (D % 4 == 0), NR float4 per lane: D <= 4 GS NR.
template <int GS>
__device__ __forceinline__ float dcn_group_sum_u(float v) {       // every thread of the block calls it (uniform control flow)
    v = wsum(v);
    if (GS == 64) return v;
    __shared__ float part[4];
    __syncthreads();
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (GS == 128) { const int w = (threadIdx.x >> 6) & ~1; return part[w] + part[w + 1]; }
    return part[0] + part[1] + part[2] + part[3];
}
__device__ __forceinline__ float dot4(const float4 a, const float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float4 fma4(const float4 a, float s, const float4 c) {        // a * s + c
    return make_float4(a.x * s + c.x, a.y * s + c.y, a.z * s + c.z, a.w * s + c.w);
}
__device__ __forceinline__ float4 add4(const float4 a, const float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

template <int NR, int GS, int L>
__global__ __launch_bounds__(256) void dcn_cross_fwd_lean_kernel(const float* __restrict__ x0, int x0_ld, const float* __restrict__ w,
                                                                const float* __restrict__ bias, int B, int D, float* __restrict__ xL,
                                                                float* __restrict__ xlw) {
    constexpr int NG = 256 / GS;
    const int lig = threadIdx.x % GS, grp = threadIdx.x / GS;
    const int D4 = D / 4;
    for (int row0 = blockIdx.x * NG; row0 < B; row0 += gridDim.x * NG) {
        const int b = row0 + grp;
        const bool live = b < B;
        float4 a0[NR], xl[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int d4 = lig + GS * r;
            a0[r] = (live && d4 < D4) ? reinterpret_cast<const float4*>(x0 + (size_t)b * x0_ld)[d4] : make_float4(0.f, 0.f, 0.f, 0.f);
            xl[r] = a0[r];
        }
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const float4* wl = reinterpret_cast<const float4*>(w + (size_t)l * D);
            const float4* bl = reinterpret_cast<const float4*>(bias + (size_t)l * D);
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int d4 = lig + GS * r;
                if (d4 < D4) s += dot4(xl[r], wl[d4]);
            }
            s = dcn_group_sum_u<GS>(s);
            if (live && lig == 0) xlw[(size_t)l * B + b] = s;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int d4 = lig + GS * r;
                if (d4 < D4) xl[r] = add4(fma4(a0[r], s, xl[r]), bl[d4]);
            }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int d4 = lig + GS * r;
            if (live && d4 < D4) reinterpret_cast<float4*>(xL + (size_t)b * D)[d4] = xl[r];
        }
    }
}

template <int NR, int GS, int L>
__global__ __launch_bounds__(256) void dcn_cross_bwd_fused_kernel(const float* __restrict__ x0, int x0_ld, const float* __restrict__ xlw,
                                                                 const float* __restrict__ w, const float* __restrict__ bias,
                                                                 const float* __restrict__ dxL, int dxl_ld, int B, int D,
                                                                 float* __restrict__ dx0, int dx0_ld, float* __restrict__ dw_rows,
                                                                 float* __restrict__ db_rows) {
    constexpr int NG = 256 / GS;
    const int lig = threadIdx.x % GS, grp = threadIdx.x / GS;
    const int D4 = D / 4;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 dw[L][NR], db[L][NR];
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int r = 0; r < NR; ++r) dw[l][r] = db[l][r] = zero;
    for (int row0 = blockIdx.x * NG; row0 < B; row0 += gridDim.x * NG) {
        const int b = row0 + grp;
        const bool live = b < B;
        float4 x[L][NR], g[NR], acc0[NR];     // x[l] = x_l (x[0] = x_0)
        float sl[L];
#pragma unroll
        for (int l = 0; l < L; ++l) sl[l] = live ? xlw[(size_t)l * B + b] : 0.f;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int d4 = lig + GS * r;
            const bool ok = live && d4 < D4;
            x[0][r] = ok ? reinterpret_cast<const float4*>(x0 + (size_t)b * x0_ld)[d4] : zero;
            g[r] = ok ? reinterpret_cast<const float4*>(dxL + (size_t)b * dxl_ld)[d4] : zero;
            acc0[r] = zero;
        }
#pragma unroll
        for (int l = 0; l + 1 < L; ++l) {       // x_{l+1} = x_0 s_l + x_l + b_l, as the forward formed it
            const float4* bl = reinterpret_cast<const float4*>(bias + (size_t)l * D);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int d4 = lig + GS * r;
                x[l + 1][r] = (live && d4 < D4) ? add4(fma4(x[0][r], sl[l], x[l][r]), bl[d4]) : zero;
            }
        }
#pragma unroll
        for (int l = L - 1; l >= 0; --l) {
            const float4* wl = reinterpret_cast<const float4*>(w + (size_t)l * D);
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < NR; ++r) t += dot4(g[r], x[0][r]);
            t = dcn_group_sum_u<GS>(t);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int d4 = lig + GS * r;
                if (live && d4 < D4) {
                    db[l][r] = add4(db[l][r], g[r]);
                    dw[l][r] = fma4(x[l][r], t, dw[l][r]);
                    acc0[r] = fma4(g[r], sl[l], acc0[r]);
                    g[r] = fma4(wl[d4], t, g[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int d4 = lig + GS * r;
            if (live && d4 < D4) {
                float4* p = reinterpret_cast<float4*>(dx0 + (size_t)b * dx0_ld) + d4;
                const float4 o = *p;
                *p = make_float4(o.x + acc0[r].x + g[r].x, o.y + acc0[r].y + g[r].y, o.z + acc0[r].z + g[r].z, o.w + acc0[r].w + g[r].w);
            }
        }
    }
    // the block's column sums -> its row of [blocks, L D] (groups of one block first, through LDS)
    __shared__ float4 red[256];
    float* wslab = dw_rows + (size_t)blockIdx.x * L * D;
    float* bslab = db_rows + (size_t)blockIdx.x * L * D;
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                float4 v = which ? db[l][r] : dw[l][r];
                if (NG > 1) {
                    __syncthreads();
                    red[threadIdx.x] = v;
                    __syncthreads();
                    if (grp == 0)
                        for (int q = 1; q < NG; ++q) v = add4(v, red[q * GS + lig]);
                }
                const int d4 = lig + GS * r;
                if (grp == 0 && d4 < D4) reinterpret_cast<float4*>((which ? bslab : wslab) + (size_t)l * D)[d4] = v;
            }
}

// -> false: shape outside what the pair covers (the caller keeps the op-level kernels)
bool dcn_cross_lean_ok(int D, int L) { return D % 4 == 0 && D <= 4 * 256 * 3 && L >= 1 && L <= 4; }
// blocks of the backward = rows of its two [rows, L D] outputs (two blocks per CU at most)
int dcn_cross_bwd_rows(int B, int D) {
    const int gs = D / 4 <= 64 * 3 ? 64 : (D / 4 <= 128 * 3 ? 128 : 256);
    return std::min(ceil_div(B, 256 / gs), 512);
}

template <int NR, int GS>
static int dcn_lean_fwd_L(const float* x0, int x0_ld, const float* w, const float* b, int B, int D, int L, float* xL, float* xlw, hipStream_t st) {
    const int grid = std::min(ceil_div(B, 256 / GS), 2048);
    switch (L) {
#define DCTR_L(LL) case LL: dcn_cross_fwd_lean_kernel<NR, GS, LL><<<grid, 256, 0, st>>>(x0, x0_ld, w, b, B, D, xL, xlw); break
        DCTR_L(1); DCTR_L(2); DCTR_L(3); DCTR_L(4);
#undef DCTR_L
        default: set_error("dcn_cross (lean): %d layers", L); return DCTR_ERR_UNSUPPORTED;
    }
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}
template <int NR, int GS>
static int dcn_lean_bwd_L(const float* x0, int x0_ld, const float* xlw, const float* w, const float* bias, const float* dxL, int dxl_ld,
                          int B, int D, int L, float* dx0, int dx0_ld, float* dw_rows, float* db_rows, hipStream_t st) {
    const int grid = dcn_cross_bwd_rows(B, D);
    switch (L) {
#define DCTR_L(LL) case LL: DCTR_LAUNCH_RIDE((dcn_cross_bwd_fused_kernel<NR, GS, LL>), dim3(grid), dim3(256), 0u, st, x0, x0_ld, xlw, w, bias, dxL, dxl_ld, B, D, dx0, dx0_ld, dw_rows, db_rows); break
        DCTR_L(1); DCTR_L(2); DCTR_L(3); DCTR_L(4);
#undef DCTR_L
        default: set_error("dcn_cross (lean): %d layers", L); return DCTR_ERR_UNSUPPORTED;
    }
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

// forward of a training or inference step: x_L [B, D] and s [L, B] only
int dcn_cross_fwd_lean(const float* x0, int x0_ld, const float* w, const float* b, int B, int D, int L, float* xL, float* xlw, hipStream_t st) {
    if (B <= 0) return DCTR_OK;
    DCTR_REQUIRE(dcn_cross_lean_ok(D, L) && x0_ld % 4 == 0, "dcn_cross (lean): D=%d L=%d ld=%d", D, L, x0_ld);
    const int D4 = D / 4;
    if (D4 <= 64 * 3) return dcn_lean_fwd_L<3, 64>(x0, x0_ld, w, b, B, D, L, xL, xlw, st);
    if (D4 <= 128 * 3) return dcn_lean_fwd_L<3, 128>(x0, x0_ld, w, b, B, D, L, xL, xlw, st);
    return dcn_lean_fwd_L<3, 256>(x0, x0_ld, w, b, B, D, L, xL, xlw, st);
}

// backward: dx0 += dL/dx_0 through the cross network; every block's sums of the cross parameters' gradients go to row `block` of
// dw_rows / db_rows [dcn_cross_bwd_rows(B, D), L D]
int dcn_cross_bwd_fused(const float* x0, int x0_ld, const float* xlw, const float* w, const float* bias, const float* dxL, int dxl_ld, int B,
                        int D, int L, float* dx0, int dx0_ld, float* dw_rows, float* db_rows, hipStream_t st) {
    if (B <= 0) return DCTR_OK;
    DCTR_REQUIRE(dcn_cross_lean_ok(D, L) && x0_ld % 4 == 0 && dxl_ld % 4 == 0 && dx0_ld % 4 == 0, "dcn_cross (lean): D=%d L=%d", D, L);
    const int D4 = D / 4;
    if (D4 <= 64 * 3) return dcn_lean_bwd_L<3, 64>(x0, x0_ld, xlw, w, bias, dxL, dxl_ld, B, D, L, dx0, dx0_ld, dw_rows, db_rows, st);
    if (D4 <= 128 * 3) return dcn_lean_bwd_L<3, 128>(x0, x0_ld, xlw, w, bias, dxL, dxl_ld, B, D, L, dx0, dx0_ld, dw_rows, db_rows, st);
    return dcn_lean_bwd_L<3, 256>(x0, x0_ld, xlw, w, bias, dxL, dxl_ld, B, D, L, dx0, dx0_ld, dw_rows, db_rows, st);
}

// ... and those rows folded into the parameters' partial slabs (slab s of layer l at part + s * part_stride + l * D): one launch
int dcn_cross_param_slabs(const float* dw_rows, const float* db_rows, int B, int D, int L, float* dw_part, float* db_part, int n_part,
                          int64_t part_stride, hipStream_t st) {
    if (B <= 0) return DCTR_OK;
    ColsumJobs J{};
    J.Y[0] = dw_rows; J.rs[0] = nullptr; J.out[0] = dw_part;
    J.Y[1] = db_rows; J.rs[1] = nullptr; J.out[1] = db_part;
    J.n = 2;
    return colsum_partials_batch(J, L * D, dcn_cross_bwd_rows(B, D), L * D, n_part, part_stride, st);
}

// ---- DeepMVM "all-order" product (DeepMVM.py:144-150): x_mvm[b,k] = prod_f (e[b,f,k] + mvm_b[f,k]) -----------------------------
// One lane per (example, k): the F factors of a lane are strided K apart in e, consecutive lanes read consecutive k.
constexpr int MVM_MAXF = 64;

__global__ __launch_bounds__(256) void mvm_fwd_kernel(const float* __restrict__ e, int e_ld, const float* __restrict__ mb, int B, int F,
                                                     int K, float* __restrict__ xm) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * K) return;
    const int b = (int)(i / K), k = (int)(i % K);
    float p = 1.f;
    for (int f = 0; f < F; ++f) p *= e[(size_t)b * e_ld + f * K + k] + mb[f * K + k];
    xm[(size_t)b * K + k] = p;
}

// backward: with a_f = e_f + mvm_b_f,  d a_f = dxm * prod_{g != f} a_g  (prefix/suffix products: no division, zeros are fine);
// dE[b,f,k] += d a_f,  d mvm_b[f,k] = sum_b d a_f  (block-level LDS accumulation, one partial slab per block).
__global__ __launch_bounds__(256) void mvm_bwd_kernel(const float* __restrict__ e, int e_ld, const float* __restrict__ mb,
                                                     const float* __restrict__ dxm, int B, int F, int K, int rows_per_block,
                                                     float* __restrict__ dE, int de_ld, float* __restrict__ dmb_part, int64_t part_stride) {
    extern __shared__ float acc[];          // [F*K]
    for (int i = threadIdx.x; i < F * K; i += blockDim.x) acc[i] = 0.f;
    __syncthreads();
    const int epb = blockDim.x / K;         // examples per pass
    const int k = threadIdx.x % K, bl = threadIdx.x / K;
    const int bbeg = blockIdx.x * rows_per_block, bend = min(B, bbeg + rows_per_block);
    if (bl < epb) {
        for (int b = bbeg + bl; b < bend; b += epb) {
            float a[MVM_MAXF];
            float p = 1.f;
            for (int f = 0; f < F; ++f) { a[f] = e[(size_t)b * e_ld + f * K + k] + mb[f * K + k]; }
            // prefix products in place of a: pre[f] = prod_{g<f} a_g ; keep a suffix running product
            float pre[MVM_MAXF];
            for (int f = 0; f < F; ++f) { pre[f] = p; p *= a[f]; }
            const float g = dxm[(size_t)b * K + k];
            float suf = 1.f;
            for (int f = F - 1; f >= 0; --f) {
                const float d = g * pre[f] * suf;
                suf *= a[f];
                dE[(size_t)b * de_ld + f * K + k] += d;
                atomicAdd(&acc[f * K + k], d);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < F * K; i += blockDim.x) dmb_part[(size_t)blockIdx.x * part_stride + i] = acc[i];
}

int mvm_fwd(const float* e, int e_ld, const float* mb, int B, int F, int K, float* xm, hipStream_t st) {
    if (B <= 0) return DCTR_OK;
    mvm_fwd_kernel<<<ceil_div((int64_t)B * K, 256), 256, 0, st>>>(e, e_ld, mb, B, F, K, xm);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

int mvm_bwd(const float* e, int e_ld, const float* mb, const float* dxm, int B, int F, int K, float* dE, int de_ld, float* dmb_part,
            int64_t part_stride, int splits, hipStream_t st) {
    DCTR_REQUIRE(F <= MVM_MAXF && K <= 256 && F * K * 4 <= 64 * 1024, "DeepMVM backward: field_size <= %d and F*K <= 16384 supported", MVM_MAXF);
    if (B <= 0) return DCTR_OK;
    mvm_bwd_kernel<<<splits, 256, (size_t)F * K * sizeof(float), st>>>(e, e_ld, mb, dxm, B, F, K, ceil_div(B, splits), dE, de_ld, dmb_part,
                                                                     part_stride);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

}  // namespace dctr

using namespace dctr;

extern "C" {

int dctr_pnn_inner_fwd(const float* d_e, int e_ld, int B, int F, int K, float* d_ip, int ip_ld, void* stream) {
    return pnn_inner_fwd(d_e, e_ld, B, F, K, d_ip, ip_ld, as_stream(stream));
}
int dctr_pnn_inner_bwd(const float* d_e, int e_ld, const float* d_dip, int dip_ld, int B, int F, int K, float* d_dE,
                       int de_ld, void* stream) {
    return pnn_inner_bwd(d_e, e_ld, d_dip, dip_ld, B, F, K, d_dE, de_ld, as_stream(stream));
}
int dctr_pnn_outer_fwd(const float* d_e, int e_ld, int B, int F, int K, float* d_op, int64_t op_ld, void* stream) {
    return pnn_outer_fwd(d_e, e_ld, B, F, K, d_op, op_ld, as_stream(stream));
}
int dctr_pnn_outer_bwd(const float* d_e, int e_ld, const float* d_dop, int64_t dop_ld, int B, int F, int K, float* d_dE,
                       int de_ld, void* stream) {
    return pnn_outer_bwd(d_e, e_ld, d_dop, dop_ld, B, F, K, d_dE, de_ld, as_stream(stream));
}
int dctr_dcn_cross_fwd(const float* d_x0, int x0_ld, const float* d_w, const float* d_b, int B, int D, int L,
                       float* d_xs, float* d_xlw, void* stream) {
    return dcn_cross_fwd(d_x0, x0_ld, d_w, d_b, B, D, L, d_xs, d_xlw, as_stream(stream));
}
int dctr_dcn_cross_bwd(const float* d_xs, const float* d_xlw, const float* d_w, const float* d_dxL, int dxl_ld, int B,
                       int D, int L, float* d_dx0, int dx0_ld, float* d_dw, float* d_db, float* d_workspace,
                       size_t workspace_bytes, void* stream) {
    // workspace: G [L,B,D] + T [L,B]; single-slab column sums straight into d_dw / d_db ([L,D] each)
    const size_t need = ((size_t)L * B * D + (size_t)L * B) * sizeof(float);
    DCTR_REQUIRE(d_workspace != nullptr && workspace_bytes >= need, "dcn_cross_bwd: workspace of %zu bytes required", need);
    return dcn_cross_bwd(d_xs, d_xlw, d_w, d_dxL, dxl_ld, B, D, L, d_dx0, dx0_ld, d_dw, d_db, 1, 0, d_workspace,
                         as_stream(stream));
}

}  // extern "C"
