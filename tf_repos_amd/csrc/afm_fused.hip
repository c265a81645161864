// AFM attention network fused over the pair rows (AFM.py:142-147): for every pair row r = (b, p)
//     s[r] = b_o + sum_a relu( sum_k pp[r,k] W[k,a] + b_a[a] ) * w_o[a]
// without materialising the [B*P, A] hidden activations (3.1 GB at B=4096, A=256, written once and read three times by the
// unfused path).  MFMA-bound: 2*K*A flops per row on v_mfma_f32_32x32x2_f32 against K*4 bytes of pp.
//   block = 4 waves; a wave owns 32 pair rows at a time (persistent loop over row tiles), W / b_a / w_o live in LDS.
//   per 32-row tile and 32-column chunk of the hidden layer: K/2 MFMAs (A operand = pp fragment kept in registers for all
//   chunks, B operand = W chunk from LDS), then bias + ReLU + the dot with w_o on the accumulator registers; the per-row partial
//   sums of all chunks are reduced across lanes once per tile.
// The backward recomputes the hidden chunk the same way and feeds it straight into the two products that need it
// (d pp = d_ah W^T and dW = pp^T d_ah, both on MFMA) plus the three column sums (db_a, dw_o via ah, db_o).
#include "engine.h"

namespace dctr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int AFM_MAXK = 32;        // fused path: embedding_size <= 32 (the pp fragment and the dW accumulators live in registers)

// C/D layout of the 32x32 MFMA: register r of lane l holds row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

template <int K>
__global__ __launch_bounds__(256) void afm_att_fwd_kernel(const float* __restrict__ pp, const float* __restrict__ W,
                                                         const float* __restrict__ ba, const float* __restrict__ wo,
                                                         const float* __restrict__ bo, int64_t rows, int A, float* __restrict__ sc) {
    extern __shared__ float lds[];                  // W [K][AP] | ba [AP] | wo [AP], AP = A rounded up to 32 (zero padded)
    const int AP = (A + 31) & ~31;
    float* Wl = lds;
    float* bal = lds + (size_t)K * AP;
    float* wol = bal + AP;
    for (int i = threadIdx.x; i < K * AP; i += 256) { const int k = i / AP, a = i - k * AP; Wl[i] = a < A ? W[(size_t)k * A + a] : 0.f; }
    for (int i = threadIdx.x; i < AP; i += 256) { bal[i] = i < A ? ba[i] : 0.f; wol[i] = i < A ? wo[i] : 0.f; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, col = lane & 31;
    const float bias_o = bo[0];
    const int64_t n_tiles = (rows + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t row = tile * 32 + col;            // the pair row this lane feeds as A operand
        // A fragment: pp[row][k] for k = 2j + half
        float a[K / 2];
        if (row < rows) {
            const float4* pr = reinterpret_cast<const float4*>(pp + (size_t)row * K);
#pragma unroll
            for (int q = 0; q < K / 4; ++q) {
                const float4 v = pr[q];
                a[2 * q] = half ? v.y : v.x;            // k = 4q + half, 4q + 2 + half
                a[2 * q + 1] = half ? v.w : v.z;
            }
        } else {
#pragma unroll
            for (int j = 0; j < K / 2; ++j) a[j] = 0.f;
        }
        float srow[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) srow[r] = 0.f;
        for (int c = 0; c < AP; c += 32) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int j = 0; j < K / 2; ++j)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], Wl[(size_t)(2 * j + half) * AP + c + col], acc, 0, 0, 0);
            const float bb = bal[c + col], ww = wol[c + col];
#pragma unroll
            for (int r = 0; r < 16; ++r) srow[r] += fmaxf(acc[r] + bb, 0.f) * ww;
        }
        // sum over the 32 columns held by the 32 lanes of each half
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = srow[r];
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);
            srow[r] = v;
        }
        if (col == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t rr = tile * 32 + acc_row(r, half);
                if (rr < rows) sc[rr] = srow[r] + bias_o;
            }
        }
    }
}

int afm_att_fwd(const float* pp, const float* W, const float* ba, const float* wo, const float* bo, int64_t rows, int K, int A, float* sc,
                hipStream_t st) {
    DCTR_REQUIRE(K % 4 == 0 && K <= AFM_MAXK && A >= 1, "fused AFM attention: K=%d (<= %d, multiple of 4) / A=%d unsupported", K, AFM_MAXK, A);
    const size_t AP = (size_t)((A + 31) & ~31);
    const size_t lds = ((size_t)K * AP + 2 * AP) * sizeof(float);
    DCTR_REQUIRE(lds <= 64 * 1024, "fused AFM attention: K*A too large for LDS staging");
    if (rows <= 0) return DCTR_OK;
    const int grid = (int)std::min<int64_t>((rows + 127) / 128, 256 * 4);
    switch (K) {
#define DCTR_F(KK) case KK: afm_att_fwd_kernel<KK><<<grid, 256, lds, st>>>(pp, W, ba, wo, bo, rows, A, sc); break
        DCTR_F(4); DCTR_F(8); DCTR_F(12); DCTR_F(16); DCTR_F(20); DCTR_F(24); DCTR_F(28); DCTR_F(32);
#undef DCTR_F
        default: set_error("fused AFM attention: K=%d unsupported", K); return DCTR_ERR_UNSUPPORTED;
    }
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

// ---- backward --------------------------------------------------------------------------------------------------------
// Inputs: pp [rows,K], dsc [rows] = dL/ds.  A group of WPG waves owns one 32-row tile at a time; each wave of the group owns CPW
// of the NCH hidden chunks (so its dW accumulators are CPW x 8|16 registers and several blocks fit a CU).  Per tile and chunk the
// wave recomputes ah = relu(pp W + b_a), forms d_ah = dsc * w_o * (ah > 0) on the accumulator registers, passes it through a
// wave-private LDS tile to re-read it in operand layout, and feeds it to  d pp += d_ah W^T  and  dW += pp^T d_ah  (both MFMA).
// The per-wave d pp partials are summed through LDS and stored coalesced.  dW, db_a, dw_o (= sum ah dsc) and db_o (= sum dsc)
// stay in registers for the whole kernel and are added into `n_slabs` partial slabs at the end.
template <int K, int NCH>
__global__ __launch_bounds__(256) void afm_att_bwd_kernel(const float* __restrict__ pp, const float* __restrict__ W,
                                                         const float* __restrict__ ba, const float* __restrict__ wo,
                                                         const float* __restrict__ dsc, int64_t rows, int A, float* __restrict__ dpp2,
                                                         float* __restrict__ dW_part, int64_t dW_stride, float* __restrict__ dba_part,
                                                         int64_t dba_stride, float* __restrict__ dwo_part, int64_t dwo_stride,
                                                         float* __restrict__ dbo_part, int64_t dbo_stride, int n_slabs) {
    constexpr int AP = 32 * NCH;
    constexpr int WPG = NCH < 4 ? NCH : 4;          // waves per group (one tile per group at a time)
    constexpr int G = 4 / WPG;                      // groups per block
    constexpr int CPW = NCH / WPG;                  // hidden chunks per wave
    constexpr int KS = K + 1;                       // LDS row stride of the pp / d pp tiles
    // K <= 16: d pp and dW go through the 16x16x4 MFMA (a 32x32 output block would be half padding): 2 blocks x 4 registers each
    constexpr bool SMALL = K <= 16;
    constexpr int NACC = SMALL ? 8 : 16;
    extern __shared__ float lds[];                  // W [K][AP] | ba | wo | per wave T [32][33], R [32][KS] | per group P [32][KS], S [32]
    float* Wl = lds;
    float* bal = Wl + (size_t)K * AP;
    float* wol = bal + AP;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = wave / WPG, wi = wave % WPG, gl = (wave % WPG) * 64 + lane;       // group, wave in group, lane in group
    float* T = wol + AP + (size_t)wave * (32 * 33 + 32 * KS);
    float* R = T + 32 * 33;
    float* P = wol + AP + (size_t)4 * (32 * 33 + 32 * KS) + (size_t)grp * (32 * KS + 32);
    float* S = P + 32 * KS;
    float* Rg = wol + AP + (size_t)(grp * WPG) * (32 * 33 + 32 * KS) + 32 * 33;      // R of the group's first wave (stride 32*33 + 32*KS)
    for (int i = threadIdx.x; i < K * AP; i += 256) { const int k = i / AP, a = i - k * AP; Wl[i] = a < A ? W[(size_t)k * A + a] : 0.f; }
    for (int i = threadIdx.x; i < AP; i += 256) { bal[i] = i < A ? ba[i] : 0.f; wol[i] = i < A ? wo[i] : 0.f; }
    const int half = lane >> 5, col = lane & 31, l15 = lane & 15, l4 = lane >> 4;
    float dWacc[CPW][NACC];
    float dba_acc[CPW], dwo_acc[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        dba_acc[c] = dwo_acc[c] = 0.f;
#pragma unroll
        for (int r = 0; r < NACC; ++r) dWacc[c][r] = 0.f;
    }
    float dbo_acc = 0.f;
    const int64_t n_tiles = (rows + 31) / 32;
    const int64_t n_iter = (n_tiles + (int64_t)gridDim.x * G - 1) / ((int64_t)gridDim.x * G);      // block-uniform trip count
    for (int64_t it = 0; it < n_iter; ++it) {
        const int64_t tile = (it * gridDim.x + blockIdx.x) * G + grp;           // (tiles past the end run masked: zero rows)
        const int64_t row0 = tile * 32;
        __syncthreads();                                // the previous tile's P / S / R are no longer read
        for (int i = gl; i < 32 * (K / 4); i += WPG * 64) {
            const int r = i / (K / 4), q = i - r * (K / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row0 + r < rows) v = reinterpret_cast<const float4*>(pp + (size_t)(row0 + r) * K)[q];
            float* d = P + r * KS + 4 * q;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        if (gl < 32) {
            const float dv = row0 + gl < rows ? dsc[row0 + gl] : 0.f;
            S[gl] = dv;
            dbo_acc += dv;
        }
        __syncthreads();
        float a[K / 2];
#pragma unroll
        for (int j = 0; j < K / 2; ++j) a[j] = P[col * KS + 2 * j + half];
        float dscr[16];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const float4 v = *reinterpret_cast<const float4*>(S + 8 * r4 + 4 * half);       // rows acc_row(4*r4 .. 4*r4+3, half)
            dscr[4 * r4 + 0] = v.x; dscr[4 * r4 + 1] = v.y; dscr[4 * r4 + 2] = v.z; dscr[4 * r4 + 3] = v.w;
        }
        f32x16 dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[r] = 0.f;
        f32x4 dps[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int cc = 0; cc < CPW; ++cc) {
            const int c = wi * CPW + cc;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int j = 0; j < K / 2; ++j)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], Wl[(size_t)(2 * j + half) * AP + 32 * c + col], acc, 0, 0, 0);
            const float bb = bal[32 * c + col], ww = wol[32 * c + col];
            __builtin_amdgcn_wave_barrier();            // (the previous chunk's reads of T are done: in-order LDS queue per wave)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float ah = fmaxf(acc[r] + bb, 0.f);
                const float g = ah > 0.f ? dscr[r] * ww : 0.f;
                dwo_acc[cc] += ah * dscr[r];
                dba_acc[cc] += g;
                T[acc_row(r, half) * 33 + col] = g;     // d_ah chunk [32 rows][32 cols]
            }
            __builtin_amdgcn_wave_barrier();
            if constexpr (SMALL) {
                // d pp [32 rows x 16] += d_ah [32 x 32] W_chunk^T [32 x 16]: two 16-row blocks, 8 steps of 4 (columns >= K fed zeros)
#pragma unroll
                for (int s4 = 0; s4 < 8; ++s4) {
                    const float bop = l15 < K ? Wl[(size_t)l15 * AP + 32 * c + 4 * s4 + l4] : 0.f;
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
                        dps[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(T[(16 * rb + l15) * 33 + 4 * s4 + l4], bop, dps[rb], 0, 0, 0);
                }
                // dW_chunk [16 x 32] += pp^T [16 x 32 rows] d_ah [32 rows x 32]: two 16-column blocks (rows >= K fed zeros)
                f32x4 w0 = {dWacc[cc][0], dWacc[cc][1], dWacc[cc][2], dWacc[cc][3]};
                f32x4 w1 = {dWacc[cc][4], dWacc[cc][5], dWacc[cc][6], dWacc[cc][7]};
#pragma unroll
                for (int s4 = 0; s4 < 8; ++s4) {
                    const float aop = l15 < K ? P[(4 * s4 + l4) * KS + l15] : 0.f;
                    w0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aop, T[(4 * s4 + l4) * 33 + l15], w0, 0, 0, 0);
                    w1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aop, T[(4 * s4 + l4) * 33 + 16 + l15], w1, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) { dWacc[cc][r] = w0[r]; dWacc[cc][4 + r] = w1[r]; }
            } else {
                // d pp [32 rows x K] += d_ah [32 x 32] W_chunk^T [32 x K]
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float bop = col < K ? Wl[(size_t)col * AP + 32 * c + 2 * j + half] : 0.f;
                    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(T[col * 33 + 2 * j + half], bop, dp, 0, 0, 0);
                }
                // dW_chunk [K x 32] += pp^T [K x 32 rows] d_ah [32 rows x 32]
                f32x16 w;
#pragma unroll
                for (int r = 0; r < 16; ++r) w[r] = dWacc[cc][r];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float aop = col < K ? P[(2 * j + half) * KS + col] : 0.f;
                    w = __builtin_amdgcn_mfma_f32_32x32x2f32(aop, T[(2 * j + half) * 33 + col], w, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) dWacc[cc][r] = w[r];
            }
        }
        // this wave's d pp partial -> R (accumulator column = k), then the group sums its WPG partials and stores the tile
        if constexpr (SMALL) {
            if (l15 < K) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) R[(16 * rb + 4 * l4 + r) * KS + l15] = dps[rb][r];
            }
        } else if (col < K) {
#pragma unroll
            for (int r = 0; r < 16; ++r) R[acc_row(r, half) * KS + col] = dp[r];
        }
        __syncthreads();
        for (int i = gl; i < 32 * K; i += WPG * 64) {
            const int r = i / K, k = i - r * K;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < WPG; ++w) v += Rg[(size_t)w * (32 * 33 + 32 * KS) + r * KS + k];
            if (row0 + r < rows) dpp2[(size_t)(row0 + r) * K + k] = v;
        }
    }
    // partial sums of this wave -> slab (waves are spread over the slabs; a few adds per address)
    const int slab = (int)((blockIdx.x * 4 + wave) % n_slabs);
#pragma unroll
    for (int cc = 0; cc < CPW; ++cc) {
        const int c = wi * CPW + cc;
        if constexpr (SMALL) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int k = 4 * l4 + (r & 3), acol = 32 * c + 16 * (r >> 2) + l15;
                if (k < K && acol < A) atomicAdd(&dW_part[(size_t)slab * dW_stride + (size_t)k * A + acol], dWacc[cc][r]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = acc_row(r, half), acol = 32 * c + col;
                if (k < K && acol < A) atomicAdd(&dW_part[(size_t)slab * dW_stride + (size_t)k * A + acol], dWacc[cc][r]);
            }
        }
        const float vb = dba_acc[cc] + __shfl_xor(dba_acc[cc], 32);      // the two halves hold the same column, different rows
        const float vw = dwo_acc[cc] + __shfl_xor(dwo_acc[cc], 32);
        if (half == 0 && 32 * c + col < A) {
            atomicAdd(&dba_part[(size_t)slab * dba_stride + 32 * c + col], vb);
            atomicAdd(&dwo_part[(size_t)slab * dwo_stride + 32 * c + col], vw);
        }
    }
    float vo = dbo_acc;                                 // (non-zero in lanes 0..31 of the group's first wave)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vo += __shfl_xor(vo, o);
    if (lane == 0 && wi == 0) atomicAdd(&dbo_part[(size_t)slab * dbo_stride], vo);
}

int afm_att_bwd(const float* pp, const float* W, const float* ba, const float* wo, const float* dsc, int64_t rows, int K, int A, float* dpp2,
                float* dW_part, int64_t dW_stride, float* dba_part, int64_t dba_stride, float* dwo_part, int64_t dwo_stride, float* dbo_part,
                int64_t dbo_stride, int n_slabs, hipStream_t st) {
    DCTR_REQUIRE(afm_fused_supported(K, A), "fused AFM attention backward: K=%d / A=%d unsupported", K, A);
    // the slabs are accumulated with atomics: clear them first
    DCTR_HIP_CHECK(hipMemsetAsync(dW_part, 0, sizeof(float) * ((size_t)(n_slabs - 1) * dW_stride + (size_t)K * A), st));
    DCTR_HIP_CHECK(hipMemsetAsync(dba_part, 0, sizeof(float) * ((size_t)(n_slabs - 1) * dba_stride + A), st));
    DCTR_HIP_CHECK(hipMemsetAsync(dwo_part, 0, sizeof(float) * ((size_t)(n_slabs - 1) * dwo_stride + A), st));
    DCTR_HIP_CHECK(hipMemsetAsync(dbo_part, 0, sizeof(float) * ((size_t)(n_slabs - 1) * dbo_stride + 1), st));
    if (rows <= 0) return DCTR_OK;
    const int nch = (A + 31) / 32;
    const int NCH = nch <= 1 ? 1 : (nch <= 2 ? 2 : (nch <= 4 ? 4 : 8));
    const int G = NCH < 4 ? 4 / NCH : 1;
    const size_t lds = ((size_t)K * 32 * NCH + 2 * 32 * NCH + 4 * (32 * 33 + 32 * (K + 1)) + (size_t)G * (32 * (K + 1) + 32)) * sizeof(float);
    static const int per_cu = getenv("DCTR_AFM_BWD_BLOCKS") ? atoi(getenv("DCTR_AFM_BWD_BLOCKS")) : 2;
    const int grid = (int)std::min<int64_t>(((rows + 31) / 32 + G - 1) / G, (int64_t)256 * per_cu);
#define DCTR_B(KK, NN)                                                                                                              \
    if (K == KK && NCH == NN) {                                                                                                     \
        afm_att_bwd_kernel<KK, NN><<<grid, 256, lds, st>>>(pp, W, ba, wo, dsc, rows, A, dpp2, dW_part, dW_stride, dba_part, dba_stride, \
                                                           dwo_part, dwo_stride, dbo_part, dbo_stride, n_slabs);                   \
        DCTR_LAUNCH_CHECK();                                                                                                        \
        return DCTR_OK;                                                                                                             \
    }
    DCTR_B(4, 1) DCTR_B(4, 2) DCTR_B(4, 4) DCTR_B(4, 8) DCTR_B(8, 1) DCTR_B(8, 2) DCTR_B(8, 4) DCTR_B(8, 8)
    DCTR_B(16, 1) DCTR_B(16, 2) DCTR_B(16, 4) DCTR_B(16, 8) DCTR_B(32, 1) DCTR_B(32, 2) DCTR_B(32, 4) DCTR_B(32, 8)
#undef DCTR_B
    set_error("fused AFM attention backward: K=%d unsupported", K);
    return DCTR_ERR_UNSUPPORTED;
}

bool afm_fused_supported(int K, int A) {
    return (K == 4 || K == 8 || K == 16 || K == 32) && A >= 1 && A <= 256;      // (dW accumulators: A/32 <= 8 chunks of 16 registers)
}

}  // namespace dctr
