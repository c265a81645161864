// K6, second kernel family: "direct-to-register, wave-split-K" exact-f32 MFMA GEMM for the SMALL dense products of the CTR
// step (4096 x 400 x 624 and friends: ~2 GFLOP, i.e. 13 us of matrix-core time -- one 80x80 output patch per CU).
// Replaces contrib.layers.fully_connected forward (DeepFM.py:156-158,165-166) and its MatMul gradients (DeepFM.py:213) like
// gemm.hip's LDS-tiled kernel does; the host-side chooser (gemm.hip) picks per shape.
//
// Why a second design.  A 64x64-tile kernel cuts 4096x400 into 448 tiles for 256 CUs (1.75 waves of tiles, N padded to 448);
// here ONE block per CU owns a (16 TM) x (16 TN) tile chosen so that the grid is <= 256 blocks and as close to 256 as the
// shape allows (32 x 208: 128 x 2 = 256 blocks, 26 of the 25x256/256 = 25 ideal 16x16 tiles per CU = 96 %).  The four
// waves of a block (one per SIMD) split the REDUCTION range in four contiguous quarters; each wave owns the whole tile
// (TM x TN accumulators of v_mfma_f32_16x16x4_f32, bitwise an fmaf chain) and streams ITS k-slices of both operands
// straight from L2 into MFMA fragments -- no LDS staging, no barrier in the main loop, every operand element enters the
// CU exactly once.  The four partial tiles meet in LDS at the end; the epilogue re-stages the tile row-major so that the
// global stores are whole rows (float4 per lane).
//
// Fragment / k assignment.  16x16x4: lane l = (c = l & 15, q = l >> 4) supplies A[row c][k_q] and B[k_q][col c].  The order in
// which k is consumed is free, so within a group of 16 consecutive k lane-quarter q takes k = 4 q + s at step s: an operand
// whose reduction dimension is contiguous in memory ("RC": X[M,K] for fwd/dgrad, W[Kin,N] read as B^T for dgrad) gets its four
// steps with ONE 16-byte load per lane; the other layout ("NC": W[K,N] for fwd, X and dY for wgrad) takes one dword per step,
// 16 lanes = 64 contiguous bytes.  All loads are buffer loads: the 15-20 lane offsets are computed once, the per-group part
// is a scalar offset, so the main loop has no address arithmetic at all; num_records is the true end of the operand, so whatever
// a clamped tail read or an empty wave addresses beyond it comes back as 0.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>
#include <utility>

#ifdef DR_STAMPS            // tools/gemm_dr_probe.hip only: cycle stamps per wave at the phase boundaries
#define DR_STAMP(i) do { if (lane == 0) dr_stamps[((blockIdx.y * gridDim.x + blockIdx.x) * 4 + w) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
extern __device__ long long dr_stamps[];
#else
#define DR_STAMP(i) do { } while (0)
#endif

namespace dctr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

enum { DR_STORE = 0, DR_BIAS_ACT = 1, DR_MASK = 2 };
// A operand generated on the fly (Outer-PNN, PNN.py:139-153 'Outer': the [B, P K K] product tensor is never written):
//   DR_AGEN_OUTER_FWD    A[b][(p,a,c)] = e[b][i_p][a] e[b][j_p][c], reduction over (p,a,c)   (first MLP layer forward)
//   DR_AGEN_OUTER_WGRAD  A^T[(p,a,c)][b], reduction over b                                   (its weight gradient)
//   DR_BGATE_WGRAD       B (NC, = the layer's ReLU OUTPUT h [rows, N]) enters as the gradient it implies under a rank-one output
//                        gradient: B'[row][n] = rowscale[row] 1[h[row][n] > 0]; the stores scale column n by colscale[n]; besides
//                        the column sums of B' (the bias gradient / colscale) a second set, sum_row rowscale[row] h[row][n], leaves
//                        in colsum2 -- the weight gradient of the (N -> 1) layer above.  AFM's last attention layer at K = 256:
//                        d ah = d score (x) w_o . 1[ah > 0] (AFM.py:147) is 3.1 GB that is neither written nor read.
enum { DR_AGEN_NONE = 0, DR_AGEN_OUTER_FWD = 1, DR_AGEN_OUTER_WGRAD = 2, DR_BGATE_WGRAD = 3 };

struct DrOuter {
    const float* e;             // gathered embeddings [rows][F K], row stride e_ld
    int e_ld;
    int rows;                   // true batch rows (the reduction length handed to the kernel may be rounded up: rows beyond read as 0)
    const int* pairs;           // [P] field pairs, i << 16 | j (i < j), the reference's double loop order
    int logk;                   // K = 1 << logk, K >= 16: a group of 16 reduction steps never straddles two (p, a)
};

struct DrEpilogue {
    const float* bias;          // DR_BIAS_ACT: [N] or null
    int relu;
    float keep;                 // dropout keep_prob of this layer (1 = off)
    uint64_t seed;
    const uint64_t* seed_ptr;   // per-step seed in device memory (graph replay), may be null
    const float* act;           // DR_MASK: stored output of the producing layer (ReLU mask), row stride ldact
    int ldact;
    float inv_keep;
    int64_t split_stride;       // DR_STORE with gridDim.y > 1: C + y * split_stride
    float* colsum;              // wgrad: column sums of B (= dY) over this block's reduction range -> colsum[y * colsum_stride + n]
    int64_t colsum_stride;
    const float* rowscale;      // DR_BGATE_WGRAD: [K] (the reduction runs over rows)
    const float* colscale;      // DR_BGATE_WGRAD: [N]
    float* colsum2;             // DR_BGATE_WGRAD: -> colsum2[y * colsum2_stride + n]
    int64_t colsum2_stride;
    // Producer-side split (round 6): besides C, the STORED output (bias / ReLU / dropout or the ReLU mask applied) leaves as three bf16
    // planes blocked along its ROWS, cf[plane][row / 8][N][8] -- the form a weight-gradient product reads both of its operands in (the
    // reduction runs over the batch rows: a lane's 8 k of an MFMA fragment are one 16-byte load, dr3 A_PRE / B_PRE) -- so that the split
    // happens ONCE per element here instead of once per element and consuming block there (x was re-split by 4 column-tile blocks, dy by
    // 10 row-tile blocks).  Rows >= M of the last row block are written as zeros (they enter the consumer's reduction).
    unsigned* cf;               // plane 0, or null
    int64_t cf_plane;           // bytes from one plane to the next
    // ... and its column sums over this block's rows (the bias gradient the weight-gradient kernel used to take from its raw B operand):
    // csp[bm * csp_stride + n]; the blocks of the LAST row tile also zero the slabs bm + 1 .. csp_slabs - 1 (a shorter batch than the one
    // the slab count was sized for)
    float* csp;
    int64_t csp_stride;
    int csp_slabs;
    int low_prio;               // 1: the block's waves do NOT raise their issue priority (a product that runs beside a kernel the step is waiting for)
};

__device__ __forceinline__ float dr_dropout_scale(uint64_t seed, uint64_t idx, float keep);   // = dropout_scale of common.h (defined by the includer)

// ---- the three-plane bf16 split of an f32 value (the split-precision kernel further down; dr_finish emits planes too)
typedef __bf16 dr_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 dr_bf16x2 __attribute__((ext_vector_type(2)));
typedef float dr_f32x2 __attribute__((ext_vector_type(2)));

template <class F, int... I>
__device__ __forceinline__ void dr_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void dr_static_for(F&& f) { dr_static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

struct DrPlanes { u32x4 h, m, l; };

__device__ __forceinline__ unsigned dr_pk_bf16(float a, float b) {          // v_cvt_pk_bf16_f32: a -> low half, b -> high half, RNE
    return __builtin_bit_cast(unsigned, __builtin_convertvector(dr_f32x2{a, b}, dr_bf16x2));
}
// (plain v_sub_f32: hipcc's SLP pass would pair them into v_pk_add_f32, which costs more issue time beside MFMAs than two subs)
__device__ __forceinline__ float dr_sub(float a, float b) {
#ifdef DR3_PK_SUB
    return a - b;
#else
    float r;
    asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#endif
}
__device__ __forceinline__ void dr_split3(const float (&x)[8], DrPlanes& p) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float x0 = x[2 * t], x1 = x[2 * t + 1];
        const unsigned h = dr_pk_bf16(x0, x1);
        const float r0 = dr_sub(x0, __uint_as_float(h << 16)), r1 = dr_sub(x1, __uint_as_float(h & 0xffff0000u));
        const unsigned m = dr_pk_bf16(r0, r1);
        const float s0 = dr_sub(r0, __uint_as_float(m << 16)), s1 = dr_sub(r1, __uint_as_float(m & 0xffff0000u));
        p.h[t] = h;
        p.m[t] = m;
        p.l[t] = dr_pk_bf16(s0, s1);
    }
}
__device__ __forceinline__ f32x4 dr_mfma_bf16(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dr_bf16x8, a), __builtin_bit_cast(dr_bf16x8, b), c, 0, 0, 0);
}


// Row / column of the tile that MFMA output row rho (= A-fragment lane) of A tile i / output column c (= B-fragment lane) of B
// tile j stands for.  An operand whose NON-reduction dimension is contiguous ("NC") is loaded with one dwordx4 (dwordx2) per
// lane along that dimension: lane c then holds one element of 4 (2) different 16-wide tiles, i.e. tile j = 4 jj + e covers the
// columns 64 jj + 4 c + e.  Which columns a tile covers is free -- only the epilogue needs to know.
template <int TN, bool B_RC>
__device__ __forceinline__ constexpr int dr_col(int j, int c) {
    constexpr int TQ = B_RC ? 0 : TN / 4;
    return j < 4 * TQ ? 64 * (j / 4) + 4 * c + (j % 4) : 64 * TQ + 16 * (j - 4 * TQ) + c;
}
template <int TM, bool A_RC>
__device__ __forceinline__ constexpr int dr_row(int i, int rho) {
    constexpr int VA = A_RC ? 1 : (TM % 4 == 0 ? 4 : (TM % 2 == 0 ? 2 : 1));
    return VA == 1 ? 16 * i + rho : 16 * VA * (i / VA) + VA * rho + (i % VA);
}
// reduction unit of tile (i,j): the four tiles of a quad stay together (their stage write is one b128)
template <int TN, bool B_RC>
__device__ __forceinline__ constexpr int dr_owner(int i, int j) {
    constexpr int TQ = B_RC ? 0 : TN / 4;
    constexpr int UPR = TQ + (TN - 4 * TQ);
    return (i * UPR + (j < 4 * TQ ? j / 4 : TQ + (j - 4 * TQ))) & 3;
}

// Cross-wave reduction of the four partial tiles + row-major staging, for wave W (compile-time, so that every register index is
// static and the LDS reads of all owned tiles are issued together).  The non-owners dump their registers as
// [tile][src][lane] float4 (conflict-free b128); the owner adds them to its own and writes the result into the
// [16 TM][16 TN + 4] row-major stage that aliases the slots (hence the barrier in between).
template <int TM, int TN, bool A_RC, bool B_RC, int W>
__device__ __forceinline__ void dr_reduce_tiles(f32x4 (&acc)[TM][TN], float* lds, int lane) {
    constexpr int LDS_ = 16 * TN + 4;
    constexpr int TQ = B_RC ? 0 : TN / 4;
    f32x4* slots = reinterpret_cast<f32x4*>(lds);
    const int c = lane & 15, q = lane >> 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int tid = i * TN + j, owner = dr_owner<TN, B_RC>(i, j);
            if (owner != W) slots[tid * 192 + ((W - owner - 1) & 3) * 64 + lane] = acc[i][j];
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int tid = i * TN + j, owner = dr_owner<TN, B_RC>(i, j);
            if (owner == W) {
                const f32x4* sp = slots + tid * 192 + lane;
                acc[i][j] = acc[i][j] + sp[0] + sp[64] + sp[128];
            }
        }
    __syncthreads();          // everybody has read its slots: the space becomes the output stage
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int jj = 0; jj < TQ; ++jj) {
            if (dr_owner<TN, B_RC>(i, 4 * jj) == W) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    *reinterpret_cast<f32x4*>(&lds[dr_row<TM, A_RC>(i, 4 * q + r) * LDS_ + 64 * jj + 4 * c]) =
                        f32x4{acc[i][4 * jj][r], acc[i][4 * jj + 1][r], acc[i][4 * jj + 2][r], acc[i][4 * jj + 3][r]};
            }
        }
#pragma unroll
        for (int j = 4 * TQ; j < TN; ++j) {
            if (dr_owner<TN, B_RC>(i, j) == W) {
#pragma unroll
                for (int r = 0; r < 4; ++r) lds[dr_row<TM, A_RC>(i, 4 * q + r) * LDS_ + dr_col<TN, B_RC>(j, c)] = acc[i][j][r];
            }
        }
    }
}

// Everything behind the main loop, shared by the exact kernel and the split-precision one (gemm_dr3_kernel): bias-gradient column
// sums, cross-wave reduction, row-major staging, epilogue (bias / ReLU / dropout, ReLU mask, column scale) and the coalesced stores.
template <int TM, int TN, bool A_PLAIN, bool B_RC, bool CS, int EPI, bool GATE>
__device__ __forceinline__ void dr_finish(f32x4 (&acc)[TM][TN], float (&cs)[TN], float (&cs2)[GATE ? TN : 1], const DrEpilogue& ep, float* __restrict__ C, int ldc,
                                          int M, int N, int m0, int n0, int bm, int split, uint32_t keep_lo, uint32_t keep_hi, float* dr_lds) {
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c = lane & 15, q = lane >> 4;
    // ---- wgrad: bias gradient = column sums of B (= dY) over this block's reduction range, first row of tiles only
    if (CS && ep.colsum != nullptr && bm == 0) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float v = cs[j];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (q == 0) dr_lds[w * 16 * TN + dr_col<TN, B_RC>(j, c)] = v;
        }
        __syncthreads();
        if (t < 16 * TN && n0 + t < N)
            ep.colsum[(size_t)split * ep.colsum_stride + n0 + t] = (dr_lds[t] + dr_lds[16 * TN + t] + dr_lds[32 * TN + t] + dr_lds[48 * TN + t]) *
                                                                     (GATE ? ep.colscale[n0 + t] : 1.f);
        __syncthreads();
    }
    if constexpr (GATE) {
        if (ep.colsum2 != nullptr && bm == 0) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float v = cs2[j];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                if (q == 0) dr_lds[w * 16 * TN + dr_col<TN, B_RC>(j, c)] = v;
            }
            __syncthreads();
            if (t < 16 * TN && n0 + t < N)
                ep.colsum2[(size_t)split * ep.colsum2_stride + n0 + t] = dr_lds[t] + dr_lds[16 * TN + t] + dr_lds[32 * TN + t] + dr_lds[48 * TN + t];
            __syncthreads();
        }
    }

    // this thread's output column (store phase below) and its bias, loaded here so that the latency hides behind the reduction
    constexpr int C4 = 4 * TN;                  // float4s per tile row
    constexpr int RPI = 256 / C4;               // tile rows stored per pass of the block
    const int tr = t / C4, tc = t - tr * C4;
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    if (EPI == DR_BIAS_ACT && ep.bias != nullptr && t < RPI * C4) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (n0 + 4 * tc + e < N) bias4[e] = ep.bias[n0 + 4 * tc + e];
    }
    float cscale4[4] = {1.f, 1.f, 1.f, 1.f};
    if (GATE && t < RPI * C4) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (n0 + 4 * tc + e < N) cscale4[e] = ep.colscale[n0 + 4 * tc + e];
    }
    // ---- cross-wave reduction + row-major staging (dr_reduce_tiles), one instantiation per wave id
    constexpr int LDS_ = 16 * TN + 4;           // staged row stride
    switch (w) {
        case 0: dr_reduce_tiles<TM, TN, A_PLAIN, B_RC, 0>(acc, dr_lds, lane); break;
        case 1: dr_reduce_tiles<TM, TN, A_PLAIN, B_RC, 1>(acc, dr_lds, lane); break;
        case 2: dr_reduce_tiles<TM, TN, A_PLAIN, B_RC, 2>(acc, dr_lds, lane); break;
        default: dr_reduce_tiles<TM, TN, A_PLAIN, B_RC, 3>(acc, dr_lds, lane); break;
    }
    DR_STAMP(3);
    __syncthreads();
    DR_STAMP(4);

    // ---- coalesced row-major stores: a thread keeps ONE float4 column of the tile and walks down the rows (its bias was loaded
    // before the reduction; the ReLU-mask reads of all its rows are issued together ahead of the stores).  Bias / ReLU /
    // dropout or the ReLU mask are applied here.
    constexpr int NIT = (16 * TM + RPI - 1) / RPI;
    const int gn = n0 + 4 * tc;
    float* Cz = C + (EPI == DR_STORE ? (size_t)split * ep.split_stride : 0);
    uint64_t seed = 0;
    if (EPI == DR_BIAS_ACT) seed = ep.seed ^ (ep.seed_ptr ? *ep.seed_ptr : 0ull);
    // fast path: every float4 of the tile is either whole or absent, and 16-byte aligned on both sides
    const bool fast = ((ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(Cz) & 15) == 0) && ((N & 3) == 0) &&
                      (EPI != DR_MASK || (((ep.ldact & 3) == 0) && ((reinterpret_cast<uintptr_t>(ep.act) & 15) == 0)));
    if (fast) {
        const bool col_on = t < RPI * C4 && gn < N;
        float4 am[NIT];
        if (EPI == DR_MASK) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = tr + RPI * it, gm = m0 + row;
                am[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (col_on && row < 16 * TM && gm < M) am[it] = *reinterpret_cast<const float4*>(ep.act + (size_t)gm * ep.ldact + gn);
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = tr + RPI * it, gm = m0 + row;
            if (col_on && row < 16 * TM && gm < M) {
                const float4 v4 = *reinterpret_cast<const float4*>(&dr_lds[row * LDS_ + 4 * tc]);
                float v[4] = {v4.x, v4.y, v4.z, v4.w};
                const float a[4] = {am[it].x, am[it].y, am[it].z, am[it].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (EPI == DR_BIAS_ACT) {
                        v[e] += bias4[e];
                        if (ep.relu) v[e] = fmaxf(v[e], 0.f);
                        if (ep.keep < 1.0f) v[e] *= (((it * 4 + e < 32 ? keep_lo : keep_hi) >> ((it * 4 + e) & 31)) & 1u) ? 1.0f / ep.keep : 0.0f;
                    } else if (EPI == DR_MASK) {
                        v[e] = (a[e] > 0.f) ? v[e] * ep.inv_keep : 0.f;
                    } else if (GATE) {
                        v[e] *= cscale4[e];
                    }
                }
                *reinterpret_cast<float4*>(Cz + (size_t)gm * ldc + gn) = make_float4(v[0], v[1], v[2], v[3]);
                if (ep.cf != nullptr || ep.csp != nullptr) *reinterpret_cast<float4*>(&dr_lds[row * LDS_ + 4 * tc]) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        if (ep.cf != nullptr || ep.csp != nullptr) {
            __syncthreads();          // the stage holds the stored values
            if (ep.cf != nullptr) {
                // entry (rb, col): rows 8 rb .. 8 rb + 7 of tile column col -> 16 bytes per plane; consecutive threads take consecutive columns
                // (LDS: conflict-free; global: 16 TN x 16 contiguous bytes per row block and plane)
                char* cfb = reinterpret_cast<char*>(ep.cf);
                for (int idx = t; idx < 2 * TM * 16 * TN; idx += 256) {
                    const int rb = idx / (16 * TN), col = idx - rb * (16 * TN);
                    const int gm0 = m0 + 8 * rb, gc = n0 + col;
                    if (gc < N && gm0 < M) {
                        float x[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) x[e] = gm0 + e < M ? dr_lds[(8 * rb + e) * LDS_ + col] : 0.f;
                        DrPlanes pl;
                        dr_split3(x, pl);
                        char* o = cfb + ((size_t)(gm0 >> 3) * N + gc) * 16;
                        *reinterpret_cast<u32x4*>(o) = pl.h;
                        *reinterpret_cast<u32x4*>(o + ep.cf_plane) = pl.m;
                        *reinterpret_cast<u32x4*>(o + 2 * ep.cf_plane) = pl.l;
                    }
                }
            }
            if (ep.csp != nullptr && t < 16 * TN && n0 + t < N) {
                float sum = 0.f;
#pragma unroll 8
                for (int row = 0; row < 16 * TM; ++row) sum += m0 + row < M ? dr_lds[row * LDS_ + t] : 0.f;
                ep.csp[(size_t)bm * ep.csp_stride + n0 + t] = sum;
                if (m0 + 16 * TM >= M)
                    for (int sl = bm + 1; sl < ep.csp_slabs; ++sl) ep.csp[(size_t)sl * ep.csp_stride + n0 + t] = 0.f;
            }
        }
    } else {                     // odd strides / widths: element by element (not a tuned path)
        for (int idx = t; idx < 16 * TM * 16 * TN; idx += 256) {
            const int row = idx / (16 * TN), col = idx - row * (16 * TN);
            const int gm = m0 + row, gc = n0 + col;
            if (gm < M && gc < N) {
                float v = dr_lds[row * LDS_ + col];
                if (EPI == DR_BIAS_ACT) {
                    if (ep.bias != nullptr) v += ep.bias[gc];
                    if (ep.relu) v = fmaxf(v, 0.f);
                    if (ep.keep < 1.0f) v *= dr_dropout_scale(seed, ((ep.seed_ptr ? ep.seed_ptr[1] : 0ull) + (uint64_t)gm) * (uint64_t)N + gc, ep.keep);
                } else if (EPI == DR_MASK) {
                    v = (ep.act[(size_t)gm * ep.ldact + gc] > 0.f) ? v * ep.inv_keep : 0.f;
                } else if (GATE) {
                    v *= ep.colscale[gc];
                }
                Cz[(size_t)gm * ldc + gc] = v;
            }
        }
    }
}

// (the 2 x 8 weight-gradient variant is compiled for TWO blocks per CU: it is the tile of choice when the reduction runs over
//  millions of rows that stream from HBM -- AFM's attention weight over 3 M pair rows -- where one resident block per CU, one
//  16-row group of prefetch ahead of its MFMAs, spends more time waiting for memory than multiplying)
// Waves per SIMD the register allocation is held to (launch bounds).  DR_WAVES2 (a build-time experiment): every tile
// of up to 28 accumulator tiles is compiled for TWO blocks per CU (<= 256 registers per lane, <= 80 KB of LDS), so that two
// products that run side by side in the step -- a layer's dgrad and the weight gradient of the layer above -- share each CU's
// matrix pipes instead of taking turns at whole CUs: one block's prologue, cross-wave reduction and stores run under the other's MFMAs.
#ifdef DR_WAVES2
#define DR_MIN_WAVES(TM, TN, CS, AGEN) ((((TM) * (TN) <= 28 && (AGEN) == DR_AGEN_NONE) || ((CS) && (TM) * (TN) <= 16)) ? 2 : 1)
#else
#define DR_MIN_WAVES(TM, TN, CS, AGEN) (((CS) && (TM) * (TN) <= 16) ? 2 : 1)
#endif
template <int TM, int TN, bool A_RC, bool B_RC, bool CS, int EPI, int AGEN = DR_AGEN_NONE>
__global__ __launch_bounds__(256, DR_MIN_WAVES(TM, TN, CS, AGEN)) void gemm_dr_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                         float* __restrict__ C, int ldc, int M, int N, int K, int kchunk, int nbn,
                                                         DrEpilogue ep, DrOuter og) {
    static_assert(AGEN == DR_AGEN_NONE || (AGEN == DR_AGEN_OUTER_FWD && A_RC) || (AGEN == DR_AGEN_OUTER_WGRAD && !A_RC) ||
                  (AGEN == DR_BGATE_WGRAD && !A_RC && !B_RC && CS && EPI == DR_STORE), "generated A: fwd is RC, wgrad is NC; gated B: a weight gradient");
    constexpr bool OUTER = AGEN == DR_AGEN_OUTER_FWD || AGEN == DR_AGEN_OUTER_WGRAD;
    constexpr bool GATE = AGEN == DR_BGATE_WGRAD;
    constexpr bool A_PLAIN = A_RC || OUTER;                  // tile i = rows 16 i .. 16 i + 15 (no interleave)
    constexpr int TQ = B_RC ? 0 : TN / 4;                    // quads of B tiles sharing one dwordx4 per lane (NC only)
    constexpr int VA = A_PLAIN ? 1 : (TM % 4 == 0 ? 4 : (TM % 2 == 0 ? 2 : 1));
    constexpr int AG = TM / VA;                              // A load groups per step (NC only)
    extern __shared__ __attribute__((aligned(16))) float dr_lds[];
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c = lane & 15, q = lane >> 4;
    int bm, bn, split;
    {   // XCD-aware order: hardware puts linear block b (x fastest, then y) on XCD b % 8; every XCD gets a contiguous run of
        // (split, tile) pairs -- split slowest, n fastest -- so the blocks sharing an A row-panel, and the tiles of one reduction
        // split (which all stream the SAME rows of both operands: the weight gradients), sit behind the same L2.  Before round 3
        // only x was remapped: XCD k then held tile column k of EVERY split and each operand row crossed the fabric once per XCD
        // (PMC: 69 MB per launch for 20 MB of operands at c2's first layer; AFM's 3 M-row weight gradient re-read its 3.1 GB
        // operand 8 times).  Speed only: any placement computes the same result.
        const int nwg = gridDim.x * gridDim.y, b = blockIdx.y * gridDim.x + blockIdx.x;
        const int qq = nwg / 8, r = nwg % 8, xcd = b % 8, idx = b / 8;
        const int lb = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
        split = __builtin_amdgcn_readfirstlane(lb / (int)gridDim.x);
        const int tile = lb - split * (int)gridDim.x;
        bn = tile % nbn;
        bm = tile / nbn;
    }
    DR_STAMP(0);
    __builtin_amdgcn_s_setprio(3);          // the step runs background kernels beside the MLP: GEMM waves go first at the issue arbiter
    const int m0 = bm * 16 * TM, n0 = bn * 16 * TN;
    const int kb0 = split * kchunk, kb1 = min(K, kb0 + kchunk);
    // (wave-uniform by construction; the readfirstlane's make the compiler believe it -- a scalar offset or descriptor it cannot
    // PROVE uniform gets every buffer load wrapped in a waterfall loop: ~10 instructions per load, no overlap between loads)
    const int kw = ((max(kb1 - kb0, 0) + 15) / 16) * 4;                 // k per wave, a multiple of 4
    const int kbeg = __builtin_amdgcn_readfirstlane(min(kb0 + w * kw, kb1));
    const int kend = __builtin_amdgcn_readfirstlane(min(kb1, kbeg + kw));
    const int Gf = __builtin_amdgcn_readfirstlane((kend - kbeg) / 16);  // full groups of 16 k
    const int kt = kbeg + 16 * Gf;                                      // tail: < 16 k
    const int ns = __builtin_amdgcn_readfirstlane((kend - kt + 3) / 4); // tail steps (0..4), step-major k = kt + 4 s + q

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // wave-uniform bases + 32-bit lane offsets.  num_records is the true end of the operand behind the base, so rows / columns
    // beyond the matrix need no clamping and no predication: a load past the end returns 0 without touching memory, and
    // whatever a load finds beyond M or N INSIDE the buffer only feeds outputs nobody stores.
    // generated A: the buffer is the embedding matrix, whole rows from the block's first batch row (fwd: m0, wgrad: kbeg) on
    const float* Ab = AGEN == DR_AGEN_OUTER_FWD ? og.e + (size_t)m0 * og.e_ld : AGEN == DR_AGEN_OUTER_WGRAD ? og.e + (size_t)kbeg * og.e_ld
                      : A + (A_RC ? (size_t)m0 * lda + kbeg : (size_t)kbeg * lda + m0);
    const float* Bb = B + (B_RC ? (size_t)n0 * ldb + kbeg : (size_t)kbeg * ldb + n0);
    // num_records: the true end of the operand where this wave could reach it, i.e. clipped to the wave's own rows -- an
    // operand of several GB (AFM's 3 M pair rows) stays addressable with 32-bit offsets behind the per-wave base
    const int Kb = AGEN == DR_AGEN_OUTER_WGRAD ? min(kend, og.rows) : kend;      // last reduction index this wave reads, exclusive
    // (32-bit compares only: HIP's min / max on int64 resolve to the double overloads -- f64 VALU code, and a descriptor word that
    // leaves the scalar unit costs every buffer load a waterfall loop)
    auto clip31 = [](int64_t floats) -> int {
        const int hi = (int)(floats >> 32);
        const unsigned top = (unsigned)((uint64_t)floats >> 29);        // != 0: negative, or 2^31 bytes and more
        return hi < 0 ? 0 : (top != 0u ? 0x7ffffff0 : (int)((unsigned)floats * 4u));
    };
    const int bytesA = AGEN == DR_AGEN_OUTER_FWD ? clip31((int64_t)min(og.rows - m0, 16 * TM) * og.e_ld)
                       : AGEN == DR_AGEN_OUTER_WGRAD ? clip31((int64_t)(min(kend, og.rows) - kbeg) * og.e_ld)
                       : clip31(A_RC ? (int64_t)(min(M - m0, 16 * TM) - 1) * lda + (K - kbeg) : (int64_t)(kend - kbeg - 1) * lda + (M - m0));
    const int bytesB = clip31(B_RC ? (int64_t)(min(N - n0, 16 * TN) - 1) * ldb + (K - kbeg) : (int64_t)(Kb - kbeg - 1) * ldb + (N - n0));
    auto uni_ptr = [](const float* p) {
        const uint64_t v = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
    };
    const auto ra = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(Ab), 0, __builtin_amdgcn_readfirstlane(bytesA), 0x00020000);
    const auto rb = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(Bb), 0, __builtin_amdgcn_readfirstlane(bytesB), 0x00020000);
    constexpr int NA = A_RC ? TM : AG, NB = B_RC ? TN : TQ + (TN - 4 * TQ);         // lane offsets per operand
    int aoff[NA], boff[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
        aoff[i] = AGEN == DR_AGEN_OUTER_FWD ? 4 * ((16 * i + c) * og.e_ld + 4 * q) : AGEN == DR_AGEN_OUTER_WGRAD ? 4 * (4 * q * og.e_ld + c)
                  : 4 * (A_RC ? (16 * i + c) * lda + 4 * q : 4 * q * lda + 16 * VA * i + VA * c);
    // generated A: lane offset of the e_i factor (fwd: this lane's row, one scalar for its 4 steps; wgrad: row 4 q + s, per step) and,
    // wgrad, the block's fixed (pair, a, c0) per tile as scalar offsets
    int aoffI[OUTER ? NA : 1];
    unsigned sJt[AGEN == DR_AGEN_OUTER_WGRAD ? TM : 1], sIt[AGEN == DR_AGEN_OUTER_WGRAD ? TM : 1];
    if constexpr (AGEN == DR_AGEN_OUTER_FWD) {
#pragma unroll
        for (int i = 0; i < NA; ++i) aoffI[i] = 4 * (16 * i + c) * og.e_ld;
    }
    if constexpr (AGEN == DR_AGEN_OUTER_WGRAD) {
        const int km = (1 << og.logk) - 1;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            aoffI[i] = 4 * (4 * q * og.e_ld);
            const int mm = m0 + 16 * i;
            const int pr = og.pairs[__builtin_amdgcn_readfirstlane(mm >> (2 * og.logk))];
            sJt[i] = __builtin_amdgcn_readfirstlane(4u * (unsigned)(((pr & 0xffff) << og.logk) + (mm & km)));
            sIt[i] = __builtin_amdgcn_readfirstlane(4u * (unsigned)(((pr >> 16) << og.logk) + ((mm >> og.logk) & km)));
        }
    }
    const unsigned strideE = 4u * (unsigned)og.e_ld;
#pragma unroll
    for (int j = 0; j < NB; ++j)
        boff[j] = 4 * (B_RC ? (16 * j + c) * ldb + 4 * q : 4 * q * ldb + (j < TQ ? 64 * j + 4 * c : 64 * TQ + 16 * (j - TQ) + c));
    const unsigned strideA = 4u * (unsigned)lda, strideB = 4u * (unsigned)ldb;    // bytes per k step of an NC operand

    struct Frag { float a[TM][4]; float b[TN][4]; float ai[OUTER ? TM : 1][AGEN == DR_AGEN_OUTER_WGRAD ? 4 : 1]; float rs[GATE ? 4 : 1]; };
    // gated B: the row scales of this wave's reduction rows behind their own descriptor (lane (c, q), step s <-> row 16 g + 4 q + s)
    const auto rr = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(GATE ? ep.rowscale + kbeg : nullptr), 0, __builtin_amdgcn_readfirstlane(GATE ? max(kend - kbeg, 0) * 4 : 0), 0x00020000);
    auto ldv = [](auto rs, int voff, unsigned soff_, auto nt, float* d, int stride) {      // nt dwords -> d[0], d[stride], ...
        constexpr int NV = decltype(nt)::value;
        const unsigned soff = __builtin_amdgcn_readfirstlane(soff_);
        if constexpr (NV == 4) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e * stride] = __uint_as_float(v[e]);
        } else if constexpr (NV == 2) {
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
            d[0] = __uint_as_float(v[0]);
            d[stride] = __uint_as_float(v[1]);
        } else {
            d[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
        }
    };
    using I1 = std::integral_constant<int, 1>;
    using I4 = std::integral_constant<int, 4>;
    using IVA = std::integral_constant<int, VA>;
    // the A loads of one group (RC: one dwordx4 per tile = its 4 steps; NC: per step, one VA-wide load per VA tiles)
    auto loadA = [&](Frag& f, int g, unsigned sA, unsigned dA) { // g: group; sA: scalar offset of its step 0, dA: per step (NC)
        if constexpr (AGEN == DR_AGEN_OUTER_FWD) {
            // the 16 k of group g share (pair, a); lane-quarter q takes c = c0 + 4 q .. + 3 of e_j (one dwordx4) and the scalar e_i[a]
            const int kk = kbeg + 16 * g, km = (1 << og.logk) - 1;
            const int pr = og.pairs[__builtin_amdgcn_readfirstlane(kk >> (2 * og.logk))];
            const unsigned sJ = 4u * (unsigned)(((pr & 0xffff) << og.logk) + (kk & km));
            const unsigned sI = 4u * (unsigned)(((pr >> 16) << og.logk) + ((kk >> og.logk) & km));
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                ldv(ra, aoff[i], sJ, I4{}, &f.a[i][0], 1);
                ldv(ra, aoffI[i], sI, I1{}, &f.ai[i][0], 1);
            }
            return;
        }
        if constexpr (AGEN == DR_AGEN_OUTER_WGRAD) {
            // step s, lane (c, q): batch row kbeg + 16 g + 4 q + s; e_j[c0 + c] (16 lanes = 64 contiguous bytes) and e_i[a] (broadcast)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    ldv(ra, aoff[i], sJt[i] + (16u * g + s) * strideE, I1{}, &f.a[i][s], 4);
                    ldv(ra, aoffI[i], sIt[i] + (16u * g + s) * strideE, I1{}, &f.ai[i][s], 4);
                }
            return;
        }
        if constexpr (GATE) ldv(rr, 16 * q, 64u * g, I4{}, &f.rs[0], 1);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (A_RC) ldv(ra, aoff[i], sA, I4{}, &f.a[i][0], 1);
            else {
#pragma unroll
                for (int s = 0; s < 4; ++s) ldv(ra, aoff[i], sA + s * dA, IVA{}, &f.a[VA * i][s], 4);
            }
        }
    };
    // the B loads of step s (NC) / all of them with s == 0 (RC)
    auto loadB_unit = [&](Frag& f, int u, int s, unsigned sB) {
        if (B_RC) ldv(rb, boff[u], sB, I4{}, &f.b[u][0], 1);
        else if (u < TQ) ldv(rb, boff[u], sB, I4{}, &f.b[4 * u][s], 4);
        else ldv(rb, boff[u], sB, I1{}, &f.b[4 * TQ + (u - TQ)][s], 4);
    };
    float cs2[GATE ? TN : 1];                       // gated B: sum_row rowscale[row] h[row][n]
#pragma unroll
    for (int j = 0; j < (GATE ? TN : 1); ++j) cs2[j] = 0.f;
    auto fin = [&](Frag& f) {                       // generated A: the products, once per fragment set, right before its MFMAs
        if constexpr (OUTER) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int s = 0; s < 4; ++s) f.a[i][s] *= f.ai[i][AGEN == DR_AGEN_OUTER_WGRAD ? s : 0];
        }
    };
    float cs[TN];                                   // wgrad: column sums of B (= dY), from the fragments as they pass by
#pragma unroll
    for (int j = 0; j < TN; ++j) cs[j] = 0.f;
    auto mma_tile = [&](const Frag& f, int j, int s) {
        // gated B: the element becomes the gradient it stands for right before its MFMAs (three VALU ops that issue under the
        // matrix pipe's passes; done for the whole fragment set ahead of the MFMAs they cost the weight gradient 19 %)
        float bv = f.b[j][s];
        if constexpr (GATE) {
            cs2[j] += bv * f.rs[s];
            bv = bv > 0.f ? f.rs[s] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[i][s], bv, acc[i][j], 0, 0, 0);
        if (CS) cs[j] += bv;
    };
    auto gA = [&](int g) -> unsigned { return A_RC ? 64u * g : 16u * g * strideA; };
    auto gB = [&](int g) -> unsigned { return B_RC ? 64u * g : 16u * g * strideB; };
    // one pipeline half: prefetch group gn into nxt while the MFMAs of cur run.  The sched_group_barrier pipeline asks for the
    // issue order  A loads, then {1 B load, the MFMAs of the tiles it feeds} repeated, so the VMEM issue slots hide inside the
    // matrix pipe's 32-cycle passes instead of forming a burst during which the pipe drains.
    constexpr int A_LOADS = AGEN == DR_AGEN_OUTER_FWD ? 2 * TM : AGEN == DR_AGEN_OUTER_WGRAD ? 8 * TM : (A_RC ? TM : 4 * AG) + (GATE ? 1 : 0);
    auto half = [&](Frag& nxt, int gn, Frag& cur) {
        __builtin_amdgcn_sched_barrier(0);
        fin(cur);
        loadA(nxt, gn, gA(gn), strideA);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                // NC: unit u's load of step s.  RC: one dwordx4 carries a tile's 4 steps -> one load every 4th slot, spread over the half
                if (!B_RC) loadB_unit(nxt, u, s, gB(gn) + s * strideB);
                else if (((s * NB + u) & 3) == 0) loadB_unit(nxt, (s * NB + u) >> 2, 0, gB(gn));
                if (!B_RC && u < TQ) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) mma_tile(cur, 4 * u + e, s);
                } else {
                    mma_tile(cur, B_RC ? u : 4 * TQ + (u - TQ), s);
                }
            }
        }
        __builtin_amdgcn_sched_group_barrier(0x20, A_LOADS, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int u = 0; u < TQ; ++u) {                    // (NC quads)
                __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, 4 * TM, 0);
            }
#pragma unroll
            for (int u = TQ; u < NB; ++u) {
                if (!B_RC || ((s * NB + u) & 3) == 0) __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, TM, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto load_all = [&](Frag& f, int g) {        // same issue order as half(): the compiler's vmcnt counts then agree on both loop entries
        loadA(f, g, gA(g), strideA);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int u = 0; u < NB; ++u)
                if (!B_RC || s == 0) loadB_unit(f, u, s, gB(g) + (B_RC ? 0u : s * strideB));
    };
    auto mma_all = [&](Frag& f, int nsteps) {
        fin(f);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < nsteps) {
#pragma unroll
                for (int j = 0; j < TN; ++j) mma_tile(f, j, s);
            }
        }
    };

    Frag f0, f1;
    // first groups: an odd count computes group 0 alone so that the pair loop below has no remainder
    int g = Gf & 1;
    if (Gf > 0) {
        if (Gf & 1) load_all(f1, 0);
        load_all(f0, min(g, Gf - 1));
    }
    // ---- dropout keep bits of the elements this thread will store (forward epilogue), computed HERE, while the first operand
    // loads are in flight: the mask is a 64-bit hash per element (common.h dropout_scale: three 64-bit multiplies, quarter-rate
    // integer ops) -- 26 elements per thread in a 2 x 13 tile -- and in the epilogue it was most of the store phase (cycle stamps:
    // ~2.6 k of the 3.2 k cycles between the reduction and the end of the kernel); here it costs nothing, the wave is waiting
    // for memory anyway.  Same function, same element index, same bits.
    constexpr int C4_ = 4 * TN, RPI_ = 256 / C4_, NIT_ = (16 * TM + RPI_ - 1) / RPI_;
    static_assert(NIT_ * 4 <= 64, "keep bits of a thread's output elements fit two words");
    uint32_t keep_lo = 0xffffffffu, keep_hi = 0xffffffffu;
    if constexpr (EPI == DR_BIAS_ACT) {
        if (ep.keep < 1.0f) {
            const uint64_t sd = ep.seed ^ (ep.seed_ptr ? *ep.seed_ptr : 0ull);
            const uint64_t row0 = ep.seed_ptr ? ep.seed_ptr[1] : 0ull;          // (StepState::row0: this rank's first example in the global batch)
            const int tr_ = t / C4_, tc_ = t - tr_ * C4_;
            keep_lo = keep_hi = 0u;
#pragma unroll
            for (int it = 0; it < NIT_; ++it) {
                const uint64_t base = (row0 + (uint64_t)(m0 + tr_ + RPI_ * it)) * (uint64_t)N + (uint64_t)(n0 + 4 * tc_);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t bit = dr_dropout_scale(sd, base + e, ep.keep) != 0.f ? 1u : 0u;
                    if (it * 4 + e < 32) keep_lo |= bit << ((it * 4 + e) & 31);
                    else keep_hi |= bit << ((it * 4 + e) & 31);
                }
            }
        }
    }
    DR_STAMP(1);
    __builtin_amdgcn_sched_barrier(0);
    if (Gf > 0) {
        if (Gf & 1) mma_all(f1, 4);
        for (; g < Gf; g += 2) {          // both halves unconditional: a load whose only use sits in a branch gets sunk into it
            half(f1, g + 1, f0);          // (g + 1 < Gf always: an even number of groups is left)
            half(f0, min(g + 2, Gf - 1), f1);      // the last prefetch re-reads a loaded group; nobody consumes it
        }
    }
    if (!OUTER && ns > 0) {       // (generated A: the host rounds the reduction length so that no wave has a tail)
        // ---- tail (< 16 k), after the main loop and into f0's registers: a third fragment set alive across the loop would push the
        // kernel past 304 VGPRs, the most that still shares a SIMD with two waves of the background table pass (104 each) -- and a
        // GEMM block that cannot be placed beside them waits for them to END.  Its load latency (~0.7 us) is exposed instead.
        // Step-major k assignment (k = kt + 4 s + q); a lane whose k lies beyond the wave's range reads at an offset past
        // num_records: the hardware returns 0 (no select after the load, no branch).
    #pragma unroll
        for (int s = 0; s < 4; ++s) {           // all four steps, unconditionally: a step beyond the tail is all-OOB (zeros, no traffic)
            const int k = kt + 4 * s + q;
            const bool ok = k < kend;
            const int kr = k - kbeg;
            if constexpr (GATE) ldv(rr, ok ? 4 * kr : 0x7ffffff0, 0u, I1{}, &f0.rs[s], 1);
    #pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int off = 4 * (A_RC ? (16 * i + c) * lda + kr : kr * lda + 16 * VA * i + VA * c);
                if (A_RC) ldv(ra, ok ? off : 0x7ffffff0, 0u, I1{}, &f0.a[i][s], 4);
                else ldv(ra, ok ? off : 0x7ffffff0, 0u, IVA{}, &f0.a[VA * i][s], 4);
            }
    #pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int off = 4 * (B_RC ? (16 * u + c) * ldb + kr : kr * ldb + (u < TQ ? 64 * u + 4 * c : 64 * TQ + 16 * (u - TQ) + c));
                if (B_RC) ldv(rb, ok ? off : 0x7ffffff0, 0u, I1{}, &f0.b[u][s], 4);
                else if (u < TQ) ldv(rb, ok ? off : 0x7ffffff0, 0u, I4{}, &f0.b[4 * u][s], 4);
                else ldv(rb, ok ? off : 0x7ffffff0, 0u, I1{}, &f0.b[4 * TQ + (u - TQ)][s], 4);
            }
        }
        mma_all(f0, ns);
    }
    DR_STAMP(2);

    dr_finish<TM, TN, A_PLAIN, B_RC, CS, EPI, GATE>(acc, cs, cs2, ep, C, ldc, M, N, m0, n0, bm, split, keep_lo, keep_hi, dr_lds);
    DR_STAMP(5);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Split-precision variant (dctr_config.gemm_mode = 1): the same direct-to-register wave-split-K product with every f32 operand
// element split IN REGISTERS into three bf16 planes  x = h + m + l  (round-to-nearest-even each time: h = bf16(x), m = bf16(x - h),
// l = bf16(x - h - m); the two differences are exact in f32 and after two 8-bit roundings at most 8 significant bits are left, so
// the three planes carry all 24 bits of x) and the six leading plane products
//     h h,  h m,  m h,  h l,  l h,  m m
// accumulated in f32 by v_mfma_f32_16x16x32_bf16 (products of two 8-bit significands are exact; the accumulator is f32).  The
// three dropped products (m l, l m, l l) are below 2^-24 |x y| each: the result is an f32 dot product to within the rounding of
// its f32 accumulation, like the exact kernel's -- NOT a bf16 product.  Why: the f32-input MFMA runs at 1/16 of the bf16 rate
// (MI355X_MICROARCH.md), so six bf16 products cost 6/16 of the matrix-pipe time of one f32 product.  What it costs instead is
// VALU: 11 ops per PAIR of elements (3 v_cvt_pk_bf16_f32, 2 shifts, 2 ands, 4 subtractions), once per element and wave because
// the waves split the reduction and each element enters the CU once -- 44 (TM + TN) VALU ops against 6 TM TN MFMAs per 32 k, which
// wants the squarer tiles (4 x 7, 4 x 10) rather than 2 x 13.
//
// k assignment: a group is 32 consecutive k; lane (c, q) holds k = 8 q + e, e = 0..7, of row / column c -- the A and B fragments
// of one 16x16x32 MFMA (any consistent order of k is a dot product).  A reduction-contiguous operand gives a lane its 8 values as
// two dwordx4; the other kind one dword (or a dwordx4 over four tiles) per e.  The raw f32 registers of a tile are free again as
// soon as the tile is split, and are refilled THEN with the next group's values: one raw set, a whole group of latency slack.
// Replaces the same TF ops as gemm_dr_kernel (DeepFM.py:156-158,165-166,213).
// Tiles of up to 14 accumulators (2 x 7: 32 x 112) are compiled for TWO blocks per CU (<= 256 registers, 42 KB of LDS): one wave
// per SIMD issues about one instruction per 5 cycles (MI355X_MICROARCH.md: <= 5 fillers per 32-cycle MFMA), which makes the
// ~2.9 split VALU ops per 17-cycle MFMA the bound; with a second block's wave on the same SIMD the two streams fill each other's
// issue gaps.  (The exact kernel gains nothing from that -- DESIGN 5b, DR_WAVES2 -- because its VALU stream is nearly empty.)
#define DR3_MIN_WAVES(TM, TN) ((TM) * (TN) <= 14 ? 2 : 1)
// B_PRE: the B operand (the layer's WEIGHT in the forward and dgrad products) arrives already split -- three bf16 planes laid out
// [plane][k / 8][column][8] (dr_wsplit_kernel writes them once per optimizer step; `B` points at plane 0, `ldb` = columns of a plane
// row, `bplane` = bytes from one plane to the next).  Every CU of a row block needs every weight element, so splitting weights
// in the product costs 64 x the VALU work of splitting them where they are written; the activations (A) are still split here.
// A_PRE (round 6): the A operand arrives as planes too, [plane][k / 8][row][8] (`lda` = rows of a plane row, `aplane` = bytes between
// planes): the weight-gradient product dW = X^T dY reduces over the BATCH rows, and both of its operands leave their producers'
// epilogues blocked along those rows (DrEpilogue::cf) -- the main loop is then loads and MFMAs only, 3 (TM + TN) 16-byte loads per 6 TM TN
// MFMAs, and a 2 x 7 tile fits two blocks per CU beside the dgrad blocks of the layer below.  Hybrid (A f32, reduction rows, split in
// registers; B planes) is the first layer's weight gradient: its A is the gathered embeddings, which nobody's epilogue writes.
//
// The kernel body is a device function of a VIRTUAL block index (vb of gx tiles x gy reduction splits) so that two products can share
// one launch (gemm_dr3_pair_kernel: a layer's weight gradient and the dgrad of the layer below read the same dY and are independent).
template <int TM, int TN, bool A_RC, bool B_RC, bool CS, int EPI, bool B_PRE = false, bool A_PRE = false>
__device__ __forceinline__ void gemm_dr3_body(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc, int M, int N,
                                              int K, int kchunk, int nbn, const DrEpilogue& ep, int64_t bplane, int64_t aplane, int vb, int gx, int gy, float* dr_lds) {
    static_assert(!B_PRE || (B_RC && !CS), "pre-split B: tiles cover plain column blocks (B_RC = true for the epilogue's column map); no column sums from raw registers");
    static_assert(!A_PRE || (A_RC && B_PRE), "pre-split A: plain row blocks (A_RC = true for the epilogue's row map), with pre-split B");
    constexpr int TQ = B_RC ? 0 : TN / 4;                    // quads of B tiles sharing one dwordx4 per lane (NC only)
    constexpr int VA = A_RC ? 1 : (TM % 4 == 0 ? 4 : (TM % 2 == 0 ? 2 : 1));
    constexpr int AG = TM / VA;                              // A load groups per e (NC only)
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c = lane & 15, q = lane >> 4;
    int bm, bn, split;
    {   // XCD-aware order, as gemm_dr_kernel
        const int nwg = gx * gy, b = vb;
        const int qq = nwg / 8, r = nwg % 8, xcd = b % 8, idx = b / 8;
        const int lb = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
        split = __builtin_amdgcn_readfirstlane(lb / gx);
        const int tile = lb - split * gx;
        bn = tile % nbn;
        bm = tile / nbn;
    }
    DR_STAMP(0);
    if (ep.low_prio == 0) __builtin_amdgcn_s_setprio(3);
    const int m0 = bm * 16 * TM, n0 = bn * 16 * TN;
    const int kb0 = split * kchunk, kb1 = min(K, kb0 + kchunk);            // (kchunk: a multiple of 32)
    const int kw = ((max(kb1 - kb0, 0) + 31) / 32) * 8;                    // k per wave, a multiple of 8
    const int kbeg = __builtin_amdgcn_readfirstlane(min(kb0 + w * kw, kb1));
    const int kend = __builtin_amdgcn_readfirstlane(min(kb1, kbeg + kw));  // (a multiple of 4 for a reduction-contiguous operand: the host requires K % 4 == 0)
    const int G = __builtin_amdgcn_readfirstlane((kend - kbeg + 31) / 32); // groups of 32 k, the last one possibly partial

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // (pre-split: plane rows of 8 k, 16 bytes per (k-block, row / column); kbeg is a multiple of 8)
    const float* Ab = A_PRE ? A + ((size_t)(kbeg / 8) * lda + m0) * 4 : A + (A_RC ? (size_t)m0 * lda + kbeg : (size_t)kbeg * lda + m0);
    const float* Bb = B_PRE ? B + ((size_t)(kbeg / 8) * ldb + n0) * 4 : B + (B_RC ? (size_t)n0 * ldb + kbeg : (size_t)kbeg * ldb + n0);
    auto clip31 = [](int64_t floats) -> int {
        const int hi = (int)(floats >> 32);
        const unsigned top = (unsigned)((uint64_t)floats >> 29);
        return hi < 0 ? 0 : (top != 0u ? 0x7ffffff0 : (int)((unsigned)floats * 4u));
    };
    // num_records: an operand whose rows ARE the reduction ends at this wave's last row -- whatever a partial or surplus group
    // addresses beyond it reads as 0 without touching memory; a reduction-contiguous operand ends at the end of the matrix and
    // its k beyond kend (the next wave's) are masked by lane offsets below
    const int bytesA = A_PRE ? clip31(((int64_t)((kend - kbeg + 7) / 8 - 1) * lda + (lda - m0)) * 4)        // this wave's k-blocks of ONE plane (in floats: 4 per entry)
                       : clip31(A_RC ? (int64_t)(min(M - m0, 16 * TM) - 1) * lda + (K - kbeg) : (int64_t)(kend - kbeg - 1) * lda + (M - m0));
    const int bytesB = B_PRE ? clip31(((int64_t)((kend - kbeg + 7) / 8 - 1) * ldb + (ldb - n0)) * 4)
                       : clip31(B_RC ? (int64_t)(min(N - n0, 16 * TN) - 1) * ldb + (K - kbeg) : (int64_t)(kend - kbeg - 1) * ldb + (N - n0));
    auto uni_ptr = [](const float* p) {
        const uint64_t v = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
    };
    const auto ra = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(Ab), 0, __builtin_amdgcn_readfirstlane(kend > kbeg ? bytesA : 0), 0x00020000);
    const auto rb = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(Bb), 0, __builtin_amdgcn_readfirstlane(kend > kbeg ? bytesB : 0), 0x00020000);
    // (pre-split: one descriptor per plane, so that each plane ends at this wave's last k-block)
    const float* Bb1 = B_PRE ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(Bb) + bplane) : Bb;
    const float* Bb2 = B_PRE ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(Bb) + 2 * bplane) : Bb;
    const auto rb1 = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(Bb1), 0, __builtin_amdgcn_readfirstlane(kend > kbeg ? bytesB : 0), 0x00020000);
    const auto rb2 = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(Bb2), 0, __builtin_amdgcn_readfirstlane(kend > kbeg ? bytesB : 0), 0x00020000);
    const float* Ab1 = A_PRE ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(Ab) + aplane) : Ab;
    const float* Ab2 = A_PRE ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(Ab) + 2 * aplane) : Ab;
    const auto ra1 = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(Ab1), 0, __builtin_amdgcn_readfirstlane(kend > kbeg ? bytesA : 0), 0x00020000);
    const auto ra2 = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(Ab2), 0, __builtin_amdgcn_readfirstlane(kend > kbeg ? bytesA : 0), 0x00020000);
    constexpr int NA = A_PRE ? 1 : (A_RC ? TM : AG), NB = B_PRE ? 1 : (B_RC ? TN : TQ + (TN - 4 * TQ));         // lane offsets per operand
    int aoff[NA], boff[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
        aoff[i] = A_PRE ? 16 * (q * lda + c)                 // (tile i: + 256 i bytes, group g: + 64 g lda bytes -- scalar)
                  : 4 * (A_RC ? (16 * i + c) * lda + 8 * q : 8 * q * lda + 16 * VA * i + VA * c);
#pragma unroll
    for (int j = 0; j < NB; ++j)
        boff[j] = B_PRE ? 16 * (q * ldb + c)                 // (tile j: + 256 j bytes, group g: + 64 g ldb bytes -- scalar)
                  : 4 * (B_RC ? (16 * j + c) * ldb + 8 * q : 8 * q * ldb + (j < TQ ? 64 * j + 4 * c : 64 * TQ + 16 * (j - TQ) + c));
    const unsigned strideA = 4u * (unsigned)lda, strideB = 4u * (unsigned)ldb;    // bytes per k of an NC operand
    // reduction-contiguous pieces (k = 32 g + 8 q + 4 p .. + 3, p = 0, 1) are real while 32 g < lim[p]
    const int lim0 = (kend - kbeg) - 8 * q, lim1 = lim0 - 4;

    struct Raw { float a[A_PRE ? 1 : TM][8]; float b[B_PRE ? 1 : TN][8]; };
    auto ld4 = [](auto rs, int voff, unsigned soff_, float* d, int stride) {
        const unsigned soff = __builtin_amdgcn_readfirstlane(soff_);
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e * stride] = __uint_as_float(v[e]);
    };
    auto ld2 = [](auto rs, int voff, unsigned soff_, float* d, int stride) {
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        const unsigned soff = __builtin_amdgcn_readfirstlane(soff_);
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
        d[0] = __uint_as_float(v[0]);
        d[stride] = __uint_as_float(v[1]);
    };
    auto ld1 = [](auto rs, int voff, unsigned soff_, float* d) {
        const unsigned soff = __builtin_amdgcn_readfirstlane(soff_);
        d[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
    };
    // the loads of group g: A tile / load group i, B unit u
    auto loadA = [&](Raw& f, int g) {
        if constexpr (A_PRE) {
        } else if constexpr (A_RC) {
            const bool ok0 = 32 * g < lim0, ok1 = 32 * g < lim1;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ld4(ra, ok0 ? aoff[i] : 0x7ffffff0, 128u * g, &f.a[i][0], 1);
                ld4(ra, ok1 ? aoff[i] + 16 : 0x7ffffff0, 128u * g, &f.a[i][4], 1);
            }
        } else {
#pragma unroll
            for (int i = 0; i < AG; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned so = (32u * g + e) * strideA;
                    if constexpr (VA == 4) ld4(ra, aoff[i], so, &f.a[4 * i][e], 8);
                    else if constexpr (VA == 2) ld2(ra, aoff[i], so, &f.a[2 * i][e], 8);
                    else ld1(ra, aoff[i], so, &f.a[i][e]);
                }
        }
    };
    auto loadB = [&](Raw& f, int u, int g) {
        if constexpr (B_PRE) {
        } else if constexpr (B_RC) {
            const bool ok0 = 32 * g < lim0, ok1 = 32 * g < lim1;
            ld4(rb, ok0 ? boff[u] : 0x7ffffff0, 128u * g, &f.b[u][0], 1);
            ld4(rb, ok1 ? boff[u] + 16 : 0x7ffffff0, 128u * g, &f.b[u][4], 1);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned so = (32u * g + e) * strideB;
                if (u < TQ) ld4(rb, boff[u], so, &f.b[4 * u][e], 8);
                else ld1(rb, boff[u], so, &f.b[4 * TQ + (u - TQ)][e]);
            }
        }
    };
    float cs[TN];                                            // wgrad: column sums of B (= dY), first row of tiles only
#pragma unroll
    for (int j = 0; j < TN; ++j) cs[j] = 0.f;
    auto mma = [&](const DrPlanes (&pa)[TM], const DrPlanes& pb, int j) {
        // six products; consecutive MFMAs go to different accumulators (i varies fastest)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = dr_mfma_bf16(pa[i].m, pb.m, acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = dr_mfma_bf16(pa[i].l, pb.h, acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = dr_mfma_bf16(pa[i].h, pb.l, acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = dr_mfma_bf16(pa[i].m, pb.h, acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = dr_mfma_bf16(pa[i].h, pb.m, acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = dr_mfma_bf16(pa[i].h, pb.h, acc[i][j]);
    };
    // B tile -> the load unit it belongs to and whether it is the unit's last tile (the unit's raw registers are free after it)
    auto unit_of = [](int j) constexpr { return B_RC ? j : (j < 4 * TQ ? j / 4 : TQ + (j - 4 * TQ)); };
    auto last_of_unit = [](int j) constexpr { return B_RC || j >= 4 * TQ || (j & 3) == 3; };

    Raw f;
    DrPlanes pa[TM], pan[TM], pb0, pb1;
    DrPlanes pw[B_PRE ? TN : 1];                              // pre-split: the planes of every B tile, refilled tile by tile
    auto loadW = [&](int j, int g) {
        const unsigned so = __builtin_amdgcn_readfirstlane(64u * g * (unsigned)ldb + 256u * j);
        pw[j].h = __builtin_amdgcn_raw_buffer_load_b128(rb, boff[0], so, 0);
        pw[j].m = __builtin_amdgcn_raw_buffer_load_b128(rb1, boff[0], so, 0);
        pw[j].l = __builtin_amdgcn_raw_buffer_load_b128(rb2, boff[0], so, 0);
    };
    // pre-split A: the planes of A tile i of group g (a group beyond the wave's range is all out of range: zeros, no traffic)
    auto loadAP = [&](DrPlanes& d, int i, int g) {
        const unsigned so = __builtin_amdgcn_readfirstlane(64u * g * (unsigned)lda + 256u * i);
        d.h = __builtin_amdgcn_raw_buffer_load_b128(ra, aoff[0], so, 0);
        d.m = __builtin_amdgcn_raw_buffer_load_b128(ra1, aoff[0], so, 0);
        d.l = __builtin_amdgcn_raw_buffer_load_b128(ra2, aoff[0], so, 0);
    };
    if (G > 0) {
        if constexpr (A_PRE) {
#pragma unroll
            for (int i = 0; i < TM; ++i) loadAP(pa[i], i, 0);
        } else loadA(f, 0);
        if constexpr (B_PRE) {
#pragma unroll
            for (int j = 0; j < TN - 1; ++j) loadW(j, 0);     // (tile TN - 1 is loaded in the first region of the first group)
        } else {
#pragma unroll
            for (int u = 0; u < NB; ++u) loadB(f, u, 0);
        }
    }
    // ---- dropout keep bits of the elements this thread will store, while the first loads are in flight (as gemm_dr_kernel)
    constexpr int C4_ = 4 * TN, RPI_ = 256 / C4_, NIT_ = (16 * TM + RPI_ - 1) / RPI_;
    static_assert(NIT_ * 4 <= 64, "keep bits of a thread's output elements fit two words");
    uint32_t keep_lo = 0xffffffffu, keep_hi = 0xffffffffu;
    if constexpr (EPI == DR_BIAS_ACT) {
        if (ep.keep < 1.0f) {
            const uint64_t sd = ep.seed ^ (ep.seed_ptr ? *ep.seed_ptr : 0ull);
            const uint64_t row0 = ep.seed_ptr ? ep.seed_ptr[1] : 0ull;
            const int tr_ = t / C4_, tc_ = t - tr_ * C4_;
            keep_lo = keep_hi = 0u;
#pragma unroll
            for (int it = 0; it < NIT_; ++it) {
                const uint64_t base = (row0 + (uint64_t)(m0 + tr_ + RPI_ * it)) * (uint64_t)N + (uint64_t)(n0 + 4 * tc_);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t bit = dr_dropout_scale(sd, base + e, ep.keep) != 0.f ? 1u : 0u;
                    if (it * 4 + e < 32) keep_lo |= bit << ((it * 4 + e) & 31);
                    else keep_hi |= bit << ((it * 4 + e) & 31);
                }
            }
        }
    }
    DR_STAMP(1);
    auto split_b = [&](int j, DrPlanes& p) {
        dr_split3(f.b[j], p);
        if constexpr (CS) {                                   // (unconditional: a branch would cut the scheduling region in two)
#pragma unroll
            for (int e = 0; e < 8; ++e) cs[j] += f.b[j][e];
        }
    };
    // A refill of the raw registers of A tile i / load group i for group g (its values of the group before have all been split)
    auto loadA_unit = [&](int i, int g) {
        if constexpr (A_PRE) {
        } else if constexpr (A_RC) {
            const bool ok0 = 32 * g < lim0, ok1 = 32 * g < lim1;
            ld4(ra, ok0 ? aoff[i] : 0x7ffffff0, 128u * g, &f.a[i][0], 1);
            ld4(ra, ok1 ? aoff[i] + 16 : 0x7ffffff0, 128u * g, &f.a[i][4], 1);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned so = (32u * g + e) * strideA;
                if constexpr (VA == 4) ld4(ra, aoff[i], so, &f.a[4 * i][e], 8);
                else if constexpr (VA == 2) ld2(ra, aoff[i], so, &f.a[2 * i][e], 8);
                else ld1(ra, aoff[i], so, &f.a[i][e]);
            }
        }
    };
    // ... of an NC operand: the two k rows (2 tt, 2 tt + 1) of every load group
    auto loadA_krows = [&](int tt, int g) {
        if constexpr (!A_PRE && !A_RC) {
#pragma unroll
            for (int u = 0; u < AG; ++u)
#pragma unroll
                for (int e = 2 * tt; e < 2 * tt + 2; ++e) {
                    const unsigned so = (32u * g + e) * strideA;
                    if constexpr (VA == 4) ld4(ra, aoff[u], so, &f.a[4 * u][e], 8);
                    else if constexpr (VA == 2) ld2(ra, aoff[u], so, &f.a[2 * u][e], 8);
                    else ld1(ra, aoff[u], so, &f.a[u][e]);
                }
        }
    };
    __builtin_amdgcn_sched_barrier(0);
    if (G > 0) {                                              // group 0's A planes and first B tile, exposed once
        if constexpr (!A_PRE) {
#pragma unroll
            for (int i = 0; i < TM; ++i) dr_split3(f.a[i], pa[i]);
        }
        if constexpr (!B_PRE) {
            split_b(0, pb0);
            if (last_of_unit(0)) loadB(f, unit_of(0), 1);     // (a group beyond the wave's range is all out of range: zeros, no traffic)
        }
        if constexpr (!A_PRE) loadA(f, 1);
    }
    // One group = TN tile regions fenced by sched_barrier(0) (nothing crosses a fence: the loads stay where they are written, next
    // to the split that frees their registers -- left alone, hipcc sinks every load without a user in the block to its end).  Region j:
    //   the 6 TM MFMAs of B tile j  ||  the split of B tile j + 1 (tile 0 of the next group in the last region), 1 / TN of the next
    //   group's A splits (pair by pair), and the refills of whatever raw registers those splits freed.
    // The A planes alternate between two sets (cur, nxt); pb0 / pb1 alternate by tile parity (TN odd: they swap roles per group,
    // which is why `par` is a parameter).  Pre-split A: the next group's A planes are REQUESTED in the first regions of the group (one tile
    // per region: a whole group of latency slack) straight into the other set.
    constexpr int NPA = 4 * TM;                               // A pairs of a group
    auto body = [&](DrPlanes (&cur)[TM], DrPlanes (&nxt)[TM], int g, auto parc) {
        constexpr int PAR = decltype(parc)::value;            // parity of (tile index -> pb set) at tile 0 of this group
        dr_static_for<TN>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            __builtin_amdgcn_sched_barrier(0);
            DrPlanes& pc = ((j + PAR) & 1) ? pb1 : pb0;
            DrPlanes& pn = ((j + PAR) & 1) ? pb0 : pb1;
            constexpr int jn = (j + 1) % TN;                              // the B tile split in this region (next group's when it wraps)
            constexpr bool LB = !B_PRE && last_of_unit(jn);               // ... completes a load unit: refill it
            if constexpr (B_PRE) {
                // the planes of the tile whose MFMAs the region before issued are free: refill them (tile TN - 1's for THIS group)
                loadW((j + TN - 1) % TN, j > 0 ? g + 1 : g);
            } else {
                split_b(jn, pn);
                if constexpr (LB) loadB(f, unit_of(jn), j + 1 < TN ? g + 1 : g + 2);
            }
            if constexpr (A_PRE) {
                if constexpr (j < TM) loadAP(nxt[j], j, g + 1);
            } else {
            // the next group's A pairs p0 .. p1 - 1.  Order: a reduction-contiguous operand tile by tile (a tile's two loads hold its 8 k: it is
            // refilled after its four pairs); an NC operand k-pair by k-pair ACROSS the tiles (one load holds ONE k of every tile of a load
            // group: once the pair (2 tt, 2 tt + 1) of every tile is split those two rows of registers are refilled for group g + 2.  Round 6:
            // refilled tile by tile, a one-group operand (TM = VA = 4: the weight gradient) got its registers back only behind the LAST pair, a
            // region ahead of their next use; measured, the k-pair order changes nothing -- 33.7 k cycles either way: the loop is issue-bound,
            // tools/mfma32_mix.hip -- it is kept because the slack is real)
            constexpr int p0 = j * NPA / TN, p1 = (j + 1) * NPA / TN;
#pragma unroll
            for (int p = p0; p < p1; ++p) {
                const int i = A_RC ? p / 4 : p % TM, tt = A_RC ? p % 4 : p / TM;
                const float x0 = f.a[i][2 * tt], x1 = f.a[i][2 * tt + 1];
                const unsigned h = dr_pk_bf16(x0, x1);
                const float r0 = dr_sub(x0, __uint_as_float(h << 16)), r1 = dr_sub(x1, __uint_as_float(h & 0xffff0000u));
                const unsigned m = dr_pk_bf16(r0, r1);
                const float s0 = dr_sub(r0, __uint_as_float(m << 16)), s1 = dr_sub(r1, __uint_as_float(m & 0xffff0000u));
                nxt[i].h[tt] = h;
                nxt[i].m[tt] = m;
                nxt[i].l[tt] = dr_pk_bf16(s0, s1);
                if constexpr (A_RC) { if (tt == 3) loadA_unit(i, g + 2); }
                else { if (i == TM - 1) loadA_krows(tt, g + 2); }
            }
            }
            if constexpr (B_PRE) mma(cur, pw[j], j);
            else mma(cur, pc, j);
            constexpr int NPAIR = A_PRE ? 0 : ((j + 1) * NPA / TN - j * NPA / TN);
            constexpr int NVAL = (B_PRE ? 0 : 44 + (CS ? 8 : 0)) + 11 * NPAIR + (LB && B_RC ? 2 : 0) + (A_RC && !A_PRE ? 4 : 0);
            constexpr int NM = 6 * TM;
            constexpr int PER = (NVAL + NM - 1) / NM;
            constexpr int NLD = A_PRE ? 3 + (j < TM ? 3 : 0) : 16;       // pre-split: spread the region's 16-byte loads over its first MFMAs
#pragma unroll
            for (int k = 0; k < NM; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                if constexpr (PER > 0) __builtin_amdgcn_sched_group_barrier(0x2, PER, 0);
                if (k < NLD) __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
            }
        });
        __builtin_amdgcn_sched_barrier(0);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, TN & 1>;
    {
        int g = 0;
        for (; g + 1 < G; g += 2) {            // pairs: the A planes alternate between pa and pan without a copy
            body(pa, pan, g, P0{});
            body(pan, pa, g + 1, P1{});
        }
        if (g < G) body(pa, pan, g, P0{});
    }
    DR_STAMP(2);
    float cs2[1] = {0.f};
    dr_finish<TM, TN, A_RC, B_RC, CS, EPI, false>(acc, cs, cs2, ep, C, ldc, M, N, m0, n0, bm, split, keep_lo, keep_hi, dr_lds);     // (B_PRE: B_RC = true -> plain column blocks)
    DR_STAMP(5);
}

template <int TM, int TN, bool A_RC, bool B_RC, bool CS, int EPI, bool B_PRE = false, bool A_PRE = false>
__global__ __launch_bounds__(256, DR3_MIN_WAVES(TM, TN)) void gemm_dr3_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                          float* __restrict__ C, int ldc, int M, int N, int K, int kchunk, int nbn, DrEpilogue ep,
                                                          int64_t bplane, int64_t aplane) {
    extern __shared__ __attribute__((aligned(16))) float dr_lds[];
    gemm_dr3_body<TM, TN, A_RC, B_RC, CS, EPI, B_PRE, A_PRE>(A, lda, B, ldb, C, ldc, M, N, K, kchunk, nbn, ep, bplane, aplane,
                                                             (int)(blockIdx.y * gridDim.x + blockIdx.x), (int)gridDim.x, (int)gridDim.y, dr_lds);
}

// Two products in ONE launch: blocks [0, n1) run product 1 (x tiles, then reduction splits), blocks [n1, n1 + n2) product 2.  A layer's
// weight gradient dW_l = X_l^T dY_l and the dgrad of the layer below dX_l = dY_l W_l^T (the input gradient through the SAME layer) read the
// same dY_l and nothing of each other: launched as one grid they need no cross-stream record on the critical stream (an event is a
// barrier packet: ~5 us of drained pipeline each, round-6 timeline), the dispatcher hands a CU that finishes a block of one the next block
// of either, and with both tiles at two blocks per CU a block's prologue / cross-wave reduction / stores run under another's MFMAs.
// Both bodies must be compiled for the same occupancy (the launch bounds are the pair's).
template <int TM_, int TN_, bool A_RC_, bool B_RC_, bool CS_, int EPI_, bool B_PRE_, bool A_PRE_>
struct Dr3Cfg {
    static constexpr int TM = TM_, TN = TN_, EPI = EPI_;
    static constexpr bool A_RC = A_RC_, B_RC = B_RC_, CS = CS_, B_PRE = B_PRE_, A_PRE = A_PRE_;
};
struct Dr3Arg {
    const float* A; int lda; const float* B; int ldb; float* C; int ldc; int M, N, K, kchunk, nbn; DrEpilogue ep; int64_t bplane, aplane;
    int gx, gy;                 // tiles, reduction splits
};
template <class P1, class P2>
__global__ __launch_bounds__(256, 2) void gemm_dr3_pair_kernel(Dr3Arg a, Dr3Arg b, int n1) {
    static_assert(P1::TM * P1::TN <= 14 && P2::TM * P2::TN <= 14, "both products at two blocks per CU");
    extern __shared__ __attribute__((aligned(16))) float dr_lds[];
    const int hb = (int)blockIdx.x;
    if (hb < n1)
        gemm_dr3_body<P1::TM, P1::TN, P1::A_RC, P1::B_RC, P1::CS, P1::EPI, P1::B_PRE, P1::A_PRE>(a.A, a.lda, a.B, a.ldb, a.C, a.ldc, a.M, a.N, a.K, a.kchunk, a.nbn, a.ep,
                                                                                                  a.bplane, a.aplane, hb, a.gx, a.gy, dr_lds);
    else
        gemm_dr3_body<P2::TM, P2::TN, P2::A_RC, P2::B_RC, P2::CS, P2::EPI, P2::B_PRE, P2::A_PRE>(b.A, b.lda, b.B, b.ldb, b.C, b.ldc, b.M, b.N, b.K, b.kchunk, b.nbn, b.ep,
                                                                                                  b.bplane, b.aplane, hb - n1, b.gx, b.gy, dr_lds);
}

// W [K][N] (row stride ldw) -> the two pre-split forms the forward and the dgrad product read (B_PRE above):
//   fwd:   [plane][K / 8][N][8]   (reduction over k: the 8 k of a block side by side)
//   dgrad: [plane][N / 8][K][8]   (reduction over n)
// planes h, m, l as in dr_split3; K8 = ceil(K / 8), N8 = ceil(N / 8): the last block of a ragged dimension is padded with zeros.
// One thread per 16-byte entry of each form; launched behind the layer's optimizer step.
struct DrWsplitJob { const float* W; int ldw, K, N; unsigned* fwd; unsigned* dgr; int64_t first; };     // first: this job's first entry in the launch
struct DrWsplitJobs { DrWsplitJob j[8]; int n; int64_t total; };
// (several weights per launch: the MLP's layers are re-split together behind the step's last optimizer launch)
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void dr_wsplit_kernel(DrWsplitJobs jobs) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= jobs.total) return;
    int ji = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k)
        if (k < jobs.n && idx >= jobs.j[k].first) ji = k;
    const DrWsplitJob& jb = jobs.j[ji];
    idx -= jb.first;
    const float* __restrict__ W = jb.W;
    const int ldw = jb.ldw, K = jb.K, N = jb.N;
    const int K8 = (K + 7) / 8, N8 = (N + 7) / 8;
    const int64_t nf = (int64_t)K8 * N, nd = (int64_t)N8 * K;
    float x[8];
    u32x4* out;
    int64_t plane;
    if (idx < nf) {                                           // entry (kb, n): W[8 kb + e][n]
        const int kb = (int)(idx / N), n = (int)(idx - (int64_t)kb * N);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = 8 * kb + e < K ? W[(size_t)(8 * kb + e) * ldw + n] : 0.f;
        out = reinterpret_cast<u32x4*>(jb.fwd) + idx;
        plane = nf;
    } else if (idx < nf + nd) {                               // entry (nb, k): W[k][8 nb + e]
        const int64_t i2 = idx - nf;
        const int nb = (int)(i2 / K), k = (int)(i2 - (int64_t)nb * K);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = 8 * nb + e < N ? W[(size_t)k * ldw + 8 * nb + e] : 0.f;
        out = reinterpret_cast<u32x4*>(jb.dgr) + i2;
        plane = nd;
    } else {
        return;
    }
    DrPlanes p;
    dr_split3(x, p);
    out[0] = p.h;
    out[plane] = p.m;
    out[2 * plane] = p.l;
}

// LDS bytes of one block: the register dump of the reduction (3 KB per 16x16 tile) or the row-major stage, whichever is larger
template <int TM, int TN>
constexpr size_t gemm_dr_lds_bytes() {
    const size_t slots = (size_t)TM * TN * 3 * 64 * 16;
    const size_t stage = (size_t)16 * TM * (16 * TN + 4) * 4;
    const size_t cs = (size_t)4 * 16 * TN * 4;
    size_t m = slots > stage ? slots : stage;
    return m > cs ? m : cs;
}

}  // namespace dctr
