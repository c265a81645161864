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
};

__device__ __forceinline__ float dr_dropout_scale(uint64_t seed, uint64_t idx, float keep);   // = dropout_scale of common.h (defined by the includer)

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
    DR_STAMP(5);
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
