// EXPERIMENT (round 6), not part of the library: the weight gradient on v_mfma_f32_32x32x16_bf16.  Included by tools/gemm_pl_probe.hip only.
// Result (profiles/r06_wgrad32_probe.txt, r06_mfma32_mix.txt): correct (max error 2.2e-8 against fp64, column sums too), NOT faster -- 24.2 us for
// c2's first layer against 23.1 us for gemm_dr3_kernel<4, 7>: the split's VALU ops do not hide under the 8-pass MFMA either.  The issue model,
// measured: 32.6 cycles per MFMA back to back (also on ONE accumulator), 36.1 with 3.5 VALU ops between two MFMAs, 49.3 with 7, 62.1 with 10.5 --
// about three ops hide, every further one costs 3.6 cycles, the same slope as beside the 4-pass 16x16x32.  The split is 5.5 VALU ops per
// element whatever the MFMA shape; only splitting each element ONCE (DrEpilogue::cf, A_PRE / B_PRE) removes them, at the price of 6 more bytes
// stored per element by the producer (measured: +2.7..3.5 us per producing launch against -4.6..5 us per weight gradient: no net gain).
#pragma once
#include "../tf_repos_amd/csrc/gemm_dr.h"

namespace dctr {

// ---------------------------------------------------------------------------------------------------------------------------
// The weight gradient on the 8-pass MFMA (round 6): dW[M][N] = sum_k A[k][M] B[k][N], both operands activations whose NON-reduction
// dimension is contiguous (A = X [rows][K_in], B = dY [rows][N]), both split in registers -- 44 (TM + TN) VALU ops against 6 TM TN MFMAs per
// k-step whatever the MFMA shape.  gemm_dr3_kernel issues them beside v_mfma_f32_16x16x32_bf16, a 4-pass instruction under which NOTHING
// hides (tools/mfma_valu_mix.hip: ~3.4 cycles per VALU op on top of the MFMA's 17): its 4 x 7 weight-gradient main loop runs at 2 x its
// MFMA-issue bound (33.7 k cycles for 17.1 k), the matrix pipe 30 % busy.  v_mfma_f32_32x32x16_bf16 is 8 passes = 32 cycles per issue, under
// which ~5-7 VALU / VMEM issue slots fit (MI355X_MICROARCH.md): with a 64 x 128 tile (2 x 4 accumulators of 32 x 32) the split is 5.5 VALU
// ops per MFMA + the bias gradient's column sums -- about what hides.  Same six plane products, same order, f32 accumulation: the
// arithmetic of gemm_dr3_kernel (the sums associate differently: 16 k per MFMA instead of 32).
//
// Fragments: lane l = (c = l & 31, h = l >> 5) supplies A[k = 8 h + e][row c] and B[k = 8 h + e][col c], e = 0..7, of a 16-k step; the
// accumulator's register 4 gq + r holds row 8 gq + 4 h + r, column c.  A lane loads TWO neighbouring rows / columns per dwordx2 (both
// operands: "NC"), so tile 2 u + e' covers rows / columns 64 u + 2 c + e' of the block's patch.  Raw registers in two sets per operand
// (the loads of step g + 2 are requested a whole step ahead of their split).  The four waves of a block split the reduction as in
// gemm_dr_kernel and meet in LDS; wave W owns the register group gq = W of every accumulator.
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int TM, int TN>       // accumulators of 32 x 32 per block patch: (32 TM) x (32 TN); both even
constexpr size_t gemm_dr3w_lds_bytes() {
    const size_t slots = (size_t)TM * TN * 4 * 3 * 64 * 16;
    const size_t stage = (size_t)32 * TM * (32 * TN + 4) * 4;
    return slots > stage ? slots : stage;
}

template <int TM, int TN>
__global__ __launch_bounds__(256, 1) void gemm_dr3w_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
                                                           int M, int N, int K, int kchunk, int nbn, DrEpilogue ep) {
    static_assert(TM % 2 == 0 && TN % 2 == 0, "dwordx2 loads: tiles come in pairs");
    extern __shared__ __attribute__((aligned(16))) float dr_lds[];
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c = lane & 31, h = lane >> 5;
    int bm, bn, split;
    {   // XCD-aware order, as gemm_dr_kernel
        const int gx = (int)gridDim.x, nwg = gx * (int)gridDim.y, b = (int)(blockIdx.y * gridDim.x + blockIdx.x);
        const int qq = nwg / 8, r = nwg % 8, xcd = b % 8, idx = b / 8;
        const int lb = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
        split = __builtin_amdgcn_readfirstlane(lb / gx);
        const int tile = lb - split * gx;
        bn = tile % nbn;
        bm = tile / nbn;
    }
    DR_STAMP(0);
    if (ep.low_prio == 0) __builtin_amdgcn_s_setprio(3);
    const int m0 = bm * 32 * TM, n0 = bn * 32 * TN;
    const int kb0 = split * kchunk, kb1 = min(K, kb0 + kchunk);            // (kchunk: a multiple of 64)
    const int kw = ((max(kb1 - kb0, 0) + 63) / 64) * 16;                   // k per wave: whole 16-k steps
    const int kbeg = __builtin_amdgcn_readfirstlane(min(kb0 + w * kw, kb1));
    const int kend = __builtin_amdgcn_readfirstlane(min(kb1, kbeg + kw));
    const int G = __builtin_amdgcn_readfirstlane((kend - kbeg + 15) / 16);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto clip31 = [](int64_t floats) -> int {
        const int hi = (int)(floats >> 32);
        const unsigned top = (unsigned)((uint64_t)floats >> 29);
        return hi < 0 ? 0 : (top != 0u ? 0x7ffffff0 : (int)((unsigned)floats * 4u));
    };
    auto uni_ptr = [](const float* p) {
        const uint64_t v = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
    };
    // (an operand whose rows ARE the reduction ends at this wave's last row: whatever a partial or surplus step addresses beyond it reads as 0)
    const float* Ab = A + (size_t)kbeg * lda + m0;
    const float* Bb = B + (size_t)kbeg * ldb + n0;
    const int bytesA = clip31((int64_t)(kend - kbeg - 1) * lda + (M - m0)), bytesB = clip31((int64_t)(kend - kbeg - 1) * ldb + (N - n0));
    const auto ra = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(Ab), 0, __builtin_amdgcn_readfirstlane(kend > kbeg ? bytesA : 0), 0x00020000);
    const auto rb = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(Bb), 0, __builtin_amdgcn_readfirstlane(kend > kbeg ? bytesB : 0), 0x00020000);
    const int aoff = 4 * (8 * h * lda + 2 * c), boff = 4 * (8 * h * ldb + 2 * c);      // (+ 256 u bytes for the pair of tiles u)
    const unsigned strideA = 4u * (unsigned)lda, strideB = 4u * (unsigned)ldb;

    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    struct Raw { float a[TM][8]; float b[TN][8]; };
    auto loadA = [&](Raw& f, int g) {           // step g: TM / 2 pairs of tiles x 8 k
#pragma unroll
        for (int u = 0; u < TM / 2; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned so = __builtin_amdgcn_readfirstlane((16u * g + e) * strideA + 256u * u);
                const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(ra, aoff, so, 0);
                f.a[2 * u][e] = __uint_as_float(v[0]);
                f.a[2 * u + 1][e] = __uint_as_float(v[1]);
            }
    };
    auto loadB = [&](Raw& f, int g) {
#pragma unroll
        for (int u = 0; u < TN / 2; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned so = __builtin_amdgcn_readfirstlane((16u * g + e) * strideB + 256u * u);
                const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rb, boff, so, 0);
                f.b[2 * u][e] = __uint_as_float(v[0]);
                f.b[2 * u + 1][e] = __uint_as_float(v[1]);
            }
    };
    float cs[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) cs[j] = 0.f;
    auto mfma = [](const u32x4& a, const u32x4& b, const f32x16& cc) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dr_bf16x8, a), __builtin_bit_cast(dr_bf16x8, b), cc, 0, 0, 0);
    };
    auto mma = [&](const DrPlanes (&pa)[TM], const DrPlanes& pb, int j) {        // the six products, smallest first (as gemm_dr3_kernel)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = mfma(pa[i].m, pb.m, acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = mfma(pa[i].l, pb.h, acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = mfma(pa[i].h, pb.l, acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = mfma(pa[i].m, pb.h, acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = mfma(pa[i].h, pb.m, acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = mfma(pa[i].h, pb.h, acc[i][j]);
    };
    auto split_b = [&](const Raw& f, int j, DrPlanes& p) {
#ifdef DR3W_NOSPLIT          // (timing experiment: no split arithmetic -- wrong results)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) { p.h[tt] = __float_as_uint(f.b[j][2 * tt]); p.m[tt] = __float_as_uint(f.b[j][2 * tt + 1]); p.l[tt] = p.h[tt] ^ p.m[tt]; }
#else
        dr_split3(f.b[j], p);
#ifndef DR3W_NOCS
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[j] += f.b[j][e];           // the bias gradient (first row of tiles stores it)
#endif
#endif
    };
    Raw f0, f1;                                  // raw registers of even / odd steps
    DrPlanes pa[TM], pan[TM], pb0, pb1;
    if (G > 0) { loadA(f0, 0); loadB(f0, 0); loadA(f1, 1); loadB(f1, 1); }
    DR_STAMP(1);
    __builtin_amdgcn_sched_barrier(0);
    if (G > 0) {                                 // step 0's A planes and first B tile, exposed once
#pragma unroll
        for (int i = 0; i < TM; ++i) dr_split3(f0.a[i], pa[i]);
        split_b(f0, 0, pb0);
    }
    // One step = TN regions fenced by sched_barrier(0).  Region j: the 6 TM MFMAs of B tile j  ||  the split of B tile j + 1 (tile 0 of the
    // next step in the last region) and 1 / TN of the next step's A pairs.  `cur` holds this step's raw values, `nxt` the next step's: cur's A
    // registers were all split during the step before and are refilled (step g + 2) in region 0, its B registers after tile TN - 1 is split.
    constexpr int NPA = 4 * TM;
    auto body = [&](DrPlanes (&pc)[TM], DrPlanes (&pn)[TM], Raw& cur, Raw& nxt, int g, auto parc) {
        constexpr int PAR = decltype(parc)::value;
        dr_static_for<TN>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            __builtin_amdgcn_sched_barrier(0);
            DrPlanes& bc = ((j + PAR) & 1) ? pb1 : pb0;
            DrPlanes& bn_ = ((j + PAR) & 1) ? pb0 : pb1;
            if constexpr (j == 0) loadA(cur, g + 2);
            if constexpr (j + 1 < TN) split_b(cur, j + 1, bn_);
            else { split_b(nxt, 0, bn_); }
            if constexpr (j == TN - 1) loadB(cur, g + 2);               // (after this step's last B tile was split, in region TN - 2)
            constexpr int p0 = j * NPA / TN, p1 = (j + 1) * NPA / TN;
#pragma unroll
            for (int p = p0; p < p1; ++p) {
                const int i = p / 4, tt = p % 4;
                const float x0 = nxt.a[i][2 * tt], x1 = nxt.a[i][2 * tt + 1];
#ifdef DR3W_NOSPLIT
                pn[i].h[tt] = __float_as_uint(x0); pn[i].m[tt] = __float_as_uint(x1); pn[i].l[tt] = pn[i].h[tt] ^ pn[i].m[tt];
#else
                const unsigned hh = dr_pk_bf16(x0, x1);
                const float r0 = dr_sub(x0, __uint_as_float(hh << 16)), r1 = dr_sub(x1, __uint_as_float(hh & 0xffff0000u));
                const unsigned mm = dr_pk_bf16(r0, r1);
                const float s0 = dr_sub(r0, __uint_as_float(mm << 16)), s1 = dr_sub(r1, __uint_as_float(mm & 0xffff0000u));
                pn[i].h[tt] = hh;
                pn[i].m[tt] = mm;
                pn[i].l[tt] = dr_pk_bf16(s0, s1);
#endif
            }
            mma(pc, bc, j);
            constexpr int NVAL = 44 + 8 + 11 * (p1 - p0);
            constexpr int NM = 6 * TM;
            constexpr int PER = (NVAL + NM - 1) / NM;
            constexpr int NLD = (j == 0 ? 4 * TM : 0) + (j == TN - 1 ? 4 * TN : 0);
#pragma unroll
            for (int k = 0; k < NM; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, PER, 0);
                if (k * ((NLD + NM - 1) / NM) < NLD) __builtin_amdgcn_sched_group_barrier(0x20, (NLD + NM - 1) / NM, 0);
            }
        });
        __builtin_amdgcn_sched_barrier(0);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, TN & 1>;
    {
        int g = 0;
        for (; g + 1 < G; g += 2) {
            body(pa, pan, f0, f1, g, P0{});
            body(pan, pa, f1, f0, g + 1, P1{});
        }
        if (g < G) body(pa, pan, f0, f1, g, P0{});
    }
    DR_STAMP(2);

    // ---- bias gradient: column sums of B over this block's reduction range (first row of tiles)
    if (ep.colsum != nullptr && bm == 0) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float v = cs[j];
            v += __shfl_xor(v, 32);
            if (h == 0) dr_lds[w * 32 * TN + 64 * (j / 2) + 2 * c + (j & 1)] = v;
        }
        __syncthreads();
        if (t < 32 * TN && n0 + t < N)
            ep.colsum[(size_t)split * ep.colsum_stride + n0 + t] = dr_lds[t] + dr_lds[32 * TN + t] + dr_lds[64 * TN + t] + dr_lds[96 * TN + t];
        __syncthreads();
    }
    // ---- cross-wave reduction: wave W owns register group gq = W of every accumulator; the others' copies meet in LDS
    {
        f32x4* slots = reinterpret_cast<f32x4*>(dr_lds);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    if (gq != w) {          // (w is wave-uniform: a scalar branch around each store)
                        const int unit = (i * TN + j) * 4 + gq;
                        slots[unit * 192 + ((w - gq - 1) & 3) * 64 + lane] = f32x4{acc[i][j][4 * gq], acc[i][j][4 * gq + 1], acc[i][j][4 * gq + 2], acc[i][j][4 * gq + 3]};
                    }
        __syncthreads();
        f32x4 own[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x4 mine = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    if (gq == w) mine = f32x4{acc[i][j][4 * gq], acc[i][j][4 * gq + 1], acc[i][j][4 * gq + 2], acc[i][j][4 * gq + 3]};
                const f32x4* sp = slots + ((i * TN + j) * 4 + w) * 192 + lane;
                own[i][j] = mine + sp[0] + sp[64] + sp[128];
            }
        DR_STAMP(3);
        __syncthreads();          // the slots become the row-major stage [32 TM][32 TN + 4]
        constexpr int LDS_ = 32 * TN + 4;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 64 * (i / 2) + 2 * (8 * w + 4 * h + r) + (i & 1);
#pragma unroll
                for (int u = 0; u < TN / 2; ++u)
                    *reinterpret_cast<float2*>(&dr_lds[row * LDS_ + 64 * u + 2 * c]) = make_float2(own[i][2 * u][r], own[i][2 * u + 1][r]);
            }
        __syncthreads();
        DR_STAMP(4);
        // ---- coalesced row-major stores of the partial slab
        constexpr int C4 = 8 * TN, RPI = 256 / C4, NIT = (32 * TM + RPI - 1) / RPI;
        const int tr = t / C4, tc = t - tr * C4, gn = n0 + 4 * tc;
        float* Cz = C + (size_t)split * ep.split_stride;
        if (t < RPI * C4 && gn < N) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = tr + RPI * it, gm = m0 + row;
                if (row < 32 * TM && gm < M) *reinterpret_cast<float4*>(Cz + (size_t)gm * ldc + gn) = *reinterpret_cast<const float4*>(&dr_lds[row * LDS_ + 4 * tc]);
            }
        }
    }
    DR_STAMP(5);
}


}  // namespace dctr
