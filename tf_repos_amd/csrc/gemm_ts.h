// K6, fifth kernel shape: split-precision products for TALL operands -- the three 398-GFLOP products of AFM's attention layer at the
// reference's K = A = 256 (AFM.py:142-147 and their gradients over the B * P = 3.0 M pair rows; run.sh:18).  Round 6.
//
// gemm_ws.hip runs them on the f32 matrix pipe (157 TF peak, 116-124 TF sustained: 3.2-3.4 ms each).  Here every f32 value enters as
// three bf16 planes (x = h + m + l exactly, gemm_dr.h dr_split3), the six products that matter (h h, h m, m h, h l, l h, m m) run on
// v_mfma_f32_16x16x32_bf16 (2.5 PF peak) and accumulate in f32: the arithmetic of gemm_dr3, for a shape where it is nearly free --
//   * a wave owns 64 rows x ALL 16 N output columns, so the split of one A fragment (44 VALU ops) is shared by 6 N MFMAs (96 at
//     N = 256: 0.46 VALU ops per MFMA, where the MLP kernels pay 1.6-3.2);
//   * the small operand (the weight, <= 256 x 256) is split ONCE per step by ts_wsplit_kernel into planes laid out in B-fragment
//     order, and streamed group by group (32 reduction steps: 3 planes x 16 KB) through a triple-buffered LDS image by the LDS-DMA
//     path (global_load_lds, 1 KB per wave instruction, no registers, no ds_write) -- the four waves of a block share every
//     fragment they read;
//   * the ReLU gate of the input gradient (d ah = dsc (x) w_o . 1[ah > 0]) is ONE exact bf16 plane: three products, no split.
// A block = 256 rows of the tall operand per pass (4 waves of 64 rows, one per SIMD, or 8 waves of 32), persistent over the row tiles (grid = CUs).
// Operands are SWAPPED at the MFMA (srcA = weight planes, srcB = rows): lane (c, q) then holds 4 consecutive output columns
// (16 tt + 4 q ..+3) of row c -- the epilogue stores float4s, no LDS staging.
//
// k order inside a group of 32: fragment slot (q, e) holds k = 32 g + (e < 4 ? 4 q + e : 16 + 4 q + e - 4), so that each of a lane's
// two 16-byte loads of a row-major operand covers, across the four q, 64 contiguous bytes of the row.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "gemm_dr.h"

#ifndef TS_SKIP         // tools/gemm_ts_probe.hip only: leave out the A loads (1), the LDS staging (2), the stores (4) -- timing experiments, wrong results
#define TS_SKIP 0
#endif

namespace dctr {

__host__ __device__ constexpr int ts_k(int q, int e) { return e < 4 ? 4 * q + e : 16 + 4 * q + (e - 4); }

// planes[p][kb][n] (16 bytes = 8 bf16 each; kb = 4 g + q): the (R x N) small operand in B-fragment order.
//   trans = 0: element (k, n) = W[k * ldw + n]            (forward: W [R = in, N = out])
//   trans = 1: element (k, n) = W[n * ldw + k] * kscale[k] (input gradient: W [N = in, R = out], the reduction runs over the layer's outputs)
// One launch writes up to two plane sets of the same weight (blockIdx.y: job 0 / 1) -- the forward's and the input gradient's.
struct TsSplitJob { int trans; const float* kscale; int R, N; u32x4* planes; };
__global__ __launch_bounds__(256) void ts_wsplit_kernel(const float* __restrict__ W, int ldw, TsSplitJob j0, TsSplitJob j1) {
    const TsSplitJob& jb = blockIdx.y == 0 ? j0 : j1;
    const int R = jb.R, N = jb.N;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int NKB = R / 8;
    if (idx >= NKB * N) return;
    const int kb = idx / N, n = idx - kb * N;
    const int g = kb >> 2, q = kb & 3;
    float x[8];
    if (jb.trans) {             // the lane's 8 k are two runs of four along W's rows
        const f32x4 lo = *reinterpret_cast<const f32x4*>(W + (size_t)n * ldw + 32 * g + 4 * q), hi = *reinterpret_cast<const f32x4*>(W + (size_t)n * ldw + 32 * g + 16 + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[e] = lo[e]; x[4 + e] = hi[e]; }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = W[(size_t)(32 * g + ts_k(q, e)) * ldw + n];
    }
    if (jb.kscale != nullptr) {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] *= jb.kscale[32 * g + ts_k(q, e)];
    }
    DrPlanes p;
    dr_split3(x, p);
    jb.planes[idx] = p.h;
    jb.planes[(size_t)NKB * N + idx] = p.m;
    jb.planes[(size_t)2 * NKB * N + idx] = p.l;
}

struct TsArgs {
    const float* A;             // the tall operand [M, 32 KG], row-major (GEN: unused)
    int lda;
    const u32x4* planes;        // ts_wsplit_kernel's output for the small operand
    float* C;                   // [M, 16 NT] (forward: may be null -- the output itself is not stored)
    int ldc;
    int64_t M;
    const float* bias;          // forward: [N]
    const float* dot_w;         // forward: dot_out[row] = sum_n C[row, n] dot_w[n] ([N], never null; dot_out may be)
    float* dot_out;
    int64_t dot_stride;         // column halves (NTF = 2 NT): half h writes its part of the dot to dot_out[h * dot_stride + row] (the caller adds the two)
    const float* rowscale;      // gate: C[row, :] *= rowscale[row]
    // the tall operand's SIGN bits, one u64 per (row, lane quarter q): bit 4 tt + r = 1[C[row, 16 tt + 4 q + r] > 0] -- exactly the 8 reduction
    // slots per group the gate kernel's lane (row, q) feeds its MFMAs with (ts_k), so the input gradient of the layer reads 32 bytes per row
    // instead of the row (1 KB at N = 256): written by the forward (bits_out, may be null), read by TS_GATE with GEN = true (bits_in)
    unsigned long long* bits_out;
    const unsigned long long* bits_in;
    // GEN (forward): row r = (example b = r / P, pair p = r % P) of the tall operand is e[b, pair_i[p], :] . e[b, pair_j[p], :], formed in
    // the registers from the gathered embeddings e [examples, e_ld] (field f at f * 32 KG) -- AFM.py:130-139's element-wise products
    // without the [B P, K] tensor
    const float* e;
    int e_ld;
    int64_t e_floats;           // examples * e_ld
    const int16_t* pair_i;
    const int16_t* pair_j;
    int P;
};

enum { TS_FWD = 0, TS_GATE = 1 };

typedef __attribute__((address_space(3))) void ts_lds_ptr;

// KG: reduction length / 32; NT: output columns / 16.
//   TS_FWD : C = relu(A W + bias), dot_out = C . dot_w        (A split in registers: 6 products)
//   TS_GATE: C = rowscale (x) (1[A > 0] Wt')                  (A is a ReLU output: one exact plane, 3 products; GEN: its sign bits are read, not A)
// TM: 16-row tiles per wave; a block is 16 / TM waves = 256 rows.  TM = 4: four waves, one per SIMD, 256 accumulator registers each (the
// AGPR half of the file).  TM = 2: EIGHT waves, two per SIMD with 256 registers each -- one wave's row loads, split and epilogue stores run
// under the other's MFMAs (with one wave per SIMD every one of them idles the matrix pipe), at twice the LDS fragment traffic: fine for the
// six-product forward (LDS 50 % busy), not for the three-product gate (100 %).
// NTF: column tiles of the whole output when a block pass covers only NT of them (NTF = 2 NT: two column HALVES, each a tile of the block
// loop of its own).  Half the accumulators (128 registers at TM = 4) and half the LDS images (72 KB) let TWO blocks share a CU: they are not
// in step with each other, so one block's epilogue stores (3.1 GB at the reference point: 0.5-0.6 ms that no MFMA covers when all waves of
// a CU reach their epilogue together) run under the other's MFMAs.  It pays only where the row operand is cheap to form twice -- the gate
// from sign bits: 1.45 -> 1.20 ms; the forward (rows split twice) LOSES, 2.30 -> 2.77 ms stored rows / 2.37 -> 3.74 ms generated, and is not
// dispatched that way (tools/gemm_ts_probe measures both).
template <int KG, int NT, int MODE, bool GEN = false, int TM = 4, int NTF = NT>
__global__ __launch_bounds__(64 * (16 / TM), NTF == NT ? 1 : 2) void gemm_ts_kernel(TsArgs a) {
    static_assert(TM == 4 || TM == 2, "four or eight waves");
    static_assert(NTF == NT || NTF == 2 * NT, "whole rows or two column halves");
    constexpr int NW = 16 / TM;
    constexpr int H = NTF / NT, NF = 16 * NTF;
    constexpr int N = 16 * NT;
    constexpr int PLANE = 4 * N * 16;                  // bytes of one plane of one group
    constexpr int BUF = 3 * PLANE;
    constexpr int PIECES = BUF / 1024;                 // 1 KB LDS-DMA pieces per group
    extern __shared__ __attribute__((aligned(16))) char ts_lds[];
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int64_t nbt = (a.M + 255) / 256 * H;          // block tiles: (row tile, column half), the half fastest
    int64_t bt = blockIdx.x;
    if (bt >= nbt) return;

    auto uni_ptr = [](const void* p) {
        const uint64_t v = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
    };
    // the planes of group g -> LDS image `buf`, contiguous in both: wave w takes the 1 KB pieces w, w + NW, ...  (buffer form: the lane part
    // of the address is ONE register, lane x 16, and the piece is a scalar offset -- with global_load_lds hipcc hoisted a 64-bit per-lane
    // address per piece and group out of the loop: 192 registers, spills)
    const auto rp = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(a.planes), 0, 3 * KG * PLANE * H, 0x00020000);
    auto stage = [&](int g, int bufoff, int h) {         // bufoff: byte offset of the LDS image (a multiple of BUF); h: column half
        constexpr int PR = N * 16 / 1024;               // 1 KB pieces per (plane, q) row of this pass's columns
#pragma unroll
        for (int j = 0; j < PIECES / NW; ++j) {
            const int piece = NW * j + w;
            const int pq = piece / PR, o = piece - pq * PR;           // pq = 4 p + q
            const int p = pq >> 2, qq = pq & 3;
            const int so = __builtin_amdgcn_readfirstlane((((p * KG + g) * 4 + qq) * NF + h * N) * 16 + o * 1024);
#if defined(__HIP_DEVICE_COMPILE__)     // (hipcc's HOST pass drops the kernel's stub without a diagnostic when it meets this builtin)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (ts_lds_ptr*)(ts_lds + bufoff + piece * 1024), 16, lane * 16, so, 0, 0);
#else
            (void)so; (void)rp;
#endif
        }
    };
    // the rows of one block tile behind a per-wave base (rows beyond M: num_records ends at the last real row -> zeros, no traffic)
    auto rows_of = [&](int64_t tl) { const int64_t m0 = tl * 256 + 16 * TM * w; return (int)(a.M - m0 < 16 * TM ? (a.M - m0 > 0 ? a.M - m0 : 0) : 16 * TM); };
    auto a_rsrc = [&](int64_t tl) {
        if constexpr (GEN && MODE == TS_GATE) {          // the tile's sign words: 32 bytes per row
            const int rows = rows_of(tl);
            return __builtin_amdgcn_make_buffer_rsrc(uni_ptr(a.bits_in + (size_t)(tl * 256 + 16 * TM * w) * 4), 0, __builtin_amdgcn_readfirstlane(rows * 32), 0x00020000);
        } else if constexpr (GEN) {        // (one descriptor over all of e: the rows of a tile are anywhere in it)
            return __builtin_amdgcn_make_buffer_rsrc(uni_ptr(a.e), 0, __builtin_amdgcn_readfirstlane((int)(a.e_floats * 4)), 0x00020000);
        } else {
            const int rows = rows_of(tl);
            return __builtin_amdgcn_make_buffer_rsrc(uni_ptr(a.A + (size_t)(tl * 256 + 16 * TM * w) * a.lda), 0,
                                                     __builtin_amdgcn_readfirstlane(rows > 0 ? ((rows - 1) * a.lda + 32 * KG) * 4 : 0), 0x00020000);
        }
    };
    // lane offsets of the row tiles: into the tile's rows, or (GEN) into e for the two factors of each row's pair
    struct Offs { int i[TM]; int j[GEN ? TM : 1]; };
    auto offs_of = [&](int64_t tl, Offs& o) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if constexpr (GEN && MODE == TS_GATE) {
                o.i[i] = (16 * i + c) * 32 + 8 * q;
                o.j[i] = 0;
            } else if constexpr (GEN) {
                int64_t row = tl * 256 + 16 * TM * w + 16 * i + c;
                if (row >= a.M) row = a.M - 1;          // (a row past the end: any real pair -- nobody stores it)
                const unsigned b = (unsigned)row / (unsigned)a.P, p = (unsigned)row - b * (unsigned)a.P;
                o.i[i] = 4 * ((int)b * a.e_ld + a.pair_i[p] * 32 * KG + 4 * q);
                o.j[i] = 4 * ((int)b * a.e_ld + a.pair_j[p] * 32 * KG + 4 * q);
            } else {
                o.i[i] = 4 * ((16 * i + c) * a.lda + 4 * q);
            }
        }
    };
    int coff[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) coff[i] = 4 * ((16 * i + c) * a.ldc + 4 * q);
    struct Raw { float a[TM][8]; float b[GEN ? TM : 1][8]; };
    auto loadA = [&](Raw& raw, decltype(a_rsrc(0)) rs, const Offs& o, int g) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(rs, o.i[i], 128 * g, 0);
            const u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(rs, o.i[i] + 64, 128 * g, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) { raw.a[i][e] = __uint_as_float(v0[e]); raw.a[i][4 + e] = __uint_as_float(v1[e]); }
            if constexpr (GEN) {
                const u32x4 u0 = __builtin_amdgcn_raw_buffer_load_b128(rs, o.j[i], 128 * g, 0);
                const u32x4 u1 = __builtin_amdgcn_raw_buffer_load_b128(rs, o.j[i] + 64, 128 * g, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) { raw.b[i][e] = __uint_as_float(u0[e]); raw.b[i][4 + e] = __uint_as_float(u1[e]); }
            }
        }
    };
    // raw -> the MFMA operand(s) of the rows
    struct Ops { u32x4 h[TM], m[MODE == TS_FWD ? TM : 1], l[MODE == TS_FWD ? TM : 1]; };
    auto convert = [&](const Raw& raw, Ops& o) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if constexpr (MODE == TS_FWD) {
                DrPlanes p;
                if constexpr (GEN) {
                    float x[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = raw.a[i][e] * raw.b[i][e];       // (the product afm_pair_fwd_kernel stores: one f32 multiply)
                    dr_split3(x, p);
                } else {
                    dr_split3(raw.a[i], p);
                }
                o.h[i] = p.h; o.m[i] = p.m; o.l[i] = p.l;
            } else {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
                    o.h[i][tt] = (raw.a[i][2 * tt] > 0.f ? 0x3f80u : 0u) | (raw.a[i][2 * tt + 1] > 0.f ? 0x3f800000u : 0u);
            }
        }
    };
    // sign-bit form of the gate operand: a tile's words (one u64 per lane and row tile) and the expansion of group g's byte into the fragment
    struct Bits { unsigned lo[TM], hi[TM]; };
    auto loadBits = [&](Bits& b, decltype(a_rsrc(0)) rs, const Offs& o) {
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, o.i[i], 0, 0);
            b.lo[i] = v[0];
            b.hi[i] = v[1];
        }
    };
    auto expand = [&](const Bits& b, int g, Ops& o) {    // g: compile-time after unrolling
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const unsigned byte = g < 4 ? b.lo[i] >> (8 * g) : b.hi[i] >> (8 * (g - 4));
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const unsigned m0_ = (unsigned)((int)(byte << (31 - 2 * tt)) >> 31), m1_ = (unsigned)((int)(byte << (30 - 2 * tt)) >> 31);
                o.h[i][tt] = (m0_ & 0x3f80u) | (m1_ & 0x3f800000u);
            }
        }
    };
    const int boff = (q * N + c) * 16;                  // this lane's fragment of (plane 0, column tile 0)
    f32x4 acc[TM][NT];
    // One region per column tile, fenced: the three fragments of tile tt + 1 are asked for ahead of tile tt's MFMAs and nothing else moves
    // across (left alone, hipcc hoists all 3 NT fragment reads of a group to its top: 192 registers, spills).
    auto products = [&](const Ops& o, int bufoff) {
        const char* base = ts_lds + bufoff + boff;
        u32x4 fb[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) fb[0][p] = *reinterpret_cast<const u32x4*>(base + p * PLANE);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            if (tt + 1 < NT) {
#pragma unroll
                for (int p = 0; p < 3; ++p) fb[(tt + 1) & 1][p] = *reinterpret_cast<const u32x4*>(base + p * PLANE + 256 * (tt + 1));
            }
            const u32x4 bh = fb[tt & 1][0], bm = fb[tt & 1][1], bl = fb[tt & 1][2];
            if constexpr (MODE == TS_FWD) {
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][tt] = dr_mfma_bf16(bm, o.m[i], acc[i][tt]);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][tt] = dr_mfma_bf16(bh, o.l[i], acc[i][tt]);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][tt] = dr_mfma_bf16(bl, o.h[i], acc[i][tt]);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][tt] = dr_mfma_bf16(bh, o.m[i], acc[i][tt]);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][tt] = dr_mfma_bf16(bm, o.h[i], acc[i][tt]);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][tt] = dr_mfma_bf16(bh, o.h[i], acc[i][tt]);
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][tt] = dr_mfma_bf16(bl, o.h[i], acc[i][tt]);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][tt] = dr_mfma_bf16(bm, o.h[i], acc[i][tt]);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][tt] = dr_mfma_bf16(bh, o.h[i], acc[i][tt]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // The rows of group g + 1 are asked for at the top of group g and waited for at its end, one group (3-6 k cycles of MFMAs) later.
    // (Two groups ahead -- a second raw set -- changed nothing: 1.55 ms either way for the input gradient, tools/gemm_ts_probe; what the
    // kernel waits for is not the row loads.)
    // THREE LDS images: the planes of group s + 2 are asked for at the top of group s, retired by this wave's vmcnt(0) at its end and
    // read in group s + 2, TWO barriers later.  With two images (asked for one group ahead, read right behind the barrier that follows the
    // wait) the eight-wave gate kernel -- the one variant that keeps the LDS pipe saturated with fragment reads -- returned a few wrong
    // tiles per 10^5, differently from run to run: a piece the counter had retired was not yet what a ds_read of ANOTHER wave saw one
    // barrier later (delays in front of the barrier lowered the rate; tools/gemm_ts_probe compares kernels word by word).
    Raw raw;
    Ops ops[2];                                         // this group's operands / the next group's, alternating (KG is even: no copy)
    Offs off, noff;
    auto rs = a_rsrc(bt / H);
    offs_of(bt / H, off);
    noff = off;
    constexpr bool BITS = GEN && MODE == TS_GATE;
    Bits cb, nb;                                        // (BITS) this tile's sign words / the next tile's, asked for a whole tile ahead
    stage(0, 0, (int)(bt % H));
    stage(1, BUF, (int)(bt % H));
    if constexpr (BITS) { loadBits(cb, rs, off); nb = cb; } else loadA(raw, rs, off, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (BITS) expand(cb, 0, ops[0]); else convert(raw, ops[0]);
    int rd = 0, wr = 2 * BUF;                           // byte offsets of the image read in this group / written for the group after next
    while (true) {
        const int64_t next = bt + gridDim.x;
        const bool more = next < nbt;
        const int64_t rt = bt / H, rt_next = (more ? next : bt) / H;           // row tiles
        const int hh = (int)(bt % H), hh_next = (int)((more ? next : bt) % H);
        auto rs_next = a_rsrc(rt_next);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) acc[i][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int64_t m0 = rt * 256 + 16 * TM * w;
        // the per-row scalars of the epilogue, asked for before the products (in the epilogue they would queue behind the next tile's loads)
        float rsc[TM];
        if constexpr (MODE == TS_GATE) {
#pragma unroll
            for (int i = 0; i < TM; ++i) rsc[i] = m0 + 16 * i + c < a.M ? a.rowscale[m0 + 16 * i + c] : 0.f;
        }
        if constexpr (BITS) loadBits(nb, rs_next, off);  // (the last tile re-reads its own words: nobody consumes them)
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            // planes two groups ahead, rows one group ahead: both land under MFMAs and are waited for at this group's end
            if (!(TS_SKIP & 2)) {
                if (g + 2 < KG) stage(g + 2, wr, hh);
                else if (more) stage(g + 2 - KG, wr, hh_next);
            }
            if constexpr (!BITS) {
                if (g + 1 < KG) {
                    if (!(TS_SKIP & 1)) loadA(raw, rs, off, g + 1);
                    // (GEN: the next tile's pair lookups, two loads per row, a group ahead of the offsets' first use)
                    if (GEN && g + 2 == KG) offs_of(rt_next, noff);
                } else {
                    if (!(TS_SKIP & 1)) loadA(raw, rs_next, GEN ? noff : off, 0);     // (the last tile re-reads its own first group: nobody consumes it)
                }
            }
            products(ops[g & 1], rd);
            if constexpr (BITS) {
                if (g + 1 < KG) expand(cb, g + 1, ops[(g + 1) & 1]);
                else expand(nb, 0, ops[(g + 1) & 1]);
            } else {
                convert(raw, ops[(g + 1) & 1]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (everything this wave asked for has landed before it enters the barrier)
            __syncthreads();
            rd = rd == 2 * BUF ? 0 : rd + BUF;
            wr = wr == 2 * BUF ? 0 : wr + BUF;
        }
        // ---- epilogue on the accumulators: register r of lane (c, q) is row 16 i + c, column 16 tt + 4 q + r
        const int rows = rows_of(rt);
        // (C == null -- the forward when nobody reads its output, only the score and the sign words: an empty descriptor, every store is dropped)
        const auto rc = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(a.C != nullptr ? a.C + (size_t)m0 * a.ldc + hh * N : a.bias), 0,
                                                          __builtin_amdgcn_readfirstlane(rows > 0 && a.C != nullptr ? ((rows - 1) * a.ldc + N) * 4 : 0), 0x00020000);
        float dot[TM];
        unsigned sgn_lo[TM], sgn_hi[TM];                 // forward: the sign words of this lane's 4 NT outputs per row tile
#pragma unroll
        for (int i = 0; i < TM; ++i) { dot[i] = 0.f; sgn_lo[i] = sgn_hi[i] = 0u; }
        auto ld4 = [&](const float* p, int tt) { return *reinterpret_cast<const f32x4*>(p + hh * N + 16 * tt + 4 * q); };
        f32x4 bc = f32x4{0.f, 0.f, 0.f, 0.f}, dc = bc;
        if constexpr (MODE == TS_FWD) { bc = ld4(a.bias, 0); dc = ld4(a.dot_w, 0); }
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            f32x4 bn = bc, dn = dc;
            if constexpr (MODE == TS_FWD) {
                if (tt + 1 < NT) { bn = ld4(a.bias, tt + 1); dn = ld4(a.dot_w, tt + 1); }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                f32x4 v = acc[i][tt];
                if constexpr (MODE == TS_FWD) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = fmaxf(v[r] + bc[r], 0.f);
                        dot[i] += v[r] * dc[r];
                    }
                    const unsigned sm = (v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u);
                    if (tt < 8) sgn_lo[i] |= sm << (4 * (tt & 7));
                    else sgn_hi[i] |= sm << (4 * (tt & 7));
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= rsc[i];
                }
                // The column tile goes into the LANE offset, computed right here (the empty asm keeps hipcc from precomputing all 4 NT offsets
                // outside the loop: 64 registers).  NOT into the scalar offset: a 16-byte buffer store whose soffset SGPR is rewritten for the
                // next store right behind it put part of its data (the first register of lanes 12-15 of every 16) at the NEXT store's offset
                // whenever two waves shared the SIMD -- the eight-wave and two-blocks-per-CU gate kernels returned a few wrong 64-byte row
                // segments per 10^5, differently run to run, until the scalar offset went (8-byte stores with it are fine; tools/gemm_ts_probe).
#ifdef TS_STORE_SOFFSET         // tools/gemm_ts_probe.hip -DTS_STORE_SOFFSET: the scalar-offset form, to show the effect (profiles/r06_store_soffset_ab.txt)
                if (!(TS_SKIP & 4)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rc, coff[i], 64 * tt, 0);
#else
                if (!(TS_SKIP & 4)) {
                    int vo = coff[i] + 64 * tt;
                    asm volatile("" : "+v"(vo));
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rc, vo, 0, 0);
                }
#endif
            }
            bc = bn; dc = dn;
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (MODE == TS_FWD) {
            if (a.bits_out != nullptr) {                // (uniform)
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                const auto rb = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(a.bits_out + (size_t)m0 * 4), 0, __builtin_amdgcn_readfirstlane(rows * 32), 0x00020000);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if constexpr (H == 1) __builtin_amdgcn_raw_buffer_store_b64(u32x2{sgn_lo[i], sgn_hi[i]}, rb, (16 * i + c) * 32 + 8 * q, 0, 0);
                    else if constexpr (NT == 8) __builtin_amdgcn_raw_buffer_store_b32(sgn_lo[i], rb, (16 * i + c) * 32 + 8 * q + 4 * hh, 0, 0);       // (half h = bits 32 h ..)
                    else __builtin_amdgcn_raw_buffer_store_b16((unsigned short)sgn_lo[i], rb, (16 * i + c) * 32 + 8 * q + 2 * hh, 0, 0);          // (NT = 4: bits 16 h ..)
                }
            }
            if (a.dot_out != nullptr) {                 // (uniform) the four q-lanes of a row hold its four column quarters
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float d = dot[i];
                    d += __shfl_xor(d, 16);
                    d += __shfl_xor(d, 32);
                    if (q == 0 && m0 + 16 * i + c < a.M) a.dot_out[hh * a.dot_stride + m0 + 16 * i + c] = d;
                }
            }
        }
        if (!more) break;
        bt = next;
        rs = rs_next;
        if constexpr (BITS) cb = nb;
        else if constexpr (GEN) off = noff;
    }
}

// ---- the weight gradient of the same layer under its rank-one output gradient (gemm_dr.h DR_BGATE_WGRAD, in split precision) ----------------
//   dW[k, a] = colscale[a] * sum_r (rowscale[r] X[r, k]) * 1[H[r, a] > 0]       (X = the pair products, H = the layer's ReLU output)
//   db[a]    = colscale[a] * sum_r rowscale[r] 1[H[r, a] > 0]
//   dwo[a]   = sum_r rowscale[r] H[r, a]                                         (the weight gradient of the (A -> 1) layer behind H)
// The row scale moves to the X side (one multiply per element), which leaves the gate as the B operand: ONE exact bf16 plane, three
// products per tile pair instead of six.  The reduction runs over the ROWS, so a block owns the whole [Kd, A] output for a
// contiguous range of rows (one partial slab per block: grid slabs) and the product is HBM-bound (each operand row is read once: 6.2 GB at
// the reference point against 0.5 ms of matrix pipe) -- the schedule is the simple one: convert, barrier, multiply, with the loads of two
// groups (32 rows each) in flight.  Wave w splits X's columns [16 TK w, 16 TK (w + 1)) in registers and turns H's columns
// [16 TA w, 16 TA (w + 1)) into gate fragments that all the waves read from LDS (16 KB per group at A = 256, double-buffered); it also owns those
// columns' two sums.  Both operands are read with one 16-byte (TK, TA = 4) or 8-byte (= 2) load per lane and row: lane (c, q) takes columns
// T c .. T c + T - 1 of rows 8 q .. 8 q + 7, i.e. element i of the load belongs to tile i, whose fragment column c is matrix column T c + i.
struct TswArgs {
    const float* X; int ldx;            // [M, Kd] (GEN: unused)
    const float* H; int ldh;            // [M, A]
    const float* rowscale;              // [M]
    const float* colscale;              // [A]
    float* dw; int64_t dw_stride;       // slab b: dw + b * dw_stride, row-major [Kd, A]
    float* db; int64_t db_stride;       // slab b: db + b * db_stride, [A]
    float* dwo; int64_t dwo_stride;
    int64_t M;
    int rows_per_block;                 // a multiple of 32; gridDim.x * rows_per_block >= M
    // GEN: X[r, :] = e[b, pair_i[p], :] . e[b, pair_j[p], :] for r = b P + p, formed in the registers from the gathered embeddings
    // (as TsArgs; field f at f * Kd of an example's e_ld floats)
    const float* e;
    int e_ld;
    int64_t e_floats;
    const int16_t* pair_i;
    const int16_t* pair_j;
    int P;
    // HB: H is not read.  Its signs come from the forward's sign words (TsArgs::bits_out: 32 bytes per row), and the second column sums
    // follow from the product itself:  sum_r rowscale[r] H[r, a] = sum_r rs[r] 1[H > 0] (sum_k X[r, k] W[k, a] + b[a])
    //                                                            = sum_k W[k, a] dWraw[k, a] + b[a] dbraw[a]
    // with dWraw / dbraw this kernel's accumulators before the column scale -- 32 bytes per row instead of 1 KB, and nobody reads H any more.
    const unsigned long long* bits;
    const float* W;             // [Kd, A] row-major: the layer's weight
    const float* bias;          // [A]
};

// NW waves per block: X has 16 TK NW columns, H has 16 TA NW.  At 256 x 256 the block is EIGHT waves of a [32, 256] output strip each (two
// per SIMD, 128 accumulator registers): with four waves of [64, 256] the 256 accumulators fill the AGPR half of the register file and the
// two raw sets + planes no longer fit the other half (355 registers spilled) -- and two waves per SIMD overlap one's conversion with the
// other's MFMAs for free.
template <int NW, int TK, int TA, bool GEN = false, bool HB = false>
__global__ __launch_bounds__(64 * NW, 1) void gemm_tsw_kernel(TswArgs a) {
    static_assert((TK == 2 || TK == 4) && (TA == 2 || TA == 4) && (NW == 4 || NW == 8), "two or four 16-column tiles per wave and operand");
    constexpr int NJ = NW * TA;                         // gate tiles of the block
    constexpr int KD = 16 * TK * NW;
    extern __shared__ __attribute__((aligned(16))) char ts_lds[];          // [2][NJ][64 lanes] x 16 bytes (+ GEN: the pair table, P words)
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_block;
    const int rows = (int)(a.M - r0 < a.rows_per_block ? (a.M - r0 > 0 ? a.M - r0 : 0) : a.rows_per_block);
    const int G = __builtin_amdgcn_readfirstlane((rows + 31) / 32);
    auto uni_ptr = [](const void* p) {
        const uint64_t v = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
    };
    // this block's rows behind per-block bases: whatever a partial or surplus group addresses beyond them reads as 0 without touching memory
    const auto rx = GEN ? __builtin_amdgcn_make_buffer_rsrc(uni_ptr(a.e), 0, __builtin_amdgcn_readfirstlane((int)(a.e_floats * 4)), 0x00020000)
                        : __builtin_amdgcn_make_buffer_rsrc(uni_ptr(a.X + (size_t)r0 * a.ldx), 0, __builtin_amdgcn_readfirstlane(rows > 0 ? ((rows - 1) * a.ldx + KD) * 4 : 0), 0x00020000);
    const auto rh = HB ? __builtin_amdgcn_make_buffer_rsrc(uni_ptr(a.bits + (size_t)r0 * 4), 0, __builtin_amdgcn_readfirstlane(rows * 32), 0x00020000)
                       : __builtin_amdgcn_make_buffer_rsrc(uni_ptr(a.H + (size_t)r0 * a.ldh), 0, __builtin_amdgcn_readfirstlane(rows > 0 ? ((rows - 1) * a.ldh + 16 * TA * NW) * 4 : 0), 0x00020000);
    // HB: this lane's TA columns 16 TA w + TA c .. are bits sh0 .. sh0 + TA - 1 of one 32-bit half of the sign word (row, q' = column quarter)
    const int col0 = 16 * TA * w + TA * c;
    const int sh0 = 4 * ((col0 >> 4) & 7) + (col0 & 3);
    const int hbo = 8 * q * 32 + ((col0 & 15) >> 2) * 8 + ((col0 >> 4) >= 8 ? 4 : 0);          // byte offset of (row 8 q, that word half)
    const auto rr = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(a.rowscale + r0), 0, __builtin_amdgcn_readfirstlane(rows * 4), 0x00020000);
    const int xoff = 4 * ((GEN ? 0 : 8 * q * a.ldx) + 16 * TK * w + TK * c), hoff = HB ? hbo : 4 * (8 * q * a.ldh + 16 * TA * w + TA * c);
    const unsigned sx = 4u * (unsigned)a.ldx, sh = HB ? 32u : 4u * (unsigned)a.ldh;
    // GEN: pair p -> the two fields' offsets inside an example (floats, 16 bits each), in LDS; this lane's (example, pair) of the first of
    // its 8 rows in the next group to load, advanced by 32 rows per group (P > 32: at most one carry)
    unsigned* tab = reinterpret_cast<unsigned*>(ts_lds + 2 * NJ * 1024);
    int bx = 0, px = 0;
    const int nex = GEN ? (int)(a.e_floats / a.e_ld) : 0;
    if constexpr (GEN) {
        for (int p = t; p < a.P; p += 64 * NW) tab[p] = (unsigned)(a.pair_i[p] * KD) | ((unsigned)(a.pair_j[p] * KD) << 16);
        const unsigned row = (unsigned)(r0 + 8 * q);
        bx = (int)(row / (unsigned)a.P);
        px = (int)(row - (unsigned)bx * (unsigned)a.P);
        __syncthreads();
    }

    struct RawH { float h[HB ? 1 : 8][TA]; unsigned hb[HB ? 8 : 1]; float rs[8]; };
    struct RawX { float x[8][TK]; float y[GEN ? 8 : 1][TK]; };
    auto ldn = [](auto rs, int voff, unsigned soff_, auto nt, float* d) {
        constexpr int NV = decltype(nt)::value;
        const unsigned soff = __builtin_amdgcn_readfirstlane(soff_);
        if constexpr (NV == 4) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = __uint_as_float(v[e]);
        } else {
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
            d[0] = __uint_as_float(v[0]);
            d[1] = __uint_as_float(v[1]);
        }
    };
    using ITK = std::integral_constant<int, TK>;
    using ITA = std::integral_constant<int, TA>;
    using I4 = std::integral_constant<int, 4>;
    auto loadH = [&](RawH& f, int g) {
        ldn(rr, 32 * q, 128u * g, I4{}, &f.rs[0]);
        ldn(rr, 32 * q + 16, 128u * g, I4{}, &f.rs[4]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if constexpr (HB) f.hb[e] = __builtin_amdgcn_raw_buffer_load_b32(rh, hoff, __builtin_amdgcn_readfirstlane((32u * g + e) * sh), 0);
            else ldn(rh, hoff, (32u * g + e) * sh, ITA{}, &f.h[e][0]);
        }
    };
    auto loadX = [&](RawX& f, int g) {
        if constexpr (GEN) {
            (void)g;                                    // (the rows of the group after the one loaded last: bx, px)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                int pe = px + e, be = bx;
                if (pe >= a.P) { pe -= a.P; ++be; }
                if (be >= nex) be = nex - 1;            // (rows past the end: any real pair; their row scale and gate are zero)
                const unsigned tw = tab[pe];
                const int base = be * a.e_ld;
                ldn(rx, xoff + 4 * (base + (int)(tw & 0xffffu)), 0u, ITK{}, &f.x[e][0]);
                ldn(rx, xoff + 4 * (base + (int)(tw >> 16)), 0u, ITK{}, &f.y[e][0]);
            }
            px += 32;
            if (px >= a.P) { px -= a.P; ++bx; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) ldn(rx, xoff, (32u * g + e) * sx, ITK{}, &f.x[e][0]);
        }
    };
    f32x4 acc[TK][NJ];
#pragma unroll
    for (int i = 0; i < TK; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float cs[TA], cs2[TA];
#pragma unroll
    for (int j = 0; j < TA; ++j) cs[j] = cs2[j] = 0.f;
    DrPlanes px_[TK];
    // raw -> X planes (registers), gate fragments (LDS image `buf`), the two column sums
    auto convert = [&](const RawX& fx, const RawH& f, int buf) {
#pragma unroll
        for (int i = 0; i < TK; ++i) {
            float xs[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) xs[e] = (GEN ? fx.x[e][i] * fx.y[e][i] : fx.x[e][i]) * f.rs[e];      // (GEN: the f32 product afm_pair_fwd_kernel stores, then the scale)
            dr_split3(xs, px_[i]);
        }
#pragma unroll
        for (int j = 0; j < TA; ++j) {
            u32x4 gw;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                if constexpr (HB) {
                    const bool g0 = (f.hb[2 * tt] >> (sh0 + j)) & 1u, g1 = (f.hb[2 * tt + 1] >> (sh0 + j)) & 1u;
                    gw[tt] = (g0 ? 0x3f80u : 0u) | (g1 ? 0x3f800000u : 0u);
                    cs[j] += (g0 ? f.rs[2 * tt] : 0.f) + (g1 ? f.rs[2 * tt + 1] : 0.f);
                } else {
                    const float h0 = f.h[2 * tt][j], h1 = f.h[2 * tt + 1][j];
                    gw[tt] = (h0 > 0.f ? 0x3f80u : 0u) | (h1 > 0.f ? 0x3f800000u : 0u);
                    cs[j] += (h0 > 0.f ? f.rs[2 * tt] : 0.f) + (h1 > 0.f ? f.rs[2 * tt + 1] : 0.f);
                    cs2[j] = fmaf(h0, f.rs[2 * tt], cs2[j]);
                    cs2[j] = fmaf(h1, f.rs[2 * tt + 1], cs2[j]);
                }
            }
            *reinterpret_cast<u32x4*>(ts_lds + ((buf * NJ + TA * w + j) * 64 + lane) * 16) = gw;
        }
    };
    auto products = [&](int buf) {
        const char* base = ts_lds + (buf * NJ * 64 + lane) * 16;
        u32x4 fg[2];
        fg[0] = *reinterpret_cast<const u32x4*>(base);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (j + 1 < NJ) fg[(j + 1) & 1] = *reinterpret_cast<const u32x4*>(base + 1024 * (j + 1));
            const u32x4 gq = fg[j & 1];
#pragma unroll
            for (int i = 0; i < TK; ++i) acc[i][j] = dr_mfma_bf16(px_[i].l, gq, acc[i][j]);
#pragma unroll
            for (int i = 0; i < TK; ++i) acc[i][j] = dr_mfma_bf16(px_[i].m, gq, acc[i][j]);
#pragma unroll
            for (int i = 0; i < TK; ++i) acc[i][j] = dr_mfma_bf16(px_[i].h, gq, acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // PIPE (the engine's form: sign bits, generated rows, eight waves): the conversion of group g + 1 is cut into NJ pieces -- one pair of
    // rows of one X tile (product, scale, three-plane split) or of one gate tile (bit tests, the first column sum) -- and piece j sits in
    // region j of group g's multiplication, between that column tile's six MFMAs: the VALU stream runs under the matrix pipe instead of in
    // front of the barrier (measured before: 0.67 ms of MFMAs + 0.65 ms of everything else, no overlap).  A second plane set takes the
    // conversion while the first feeds the MFMAs; the gate images alternate as before; one barrier per group.
    constexpr bool PIPE = HB && GEN && NW == 8 && TK == 2 && TA == 2;
    if constexpr (PIPE) {
        static_assert(NJ == 16, "eight X pieces + eight gate pieces");
        RawH hh;
        RawX xx;
        DrPlanes pq[2][TK];                             // [g & 1]: planes of group g
        u32x4 gwt;                                      // the gate tile being assembled
        // piece j of the conversion of (xx, hh) into pq[pn] / gate image `img`
        auto piece = [&](auto jc, int pn, int img) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j < 8) {
                constexpr int i = j / 4, tt = j % 4;
                const float x0 = xx.x[2 * tt][i] * xx.y[2 * tt][i] * hh.rs[2 * tt], x1 = xx.x[2 * tt + 1][i] * xx.y[2 * tt + 1][i] * hh.rs[2 * tt + 1];
                const unsigned h = dr_pk_bf16(x0, x1);
                const float r0_ = dr_sub(x0, __uint_as_float(h << 16)), r1_ = dr_sub(x1, __uint_as_float(h & 0xffff0000u));
                const unsigned m = dr_pk_bf16(r0_, r1_);
                const float s0 = dr_sub(r0_, __uint_as_float(m << 16)), s1 = dr_sub(r1_, __uint_as_float(m & 0xffff0000u));
                pq[pn][i].h[tt] = h; pq[pn][i].m[tt] = m; pq[pn][i].l[tt] = dr_pk_bf16(s0, s1);
            } else {
                constexpr int jj = (j - 8) / 4, tt = (j - 8) % 4;
                const bool g0 = (hh.hb[2 * tt] >> (sh0 + jj)) & 1u, g1 = (hh.hb[2 * tt + 1] >> (sh0 + jj)) & 1u;
                gwt[tt] = (g0 ? 0x3f80u : 0u) | (g1 ? 0x3f800000u : 0u);
                cs[jj] += (g0 ? hh.rs[2 * tt] : 0.f) + (g1 ? hh.rs[2 * tt + 1] : 0.f);
                if constexpr (tt == 3) *reinterpret_cast<u32x4*>(ts_lds + ((img * NJ + TA * w + jj) * 64 + lane) * 16) = gwt;
            }
        };
        if (G > 0) {
            loadH(hh, 0); loadX(xx, 0);
            dr_static_for<NJ>([&](auto jc) { piece(jc, 0, 0); });
            loadH(hh, 1); loadX(xx, 1);                 // (beyond the block's rows: zeros, no traffic)
        }
        __syncthreads();
        auto group = [&](int g, auto pc) {
            constexpr int pcur = decltype(pc)::value, pnxt = pcur ^ 1;
            const char* base = ts_lds + (pcur * NJ * 64 + lane) * 16;
            u32x4 fg[2];
            fg[0] = *reinterpret_cast<const u32x4*>(base);
            dr_static_for<NJ>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if constexpr (j + 1 < NJ) fg[(j + 1) & 1] = *reinterpret_cast<const u32x4*>(base + 1024 * (j + 1));
                const u32x4 gq = fg[j & 1];
#pragma unroll
                for (int i = 0; i < TK; ++i) acc[i][j] = dr_mfma_bf16(pq[pcur][i].l, gq, acc[i][j]);
#pragma unroll
                for (int i = 0; i < TK; ++i) acc[i][j] = dr_mfma_bf16(pq[pcur][i].m, gq, acc[i][j]);
#pragma unroll
                for (int i = 0; i < TK; ++i) acc[i][j] = dr_mfma_bf16(pq[pcur][i].h, gq, acc[i][j]);
                piece(jc, pnxt, pnxt);
                __builtin_amdgcn_sched_barrier(0);
            });
            // (asking for the X rows of group g + 2 pair by pair from region 8 on, as their registers free up, costs 151 spilled registers: 2.4 ms)
            loadH(hh, g + 2); loadX(xx, g + 2);
            __syncthreads();
        };
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        for (int g = 0; g < G; g += 2) {
            group(g, P0{});
            if (g + 1 < G) group(g + 1, P1{});          // (uniform)
        }
    } else {
    // H and the row scales (HBM) two groups ahead in two sets; X beside them -- or, generated from the cache-resident embeddings, ONE group
    // ahead in one set (twice the registers per row)
    RawH h0, h1;
    RawX x0, x1;
    if (G > 0) { loadH(h0, 0); loadX(x0, 0); }
    if (G > 1) { loadH(h1, 1); if (!GEN) loadX(x1, 1); }
    for (int g = 0; g < G; g += 2) {
        convert(x0, h0, 0);
        loadH(h0, g + 2);                               // (beyond the block's rows: zeros, no traffic)
        loadX(GEN ? x0 : x0, GEN ? g + 1 : g + 2);
        __syncthreads();
        products(0);
        if (g + 1 < G) {                                // (uniform)
            convert(GEN ? x0 : x1, h1, 1);
            loadH(h1, g + 3);
            loadX(GEN ? x0 : x1, GEN ? g + 2 : g + 3);
            __syncthreads();
            products(1);
        }
    }
    }
    // ---- epilogue: register r of lane (c, q), tile (i, j = TA wj + jj) is dW[16 TK w + TK (4 q + r) + i][16 TA wj + TA c + jj]
    constexpr int A_ = 16 * TA * NW;
    if constexpr (HB) {
        // the second column sums from the raw accumulators: this wave's share sum_{k in its strip} W[k, a] dWraw[k, a] for all A columns,
        // summed over the four q in the wave and over the NW waves through LDS (the gate images are free now)
        __syncthreads();
        float* red = reinterpret_cast<float*>(ts_lds);           // [NW + 1][A_]: the waves' shares, then the raw first sums
#pragma unroll
        for (int wj = 0; wj < NW; ++wj) {
            float sw[TA];
#pragma unroll
            for (int jj = 0; jj < TA; ++jj) sw[jj] = 0.f;
#pragma unroll
            for (int i = 0; i < TK; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float* wp = a.W + (size_t)(16 * TK * w + TK * (4 * q + r) + i) * A_ + 16 * TA * wj + TA * c;
#pragma unroll
                    for (int jj = 0; jj < TA; ++jj) sw[jj] = fmaf(wp[jj], acc[i][TA * wj + jj][r], sw[jj]);
                }
#pragma unroll
            for (int jj = 0; jj < TA; ++jj) {
                float v = sw[jj];
                v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
                if (q == 0) red[w * A_ + 16 * TA * wj + TA * c + jj] = v;
            }
        }
#pragma unroll
        for (int j = 0; j < TA; ++j) {
            float s1 = cs[j];
            s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
            cs[j] = s1;
            if (q == 0) red[NW * A_ + 16 * TA * w + TA * c + j] = s1;
        }
        __syncthreads();
        for (int col = t; col < A_; col += 64 * NW) {
            float v = a.bias[col] * red[NW * A_ + col];
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) v += red[ww * A_ + col];
            if (a.dwo != nullptr) a.dwo[(size_t)blockIdx.x * a.dwo_stride + col] = v;
            if (a.db != nullptr) a.db[(size_t)blockIdx.x * a.db_stride + col] = red[NW * A_ + col] * a.colscale[col];
        }
    }
    // (buffer stores: the lane part of the address is one register, tile and row are a scalar offset)
    const auto rw = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(a.dw + (size_t)blockIdx.x * a.dw_stride), 0, 16 * TK * NW * A_ * 4, 0x00020000);
    const int woff = 4 * (TK * 4 * q * A_ + TA * c);
#pragma unroll
    for (int wj = 0; wj < NW; ++wj) {
        float sc[TA];
#pragma unroll
        for (int jj = 0; jj < TA; ++jj) sc[jj] = a.colscale[16 * TA * wj + TA * c + jj];
#pragma unroll
        for (int i = 0; i < TK; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned so = __builtin_amdgcn_readfirstlane(4u * (unsigned)((16 * TK * w + TK * r + i) * A_ + 16 * TA * wj));
                if constexpr (TA == 4) {
                    // (16-byte stores: no scalar offset -- see gemm_ts_kernel's epilogue)
                    const f32x4 v = f32x4{acc[i][4 * wj][r] * sc[0], acc[i][4 * wj + 1][r] * sc[1], acc[i][4 * wj + 2][r] * sc[2], acc[i][4 * wj + 3][r] * sc[3]};
                    int vo = woff + (int)so;
                    asm volatile("" : "+v"(vo));
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rw, vo, 0, 0);
                } else {
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                    const f32x2 v = f32x2{acc[i][2 * wj][r] * sc[0], acc[i][2 * wj + 1][r] * sc[1]};
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rw, woff, so, 0);
                }
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    // the column sums of this wave's H columns: the four q hold different rows of the same columns
    if constexpr (!HB)
#pragma unroll
    for (int j = 0; j < TA; ++j) {
        float s1 = cs[j], s2 = cs2[j];
        s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
        const int col = 16 * TA * w + TA * c + j;
        if (q == 0) {
            if (a.db != nullptr) a.db[(size_t)blockIdx.x * a.db_stride + col] = s1 * a.colscale[col];
            if (a.dwo != nullptr) a.dwo[(size_t)blockIdx.x * a.dwo_stride + col] = s2;
        }
    }
}

}  // namespace dctr
