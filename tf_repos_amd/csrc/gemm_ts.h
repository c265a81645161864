// K6, fifth kernel shape: split-precision products for TALL operands -- the three 398-GFLOP products of AFM's attention layer at the
// reference's K = A = 256 (AFM.py:142-147 and their gradients over the B * P = 3.0 M pair rows; run.sh:18).  Round 6.
//
// gemm_ws.hip runs them on the f32 matrix pipe (157 TF peak, 116-124 TF sustained: 3.2-3.4 ms each).  Here every f32 value enters as
// three bf16 planes (x = h + m + l exactly, gemm_dr.h dr_split3), the six products that matter (h h, h m, m h, h l, l h, m m) run on
// v_mfma_f32_16x16x32_bf16 (2.5 PF peak) and accumulate in f32: the arithmetic of gemm_dr3, for a shape where it is nearly free --
//   * a wave owns 64 rows x ALL 16 N output columns, so the split of one A fragment (44 VALU ops) is shared by 6 N MFMAs (96 at
//     N = 256: 0.46 VALU ops per MFMA, where the MLP kernels pay 1.6-3.2);
//   * the small operand (the weight, <= 256 x 256) is split ONCE per step by ts_wsplit_kernel into planes laid out in B-fragment
//     order, and streamed group by group (32 reduction steps: 3 planes x 16 KB) through a double-buffered LDS image by the LDS-DMA
//     path (global_load_lds, 1 KB per wave instruction, no registers, no ds_write) -- the four waves of a block share every
//     fragment they read;
//   * the ReLU gate of the input gradient (d ah = dsc (x) w_o . 1[ah > 0]) is ONE exact bf16 plane: three products, no split.
// A block = 4 waves (one per SIMD) = 256 rows of the tall operand per pass, persistent over the row tiles (grid = CUs).
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
    const float* A;             // the tall operand [M, 32 KG], row-major
    int lda;
    const u32x4* planes;        // ts_wsplit_kernel's output for the small operand
    float* C;                   // [M, 16 NT]
    int ldc;
    int64_t M;
    const float* bias;          // forward: [N]
    const float* dot_w;         // forward: dot_out[row] = sum_n C[row, n] dot_w[n] ([N], never null; dot_out may be)
    float* dot_out;
    const float* rowscale;      // gate: C[row, :] *= rowscale[row]
};

enum { TS_FWD = 0, TS_GATE = 1 };

typedef __attribute__((address_space(3))) void ts_lds_ptr;

// KG: reduction length / 32; NT: output columns / 16.
//   TS_FWD : C = relu(A W + bias), dot_out = C . dot_w        (A split in registers: 6 products)
//   TS_GATE: C = rowscale (x) (1[A > 0] Wt')                  (A is a ReLU output: one exact plane, 3 products)
template <int KG, int NT, int MODE>
__global__ __launch_bounds__(256, 1) void gemm_ts_kernel(TsArgs a) {
    constexpr int N = 16 * NT;
    constexpr int PLANE = 4 * N * 16;                  // bytes of one plane of one group
    constexpr int BUF = 3 * PLANE;
    constexpr int PIECES = BUF / 1024;                 // 1 KB LDS-DMA pieces per group
    extern __shared__ __attribute__((aligned(16))) char ts_lds[];
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int64_t nbt = (a.M + 255) / 256;
    int64_t bt = blockIdx.x;
    if (bt >= nbt) return;

    auto uni_ptr = [](const void* p) {
        const uint64_t v = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
    };
    // the planes of group g -> LDS image `buf`, contiguous in both: wave w takes the 1 KB pieces w, w + 4, ...  (buffer form: the lane part
    // of the address is ONE register, lane x 16, and the piece is a scalar offset -- with global_load_lds hipcc hoisted a 64-bit per-lane
    // address per piece and group out of the loop: 192 registers, spills)
    const auto rp = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(a.planes), 0, 3 * KG * PLANE, 0x00020000);
    auto stage = [&](int g, int buf) {
#pragma unroll
        for (int j = 0; j < PIECES / 4; ++j) {
            const int piece = 4 * j + w;
            const int p = piece / (PLANE / 1024), o = piece - p * (PLANE / 1024);
            const int so = __builtin_amdgcn_readfirstlane((p * KG + g) * PLANE + o * 1024);
#if defined(__HIP_DEVICE_COMPILE__)     // (hipcc's HOST pass drops the kernel's stub without a diagnostic when it meets this builtin)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (ts_lds_ptr*)(ts_lds + buf * BUF + piece * 1024), 16, lane * 16, so, 0, 0);
#else
            (void)so; (void)rp;
#endif
        }
    };
    // the rows of one block tile behind a per-wave base (rows beyond M: num_records ends at the last real row -> zeros, no traffic)
    auto rows_of = [&](int64_t tl) { const int64_t m0 = tl * 256 + 64 * w; return (int)(a.M - m0 < 64 ? (a.M - m0 > 0 ? a.M - m0 : 0) : 64); };
    auto a_rsrc = [&](int64_t tl) {
        const int rows = rows_of(tl);
        return __builtin_amdgcn_make_buffer_rsrc(uni_ptr(a.A + (size_t)(tl * 256 + 64 * w) * a.lda), 0,
                                                 __builtin_amdgcn_readfirstlane(rows > 0 ? ((rows - 1) * a.lda + 32 * KG) * 4 : 0), 0x00020000);
    };
    int aoff[4], coff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        aoff[i] = 4 * ((16 * i + c) * a.lda + 4 * q);
        coff[i] = 4 * ((16 * i + c) * a.ldc + 4 * q);
    }
    auto loadA = [&](float (&raw)[4][8], decltype(a_rsrc(0)) rs, int g) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(rs, aoff[i], 128 * g, 0);
            const u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(rs, aoff[i] + 64, 128 * g, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) { raw[i][e] = __uint_as_float(v0[e]); raw[i][4 + e] = __uint_as_float(v1[e]); }
        }
    };
    // raw -> the MFMA operand(s) of the rows
    struct Ops { u32x4 h[4], m[MODE == TS_FWD ? 4 : 1], l[MODE == TS_FWD ? 4 : 1]; };
    auto convert = [&](const float (&raw)[4][8], Ops& o) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (MODE == TS_FWD) {
                DrPlanes p;
                dr_split3(raw[i], p);
                o.h[i] = p.h; o.m[i] = p.m; o.l[i] = p.l;
            } else {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
                    o.h[i][tt] = (raw[i][2 * tt] > 0.f ? 0x3f80u : 0u) | (raw[i][2 * tt + 1] > 0.f ? 0x3f800000u : 0u);
            }
        }
    };
    const int boff = (q * N + c) * 16;                  // this lane's fragment of (plane 0, column tile 0)
    f32x4 acc[4][NT];
    // One region per column tile, fenced: the three fragments of tile tt + 1 are asked for ahead of tile tt's MFMAs and nothing else moves
    // across (left alone, hipcc hoists all 3 NT fragment reads of a group to its top: 192 registers, spills).
    auto products = [&](const Ops& o, int buf) {
        const char* base = ts_lds + buf * BUF + boff;
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
                for (int i = 0; i < 4; ++i) acc[i][tt] = dr_mfma_bf16(bm, o.m[i], acc[i][tt]);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][tt] = dr_mfma_bf16(bh, o.l[i], acc[i][tt]);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][tt] = dr_mfma_bf16(bl, o.h[i], acc[i][tt]);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][tt] = dr_mfma_bf16(bh, o.m[i], acc[i][tt]);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][tt] = dr_mfma_bf16(bm, o.h[i], acc[i][tt]);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][tt] = dr_mfma_bf16(bh, o.h[i], acc[i][tt]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][tt] = dr_mfma_bf16(bl, o.h[i], acc[i][tt]);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][tt] = dr_mfma_bf16(bm, o.h[i], acc[i][tt]);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][tt] = dr_mfma_bf16(bh, o.h[i], acc[i][tt]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    float raw[4][8];
    Ops cur, nxt;
    auto rs = a_rsrc(bt);
    stage(0, 0);
    loadA(raw, rs, 0);
    __syncthreads();                                    // (drains the LDS-DMA queue: vmcnt(0) before the barrier)
    convert(raw, cur);
    int buf = 0;
    while (true) {
        const int64_t next = bt + gridDim.x;
        const bool more = next < nbt;
        auto rs_next = a_rsrc(more ? next : bt);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) acc[i][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int64_t m0 = bt * 256 + 64 * w;
        // the per-row scalars of the epilogue, asked for before the products (in the epilogue they would queue behind the next tile's loads)
        float rsc[4];
        if constexpr (MODE == TS_GATE) {
#pragma unroll
            for (int i = 0; i < 4; ++i) rsc[i] = m0 + 16 * i + c < a.M ? a.rowscale[m0 + 16 * i + c] : 0.f;
        }
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            // everything asked for here lands under this group's MFMAs (3-6 k cycles) and is waited for at its end
            if (g + 1 < KG) {
                if (!(TS_SKIP & 2)) stage(g + 1, buf ^ 1);
                if (!(TS_SKIP & 1)) loadA(raw, rs, g + 1);
            } else {
                if (more && !(TS_SKIP & 2)) stage(0, buf ^ 1);
                if (!(TS_SKIP & 1)) loadA(raw, rs_next, 0);                 // (the last tile re-reads its own first group: nobody consumes it)
            }
            products(cur, buf);
            convert(raw, nxt);
            __syncthreads();
            cur = nxt;
            buf ^= 1;
        }
        // ---- epilogue on the accumulators: register r of lane (c, q) is row 16 i + c, column 16 tt + 4 q + r
        const int rows = rows_of(bt);
        const auto rc = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(a.C + (size_t)m0 * a.ldc), 0,
                                                          __builtin_amdgcn_readfirstlane(rows > 0 ? ((rows - 1) * a.ldc + N) * 4 : 0), 0x00020000);
        float dot[4] = {0.f, 0.f, 0.f, 0.f};
        auto ld4 = [&](const float* p, int tt) { return *reinterpret_cast<const f32x4*>(p + 16 * tt + 4 * q); };
        f32x4 bc = f32x4{0.f, 0.f, 0.f, 0.f}, dc = bc;
        if constexpr (MODE == TS_FWD) { bc = ld4(a.bias, 0); dc = ld4(a.dot_w, 0); }
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            f32x4 bn = bc, dn = dc;
            if constexpr (MODE == TS_FWD) {
                if (tt + 1 < NT) { bn = ld4(a.bias, tt + 1); dn = ld4(a.dot_w, tt + 1); }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v = acc[i][tt];
                if constexpr (MODE == TS_FWD) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = fmaxf(v[r] + bc[r], 0.f);
                        dot[i] += v[r] * dc[r];
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= rsc[i];
                }
                // (the column tile as the SCALAR offset: written into the lane offset, hipcc precomputes all 4 NT of them outside the loop)
                if (!(TS_SKIP & 4)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rc, coff[i], 64 * tt, 0);
            }
            bc = bn; dc = dn;
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (MODE == TS_FWD) {
            if (a.dot_out != nullptr) {                 // (uniform) the four q-lanes of a row hold its four column quarters
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float d = dot[i];
                    d += __shfl_xor(d, 16);
                    d += __shfl_xor(d, 32);
                    if (q == 0 && m0 + 16 * i + c < a.M) a.dot_out[m0 + 16 * i + c] = d;
                }
            }
        }
        if (!more) break;
        bt = next;
        rs = rs_next;
    }
}

}  // namespace dctr
