// Outer-PNN, backward of the first MLP layer's pair-product rows without the [B, P K K] gradient tensor
// (replaces tf.gradients through PNN.py:139-153 'Outer' + the first fully_connected, PNN.py:159-166):
//
//     dOP[b][(p,a,c)] = sum_h dh0[b][h] W0[F K + (p K + a) K + c][h]          (an exact-f32 MFMA product, reduction over h <= 256)
//     dE[b][i_p][a]  += sum_c dOP[b][(p,a,c)] e[b][j_p][c]
//     dE[b][j_p][c]  += sum_a dOP[b][(p,a,c)] e[b][i_p][a]
//
// One block owns 32 batch rows (two 16-row MFMA tiles) and a contiguous range of field pairs; its dh0 rows sit in LDS (32 KB) as the
// A operand of every product.  Each of its 12 waves (three per SIMD; two for K = 64: one contracts while the others multiply) takes a contiguous
// slice of the block's pairs and streams ITS rows of W0 straight from L2 into B fragments, double-buffered one chunk (64 MFMAs) ahead
// (the weight rows of consecutive (p, a) are one linear stream).  A (p, a) step is a 32 x K patch of dOP held in 4 K/16
// accumulators; it never leaves the registers: de_j accumulates over a in registers (one float atomic per element and pair), de_i
// is reduced over the 16 lanes of a row with DPP and added to dE right away.  dE already holds the flat rows' share (the ordinary
// dgrad product ran before), hence atomics: the waves of a block, and the blocks of a pair-split launch, meet in the same rows.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "common.h"
#include "ops.h"

namespace dctr {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a DPP row (every lane ends with the total)
__device__ __forceinline__ float row16_sum(float v) {
    v = dpp_add<0xB1>(v);       // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);       // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);      // row_half_mirror
    return dpp_add<0x140>(v);   // row_mirror
}

// waves per block: three per SIMD where the registers allow it (K <= 32: 162 VGPRs), else two.  With three, c4's backward went
// 76.4 -> 74.0 ms/step: one more wave's MFMAs to run under a neighbour's contraction.
constexpr int opnn_dgrad_waves(int KT) { return KT <= 2 ? 12 : 8; }

// KT = K / 16 column tiles per (p, a); NG = groups of 16 h (H <= 16 NG; columns beyond H are zero in the LDS copy of dh0)
template <int KT, int NG>
__global__ __launch_bounds__(64 * opnn_dgrad_waves(KT)) void opnn_dgrad_kernel(const float* __restrict__ dh, int lddh, int H, const float* __restrict__ w_outer,
                                                         const float* __restrict__ e, int e_ld, const int* __restrict__ pairs, int P,
                                                         int pairs_per_block, int B, float* __restrict__ dE, int de_ld) {
    // CG: groups of 16 h per B chunk (two chunk buffers of KT CG 4 registers each: <= 64 VGPRs); HS: LDS row stride
    constexpr int NWV = opnn_dgrad_waves(KT);
    constexpr int K = 16 * KT, CG = (NG / 2 < 8 / KT) ? NG / 2 : 8 / KT, NCH = NG / CG, HS = 16 * NG + 4;
    static_assert(NCH % 2 == 0 && CG >= 1, "an even number of chunks per (p, a): the buffer of a chunk is a compile-time choice");
    extern __shared__ __attribute__((aligned(16))) float dh_lds[];      // [32][HS]
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * 32;
    // ---- dh0 rows of this block -> LDS (zero beyond B rows / H columns)
    for (int idx = t; idx < 32 * 4 * NG; idx += 64 * NWV) {
        const int r = idx / (4 * NG), h4 = idx - r * (4 * NG);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m0 + r < B && 4 * h4 < H) v = *reinterpret_cast<const f32x4*>(dh + (size_t)(m0 + r) * lddh + 4 * h4);
        *reinterpret_cast<f32x4*>(&dh_lds[r * HS + 4 * h4]) = v;
    }
    __syncthreads();
    // ---- this wave's pairs
    const int pb = blockIdx.y * pairs_per_block, pe = min(P, pb + pairs_per_block);
    const int np = max(pe - pb, 0);
    const int p0 = __builtin_amdgcn_readfirstlane(pb + (int)((int64_t)np * w / NWV)), p1 = __builtin_amdgcn_readfirstlane(pb + (int)((int64_t)np * (w + 1) / NWV));
    if (p0 >= p1) return;

    auto uni_ptr = [](const float* p) {
        const uint64_t v = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
    };
    // weight rows from pair p0 on ((P - p0) K K rows of H floats: < 2^31 bytes, checked by the host); embeddings from row m0 on
    const size_t wrow0 = (size_t)p0 * K * K;
    const auto rw = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(w_outer + wrow0 * H), 0,
                                                      __builtin_amdgcn_readfirstlane((int)(((size_t)(P - p0) * K * K) * H * 4)), 0x00020000);
    const auto re = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(e + (size_t)m0 * e_ld), 0, __builtin_amdgcn_readfirstlane(max(B - m0, 0) * e_ld * 4), 0x00020000);
    int boff[KT];                       // B fragment lane offsets: weight row 16 t + c, h = 4 q (+ 16 per group)
#pragma unroll
    for (int tt = 0; tt < KT; ++tt) boff[tt] = 4 * ((16 * tt + c) * H + 4 * q);
    int eoff[2][4];                     // embedding row 16 i + 4 q + r (the rows of this lane's accumulator registers)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) eoff[i][r] = 4 * (16 * i + 4 * q + r) * e_ld;
    const int aoff = (c * HS + 4 * q) * 4;        // LDS byte offset of this lane's A fragment (tile 0, group 0)

    struct Chunk { float b[KT][CG][4]; };
    auto load_chunk = [&](Chunk& ck, unsigned soff_) {
        const unsigned soff = __builtin_amdgcn_readfirstlane(soff_);
#pragma unroll
        for (int g = 0; g < CG; ++g)
#pragma unroll
            for (int tt = 0; tt < KT; ++tt) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, boff[tt] + 64 * g, soff, 0);
#pragma unroll
                for (int s = 0; s < 4; ++s) ck.b[tt][g][s] = __uint_as_float(v[s]);
            }
    };
    f32x4 acc[2][KT];
    auto mma_chunk = [&](const Chunk& ck, int x) {              // chunk x of the current (p, a): h in [16 CG x, 16 CG (x + 1))
#pragma unroll
        for (int g = 0; g < CG; ++g) {
            f32x4 a[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                a[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(dh_lds) + aoff + 4 * (16 * i * HS + 16 * (x * CG + g)));
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int tt = 0; tt < KT; ++tt)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], ck.b[tt][g][s], acc[i][tt], 0, 0, 0);
        }
    };

    const unsigned pa_bytes = (unsigned)K * (unsigned)H * 4u;           // weight bytes per (p, a): K rows of H floats
    Chunk c0, c1;
    unsigned soff = 0;                                                  // byte offset of the current (p, a) behind wrow0
    load_chunk(c0, 0u);
    for (int p = p0; p < p1; ++p) {
        const int pr = pairs[p];
        const int ip = __builtin_amdgcn_readfirstlane(pr >> 16), jp = __builtin_amdgcn_readfirstlane(pr & 0xffff);
        float ej[2][KT][4], dj[2][KT][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int tt = 0; tt < KT; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ej[i][tt][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(re, eoff[i][r] + 4 * (16 * tt + c), 4u * (unsigned)(jp * K), 0));
                    dj[i][tt][r] = 0.f;
                }
        for (int a = 0; a < K; ++a) {
            // (the A fragments are re-read from LDS for every (p, a): hoisted out of this loop they would take 8 NG registers, and the
            // kernel lives on two waves per SIMD)
            asm volatile("" ::: "memory");
            float ei[2][4];
            const unsigned si = 4u * (unsigned)(ip * K + a);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) ei[i][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(re, eoff[i][r], si, 0));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int tt = 0; tt < KT; ++tt) acc[i][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int x = 0; x < NCH; x += 2) {
                load_chunk(c1, soff + 64u * CG * (x + 1));
                mma_chunk(c0, x);
                // (the last prefetch is chunk 0 of the next (p, a); beyond the wave's range it is read and never used)
                load_chunk(c0, x + 2 < NCH ? soff + 64u * CG * (x + 2) : soff + pa_bytes);
                mma_chunk(c1, x + 1);
            }
            soff += pa_bytes;
            // ---- contraction of the 32 x K patch
            float vi[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = 0.f;
#pragma unroll
                    for (int tt = 0; tt < KT; ++tt) {
                        dj[i][tt][r] += acc[i][tt][r] * ei[i][r];
                        v += acc[i][tt][r] * ej[i][tt][r];
                    }
                    vi[i][r] = row16_sum(v);
                }
            // lane c = 4 i + r (c < 8) commits de_i[a] of row 16 i + 4 q + r
            float mine = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) mine = (c == 4 * i + r) ? vi[i][r] : mine;
            const int row = m0 + 16 * (c >> 2) + 4 * q + (c & 3);
            if (c < 8 && row < B) unsafeAtomicAdd(dE + (size_t)row * de_ld + ip * K + a, mine);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int tt = 0; tt < KT; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + 16 * i + 4 * q + r;
                    if (row < B) unsafeAtomicAdd(dE + (size_t)row * de_ld + jp * K + 16 * tt + c, dj[i][tt][r]);
                }
    }
}

template <int KT, int NG>
int launch(const float* dh, int lddh, int H, const float* w_outer, const float* e, int e_ld, const int* pairs, int P, int B, float* dE, int de_ld,
           hipStream_t st) {
    auto kern = opnn_dgrad_kernel<KT, NG>;
    constexpr int NWV = opnn_dgrad_waves(KT);
    constexpr size_t lds = (size_t)32 * (16 * NG + 4) * sizeof(float);
    const int nbm = ceil_div(B, 32);
    int splits = std::max(1, 256 / nbm);                    // small batches: the pairs are split over blocks as well
    splits = std::min(splits, std::max(1, P / NWV));        // (>= one pair per wave)
    const int ppb = ceil_div(P, splits);
    kern<<<dim3((unsigned)nbm, (unsigned)ceil_div(P, ppb)), 64 * NWV, lds, st>>>(dh, lddh, H, w_outer, e, e_ld, pairs, P, ppb, B, dE, de_ld);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

}  // namespace

bool opnn_dgrad_fused_ok(int K, int H) { return (K == 16 || K == 32 || K == 64) && H <= 256 && (H & 3) == 0; }

// dE[B, >= F K] += the pair-product rows' share of dL/de.  dh: [B, H] gradient at the first layer's pre-activation; w_outer: rows
// [F K, F K + P K K) of the first layer's weight.
int opnn_outer_dgrad_fused(const float* dh, int lddh, int H, const float* w_outer, const float* e, int e_ld, const int* pairs, int B, int F, int K,
                           float* dE, int de_ld, hipStream_t st) {
    if (B <= 0) return DCTR_OK;
    const int P = F * (F - 1) / 2;
    DCTR_REQUIRE(opnn_dgrad_fused_ok(K, H) && (lddh & 3) == 0 && (reinterpret_cast<uintptr_t>(dh) & 15) == 0 && (reinterpret_cast<uintptr_t>(w_outer) & 15) == 0 &&
                     (int64_t)P * K * K * H * 4 < (int64_t)0x7fff0000 && (int64_t)B * e_ld * 4 < (int64_t)0x7fff0000,
                 "opnn_outer_dgrad_fused: unsupported shape / alignment");
    const int ng = H <= 64 ? 4 : (H <= 128 ? 8 : 16);
#define DCTR_OPNN_DGRAD(KT_)                                                                                                  \
    switch (ng) {                                                                                                             \
        case 4: return launch<KT_, 4>(dh, lddh, H, w_outer, e, e_ld, pairs, P, B, dE, de_ld, st);                             \
        case 8: return launch<KT_, 8>(dh, lddh, H, w_outer, e, e_ld, pairs, P, B, dE, de_ld, st);                             \
        default: return launch<KT_, 16>(dh, lddh, H, w_outer, e, e_ld, pairs, P, B, dE, de_ld, st);                           \
    }
    if (K == 16) { DCTR_OPNN_DGRAD(1) }
    if (K == 32) { DCTR_OPNN_DGRAD(2) }
    DCTR_OPNN_DGRAD(4)
#undef DCTR_OPNN_DGRAD
}

}  // namespace dctr

// ---- op-level C ABI of the fused Outer-PNN first layer (include/deepctr_hip.h) ---------------------------------------------------
#include <map>
#include <mutex>
#include <vector>

using namespace dctr;

namespace {

// device copy of the pair table (i << 16 | j, the reference's double loop order, PNN.py:142-146), one per field count
int pair_table(int F, const int** out) {
    static std::mutex mu;
    static std::map<int, int*> tables;
    std::lock_guard<std::mutex> lock(mu);
    auto it = tables.find(F);
    if (it == tables.end()) {
        std::vector<int> pairs;
        for (int i = 0; i < F; ++i)
            for (int j = i + 1; j < F; ++j) pairs.push_back(i << 16 | j);
        int* d = nullptr;
        DCTR_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d), std::max<size_t>(pairs.size(), 1) * sizeof(int)));
        DCTR_HIP_CHECK(hipMemcpy(d, pairs.data(), pairs.size() * sizeof(int), hipMemcpyHostToDevice));
        it = tables.emplace(F, d).first;
    }
    *out = it->second;
    return DCTR_OK;
}

}  // namespace

extern "C" {

size_t dctr_pnn_outer_fc_workspace_bytes(int max_batch, int H) { return (size_t)opnn_fwd_ws_floats_max(max_batch, H) * sizeof(float); }

int dctr_pnn_outer_fc_fwd(const float* d_e, int e_ld, int B, int F, int K, const float* d_w, const float* d_b, float* d_y, int ldy, int H, int relu,
                          float keep, uint64_t seed, float* d_workspace, size_t workspace_bytes, void* stream) {
    DCTR_REQUIRE(keep > 0.f && keep <= 1.f, "keep_prob must be in (0,1], got %f", keep);
    DCTR_REQUIRE(opnn_fused_ok(F, K, H), "dctr_pnn_outer_fc_fwd: needs K a power of two >= 16 and H a multiple of 4 (got K=%d H=%d)", K, H);
    DCTR_REQUIRE(d_workspace != nullptr && workspace_bytes >= dctr_pnn_outer_fc_workspace_bytes(B, H), "dctr_pnn_outer_fc_fwd: workspace too small");
    const int* pairs = nullptr;
    DCTR_TRY(pair_table(F, &pairs));
    hipStream_t st = as_stream(stream);
    const int D = F * K;
    DCTR_TRY(fc_fwd(d_e, e_ld, d_w, nullptr, d_y, ldy, B, D, H, 0, 1.f, nullptr, 0, st));            // rows [0, F K): the flat embeddings
    return opnn_outer_fwd(d_e, e_ld, B, F, K, pairs, d_w + (size_t)D * H, d_b, d_y, ldy, H, relu, keep, nullptr, seed, d_workspace, st);
}

int dctr_pnn_outer_fc_bwd_weights(const float* d_e, int e_ld, int B, int F, int K, const float* d_dy, int lddy, int H, float* d_dw, float* d_db,
                                  void* stream) {
    DCTR_REQUIRE(opnn_fused_ok(F, K, H), "dctr_pnn_outer_fc_bwd_weights: needs K a power of two >= 16 and H a multiple of 4 (got K=%d H=%d)", K, H);
    const int* pairs = nullptr;
    DCTR_TRY(pair_table(F, &pairs));
    hipStream_t st = as_stream(stream);
    const int D = F * K;
    DCTR_TRY(fc_bwd_weights_partials(d_e, e_ld, d_dy, lddy, d_dw, 0, d_db, 0, B, D, H, 1, st));
    return opnn_outer_wgrad(d_e, e_ld, B, F, K, pairs, d_dy, lddy, H, d_dw + (size_t)D * H, st);
}

int dctr_pnn_outer_fc_bwd_data(const float* d_e, int e_ld, int B, int F, int K, const float* d_dy, int lddy, int H, const float* d_w, float* d_dE,
                               int de_ld, void* stream) {
    DCTR_REQUIRE(opnn_fused_ok(F, K, H) && opnn_dgrad_fused_ok(K, H),
                 "dctr_pnn_outer_fc_bwd_data: needs K in {16, 32, 64} and H <= 256, a multiple of 4 (got K=%d H=%d)", K, H);
    const int* pairs = nullptr;
    DCTR_TRY(pair_table(F, &pairs));
    hipStream_t st = as_stream(stream);
    const int D = F * K;
    DCTR_TRY(fc_bwd_data(d_dy, lddy, d_w, d_dE, de_ld, B, D, H, nullptr, 0, 1.f, st));             // overwrites dE[:, :F K] with the flat rows' share
    return opnn_outer_dgrad_fused(d_dy, lddy, H, d_w + (size_t)D * H, d_e, e_ld, pairs, B, F, K, d_dE, de_ld, st);
}

}  // extern "C"
