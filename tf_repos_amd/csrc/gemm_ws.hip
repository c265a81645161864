// K6, third kernel shape: "weights-stationary" exact-f32 MFMA product for TALL operands -- C[M, N] = A[M, R] B[R, N] with M in the
// millions and R, N <= 256: AFM's attention layer over the B*P pair rows (AFM.py:142-145 at the reference's K = A = 256: 3.0 M
// rows, 398 GFLOP per product).  Replaces contrib.layers.fully_connected forward and its input gradient like gemm.hip does.
//
// The LDS-tiled kernel (gemm.hip) re-stages the same 256 KB of weights for every 64-row tile of the tall operand and pays a
// barrier per 16 k; the direct kernel (gemm_dr.h) is built for one-round grids.  Here the small operand is loaded ONCE per block
// into LDS (a 128-column slab: 128 KB), already permuted into MFMA B-fragment order, and stays; the block's 8 waves (two per
// SIMD) are independent from then on: each walks its own 32-row tiles of A, streaming the rows straight from HBM into A
// fragments (one dwordx4 per lane and 16 k, double-buffered 4 groups = 256 MFMAs ahead), with one ds_read_b128 per B fragment.
// No split of the reduction, no cross-wave reduction, no barrier after the fill.  The epilogue (bias / ReLU, or the
// ReLU mask of the layer below) works on the accumulators and stores 64-byte row segments.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "common.h"
#include "ops.h"

namespace dctr {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// WS_GATE: the tall operand is a ReLU OUTPUT and only its sign enters the product (1 where > 0) --
//   C[row, :] = rowscale[row] * sum_k 1[A[row, k] > 0] * (kscale[k] B[k, :])
// i.e. the input gradient of a layer whose own output gradient is rank one under the ReLU mask, d h = rowscale (x) kscale . 1[h > 0]
// (AFM's last attention layer: rowscale = d score, kscale = attention_out's weight), without that [M, R] gradient in memory.
enum { WS_STORE = 0, WS_BIAS_ACT = 1, WS_MASK = 2, WS_GATE = 3 };

struct WsEpilogue {
    const float* bias;
    int relu;
    float keep;
    uint64_t seed;
    const uint64_t* seed_ptr;
    const float* act;
    int ldact;
    float inv_keep;
    const float* rowscale;   // WS_GATE
    const float* kscale;     // WS_GATE
    const float* dot_w;      // WS_BIAS_ACT: also dot_out[blockIdx.y * dot_stride + row] = sum over this block's columns of C[row, col] * dot_w[col]
    float* dot_out;
    int64_t dot_stride;
};

constexpr int WS_NT = 8;            // 16-column tiles per block: a 128-column slab of the stationary operand
constexpr int WS_WAVES = 8;
constexpr int WS_CG = 4;            // groups of 16 reduction steps per A chunk

// KG = reduction length / 16 (a multiple of 2 WS_CG: the chunk buffers alternate statically)
template <int KG, int EPI>
__global__ __launch_bounds__(64 * WS_WAVES) void gemm_ws_kernel(const float* __restrict__ A, int lda, const float* __restrict__ Bm, int ldb,
                                                                int b_trans, float* __restrict__ C, int ldc, int64_t M, int N, int R,
                                                                WsEpilogue ep) {
    constexpr int NCH = KG / WS_CG;
    static_assert(KG % (2 * WS_CG) == 0, "an even number of chunks");
    extern __shared__ __attribute__((aligned(16))) float ws_lds[];     // [KG][4 q][NT][16 c][4 s]
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int n0 = blockIdx.y * 16 * WS_NT;
    // ---- the stationary slab -> LDS in B-fragment order: lane (c, q) of tile tt, group g reads its 4 steps as ONE float4
    for (int idx = t; idx < 16 * KG * 16 * WS_NT; idx += 64 * WS_WAVES) {
        const int k = idx / (16 * WS_NT), n = idx - k * (16 * WS_NT);
        float v = 0.f;
        if (k < R && n0 + n < N) v = b_trans ? Bm[(size_t)(n0 + n) * ldb + k] : Bm[(size_t)k * ldb + n0 + n];
        if (EPI == WS_GATE && k < R) v *= ep.kscale[k];
        const int g = k >> 4, qq = (k >> 2) & 3, s = k & 3, tt = n >> 4, cc = n & 15;
        ws_lds[((((g * 4 + qq) * WS_NT + tt) * 16 + cc) << 2) + s] = v;
    }
    __syncthreads();
    const int boff = ((q * WS_NT) * 16 + c) * 16;              // byte offset of this lane's fragment of (g = 0, tile 0)

    struct Chunk { float a[2][WS_CG][4]; };
    const int64_t n_tiles = (M + 31) / 32;
    const int64_t stride = (int64_t)gridDim.x * WS_WAVES;
    int64_t tile = (int64_t)blockIdx.x * WS_WAVES + w;
    if (tile >= n_tiles) return;
    auto uni_ptr = [](const float* p) {
        const uint64_t v = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
    };
    // A rows of one tile behind a per-tile base: 32 rows of lda floats (rows beyond M: num_records ends at the last real row)
    auto tile_rsrc = [&](int64_t tl) {
        const int64_t m0 = tl * 32;
        const int rows = (int)(M - m0 < 32 ? (M - m0 > 0 ? M - m0 : 0) : 32);
        return __builtin_amdgcn_make_buffer_rsrc(uni_ptr(A + (size_t)m0 * lda), 0, __builtin_amdgcn_readfirstlane(rows > 0 ? ((rows - 1) * lda + R) * 4 : 0),
                                                 0x00020000);
    };
    int aoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) aoff[i] = 4 * ((16 * i + c) * lda + 4 * q);
    auto load_chunk = [&](Chunk& ck, decltype(tile_rsrc(0)) rs, int x) {      // groups [WS_CG x, WS_CG (x + 1)) of the tile behind rs
#pragma unroll
        for (int g = 0; g < WS_CG; ++g)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, aoff[i] + 64 * g, 64u * WS_CG * x, 0);
#pragma unroll
                for (int s = 0; s < 4; ++s) ck.a[i][g][s] = __uint_as_float(v[s]);
            }
    };
    f32x4 acc[2][WS_NT];
    auto mma_chunk = [&](const Chunk& ck, int x) {
#pragma unroll
        for (int g = 0; g < WS_CG; ++g) {
            f32x4 b[WS_NT];
#pragma unroll
            for (int tt = 0; tt < WS_NT; ++tt)
                b[tt] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(ws_lds) + boff + 16 * 16 * tt + 16 * 16 * WS_NT * 4 * (x * WS_CG + g));
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int tt = 0; tt < WS_NT; ++tt)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[i][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(EPI == WS_GATE ? (ck.a[i][g][s] > 0.f ? 1.f : 0.f) : ck.a[i][g][s], b[tt][s], acc[i][tt], 0, 0, 0);
        }
    };
    float bias[WS_NT];
#pragma unroll
    for (int tt = 0; tt < WS_NT; ++tt) bias[tt] = (EPI == WS_BIAS_ACT && ep.bias != nullptr && n0 + 16 * tt + c < N) ? ep.bias[n0 + 16 * tt + c] : 0.f;
    float dotw[WS_NT];
#pragma unroll
    for (int tt = 0; tt < WS_NT; ++tt) dotw[tt] = (EPI == WS_BIAS_ACT && ep.dot_w != nullptr && n0 + 16 * tt + c < N) ? ep.dot_w[n0 + 16 * tt + c] : 0.f;

    Chunk c0, c1;
    auto rs = tile_rsrc(tile);
    load_chunk(c0, rs, 0);
    while (true) {
        // (the fragments of the stationary operand are re-read from LDS for every tile: hoisted they would take 4 KG NT registers)
        asm volatile("" ::: "memory");
        const int64_t next = tile + stride;
        const bool more = next < n_tiles;
        auto rs_next = tile_rsrc(more ? next : tile);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int tt = 0; tt < WS_NT; ++tt) acc[i][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
        // WS_GATE: this tile's row scales, asked for BEFORE the products -- loaded in the epilogue they sit behind the next tile's
        // first A chunk in the (in-order) load queue, and every tile waited out an HBM latency for them (3.33 -> 3.88 ms over 3 M rows)
        float rsc[2][4];
        if (EPI == WS_GATE) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t row = tile * 32 + 16 * i + 4 * q + r;
                    rsc[i][r] = row < M ? ep.rowscale[row] : 0.f;
                }
        }
#pragma unroll
        for (int x = 0; x < NCH; x += 2) {
            load_chunk(c1, rs, x + 1);
            mma_chunk(c0, x);
            if (x + 2 < NCH) load_chunk(c0, rs, x + 2);
            else load_chunk(c0, rs_next, 0);              // chunk 0 of the next tile (the last tile re-reads its own: never used)
            mma_chunk(c1, x + 1);
        }
        // ---- epilogue on the accumulators: register r of lane (c, q) is row 16 i + 4 q + r, column 16 tt + c
        const int64_t m0 = tile * 32;
        float dots[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = m0 + 16 * i + 4 * q + r;
                const bool live = row < M;
                const float rs = EPI == WS_GATE ? rsc[i][r] : 1.f;
                float dot = 0.f;
#pragma unroll
                for (int tt = 0; tt < WS_NT; ++tt) {
                    const int col = n0 + 16 * tt + c;
                    float v = acc[i][tt][r];
                    if (EPI == WS_BIAS_ACT) {
                        v += bias[tt];
                        if (ep.relu) v = fmaxf(v, 0.f);
                        dot += v * dotw[tt];
                    } else if (EPI == WS_MASK) {
                        if (live && col < N) v = ep.act[(size_t)row * ep.ldact + col] > 0.f ? v * ep.inv_keep : 0.f;
                    } else if (EPI == WS_GATE) {
                        v *= rs;
                    }
                    if (live && col < N) C[(size_t)row * ldc + col] = v;
                }
                dots[i][r] = dot;
            }
        if (EPI == WS_BIAS_ACT && ep.dot_out != nullptr) {       // (wave-uniform branch)
            // the 16 lanes of quarter q hold pieces of the same 8 rows: butterfly sums leave every lane with all 8 totals; lane
            // c < 8 keeps total (i, r) = (c >> 2, c & 3) -- ONE store of 32 rows per tile instead of eight 4-lane ones
            float mine = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float d = dots[i][r];
                    d += __shfl_xor(d, 1); d += __shfl_xor(d, 2); d += __shfl_xor(d, 4); d += __shfl_xor(d, 8);
                    if (c == 4 * i + r) mine = d;
                }
            const int64_t row = m0 + 16 * (c >> 2) + 4 * q + (c & 3);
            if (c < 8 && row < M) ep.dot_out[(size_t)blockIdx.y * ep.dot_stride + row] = mine;
        }
        if (!more) break;
        tile = next;
        rs = rs_next;
    }
}

template <int KG, int EPI>
int launch_ws(const float* A, int lda, const float* Bm, int ldb, int b_trans, float* C, int ldc, int64_t M, int N, int R, const WsEpilogue& ep,
              hipStream_t st) {
    auto kern = gemm_ws_kernel<KG, EPI>;
    constexpr size_t lds = (size_t)16 * KG * 16 * WS_NT * sizeof(float);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    DCTR_HIP_CHECK(attr);
    const int ny = ceil_div(N, 16 * WS_NT);
    const int nx = std::max(1, 256 / ny);
    kern<<<dim3((unsigned)nx, (unsigned)ny), 64 * WS_WAVES, lds, st>>>(A, lda, Bm, ldb, b_trans, C, ldc, M, N, R, ep);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

// A/B knob DCTR_GEMM_WS=0: tall products back on the LDS-tiled kernel.  AFM at K = A = 256, B = 4096 (3.0 M pair rows, 398 GF per
// product): forward 4.15 -> 3.46 ms (115 TF = 0.73 of peak), input gradient 4.46 -> 3.35 ms, step 17.8 -> 16.7 ms.
bool ws_enabled() {
    static const bool on = [] { const char* e = getenv("DCTR_GEMM_WS"); return e == nullptr || e[0] != '0'; }();
    return on;
}
bool ws_shape_ok(int64_t M, int R, int N) { return M >= 65536 && (R == 128 || R == 256) && N >= 64 && N <= 512; }
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int EPI>
int dispatch_ws(const float* A, int lda, const float* Bm, int ldb, int b_trans, float* C, int ldc, int64_t M, int N, int R, const WsEpilogue& ep,
                hipStream_t st) {
    if (R == 128) return launch_ws<8, EPI>(A, lda, Bm, ldb, b_trans, C, ldc, M, N, R, ep, st);
    return launch_ws<16, EPI>(A, lda, Bm, ldb, b_trans, C, ldc, M, N, R, ep, st);
}

}  // namespace

// host logic: would a product C[M, N] over a reduction of R take this kernel (alignment permitting, no dropout epilogue)?
bool ws_takes(int64_t M, int R, int N) { return ws_enabled() && ws_shape_ok(M, R, N); }

// Y[M,N] = act(X[M,K] W[K,N] + b)
int ws_fc_fwd(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int M, int K, int N, int relu, float keep,
              const uint64_t* seed_ptr, uint64_t seed, hipStream_t st, bool* done) {
    *done = false;
    if (keep < 1.0f) return DCTR_OK;          // (no dropout epilogue here: 64 counter-RNG evaluations per lane and tile do not fit the registers)
    if (!ws_enabled() || !ws_shape_ok(M, K, N) || !al16(x) || (ldx & 3) != 0 || (int64_t)32 * ldx * 4 >= (int64_t)0x7fff0000) return DCTR_OK;
    WsEpilogue ep{};
    ep.bias = b; ep.relu = relu; ep.keep = keep; ep.seed = seed; ep.seed_ptr = seed_ptr;
    *done = true;
    return dispatch_ws<WS_BIAS_ACT>(x, ldx, w, N, 0, y, ldy, M, N, K, ep, st);
}

// the same with the score dot of the NEXT (N -> 1) layer taken from the accumulators: dot_parts[j * dot_stride + row] = sum over
// column slab j (128 columns each, *n_parts of them) of Y[row, col] * dot_w[col] -- the caller adds the parts (and the bias)
int ws_fc_fwd_dot(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int M, int K, int N, int relu,
                  const float* dot_w, float* dot_parts, int64_t dot_stride, int* n_parts, hipStream_t st, bool* done) {
    *done = false;
    if (!ws_enabled() || !ws_shape_ok(M, K, N) || !al16(x) || (ldx & 3) != 0 || (int64_t)32 * ldx * 4 >= (int64_t)0x7fff0000) return DCTR_OK;
    WsEpilogue ep{};
    ep.bias = b; ep.relu = relu; ep.keep = 1.f; ep.dot_w = dot_w; ep.dot_out = dot_parts; ep.dot_stride = dot_stride;
    *n_parts = ceil_div(N, 16 * WS_NT);
    *done = true;
    return dispatch_ws<WS_BIAS_ACT>(x, ldx, w, N, 0, y, ldy, M, N, K, ep, st);
}

// dX[M,K] = (rowscale (x) kscale . 1[H > 0]) W[K,N]^T with H [M,N] the layer's ReLU output (WS_GATE above)
int ws_fc_bwd_data_gate(const float* h, int ldh, const float* rowscale, const float* kscale, const float* w, float* dx, int lddx, int M, int K,
                        int N, hipStream_t st, bool* done) {
    *done = false;
    if (!ws_enabled() || !ws_shape_ok(M, N, K) || !al16(h) || (ldh & 3) != 0 || (int64_t)32 * ldh * 4 >= (int64_t)0x7fff0000) return DCTR_OK;
    WsEpilogue ep{};
    ep.rowscale = rowscale; ep.kscale = kscale;
    *done = true;
    return dispatch_ws<WS_GATE>(h, ldh, w, N, 1, dx, lddx, M, K, N, ep, st);
}

// dX[M,K] = dY[M,N] W[K,N]^T (x ReLU mask of the producing layer): the stationary operand is W^T
int ws_fc_bwd_data(const float* dy, int lddy, const float* w, float* dx, int lddx, int M, int K, int N, const float* act, int ldact,
                   float keep_prev, hipStream_t st, bool* done) {
    *done = false;
    if (!ws_enabled() || !ws_shape_ok(M, N, K) || !al16(dy) || (lddy & 3) != 0 || (int64_t)32 * lddy * 4 >= (int64_t)0x7fff0000) return DCTR_OK;
    WsEpilogue ep{};
    ep.act = act; ep.ldact = ldact; ep.inv_keep = act ? 1.0f / keep_prev : 1.f;
    *done = true;
    if (act != nullptr) return dispatch_ws<WS_MASK>(dy, lddy, w, N, 1, dx, lddx, M, K, N, ep, st);
    return dispatch_ws<WS_STORE>(dy, lddy, w, N, 1, dx, lddx, M, K, N, ep, st);
}

}  // namespace dctr
