// K6, second kernel family (gemm_dr.h): host side -- instantiations, the per-shape tile chooser and the three layer products.
// gemm.hip's fc_fwd / fc_bwd_data / fc_bwd_weights_partials try these first and fall back to the LDS-tiled kernel when no
// tile of the list below fits the shape well (or an operand is not 16-byte aligned).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>

#include "common.h"
#include "gemm_dr.h"
#include "ops.h"

namespace dctr {

__device__ __forceinline__ float dr_dropout_scale(uint64_t seed, uint64_t idx, float keep) { return dropout_scale(seed, idx, keep); }

namespace {

constexpr int CUS = 256;

struct Tile { int tm, tn; };

// share of the chip's matrix-core time that an (Mo x No) output over a reduction of R cut in S splits turns into useful flops
// with tile (tm, tn): ideal cycles (64 flop / cycle / SIMD, 1024 SIMDs) over rounds x cycles of one block, the latter being its
// MFMA steps (whole 4-k steps per wave, tm*tn instructions of 32 cycles each) plus ~9000 cycles of prologue latency, cross-wave
// reduction and stores (tools/gemm_dr_probe.hip stamps)
double dr_efficiency(int Mo, int No, int64_t R, int S, Tile t, int* blocks_out) {
    const int64_t tiles = (int64_t)ceil_div(Mo, 16 * t.tm) * ceil_div(No, 16 * t.tn);
    const int64_t blocks = tiles * S;
    const int64_t rounds = (blocks + CUS - 1) / CUS;
    if (blocks_out) *blocks_out = (int)std::min<int64_t>(blocks, 1 << 30);
    const int64_t kchunk = round_up(ceil_div(R, S), 16);
    const int64_t kw = ((kchunk + 15) / 16) * 4;                  // k per wave
    const double steps = (double)((kw + 3) / 4);                  // MFMA steps per wave
    const double ideal = (double)Mo * No * (double)R / 32768.0;
    const double block_cycles = steps * t.tm * t.tn * 32.0 + 9000.0;
    return ideal / ((double)rounds * block_cycles);
}

// DCTR_GEMM = "lds": never; otherwise a subset of "fdw" (forward / dgrad / wgrad products that may take the direct kernel).
// Default "fdw".  Alone the direct kernel wins all three (c2 layer 0: fwd 25 vs 32 us, dgrad 26 vs 29, wgrad 27 vs 30).  In the
// training step each dgrad runs beside the weight gradient of the layer above; measured ms/step at c2 (3 x 600 steps, next-batch
// hint on): lds 0.338, f 0.324, fw 0.3186, fdw 0.3166.  (Before the background table pass was capped at 96 VGPRs and the kernels
// here at <= 312, a direct dgrad could not share a SIMD with anything and "fdw" LOST: 0.347 vs 0.329 for "fw".)
bool dr_enabled(char op) {
    static const std::string ops = [] {
        const char* e = getenv("DCTR_GEMM");
        if (e == nullptr) return std::string("fdw");
        if (!strcmp(e, "lds") || !strcmp(e, "LDS")) return std::string();
        return std::string(e);
    }();
    return ops.find(op) != std::string::npos;
}
double dr_threshold() {
    static const double th = [] { const char* e = getenv("DCTR_GEMM_DR_MIN_EFF"); return e ? atof(e) : 0.50; }();
    return th;
}

template <int TM, int TN, bool A_RC, bool B_RC, bool CS, int EPI>
int dr_launch(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, int splits, const DrEpilogue& ep,
              hipStream_t st) {
    auto kern = gemm_dr_kernel<TM, TN, A_RC, B_RC, CS, EPI>;
    constexpr size_t lds = gemm_dr_lds_bytes<TM, TN>();
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    DCTR_HIP_CHECK(attr);
    const int nbm = ceil_div(M, 16 * TM), nbn = ceil_div(N, 16 * TN);
    const int kchunk = (int)round_up(ceil_div(K, splits), 16);
    kern<<<dim3((unsigned)(nbm * nbn), (unsigned)splits), 256, lds, st>>>(A, lda, B, ldb, C, ldc, M, N, K, kchunk, nbn, ep);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// an operand the kernel addresses with 32-bit byte offsets behind a per-block base
inline bool fits31(int64_t rows, int64_t ld) { return rows * ld * 4 < (int64_t)0x7fff0000; }

template <size_t NT>
int pick(const Tile (&list)[NT], int Mo, int No, int64_t R, int S, double* eff) {
    int best = -1;
    *eff = 0.0;
    for (size_t i = 0; i < NT; ++i) {
        const double e = dr_efficiency(Mo, No, R, S, list[i], nullptr);
        if (e > *eff) { *eff = e; best = (int)i; }
    }
    return best;
}

constexpr Tile FWD_TILES[] = {{2, 13}, {4, 7}, {2, 8}};
constexpr Tile DGRAD_TILES[] = {{2, 13}, {4, 10}, {2, 8}};
// (no 3x13 here: 392 VGPRs -- a wgrad runs beside the background table pass, whose two waves per SIMD leave room for 320)
constexpr Tile WGRAD_TILES[] = {{2, 13}, {2, 8}};

}  // namespace

// ---- Y = act(X W + b)  (A = X [M,K] reduction-contiguous, B = W [K,N])
int dr_fc_fwd(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int M, int K, int N, int relu, float keep,
              const uint64_t* seed_ptr, uint64_t seed, hipStream_t st, bool* done) {
    *done = false;
    if (!dr_enabled('f') || M <= 0 || N <= 0 || K < 64) return DCTR_OK;
    if (!al16(x) || !al16(w) || (ldx & 3) || (N & 3) || !fits31(M, ldx) || !fits31(K, N)) return DCTR_OK;
    double eff;
    const int t = pick(FWD_TILES, M, N, K, 1, &eff);
    if (t < 0 || eff < dr_threshold()) return DCTR_OK;
    DrEpilogue ep{};
    ep.bias = b; ep.relu = relu; ep.keep = keep; ep.seed = seed; ep.seed_ptr = seed_ptr;
    *done = true;
    switch (t) {
        case 0: return dr_launch<2, 13, true, false, false, DR_BIAS_ACT>(x, ldx, w, N, y, ldy, M, N, K, 1, ep, st);
        case 1: return dr_launch<4, 7, true, false, false, DR_BIAS_ACT>(x, ldx, w, N, y, ldy, M, N, K, 1, ep, st);
        default: return dr_launch<2, 8, true, false, false, DR_BIAS_ACT>(x, ldx, w, N, y, ldy, M, N, K, 1, ep, st);
    }
}

// ---- dX[M,K] = dY[M,N] W[K,N]^T (x ReLU mask of the producing layer): reduction over N; A = dY [M,N], B^T = W [K,N], both
// reduction-contiguous
int dr_fc_bwd_data(const float* dy, int lddy, const float* w, float* dx, int lddx, int M, int K, int N, const float* act, int ldact,
                   float keep_prev, hipStream_t st, bool* done) {
    *done = false;
    if (!dr_enabled('d') || M <= 0 || K <= 0 || N < 64) return DCTR_OK;
    if (!al16(dy) || !al16(w) || (lddy & 3) || (N & 3) || !fits31(M, lddy) || !fits31(K, N)) return DCTR_OK;
    double eff;
    const int t = pick(DGRAD_TILES, M, K, N, 1, &eff);
    if (t < 0 || eff < dr_threshold()) return DCTR_OK;
    DrEpilogue ep{};
    ep.act = act; ep.ldact = ldact; ep.inv_keep = act ? 1.0f / keep_prev : 1.f;
    *done = true;
    if (act != nullptr) {
        switch (t) {
            case 0: return dr_launch<2, 13, true, true, false, DR_MASK>(dy, lddy, w, N, dx, lddx, M, K, N, 1, ep, st);
            case 1: return dr_launch<4, 10, true, true, false, DR_MASK>(dy, lddy, w, N, dx, lddx, M, K, N, 1, ep, st);
            default: return dr_launch<2, 8, true, true, false, DR_MASK>(dy, lddy, w, N, dx, lddx, M, K, N, 1, ep, st);
        }
    }
    switch (t) {
        case 0: return dr_launch<2, 13, true, true, false, DR_STORE>(dy, lddy, w, N, dx, lddx, M, K, N, 1, ep, st);
        case 1: return dr_launch<4, 10, true, true, false, DR_STORE>(dy, lddy, w, N, dx, lddx, M, K, N, 1, ep, st);
        default: return dr_launch<2, 8, true, true, false, DR_STORE>(dy, lddy, w, N, dx, lddx, M, K, N, 1, ep, st);
    }
}

// the batch split the direct kernel wants for dW[K,N] = X^T dY over M rows: as many splits as keep the grid within one round of
// the chip (0: no tile of the list fits this shape well -> the LDS-tiled kernel and its own split rule)
int dr_wgrad_splits(int M, int K, int N) {
    if (!dr_enabled('w') || M < 256) return 0;
    int best_s = 0;
    double best = 0.0;
    for (const Tile& t : WGRAD_TILES) {
        const int tiles = ceil_div(K, 16 * t.tm) * ceil_div(N, 16 * t.tn);
        int s = CUS / tiles;
        if (s < 1) continue;                                    // (multi-round wgrad: the LDS kernel's territory)
        s = std::min(s, ceil_div(M, 256));                      // >= 4 groups per wave
        const double e = dr_efficiency(K, N, M, s, t, nullptr);
        if (e > best) { best = e; best_s = s; }
    }
    return best >= dr_threshold() ? best_s : 0;
}

// dW partial slabs (split over the batch) + bias-gradient partials; A = X^T stored as X [M,K], B = dY [M,N]: both "NC"
int dr_fc_bwd_weights_partials(const float* x, int ldx, const float* dy, int lddy, float* dw_part, int64_t dw_stride, float* db_part,
                               int64_t db_stride, int M, int K, int N, int splits, hipStream_t st, bool* done) {
    *done = false;
    if (!dr_enabled('w') || M <= 0 || splits < 1 || (int64_t)ceil_div(M, splits) < 64) return DCTR_OK;
    if (!al16(x) || !al16(dy) || (ldx & 3) || (lddy & 3) || (N & 3) || !fits31(M, ldx) || !fits31(M, lddy) || !al16(dw_part) || (dw_stride & 3))
        return DCTR_OK;
    double eff;
    const int t = pick(WGRAD_TILES, K, N, M, splits, &eff);
    if (t < 0 || eff < dr_threshold()) return DCTR_OK;
    int blocks = 0;
    dr_efficiency(K, N, M, splits, WGRAD_TILES[t], &blocks);
    if (blocks > CUS) return DCTR_OK;
    DrEpilogue ep{};
    ep.split_stride = dw_stride;
    ep.colsum = db_part;
    ep.colsum_stride = db_stride;
    *done = true;
    switch (t) {
        case 0: return dr_launch<2, 13, false, false, true, DR_STORE>(x, ldx, dy, lddy, dw_part, N, K, N, M, splits, ep, st);
        default: return dr_launch<2, 8, false, false, true, DR_STORE>(x, ldx, dy, lddy, dw_part, N, K, N, M, splits, ep, st);
    }
}

}  // namespace dctr
