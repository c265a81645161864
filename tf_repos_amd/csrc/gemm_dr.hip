// K6, second kernel family (gemm_dr.h): host side -- instantiations, the per-shape tile chooser and the three layer products.
// gemm.hip's fc_fwd / fc_bwd_data / fc_bwd_weights_partials try these first and fall back to the LDS-tiled kernel when no
// tile of the list below fits the shape well (or an operand is not 16-byte aligned).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>

#include <hip/hip_ext.h>

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
    // one block per CU at a time (launch bounds, LDS): nothing hides a block's prologue and reduction behind its neighbour's MFMAs,
    // so a grid of many rounds is the LDS-tiled kernel's territory (AFM's 3 M-row attention dgrad: 9.6 ms here, 5.8 ms there)
    if (rounds > 4) return 0.0;
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

template <int TM, int TN, bool A_RC, bool B_RC, bool CS, int EPI, int AGEN = DR_AGEN_NONE>
int dr_launch(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, int splits, const DrEpilogue& ep,
              hipStream_t st) {
    auto kern = gemm_dr_kernel<TM, TN, A_RC, B_RC, CS, EPI, AGEN>;
    constexpr size_t lds = gemm_dr_lds_bytes<TM, TN>();
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    DCTR_HIP_CHECK(attr);
    const int nbm = ceil_div(M, 16 * TM), nbn = ceil_div(N, 16 * TN);
    const int kchunk = (int)round_up(ceil_div(K, splits), 16);
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (take_timer_events(&t0, &t1)) {              // (dctr_step_timer mode 2: this dispatch's own start / stop events; a launch has ONE stop event -- the
                                                    //  engine does not arm a fork on a launch it times)
        hipExtLaunchKernelGGL(kern, dim3((unsigned)(nbm * nbn), (unsigned)splits), dim3(256), (uint32_t)lds, st, t0, t1, 0, A, lda, B, ldb, C, ldc, M, N, K,
                              kchunk, nbn, ep, DrOuter{});
    } else if (hipEvent_t stop = take_stop_event()) {      // (the engine's next cross-stream record rides on this launch: common.h)
        hipExtLaunchKernelGGL(kern, dim3((unsigned)(nbm * nbn), (unsigned)splits), dim3(256), (uint32_t)lds, st, nullptr, stop, 0, A, lda, B, ldb, C, ldc, M, N, K,
                              kchunk, nbn, ep, DrOuter{});
    } else {
        kern<<<dim3((unsigned)(nbm * nbn), (unsigned)splits), 256, lds, st>>>(A, lda, B, ldb, C, ldc, M, N, K, kchunk, nbn, ep, DrOuter{});
    }
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

// generated-A products of the Outer-PNN first layer (gemm_dr.h DR_AGEN_*): R = reduction length (a multiple of 64), kchunk a multiple of 64
template <int TN, int AGEN>
int dr_launch_outer(const DrOuter& og, const float* Bm, int ldb, float* C, int ldc, int M, int N, int R, int splits, int64_t split_stride,
                    hipStream_t st) {
    constexpr bool FWD = AGEN == DR_AGEN_OUTER_FWD;
    auto kern = gemm_dr_kernel<2, TN, FWD, false, false, DR_STORE, AGEN>;
    constexpr size_t lds = gemm_dr_lds_bytes<2, TN>();
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    DCTR_HIP_CHECK(attr);
    const int nbm = ceil_div(M, 32), nbn = ceil_div(N, 16 * TN);
    const int kchunk = (int)round_up(ceil_div(R, splits), 64);
    DrEpilogue ep{};
    ep.split_stride = split_stride;
    kern<<<dim3((unsigned)(nbm * nbn), (unsigned)ceil_div(R, kchunk)), 256, lds, st>>>(nullptr, 0, Bm, ldb, C, ldc, M, N, R, kchunk, nbn, ep, og);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// the kernel addresses an operand with 32-bit byte offsets behind a per-WAVE base: what must fit is the wave's own extent -- its
// tile's rows of a reduction-contiguous operand, its quarter of the (split) reduction range of the other kind
inline bool fits31(int64_t rows, int64_t ld) { return rows * ld * 4 < (int64_t)0x7fff0000; }
inline int64_t wave_rows(int64_t R, int S) { return (round_up(ceil_div(R, S), 16) + 15) / 16 * 4 + 16; }

// A/B knob DCTR_DR_TILE="<op><index>[,<op><index>...]" (op in f, d, w): that product takes tile `index` of its list whatever the model says
int forced_tile(char op) {
    static const std::string spec = [] { const char* e = getenv("DCTR_DR_TILE"); return std::string(e ? e : ""); }();
    for (size_t i = 0; i + 1 < spec.size(); ++i)
        if (spec[i] == op && spec[i + 1] >= '0' && spec[i + 1] <= '9') return spec[i + 1] - '0';
    return -1;
}
template <size_t NT>
int pick(const Tile (&list)[NT], int Mo, int No, int64_t R, int S, double* eff, char op = 0) {
    int best = -1;
    *eff = 0.0;
    const int forced = op ? forced_tile(op) : -1;
    if (forced >= 0 && forced < (int)NT) { *eff = 1.0; return forced; }
    for (size_t i = 0; i < NT; ++i) {
        const double e = dr_efficiency(Mo, No, R, S, list[i], nullptr);
        if (e > *eff) { *eff = e; best = (int)i; }
    }
    return best;
}
// An output of fewer than 128 of the LDS-tiled kernel's 64 x 64 tiles leaves more than half of the chip idle there, each block
// walking the whole reduction behind a barrier per 16 k (c1, B = 256: 28 blocks, 16-19 us for 64 MFLOP).  No tile of the direct
// kernel reaches the efficiency threshold on such a shape either -- its fixed cost dominates -- but the fastest of them wins by
// 3x: the best efficiency is the least rounds x cycles.
bool small_problem(int Mo, int No) { return (int64_t)ceil_div(Mo, 64) * ceil_div(No, 64) < 128; }

// (1x4: small batches -- c1's B = 256, serving -- where the output has too few big tiles for the chip: 16 x 64 patches, 112 blocks
//  for 256 x 400, each wave a quarter of the reduction; see small_problem())
// (4x8: a 256-wide layer over a long reduction -- c4's first PNN layer, 8192 x 1989 -> 256 -- in ONE round of 64 x 128 tiles where
//  2x8 takes two and loads every weight column block twice as often)
constexpr Tile FWD_TILES[] = {{2, 13}, {4, 7}, {2, 8}, {1, 4}, {4, 8}};
constexpr Tile DGRAD_TILES[] = {{2, 13}, {4, 10}, {2, 8}, {1, 4}};
// (no 3x13 here: 392 VGPRs -- a wgrad runs beside the background table pass, whose two waves per SIMD leave room for 320)
// (2x16: a 256-wide output in one column block -- AFM's attention weight over 3 M pair rows, 8 row tiles x 32 batch splits)
// (4x8: 64 x 128 per block -- a reduction over millions of rows streams BOTH operands, and every wave loads its whole tile's rows:
//  16 (16 TM + 16 TN) 4 bytes per 16-row group, i.e. 14 flop/byte for 2x16 but 21 for 4x8; at the MFMA rate the 2x16 tile asks the
//  L2s for 9 TB/s)
constexpr Tile WGRAD_TILES[] = {{2, 13}, {2, 16}, {2, 8}, {4, 8}};

}  // namespace

// Which kernel a layer product of this shape takes (host logic only, no launch): "ws" (gemm_ws.hip), "dr TMxTN[ xS]" or "lds[ xS]".  Alignment is
// assumed (16-byte bases, leading dimensions multiples of 4), as the engine's buffers have it.
// a reduction over this many rows streams its operands from HBM (no cache holds them): the 2 x 8 tile, two blocks per CU
inline bool streaming_rows(int64_t M, int K = 32, int N = 128) {
    // measured on AFM's attention weight (3 M pair rows, 256 x 256): 8.5 ms against 6.5 ms for the 2 x 16 tile at one block per CU --
    // the second column block re-reads the 3.1 GB operand; kept as an A/B knob (DCTR_WGRAD_STREAM=1)
    static const bool on = getenv("DCTR_WGRAD_STREAM") != nullptr;
    return on && M >= (1 << 20) && K >= 32 && N >= 128;                         // (whole 32 x 128 tiles)
}
int dr_wgrad_splits(int M, int K, int N);
// a weight gradient over millions of rows whose output is whole 64 x 128 tiles: the 4 x 8 tile (A/B knob DCTR_WGRAD_48=0)
inline bool tall_square(int64_t M, int K, int N) {
    static const bool off = [] { const char* v = getenv("DCTR_WGRAD_48"); return v != nullptr && v[0] == '0'; }();
    return !off && M >= (1 << 20) && K % 64 == 0 && N % 128 == 0;
}
int gemm_plan(char op, int M, int K, int N, char* out, int out_len) {
    double eff = 0.0;
    int t = -1, s = 1;
    const Tile* tile = nullptr;
    if (op == 'f') {
        if (dr_enabled('f') && M > 0 && N > 0 && K >= 64 && (N & 3) == 0) {
            t = pick(FWD_TILES, M, N, K, 1, &eff, 'f');
            if (t >= 0 && (eff >= dr_threshold() || small_problem(M, N))) tile = &FWD_TILES[t];
        }
    } else if (op == 'd') {
        if (dr_enabled('d') && M > 0 && K > 0 && N >= 64 && (N & 3) == 0) {
            t = pick(DGRAD_TILES, M, K, N, 1, &eff, 'd');
            if (t >= 0 && (eff >= dr_threshold() || small_problem(M, K))) tile = &DGRAD_TILES[t];
        }
    } else if (op == 'w') {
        s = choose_wgrad_splits(M, K, N);
        if (dr_wgrad_splits(M, K, N) == s && s > 0 && (int64_t)ceil_div(M, s) >= 64 && (N & 3) == 0 && streaming_rows(M, K, N) &&
            (int64_t)ceil_div(K, 32) * ceil_div(N, 128) * s <= 2 * CUS) {
            tile = &WGRAD_TILES[2];           // millions of rows from HBM: 2 x 8, two blocks per CU
        } else if (dr_wgrad_splits(M, K, N) == s && s > 0 && (int64_t)ceil_div(M, s) >= 64 && tall_square(M, K, N) && (int64_t)(K / 64) * (N / 128) * s <= CUS) {
            tile = &WGRAD_TILES[3];
        } else if (dr_wgrad_splits(M, K, N) == s && s > 0 && (int64_t)ceil_div(M, s) >= 64 && (N & 3) == 0) {
            t = pick(WGRAD_TILES, K, N, M, s, &eff);
            int blocks = 0;
            if (t >= 0) dr_efficiency(K, N, M, s, WGRAD_TILES[t], &blocks);
            if (t >= 0 && eff >= dr_threshold() && blocks <= CUS) tile = &WGRAD_TILES[t];
        }
    } else {
        set_error("gemm_plan: op must be 'f', 'd' or 'w'");
        return DCTR_ERR_INVALID_ARG;
    }
    if ((op == 'f' && ws_takes(M, K, N)) || (op == 'd' && ws_takes(M, N, K))) { snprintf(out, (size_t)out_len, "ws"); return DCTR_OK; }
    if (tile != nullptr) snprintf(out, (size_t)out_len, op == 'w' ? "dr %dx%d x%d" : "dr %dx%d", tile->tm, tile->tn, s);
    else snprintf(out, (size_t)out_len, op == 'w' ? "lds x%d" : "lds", s);
    return DCTR_OK;
}

// ---- Y = act(X W + b)  (A = X [M,K] reduction-contiguous, B = W [K,N])
int dr_fc_fwd(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int M, int K, int N, int relu, float keep,
              const uint64_t* seed_ptr, uint64_t seed, hipStream_t st, bool* done) {
    *done = false;
    if (!dr_enabled('f') || M <= 0 || N <= 0 || K < 64) return DCTR_OK;
    if (!al16(x) || !al16(w) || (ldx & 3) || (N & 3) || !fits31(64, ldx) || !fits31(wave_rows(K, 1), N)) return DCTR_OK;
    double eff;
    const int t = pick(FWD_TILES, M, N, K, 1, &eff, 'f');
    if (t < 0 || (eff < dr_threshold() && !small_problem(M, N))) return DCTR_OK;
    DrEpilogue ep{};
    ep.bias = b; ep.relu = relu; ep.keep = keep; ep.seed = seed; ep.seed_ptr = seed_ptr;
    *done = true;
    switch (t) {
        case 0: return dr_launch<2, 13, true, false, false, DR_BIAS_ACT>(x, ldx, w, N, y, ldy, M, N, K, 1, ep, st);
        case 1: return dr_launch<4, 7, true, false, false, DR_BIAS_ACT>(x, ldx, w, N, y, ldy, M, N, K, 1, ep, st);
        case 2: return dr_launch<2, 8, true, false, false, DR_BIAS_ACT>(x, ldx, w, N, y, ldy, M, N, K, 1, ep, st);
        case 4: return dr_launch<4, 8, true, false, false, DR_BIAS_ACT>(x, ldx, w, N, y, ldy, M, N, K, 1, ep, st);
        default: return dr_launch<1, 4, true, false, false, DR_BIAS_ACT>(x, ldx, w, N, y, ldy, M, N, K, 1, ep, st);
    }
}

// ---- dX[M,K] = dY[M,N] W[K,N]^T (x ReLU mask of the producing layer): reduction over N; A = dY [M,N], B^T = W [K,N], both
// reduction-contiguous
int dr_fc_bwd_data(const float* dy, int lddy, const float* w, float* dx, int lddx, int M, int K, int N, const float* act, int ldact,
                   float keep_prev, hipStream_t st, bool* done) {
    *done = false;
    if (!dr_enabled('d') || M <= 0 || K <= 0 || N < 64) return DCTR_OK;
    if (!al16(dy) || !al16(w) || (lddy & 3) || (N & 3) || !fits31(64, lddy) || !fits31(256, N)) return DCTR_OK;
    double eff;
    const int t = pick(DGRAD_TILES, M, K, N, 1, &eff, 'd');
    if (t < 0 || (eff < dr_threshold() && !small_problem(M, K))) return DCTR_OK;
    DrEpilogue ep{};
    ep.act = act; ep.ldact = ldact; ep.inv_keep = act ? 1.0f / keep_prev : 1.f;
    *done = true;
    if (act != nullptr) {
        switch (t) {
            case 0: return dr_launch<2, 13, true, true, false, DR_MASK>(dy, lddy, w, N, dx, lddx, M, K, N, 1, ep, st);
            case 1: return dr_launch<4, 10, true, true, false, DR_MASK>(dy, lddy, w, N, dx, lddx, M, K, N, 1, ep, st);
            case 2: return dr_launch<2, 8, true, true, false, DR_MASK>(dy, lddy, w, N, dx, lddx, M, K, N, 1, ep, st);
            default: return dr_launch<1, 4, true, true, false, DR_MASK>(dy, lddy, w, N, dx, lddx, M, K, N, 1, ep, st);
        }
    }
    switch (t) {
        case 0: return dr_launch<2, 13, true, true, false, DR_STORE>(dy, lddy, w, N, dx, lddx, M, K, N, 1, ep, st);
        case 1: return dr_launch<4, 10, true, true, false, DR_STORE>(dy, lddy, w, N, dx, lddx, M, K, N, 1, ep, st);
        case 2: return dr_launch<2, 8, true, true, false, DR_STORE>(dy, lddy, w, N, dx, lddx, M, K, N, 1, ep, st);
        default: return dr_launch<1, 4, true, true, false, DR_STORE>(dy, lddy, w, N, dx, lddx, M, K, N, 1, ep, st);
    }
}

// the batch split the direct kernel wants for dW[K,N] = X^T dY over M rows: as many splits as keep the grid within one round of
// the chip (0: no tile of the list fits this shape well -> the LDS-tiled kernel and its own split rule)
int dr_wgrad_splits(int M, int K, int N) {
    if (!dr_enabled('w') || M < 256) return 0;
    if (streaming_rows(M, K, N)) {
        const int tiles = ceil_div(K, 32) * ceil_div(N, 128);
        if (tiles <= 2 * CUS) return std::max(1, 2 * CUS / tiles);
    }
    if (tall_square(M, K, N) && (K / 64) * (N / 128) <= CUS) return std::max(1, CUS / ((K / 64) * (N / 128)));
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

// which weight-gradient tile (index into WGRAD_TILES) a shape takes with this many splits, -1: not the direct kernel's
static int wgrad_tile(const float* x, int ldx, const float* dy, int lddy, const float* dw_part, int64_t dw_stride, int M, int K, int N, int splits) {
    if (!dr_enabled('w') || M <= 0 || splits < 1 || (int64_t)ceil_div(M, splits) < 64) return -1;
    if (!al16(x) || !al16(dy) || (ldx & 3) || (lddy & 3) || (N & 3) || !fits31(wave_rows(M, splits), ldx) || !fits31(wave_rows(M, splits), lddy) ||
        !al16(dw_part) || (dw_stride & 3))
        return -1;
    double eff;
    int t = pick(WGRAD_TILES, K, N, M, splits, &eff);
    const bool stream = streaming_rows(M, K, N) && (int64_t)ceil_div(K, 32) * ceil_div(N, 128) * splits <= 2 * CUS;
    if (stream) t = 2;
    else if (tall_square(M, K, N) && (int64_t)(K / 64) * (N / 128) * splits <= CUS) t = 3;
    else {
        if (t < 0 || eff < dr_threshold()) return -1;
        int blocks = 0;
        dr_efficiency(K, N, M, splits, WGRAD_TILES[t], &blocks);
        if (blocks > CUS) return -1;
    }
    return t;
}

// dW partial slabs (split over the batch) + bias-gradient partials; A = X^T stored as X [M,K], B = dY [M,N]: both "NC"
int dr_fc_bwd_weights_partials(const float* x, int ldx, const float* dy, int lddy, float* dw_part, int64_t dw_stride, float* db_part,
                               int64_t db_stride, int M, int K, int N, int splits, hipStream_t st, bool* done) {
    *done = false;
    const int t = wgrad_tile(x, ldx, dy, lddy, dw_part, dw_stride, M, K, N, splits);
    if (t < 0) return DCTR_OK;
    DrEpilogue ep{};
    ep.split_stride = dw_stride;
    ep.colsum = db_part;
    ep.colsum_stride = db_stride;
    *done = true;
    switch (t) {
        case 0: return dr_launch<2, 13, false, false, true, DR_STORE>(x, ldx, dy, lddy, dw_part, N, K, N, M, splits, ep, st);
        case 1: return dr_launch<2, 16, false, false, true, DR_STORE>(x, ldx, dy, lddy, dw_part, N, K, N, M, splits, ep, st);
        case 2: return dr_launch<2, 8, false, false, true, DR_STORE>(x, ldx, dy, lddy, dw_part, N, K, N, M, splits, ep, st);
        default: return dr_launch<4, 8, false, false, true, DR_STORE>(x, ldx, dy, lddy, dw_part, N, K, N, M, splits, ep, st);
    }
}

// The same for a layer whose output gradient is rank one under its ReLU mask, dY = rowscale (x) colscale . 1[H > 0] (H [M,N] the
// layer's stored output), formed on the B-operand loads (gemm_dr.h DR_BGATE_WGRAD) instead of read from memory; dwo_part takes the
// second column sums, sum_row rowscale[row] H[row, n] -- the weight gradient of the (N -> 1) layer that produced the rank-one form.
int dr_fc_bwd_weights_partials_gate(const float* x, int ldx, const float* h, int ldh, const float* rowscale, const float* colscale,
                                    float* dw_part, int64_t dw_stride, float* db_part, int64_t db_stride, float* dwo_part, int64_t dwo_stride,
                                    int M, int K, int N, int splits, hipStream_t st, bool* done) {
    *done = false;
    const int t = wgrad_tile(x, ldx, h, ldh, dw_part, dw_stride, M, K, N, splits);
    if (t < 0 || !al16(rowscale)) return DCTR_OK;
    DrEpilogue ep{};
    ep.split_stride = dw_stride;
    ep.colsum = db_part;
    ep.colsum_stride = db_stride;
    ep.rowscale = rowscale;
    ep.colscale = colscale;
    ep.colsum2 = dwo_part;
    ep.colsum2_stride = dwo_stride;
    *done = true;
    switch (t) {
        case 0: return dr_launch<2, 13, false, false, true, DR_STORE, DR_BGATE_WGRAD>(x, ldx, h, ldh, dw_part, N, K, N, M, splits, ep, st);
        case 1: return dr_launch<2, 16, false, false, true, DR_STORE, DR_BGATE_WGRAD>(x, ldx, h, ldh, dw_part, N, K, N, M, splits, ep, st);
        case 2: return dr_launch<2, 8, false, false, true, DR_STORE, DR_BGATE_WGRAD>(x, ldx, h, ldh, dw_part, N, K, N, M, splits, ep, st);
        default: return dr_launch<4, 8, false, false, true, DR_STORE, DR_BGATE_WGRAD>(x, ldx, h, ldh, dw_part, N, K, N, M, splits, ep, st);
    }
}

// ---- Outer-PNN without the [B, P K K] product tensor (PNN.py:139-153 'Outer' + the first fully_connected, PNN.py:159-166) ---------------
// The first MLP layer's weight is [F K + P K K, H]: rows [0, F K) meet the flat embeddings (an ordinary layer product, done by the
// caller), rows F K + (p K + a) K + c meet e_i[a] e_j[c] of pair p = (i, j).  Here those products are formed in the MFMA A fragments
// from the gathered embeddings (41 MB at B = 8192, L2 / Infinity-Cache resident) instead of being written and re-read (24.9 GB each
// way at the run.sh operating point K = 32).
bool opnn_fused_ok(int F, int K, int H) {
    static const bool off = getenv("DCTR_OPNN_MATERIALISE") != nullptr;         // A/B knob: the materialising path
    if (off || K < 16 || (K & (K - 1)) != 0 || F < 2 || F > 0x7fff || (H & 3) != 0 || H < 4) return false;
    const int64_t L = (int64_t)F * (F - 1) / 2 * K * K;
    return fits31(L, H);
}

int opnn_fwd_splits(int B, int H) {
    const int blocks = ceil_div(B, 32) * ceil_div(H, 256);
    return std::max(1, CUS / blocks);
}

__global__ __launch_bounds__(256) void opnn_finish_kernel(float* __restrict__ y, int ldy, const float* __restrict__ ws, int64_t ws_stride, int splits,
                                                          const float* __restrict__ bias, int B, int H, int relu, float keep, uint64_t seed,
                                                          const uint64_t* __restrict__ seed_ptr) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;        // one float4 of the [B, H] output
    const int h4 = H / 4;
    if (idx >= (int64_t)B * h4) return;
    const int b = (int)(idx / h4), col = 4 * (int)(idx - (int64_t)b * h4);
    float4 v = *reinterpret_cast<const float4*>(y + (size_t)b * ldy + col);
    for (int s = 0; s < splits; ++s) {
        const float4 p = *reinterpret_cast<const float4*>(ws + (size_t)s * ws_stride + (size_t)b * H + col);
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    float o[4] = {v.x, v.y, v.z, v.w};
    const uint64_t sd = seed ^ (seed_ptr ? *seed_ptr : 0ull), row0 = dropout_row0(seed_ptr);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (bias != nullptr) o[e] += bias[col + e];
        if (relu) o[e] = fmaxf(o[e], 0.f);
        if (keep < 1.0f) o[e] *= dropout_scale(sd, (row0 + (uint64_t)b) * (uint64_t)H + col + e, keep);
    }
    *reinterpret_cast<float4*>(y + (size_t)b * ldy + col) = make_float4(o[0], o[1], o[2], o[3]);
}

// y[B, H] holds the flat part's product on entry; on return act(y + outer part + bias).  ws: opnn_fwd_ws_floats_max() floats.
// (splits * ceil(B / 32) <= 256 whenever splits > 1, so no batch needs more than max(8192 + 32 splits, B) slab rows)
int64_t opnn_fwd_ws_floats_max(int max_batch, int H) { return ((int64_t)std::max(max_batch, 8192) + 32 * CUS) * H; }

int opnn_outer_fwd(const float* e, int e_ld, int B, int F, int K, const int* pairs, const float* w_outer, const float* bias, float* y, int ldy,
                   int H, int relu, float keep, const uint64_t* seed_ptr, uint64_t seed, float* ws, hipStream_t st) {
    if (B <= 0) return DCTR_OK;
    DCTR_REQUIRE(opnn_fused_ok(F, K, H) && al16(e) && al16(w_outer) && al16(y) && al16(ws) && (e_ld & 3) == 0 && (ldy & 3) == 0 && fits31(B, e_ld),
                 "opnn_outer_fwd: unsupported shape / alignment");
    DrOuter og{e, e_ld, B, pairs, 31 - __builtin_clz((unsigned)K)};
    const int L = F * (F - 1) / 2 * K * K, S = opnn_fwd_splits(B, H);
    const int64_t stride = (int64_t)B * H;
    if (H > 208) DCTR_TRY((dr_launch_outer<16, DR_AGEN_OUTER_FWD>(og, w_outer, H, ws, H, B, H, L, S, stride, st)));
    else if (H > 128) DCTR_TRY((dr_launch_outer<13, DR_AGEN_OUTER_FWD>(og, w_outer, H, ws, H, B, H, L, S, stride, st)));
    else DCTR_TRY((dr_launch_outer<8, DR_AGEN_OUTER_FWD>(og, w_outer, H, ws, H, B, H, L, S, stride, st)));
    const int splits = ceil_div(L, (int)round_up(ceil_div(L, S), 64));
    opnn_finish_kernel<<<(unsigned)ceil_div((int64_t)B * (H / 4), 256), 256, 0, st>>>(y, ldy, ws, stride, splits, bias, B, H, relu, keep, seed, seed_ptr);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

// dW_outer[(p,a,c)][h] = sum_b e_i[b][a] e_j[b][c] dY[b][h]   (one slab: P K K / 32 row blocks are far more than the chip has CUs)
int opnn_outer_wgrad(const float* e, int e_ld, int B, int F, int K, const int* pairs, const float* dy, int lddy, int H, float* dw_outer, hipStream_t st) {
    DCTR_REQUIRE(opnn_fused_ok(F, K, H) && al16(e) && al16(dy) && al16(dw_outer) && (e_ld & 3) == 0 && (lddy & 3) == 0 && fits31(B, e_ld) && fits31(B, lddy),
                 "opnn_outer_wgrad: unsupported shape / alignment");
    const int L = F * (F - 1) / 2 * K * K;
    if (B <= 0) { DCTR_HIP_CHECK(hipMemsetAsync(dw_outer, 0, (size_t)L * H * sizeof(float), st)); return DCTR_OK; }
    DrOuter og{e, e_ld, B, pairs, 31 - __builtin_clz((unsigned)K)};
    const int R = (int)round_up(B, 64);
    if (H > 208) return dr_launch_outer<16, DR_AGEN_OUTER_WGRAD>(og, dy, lddy, dw_outer, H, L, H, R, 1, 0, st);
    if (H > 128) return dr_launch_outer<13, DR_AGEN_OUTER_WGRAD>(og, dy, lddy, dw_outer, H, L, H, R, 1, 0, st);
    return dr_launch_outer<8, DR_AGEN_OUTER_WGRAD>(og, dy, lddy, dw_outer, H, L, H, R, 1, 0, st);
}

}  // namespace dctr

extern "C" int dctr_gemm_plan(char op, int M, int K, int N, char* out, int out_len) {
    DCTR_REQUIRE(out != nullptr && out_len >= 24, "dctr_gemm_plan: output buffer of >= 24 bytes");
    return dctr::gemm_plan(op, M, K, N, out, out_len);
}
