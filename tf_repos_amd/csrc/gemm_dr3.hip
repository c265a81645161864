// K6, split-precision mode of the direct-to-register GEMM family (gemm_dr.h: gemm_dr3_kernel, dr_wsplit_kernel): host side.
// dctr_config.gemm_mode = 1.  The three layer products of contrib.layers.fully_connected (DeepFM.py:156-158,165-166,213) with every
// f32 operand element carried as three bf16 planes (24 significand bits) and the six leading plane products accumulated in f32 --
// f32-equivalent results (max error against an fp64 product measured at or below the exact f32-MFMA kernel's, tools/gemm_dr_probe)
// at 6/16 of the f32 MFMA's matrix-pipe time.  The layer's WEIGHT arrives pre-split (the planes are rewritten behind its optimizer
// step: dr3_wsplit), activations are split in registers.  A shape none of the tiles below takes falls through to the exact kernels.
//
// This file is compiled with -fno-slp-vectorize (build.py): hipcc's SLP pass pairs the split's subtractions into v_pk_add_f32, which
// costs more issue time beside MFMAs than two v_sub_f32 (MI355X_MICROARCH.md, "packed f32 VALU ... an anti-lever beside MFMAs").
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <string>

#include <hip/hip_ext.h>

#include "common.h"
#include "gemm_dr.h"
#include "ops.h"

namespace dctr {

__device__ __forceinline__ float dr_dropout_scale(uint64_t seed, uint64_t idx, float keep) { return dropout_scale(seed, idx, keep); }

std::atomic<int64_t> g_dr3_launches{0};

namespace {

constexpr int CUS = 256;

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool fits31(int64_t rows, int64_t ld) { return rows * ld * 4 < (int64_t)0x7fff0000; }

template <int TM, int TN, bool A_RC, bool B_RC, bool CS, int EPI, bool B_PRE>
int dr3_launch(const float* A, int lda, const float* B, int ldb, int64_t bplane, float* C, int ldc, int M, int N, int K, int splits, const DrEpilogue& ep,
               hipStream_t st) {
    auto kern = gemm_dr3_kernel<TM, TN, A_RC, B_RC, CS, EPI, B_PRE>;
    constexpr size_t lds = gemm_dr_lds_bytes<TM, TN>();
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    DCTR_HIP_CHECK(attr);
    const int nbm = ceil_div(M, 16 * TM), nbn = ceil_div(N, 16 * TN);
    const int kchunk = (int)round_up(ceil_div(K, splits), 32);
    const dim3 grid((unsigned)(nbm * nbn), (unsigned)splits);
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (take_timer_events(&t0, &t1)) {              // (dctr_step_timer mode 2: this dispatch's own start / stop events)
        hipExtLaunchKernelGGL(kern, grid, dim3(256), (uint32_t)lds, st, t0, t1, 0, A, lda, B, ldb, C, ldc, M, N, K, kchunk, nbn, ep, bplane, (int64_t)0);
    } else if (hipEvent_t stop = take_stop_event()) {      // (the engine's next cross-stream record rides on this launch: common.h)
        hipExtLaunchKernelGGL(kern, grid, dim3(256), (uint32_t)lds, st, nullptr, stop, 0, A, lda, B, ldb, C, ldc, M, N, K, kchunk, nbn, ep, bplane, (int64_t)0);
    } else {
        kern<<<grid, 256, lds, st>>>(A, lda, B, ldb, C, ldc, M, N, K, kchunk, nbn, ep, bplane, (int64_t)0);
    }
    DCTR_LAUNCH_CHECK();
    g_dr3_launches.fetch_add(1, std::memory_order_relaxed);
    return DCTR_OK;
}

// One round of the chip or nothing: the tile (index) whose grid fills the most of <= 256 blocks with the least padded work.
struct Tile3 { int tm, tn; };
constexpr Tile3 FWD3[] = {{4, 7}, {4, 8}, {4, 10}};
// The forward and dgrad products of a 4096 x 400 output take the 2 x 7 tile at TWO blocks per CU (512 blocks) rather than 4 x 7 at one:
// alone the two are equal (18.0 us, tools/gemm_dr_probe), in the step the small tile wins -- a CU holds blocks of two products side by
// side (a dgrad's beside the weight gradient's of the layer above), one block's prologue / cross-wave reduction / stores under the
// other's MFMAs, and co-resident blocks of one product share their weight planes in L1 (same column block, XCD-aware order): c2
// 0.2271 -> 0.2172 ms/step with "fd", 0.2185 with "fdw" (the weight gradient splits BOTH operands in registers: 4.7 VALU ops per MFMA at
// 2 x 7 against 2.9 at 4 x 7), 0.2206 "f", 0.2188 "d", 0.2316 "w" (profiles/r05_ab_small_tiles.txt).  A/B knob DCTR_DR3_SMALL=<subset of fdw> | none.
bool small_tile(char op, int Mo, int No, int splits) {
    static const std::string ops = [] { const char* e = getenv("DCTR_DR3_SMALL"); return std::string(e ? e : "fd"); }();
    return ops != "none" && ops.find(op) != std::string::npos && (int64_t)ceil_div(Mo, 32) * ceil_div(No, 112) * splits <= 2 * CUS;
}
template <size_t NT>
int pick3(const Tile3 (&list)[NT], int Mo, int No, int splits) {
    int best = -1;
    double best_cost = 0.0;
    for (size_t i = 0; i < NT; ++i) {
        const int64_t blocks = (int64_t)ceil_div(Mo, 16 * list[i].tm) * ceil_div(No, 16 * list[i].tn) * splits;
        if (blocks > CUS) continue;
        const double cost = (double)list[i].tm * list[i].tn;            // MFMAs per block and group: the blocks run side by side
        if (best < 0 || cost < best_cost) { best = (int)i; best_cost = cost; }
    }
    return best;
}

}  // namespace

// (reduction lengths up to 16 384: what the error measurements against fp64 cover -- layer widths; a reduction over hundreds of
//  thousands of terms, like the materialised Outer-PNN first layer's 760 032, stays on the exact kernels, and so do its 1.2 GB of planes)
// (and products of at least 0.8 GFLOP: below that the kernels' fixed costs dominate either way and the re-split launch behind the
//  optimizer is a net loss -- c4's 8192 x 256 x 128 layer: NFM 0.226 -> 0.236 ms/step with it, profiles/r05_configs.txt)
bool dr3_shape_ok(int M, int K, int N) {
    return M >= 1024 && K >= 64 && N >= 64 && K <= 16384 && N <= 16384 && (K & 7) == 0 && (N & 7) == 0 && (int64_t)M * K * N >= (int64_t)400 * 1000 * 1000;
}

// bytes of ONE plane of each pre-split form of a [K][N] weight (three planes each)
int64_t dr3_fwd_plane_bytes(int K, int N) { return (int64_t)ceil_div(K, 8) * N * 16; }
int64_t dr3_dgr_plane_bytes(int K, int N) { return (int64_t)ceil_div(N, 8) * K * 16; }

// up to 8 weights in one launch
int dr3_wsplit_multi(const WsplitJob* jobs, int n, hipStream_t st) {
    DrWsplitJobs J{};
    DCTR_REQUIRE(n >= 0 && n <= 8, "dr3_wsplit_multi: at most 8 weights per launch");
    if (n == 0) return DCTR_OK;
    int64_t total = 0;
    for (int i = 0; i < n; ++i) {
        J.j[i] = DrWsplitJob{jobs[i].w, jobs[i].ldw, jobs[i].K, jobs[i].N, jobs[i].fwd, jobs[i].dgr, total};
        total += (int64_t)ceil_div(jobs[i].K, 8) * jobs[i].N + (int64_t)ceil_div(jobs[i].N, 8) * jobs[i].K;
    }
    J.n = n;
    J.total = total;
    dr_wsplit_kernel<0><<<(unsigned)ceil_div(total, 256), 256, 0, st>>>(J);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}
int dr3_wsplit(const float* w, int ldw, int K, int N, unsigned* fwd, unsigned* dgr, hipStream_t st) {
    const WsplitJob j{w, ldw, K, N, fwd, dgr};
    return dr3_wsplit_multi(&j, 1, st);
}

// ---- Y = act(X W + b): A = X [M,K] split in registers, B = the k-blocked planes of W [K,N]
int dr3_fc_fwd(const float* x, int ldx, const unsigned* wp, int64_t plane, const float* b, float* y, int ldy, int M, int K, int N, int relu, float keep,
               const uint64_t* seed_ptr, uint64_t seed, hipStream_t st, bool* done) {
    *done = false;
    if (wp == nullptr || !dr3_shape_ok(M, K, N) || !al16(x) || (ldx & 3) || !fits31(64, ldx) || !fits31(ceil_div(K, 8) + 8, (int64_t)N * 4)) return DCTR_OK;
    const int t = pick3(FWD3, M, N, 1);
    if (t < 0) return DCTR_OK;
    DrEpilogue ep{};
    ep.bias = b; ep.relu = relu; ep.keep = keep; ep.seed = seed; ep.seed_ptr = seed_ptr;
    *done = true;
    const float* B = reinterpret_cast<const float*>(wp);
    if (small_tile('f', M, N, 1)) return dr3_launch<2, 7, true, true, false, DR_BIAS_ACT, true>(x, ldx, B, N, plane, y, ldy, M, N, K, 1, ep, st);
    switch (t) {
        case 0: return dr3_launch<4, 7, true, true, false, DR_BIAS_ACT, true>(x, ldx, B, N, plane, y, ldy, M, N, K, 1, ep, st);
        case 1: return dr3_launch<4, 8, true, true, false, DR_BIAS_ACT, true>(x, ldx, B, N, plane, y, ldy, M, N, K, 1, ep, st);
        default: return dr3_launch<4, 10, true, true, false, DR_BIAS_ACT, true>(x, ldx, B, N, plane, y, ldy, M, N, K, 1, ep, st);
    }
}

// ---- dX[M,K] = dY[M,N] W[K,N]^T (x ReLU mask): reduction over N; A = dY split in registers, B = the n-blocked planes of W
int dr3_fc_bwd_data(const float* dy, int lddy, const unsigned* wp, int64_t plane, float* dx, int lddx, int M, int K, int N, const float* act, int ldact,
                    float keep_prev, hipStream_t st, bool* done) {
    *done = false;
    if (wp == nullptr || !dr3_shape_ok(M, N, K) || !al16(dy) || (lddy & 3) || !fits31(64, lddy) || !fits31(ceil_div(N, 8) + 8, (int64_t)K * 4)) return DCTR_OK;
    const int t = pick3(FWD3, M, K, 1);
    if (t < 0) return DCTR_OK;
    DrEpilogue ep{};
    ep.act = act; ep.ldact = ldact; ep.inv_keep = act ? 1.0f / keep_prev : 1.f;
    *done = true;
    const float* B = reinterpret_cast<const float*>(wp);
    if (act != nullptr && small_tile('d', M, K, 1)) return dr3_launch<2, 7, true, true, false, DR_MASK, true>(dy, lddy, B, K, plane, dx, lddx, M, K, N, 1, ep, st);
    if (act != nullptr) {
        switch (t) {
            case 0: return dr3_launch<4, 7, true, true, false, DR_MASK, true>(dy, lddy, B, K, plane, dx, lddx, M, K, N, 1, ep, st);
            case 1: return dr3_launch<4, 8, true, true, false, DR_MASK, true>(dy, lddy, B, K, plane, dx, lddx, M, K, N, 1, ep, st);
            default: return dr3_launch<4, 10, true, true, false, DR_MASK, true>(dy, lddy, B, K, plane, dx, lddx, M, K, N, 1, ep, st);
        }
    }
    switch (t) {
        case 0: return dr3_launch<4, 7, true, true, false, DR_STORE, true>(dy, lddy, B, K, plane, dx, lddx, M, K, N, 1, ep, st);
        case 1: return dr3_launch<4, 8, true, true, false, DR_STORE, true>(dy, lddy, B, K, plane, dx, lddx, M, K, N, 1, ep, st);
        default: return dr3_launch<4, 10, true, true, false, DR_STORE, true>(dy, lddy, B, K, plane, dx, lddx, M, K, N, 1, ep, st);
    }
}

// ---- dW partial slabs (split over the batch) + bias-gradient partials: both operands are activations, both split in registers
int dr3_fc_bwd_weights_partials(const float* x, int ldx, const float* dy, int lddy, float* dw_part, int64_t dw_stride, float* db_part,
                                int64_t db_stride, int M, int K, int N, int splits, hipStream_t st, bool* done, int low_prio) {
    *done = false;
    if (splits < 1 || M < 1024 || K < 64 || N < 64 || (N & 3) || (K & 3) || !al16(x) || !al16(dy) || (ldx & 3) || (lddy & 3) || !al16(dw_part) ||
        (dw_stride & 3) || (int64_t)ceil_div(M, splits) < 128 || (int64_t)ceil_div(M, splits) > 65536 || K > 16384 || N > 16384 ||
        (int64_t)M * K * N < (int64_t)400 * 1000 * 1000)
        return DCTR_OK;
    const int64_t rows_w = round_up(ceil_div(M, splits), 32) / 4 + 64;            // a wave's rows (+ the groups it prefetches beyond its range)
    if (!fits31(rows_w, ldx) || !fits31(rows_w, lddy)) return DCTR_OK;
    if ((int64_t)ceil_div(K, 64) * ceil_div(N, 112) * splits > CUS) return DCTR_OK;
    DrEpilogue ep{};
    ep.split_stride = dw_stride;
    ep.colsum = db_part;
    ep.colsum_stride = db_stride;
    ep.low_prio = low_prio;
    *done = true;
    if (small_tile('w', K, N, splits)) return dr3_launch<2, 7, false, false, true, DR_STORE, false>(x, ldx, dy, lddy, 0, dw_part, N, K, N, M, splits, ep, st);
    return dr3_launch<4, 7, false, false, true, DR_STORE, false>(x, ldx, dy, lddy, 0, dw_part, N, K, N, M, splits, ep, st);
}

}  // namespace dctr

using namespace dctr;

extern "C" {

int dctr_gemm_split_plane_bytes(int K, int N, int64_t* fwd_bytes, int64_t* dgr_bytes) {
    DCTR_REQUIRE(K > 0 && N > 0 && fwd_bytes != nullptr && dgr_bytes != nullptr, "dctr_gemm_split_plane_bytes: K, N > 0 and two outputs");
    *fwd_bytes = 3 * dr3_fwd_plane_bytes(K, N);
    *dgr_bytes = 3 * dr3_dgr_plane_bytes(K, N);
    return DCTR_OK;
}

int dctr_gemm_wsplit(const float* d_w, int K, int N, void* d_fwd_planes, void* d_dgr_planes, void* stream) {
    DCTR_REQUIRE(d_w != nullptr && d_fwd_planes != nullptr && d_dgr_planes != nullptr && K > 0 && N > 0, "dctr_gemm_wsplit: null argument / empty weight");
    return dr3_wsplit(d_w, N, K, N, static_cast<unsigned*>(d_fwd_planes), static_cast<unsigned*>(d_dgr_planes), as_stream(stream));
}

int dctr_fc_fwd_split(const float* d_x, int ldx, const void* d_fwd_planes, const float* d_b, float* d_y, int ldy, int M, int K, int N, int relu, float keep,
                      uint64_t seed, void* stream) {
    DCTR_REQUIRE(keep > 0.f && keep <= 1.f, "keep_prob must be in (0,1], got %f", keep);
    bool done = false;
    DCTR_TRY(dr3_fc_fwd(d_x, ldx, static_cast<const unsigned*>(d_fwd_planes), dr3_fwd_plane_bytes(K, N), d_b, d_y, ldy, M, K, N, relu, keep, nullptr, seed,
                        as_stream(stream), &done));
    if (!done) { set_error("dctr_fc_fwd_split: no split-precision kernel takes M=%d K=%d N=%d (use dctr_fc_fwd)", M, K, N); return DCTR_ERR_UNSUPPORTED; }
    return DCTR_OK;
}

int dctr_fc_bwd_data_split(const float* d_dy, int lddy, const void* d_dgr_planes, float* d_dx, int lddx, int M, int K, int N, const float* d_act, int ldact,
                           float keep_prev, void* stream) {
    bool done = false;
    DCTR_TRY(dr3_fc_bwd_data(d_dy, lddy, static_cast<const unsigned*>(d_dgr_planes), dr3_dgr_plane_bytes(K, N), d_dx, lddx, M, K, N, d_act, ldact, keep_prev,
                             as_stream(stream), &done));
    if (!done) { set_error("dctr_fc_bwd_data_split: no split-precision kernel takes M=%d K=%d N=%d (use dctr_fc_bwd_data)", M, K, N); return DCTR_ERR_UNSUPPORTED; }
    return DCTR_OK;
}

int dctr_fc_bwd_weights_split(const float* d_x, int ldx, const float* d_dy, int lddy, float* d_dw, float* d_db, int M, int K, int N, float* d_workspace,
                              size_t workspace_bytes, void* stream) {
    hipStream_t st = as_stream(stream);
    int splits = choose_wgrad_splits(M, K, N);
    const size_t per = ((size_t)K * N + N) * sizeof(float);
    DCTR_REQUIRE(d_workspace != nullptr && workspace_bytes >= per, "dctr_fc_bwd_weights_split: workspace of at least (K N + N) floats");
    if ((size_t)splits * per > workspace_bytes) splits = (int)(workspace_bytes / per);
    float* wpart = d_workspace;
    float* bpart = d_workspace + (size_t)splits * K * N;
    bool done = false;
    DCTR_TRY(dr3_fc_bwd_weights_partials(d_x, ldx, d_dy, lddy, wpart, (int64_t)K * N, d_db ? bpart : nullptr, N, M, K, N, splits, st, &done));
    if (!done) { set_error("dctr_fc_bwd_weights_split: no split-precision kernel takes M=%d K=%d N=%d (use dctr_fc_bwd_weights)", M, K, N); return DCTR_ERR_UNSUPPORTED; }
    DCTR_TRY(sum_partials(wpart, (int64_t)K * N, splits, (int64_t)K * N, d_dw, st));
    if (d_db) DCTR_TRY(sum_partials(bpart, N, splits, N, d_db, st));
    return DCTR_OK;
}

int64_t dctr_gemm_split_launches(void) { return g_dr3_launches.load(std::memory_order_relaxed); }

}  // extern "C"
