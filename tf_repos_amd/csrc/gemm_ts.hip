// K6, host side of the tall split-precision products (gemm_ts.h): AFM's attention layer over the B * P pair rows (AFM.py:142-147) and its
// gradients in dctr_config.gemm_mode = 1.  The small operand's planes are rewritten per call by ts_wsplit_kernel (65 k elements: ~3 us
// beside a product of milliseconds) into a workspace the caller owns (ts_plane_bytes); the engine writes both plane sets of a step with one launch (ts_prepare).  A shape these kernels do not take leaves
// *done = false: the caller runs the exact kernels (gemm_ws.hip / gemm_dr.hip).
//
// Compiled with -fno-slp-vectorize like gemm_dr3.hip (build.py): the split's subtractions stay two v_sub_f32.
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "common.h"
#include "gemm_ts.h"
#include "ops.h"

namespace dctr {

__device__ __forceinline__ float dr_dropout_scale(uint64_t seed, uint64_t idx, float keep) { return dropout_scale(seed, idx, keep); }

extern std::atomic<int64_t> g_dr3_launches;          // (gemm_dr3.hip) split-kernel launches of this process: dctr_gemm_split_launches

namespace {

constexpr int CUS = 256;
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

bool ts_enabled() {
    static const bool on = [] { const char* e = getenv("DCTR_GEMM_TS"); return e == nullptr || e[0] != '0'; }();       // A/B knob
    return on;
}
bool ts_dim_ok(int d) { return d == 128 || d == 256; }

template <int KG, int NT, int MODE>
int ts_launch(const TsArgs& a, hipStream_t st) {
    auto kern = gemm_ts_kernel<KG, NT, MODE>;
    constexpr int lds = 2 * 3 * 4 * 16 * NT * 16;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    DCTR_HIP_CHECK(attr);
    const int grid = (int)std::min<int64_t>((a.M + 255) / 256, CUS);
    kern<<<grid, 256, lds, st>>>(a);
    DCTR_LAUNCH_CHECK();
    g_dr3_launches.fetch_add(1, std::memory_order_relaxed);
    return DCTR_OK;
}
template <int MODE>
int ts_dispatch(int R, int N, const TsArgs& a, hipStream_t st) {
    if (R == 128 && N == 128) return ts_launch<4, 8, MODE>(a, st);
    if (R == 128 && N == 256) return ts_launch<4, 16, MODE>(a, st);
    if (R == 256 && N == 128) return ts_launch<8, 8, MODE>(a, st);
    return ts_launch<8, 16, MODE>(a, st);
}
int ts_split(const float* w, int ldw, const TsSplitJob& j0, const TsSplitJob* j1, hipStream_t st) {
    const int n0 = j0.R / 8 * j0.N, n1 = j1 ? j1->R / 8 * j1->N : 0;
    ts_wsplit_kernel<<<dim3(ceil_div(std::max(n0, n1), 256), j1 ? 2 : 1), 256, 0, st>>>(w, ldw, j0, j1 ? *j1 : j0);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

}  // namespace

// host logic: would a product over M rows with a reduction of R and N output columns take these kernels (alignment permitting)?
bool ts_takes(int64_t M, int R, int N) { return ts_enabled() && M >= 65536 && ts_dim_ok(R) && ts_dim_ok(N); }
size_t ts_plane_bytes(int R, int N) { return (size_t)3 * R * N * 2; }

// both plane sets of one layer weight W [K, N] in one launch: the forward's into fwd_planes, the gated input gradient's (x kscale) into dgr_planes
int ts_prepare(const float* w, int K, int N, const float* kscale, void* fwd_planes, void* dgr_planes, hipStream_t st) {
    const TsSplitJob jd{1, kscale, N, K, static_cast<u32x4*>(dgr_planes)};
    return ts_split(w, N, TsSplitJob{0, nullptr, K, N, static_cast<u32x4*>(fwd_planes)}, &jd, st);
}

// Y[M,N] = relu(X[M,K] W[K,N] + b), dot_out[row] = <Y[row,:], dot_w> (dot_out may be null)
int ts_fc_fwd_dot(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int64_t M, int K, int N, const float* dot_w,
                  float* dot_out, void* planes_ws, bool split_here, hipStream_t st, bool* done) {
    *done = false;
    if (!ts_takes(M, K, N) || planes_ws == nullptr || !al16(w) || !al16(x) || (ldx & 3) || !al16(y) || (ldy & 3) || !al16(b) || !al16(dot_w) || b == nullptr ||
        dot_w == nullptr || ldx > (1 << 20) || ldy > (1 << 20))
        return DCTR_OK;
    if (split_here) DCTR_TRY(ts_split(w, N, TsSplitJob{0, nullptr, K, N, static_cast<u32x4*>(planes_ws)}, nullptr, st));
    TsArgs a{};
    a.A = x; a.lda = ldx; a.planes = static_cast<const u32x4*>(planes_ws); a.C = y; a.ldc = ldy; a.M = M; a.bias = b; a.dot_w = dot_w; a.dot_out = dot_out;
    *done = true;
    return ts_dispatch<TS_FWD>(K, N, a, st);
}

// dX[M,K] = (rowscale (x) kscale . 1[H > 0]) W[K,N]^T with H [M,N] the layer's ReLU output (gemm_ws.hip WS_GATE, three bf16 products)
int ts_fc_bwd_data_gate(const float* h, int ldh, const float* rowscale, const float* kscale, const float* w, float* dx, int lddx, int64_t M, int K,
                        int N, void* planes_ws, bool split_here, hipStream_t st, bool* done) {
    *done = false;
    if (!ts_takes(M, N, K) || planes_ws == nullptr || !al16(w) || !al16(h) || (ldh & 3) || !al16(dx) || (lddx & 3) || rowscale == nullptr || kscale == nullptr ||
        ldh > (1 << 20) || lddx > (1 << 20))
        return DCTR_OK;
    if (split_here) DCTR_TRY(ts_split(w, N, TsSplitJob{1, kscale, N, K, static_cast<u32x4*>(planes_ws)}, nullptr, st));
    TsArgs a{};
    a.A = h; a.lda = ldh; a.planes = static_cast<const u32x4*>(planes_ws); a.C = dx; a.ldc = lddx; a.M = M; a.rowscale = rowscale;
    *done = true;
    return ts_dispatch<TS_GATE>(N, K, a, st);
}

}  // namespace dctr

using namespace dctr;

extern "C" {

int dctr_ts_plane_bytes(int R, int N, int64_t* bytes) {
    DCTR_REQUIRE(R > 0 && N > 0 && bytes != nullptr, "dctr_ts_plane_bytes: R, N > 0 and an output");
    *bytes = (int64_t)ts_plane_bytes(R, N);
    return DCTR_OK;
}

int dctr_fc_fwd_dot_split(const float* d_x, int ldx, const float* d_w, const float* d_b, float* d_y, int ldy, int64_t M, int K, int N,
                          const float* d_dot_w, float* d_dot_out, void* d_planes_ws, void* stream) {
    bool done = false;
    DCTR_TRY(ts_fc_fwd_dot(d_x, ldx, d_w, d_b, d_y, ldy, M, K, N, d_dot_w, d_dot_out, d_planes_ws, true, as_stream(stream), &done));
    if (!done) { set_error("dctr_fc_fwd_dot_split: no tall split-precision kernel takes M=%lld K=%d N=%d with these pointers", (long long)M, K, N); return DCTR_ERR_UNSUPPORTED; }
    return DCTR_OK;
}

int dctr_fc_bwd_data_gate_split(const float* d_h, int ldh, const float* d_rowscale, const float* d_kscale, const float* d_w, float* d_dx, int lddx,
                                int64_t M, int K, int N, void* d_planes_ws, void* stream) {
    bool done = false;
    DCTR_TRY(ts_fc_bwd_data_gate(d_h, ldh, d_rowscale, d_kscale, d_w, d_dx, lddx, M, K, N, d_planes_ws, true, as_stream(stream), &done));
    if (!done) { set_error("dctr_fc_bwd_data_gate_split: no tall split-precision kernel takes M=%lld K=%d N=%d with these pointers", (long long)M, K, N); return DCTR_ERR_UNSUPPORTED; }
    return DCTR_OK;
}

}  // extern "C"
