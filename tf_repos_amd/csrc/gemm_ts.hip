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

// Eight waves of 32 rows per block (gemm_ts.h TM = 2) for the products over stored / generated rows: forward 2.63 -> 2.10 ms, gated input
// gradient 1.70 -> 1.55 ms at 3.0 M rows against four waves of 64, bit-identical (tools/gemm_ts_probe; DCTR_GEMM_TS_WAVES=4 is the A/B
// knob).  The gate from the forward's sign bits runs in two column halves at two blocks per CU: 1.20 ms.
template <int KG, int NT, int MODE, bool GEN, int TM, int NTF = NT>
int ts_launch(const TsArgs& a, hipStream_t st) {
    auto kern = gemm_ts_kernel<KG, NT, MODE, GEN, TM, NTF>;
    constexpr int lds = 3 * 3 * 4 * 16 * NT * 16;          // three images of a group's planes (of this pass's columns)
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    DCTR_HIP_CHECK(attr);
    constexpr int H = NTF / NT;                             // column halves: two blocks per CU
    // Persistent blocks (one per CU slot, looping over the row tiles) save the prologue of every tile but hold every CU until the kernel ends;
    // with few tiles per slot the step's other streams (next batch's grouping, table sweep) pay for that with a wait per product.
    // AFM at run.sh:18's B = 128, A = 128 (371 row tiles): 0.73 ms/step with looping blocks, 0.41 with one block per tile (the f32 kernels: 0.58).
    // DCTR_GEMM_TS_PERSIST_TILES: from this many tiles per slot on the blocks loop (default 2); below, one block per tile.
    static const int persist_from = getenv("DCTR_GEMM_TS_PERSIST_TILES") ? atoi(getenv("DCTR_GEMM_TS_PERSIST_TILES")) : 2;     // A/B knob
    const int64_t tiles = (a.M + 255) / 256 * H;
    const int grid = tiles >= (int64_t)persist_from * CUS * H ? CUS * H : (int)std::min<int64_t>(tiles, 1 << 20);
    kern<<<grid, 64 * (16 / TM), lds, st>>>(a);
    DCTR_LAUNCH_CHECK();
    g_dr3_launches.fetch_add(1, std::memory_order_relaxed);
    return DCTR_OK;
}
template <int MODE, bool GEN, int TM>
int ts_dispatch_t(int R, int N, const TsArgs& a, hipStream_t st) {
    if (R == 128 && N == 128) return ts_launch<4, 8, MODE, GEN, TM>(a, st);
    if (R == 128 && N == 256) return ts_launch<4, 16, MODE, GEN, TM>(a, st);
    if (R == 256 && N == 128) return ts_launch<8, 8, MODE, GEN, TM>(a, st);
    return ts_launch<8, 16, MODE, GEN, TM>(a, st);
}
template <int MODE, bool GEN = false>
int ts_dispatch(int R, int N, const TsArgs& a, hipStream_t st) {
    static const bool four = getenv("DCTR_GEMM_TS_WAVES") != nullptr && atoi(getenv("DCTR_GEMM_TS_WAVES")) == 4;
    return four ? ts_dispatch_t<MODE, GEN, 4>(R, N, a, st) : ts_dispatch_t<MODE, GEN, 2>(R, N, a, st);
}
// the gate from the forward's sign bits, in two column halves at two blocks per CU (gemm_ts.h NTF)
int ts_dispatch_bits(int R, int N, const TsArgs& a, hipStream_t st) {
    if (R == 128 && N == 128) return ts_launch<4, 4, TS_GATE, true, 4, 8>(a, st);
    if (R == 128 && N == 256) return ts_launch<4, 8, TS_GATE, true, 4, 16>(a, st);
    if (R == 256 && N == 128) return ts_launch<8, 4, TS_GATE, true, 4, 8>(a, st);
    return ts_launch<8, 8, TS_GATE, true, 4, 16>(a, st);
}
int ts_split(const float* w, int ldw, const TsSplitJob& j0, const TsSplitJob* j1, hipStream_t st) {
    const int n0 = j0.R / 8 * j0.N, n1 = j1 ? j1->R / 8 * j1->N : 0;
    ts_wsplit_kernel<<<dim3(ceil_div(std::max(n0, n1), 256), j1 ? 2 : 1), 256, 0, st>>>(w, ldw, j0, j1 ? *j1 : j0);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

template <int NW, int TK, int TA, bool GEN = false, bool HB = false>
int tsw_launch(const TswArgs& a, int grid, hipStream_t st) {
    auto kern = gemm_tsw_kernel<NW, TK, TA, GEN, HB>;
    const int lds = 2 * NW * TA * 1024 + (GEN ? a.P * 4 : 0);
    kern<<<grid, 64 * NW, lds, st>>>(a);
    DCTR_LAUNCH_CHECK();
    g_dr3_launches.fetch_add(1, std::memory_order_relaxed);
    return DCTR_OK;
}

}  // namespace

// host logic: would a product over M rows with a reduction of R and N output columns take these kernels (alignment permitting)?
bool ts_takes(int64_t M, int R, int N) { return ts_enabled() && M >= 65536 && ts_dim_ok(R) && ts_dim_ok(N); }
// are the forward's sign words used by the two gradient products (DCTR_GEMM_TS_BITS=0: they read the layer's output itself)?
bool ts_bits_enabled() {
    static const bool off = getenv("DCTR_GEMM_TS_BITS") != nullptr && atoi(getenv("DCTR_GEMM_TS_BITS")) == 0;      // A/B knob
    return !off;
}
size_t ts_plane_bytes(int R, int N) { return (size_t)3 * R * N * 2; }

// both plane sets of one layer weight W [K, N] in one launch: the forward's into fwd_planes, the gated input gradient's (x kscale) into dgr_planes
int ts_prepare(const float* w, int K, int N, const float* kscale, void* fwd_planes, void* dgr_planes, hipStream_t st) {
    const TsSplitJob jd{1, kscale, N, K, static_cast<u32x4*>(dgr_planes)};
    return ts_split(w, N, TsSplitJob{0, nullptr, K, N, static_cast<u32x4*>(fwd_planes)}, &jd, st);
}

// rows generated from pairs of embeddings: all of e behind one 31-bit descriptor, 16-byte loads, field offsets in 16 bits, every row in range
bool ts_pairs_ok(const TsPairs* g, int64_t M, int K) {
    return g != nullptr && g->e != nullptr && al16(g->e) && (g->e_ld & 3) == 0 && g->examples > 0 && g->P > 32 && g->P < 32768 && g->pair_i != nullptr &&
           g->pair_j != nullptr && (int64_t)g->examples * g->e_ld * 4 < (int64_t)0x7fff0000 && g->e_ld <= 65535 && (int64_t)g->e_ld >= K &&
           M <= (int64_t)g->examples * g->P;
}

// Y[M,N] = relu(X[M,K] W[K,N] + b), dot_out[row] = <Y[row,:], dot_w> (dot_out may be null).  pairs != null: X is not read -- row b P + p is
// e[b, pair_i[p], :] . e[b, pair_j[p], :] (TsPairs), formed in the registers.
int ts_fc_fwd_dot(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int64_t M, int K, int N, const float* dot_w,
                  float* dot_out, void* planes_ws, bool split_here, hipStream_t st, bool* done, const TsPairs* pairs, void* sign_bits_out) {
    *done = false;
    if (pairs != nullptr && !ts_pairs_ok(pairs, M, K)) return DCTR_OK;
    if (pairs != nullptr) { x = pairs->e; ldx = 4; }
    if (!ts_takes(M, K, N) || planes_ws == nullptr || !al16(w) || !al16(x) || (ldx & 3) || !al16(y) || (ldy & 3) || !al16(b) || !al16(dot_w) || b == nullptr ||
        dot_w == nullptr || ldx > (1 << 20) || ldy > (1 << 20) || (y == nullptr && sign_bits_out == nullptr))
        return DCTR_OK;
    if (split_here) DCTR_TRY(ts_split(w, N, TsSplitJob{0, nullptr, K, N, static_cast<u32x4*>(planes_ws)}, nullptr, st));
    TsArgs a{};
    a.A = x; a.lda = ldx; a.planes = static_cast<const u32x4*>(planes_ws); a.C = y; a.ldc = ldy; a.M = M; a.bias = b; a.dot_w = dot_w; a.dot_out = dot_out;
    a.bits_out = static_cast<unsigned long long*>(sign_bits_out);
    *done = true;
    if (pairs != nullptr) {
        a.e = pairs->e; a.e_ld = pairs->e_ld; a.e_floats = (int64_t)pairs->examples * pairs->e_ld; a.pair_i = pairs->pair_i; a.pair_j = pairs->pair_j; a.P = pairs->P;
        return ts_dispatch<TS_FWD, true>(K, N, a, st);
    }
    return ts_dispatch<TS_FWD>(K, N, a, st);
}

// dX[M,K] = (rowscale (x) kscale . 1[H > 0]) W[K,N]^T with H [M,N] the layer's ReLU output (gemm_ws.hip WS_GATE, three bf16 products).
// sign_bits: the words ts_fc_fwd_dot wrote beside H (32 bytes per row) -- read instead of H when given (h may then be null... of the same forward)
int ts_fc_bwd_data_gate(const float* h, int ldh, const float* rowscale, const float* kscale, const float* w, float* dx, int lddx, int64_t M, int K,
                        int N, void* planes_ws, bool split_here, hipStream_t st, bool* done, const void* sign_bits) {
    *done = false;
    if (sign_bits != nullptr && h == nullptr) { h = rowscale; ldh = 4; }      // (the rows are not read: any aligned pointer passes the checks below)
    if (!ts_takes(M, N, K) || planes_ws == nullptr || !al16(w) || !al16(h) || (ldh & 3) || !al16(dx) || (lddx & 3) || rowscale == nullptr || kscale == nullptr ||
        ldh > (1 << 20) || lddx > (1 << 20))
        return DCTR_OK;
    if (split_here) DCTR_TRY(ts_split(w, N, TsSplitJob{1, kscale, N, K, static_cast<u32x4*>(planes_ws)}, nullptr, st));
    TsArgs a{};
    a.A = h; a.lda = ldh; a.planes = static_cast<const u32x4*>(planes_ws); a.C = dx; a.ldc = lddx; a.M = M; a.rowscale = rowscale;
    *done = true;
    if (sign_bits != nullptr && ts_bits_enabled() && (reinterpret_cast<uintptr_t>(sign_bits) & 7) == 0) {
        a.bits_in = static_cast<const unsigned long long*>(sign_bits);
        return ts_dispatch_bits(N, K, a, st);
    }
    return ts_dispatch<TS_GATE>(N, K, a, st);
}

// The layer's weight gradient under the rank-one output gradient d H = rowscale (x) colscale . 1[H > 0] (gemm_dr.hip dr_fc_bwd_weights_partials_gate in
// split precision): `splits` partial slabs over contiguous row ranges -- dw_part [K, N], db_part [N] (bias gradient), dwo_part [N] = the
// second column sums sum_r rowscale[r] H[r, :] (the weight gradient of the (N -> 1) layer that produced the rank-one form).  Any M >= 0
// (slabs beyond the rows are written as zeros); one block per slab: 256 slabs fill the chip.
int ts_fc_bwd_weights_gate(const float* x, int ldx, const float* h, int ldh, const float* rowscale, const float* colscale, float* dw_part,
                           int64_t dw_stride, float* db_part, int64_t db_stride, float* dwo_part, int64_t dwo_stride, int64_t M, int K, int N, int splits,
                           hipStream_t st, bool* done, const TsPairs* pairs, const void* sign_bits, const float* w, const float* bias) {
    *done = false;
    if (pairs != nullptr && !ts_pairs_ok(pairs, M, K)) return DCTR_OK;
    if (pairs != nullptr) { x = pairs->e; ldx = 4; }
    // the gate from the forward's sign words, the second sums from the product (gemm_ts.h HB): with generated rows only (what the engine runs)
    const bool hb = sign_bits != nullptr && pairs != nullptr && w != nullptr && bias != nullptr && ts_bits_enabled() && (reinterpret_cast<uintptr_t>(sign_bits) & 7) == 0;
    if (hb && h == nullptr) { h = rowscale; ldh = 4; }
    if (!ts_enabled() || !ts_dim_ok(K) || !ts_dim_ok(N) || splits < 1 || splits > 65535 || M < 0 || !al16(x) || !al16(h) || (ldx & 3) || (ldh & 3) ||
        !al16(rowscale) || colscale == nullptr || !al16(dw_part) || (dw_stride & 3) || ldx > (1 << 16) || ldh > (1 << 16))
        return DCTR_OK;
    const int64_t rpb = round_up(ceil_div(std::max<int64_t>(M, 1), (int64_t)splits), 32);
    if ((rpb + 64) * std::max(ldx, ldh) * 4 >= (int64_t)0x7fff0000) return DCTR_OK;        // (32-bit byte offsets inside a block's rows)
    TswArgs a{};
    a.X = x; a.ldx = ldx; a.H = h; a.ldh = ldh; a.rowscale = rowscale; a.colscale = colscale; a.dw = dw_part; a.dw_stride = dw_stride;
    a.db = db_part; a.db_stride = db_stride; a.dwo = dwo_part; a.dwo_stride = dwo_stride; a.M = M; a.rows_per_block = (int)rpb;
    *done = true;
    if (hb) {
        a.e = pairs->e; a.e_ld = pairs->e_ld; a.e_floats = (int64_t)pairs->examples * pairs->e_ld; a.pair_i = pairs->pair_i; a.pair_j = pairs->pair_j; a.P = pairs->P;
        a.bits = static_cast<const unsigned long long*>(sign_bits); a.W = w; a.bias = bias; a.H = nullptr;
        if (K == 256 && N == 256) return tsw_launch<8, 2, 2, true, true>(a, splits, st);
        if (K == 256 && N == 128) return tsw_launch<4, 4, 2, true, true>(a, splits, st);
        if (K == 128 && N == 256) return tsw_launch<4, 2, 4, true, true>(a, splits, st);
        return tsw_launch<4, 2, 2, true, true>(a, splits, st);
    }
    if (pairs != nullptr) {
        a.e = pairs->e; a.e_ld = pairs->e_ld; a.e_floats = (int64_t)pairs->examples * pairs->e_ld; a.pair_i = pairs->pair_i; a.pair_j = pairs->pair_j; a.P = pairs->P;
        if (K == 256 && N == 256) return tsw_launch<8, 2, 2, true>(a, splits, st);
        if (K == 256 && N == 128) return tsw_launch<4, 4, 2, true>(a, splits, st);
        if (K == 128 && N == 256) return tsw_launch<4, 2, 4, true>(a, splits, st);
        return tsw_launch<4, 2, 2, true>(a, splits, st);
    }
    if (K == 256 && N == 256) return tsw_launch<8, 2, 2>(a, splits, st);
    if (K == 256 && N == 128) return tsw_launch<4, 4, 2>(a, splits, st);
    if (K == 128 && N == 256) return tsw_launch<4, 2, 4>(a, splits, st);
    return tsw_launch<4, 2, 2>(a, splits, st);
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
                          const float* d_dot_w, float* d_dot_out, void* d_sign_bits, void* d_planes_ws, void* stream) {
    bool done = false;
    DCTR_TRY(ts_fc_fwd_dot(d_x, ldx, d_w, d_b, d_y, ldy, M, K, N, d_dot_w, d_dot_out, d_planes_ws, true, as_stream(stream), &done, nullptr, d_sign_bits));
    if (!done) { set_error("dctr_fc_fwd_dot_split: no tall split-precision kernel takes M=%lld K=%d N=%d with these pointers", (long long)M, K, N); return DCTR_ERR_UNSUPPORTED; }
    return DCTR_OK;
}

int dctr_fc_bwd_data_gate_split(const float* d_h, int ldh, const void* d_sign_bits, const float* d_rowscale, const float* d_kscale, const float* d_w,
                                float* d_dx, int lddx, int64_t M, int K, int N, void* d_planes_ws, void* stream) {
    bool done = false;
    DCTR_TRY(ts_fc_bwd_data_gate(d_h, ldh, d_rowscale, d_kscale, d_w, d_dx, lddx, M, K, N, d_planes_ws, true, as_stream(stream), &done, d_sign_bits));
    if (!done) { set_error("dctr_fc_bwd_data_gate_split: no tall split-precision kernel takes M=%lld K=%d N=%d with these pointers", (long long)M, K, N); return DCTR_ERR_UNSUPPORTED; }
    return DCTR_OK;
}

static int bwd_weights_gate_split(const float* d_x, int ldx, const TsPairs* pairs, const float* d_h, int ldh, const void* sign_bits, const float* d_w, const float* d_bias,
                                  const float* d_rowscale, const float* d_colscale, float* d_dw, float* d_db, float* d_dwo, int64_t M, int K, int N, float* d_workspace,
                                  size_t workspace_bytes, void* stream) {
    hipStream_t st = as_stream(stream);
    DCTR_REQUIRE(d_dw != nullptr && d_db != nullptr && d_dwo != nullptr, "fc_bwd_weights_gate_split: three outputs");
    const size_t per = ((size_t)K * N + 2 * (size_t)N) * sizeof(float);
    DCTR_REQUIRE(d_workspace != nullptr && workspace_bytes >= per, "fc_bwd_weights_gate_split: workspace of at least (K N + 2 N) floats");
    int splits = (int)std::min<int64_t>(CUS, std::max<int64_t>(1, M / 256));
    if ((size_t)splits * per > workspace_bytes) splits = (int)(workspace_bytes / per);
    float* wpart = d_workspace;
    float* bpart = wpart + (size_t)splits * K * N;
    float* opart = bpart + (size_t)splits * N;
    bool done = false;
    DCTR_TRY(ts_fc_bwd_weights_gate(d_x, ldx, d_h, ldh, d_rowscale, d_colscale, wpart, (int64_t)K * N, bpart, N, opart, N, M, K, N, splits, st, &done, pairs, sign_bits, d_w, d_bias));
    if (!done) { set_error("fc_bwd_weights_gate_split: no tall split-precision kernel takes M=%lld K=%d N=%d with these pointers", (long long)M, K, N); return DCTR_ERR_UNSUPPORTED; }
    DCTR_TRY(sum_partials(wpart, (int64_t)K * N, splits, (int64_t)K * N, d_dw, st));
    DCTR_TRY(sum_partials(bpart, N, splits, N, d_db, st));
    DCTR_TRY(sum_partials(opart, N, splits, N, d_dwo, st));
    return DCTR_OK;
}

int dctr_fc_bwd_weights_gate_split(const float* d_x, int ldx, const float* d_h, int ldh, const float* d_rowscale, const float* d_colscale, float* d_dw,
                                   float* d_db, float* d_dwo, int64_t M, int K, int N, float* d_workspace, size_t workspace_bytes, void* stream) {
    return bwd_weights_gate_split(d_x, ldx, nullptr, d_h, ldh, nullptr, nullptr, nullptr, d_rowscale, d_colscale, d_dw, d_db, d_dwo, M, K, N, d_workspace, workspace_bytes, stream);
}

int dctr_pairs_fc_fwd_dot_split(const float* d_e, int e_ld, int examples, const int16_t* d_pair_i, const int16_t* d_pair_j, int P, const float* d_w,
                                const float* d_b, float* d_y, int ldy, int64_t M, int K, int N, const float* d_dot_w, float* d_dot_out, void* d_sign_bits,
                                void* d_planes_ws, void* stream) {
    const TsPairs pairs{d_e, e_ld, examples, d_pair_i, d_pair_j, P};
    bool done = false;
    DCTR_TRY(ts_fc_fwd_dot(nullptr, 0, d_w, d_b, d_y, ldy, M, K, N, d_dot_w, d_dot_out, d_planes_ws, true, as_stream(stream), &done, &pairs, d_sign_bits));
    if (!done) { set_error("dctr_pairs_fc_fwd_dot_split: no tall split-precision kernel takes M=%lld K=%d N=%d over these embeddings", (long long)M, K, N); return DCTR_ERR_UNSUPPORTED; }
    return DCTR_OK;
}

int dctr_pairs_fc_bwd_weights_gate_split(const float* d_e, int e_ld, int examples, const int16_t* d_pair_i, const int16_t* d_pair_j, int P, const float* d_h,
                                         int ldh, const void* d_sign_bits, const float* d_w, const float* d_b, const float* d_rowscale, const float* d_colscale,
                                         float* d_dw, float* d_db, float* d_dwo, int64_t M, int K, int N, float* d_workspace, size_t workspace_bytes, void* stream) {
    const TsPairs pairs{d_e, e_ld, examples, d_pair_i, d_pair_j, P};
    return bwd_weights_gate_split(nullptr, 0, &pairs, d_h, ldh, d_sign_bits, d_w, d_b, d_rowscale, d_colscale, d_dw, d_db, d_dwo, M, K, N, d_workspace, workspace_bytes,
                                  stream);
}

}  // extern "C"
