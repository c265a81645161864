// Round 6 probe: producer-side split for the weight gradient (tf_repos_amd/csrc/gemm_dr.h: gemm_dr3 A_PRE / B_PRE, DrEpilogue::cf / csp,
// gemm_dr3_pair_kernel).  Correctness against an fp64-accumulated reference and timing with cycle stamps, one MI355X.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o tools/_bin/gemm_pl_probe tools/gemm_pl_probe.hip && tools/_bin/gemm_pl_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define DR_STAMPS
__device__ long long dr_stamps[16384 * 8];
#include "../tf_repos_amd/csrc/gemm_dr.h"
#include "gemm_dr3w_experiment.h"

namespace dctr {
__device__ __forceinline__ float dr_dropout_scale(uint64_t, uint64_t, float) { return 1.f; }
}
using namespace dctr;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// C[m][n] = sum_k A(m,k) B(k,n) in double; A given as [K][M] (kmajor) or [M][K]; B as [K][N] or [N][K]
__global__ void ref_gemm(const float* A, int lda, bool a_kmajor, const float* B, int ldb, bool b_kmajor, float* C, int M, int N, int K) {
    int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    double s = 0;
    for (int k = 0; k < K; ++k) {
        float a = a_kmajor ? A[(size_t)k * lda + m] : A[(size_t)m * lda + k];
        float b = b_kmajor ? B[(size_t)k * ldb + n] : B[(size_t)n * ldb + k];
        s += (double)a * b;
    }
    C[(size_t)m * N + n] = (float)s;
}

static void stamps_report(int nblocks) {
    const int nw = std::min(nblocks, 1024) * 4;
    std::vector<long long> hs(nw * 8);
    CK(hipMemcpyFromSymbol(hs.data(), HIP_SYMBOL(dr_stamps), hs.size() * 8));
    double avg[6] = {0}, mx[6] = {0};
    for (int b = 0; b < nw; ++b) {
        if (hs[b * 8 + 5] < hs[b * 8]) continue;
        for (int z = 1; z < 6; ++z) { double d = (double)(hs[b * 8 + z] - hs[b * 8 + z - 1]); avg[z] += d / nw; mx[z] = fmax(mx[z], d); }
    }
    printf("      cycles avg (max): setup+loads-issued %.0f (%.0f)  main %.0f (%.0f)  reduce+stage %.0f (%.0f)  barrier %.0f (%.0f)  stores(+planes) %.0f (%.0f)\n",
           avg[1], mx[1], avg[2], mx[2], avg[3], mx[3], avg[4], mx[4], avg[5], mx[5]);
}

template <class F>
static double time_us(F&& launch, int it = 100) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 10; ++i) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < it; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms * 1e3 / it;
}

// CF planes of X [R][W] (row-major, ld = W): dr_wsplit_kernel's "fwd" form of a [K = R][N = W] matrix
static void make_cf(const float* X, int R, int W, unsigned** cf, int64_t* plane_bytes) {
    const int64_t nfe = (int64_t)((R + 7) / 8) * W, nde = (int64_t)((W + 7) / 8) * R;
    unsigned* dgr;
    CK(hipMalloc(cf, nfe * 48)); CK(hipMalloc(&dgr, nde * 48));
    DrWsplitJobs J{};
    J.j[0] = DrWsplitJob{X, W, R, W, *cf, dgr, 0};
    J.n = 1; J.total = nfe + nde;
    dr_wsplit_kernel<0><<<(unsigned)((nfe + nde + 255) / 256), 256>>>(J);
    CK(hipDeviceSynchronize());
    hipFree(dgr);
    *plane_bytes = nfe * 16;
}

struct Data {
    int Mb, Kin, N;
    float *x, *dy, *w, *ref_dw, *part, *bias;
    unsigned *cfx, *cfdy; int64_t cfx_plane, cfdy_plane;
    std::vector<float> hx, hdy;
};
static Data make_data(int Mb, int Kin, int N) {
    Data d; d.Mb = Mb; d.Kin = Kin; d.N = N;
    d.hx.resize((size_t)Mb * Kin); d.hdy.resize((size_t)Mb * N);
    srand(Mb + Kin + N);
    for (auto& v : d.hx) v = (rand() / (float)RAND_MAX) * 2 - 1;
    for (auto& v : d.hdy) v = ((rand() / (float)RAND_MAX) * 2 - 1) * 1e-3f;
    std::vector<float> hw((size_t)Kin * N);
    for (auto& v : hw) v = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.05f;
    CK(hipMalloc(&d.x, d.hx.size() * 4)); CK(hipMalloc(&d.dy, d.hdy.size() * 4)); CK(hipMalloc(&d.w, hw.size() * 4));
    CK(hipMalloc(&d.ref_dw, (size_t)Kin * N * 4)); CK(hipMalloc(&d.part, (size_t)16 * Kin * N * 4)); CK(hipMalloc(&d.bias, N * 4));
    CK(hipMemset(d.bias, 0, N * 4));
    CK(hipMemcpy(d.x, d.hx.data(), d.hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d.dy, d.hdy.data(), d.hdy.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d.w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    make_cf(d.x, Mb, Kin, &d.cfx, &d.cfx_plane);
    make_cf(d.dy, Mb, N, &d.cfdy, &d.cfdy_plane);
    ref_gemm<<<dim3((N + 255) / 256, Kin), 256>>>(d.x, Kin, true, d.dy, N, true, d.ref_dw, Kin, N, Mb);
    CK(hipDeviceSynchronize());
    return d;
}

static double check_slabs(const Data& d, int splits) {
    std::vector<float> hp((size_t)splits * d.Kin * d.N), hr((size_t)d.Kin * d.N);
    CK(hipMemcpy(hp.data(), d.part, hp.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hr.data(), d.ref_dw, hr.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (size_t i = 0; i < hr.size(); ++i) {
        double s = 0;
        for (int z = 0; z < splits; ++z) s += hp[(size_t)z * d.Kin * d.N + i];
        worst = fmax(worst, fabs(s - hr[i]));
    }
    return worst;
}

// MODE 0: both operands split in registers (round 5), 1: hybrid (A = x f32 in registers, B = CF(dy)), 2: both pre-split
template <int TM, int TN, int MODE>
static void run_wgrad(const char* name, const Data& d, int splits) {
    const int M = d.Kin, N = d.N, K = d.Mb;
    const int nbm = (M + 16 * TM - 1) / (16 * TM), nbn = (N + 16 * TN - 1) / (16 * TN);
    const size_t lds = gemm_dr_lds_bytes<TM, TN>();
    const int kchunk = ((K + splits - 1) / splits + 31) / 32 * 32;
    DrEpilogue ep{};
    ep.split_stride = (int64_t)M * N;
    CK(hipMemset(d.part, 0, (size_t)splits * M * N * 4));
    auto launch = [&]() {
        if constexpr (MODE == 0) {
            auto k = gemm_dr3_kernel<TM, TN, false, false, true, DR_STORE, false, false>;
            static bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true); (void)once;
            k<<<dim3(nbm * nbn, splits), 256, lds, 0>>>(d.x, M, d.dy, N, d.part, N, M, N, K, kchunk, nbn, ep, 0, 0);
        } else if constexpr (MODE == 1) {
            auto k = gemm_dr3_kernel<TM, TN, false, true, false, DR_STORE, true, false>;
            static bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true); (void)once;
            k<<<dim3(nbm * nbn, splits), 256, lds, 0>>>(d.x, M, (const float*)d.cfdy, N, d.part, N, M, N, K, kchunk, nbn, ep, d.cfdy_plane, 0);
        } else {
            auto k = gemm_dr3_kernel<TM, TN, true, true, false, DR_STORE, true, true>;
            static bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true); (void)once;
            k<<<dim3(nbm * nbn, splits), 256, lds, 0>>>((const float*)d.cfx, M, (const float*)d.cfdy, N, d.part, N, M, N, K, kchunk, nbn, ep, d.cfdy_plane, d.cfx_plane);
        }
    };
    launch();
    CK(hipDeviceSynchronize());
    const double err = check_slabs(d, splits);
    const double us = time_us(launch);
    printf("[wgrad %s] %-10s out %dx%d over %d rows  TM=%d TN=%d splits=%d blocks=%d lds=%zuKB  %.2f us  %.1f TF  maxerr %.2e\n",
           MODE == 0 ? "both in registers" : MODE == 1 ? "hybrid: x f32, dy planes" : "both pre-split", name, M, N, K, TM, TN, splits, nbm * nbn * splits, lds / 1024, us,
           2.0 * M * N * K / us / 1e6, err);
    stamps_report(nbm * nbn * splits);
}

// the weight gradient on v_mfma_f32_32x32x16_bf16 (gemm_dr3w_kernel): both operands split in registers under the 8-pass MFMA
template <int TM, int TN>
static void run_wgrad32(const char* name, const Data& d, int splits) {
    const int M = d.Kin, N = d.N, K = d.Mb;
    const int nbm = (M + 32 * TM - 1) / (32 * TM), nbn = (N + 32 * TN - 1) / (32 * TN);
    const size_t lds = gemm_dr3w_lds_bytes<TM, TN>();
    const int kchunk = ((K + splits - 1) / splits + 63) / 64 * 64;
    DrEpilogue ep{};
    ep.split_stride = (int64_t)M * N;
    float* cs; CK(hipMalloc(&cs, (size_t)splits * N * 4)); CK(hipMemset(cs, 0, (size_t)splits * N * 4));
    ep.colsum = cs; ep.colsum_stride = N;
    CK(hipMemset(d.part, 0, (size_t)splits * M * N * 4));
    auto k = gemm_dr3w_kernel<TM, TN>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    auto launch = [&]() { k<<<dim3(nbm * nbn, splits), 256, lds, 0>>>(d.x, M, d.dy, N, d.part, N, M, N, K, kchunk, nbn, ep); };
    launch();
    CK(hipDeviceSynchronize());
    const double err = check_slabs(d, splits);
    std::vector<float> hc((size_t)splits * N);
    CK(hipMemcpy(hc.data(), cs, hc.size() * 4, hipMemcpyDeviceToHost));
    double err_cs = 0;
    for (int n = 0; n < N; ++n) {
        double sm = 0, r = 0;
        for (int z = 0; z < splits; ++z) sm += hc[(size_t)z * N + n];
        for (int m = 0; m < K; ++m) r += d.hdy[(size_t)m * N + n];
        err_cs = fmax(err_cs, fabs(sm - r));
    }
    const double us = time_us(launch);
    printf("[wgrad 32x32x16, both in registers] %-10s out %dx%d over %d rows  tile %dx%d splits=%d blocks=%d lds=%zuKB  %.2f us  %.1f TF  maxerr %.2e  colsum err %.2e\n", name, M, N, K, 32 * TM, 32 * TN,
           splits, nbm * nbn * splits, lds / 1024, us, 2.0 * M * N * K / us / 1e6, err, err_cs);
    stamps_report(nbm * nbn * splits);
    hipFree(cs);
}

// forward / dgrad product (A f32 split in registers, W pre-split) with and without the CF planes + column sums of its output
template <int TM, int TN, int EPI>
static void run_fwd(const char* name, int Mb, int K, int N, bool emit) {
    std::vector<float> hx((size_t)Mb * K), hw((size_t)K * N), hb(N), hact((size_t)Mb * N);
    srand(7);
    for (auto& v : hx) v = (rand() / (float)RAND_MAX) * 2 - 1;
    for (auto& v : hw) v = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.05f;
    for (auto& v : hb) v = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.1f;
    for (auto& v : hact) v = (rand() & 1) ? 1.f : 0.f;
    float *x, *w, *b, *y, *act, *csp;
    CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&w, hw.size() * 4)); CK(hipMalloc(&b, N * 4)); CK(hipMalloc(&y, (size_t)Mb * N * 4)); CK(hipMalloc(&act, hact.size() * 4));
    const int slabs = (Mb + 16 * TM - 1) / (16 * TM) + 3;
    CK(hipMalloc(&csp, (size_t)slabs * N * 4)); CK(hipMemset(csp, 0xff, (size_t)slabs * N * 4));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(act, hact.data(), hact.size() * 4, hipMemcpyHostToDevice));
    unsigned *pf, *pd;
    const int64_t nfe = (int64_t)((K + 7) / 8) * N, nde = (int64_t)((N + 7) / 8) * K;
    CK(hipMalloc(&pf, nfe * 48)); CK(hipMalloc(&pd, nde * 48));
    { DrWsplitJobs J{}; J.j[0] = DrWsplitJob{w, N, K, N, pf, pd, 0}; J.n = 1; J.total = nfe + nde; dr_wsplit_kernel<0><<<(unsigned)((nfe + nde + 255) / 256), 256>>>(J); }
    unsigned* cf; const int64_t cf_plane = (int64_t)((Mb + 7) / 8) * N * 16;
    CK(hipMalloc(&cf, cf_plane * 3)); CK(hipMemset(cf, 0xff, cf_plane * 3));
    const int nbm = (Mb + 16 * TM - 1) / (16 * TM), nbn = (N + 16 * TN - 1) / (16 * TN);
    const size_t lds = gemm_dr_lds_bytes<TM, TN>();
    DrEpilogue ep{};
    ep.bias = b; ep.relu = 1; ep.keep = 1.f; ep.act = act; ep.ldact = N; ep.inv_keep = 2.f;
    if (emit) { ep.cf = cf; ep.cf_plane = cf_plane; if (EPI == DR_MASK) { ep.csp = csp; ep.csp_stride = N; ep.csp_slabs = slabs; } }
    auto k = gemm_dr3_kernel<TM, TN, true, true, false, EPI, true, false>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int kchunk = (K + 31) / 32 * 32;
    auto launch = [&]() { k<<<dim3(nbm * nbn, 1), 256, lds, 0>>>(x, K, (const float*)pf, N, y, N, Mb, N, K, kchunk, nbn, ep, nfe * 16, 0); };
    launch();
    CK(hipDeviceSynchronize());
    double err_cf = 0, err_cs = 0;
    if (emit) {
        // the planes must be the split of the stored y, bit for bit: compare with dr_wsplit of y
        unsigned* cf2; int64_t pl2;
        make_cf(y, Mb, N, &cf2, &pl2);
        std::vector<unsigned> a((size_t)cf_plane * 3 / 4), c2((size_t)pl2 * 3 / 4);
        CK(hipMemcpy(a.data(), cf, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c2.data(), cf2, c2.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < a.size(); ++i) bad += a[i] != c2[i];
        err_cf = (double)bad;
        hipFree(cf2);
        if (EPI == DR_MASK) {
            std::vector<float> hy((size_t)Mb * N), hc((size_t)slabs * N);
            CK(hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hc.data(), csp, hc.size() * 4, hipMemcpyDeviceToHost));
            for (int n = 0; n < N; ++n) {
                double s = 0, r = 0;
                for (int z = 0; z < slabs; ++z) s += hc[(size_t)z * N + n];
                for (int m = 0; m < Mb; ++m) r += hy[(size_t)m * N + n];
                err_cs = fmax(err_cs, fabs(s - r));
            }
        }
    }
    const double us = time_us(launch);
    printf("[%s%s] %-10s M=%d K=%d N=%d TM=%d TN=%d blocks=%d  %.2f us  planes differing words %.0f  colsum err %.2e\n", EPI == DR_MASK ? "dgrad" : "fwd", emit ? " + CF planes" : "", name, Mb,
           K, N, TM, TN, nbm * nbn, us, err_cf, err_cs);
    stamps_report(nbm * nbn);
    hipFree(x); hipFree(w); hipFree(b); hipFree(y); hipFree(act); hipFree(csp); hipFree(pf); hipFree(pd); hipFree(cf);
}

// dgrad of layer l-1's input (A = dy f32, W pre-split, ReLU mask, CF planes out) and the weight gradient of layer l (both pre-split): two launches
// back to back vs ONE launch (gemm_dr3_pair_kernel)
template <bool WPRE>
static void run_pair(const Data& d, int splits, bool emit_cf = true) {
    using PD = Dr3Cfg<2, 7, true, true, false, DR_MASK, true, false>;
    using PW = std::conditional_t<WPRE, Dr3Cfg<2, 7, true, true, false, DR_STORE, true, true>, Dr3Cfg<2, 7, false, false, true, DR_STORE, false, false>>;
    const int Mb = d.Mb, Kin = d.Kin, N = d.N;
    // dgrad: dX [Mb][Kin] = dY [Mb][N] W^T: reduction over N, B = the n-blocked planes of W [Kin][N]
    unsigned *pf, *pd;
    const int64_t nfe = (int64_t)((Kin + 7) / 8) * N, nde = (int64_t)((N + 7) / 8) * Kin;
    CK(hipMalloc(&pf, nfe * 48)); CK(hipMalloc(&pd, nde * 48));
    { DrWsplitJobs J{}; J.j[0] = DrWsplitJob{d.w, N, Kin, N, pf, pd, 0}; J.n = 1; J.total = nfe + nde; dr_wsplit_kernel<0><<<(unsigned)((nfe + nde + 255) / 256), 256>>>(J); }
    float *dx, *act;
    CK(hipMalloc(&dx, (size_t)Mb * Kin * 4)); CK(hipMalloc(&act, (size_t)Mb * Kin * 4));
    CK(hipMemset(act, 0x3f, (size_t)Mb * Kin * 4));
    unsigned* cf; const int64_t cf_plane = (int64_t)((Mb + 7) / 8) * Kin * 16;
    CK(hipMalloc(&cf, cf_plane * 3));
    Dr3Arg a{}, b{};
    a.A = d.dy; a.lda = N; a.B = (const float*)pd; a.ldb = Kin; a.C = dx; a.ldc = Kin; a.M = Mb; a.N = Kin; a.K = N; a.kchunk = (N + 31) / 32 * 32;
    a.nbn = (Kin + 111) / 112; a.bplane = nde * 16; a.aplane = 0; a.gx = ((Mb + 31) / 32) * a.nbn; a.gy = 1;
    a.ep.act = act; a.ep.ldact = Kin; a.ep.inv_keep = 2.f;
    if (emit_cf) { a.ep.cf = cf; a.ep.cf_plane = cf_plane; }
    if (WPRE) { b.A = (const float*)d.cfx; b.lda = Kin; b.B = (const float*)d.cfdy; b.ldb = N; b.bplane = d.cfdy_plane; b.aplane = d.cfx_plane; }
    else { b.A = d.x; b.lda = Kin; b.B = d.dy; b.ldb = N; b.bplane = 0; b.aplane = 0; b.ep.colsum = d.bias; b.ep.colsum_stride = 0; }
    b.C = d.part; b.ldc = N; b.M = Kin; b.N = N; b.K = Mb;
    b.kchunk = ((Mb + splits - 1) / splits + 31) / 32 * 32; b.nbn = (N + 111) / 112;
    b.gx = ((Kin + 31) / 32) * b.nbn; b.gy = splits; b.ep.split_stride = (int64_t)Kin * N;
    const size_t lds = gemm_dr_lds_bytes<2, 7>();
    auto k1 = gemm_dr3_kernel<2, 7, true, true, false, DR_MASK, true, false>;
    auto k2 = gemm_dr3_kernel<PW::TM, PW::TN, PW::A_RC, PW::B_RC, PW::CS, PW::EPI, PW::B_PRE, PW::A_PRE>;
    auto kp = gemm_dr3_pair_kernel<PD, PW>;
    CK(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    auto two = [&]() {
        k1<<<dim3(a.gx, 1), 256, lds, 0>>>(a.A, a.lda, a.B, a.ldb, a.C, a.ldc, a.M, a.N, a.K, a.kchunk, a.nbn, a.ep, a.bplane, 0);
        k2<<<dim3(b.gx, b.gy), 256, lds, 0>>>(b.A, b.lda, b.B, b.ldb, b.C, b.ldc, b.M, b.N, b.K, b.kchunk, b.nbn, b.ep, b.bplane, b.aplane);
    };
    auto one = [&]() { kp<<<dim3(a.gx * a.gy + b.gx * b.gy), 256, lds, 0>>>(a, b, a.gx * a.gy); };
    auto one_w_first = [&]() { gemm_dr3_pair_kernel<PW, PD><<<dim3(a.gx * a.gy + b.gx * b.gy), 256, lds, 0>>>(b, a, b.gx * b.gy); };
    CK(hipFuncSetAttribute((const void*)gemm_dr3_pair_kernel<PW, PD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemset(d.part, 0, (size_t)splits * Kin * N * 4));
    one();
    CK(hipDeviceSynchronize());
    const double err = check_slabs(d, splits);
    std::vector<float> h1((size_t)Mb * Kin), h2((size_t)Mb * Kin);
    CK(hipMemcpy(h1.data(), dx, h1.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemset(dx, 0, (size_t)Mb * Kin * 4));
    two();
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h2.data(), dx, h2.size() * 4, hipMemcpyDeviceToHost));
    const bool same = memcmp(h1.data(), h2.data(), h1.size() * 4) == 0;
    const double t2 = time_us(two), t1 = time_us(one), t1w = time_us(one_w_first);
    auto only_d = [&]() { k1<<<dim3(a.gx, 1), 256, lds, 0>>>(a.A, a.lda, a.B, a.ldb, a.C, a.ldc, a.M, a.N, a.K, a.kchunk, a.nbn, a.ep, a.bplane, 0); };
    auto only_w = [&]() { k2<<<dim3(b.gx, b.gy), 256, lds, 0>>>(b.A, b.lda, b.B, b.ldb, b.C, b.ldc, b.M, b.N, b.K, b.kchunk, b.nbn, b.ep, b.bplane, b.aplane); };
    const double td = time_us(only_d), tw = time_us(only_w);
    printf("[pair] dgrad %dx%d<-%d (%s) and wgrad %dx%d over %d rows (%s, %d splits): dgrad alone %.2f, wgrad alone %.2f, two launches %.2f us, ONE launch %.2f us (dgrad blocks first) / %.2f us (wgrad blocks first); "
           "wgrad maxerr %.2e, dgrad bit-identical to its own launch: %s\n", Mb, Kin, N, emit_cf ? "+CF planes" : "f32 only", Kin, N, Mb, WPRE ? "both pre-split" : "both split in registers", splits, td, tw, t2, t1, t1w, err, same ? "yes" : "NO");
    hipFree(pf); hipFree(pd); hipFree(dx); hipFree(act); hipFree(cf);
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    Data L0 = make_data(4096, 624, 400), L1 = make_data(4096, 400, 400);
    printf("# weight gradient on the 8-pass MFMA\n");
    run_wgrad32<2, 4>("L0", L0, 6);
    run_wgrad32<2, 4>("L1", L1, 8);
    run_wgrad32<2, 4>("L1", L1, 9);
    { Data Rr = make_data(4001, 392, 616); run_wgrad32<2, 4>("ragged", Rr, 5); }
    if (argc > 1) return 0;
    printf("# weight gradient, c2 layer 0 (624 x 400 over 4096 rows) and layers 1 / 2 (400 x 400)\n");
    run_wgrad<4, 7, 0>("L0", L0, 6);
    run_wgrad<2, 7, 0>("L0", L0, 6);
    run_wgrad<4, 7, 1>("L0", L0, 6);
    run_wgrad<2, 7, 1>("L0", L0, 6);
    run_wgrad<4, 7, 2>("L0", L0, 6);
    run_wgrad<2, 7, 2>("L0", L0, 6);
    run_wgrad<2, 7, 2>("L0", L0, 4);
    run_wgrad<4, 7, 0>("L1", L1, 9);
    run_wgrad<4, 7, 2>("L1", L1, 9);
    run_wgrad<2, 7, 2>("L1", L1, 9);
    run_wgrad<2, 7, 2>("L1", L1, 8);
    run_wgrad<2, 7, 2>("L1", L1, 4);
    Data R = make_data(4001, 392, 616);       // ragged batch (not a multiple of 8), widths not multiples of the tiles
    run_wgrad<2, 7, 1>("ragged", R, 5);
    run_wgrad<2, 7, 2>("ragged", R, 5);
    printf("# producers: forward / dgrad products with the CF planes (+ column sums) of their output\n");
    run_fwd<2, 7, DR_BIAS_ACT>("L1", 4096, 400, 400, false);
    run_fwd<2, 7, DR_BIAS_ACT>("L1", 4096, 400, 400, true);
    run_fwd<2, 7, DR_BIAS_ACT>("L0", 4096, 624, 400, false);
    run_fwd<2, 7, DR_BIAS_ACT>("L0", 4096, 624, 400, true);
    run_fwd<2, 7, DR_MASK>("L1", 4096, 400, 400, false);
    run_fwd<2, 7, DR_MASK>("L1", 4096, 400, 400, true);
    run_fwd<2, 7, DR_MASK>("ragged", 4001, 392, 616, true);
    run_fwd<4, 7, DR_BIAS_ACT>("L1", 4096, 400, 400, true);
    printf("# one launch for a layer's weight gradient and the dgrad of the layer below\n");
    run_pair<true>(L1, 9);
    run_pair<true>(L1, 8);
    run_pair<false>(L1, 9, false);
    run_pair<false>(L1, 8, false);
    run_pair<false>(L0, 6, false);
    return 0;
}
