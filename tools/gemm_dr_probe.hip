// Stand-alone timing / correctness probe of tf_repos_amd/csrc/gemm_dr.h (the direct-to-register wave-split-K f32 MFMA GEMM).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/_bin/gemm_dr_probe tools/gemm_dr_probe.hip && tools/_bin/gemm_dr_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define DR_STAMPS
__device__ long long dr_stamps[4096 * 8];
#include "../tf_repos_amd/csrc/gemm_dr.h"

namespace dctr {
__device__ __forceinline__ float dr_dropout_scale(uint64_t, uint64_t, float) { return 1.f; }
}
using namespace dctr;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void ref_gemm(const float* A, int lda, bool a_rc, const float* B, int ldb, bool b_rc, float* C, int M, int N, int K, const float* bias, int relu) {
    int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    double s = 0;
    for (int k = 0; k < K; ++k) {
        float a = a_rc ? A[(size_t)m * lda + k] : A[(size_t)k * lda + m];
        float b = b_rc ? B[(size_t)n * ldb + k] : B[(size_t)k * ldb + n];
        s += (double)a * b;
    }
    float v = (float)s + (bias ? bias[n] : 0.f);
    C[(size_t)m * N + n] = (relu && v < 0) ? 0 : v;
}

template <int TM, int TN, bool A_RC, bool B_RC, bool CS, bool P3 = false, bool BP = false>
void run(const char* name, int M, int N, int K, int splits = 1) {
    std::vector<float> hA((size_t)M * K), hB((size_t)K * N), hb(N);
    srand(1);
    for (auto& v : hA) v = (rand() / (float)RAND_MAX) * 2 - 1;
    for (auto& v : hB) v = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.05f;
    for (auto& v : hb) v = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.1f;
    float *A, *B, *C, *R, *bias, *cs;
    CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&B, hB.size() * 4)); CK(hipMalloc(&C, (size_t)splits * M * N * 4)); CK(hipMalloc(&R, (size_t)M * N * 4));
    CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&cs, (size_t)splits * N * 4));
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
    const int lda = A_RC ? K : M, ldb = B_RC ? K : N;
    const int nbm = (M + 16 * TM - 1) / (16 * TM), nbn = (N + 16 * TN - 1) / (16 * TN);
    const size_t lds = gemm_dr_lds_bytes<TM, TN>();
    constexpr int EPI = CS ? DR_STORE : DR_BIAS_ACT;
    const void* kfn;
    // pre-split B (weights): both plane forms of the B matrix as stored ([K][N] for fwd, [N][K] for dgrad), the product reads one
    unsigned *pf = nullptr, *pd = nullptr;
    const int Wr = B_RC ? N : K, Wc = B_RC ? K : N;          // B as a row-major matrix [Wr][Wc]
    const int64_t nfe = (int64_t)((Wr + 7) / 8) * Wc, nde = (int64_t)((Wc + 7) / 8) * Wr;
    if constexpr (BP) {
        CK(hipMalloc(&pf, nfe * 48)); CK(hipMalloc(&pd, nde * 48));
        {
            DrWsplitJobs J{};
            J.j[0] = DrWsplitJob{B, Wc, Wr, Wc, pf, pd, 0};
            J.n = 1; J.total = nfe + nde;
            dr_wsplit_kernel<0><<<(unsigned)((nfe + nde + 255) / 256), 256>>>(J);
        }
        CK(hipDeviceSynchronize());
    }
    if constexpr (BP) kfn = (const void*)gemm_dr3_kernel<TM, TN, A_RC, true, CS, EPI, true>;
    else if constexpr (P3) kfn = (const void*)gemm_dr3_kernel<TM, TN, A_RC, B_RC, CS, EPI>;
    else kfn = (const void*)gemm_dr_kernel<TM, TN, A_RC, B_RC, CS, EPI>;
    CK(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    DrEpilogue ep{};
    ep.bias = bias; ep.relu = 1; ep.keep = 1.f;
    ep.split_stride = (int64_t)M * N;
    ep.colsum = CS ? cs : nullptr; ep.colsum_stride = N;
    const int kr = P3 ? 32 : 16;
    const int kchunk = ((K + splits - 1) / splits + kr - 1) / kr * kr;
    auto launch = [&]() {
        // fwd (B = W [K][N]): the k-blocked form of W, plane rows of N columns; dgrad (B stored [N][K], reduction over its columns): the
        // column-blocked form, plane rows of N entries
        if constexpr (BP) gemm_dr3_kernel<TM, TN, A_RC, true, CS, EPI, true><<<dim3(nbm * nbn, splits), 256, lds, 0>>>(A, lda, (const float*)(B_RC ? pd : pf), N, C, N, M, N, K, kchunk, nbn, ep, (B_RC ? nde : nfe) * 16, 0);
        else if constexpr (P3) gemm_dr3_kernel<TM, TN, A_RC, B_RC, CS, EPI><<<dim3(nbm * nbn, splits), 256, lds, 0>>>(A, lda, B, ldb, C, N, M, N, K, kchunk, nbn, ep, 0, 0);
        else gemm_dr_kernel<TM, TN, A_RC, B_RC, CS, EPI><<<dim3(nbm * nbn, splits), 256, lds, 0>>>(A, lda, B, ldb, C, N, M, N, K, kchunk, nbn, ep, DrOuter{});
    };
    launch();
    CK(hipDeviceSynchronize());
    const bool plain = EPI == DR_STORE;
    ref_gemm<<<dim3((N + 255) / 256, M), 256>>>(A, lda, A_RC, B, ldb, B_RC, R, M, N, K, plain ? nullptr : bias, plain ? 0 : 1);
    CK(hipDeviceSynchronize());
    std::vector<float> hC((size_t)splits * M * N), hR((size_t)M * N), hcs((size_t)splits * N);
    CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hR.data(), R, hR.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hcs.data(), cs, hcs.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, worst_cs = 0;
    for (size_t i = 0; i < hR.size(); ++i) {
        double s = 0;
        for (int z = 0; z < splits; ++z) s += hC[(size_t)z * M * N + i];
        worst = fmax(worst, fabs(s - hR[i]));
    }
    if (CS) {
        for (int n = 0; n < N; ++n) {
            double s = 0, r = 0;
            for (int z = 0; z < splits; ++z) s += hcs[(size_t)z * N + n];
            for (int kk = 0; kk < K; ++kk) r += hB[(size_t)kk * N + n];
            worst_cs = fmax(worst_cs, fabs(s - r));
        }
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 10; ++i) launch();
    CK(hipEventRecord(e0));
    const int it = 100;
    for (int i = 0; i < it; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / it;
    printf("%s%-14s M=%d N=%d K=%d  TM=%d TN=%d splits=%d blocks=%d lds=%zuKB  %.2f us  %.1f TF  (%.3f of 157.3)  maxerr %.2e colsum err %.2e\n", BP ? "[bf16x3 W pre-split] " : P3 ? "[bf16x3] " : "[exact]  ", name, M, N, K, TM, TN, splits,
           nbm * nbn * splits, lds / 1024, us, 2.0 * M * N * K / us / 1e6, 2.0 * M * N * K / us / 1e6 / 157.3, worst, worst_cs);
    {
        const int nw = std::min(nbm * nbn * splits, 1024) * 4;
        std::vector<long long> hs(nw * 8);
        CK(hipMemcpyFromSymbol(hs.data(), HIP_SYMBOL(dr_stamps), hs.size() * 8));
        double avg[6] = {0}, mx[6] = {0};
        long long t0 = hs[0], t1 = hs[5];
        for (int b = 0; b < nw; ++b) {
            if (hs[b * 8 + 5] < hs[b * 8]) continue;
            for (int z = 1; z < 6; ++z) { double d = (double)(hs[b * 8 + z] - hs[b * 8 + z - 1]); avg[z] += d / nw; mx[z] = fmax(mx[z], d); }
            t0 = std::min(t0, hs[b * 8]); t1 = std::max(t1, hs[b * 8 + 5]);
        }
        printf("      cycles avg (max): setup+loads-issued %.0f (%.0f)  tail+main %.0f (%.0f)  reduce+stage %.0f (%.0f)  barrier %.0f (%.0f)  stores %.0f (%.0f)  | first start..last end %lld\n",
               avg[1], mx[1], avg[2], mx[2], avg[3], mx[3], avg[4], mx[4], avg[5], mx[5], t1 - t0);
    }
    hipFree(A); hipFree(B); hipFree(C); hipFree(R); hipFree(bias); hipFree(cs);
}

int main(int argc, char** argv) {
    // weight-gradient tiles in split precision (both operands split in registers): VALU ops per MFMA fall with squarer tiles
    run<4, 7, false, false, true, true>("wgrad L0", 624, 400, 4096, 6);
    run<5, 7, false, false, true, true>("wgrad L0", 624, 400, 4096, 8);
    run<4, 7, false, false, true, true>("wgrad L1", 400, 400, 4096, 9);
    run<5, 7, false, false, true, true>("wgrad L1", 400, 400, 4096, 12);
    run<5, 5, false, false, true, true>("wgrad L1", 400, 400, 4096, 10);
    run<5, 5, false, false, true, true>("wgrad L0", 624, 400, 4096, 6);
    return 0;
}
