// Round 6 probe: the tall split-precision products (csrc/gemm_ts.h) alone on an idle chip, at AFM's reference point (3.0 M pair rows, K = A = 256)
// and on a ragged row count; results against an fp64 product on sampled rows.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I include -o tools/_bin/gemm_ts_probe tools/gemm_ts_probe.hip && tools/_bin/gemm_ts_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../tf_repos_amd/csrc/gemm_ts.h"

namespace dctr { __device__ __forceinline__ float dr_dropout_scale(uint64_t, uint64_t, float) { return 1.f; } }
using namespace dctr;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__host__ __device__ inline float hval(uint64_t i, uint32_t seed) {
    uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)((z >> 40) * (1.0 / 16777216.0)) * 2.f - 1.f;
}
__global__ void fill_kernel(float* p, int64_t n, uint32_t seed, float scale, int relu) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) {
        float v = hval(i, seed) * scale;
        p[i] = relu ? fmaxf(v, 0.f) : v;
    }
}

template <int KG, int NT>
static void run(int64_t M, int reps) {
    constexpr int R = 32 * KG, N = 16 * NT;
    printf("# M = %lld rows, reduction %d, %d output columns\n", (long long)M, R, N);
    float *X, *Y, *W, *b, *wo, *rsc, *dot, *DX;
    const size_t guard = 4096;
    CK(hipMalloc(&X, (M * R + guard) * 4)); CK(hipMalloc(&Y, (M * N + guard) * 4)); CK(hipMalloc(&DX, (M * R + guard) * 4));
    CK(hipMalloc(&W, R * N * 4)); CK(hipMalloc(&b, N * 4)); CK(hipMalloc(&wo, N * 4)); CK(hipMalloc(&rsc, (M + guard) * 4)); CK(hipMalloc(&dot, (M + guard) * 4));
    u32x4 *pf, *pg;
    CK(hipMalloc(&pf, 3 * R * N * 2)); CK(hipMalloc(&pg, 3 * R * N * 2));
    fill_kernel<<<4096, 256>>>(X, M * R, 1, 1.f, 0);
    fill_kernel<<<64, 256>>>(W, R * N, 2, 0.08f, 0);
    fill_kernel<<<1, 256>>>(b, N, 3, 0.1f, 0);
    fill_kernel<<<1, 256>>>(wo, N, 4, 0.3f, 0);
    fill_kernel<<<1024, 256>>>(rsc, M, 5, 1.f, 0);
    CK(hipMemset(Y + M * N, 0xff, guard * 4)); CK(hipMemset(DX + M * R, 0xff, guard * 4)); CK(hipMemset(dot + M, 0xff, guard * 4));
    auto kf = gemm_ts_kernel<KG, NT, TS_FWD>;
    auto kg = gemm_ts_kernel<NT / 2, 2 * KG, TS_GATE>;            // the input gradient: reduction over the N outputs, R columns out
    const int ldsf = 2 * 3 * 4 * N * 16, ldsg = 2 * 3 * 4 * R * 16;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, ldsf));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kg), hipFuncAttributeMaxDynamicSharedMemorySize, ldsg));
    const int grid = (int)std::min<int64_t>((M + 255) / 256, 256);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // ---- forward
    TsArgs af{}; af.A = X; af.lda = R; af.planes = pf; af.C = Y; af.ldc = N; af.M = M; af.bias = b; af.dot_w = wo; af.dot_out = dot;
    {   // both plane sets in one launch, timed
        const TsSplitJob j0{0, nullptr, R, N, pf}, j1{1, wo, N, R, pg};
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0));
            ts_wsplit_kernel<<<dim3((R / 8 * N + 255) / 256, 2), 256>>>(W, N, j0, j1);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r == 2) printf("weight split (both plane sets, one launch): %.1f us\n", ms * 1e3);
        }
    }
    float best = 1e9f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        kf<<<grid, 256, ldsf>>>(af);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    CK(hipGetLastError());
    const double gf = 2.0 * M * R * N * 1e-9;
    printf("forward  (6 products): %.3f ms  %.1f TF-equivalent  %.2f TB/s algorithmic\n", best, gf / best, (double)M * (R + N) * 4 / best * 1e-9);
    // ---- gate
    TsArgs ag{}; ag.A = Y; ag.lda = N; ag.planes = pg; ag.C = DX; ag.ldc = R; ag.M = M; ag.rowscale = rsc;
    best = 1e9f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        kg<<<grid, 256, ldsg>>>(ag);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    CK(hipGetLastError());
    printf("gate dgrad (3 products): %.3f ms  %.1f TF-equivalent  %.2f TB/s algorithmic\n", best, gf / best, (double)M * (R + N) * 4 / best * 1e-9);
    // ---- check sampled rows (the last rows among them) against fp64
    std::vector<float> hW(R * N), hb(N), hwo(N);
    CK(hipMemcpy(hW.data(), W, R * N * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), b, N * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hwo.data(), wo, N * 4, hipMemcpyDeviceToHost));
    double ef = 0, ed = 0, eg = 0, sf = 0, sg = 0;
    std::vector<float> x(R), y(N), dx(R);
    for (int s = 0; s < 96; ++s) {
        const int64_t row = s < 32 ? M - 1 - s : (s < 64 ? s - 32 : (int64_t)((uint64_t)s * 0x9E3779B97F4A7C15ull % (uint64_t)M));
        CK(hipMemcpy(x.data(), X + row * R, R * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), Y + row * N, N * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(dx.data(), DX + row * R, R * 4, hipMemcpyDeviceToHost));
        float hd, hr; CK(hipMemcpy(&hd, dot + row, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hr, rsc + row, 4, hipMemcpyDeviceToHost));
        double d = 0;
        for (int n = 0; n < N; ++n) {
            double v = hb[n];
            for (int k = 0; k < R; ++k) v += (double)x[k] * hW[k * N + n];
            v = v > 0 ? v : 0;
            ef = std::max(ef, std::fabs(v - y[n])); sf = std::max(sf, std::fabs(v));
            d += (double)y[n] * hwo[n];
        }
        ed = std::max(ed, std::fabs(d - hd));
        for (int k = 0; k < R; ++k) {
            double v = 0;
            for (int n = 0; n < N; ++n) if (y[n] > 0.f) v += (double)(hW[k * N + n] * hwo[n]);
            v *= hr;
            eg = std::max(eg, std::fabs(v - dx[k])); sg = std::max(sg, std::fabs(v));
        }
    }
    std::vector<uint32_t> gd(guard);
    size_t bad = 0;
    CK(hipMemcpy(gd.data(), Y + M * N, guard * 4, hipMemcpyDeviceToHost)); for (auto v : gd) bad += v != 0xffffffffu;
    CK(hipMemcpy(gd.data(), DX + M * R, guard * 4, hipMemcpyDeviceToHost)); for (auto v : gd) bad += v != 0xffffffffu;
    CK(hipMemcpy(gd.data(), dot + M, guard * 4, hipMemcpyDeviceToHost)); for (auto v : gd) bad += v != 0xffffffffu;
    printf("  max |err| forward %.3e (values to %.2f), score dot %.3e, gate dgrad %.3e (values to %.2f); words written past the ends: %zu\n", ef, sf, ed, eg, sg, bad);
    CK(hipFree(X)); CK(hipFree(Y)); CK(hipFree(DX)); CK(hipFree(W)); CK(hipFree(b)); CK(hipFree(wo)); CK(hipFree(rsc)); CK(hipFree(dot)); CK(hipFree(pf)); CK(hipFree(pg));
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    run<8, 16>(100037, 3);
    run<8, 16>(4096ll * 741, 5);
    run<8, 8>(128ll * 741, 5);              // test_afm_grad's point: K = 256, A = 128
    return 0;
}
