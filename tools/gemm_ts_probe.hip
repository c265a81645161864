// Round 6 probe: the tall split-precision products (csrc/gemm_ts.h) alone on an idle chip, at AFM's reference point (3.0 M pair rows, K = A = 256)
// and on a ragged row count; results against an fp64 product on sampled rows.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I include -o tools/_bin/gemm_ts_probe tools/gemm_ts_probe.hip && tools/_bin/gemm_ts_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

__global__ void pair_kernel(const float* e, int e_ld, const int16_t* pi, const int16_t* pj, int P, int K, int64_t M, float* X) {
    for (int64_t idx = blockIdx.x * 256ll + threadIdx.x; idx < M * K; idx += gridDim.x * 256ll) {
        const int64_t r = idx / K; const int k = (int)(idx - r * K);
        const int64_t b = r / P; const int p = (int)(r - b * P);
        X[idx] = e[b * e_ld + pi[p] * K + k] * e[b * e_ld + pj[p] * K + k];
    }
}
static void diff_show(const float* a, const float* b, size_t n, int ld) {
    std::vector<float> x(n), y(n);
    CK(hipMemcpy(x.data(), a, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), b, n * 4, hipMemcpyDeviceToHost));
    int shown = 0; long long last_row = -1;
    for (size_t i = 0; i < n && shown < 12; ++i)
        if (memcmp(&x[i], &y[i], 4) != 0) {
            const long long row = i / ld; const int col = (int)(i % ld);
            if (row == last_row) continue;
            last_row = row;
            int ncol = 0, c0 = -1, c1 = -1;
            for (int cc = 0; cc < ld; ++cc) if (memcmp(&x[row * ld + cc], &y[row * ld + cc], 4) != 0) { if (c0 < 0) c0 = cc; c1 = cc; ++ncol; }
            printf("    row %lld (tile row %lld, in-tile %lld): %d differing columns in [%d, %d]; first: ref %.6g got %.6g (ratio %.4f)\n", row, row / 256, row % 256, ncol, c0, c1, x[i], y[i], y[i] / x[i]);
            (void)col; ++shown;
        }
}
static size_t diff_words(const void* a, const void* b, size_t n) {
    std::vector<uint32_t> x(n), y(n);
    CK(hipMemcpy(x.data(), a, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), b, n * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i) bad += x[i] != y[i];
    return bad;
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
    // X = the pair products of F = 39 gathered embeddings per example (AFM.py:130-139), so that the generated-operand kernels can be compared
    const int F = 39, P = F * (F - 1) / 2, e_ld = F * R;
    const int64_t nex = (M + P - 1) / P;
    float* E_; int16_t *pi, *pj;
    CK(hipMalloc(&E_, nex * e_ld * 4)); CK(hipMalloc(&pi, P * 2)); CK(hipMalloc(&pj, P * 2));
    {
        std::vector<int16_t> hi, hj;
        for (int i = 0; i < F - 1; ++i) for (int j = i + 1; j < F; ++j) { hi.push_back((int16_t)i); hj.push_back((int16_t)j); }
        CK(hipMemcpy(pi, hi.data(), P * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(pj, hj.data(), P * 2, hipMemcpyHostToDevice));
    }
    fill_kernel<<<4096, 256>>>(E_, nex * e_ld, 1, 1.f, 0);
    pair_kernel<<<8192, 256>>>(E_, e_ld, pi, pj, P, R, M, X);
    fill_kernel<<<64, 256>>>(W, R * N, 2, 0.08f, 0);
    fill_kernel<<<1, 256>>>(b, N, 3, 0.1f, 0);
    fill_kernel<<<1, 256>>>(wo, N, 4, 0.3f, 0);
    fill_kernel<<<1024, 256>>>(rsc, M, 5, 1.f, 0);
    CK(hipMemset(Y + M * N, 0xff, guard * 4)); CK(hipMemset(DX + M * R, 0xff, guard * 4)); CK(hipMemset(dot + M, 0xff, guard * 4));
    auto kf = gemm_ts_kernel<KG, NT, TS_FWD>;
    auto kg = gemm_ts_kernel<NT / 2, 2 * KG, TS_GATE>;            // the input gradient: reduction over the N outputs, R columns out
    const int ldsf = 3 * 3 * 4 * N * 16, ldsg = 3 * 3 * 4 * R * 16;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, ldsf));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kg), hipFuncAttributeMaxDynamicSharedMemorySize, ldsg));
    const int grid = (int)std::min<int64_t>((M + 255) / 256, 256);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // ---- forward
    unsigned long long* SB; CK(hipMalloc(&SB, (M + 4096) * 32)); CK(hipMemset(SB + M * 4, 0xff, 4096 * 32));
    TsArgs af{}; af.A = X; af.lda = R; af.planes = pf; af.C = Y; af.ldc = N; af.M = M; af.bias = b; af.dot_w = wo; af.dot_out = dot; af.bits_out = SB;
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
    {   // the same product with its rows generated from the embeddings: bit-identical output
        float *Y2, *dot2;
        CK(hipMalloc(&Y2, (M * N) * 4)); CK(hipMalloc(&dot2, M * 4));
        TsArgs ag2 = af; ag2.C = Y2; ag2.dot_out = dot2; ag2.e = E_; ag2.e_ld = e_ld; ag2.e_floats = nex * e_ld; ag2.pair_i = pi; ag2.pair_j = pj; ag2.P = P;
        auto kfg = gemm_ts_kernel<KG, NT, TS_FWD, true>;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfg), hipFuncAttributeMaxDynamicSharedMemorySize, ldsf));
        best = 1e9f;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0));
            kfg<<<grid, 256, ldsf>>>(ag2);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        CK(hipGetLastError());
        printf("forward, rows generated from the embeddings: %.3f ms  %.1f TF-equivalent; words differing from the product over the stored rows: %zu (+ %zu of the scores)\n",
               best, gf / best, M <= 200000 ? diff_words(Y, Y2, M * N) : diff_words(Y, Y2, 1 << 24), diff_words(dot, dot2, M));
        CK(hipFree(Y2)); CK(hipFree(dot2));
    }
    {   // two column halves, four waves of 64 rows, two blocks per CU
        float *Y2, *dot2; unsigned long long* SB2;
        CK(hipMalloc(&Y2, (M * N) * 4)); CK(hipMalloc(&dot2, 2 * M * 4)); CK(hipMalloc(&SB2, M * 32)); CK(hipMemset(SB2, 0, M * 32));
        TsArgs ah2 = af; ah2.C = Y2; ah2.dot_out = dot2; ah2.dot_stride = M; ah2.bits_out = SB2; ah2.e = E_; ah2.e_ld = e_ld; ah2.e_floats = nex * e_ld; ah2.pair_i = pi; ah2.pair_j = pj; ah2.P = P;
        auto kh = gemm_ts_kernel<KG, NT / 2, TS_FWD, false, 4, NT>;
        auto khg = gemm_ts_kernel<KG, NT / 2, TS_FWD, true, 4, NT>;
        const int ldsh = 3 * 3 * 4 * (N / 2) * 16;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kh), hipFuncAttributeMaxDynamicSharedMemorySize, ldsh));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(khg), hipFuncAttributeMaxDynamicSharedMemorySize, ldsh));
        const int gridh = (int)std::min<int64_t>((M + 255) / 256 * 2, 512);
        for (int gen = 0; gen < 2; ++gen) {
            best = 1e9f;
            for (int r = 0; r < reps; ++r) {
                CK(hipEventRecord(e0));
                if (gen) khg<<<gridh, 256, ldsh>>>(ah2); else kh<<<gridh, 256, ldsh>>>(ah2);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms);
            }
            CK(hipGetLastError());
            printf("forward, column halves at two blocks per CU%s: %.3f ms  %.1f TF-equivalent; words differing: %zu of the output, %zu of the sign words\n", gen ? ", rows generated" : "", best, gf / best,
                   M <= 200000 ? diff_words(Y, Y2, M * N) : diff_words(Y, Y2, 1 << 24), diff_words(SB, SB2, M * 8));
        }
        CK(hipFree(Y2)); CK(hipFree(dot2)); CK(hipFree(SB2));
    }
    {   // eight waves of 32 rows (two per SIMD)
        float *Y2, *dot2;
        CK(hipMalloc(&Y2, (M * N) * 4)); CK(hipMalloc(&dot2, M * 4));
        TsArgs a8 = af; a8.C = Y2; a8.dot_out = dot2; a8.e = E_; a8.e_ld = e_ld; a8.e_floats = nex * e_ld; a8.pair_i = pi; a8.pair_j = pj; a8.P = P;
        auto k8 = gemm_ts_kernel<KG, NT, TS_FWD, false, 2>;
        auto k8g = gemm_ts_kernel<KG, NT, TS_FWD, true, 2>;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k8), hipFuncAttributeMaxDynamicSharedMemorySize, ldsf));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k8g), hipFuncAttributeMaxDynamicSharedMemorySize, ldsf));
        for (int gen = 0; gen < 2; ++gen) {
            best = 1e9f;
            for (int r = 0; r < reps; ++r) {
                CK(hipEventRecord(e0));
                if (gen) k8g<<<grid, 512, ldsf>>>(a8); else k8<<<grid, 512, ldsf>>>(a8);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms);
            }
            CK(hipGetLastError());
            printf("forward, 8 waves%s: %.3f ms  %.1f TF-equivalent; words differing: %zu (+ %zu of the scores)\n", gen ? ", rows generated" : "", best, gf / best,
                   M <= 200000 ? diff_words(Y, Y2, M * N) : diff_words(Y, Y2, 1 << 24), diff_words(dot, dot2, M));
        }
        CK(hipFree(Y2)); CK(hipFree(dot2));
    }
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
    {
        auto kg8 = gemm_ts_kernel<NT / 2, 2 * KG, TS_GATE, false, 2>;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kg8), hipFuncAttributeMaxDynamicSharedMemorySize, ldsg));
        float b8 = 1e9f;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0));
            kg8<<<grid, 512, ldsg>>>(ag);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            b8 = std::min(b8, ms);
        }
        printf("gate dgrad, 8 waves: %.3f ms  %.1f TF-equivalent\n", b8, 2.0 * M * R * N * 1e-9 / b8);
        kg<<<grid, 256, ldsg>>>(ag);                    // (DX holds the 4-wave kernel's result from here on)
        CK(hipDeviceSynchronize());
    }
    CK(hipGetLastError());
    printf("gate dgrad (3 products): %.3f ms  %.1f TF-equivalent  %.2f TB/s algorithmic\n", best, gf / best, (double)M * (R + N) * 4 / best * 1e-9);
    {   // the same gradient from the forward's sign bits (32 bytes per row instead of the row)
        float* DX2; CK(hipMalloc(&DX2, M * R * 4));
        TsArgs a2 = ag; a2.C = DX2; a2.bits_in = SB; a2.A = nullptr;
        auto kgb = gemm_ts_kernel<NT / 2, 2 * KG, TS_GATE, true, 4>;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kgb), hipFuncAttributeMaxDynamicSharedMemorySize, ldsg));
        float bb = 1e9f;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0));
            kgb<<<grid, 256, ldsg>>>(a2);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            bb = std::min(bb, ms);
        }
        CK(hipGetLastError());
        std::vector<uint32_t> gd2(4096 * 8);
        CK(hipMemcpy(gd2.data(), SB + M * 4, 4096 * 32, hipMemcpyDeviceToHost));
        size_t badb = 0; for (auto v : gd2) badb += v != 0xffffffffu;
        printf("gate dgrad from the sign bits: %.3f ms  %.1f TF-equivalent  %.2f TB/s algorithmic; words differing from the gradient read from the rows: %zu; sign words written past the end: %zu\n",
               bb, gf / bb, ((double)M * R * 4 + M * 32.0) / bb * 1e-9, diff_words(DX, DX2, M <= 200000 ? M * R : (1 << 24)), badb);
        {   // ... in two column halves, two blocks per CU
            auto kgh = gemm_ts_kernel<NT / 2, KG, TS_GATE, true, 4, 2 * KG>;
            const int ldsh = 3 * 3 * 4 * (R / 2) * 16;
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kgh), hipFuncAttributeMaxDynamicSharedMemorySize, ldsh));
            CK(hipMemset(DX2, 0, M * R * 4));
            const int gridh = (int)std::min<int64_t>((M + 255) / 256 * 2, 512);
            float bh2 = 1e9f;
            for (int r = 0; r < reps; ++r) {
                CK(hipEventRecord(e0));
                kgh<<<gridh, 256, ldsh>>>(a2);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                bh2 = std::min(bh2, ms);
            }
            CK(hipGetLastError());
            size_t dd = 0;
            for (int r = 0; r < 3; ++r) { kgh<<<gridh, 256, ldsh>>>(a2); CK(hipDeviceSynchronize()); dd += diff_words(DX, DX2, M <= 200000 ? M * R : (1 << 24)); }
            printf("gate dgrad from the sign bits, column halves, two blocks per CU: %.3f ms  %.1f TF-equivalent; words differing (3 more runs): %zu\n", bh2, gf / bh2, dd);
            if (dd) diff_show(DX, DX2, M <= 200000 ? M * R : (1 << 24), R);
        }
        CK(hipFree(DX2));
    }
    {   // run to run: the 4-wave kernel against itself, and the 8-wave form (the library's for stored rows) against it
        float* DX2; CK(hipMalloc(&DX2, M * R * 4));
        TsArgs a2 = ag; a2.C = DX2;
        size_t d4 = 0, d8 = 0;
        const size_t nw = M <= 200000 ? M * R : (1 << 24);
        for (int r = 0; r < 4; ++r) {
            kg<<<grid, 256, ldsg>>>(a2);
            CK(hipDeviceSynchronize());
            d4 += diff_words(DX, DX2, nw);
        }
        for (int r = 0; r < 4; ++r) {
            gemm_ts_kernel<NT / 2, 2 * KG, TS_GATE, false, 2><<<grid, 512, ldsg>>>(a2);
            CK(hipDeviceSynchronize());
            d8 += diff_words(DX, DX2, nw);
        }
        printf("gate dgrad, 4 more runs each: words differing from the first run's, 4 waves: %zu; 8 waves: %zu of 4 x %zu\n", d4, d8, nw);
        CK(hipFree(DX2));
    }
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
    // ---- gated weight gradient: dW[k, a] = wo[a] sum_r (rsc[r] X[r, k]) 1[Y[r, a] > 0], partial slabs per block
    {
        constexpr int NW = (R == 256 && N == 256) ? 8 : 4, TK = R / (16 * NW), TA = N / (16 * NW);
        const int grid_w = 256;
        const int rpb = (int)(((M + grid_w - 1) / grid_w + 31) / 32 * 32);
        float *dw, *db, *dwo;
        CK(hipMalloc(&dw, (size_t)grid_w * R * N * 4)); CK(hipMalloc(&db, grid_w * N * 4)); CK(hipMalloc(&dwo, grid_w * N * 4));
        TswArgs aw{}; aw.X = X; aw.ldx = R; aw.H = Y; aw.ldh = N; aw.rowscale = rsc; aw.colscale = wo; aw.dw = dw; aw.dw_stride = R * N; aw.db = db; aw.db_stride = N;
        aw.dwo = dwo; aw.dwo_stride = N; aw.M = M; aw.rows_per_block = rpb;
        auto kw = gemm_tsw_kernel<NW, TK, TA>;
        const int ldsw = 2 * NW * TA * 1024;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kw), hipFuncAttributeMaxDynamicSharedMemorySize, ldsw));
        best = 1e9f;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0));
            kw<<<grid_w, 64 * NW, ldsw>>>(aw);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        CK(hipGetLastError());
        printf("gate wgrad (3 products): %.3f ms  %.1f TF-equivalent  %.2f TB/s algorithmic (%d rows per block)\n", best, gf / best, (double)M * (R + N) * 4 / best * 1e-9, rpb);
        if (M <= 200000) {
            std::vector<float> hx((size_t)M * R), hy((size_t)M * N), hr(M), hdw((size_t)grid_w * R * N), hdb(grid_w * N), hdwo(grid_w * N);
            CK(hipMemcpy(hx.data(), X, hx.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hy.data(), Y, hy.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hr.data(), rsc, M * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hdw.data(), dw, hdw.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hdb.data(), db, hdb.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hdwo.data(), dwo, hdwo.size() * 4, hipMemcpyDeviceToHost));
            double ew = 0, sw = 0, eb = 0, sb = 0, eo = 0, so = 0;
            for (int kk = 0; kk < 24; ++kk) {
                const int k = (kk * 37 + 5) % R;
                for (int aa = 0; aa < 24; ++aa) {
                    const int an = (aa * 29 + 3) % N;
                    double v = 0;
                    for (int64_t r = 0; r < M; ++r) if (hy[r * N + an] > 0.f) v += (double)(hr[r] * hx[r * R + k]);
                    v *= hwo[an];
                    double got = 0;
                    for (int b2 = 0; b2 < grid_w; ++b2) got += hdw[(size_t)b2 * R * N + (size_t)k * N + an];
                    ew = std::max(ew, std::fabs(v - got)); sw = std::max(sw, std::fabs(v));
                }
            }
            for (int an = 0; an < N; ++an) {
                double v1 = 0, v2 = 0, g1 = 0, g2 = 0;
                for (int64_t r = 0; r < M; ++r) { if (hy[r * N + an] > 0.f) v1 += hr[r]; v2 += (double)hr[r] * hy[r * N + an]; }
                v1 *= hwo[an];
                for (int b2 = 0; b2 < grid_w; ++b2) { g1 += hdb[b2 * N + an]; g2 += hdwo[b2 * N + an]; }
                eb = std::max(eb, std::fabs(v1 - g1)); sb = std::max(sb, std::fabs(v1));
                eo = std::max(eo, std::fabs(v2 - g2)); so = std::max(so, std::fabs(v2));
            }
            printf("  max |err| gate wgrad %.3e (values to %.2f), bias grad %.3e (to %.2f), second column sums %.3e (to %.2f)\n", ew, sw, eb, sb, eo, so);
        }
        {   // generated X
            float *dw2, *db2, *dwo2;
            CK(hipMalloc(&dw2, (size_t)grid_w * R * N * 4)); CK(hipMalloc(&db2, grid_w * N * 4)); CK(hipMalloc(&dwo2, grid_w * N * 4));
            TswArgs a2 = aw; a2.dw = dw2; a2.db = db2; a2.dwo = dwo2; a2.e = E_; a2.e_ld = e_ld; a2.e_floats = nex * e_ld; a2.pair_i = pi; a2.pair_j = pj; a2.P = P;
            auto kwg = gemm_tsw_kernel<NW, TK, TA, true>;
            const int ldsg2 = ldsw + P * 4;
            best = 1e9f;
            for (int r = 0; r < reps; ++r) {
                CK(hipEventRecord(e0));
                kwg<<<grid_w, 64 * NW, ldsg2>>>(a2);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms);
            }
            CK(hipGetLastError());
            printf("gate wgrad, X generated from the embeddings: %.3f ms  %.1f TF-equivalent; words differing: %zu of dW, %zu of db, %zu of dwo\n", best, gf / best,
                   diff_words(dw, dw2, (size_t)grid_w * R * N), diff_words(db, db2, grid_w * N), diff_words(dwo, dwo2, grid_w * N));
            CK(hipFree(dw2)); CK(hipFree(db2)); CK(hipFree(dwo2));
        }
        {   // gate from the forward's sign words, second sums from the product itself; X generated
            float *dw2, *db2, *dwo2;
            CK(hipMalloc(&dw2, (size_t)grid_w * R * N * 4)); CK(hipMalloc(&db2, grid_w * N * 4)); CK(hipMalloc(&dwo2, grid_w * N * 4));
            TswArgs a2 = aw; a2.dw = dw2; a2.db = db2; a2.dwo = dwo2; a2.e = E_; a2.e_ld = e_ld; a2.e_floats = nex * e_ld; a2.pair_i = pi; a2.pair_j = pj; a2.P = P;
            a2.bits = SB; a2.W = W; a2.bias = b; a2.H = nullptr;
            auto kwb = gemm_tsw_kernel<NW, TK, TA, true, true>;
            const int ldsg2 = ldsw + P * 4;
            best = 1e9f;
            for (int r = 0; r < reps; ++r) {
                CK(hipEventRecord(e0));
                kwb<<<grid_w, 64 * NW, ldsg2>>>(a2);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms);
            }
            CK(hipGetLastError());
            // second sums: compare the slab totals with the H-reading kernel's
            std::vector<float> o1(grid_w * N), o2(grid_w * N);
            CK(hipMemcpy(o1.data(), dwo, o1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(o2.data(), dwo2, o2.size() * 4, hipMemcpyDeviceToHost));
            double em = 0, sm = 0;
            for (int an = 0; an < N; ++an) {
                double t1 = 0, t2 = 0;
                for (int b2 = 0; b2 < grid_w; ++b2) { t1 += o1[b2 * N + an]; t2 += o2[b2 * N + an]; }
                em = std::max(em, std::fabs(t1 - t2)); sm = std::max(sm, std::fabs(t1));
            }
            printf("gate wgrad from the sign bits, X generated: %.3f ms  %.1f TF-equivalent; words differing: %zu of dW, %zu of db; second column sums (W . dW + b db) against the sums over H: max |diff| %.3e (values to %.2f)\n",
                   best, gf / best, diff_words(dw, dw2, (size_t)grid_w * R * N), diff_words(db, db2, grid_w * N), em, sm);
            CK(hipFree(dw2)); CK(hipFree(db2)); CK(hipFree(dwo2));
        }
        CK(hipFree(dw)); CK(hipFree(db)); CK(hipFree(dwo));
    }
    CK(hipFree(X)); CK(hipFree(Y)); CK(hipFree(DX)); CK(hipFree(W)); CK(hipFree(b)); CK(hipFree(wo)); CK(hipFree(rsc)); CK(hipFree(dot)); CK(hipFree(SB)); CK(hipFree(pf)); CK(hipFree(pg)); CK(hipFree(E_)); CK(hipFree(pi)); CK(hipFree(pj));
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    run<8, 16>(100037, 3);
    run<8, 16>(4096ll * 741, 5);
    run<8, 8>(128ll * 741, 5);              // test_afm_grad's point: K = 256, A = 128
    return 0;
}
