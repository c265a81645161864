// Round 6 probe: the background sweep of the time-blocked table step (csrc/lag.hip: lag_advance_kernel<KQ, false, UNR>) ALONE on an idle
// chip, by grid size (blocks per CU) and rows in flight per lane -- what bounds its 21-24 us in the step (52 MB of traffic, 17 M
// element-updates: HBM floor ~9 us at the copy rate, ALU floor ~5.5 us).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -o tools/_bin/lag_probe tools/lag_probe.hip && tools/_bin/lag_probe
#include <cstdarg>
#include <cstdio>
#include <vector>
#include "../tf_repos_amd/csrc/lag.hip"

namespace dctr { void set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vprintf(fmt, a); va_end(a); printf("\n"); } }
using namespace dctr;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int UNR, bool PIPE = false>
static void run(int64_t V, int K, int period, int bpc, float4* emb, float4* s0, float4* s1, float* lin, float* l0, float* l1, int32_t* slot, uint8_t* ts, StepState* S,
                StepState& hs) {
    const int KQ = K / 4;
    const int64_t span = (V + period - 1) / period;
    const int grid = (int)std::min<int64_t>((span * KQ + 256 * UNR - 1) / (256 * UNR), 256 * bpc);
    // stamps: block b of the table was last visited at step b (so that at T = period + b it is `period` steps behind), T runs period .. 3 period
    std::vector<uint8_t> hts(V);
    for (int64_t r = 0; r < V; ++r) hts[r] = (uint8_t)(r / span);
    CK(hipMemcpy(ts, hts.data(), V, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double tot = 0; int n = 0;
    for (int64_t T = period; T < 5 * period; ++T) {
        hs.t = T;
        CK(hipMemcpy(S, &hs, sizeof(hs), hipMemcpyHostToDevice));
        CK(hipEventRecord(e0));
        if (PIPE) lag_sweep_pipe_kernel<4, UNR><<<grid, 256>>>(V, emb, s0, s1, lin, l0, l1, slot, ts, S, 1e-4f, period, KQ, 1);
        else lag_advance_kernel<4, false, UNR><<<grid, 256>>>(V, emb, s0, s1, lin, l0, l1, slot, ts, S, 1e-4f, period, 0, nullptr, nullptr, KQ, 1);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (T >= 2 * period) { tot += ms; ++n; }
    }
    const double us = tot * 1e3 / n;
    const double bytes = (6.0 * (K + 1) * 4 + 5) * span;
    printf("%s UNR=%d blocks/CU<=%d grid=%d: %.2f us per launch (each swept row replays %d steps), %.1f MB -> %.2f TB/s\n", PIPE ? "pipelined" : "one trip ", UNR, bpc, grid, us, period, bytes / 1e6, bytes / us / 1e6);
}
static void same(const void* a, const void* b, size_t n, const char* what) {
    std::vector<char> x(n), y(n);
    CK(hipMemcpy(x.data(), a, n, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), b, n, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i) bad += x[i] != y[i];
    printf("  %s: %zu differing bytes of %zu\n", what, bad, n);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int64_t V = 1000000; const int K = 16, period = 8;
    float4 *emb, *s0, *s1; float *lin, *l0, *l1; int32_t* slot; uint8_t* ts; StepState* S;
    CK(hipMalloc(&emb, V * K * 4)); CK(hipMalloc(&s0, V * K * 4)); CK(hipMalloc(&s1, V * K * 4));
    CK(hipMalloc(&lin, V * 4)); CK(hipMalloc(&l0, V * 4)); CK(hipMalloc(&l1, V * 4)); CK(hipMalloc(&slot, V * 4)); CK(hipMalloc(&ts, V)); CK(hipMalloc(&S, sizeof(StepState)));
    std::vector<float> h(V * K);
    for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 0.02f;
    CK(hipMemcpy(emb, h.data(), V * K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(lin, h.data(), V * 4, hipMemcpyHostToDevice));
    for (auto& v : h) v = fabsf(v) * 1e-3f;
    CK(hipMemcpy(s0, h.data(), V * K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(s1, h.data(), V * K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(l0, h.data(), V * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(l1, h.data(), V * 4, hipMemcpyHostToDevice));
    CK(hipMemset(slot, 0, V * 4));
    StepState hs{};
    hs.hyper.lr = 5e-4f; hs.hyper.beta1 = 0.9f; hs.hyper.beta2 = 0.999f; hs.hyper.eps = 1e-8f; hs.hyper.lr_t = 5e-4f;
    for (int i = 0; i < LR_HIST; ++i) hs.lr_hist[i] = 5e-4f * (1.f + 0.01f * i);
    // a second, identical table: the pipelined kernel must leave the bytes the one-trip kernel leaves (some rows "touched": slot != 0)
    float4 *emb2, *s02, *s12; float *lin2, *l02, *l12; uint8_t* ts2;
    CK(hipMalloc(&emb2, V * K * 4)); CK(hipMalloc(&s02, V * K * 4)); CK(hipMalloc(&s12, V * K * 4));
    CK(hipMalloc(&lin2, V * 4)); CK(hipMalloc(&l02, V * 4)); CK(hipMalloc(&l12, V * 4)); CK(hipMalloc(&ts2, V));
    {
        std::vector<int32_t> hsl(V);
        for (int64_t r = 0; r < V; ++r) hsl[r] = (r * 2654435761u >> 28) == 0 ? 1 : 0;        // ~6 % of the rows
        CK(hipMemcpy(slot, hsl.data(), V * 4, hipMemcpyHostToDevice));
    }
    CK(hipMemcpy(emb2, emb, V * K * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(s02, s0, V * K * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(s12, s1, V * K * 4, hipMemcpyDeviceToDevice));
    CK(hipMemcpy(lin2, lin, V * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(l02, l0, V * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(l12, l1, V * 4, hipMemcpyDeviceToDevice));
    run<4>(V, K, period, 2, emb, s0, s1, lin, l0, l1, slot, ts, S, hs);
    run<2, true>(V, K, period, 1, emb2, s02, s12, lin2, l02, l12, slot, ts2, S, hs);
    same(emb, emb2, V * K * 4, "emb"); same(s0, s02, V * K * 4, "m"); same(s1, s12, V * K * 4, "v"); same(lin, lin2, V * 4, "linear"); same(l0, l02, V * 4, "linear m"); same(ts, ts2, V, "stamps");
    {   // a checksum of the one-trip kernel's table, to compare two builds of this probe (-DDCTR_LAG_SELECT: the round-5 replay loop)
        std::vector<uint32_t> x(V * K);
        CK(hipMemcpy(x.data(), emb, V * K * 4, hipMemcpyDeviceToHost));
        uint64_t hsh = 1469598103934665603ull;
        for (uint32_t w : x) hsh = (hsh ^ w) * 1099511628211ull;
        CK(hipMemcpy(x.data(), s0, V * K * 4, hipMemcpyDeviceToHost));
        for (uint32_t w : x) hsh = (hsh ^ w) * 1099511628211ull;
        CK(hipMemcpy(x.data(), s1, V * K * 4, hipMemcpyDeviceToHost));
        for (uint32_t w : x) hsh = (hsh ^ w) * 1099511628211ull;
        printf("  checksum of (emb, m, v) after the one-trip kernel's 32 sweeps: %016llx\n", (unsigned long long)hsh);
    }
    CK(hipMemset(slot, 0, V * 4));
    for (int bpc : {1, 2, 4}) {
        run<4>(V, K, period, bpc, emb, s0, s1, lin, l0, l1, slot, ts, S, hs);
        run<4, true>(V, K, period, bpc, emb, s0, s1, lin, l0, l1, slot, ts, S, hs);
        run<2, true>(V, K, period, bpc, emb, s0, s1, lin, l0, l1, slot, ts, S, hs);
        run<1, true>(V, K, period, bpc, emb, s0, s1, lin, l0, l1, slot, ts, S, hs);
    }
    return 0;
}
