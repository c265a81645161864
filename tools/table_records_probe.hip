// Table layouts under LAGGING ROWS (csrc/lag.h): what a per-row [theta | m | v] record would buy the three kernels that touch table rows
// in a training step.  Round-3 verdict, item 3: with rows allowed to lag, the gather and the touched-rows step read (and the latter
// writes) SIX separate arrays per id -- emb, its two Adam slots, the linear weight, its two slots: six 128-byte granules per id at
// K = 16 -- where a record holding all of them would be two.
//
// Layouts (K = 16; floats per row):
//   separate  : emb [V,16], emb_m [V,16], emb_v [V,16], lin [V], lin_m [V], lin_v [V]   -- the engine's (variables keep the reference's shapes)
//   rec_tmv64 : one 64-float (256-byte, two granules) record per row: [theta 16 | w | pad 3 | m 16 | w_m | pad 3 | v 16 | w_v | pad 7]
//   rec_tmv51 : one 52-float record (51 used; 16-byte aligned pieces: [theta 16 | m 16 | v 16 | w w_m w_v pad]) -- records straddle granules
// Kernels: the product's own code paths re-instantiated on strided views -- the LAG gather's access pattern (row + both slots +
// linear weight and its slots read per id, replay in registers; lag.h replay loop and opt_rules.h arithmetic are #included, not
// imitated), a touched-rows step (one walker of K/4 lanes per distinct row: read row + slots + a compact gradient row, replay, Adam
// step, write back: scatter_apply_kernel's short-segment path, 85 % of a Criteo batch's distinct ids), and the 1/N background sweep.
// Sizes: c2's table (V = 1e6: inside the 256 MB Infinity Cache in every layout) and an HBM-resident one (V = 32 M rows: 6.5 - 8.2 GB).
//
// build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I tf_repos_amd/csrc tools/table_records_probe.hip -o /tmp/trp && /tmp/trp
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

#include "lag.h"

using namespace dctr;

namespace dctr {        // (the two host helpers the included headers declare; unused by the probe's device code)
void set_error(const char*, ...) {}
}

struct Lay {
    float* th; float* m; float* v;       // row r, piece kq: th + r * ld + 4 kq   (ld in floats)
    float* w; float* wm; float* wv;      // row r: w + r * wld
    int ld, wld;
};

constexpr int KQ = 4;                    // K = 16

// ---- LAG gather: gather.hip's mapping (KQ lanes x FS field groups per example, U rows in flight per lane), strided views
template <int FS, int U>
__global__ __launch_bounds__(256) void gather_lag_kernel(Lay T, const uint8_t* __restrict__ ts, const StepState* __restrict__ S, float l2,
                                                        const int32_t* __restrict__ ids, const float* __restrict__ vals, int B, int F,
                                                        float* __restrict__ e_out, float* __restrict__ yw_out) {
    constexpr int TPE = KQ * FS, EPB = 256 / TPE;
    const int sub = threadIdx.x % TPE, kq = sub % KQ, fs = sub / KQ;
    const int b = blockIdx.x * EPB + threadIdx.x / TPE;
    if (b >= B) return;
    float yw = 0.f;
    const int32_t* idr = ids + (size_t)b * F;
    const float* vr = vals + (size_t)b * F;
    float4* er = reinterpret_cast<float4*>(e_out + (size_t)b * F * 16);
    for (int f0 = fs; f0 < F; f0 += FS * U) {
        int32_t id[U]; float val[U]; float4 r[U], m[U], vv[U]; float w[U], lm[U], lv[U]; int nlag[U], nl[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int f = f0 + u * FS; id[u] = f < F ? idr[f] : 0; val[u] = f < F ? vr[f] : 0.f; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = f0 + u * FS < F;
            const size_t o = (size_t)id[u] * T.ld + 4 * kq;
            r[u] = ok ? *reinterpret_cast<const float4*>(T.th + o) : make_float4(0.f, 0.f, 0.f, 0.f);
            m[u] = ok ? *reinterpret_cast<const float4*>(T.m + o) : make_float4(0.f, 0.f, 0.f, 0.f);
            vv[u] = ok ? *reinterpret_cast<const float4*>(T.v + o) : make_float4(0.f, 0.f, 0.f, 0.f);
            nlag[u] = ok ? lag_behind(S->t - 1, ts[id[u]]) : 0;
            w[u] = lm[u] = lv[u] = 0.f;
            if (ok && kq == 0) { const size_t ow = (size_t)id[u] * T.wld; w[u] = T.w[ow]; lm[u] = T.wm[ow]; lv[u] = T.wv[ow]; }
            nl[u] = kq == 0 ? nlag[u] : 0;
        }
        lag_catch_up_rows_lin<U, true>(S, S->hyper, l2, S->t - 1, nlag, r, m, vv, nl, w, lm, lv);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int f = f0 + u * FS;
            if (f < F) {
                er[(size_t)f * KQ + kq] = make_float4(r[u].x * val[u], r[u].y * val[u], r[u].z * val[u], r[u].w * val[u]);
                yw += w[u] * val[u];
            }
        }
    }
#pragma unroll
    for (int off = KQ; off < TPE; off <<= 1) yw += __shfl_xor(yw, off);
    if (sub == 0) yw_out[b] = yw;
}

// ---- touched-rows step: one walker of KQ lanes per distinct row (the fused tail's short-segment path)
__global__ __launch_bounds__(256) void touched_step_kernel(Lay T, uint8_t* __restrict__ ts, const StepState* __restrict__ S, float l2,
                                                          const int32_t* __restrict__ uniq, int U, const float4* __restrict__ gemb,
                                                          const float* __restrict__ glin) {
    const int t = blockIdx.x * 256 + threadIdx.x, u = t / KQ, kq = t % KQ;
    if (u >= U) return;
    const int64_t r = uniq[u];
    const size_t o = (size_t)r * T.ld + 4 * kq, ow = (size_t)r * T.wld;
    float4 th = *reinterpret_cast<const float4*>(T.th + o), a = *reinterpret_cast<const float4*>(T.m + o), b = *reinterpret_cast<const float4*>(T.v + o);
    float lt = 0.f, la = 0.f, lb = 0.f;
    if (kq == 0) { lt = T.w[ow]; la = T.wm[ow]; lb = T.wv[ow]; }
    const float4 g = gemb[(size_t)u * KQ + kq];
    const float gl = kq == 0 ? glin[u] : 0.f;
    const uint8_t stamp = ts[r];
    const int n = lag_behind(S->t - 1, stamp);
    const Hyper h = S->hyper;
    lag_catch_up4_lin(S, h, l2, S->t - 1, n, th, a, b, kq == 0 ? n : 0, lt, la, lb);
    opt_update(DCTR_OPT_ADAM, h, th.x, a.x, b.x, l2 * th.x + g.x);
    opt_update(DCTR_OPT_ADAM, h, th.y, a.y, b.y, l2 * th.y + g.y);
    opt_update(DCTR_OPT_ADAM, h, th.z, a.z, b.z, l2 * th.z + g.z);
    opt_update(DCTR_OPT_ADAM, h, th.w, a.w, b.w, l2 * th.w + g.w);
    *reinterpret_cast<float4*>(T.th + o) = th; *reinterpret_cast<float4*>(T.m + o) = a; *reinterpret_cast<float4*>(T.v + o) = b;
    if (kq == 0) {
        opt_update(DCTR_OPT_ADAM, h, lt, la, lb, l2 * lt + gl);
        T.w[ow] = lt; T.wm[ow] = la; T.wv[ow] = lb;
        ts[r] = stamp;                          // (the probe writes the stamp back as drawn: every launch sees the same lags)
    }
}

// ---- background sweep over rows [r0, r1): lag.hip's lag_advance_kernel mapping (4 row pieces in flight per lane, small grid)
__global__ __launch_bounds__(256) void sweep_kernel(Lay T, const uint8_t* __restrict__ ts, const StepState* __restrict__ S, float l2, int64_t r0, int64_t r1, int nsteps) {
    constexpr int UNR = 4;
    const int64_t n_items = (r1 - r0) * KQ, stride = (int64_t)gridDim.x * 256;
    const Hyper h = S->hyper;
    for (int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x; t0 < n_items; t0 += stride * UNR) {
        float4 th[UNR], m[UNR], v[UNR]; float lt[UNR], lm[UNR], lv[UNR]; int n[UNR], nl[UNR]; size_t o[UNR], ow[UNR];
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            const int64_t t = t0 + j * stride;
            const bool live = t < n_items;
            const int64_t row = r0 + (live ? t / KQ : 0);
            const int kq = (int)(t % KQ);
            o[j] = (size_t)row * T.ld + 4 * kq; ow[j] = (size_t)row * T.wld;
            n[j] = live ? nsteps + (ts[row] & 0) : 0;           // (every row of the block is `nsteps` behind: the steady state)
            th[j] = *reinterpret_cast<const float4*>(T.th + o[j]); m[j] = *reinterpret_cast<const float4*>(T.m + o[j]); v[j] = *reinterpret_cast<const float4*>(T.v + o[j]);
            nl[j] = (live && kq == 0) ? n[j] : 0;
            lt[j] = lm[j] = lv[j] = 0.f;
            if (nl[j]) { lt[j] = T.w[ow[j]]; lm[j] = T.wm[ow[j]]; lv[j] = T.wv[ow[j]]; }
        }
        lag_catch_up_rows_lin<UNR>(S, h, l2, S->t, n, th, m, v, nl, lt, lm, lv);
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            if (n[j] <= 0) continue;
            *reinterpret_cast<float4*>(T.th + o[j]) = th[j]; *reinterpret_cast<float4*>(T.m + o[j]) = m[j]; *reinterpret_cast<float4*>(T.v + o[j]) = v[j];
            if (nl[j]) { T.w[ow[j]] = lt[j]; T.wm[ow[j]] = lm[j]; T.wv[ow[j]] = lv[j]; }
        }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Alloc { std::vector<void*> p; float* f(size_t n) { void* q; CK(hipMalloc(&q, n * 4)); CK(hipMemset(q, 0, n * 4)); p.push_back(q); return (float*)q; } ~Alloc() { for (auto q : p) hipFree(q); } };

static Lay make_layout(Alloc& A, const char* name, int64_t V) {
    Lay T{};
    if (!strcmp(name, "separate")) {
        T.th = A.f(V * 16); T.m = A.f(V * 16); T.v = A.f(V * 16); T.w = A.f(V); T.wm = A.f(V); T.wv = A.f(V); T.ld = 16; T.wld = 1;
    } else if (!strcmp(name, "rec_tmv64")) {
        float* b = A.f(V * 64);
        T.th = b; T.w = b + 16; T.m = b + 20; T.wm = b + 36; T.v = b + 40; T.wv = b + 56; T.ld = T.wld = 64;
    } else {        // rec_tmv51 (52 floats)
        float* b = A.f(V * 52);
        T.th = b; T.m = b + 16; T.v = b + 32; T.w = b + 48; T.wm = b + 49; T.wv = b + 50; T.ld = T.wld = 52;
    }
    return T;
}

int main(int argc, char** argv) {
    const int B = 4096, F = 39, PERIOD = 8, ITERS = 100;
    const int64_t Vs[2] = {1000000, 32ll << 20};
    const char* lays[3] = {"separate", "rec_tmv64", "rec_tmv51"};
    printf("%-10s %-10s %9s | %12s %12s %12s | %s\n", "rows", "layout", "table GB", "LAG gather", "touched step", "1/8 sweep", "(us per launch; B = 4096, F = 39, K = 16; 62 % of a batch's rows lag 1..8 steps)");
    for (int vi = 0; vi < 2; ++vi) {
        const int64_t V = Vs[vi];
        // one batch: Criteo shape -- 13 always-hit ids, the rest uniform over the table (worst case for caches); its distinct rows
        std::mt19937_64 rng(7);
        std::vector<int32_t> ids((size_t)B * F);
        for (int b = 0; b < B; ++b) for (int f = 0; f < F; ++f) ids[(size_t)b * F + f] = f < 13 ? f + 1 : (int32_t)(rng() % (uint64_t)V);
        std::vector<int32_t> uniq(ids); std::sort(uniq.begin(), uniq.end()); uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
        std::shuffle(uniq.begin(), uniq.end(), rng);         // (the grouping's compact order is arrival order, not sorted)
        const int U = (int)uniq.size();
        std::vector<uint8_t> ts((size_t)V);
        const int Tnow = 100;
        for (int64_t r = 0; r < V; ++r) ts[r] = (uint8_t)((rng() % 100 < 62) ? Tnow - 1 - (1 + rng() % PERIOD) : Tnow - 1);
        for (int k = 1; k <= 13; ++k) ts[k] = (uint8_t)(Tnow - 1);
        StepState hs{}; hs.t = Tnow; hs.hyper.lr = 5e-4f; hs.hyper.beta1 = 0.9f; hs.hyper.beta2 = 0.999f; hs.hyper.eps = 1e-8f; hs.hyper.lr_t = 4e-4f;
        for (int i = 0; i < LR_HIST; ++i) hs.lr_hist[i] = 4e-4f;
        for (int li = 0; li < 3; ++li) {
            Alloc A;
            Lay T = make_layout(A, lays[li], V);
            const double gb = (li == 0 ? 51.0 : (li == 1 ? 64.0 : 52.0)) * 4 * V / 1e9;
            int32_t *d_ids, *d_uniq; uint8_t* d_ts; StepState* d_S; float *d_vals, *d_e, *d_yw, *d_g, *d_gl;
            CK(hipMalloc(&d_ids, ids.size() * 4)); CK(hipMemcpy(d_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
            CK(hipMalloc(&d_uniq, (size_t)U * 4)); CK(hipMemcpy(d_uniq, uniq.data(), (size_t)U * 4, hipMemcpyHostToDevice));
            CK(hipMalloc(&d_ts, (size_t)V)); CK(hipMemcpy(d_ts, ts.data(), (size_t)V, hipMemcpyHostToDevice));
            CK(hipMalloc(&d_S, sizeof(StepState))); CK(hipMemcpy(d_S, &hs, sizeof(hs), hipMemcpyHostToDevice));
            d_vals = A.f((size_t)B * F); d_e = A.f((size_t)B * F * 16); d_yw = A.f(B); d_g = A.f((size_t)U * 16); d_gl = A.f(U);
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            auto time = [&](auto launch) {
                for (int i = 0; i < 5; ++i) launch();
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                for (int i = 0; i < ITERS; ++i) launch();
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                return 1e3 * ms / ITERS;
            };
            const double t_g = time([&] { gather_lag_kernel<8, 5><<<(B + 7) / 8, 256>>>(T, d_ts, d_S, 1e-4f, d_ids, d_vals, B, F, d_e, d_yw); });
            const double t_s = time([&] { touched_step_kernel<<<(U * KQ + 255) / 256, 256>>>(T, d_ts, d_S, 1e-4f, d_uniq, U, (const float4*)d_g, d_gl); });
            const int64_t rpb = (V + PERIOD - 1) / PERIOD;
            int blk = 0;
            const double t_w = time([&] { const int64_t r0 = (blk++ % PERIOD) * rpb; sweep_kernel<<<512, 256>>>(T, d_ts, d_S, 1e-4f, r0, std::min(V, r0 + rpb), PERIOD); });
            printf("%-10lld %-10s %9.2f | %12.2f %12.2f %12.2f | distinct rows %d\n", (long long)V, lays[li], gb, t_g, t_s, t_w, U);
            fflush(stdout);
            hipFree(d_ids); hipFree(d_uniq); hipFree(d_ts); hipFree(d_S); hipEventDestroy(e0); hipEventDestroy(e1);
        }
    }
    return 0;
}
