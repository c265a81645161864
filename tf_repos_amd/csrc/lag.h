// Time-blocked dense-exact table sweep ("lagging rows"), Adam.
//
// The reference's loss holds l2_reg * l2_loss(table) (DeepFM.py:188-190), so TF's Adam steps EVERY table row EVERY step; for the
// rows a batch does not touch the gradient is just l2 * theta and the row's (theta, m, v) follow a recurrence that needs nothing
// the step computes.  The classic path streams all V rows through HBM each step for that (c2: 412 MB, c5: 80 GB per step).
// Here a row may LAG: row_ts[r] (one byte) holds the step up to which the row has been advanced, and whoever needs the row
// next first replays the missed l2-only steps IN REGISTERS -- the same opt_update calls with the same per-step lr_t (kept in
// StepState::lr_hist), in the same order, so every row sees exactly the arithmetic of the classic sweep, step for step:
//   * the gather of step t advances a private copy of each gathered row to t-1 (nothing written back);
//   * the touched-rows step of step t (scatter_apply_kernel) advances the row to t-1, applies step t with the batch's gradient,
//     stamps it t;
//   * a background sweep visits block (t mod N) of the table -- 1/N of the rows per step -- and advances its untouched rows
//     to t; so no row ever lags more than N steps;
//   * lag_flush advances every row to the present (before predict / eval / parameter reads / a step that reports the loss,
//     whose l2 term needs sum theta^2 of all rows; the flush accumulates it).
// HBM traffic of the table step: 6 V (K+1) 4 / N bytes per step instead of 6 V (K+1) 4; the arithmetic is unchanged (temporal
// blocking of a streaming recurrence).  N = dctr_config.table_sweep_period (1 = classic sweep).
#pragma once
#include "ops.h"
#include "opt_rules.h"

namespace dctr {

constexpr int LAG_MAX_PERIOD = 24;          // < LR_HIST (the ring of per-step lr_t) and far below the 8-bit stamp's range

struct LagView {
    uint8_t* ts;                // [rows] step (mod 256) each row has been advanced to; nullptr = rows never lag (classic)
    const StepState* state;     // t, hyper, lr_hist
    float4* s0; float4* s1;     // the table's Adam slots (the gather needs them only for lagging rows)
    float* l0; float* l1;       // the linear table's
    float l2;
    int ld4 = 0;                // row stride of s0 / s1 in float4 units (0: K / 4 -- separate [rows, K] arrays); the table's own stride
    int lin_ld = 1;             // row stride of l0 / l1 in floats (records: engine.h table layout)
};

#ifdef DCTR_LAG_LANE_LOOPS      // (A/B: round 3's lane-own loops, tf_repos_amd.build --variant lanelag -DDCTR_LAG_LANE_LOOPS)
// replay steps first .. first+n-1 of a row piece that no batch touched: g = l2 * theta (what opt_table_untouched_kernel computes)
__device__ __forceinline__ void lag_catch_up4(const StepState* __restrict__ S, Hyper h, float l2, int64_t first, int n, float4& th, float4& m, float4& v) {
    for (int k = 0; k < n; ++k) {
        h.lr_t = S->lr_hist[(first + k) & (LR_HIST - 1)];
        opt_update(DCTR_OPT_ADAM, h, th.x, m.x, v.x, l2 * th.x);
        opt_update(DCTR_OPT_ADAM, h, th.y, m.y, v.y, l2 * th.y);
        opt_update(DCTR_OPT_ADAM, h, th.z, m.z, v.z, l2 * th.z);
        opt_update(DCTR_OPT_ADAM, h, th.w, m.w, v.w, l2 * th.w);
    }
}
__device__ __forceinline__ void lag_catch_up1(const StepState* __restrict__ S, Hyper h, float l2, int64_t first, int n, float& th, float& m, float& v) {
    for (int k = 0; k < n; ++k) {
        h.lr_t = S->lr_hist[(first + k) & (LR_HIST - 1)];
        opt_update(DCTR_OPT_ADAM, h, th, m, v, l2 * th);
    }
}
// R row pieces at once, each n[r] steps behind `last` (its steps last-n[r]+1 .. last are replayed): ONE loop over the steps, the
// rows' recurrences side by side -- R independent dependency chains per lane instead of R loops in sequence (the Adam update is
// a ~15-instruction chain through sqrt and rcp; alone it leaves the ALU waiting on itself)
template <int R>
__device__ __forceinline__ void lag_catch_up4_rows(const StepState* __restrict__ S, Hyper h, float l2, int64_t last, const int (&n)[R],
                                                   float4 (&th)[R], float4 (&m)[R], float4 (&v)[R]) {
    int nmax = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) nmax = n[r] > nmax ? n[r] : nmax;
    for (int k = nmax; k >= 1; --k) {            // step last - k + 1
        h.lr_t = S->lr_hist[(last - k + 1) & (LR_HIST - 1)];
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (k <= n[r]) {
                opt_update(DCTR_OPT_ADAM, h, th[r].x, m[r].x, v[r].x, l2 * th[r].x);
                opt_update(DCTR_OPT_ADAM, h, th[r].y, m[r].y, v[r].y, l2 * th[r].y);
                opt_update(DCTR_OPT_ADAM, h, th[r].z, m[r].z, v[r].z, l2 * th[r].z);
                opt_update(DCTR_OPT_ADAM, h, th[r].w, m[r].w, v[r].w, l2 * th[r].w);
            }
    }
}
template <int R>
__device__ __forceinline__ void lag_catch_up1_rows(const StepState* __restrict__ S, Hyper h, float l2, int64_t last, const int (&n)[R],
                                                   float (&th)[R], float (&m)[R], float (&v)[R]) {
    int nmax = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) nmax = n[r] > nmax ? n[r] : nmax;
    for (int k = nmax; k >= 1; --k) {
        h.lr_t = S->lr_hist[(last - k + 1) & (LR_HIST - 1)];
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (k <= n[r]) opt_update(DCTR_OPT_ADAM, h, th[r], m[r], v[r], l2 * th[r]);
    }
}
template <int R, bool SKIP = false>
__device__ __forceinline__ void lag_catch_up_rows_lin(const StepState* __restrict__ S, Hyper h, float l2, int64_t last, const int (&n)[R],
                                                      float4 (&th)[R], float4 (&m)[R], float4 (&v)[R], const int (&nl)[R], float (&lt)[R],
                                                      float (&lm)[R], float (&lv)[R]) {
    lag_catch_up4_rows<R>(S, h, l2, last, n, th, m, v);
    lag_catch_up1_rows<R>(S, h, l2, last, nl, lt, lm, lv);
}
__device__ __forceinline__ void lag_catch_up4_lin(const StepState* __restrict__ S, Hyper h, float l2, int64_t last, int n, float4& th, float4& m, float4& v,
                                                  int nlin, float& lt, float& lm, float& lv) {
    lag_catch_up4(S, h, l2, last - n + 1, n, th, m, v);
    if (nlin > 0) lag_catch_up1(S, h, l2, last - nlin + 1, nlin, lt, lm, lv);
}
#else
// ---- the replay loop.  Every caller has lanes that are behind by DIFFERENT numbers of steps (a gather's rows, a sweep block's
// recently touched rows), and the step to replay fixes the Adam lr_t to use.  The loop therefore runs over the STEPS, wave-uniformly
// (k = kmax .. 1 <-> step last - k + 1, kmax the largest lag among the wave's active lanes), so that
//   * lr_t of a step is ONE scalar load for the wave, issued an iteration ahead of its use (a lane-indexed load of the history ring
//     was a vector load plus a full wait in every iteration of every lane's own loop: the loop was a chain of memory latencies);
//   * the R rows' (and the linear weight's) updates of a step sit side by side in one basic block -- 4 R + 1 independent dependency
//     chains through sqrt and rcp for the scheduler to interleave -- and a lane that is not behind that far takes the update and
//     throws it away with a select (`k <= n`), instead of each row's update sitting in its own divergent branch.
// What is computed for a lane that takes part is, call for call, what the lane-own loops computed: opt_update with the step's lr_t
// and g = l2 * theta, oldest missed step first.
__device__ __forceinline__ int lag_wave_max_lag(int nmax) {          // largest lag among the active lanes (wave-uniform), 0 if none is behind
    for (int k = LAG_MAX_PERIOD; k >= 1; --k)
        if (__any(k <= nmax)) return k;
    return 0;
}
// SKIP: a row slot that NO lane of the wave needs at step k is skipped by a wave-uniform branch -- the gather's slots are fields, and
// Criteo's numeric fields (always-hit ids) are current on every lane: a third of the work (measured, same box: the LAG gather 15.8 us
// with lane-own loops, 20.6 with every slot computed and selected away).  The price is a basic block per slot: the sweep, whose rows
// all lag alike, leaves it off and gets 4 R + 1 interleaved chains (lag_advance_kernel 41 -> 31 us).
template <int R, bool LIN, bool SKIP = false>
__device__ __forceinline__ void lag_replay_rows(const StepState* __restrict__ S, Hyper h, float l2, int64_t last, const int (&n)[R],
                                                float4 (&th)[R], float4 (&m)[R], float4 (&v)[R], const int (&nl)[R], float (&lt)[R],
                                                float (&lm)[R], float (&lv)[R]) {
    int nmax = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) nmax = n[r] > nmax ? n[r] : nmax;
    if (LIN) {
#pragma unroll
        for (int r = 0; r < R; ++r) nmax = nl[r] > nmax ? nl[r] : nmax;
    }
    const int kmax = lag_wave_max_lag(nmax);
    if (kmax == 0) return;
    // (a row further behind than the loop can replay would be stamped current after LAG_MAX_PERIOD updates: never by the engine's own
    //  schedule -- the sweep bounds every row's lag by the period -- so it is a broken invariant, flagged instead of silently wrong)
    if (__any(nmax > LAG_MAX_PERIOD) && nmax > LAG_MAX_PERIOD) const_cast<StepState*>(S)->lag_overflow = 1;
    // (`last` is the same on every lane -- callers derive it from StepState::t; readfirstlane says so to the compiler, which would
    //  otherwise index the ring per lane: a vector load.  The ring index needs the low bits only.)
    const int ilast = __builtin_amdgcn_readfirstlane((int)last);
    float lr = S->lr_hist[(ilast - kmax + 1) & (LR_HIST - 1)];
#ifdef DCTR_LAG_SELECT          // (A/B: rounds 4-5 -- every lane takes the update, a lane that is not that far behind selects it away: 3 v_cndmask per ELEMENT and step)
    for (int k = kmax; k >= 1; --k) {                               // step last - k + 1
        const float lr_next = S->lr_hist[(ilast - k + 2) & (LR_HIST - 1)];     // (of step last - k + 2: the next iteration's; k = 1 reads one entry past `last`, unused)
        h.lr_t = lr;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (SKIP && !__any(k <= n[r] || (LIN && k <= nl[r]))) continue;
            float4 t = th[r], a = m[r], b = v[r];
            opt_update(DCTR_OPT_ADAM, h, t.x, a.x, b.x, l2 * t.x);
            opt_update(DCTR_OPT_ADAM, h, t.y, a.y, b.y, l2 * t.y);
            opt_update(DCTR_OPT_ADAM, h, t.z, a.z, b.z, l2 * t.z);
            opt_update(DCTR_OPT_ADAM, h, t.w, a.w, b.w, l2 * t.w);
            const bool on = k <= n[r];
            th[r].x = on ? t.x : th[r].x; th[r].y = on ? t.y : th[r].y; th[r].z = on ? t.z : th[r].z; th[r].w = on ? t.w : th[r].w;
            m[r].x = on ? a.x : m[r].x; m[r].y = on ? a.y : m[r].y; m[r].z = on ? a.z : m[r].z; m[r].w = on ? a.w : m[r].w;
            v[r].x = on ? b.x : v[r].x; v[r].y = on ? b.y : v[r].y; v[r].z = on ? b.z : v[r].z; v[r].w = on ? b.w : v[r].w;
            if (LIN) {
                float t1 = lt[r], a1 = lm[r], b1 = lv[r];
                opt_update(DCTR_OPT_ADAM, h, t1, a1, b1, l2 * t1);
                const bool on1 = k <= nl[r];
                lt[r] = on1 ? t1 : lt[r]; lm[r] = on1 ? a1 : lm[r]; lv[r] = on1 ? b1 : lv[r];
            }
        }
        lr = lr_next;
    }
#else
    // Round 6: the replay is VALU-bound (tools/lag_probe.hip: the 1/8 sweep of c2 is 17 M element-updates; per update the ISA of the
    // select form spent 12 VALU slots of 4 cycles + sqrt and rcp at 16: ~73 cycles per element and wave, of which 12 were the three selects
    // and 9 register copies around them).  A lane that is not behind that far now runs the step with the IDENTITY's coefficients instead
    // -- beta = 1, 1 - beta = 0, lr_t = 0: m' = 1 m + 0 g = m, v' = 1 v + (0 g) g = v, theta' = theta - (0 m') rcp(..) = theta, exactly (m, v,
    // theta finite; sqrt(v) + eps > 0) -- five selects per ROW and step in place of three per element, and the updates go to the rows'
    // registers in place.  A lane that takes part computes opt_update's Adam expressions, operand for operand.  The rows' linear weights
    // ride on their row's coefficients: nl[r] is n[r] on the lane that owns the weight and 0 elsewhere, where lt / lm / lv are zeros or
    // never stored (a zero stays a zero under the update).  PRECONDITION: th / m / v of EVERY lane are finite numbers (zeros where nothing was
    // loaded), also on lanes with n[r] = 0: 0 x NaN is NaN.
    const float c1 = 1.0f - h.beta1, c2 = 1.0f - h.beta2;
    auto step = [&](float& t, float& a, float& b, float b1, float a1, float b2, float a2, float lrk) {
        const float g = l2 * t;
        a = b1 * a + a1 * g;
        b = b2 * b + a2 * g * g;
        if constexpr (ADAM_TABLES_EXACT) t = t - lrk * a / (sqrtf(b) + h.eps);
        else t = t - (lrk * a) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(b) + h.eps);
    };
    for (int k = kmax; k >= 1; --k) {                               // step last - k + 1
        const float lr_next = S->lr_hist[(ilast - k + 2) & (LR_HIST - 1)];     // (of step last - k + 2: the next iteration's; k = 1 reads one entry past `last`, unused)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (SKIP && !__any(k <= n[r] || (LIN && k <= nl[r]))) continue;
            const bool on = k <= n[r];
            const float b1 = on ? h.beta1 : 1.f, a1 = on ? c1 : 0.f, b2 = on ? h.beta2 : 1.f, a2 = on ? c2 : 0.f, lrk = on ? lr : 0.f;
            step(th[r].x, m[r].x, v[r].x, b1, a1, b2, a2, lrk);
            step(th[r].y, m[r].y, v[r].y, b1, a1, b2, a2, lrk);
            step(th[r].z, m[r].z, v[r].z, b1, a1, b2, a2, lrk);
            step(th[r].w, m[r].w, v[r].w, b1, a1, b2, a2, lrk);
            if (LIN) step(lt[r], lm[r], lv[r], b1, a1, b2, a2, lrk);
        }
        lr = lr_next;
    }
#endif
}
// R row pieces, each n[r] steps behind `last` (its steps last - n[r] + 1 .. last are replayed)
template <int R>
__device__ __forceinline__ void lag_catch_up4_rows(const StepState* __restrict__ S, Hyper h, float l2, int64_t last, const int (&n)[R],
                                                   float4 (&th)[R], float4 (&m)[R], float4 (&v)[R]) {
    int nl[R]; float a[R], b[R], c[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { nl[r] = 0; a[r] = b[r] = c[r] = 0.f; }
    lag_replay_rows<R, false>(S, h, l2, last, n, th, m, v, nl, a, b, c);
}
// ... together with the rows' linear weights (nl[r]: the same lag on the lane that owns the weight, 0 elsewhere)
template <int R, bool SKIP = false>
__device__ __forceinline__ void lag_catch_up_rows_lin(const StepState* __restrict__ S, Hyper h, float l2, int64_t last, const int (&n)[R],
                                                      float4 (&th)[R], float4 (&m)[R], float4 (&v)[R], const int (&nl)[R], float (&lt)[R],
                                                      float (&lm)[R], float (&lv)[R]) {
    lag_replay_rows<R, true, SKIP>(S, h, l2, last, n, th, m, v, nl, lt, lm, lv);
}
// one row piece: steps first .. first + n - 1 (what opt_table_untouched_kernel computes for a row no batch touched, step by step)
__device__ __forceinline__ void lag_catch_up4(const StepState* __restrict__ S, Hyper h, float l2, int64_t first, int n, float4& th, float4& m, float4& v) {
    int nn[1] = {n}, nl[1] = {0};
    float4 t[1] = {th}, a[1] = {m}, b[1] = {v};
    float x[1] = {0.f}, y[1] = {0.f}, z[1] = {0.f};
    lag_replay_rows<1, false>(S, h, l2, first + n - 1, nn, t, a, b, nl, x, y, z);
    th = t[0]; m = a[0]; v = b[0];
}
// one row piece and its linear weight (nlin = n on the lane that owns the weight, else 0)
__device__ __forceinline__ void lag_catch_up4_lin(const StepState* __restrict__ S, Hyper h, float l2, int64_t last, int n, float4& th, float4& m, float4& v,
                                                  int nlin, float& lt, float& lm, float& lv) {
    int nn[1] = {n}, nl[1] = {nlin};
    float4 t[1] = {th}, a[1] = {m}, b[1] = {v};
    float x[1] = {lt}, y[1] = {lm}, z[1] = {lv};
    lag_replay_rows<1, true>(S, h, l2, last, nn, t, a, b, nl, x, y, z);
    th = t[0]; m = a[0]; v = b[0]; lt = x[0]; lm = y[0]; lv = z[0];
}
__device__ __forceinline__ void lag_catch_up1(const StepState* __restrict__ S, Hyper h, float l2, int64_t first, int n, float& th, float& m, float& v) {
    // (a linear weight alone: the row-sharded pack of a lin-only record)
    const int kmax = lag_wave_max_lag(n);
    const int ilast = __builtin_amdgcn_readfirstlane((int)(first + n - 1));      // (callers pass first = last - n + 1: the same `last` on every lane)
    for (int k = kmax; k >= 1; --k) {
        h.lr_t = S->lr_hist[(ilast - k + 1) & (LR_HIST - 1)];
        float t = th, a = m, b = v;
        opt_update(DCTR_OPT_ADAM, h, t, a, b, l2 * t);
        const bool on = k <= n;
        th = on ? t : th; m = on ? a : m; v = on ? b : v;
    }
}
#endif
// steps a row stamped `ts` is behind `target` (both compared mod 256)
__device__ __forceinline__ int lag_behind(int64_t target, uint8_t ts) { return (int)(uint8_t)((uint8_t)target - ts); }

// (ld / lin_ld: row strides in floats of emb, s0, s1 / lin, l0, l1; 0 / 1 = separate dense arrays [rows, K] / [rows])
int lag_sweep(int K, int64_t rows, float* emb, float* s0, float* s1, float* lin, float* l0, float* l1, const int32_t* slot, uint8_t* ts,
              const StepState* state, float l2, int period, hipStream_t st, int ld = 0, int lin_ld = 1);
int lag_flush(int K, int64_t rows, float* emb, float* s0, float* s1, float* lin, float* l0, float* l1, uint8_t* ts, const StepState* state,
              float l2, int target_offset, float* sumsq_emb, float* sumsq_lin, hipStream_t st, int ld = 0, int lin_ld = 1);
int lag_stamp(uint8_t* ts, int64_t rows, const StepState* state, hipStream_t st);

}  // namespace dctr
