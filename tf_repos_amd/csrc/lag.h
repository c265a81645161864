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
};

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
// steps a row stamped `ts` is behind `target` (both compared mod 256)
__device__ __forceinline__ int lag_behind(int64_t target, uint8_t ts) { return (int)(uint8_t)((uint8_t)target - ts); }

int lag_sweep(int K, int64_t rows, float* emb, float* s0, float* s1, float* lin, float* l0, float* l1, const int32_t* slot, uint8_t* ts,
              const StepState* state, float l2, int period, hipStream_t st);
int lag_flush(int K, int64_t rows, float* emb, float* s0, float* s1, float* lin, float* l0, float* l1, uint8_t* ts, const StepState* state,
              float l2, int target_offset, float* sumsq_emb, float* sumsq_lin, hipStream_t st);
int lag_stamp(uint8_t* ts, int64_t rows, const StepState* state, hipStream_t st);

}  // namespace dctr
