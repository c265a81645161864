// Kernels special to the canned-estimator models of wide_n_deep.py (LinearClassifier / DNNClassifier /
// DNNLinearCombinedClassifier, wide_n_deep.py:113-151): the numeric columns (wide_n_deep.py:94) enter as DENSE inputs, and
// the linear side has an optimizer of its own (TF default Ftrl) applied sparsely to the touched rows of its table.
#include "opt_rules.h"

namespace dctr {

// input_layer's numeric columns: x_in[b, col0 + j] = dense[b, j]  (x_in == nullptr: no DNN side)
// linear_model's numeric columns: yw[b] += sum_j wd[j] * dense[b, j]  (wd == nullptr: no linear side)
__global__ __launch_bounds__(256) void wnd_dense_fwd_kernel(const float* __restrict__ dense, int nd, const float* __restrict__ wd,
                                                           int B, float* __restrict__ x_in, int ldx, int col0,
                                                           float* __restrict__ yw) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float acc = 0.f;
    for (int j = 0; j < nd; ++j) {
        const float v = dense[(size_t)b * nd + j];
        if (x_in != nullptr) x_in[(size_t)b * ldx + col0 + j] = v;
        if (wd != nullptr) acc += wd[j] * v;
    }
    if (wd != nullptr) yw[b] += acc;
}

int wnd_dense_fwd(const float* dense, int nd, const float* wd, int B, float* x_in, int ldx, int col0, float* yw, hipStream_t st) {
    if (B <= 0 || nd <= 0) return DCTR_OK;
    DCTR_REQUIRE(dense != nullptr, "dense inputs missing: call dctr_set_dense_input before the step");
    wnd_dense_fwd_kernel<<<ceil_div(B, 256), 256, 0, st>>>(dense, nd, wd, B, x_in, ldx, col0, yw);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

// sparse apply on the linear table: only the batch's distinct rows move (TF applies an IndexedSlices gradient with
// duplicates summed, [TF-1.4] training_ops sparse_apply_*)
template <int KIND>
__global__ __launch_bounds__(256) void opt_lin_touched_kernel(const Hyper* __restrict__ hdev, Hyper hval, float* __restrict__ lin,
                                                             float* __restrict__ l0, float* __restrict__ l1,
                                                             const int32_t* __restrict__ uniq, const int32_t* __restrict__ counters,
                                                             const float* __restrict__ glin) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= counters[0]) return;
    const Hyper h = load_hyper(hdev, hval);
    const int r = uniq[u];
    float th = lin[r], a = l0[r];
    float b = (KIND == DCTR_OPT_ADAM || KIND == DCTR_OPT_FTRL) ? l1[r] : 0.f;
    opt_update(KIND, h, th, a, b, glin[u]);
    lin[r] = th; l0[r] = a;
    if (KIND == DCTR_OPT_ADAM || KIND == DCTR_OPT_FTRL) l1[r] = b;
}

int opt_lin_touched(int kind, const Hyper* hdev, const Hyper& hval, float* lin, float* l0, float* l1, const int32_t* uniq,
                    const int32_t* counters, int64_t max_entries, const float* glin, hipStream_t st) {
    const int grid = ceil_div(max_entries, 256);
    switch (kind) {
#define DCTR_K(KD) case KD: opt_lin_touched_kernel<KD><<<grid, 256, 0, st>>>(hdev, hval, lin, l0, l1, uniq, counters, glin); break
        DCTR_K(DCTR_OPT_ADAM); DCTR_K(DCTR_OPT_ADAGRAD); DCTR_K(DCTR_OPT_MOMENTUM); DCTR_K(DCTR_OPT_FTRL);
#undef DCTR_K
        default: set_error("unknown optimizer kind %d", kind); return DCTR_ERR_INVALID_ARG;
    }
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

}  // namespace dctr
