// ESMM (DeepMTL/Model_pipeline/DeepCvrMTL.py): the multi-task head over the CTR and CVR towers' last hidden layers.
//   y_ctr = h_ctr . w_ctr + b_ctr, y_cvr = h_cvr . w_cvr + b_cvr           (ctr_out / cvr_out, :182-184,203-205)
//   pctr = sigmoid(y_ctr), pcvr = sigmoid(y_cvr), pctcvr = pctr * pcvr       (:207-210)
//   loss = w * mean xent(y_ctr, y) + (1 - w) * mean log_loss(pctcvr, z)      (:222-225; tf.losses.log_loss eps = 1e-7 [TF-1.4])
// One wave per example: the two dot products, then lane 0 does the scalar tail and writes dL/dy of both towers.
#include "ops.h"

namespace dctr {

__global__ __launch_bounds__(256) void esmm_head_kernel(const float* __restrict__ h_ctr, int ld_ctr, const float* __restrict__ w_ctr,
                                                       const float* __restrict__ b_ctr, int n_ctr, const float* __restrict__ h_cvr,
                                                       int ld_cvr, const float* __restrict__ w_cvr, const float* __restrict__ b_cvr,
                                                       int n_cvr, const float* __restrict__ y, const float* __restrict__ z, int B,
                                                       float inv_b, float wgt, float* __restrict__ y_ctr, float* __restrict__ y_cvr,
                                                       float* __restrict__ pctr, float* __restrict__ pcvr, float* __restrict__ pctcvr,
                                                       float* __restrict__ dy_ctr, float* __restrict__ dy_cvr,
                                                       float* __restrict__ loss_shards) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    float a = 0.f, c = 0.f;
    for (int j = lane; j < n_ctr; j += 64) a += h_ctr[(size_t)b * ld_ctr + j] * w_ctr[j];
    for (int j = lane; j < n_cvr; j += 64) c += h_cvr[(size_t)b * ld_cvr + j] * w_cvr[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); c += __shfl_xor(c, o); }
    if (lane != 0) return;
    const float yc = a + b_ctr[0], yv = c + b_cvr[0];
    const float pc = 1.0f / (1.0f + expf(-yc)), pv = 1.0f / (1.0f + expf(-yv));
    const float pj = pc * pv;
    y_ctr[b] = yc; y_cvr[b] = yv; pctr[b] = pc; pcvr[b] = pv; pctcvr[b] = pj;
    if (y == nullptr) return;
    const float yl = y[b], zl = z[b];
    constexpr float eps = 1e-7f;
    // d log_loss / d pctcvr, then through the product and the two sigmoids
    const float dl = -zl / (pj + eps) + (1.0f - zl) / (1.0f - pj + eps);
    dy_ctr[b] = inv_b * (wgt * (pc - yl) + (1.0f - wgt) * dl * pv * pc * (1.0f - pc));
    dy_cvr[b] = inv_b * ((1.0f - wgt) * dl * pc * pv * (1.0f - pv));
    const float xent = fmaxf(yc, 0.f) - yc * yl + log1pf(expf(-fabsf(yc)));
    const float ll = -zl * logf(pj + eps) - (1.0f - zl) * logf(1.0f - pj + eps);
    atomicAdd(&loss_shards[blockIdx.x % SUMSQ_SHARDS], wgt * xent + (1.0f - wgt) * ll);
}

int esmm_head(const float* h_ctr, int ld_ctr, const float* w_ctr, const float* b_ctr, int n_ctr, const float* h_cvr, int ld_cvr,
              const float* w_cvr, const float* b_cvr, int n_cvr, const float* y, const float* z, int B, float inv_b, float wgt,
              float* y_ctr, float* y_cvr, float* pctr, float* pcvr, float* pctcvr, float* dy_ctr, float* dy_cvr, float* loss_shards,
              hipStream_t st) {
    if (B <= 0) return DCTR_OK;
    esmm_head_kernel<<<ceil_div(B, 4), 256, 0, st>>>(h_ctr, ld_ctr, w_ctr, b_ctr, n_ctr, h_cvr, ld_cvr, w_cvr, b_cvr, n_cvr, y, z, B, inv_b,
                                                     wgt, y_ctr, y_cvr, pctr, pcvr, pctcvr, dy_ctr, dy_cvr, loss_shards);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

// a += b (float4 pieces): the second tower's dL/dx joins the first's before the table backward
__global__ __launch_bounds__(256) void add_inplace_kernel(float4* __restrict__ a, const float4* __restrict__ b, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 x = a[i];
    const float4 y = b[i];
    x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
    a[i] = x;
}

int add_inplace(float* a, const float* b, int64_t n, hipStream_t st) {
    DCTR_REQUIRE(n % 4 == 0, "add_inplace: n must be a multiple of 4");
    if (n <= 0) return DCTR_OK;
    add_inplace_kernel<<<ceil_div(n / 4, 256), 256, 0, st>>>(reinterpret_cast<float4*>(a), reinterpret_cast<const float4*>(b), n / 4);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

}  // namespace dctr
