// K7 loss head, K9 optimizers, K10 streaming AUC, and the N=1 output layers (row dots).
// All HBM-bound streaming kernels: float4 / 16-byte-per-lane coalesced accesses, grid-stride loops,
// reductions in registers -> wave shuffles -> one atomic per block.
#include "opt_rules.h"
#include <hip/hip_ext.h>

#include "common.h"
#include "ops.h"

namespace dctr {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// ---- y[b] (+)= x[b,:] . w + bias : the [*,1] fully_connected layers (DeepFM.py:165-166, DCN.py:180-182,
// AFM.py:147,160-161) and DCN's x_l . w_l (DCN.py:144).  One wave per row.
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                    const float* __restrict__ bias, int M, int n,
                                                    float* __restrict__ y, int accumulate) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* xr = x + (size_t)row * ldx;
    float s = 0.f;
    for (int j = lane; j < n; j += 64) s += xr[j] * w[j];
    s = wave_sum(s);
    if (lane == 0) {
        if (bias != nullptr) s += bias[0];
        y[row] = accumulate ? y[row] + s : s;
    }
}

// float4 form: 16 lanes per row (4 rows per wave), 16 rows per block; a [M, 256] operand streams with 4 loads per lane and a
// 4-step shuffle reduction per 4 rows (the one-wave-per-row form: 4 scalar loads + 6 steps per row -- 2.8 TB/s at M = 1e6)
__global__ __launch_bounds__(256) void rowdot_v4_kernel(const float4* __restrict__ x, int ldx4, const float4* __restrict__ w,
                                                       const float* __restrict__ bias, int M, int n4, float* __restrict__ y,
                                                       int accumulate) {
    const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int l = threadIdx.x & 15;
    float s = 0.f;
    if (row < M) {
        const float4* xr = x + (size_t)row * ldx4;
        for (int j = l; j < n4; j += 16) {
            const float4 a = xr[j], b = w[j];
            s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
        }
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
    if (row < M && l == 0) {
        if (bias != nullptr) s += bias[0];
        y[row] = accumulate ? y[row] + s : s;
    }
}

int rowdot(const float* x, int ldx, const float* w, const float* bias, int M, int n, float* y, int accumulate,
           hipStream_t st) {
    if (M <= 0) return DCTR_OK;
    if (n % 4 == 0 && ldx % 4 == 0 && n >= 32 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0) {
        rowdot_v4_kernel<<<ceil_div(M, 16), 256, 0, st>>>(reinterpret_cast<const float4*>(x), ldx / 4, reinterpret_cast<const float4*>(w),
                                                          bias, M, n / 4, y, accumulate);
        DCTR_LAUNCH_CHECK();
        return DCTR_OK;
    }
    rowdot_kernel<<<ceil_div(M, 4), 256, 0, st>>>(x, ldx, w, bias, M, n, y, accumulate);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

// ---- dX[b,j] (+)= dy[b] * w[j], optionally masked by the producing layer's ReLU/dropout ((act>0)/keep)
__global__ __launch_bounds__(256) void rank1_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ w, int M, int n,
                                                       const float* __restrict__ act, int ldact, float inv_keep,
                                                       float* __restrict__ dx, int lddx, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)M * n) return;
    const int b = (int)(i / n), j = (int)(i - (int64_t)b * n);
    float v = dy[b] * w[j];
    if (act != nullptr) v = (act[(size_t)b * ldact + j] > 0.f) ? v * inv_keep : 0.f;
    float* o = dx + (size_t)b * lddx + j;
    *o = accumulate ? *o + v : v;
}

int rank1_bwd(const float* dy, const float* w, int M, int n, const float* act, int ldact, float keep, float* dx,
              int lddx, int accumulate, hipStream_t st) {
    if (M <= 0 || n <= 0) return DCTR_OK;
    rank1_bwd_kernel<<<ceil_div((int64_t)M * n, 256), 256, 0, st>>>(dy, w, M, n, act, ldact, 1.0f / keep, dx, lddx, accumulate);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

// ---- column sums with an optional per-row scale: out[s][c] = sum_{r in split s} rs[r] * Y[r,c]
__global__ __launch_bounds__(256) void colsum_scaled_kernel(const float* __restrict__ Y, int ldy, const float* __restrict__ rs,
                                                           int M, int N, int rows_per_split, float* __restrict__ out,
                                                           int64_t split_stride) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int rbeg = blockIdx.y * rows_per_split, rend = min(M, rbeg + rows_per_split);
    float s = 0.f;
    if (c < N) {
        if (rs != nullptr)
            for (int r = rbeg + rl; r < rend; r += 4) s += rs[r] * Y[(size_t)r * ldy + c];
        else
            for (int r = rbeg + rl; r < rend; r += 4) s += Y[(size_t)r * ldy + c];
    }
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < N) {
        const int x = threadIdx.x;
        out[(size_t)blockIdx.y * split_stride + c] = red[0][x] + red[1][x] + red[2][x] + red[3][x];
    }
}

// float4 variant: 16 column groups (64 columns) x 16 row lanes per block, LDS reduction over the row lanes
__global__ __launch_bounds__(256) void colsum_scaled_v4_kernel(const float* __restrict__ Y, int ldy, const float* __restrict__ rs,
                                                              int M, int N, int rows_per_split, float* __restrict__ out,
                                                              int64_t split_stride) {
    __shared__ float4 red[16][17];
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cg * 4;
    const int rbeg = blockIdx.y * rows_per_split, rend = min(M, rbeg + rows_per_split);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < N) {     // N % 4 == 0 on this path
        for (int r = rbeg + rl; r < rend; r += 16) {
            const float4 v = *reinterpret_cast<const float4*>(Y + (size_t)r * ldy + c);
            const float k = rs ? rs[r] : 1.0f;
            s.x += k * v.x; s.y += k * v.y; s.z += k * v.z; s.w += k * v.w;
        }
    }
    red[rl][cg] = s;
    __syncthreads();
    if (rl == 0 && c < N) {
        float4 a = red[0][cg];
#pragma unroll
        for (int j = 1; j < 16; ++j) { const float4 b = red[j][cg]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
        *reinterpret_cast<float4*>(out + (size_t)blockIdx.y * split_stride + c) = a;
    }
}

// several column sums of one shape in ONE launch (blockIdx.z = job): DCN's 2 L cross-parameter gradients were 2 L launches of 6-16 us
// each in a row on the weight-gradient stream (c3: 56 us, the end of the step waited for them); side by side they take the longest one
__global__ __launch_bounds__(256) void colsum_scaled_v4_batch_kernel(ColsumJobs J, int ldy, int M, int N, int rows_per_split, int64_t split_stride) {
    __shared__ float4 red[16][17];
    const float* __restrict__ Y = J.Y[blockIdx.z];
    const float* __restrict__ rs = J.rs[blockIdx.z];
    float* __restrict__ out = J.out[blockIdx.z];
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cg * 4;
    const int rbeg = blockIdx.y * rows_per_split, rend = min(M, rbeg + rows_per_split);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < N) {
        for (int r = rbeg + rl; r < rend; r += 16) {
            const float4 v = *reinterpret_cast<const float4*>(Y + (size_t)r * ldy + c);
            const float k = rs ? rs[r] : 1.0f;
            s.x += k * v.x; s.y += k * v.y; s.z += k * v.z; s.w += k * v.w;
        }
    }
    red[rl][cg] = s;
    __syncthreads();
    if (rl == 0 && c < N) {
        float4 a = red[0][cg];
#pragma unroll
        for (int j = 1; j < 16; ++j) { const float4 b = red[j][cg]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
        *reinterpret_cast<float4*>(out + (size_t)blockIdx.y * split_stride + c) = a;
    }
}

int colsum_partials_batch(const ColsumJobs& J, int ldy, int M, int N, int splits, int64_t split_stride, hipStream_t st) {
    if (N <= 0 || J.n <= 0) return DCTR_OK;
    DCTR_REQUIRE(J.n <= COLSUM_MAX_JOBS, "colsum: %d jobs in one launch (at most %d)", J.n, COLSUM_MAX_JOBS);
    bool v4 = (N % 4 == 0) && (ldy % 4 == 0) && (split_stride % 4 == 0);
    for (int j = 0; j < J.n; ++j)
        v4 = v4 && ((reinterpret_cast<uintptr_t>(J.Y[j]) | reinterpret_cast<uintptr_t>(J.out[j])) & 15) == 0;
    if (!v4) {          // (odd shapes: one launch per job, as before)
        for (int j = 0; j < J.n; ++j) DCTR_TRY(colsum_partials(J.Y[j], ldy, J.rs[j], M, N, splits, J.out[j], split_stride, st));
        return DCTR_OK;
    }
    dim3 grid(ceil_div(N, 64), splits, J.n), block(256);
    colsum_scaled_v4_batch_kernel<<<grid, block, 0, st>>>(J, ldy, M, N, ceil_div(M, splits), split_stride);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

int colsum_partials(const float* Y, int ldy, const float* rs, int M, int N, int splits, float* out,
                    int64_t split_stride, hipStream_t st) {
    if (N <= 0) return DCTR_OK;
    dim3 grid(ceil_div(N, 64), splits), block(256);
    const bool v4 = (N % 4 == 0) && (ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(Y) & 15) == 0) &&
                    ((reinterpret_cast<uintptr_t>(out) & 15) == 0) && (split_stride % 4 == 0);
    if (v4) colsum_scaled_v4_kernel<<<grid, block, 0, st>>>(Y, ldy, rs, M, N, ceil_div(M, splits), out, split_stride);
    else colsum_scaled_kernel<<<grid, block, 0, st>>>(Y, ldy, rs, M, N, ceil_div(M, splits), out, split_stride);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

// ---- backward of a [n] -> 1 output layer in ONE pass over its input x [M,n]:
//   dx[r,c]      = dy[r] * w[c]            (optionally masked by the producing ReLU/dropout: (x>0)/keep)
//   dw_part[s,c] = sum_{r in block s} dy[r] * x[r,c]
//   db_part[s]   = sum_{r in block s} dy[r]            (if requested)
// grid = S row blocks; the optimizer kernel sums the S partial slabs.
__global__ __launch_bounds__(256) void out_layer_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy,
                                                           const float* __restrict__ w, int M, int n, int rows_per_block,
                                                           int masked, float inv_keep, float* __restrict__ dx, int lddx,
                                                           float* __restrict__ dw_part, int64_t dw_stride,
                                                           float* __restrict__ db_part, int64_t db_stride) {
    const int rbeg = blockIdx.x * rows_per_block, rend = min(M, rbeg + rows_per_block);
    for (int c = threadIdx.x; c < n; c += 256) {
        const float wc = w[c];
        float acc = 0.f;
        for (int r = rbeg; r < rend; ++r) {
            const float d = dy[r];
            const float xv = x[(size_t)r * ldx + c];
            acc += d * xv;
            if (dx != nullptr) {
                float g = d * wc;
                if (masked) g = xv > 0.f ? g * inv_keep : 0.f;
                dx[(size_t)r * lddx + c] = g;
            }
        }
        dw_part[(size_t)blockIdx.x * dw_stride + c] = acc;
    }
    if (db_part != nullptr && threadIdx.x < 64) {
        float s = 0.f;
        for (int r = rbeg + threadIdx.x; r < rend; r += 64) s += dy[r];
        s = wave_sum(s);
        if (threadIdx.x == 0) db_part[(size_t)blockIdx.x * db_stride] = s;
    }
}

// float4 variant: CG column groups x 256/CG row lanes per block (CG = the power of two covering n/4, at most 128: a 256-wide layer
// -- AFM's attention_out -- keeps every lane busy with 64 x 4), 4 rows in flight per thread
template <int CG>
__global__ __launch_bounds__(256) void out_layer_bwd_v4_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy,
                                                              const float* __restrict__ w, int M, int n, int rows_per_block,
                                                              int masked, float inv_keep, float* __restrict__ dx, int lddx,
                                                              float* __restrict__ dw_part, int64_t dw_stride,
                                                              float* __restrict__ db_part, int64_t db_stride) {
    constexpr int RL = 256 / CG;
    __shared__ float4 red[RL > 1 ? (RL - 1) * CG : 1];
    const int rbeg = blockIdx.x * rows_per_block, rend = min(M, rbeg + rows_per_block);
    const int cg = threadIdx.x % CG, rl = threadIdx.x / CG;
    for (int c0 = 0; c0 < n; c0 += 4 * CG) {
        const int c = c0 + cg * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < n) {
            const float4 wc = *reinterpret_cast<const float4*>(w + c);
            for (int r = rbeg + rl; r < rend; r += 4 * RL) {
                float4 xv[4];
                float d[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int rr = r + RL * u;
                    d[u] = rr < rend ? dy[rr] : 0.f;
                    xv[u] = rr < rend ? *reinterpret_cast<const float4*>(x + (size_t)rr * ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int rr = r + RL * u;
                    acc.x += d[u] * xv[u].x; acc.y += d[u] * xv[u].y; acc.z += d[u] * xv[u].z; acc.w += d[u] * xv[u].w;
                    if (dx != nullptr && rr < rend) {
                        float4 g = make_float4(d[u] * wc.x, d[u] * wc.y, d[u] * wc.z, d[u] * wc.w);
                        if (masked) {
                            g.x = xv[u].x > 0.f ? g.x * inv_keep : 0.f; g.y = xv[u].y > 0.f ? g.y * inv_keep : 0.f;
                            g.z = xv[u].z > 0.f ? g.z * inv_keep : 0.f; g.w = xv[u].w > 0.f ? g.w * inv_keep : 0.f;
                        }
                        *reinterpret_cast<float4*>(dx + (size_t)rr * lddx + c) = g;
                    }
                }
            }
        }
        if (rl > 0) red[(rl - 1) * CG + cg] = acc;
        __syncthreads();
        if (rl == 0 && c < n) {
#pragma unroll
            for (int k = 0; k + 1 < RL; ++k) {
                const float4 o = red[k * CG + cg];
                acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
            }
            *reinterpret_cast<float4*>(dw_part + (size_t)blockIdx.x * dw_stride + c) = acc;
        }
        __syncthreads();
    }
    if (db_part != nullptr && threadIdx.x < 64) {
        float s = 0.f;
        for (int r = rbeg + threadIdx.x; r < rend; r += 64) s += dy[r];
        s = wave_sum(s);
        if (threadIdx.x == 0) db_part[(size_t)blockIdx.x * db_stride] = s;
    }
}

int out_layer_bwd(const float* x, int ldx, const float* dy, const float* w, int M, int n, int splits, int masked,
                  float keep, float* dx, int lddx, float* dw_part, int64_t dw_stride, float* db_part, int64_t db_stride,
                  hipStream_t st) {
    if (M <= 0 || n <= 0) return DCTR_OK;
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool v4 = n % 4 == 0 && ldx % 4 == 0 && (dx == nullptr || lddx % 4 == 0) && dw_stride % 4 == 0 && al(x) && al(w) && al(dw_part) &&
                    (dx == nullptr || al(dx));
    if (v4) {
        auto kern = n > 256 ? out_layer_bwd_v4_kernel<128> : n > 128 ? out_layer_bwd_v4_kernel<64> : n > 64 ? out_layer_bwd_v4_kernel<32> : out_layer_bwd_v4_kernel<16>;
        kern<<<splits, 256, 0, st>>>(x, ldx, dy, w, M, n, ceil_div(M, splits), masked, 1.0f / keep, dx, lddx, dw_part, dw_stride, db_part, db_stride);
    } else
        out_layer_bwd_kernel<<<splits, 256, 0, st>>>(x, ldx, dy, w, M, n, ceil_div(M, splits), masked, 1.0f / keep, dx, lddx,
                                                      dw_part, dw_stride, db_part, db_stride);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

// ---- K7 loss head (DeepFM.py:174-176,188) -----------------------------------------------------
__global__ __launch_bounds__(256) void loss_head_kernel(const float* __restrict__ bias, const float* __restrict__ yw,
                                                       const float* __restrict__ yv, const float* __restrict__ yd,
                                                       const float* __restrict__ labels, int B, float inv_batch,
                                                       float* __restrict__ y_out, float* __restrict__ prob,
                                                       float* __restrict__ dy, float* __restrict__ loss_sum) {
    __shared__ float red[4];
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    float l = 0.f;
    if (b < B) {
        float y = 0.f;
        if (bias) y += bias[0];
        if (yw) y += yw[b];
        if (yv) y += yv[b];
        if (yd) y += yd[b];
        // sigmoid via exp(-|y|): no overflow for large |y|
        const float en = __expf(-fabsf(y));
        const float p = (y >= 0.f) ? 1.0f / (1.0f + en) : en / (1.0f + en);
        if (y_out) y_out[b] = y;
        if (prob) prob[b] = p;
        if (labels) {
            const float z = labels[b];
            // max(x,0) - x z + log1p(exp(-|x|))   [TF-1.4 sigmoid_cross_entropy_with_logits]
            l = fmaxf(y, 0.f) - y * z + log1pf(expf(-fabsf(y)));
            if (dy) dy[b] = (p - z) * inv_batch;
        }
    }
    l = wave_sum(l);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = l;
    __syncthreads();
    if (threadIdx.x == 0 && loss_sum != nullptr && labels != nullptr) atomicAdd(loss_sum, red[0] + red[1] + red[2] + red[3]);
}

// ---- fused output layer + loss head: y_d = x1[b,:].w1 (+ x2[b,:].w2) + b_out, then the loss head of DeepFM.py:174-176,188.
// One wave per example (4 per block); the xent sum goes to one of SUMSQ_SHARDS shards (a thousand blocks on one word would
// serialise).  Replaces three launches (rowdot, rowdot, loss_head) on the critical path between forward and backward.
__global__ __launch_bounds__(256) void head_fused_kernel(const float* __restrict__ x1, int ld1, const float* __restrict__ w1, int n1,
                                                        const float* __restrict__ x2, int ld2, const float* __restrict__ w2, int n2,
                                                        const float* __restrict__ b_out, const float* __restrict__ bias,
                                                        const float* __restrict__ yw, const float* __restrict__ yv,
                                                        const float* __restrict__ labels, int B, float inv_batch,
                                                        float* __restrict__ yd_out, float* __restrict__ y_out, float* __restrict__ prob,
                                                        float* __restrict__ dy, float* __restrict__ loss_shards) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    float l = 0.f;
    if (b < B) {
        float s = 0.f;
        const float* r1 = x1 + (size_t)b * ld1;
        for (int j = lane; j < n1; j += 64) s += r1[j] * w1[j];
        if (x2 != nullptr) {
            const float* r2 = x2 + (size_t)b * ld2;
            for (int j = lane; j < n2; j += 64) s += r2[j] * w2[j];
        }
        s = wave_sum(s);
        if (lane == 0) {
            const float ydv = s + (b_out ? b_out[0] : 0.f);
            float y = ydv;
            if (bias) y += bias[0];
            if (yw) y += yw[b];
            if (yv) y += yv[b];
            const float en = __expf(-fabsf(y));
            const float p = (y >= 0.f) ? 1.0f / (1.0f + en) : en / (1.0f + en);
            yd_out[b] = ydv;
            y_out[b] = y;
            prob[b] = p;
            if (labels) {
                const float z = labels[b];
                l = fmaxf(y, 0.f) - y * z + log1pf(expf(-fabsf(y)));
                if (dy) dy[b] = (p - z) * inv_batch;
            }
        }
    }
    if (lane == 0) red[wave] = l;
    __syncthreads();
    if (threadIdx.x == 0 && loss_shards != nullptr && labels != nullptr)
        atomicAdd(loss_shards + (blockIdx.x & (SUMSQ_SHARDS - 1)), red[0] + red[1] + red[2] + red[3]);
}

int head_fused(const float* x1, int ld1, const float* w1, int n1, const float* x2, int ld2, const float* w2, int n2,
               const float* b_out, const float* bias, const float* yw, const float* yv, const float* labels, int B, float inv_batch,
               float* yd, float* y, float* prob, float* dy, float* loss_shards, hipStream_t st) {
    if (B <= 0) return DCTR_OK;
    head_fused_kernel<<<ceil_div(B, 4), 256, 0, st>>>(x1, ld1, w1, n1, x2, ld2, w2, n2, b_out, bias, yw, yv, labels, B, inv_batch, yd, y,
                                                      prob, dy, loss_shards);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

// ---- TRAIN: output layer forward + loss head + output layer backward in ONE launch.  A block owns `rows_per_block`
// examples: phase 1 (one wave per example) computes logit / prob / xent / dy into LDS, phase 2 (128 float4 column groups x 2
// row lanes) re-reads the block's rows of x (L2-resident) for dX = dy (x) w [masked] and the dW partial slab.
struct HeadSeg { const float* x; int ld; const float* w; int n; int masked; float* dx; int lddx; };

__global__ __launch_bounds__(256) void head_out_bwd_kernel(HeadSeg s1, HeadSeg s2, const float* __restrict__ b_out,
                                                          const float* __restrict__ bias, const float* __restrict__ yw,
                                                          const float* __restrict__ yv, const float* __restrict__ labels, int B,
                                                          float inv_batch, float inv_keep, int rows_per_block,
                                                          float* __restrict__ yd_out, float* __restrict__ y_out, float* __restrict__ prob,
                                                          float* __restrict__ dy_out, float* __restrict__ loss_shards,
                                                          float* __restrict__ dw_part, int64_t dw_stride, float* __restrict__ db_part,
                                                          int64_t db_stride) {
    __shared__ float dys[64];
    __shared__ float4 red[128];
    __shared__ float lred[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rbeg = blockIdx.x * rows_per_block, rend = min(B, rbeg + rows_per_block);
    float lsum = 0.f;
    // phase 1: 8 lanes per example, 32 examples at a time, float4 loads -> all of the block's dot products in flight together
    for (int rb = rbeg; rb < rend; rb += 32) {
        const int r = rb + (threadIdx.x >> 3), sub = threadIdx.x & 7;
        float s = 0.f;
        if (r < rend) {
            const float* r1 = s1.x + (size_t)r * s1.ld;
            for (int j = sub * 4; j < s1.n; j += 32) {
                const float4 a = *reinterpret_cast<const float4*>(r1 + j);
                const float4 w = *reinterpret_cast<const float4*>(s1.w + j);
                s += a.x * w.x + a.y * w.y + a.z * w.z + a.w * w.w;
            }
            if (s2.x != nullptr) {
                const float* r2 = s2.x + (size_t)r * s2.ld;
                for (int j = sub * 4; j < s2.n; j += 32) {
                    const float4 a = *reinterpret_cast<const float4*>(r2 + j);
                    const float4 w = *reinterpret_cast<const float4*>(s2.w + j);
                    s += a.x * w.x + a.y * w.y + a.z * w.z + a.w * w.w;
                }
            }
        }
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
        if (r < rend && sub == 0) {
            const float ydv = s + (b_out ? b_out[0] : 0.f);
            float y = ydv;
            if (bias) y += bias[0];
            if (yw) y += yw[r];
            if (yv) y += yv[r];
            const float en = __expf(-fabsf(y));
            const float p = (y >= 0.f) ? 1.0f / (1.0f + en) : en / (1.0f + en);
            const float z = labels[r];
            lsum += fmaxf(y, 0.f) - y * z + log1pf(expf(-fabsf(y)));
            const float d = (p - z) * inv_batch;
            yd_out[r] = ydv; y_out[r] = y; prob[r] = p; dy_out[r] = d;
            dys[r - rbeg] = d;
        }
    }
    lsum = wave_sum(lsum);
    if (lane == 0) lred[wave] = lsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(loss_shards + (blockIdx.x & (SUMSQ_SHARDS - 1)), lred[0] + lred[1] + lred[2] + lred[3]);
        float s = 0.f;
        for (int r = 0; r < rend - rbeg; ++r) s += dys[r];
        db_part[(size_t)blockIdx.x * db_stride] = s;
    }
    const int cg = threadIdx.x & 127, rl = threadIdx.x >> 7;
    int col_off = 0;
#pragma unroll
    for (int seg = 0; seg < 2; ++seg) {
        const HeadSeg& g = seg == 0 ? s1 : s2;
        if (g.x == nullptr) break;
        for (int c0 = 0; c0 < g.n; c0 += 512) {
            const int c = c0 + cg * 4;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < g.n) {
                const float4 wc = *reinterpret_cast<const float4*>(g.w + c);
                for (int r = rbeg + rl; r < rend; r += 2) {
                    const float d = dys[r - rbeg];
                    const float4 xv = *reinterpret_cast<const float4*>(g.x + (size_t)r * g.ld + c);
                    acc.x += d * xv.x; acc.y += d * xv.y; acc.z += d * xv.z; acc.w += d * xv.w;
                    float4 o = make_float4(d * wc.x, d * wc.y, d * wc.z, d * wc.w);
                    if (g.masked) {
                        o.x = xv.x > 0.f ? o.x * inv_keep : 0.f; o.y = xv.y > 0.f ? o.y * inv_keep : 0.f;
                        o.z = xv.z > 0.f ? o.z * inv_keep : 0.f; o.w = xv.w > 0.f ? o.w * inv_keep : 0.f;
                    }
                    *reinterpret_cast<float4*>(g.dx + (size_t)r * g.lddx + c) = o;
                }
            }
            if (rl == 1) red[cg] = acc;
            __syncthreads();
            if (rl == 0 && c < g.n) {
                const float4 o = red[cg];
                *reinterpret_cast<float4*>(dw_part + (size_t)blockIdx.x * dw_stride + col_off + c) =
                    make_float4(acc.x + o.x, acc.y + o.y, acc.z + o.z, acc.w + o.w);
            }
            __syncthreads();
        }
        col_off += g.n;
    }
}

// Same contract in ONE pass over x (the kernel above reads every row twice and gives a block of 256 threads 32 rows: 23 us for
// 13 MB).  A block of 1024 threads takes 32 rows at a time, 32 lanes per row; a lane keeps its NI float4 pieces of the row and of
// w in registers, so the logit's dot product, dX = dy (x) w [masked] and the lane's share of dW = x^T dy all come from the same
// loads.  A block (4 waves: light enough to be placed beside whatever else the step has on the chip) walks its rows 8 at a time,
// dW accumulating in registers; it meets across the block in two steps: the two rows of a wave by a lane swap, the 4 waves
// through LDS.  Round 5: the loads of U trips (U x 8 rows: all 32 rows of a c2 block) are requested together before the first trip's
// arithmetic -- see the loop.
template <int NI>
__global__ __launch_bounds__(256) void head_out_bwd_rows_kernel(HeadSeg s1, HeadSeg s2, const float* __restrict__ b_out,
                                                                const float* __restrict__ bias, const float* __restrict__ yw,
                                                                const float* __restrict__ yv, const float* __restrict__ labels, int B,
                                                                float inv_batch, float inv_keep, int rows_per_block,
                                                                float* __restrict__ yd_out, float* __restrict__ y_out,
                                                                float* __restrict__ prob, float* __restrict__ dy_out,
                                                                float* __restrict__ loss_shards, float* __restrict__ dw_part,
                                                                int64_t dw_stride, float* __restrict__ db_part, int64_t db_stride) {
    constexpr int NW = 4;                   // waves per block: 2 rows each
    __shared__ float4 wred[NW][NI * 32];
    __shared__ float lred[NW], dred[NW];
    const int t = threadIdx.x, l = t & 31, rr = t >> 5, wave = t >> 6;
    const int rbeg = blockIdx.x * rows_per_block, rend = min(B, rbeg + rows_per_block);
    const int n41 = s1.n >> 2, n4 = n41 + (s2.x != nullptr ? (s2.n >> 2) : 0);
    float4 wv[NI], dw[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int f = l + 32 * i;
        wv[i] = dw[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < n4) wv[i] = f < n41 ? *reinterpret_cast<const float4*>(s1.w + 4 * f) : *reinterpret_cast<const float4*>(s2.w + 4 * (f - n41));
    }
    const float b0 = (b_out ? b_out[0] : 0.f), g0 = (bias ? bias[0] : 0.f);
    float lsum = 0.f, dsum = 0.f;
    // Everything the block's next U trips (U x 8 rows) need from memory -- each row's NI pieces of x and its three per-example scalars
    // (y_w, y_v, label) -- is requested TOGETHER, in front of the first trip's arithmetic.  (The scalars used to be fetched where they
    // are used, behind the dot product's shuffles, one dependent load after the other, and every trip waited for its own row first:
    // sixteen memory latencies in a row for a block's 32 rows -- 12.3 us in the step for 13 MB.)  The trips' order, and with it every
    // sum, is what it was.
    constexpr int U = NI <= 4 ? 4 : 2;
    for (int rb = rbeg; rb < rend; rb += 2 * NW * U) {
        float4 xv[U][NI];
        float ywv[U], yvv[U], zv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = rb + 2 * NW * u + rr;
            const bool valid = r < rend;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int f = l + 32 * i;
                xv[u][i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (valid && f < n4)
                    xv[u][i] = f < n41 ? *reinterpret_cast<const float4*>(s1.x + (size_t)r * s1.ld + 4 * f)
                                       : *reinterpret_cast<const float4*>(s2.x + (size_t)r * s2.ld + 4 * (f - n41));
            }
            ywv[u] = (valid && yw) ? yw[r] : 0.f;
            yvv[u] = (valid && yv) ? yv[r] : 0.f;
            zv[u] = valid ? labels[r] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = rb + 2 * NW * u + rr;
            const bool valid = r < rend;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NI; ++i) s += xv[u][i].x * wv[i].x + xv[u][i].y * wv[i].y + xv[u][i].z * wv[i].z + xv[u][i].w * wv[i].w;
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8); s += __shfl_xor(s, 16);
            float d = 0.f;
            if (valid) {
                const float ydv = s + b0;
                float y = ydv + g0;
                if (yw) y += ywv[u];
                if (yv) y += yvv[u];
                const float en = __expf(-fabsf(y));
                const float p = (y >= 0.f) ? 1.0f / (1.0f + en) : en / (1.0f + en);
                const float z = zv[u];
                d = (p - z) * inv_batch;
                if (l == 0) {
                    lsum += fmaxf(y, 0.f) - y * z + log1pf(expf(-fabsf(y)));
                    dsum += d;
                    yd_out[r] = ydv; y_out[r] = y; prob[r] = p; dy_out[r] = d;
                }
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int f = l + 32 * i;
                if (valid && f < n4) {
                    const bool first = f < n41;
                    const HeadSeg& g = first ? s1 : s2;
                    float4 o = make_float4(d * wv[i].x, d * wv[i].y, d * wv[i].z, d * wv[i].w);
                    if (g.masked) {
                        o.x = xv[u][i].x > 0.f ? o.x * inv_keep : 0.f; o.y = xv[u][i].y > 0.f ? o.y * inv_keep : 0.f;
                        o.z = xv[u][i].z > 0.f ? o.z * inv_keep : 0.f; o.w = xv[u][i].w > 0.f ? o.w * inv_keep : 0.f;
                    }
                    *reinterpret_cast<float4*>(g.dx + (size_t)r * g.lddx + 4 * (first ? f : f - n41)) = o;
                    dw[i].x += d * xv[u][i].x; dw[i].y += d * xv[u][i].y; dw[i].z += d * xv[u][i].z; dw[i].w += d * xv[u][i].w;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {          // the wave's two rows: lanes l and l + 32
        dw[i].x += __shfl_xor(dw[i].x, 32); dw[i].y += __shfl_xor(dw[i].y, 32);
        dw[i].z += __shfl_xor(dw[i].z, 32); dw[i].w += __shfl_xor(dw[i].w, 32);
        if ((t & 63) < 32) wred[wave][l + 32 * i] = dw[i];
    }
    lsum += __shfl_xor(lsum, 32);
    dsum += __shfl_xor(dsum, 32);
    if ((t & 63) == 0) { lred[wave] = lsum; dred[wave] = dsum; }
    __syncthreads();
    for (int f = t; f < n4; f += 64 * NW) {
        const int t = f;
        float4 a = wred[0][t];
#pragma unroll
        for (int wv_ = 1; wv_ < NW; ++wv_) { const float4 o = wred[wv_][t]; a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w; }
        *reinterpret_cast<float4*>(dw_part + (size_t)blockIdx.x * dw_stride + 4 * t) = a;
    }
    if (t == 0) {
        float ls = 0.f, ds = 0.f;
#pragma unroll
        for (int wv_ = 0; wv_ < NW; ++wv_) { ls += lred[wv_]; ds += dred[wv_]; }
        atomicAdd(loss_shards + (blockIdx.x & (SUMSQ_SHARDS - 1)), ls);
        db_part[(size_t)blockIdx.x * db_stride] = ds;
    }
}

// returns DCTR_ERR_UNSUPPORTED (without setting an error) when the shapes do not meet the float4 alignment rules
int head_out_bwd(const float* x1, int ld1, const float* w1, int n1, int masked1, float* dx1, int lddx1,
                 const float* x2, int ld2, const float* w2, int n2, int masked2, float* dx2, int lddx2,
                 const float* b_out, const float* bias, const float* yw, const float* yv, const float* labels, int B, float inv_batch,
                 float keep, int splits, float* yd, float* y, float* prob, float* dy, float* loss_shards, float* dw_part,
                 int64_t dw_stride, float* db_part, int64_t db_stride, hipStream_t st) {
    auto ok = [](const void* p, int ld, int n) { return p == nullptr || ((reinterpret_cast<uintptr_t>(p) & 15) == 0 && ld % 4 == 0 && n % 4 == 0); };
    const int rpb = ceil_div(B, splits);
    if (B <= 0 || rpb > 64 || !ok(x1, ld1, n1) || !ok(dx1, lddx1, n1) || !ok(w1, 4, n1) || !ok(x2, ld2, n2) || !ok(dx2, lddx2, n2) ||
        !ok(w2, 4, n2) || dw_stride % 4 != 0 || (reinterpret_cast<uintptr_t>(dw_part) & 15) != 0)
        return DCTR_ERR_UNSUPPORTED;
    HeadSeg s1{x1, ld1, w1, n1, masked1, dx1, lddx1}, s2{x2, ld2, w2, n2, masked2, dx2, lddx2};
    static const bool two_pass = getenv("DCTR_HEAD_TWO_PASS") != nullptr;       // A/B knob: the 256-thread two-pass kernel
    const int n4 = (n1 + (x2 ? n2 : 0)) / 4;
    if (!two_pass && n4 <= 256) {
        hipEvent_t stop = take_stop_event();         // (the engine's record behind the head rides on this launch: common.h)
        if (stop != nullptr && n4 <= 128)
            hipExtLaunchKernelGGL(head_out_bwd_rows_kernel<4>, dim3(splits), dim3(256), 0u, st, nullptr, stop, 0, s1, s2, b_out, bias, yw, yv, labels, B, inv_batch,
                                  1.0f / keep, rpb, yd, y, prob, dy, loss_shards, dw_part, dw_stride, db_part, db_stride);
        else if (stop != nullptr)
            hipExtLaunchKernelGGL(head_out_bwd_rows_kernel<8>, dim3(splits), dim3(256), 0u, st, nullptr, stop, 0, s1, s2, b_out, bias, yw, yv, labels, B, inv_batch,
                                  1.0f / keep, rpb, yd, y, prob, dy, loss_shards, dw_part, dw_stride, db_part, db_stride);
        else if (n4 <= 128)
            head_out_bwd_rows_kernel<4><<<splits, 256, 0, st>>>(s1, s2, b_out, bias, yw, yv, labels, B, inv_batch, 1.0f / keep, rpb, yd, y, prob,
                                                                 dy, loss_shards, dw_part, dw_stride, db_part, db_stride);
        else
            head_out_bwd_rows_kernel<8><<<splits, 256, 0, st>>>(s1, s2, b_out, bias, yw, yv, labels, B, inv_batch, 1.0f / keep, rpb, yd, y, prob,
                                                                 dy, loss_shards, dw_part, dw_stride, db_part, db_stride);
        DCTR_LAUNCH_CHECK();
        return DCTR_OK;
    }
    head_out_bwd_kernel<<<splits, 256, 0, st>>>(s1, s2, b_out, bias, yw, yv, labels, B, inv_batch, 1.0f / keep, rpb, yd, y, prob, dy,
                                                loss_shards, dw_part, dw_stride, db_part, db_stride);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

int loss_head(const float* bias, const float* yw, const float* yv, const float* yd, const float* labels, int B,
              float inv_batch, float* y, float* prob, float* dy, float* loss_sum, hipStream_t st) {
    if (B <= 0) return DCTR_OK;
    loss_head_kernel<<<ceil_div(B, 256), 256, 0, st>>>(bias, yw, yv, yd, labels, B, inv_batch, y, prob, dy, loss_sum);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

// ---- K9 optimizers (DeepFM.py:204-211), TF-1.4 update rules ----------------------------------

// dense arena: OPT_BLOCK elements per block; per-block metadata says where the gradient partials live
template <int KIND>
__global__ __launch_bounds__(256) void opt_dense_kernel(const Hyper* __restrict__ hdev, Hyper hval,
                                                       float4* __restrict__ theta, float4* __restrict__ s0,
                                                       float4* __restrict__ s1, const float* __restrict__ parts,
                                                       const OptBlockMeta* __restrict__ meta, float4* __restrict__ gout,
                                                       int apply, float* __restrict__ sumsq) {
    const Hyper h = load_hyper(hdev, hval);
    const OptBlockMeta m = meta[blockIdx.x];
    if (m.n_part < 0) return;       // not trainable (BN moving statistics)
    const size_t i4 = (size_t)blockIdx.x * (OPT_BLOCK / 4) + threadIdx.x;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* p = reinterpret_cast<const float4*>(parts + m.part_off) + threadIdx.x;
    const size_t ps4 = (size_t)(m.part_stride / 4);
    // eight slab loads in flight per round: under a concurrent table-optimizer stream a load takes microseconds, and a
    // one-at-a-time loop made this kernel 15 dependent round trips long
    for (int s0 = 0; s0 < m.n_part; s0 += 8) {
        float4 q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = (s0 + j < m.n_part) ? p[(size_t)(s0 + j) * ps4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 8; ++j) { g.x += q[j].x; g.y += q[j].y; g.z += q[j].z; g.w += q[j].w; }
    }
    if (gout != nullptr) gout[i4] = g;       // reduced gradient (for the all-reduce in the multi-GPU path)
    if (!apply) return;
    float4 th = theta[i4];
    if (m.l2 != 0.f) {
        g.x += m.l2 * th.x; g.y += m.l2 * th.y; g.z += m.l2 * th.z; g.w += m.l2 * th.w;
        if (sumsq != nullptr) {   // l2_loss term of the reported loss (DCN.py:199), pre-update values; padding is zero
            float sq = wave_sum(th.x * th.x + th.y * th.y + th.z * th.z + th.w * th.w);
            if ((threadIdx.x & 63) == 0) atomicAdd(sumsq + (blockIdx.x & (SUMSQ_SHARDS - 1)), sq);
        }
    }
    float4 a = s0[i4];
    float4 b = (KIND == DCTR_OPT_ADAM || KIND == DCTR_OPT_FTRL) ? s1[i4] : make_float4(0.f, 0.f, 0.f, 0.f);
    opt_update<true>(KIND, h, th.x, a.x, b.x, g.x);
    opt_update<true>(KIND, h, th.y, a.y, b.y, g.y);
    opt_update<true>(KIND, h, th.z, a.z, b.z, g.z);
    opt_update<true>(KIND, h, th.w, a.w, b.w, g.w);
    theta[i4] = th;
    s0[i4] = a;
    if (KIND == DCTR_OPT_ADAM || KIND == DCTR_OPT_FTRL) s1[i4] = b;
}

int opt_dense_arena(int kind, const Hyper* hdev, const Hyper& hval, float* theta, float* s0, float* s1,
                    const float* parts, const OptBlockMeta* meta, int n_blocks, float* gout, int apply, float* sumsq,
                    hipStream_t st) {
    if (n_blocks <= 0) return DCTR_OK;
    float4* t4 = reinterpret_cast<float4*>(theta);
    float4* a4 = reinterpret_cast<float4*>(s0);
    float4* b4 = reinterpret_cast<float4*>(s1);
    float4* g4 = reinterpret_cast<float4*>(gout);
    switch (kind) {
#define DCTR_K(KD) case KD: opt_dense_kernel<KD><<<n_blocks, 256, 0, st>>>(hdev, hval, t4, a4, b4, parts, meta, g4, apply, sumsq); break
        DCTR_K(DCTR_OPT_ADAM); DCTR_K(DCTR_OPT_ADAGRAD); DCTR_K(DCTR_OPT_MOMENTUM); DCTR_K(DCTR_OPT_FTRL);
#undef DCTR_K
        default: set_error("unknown optimizer kind %d", kind); return DCTR_ERR_INVALID_ARG;
    }
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

// plain flat variant for the op-level C ABI
template <int KIND>
__global__ __launch_bounds__(256) void opt_flat_kernel(const Hyper* __restrict__ hdev, Hyper hval, float* __restrict__ theta,
                                                      float* __restrict__ s0, float* __restrict__ s1,
                                                      const float* __restrict__ grad, int n_part, int64_t part_stride,
                                                      int64_t n, float l2) {
    const Hyper h = load_hyper(hdev, hval);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float g = 0.f;
        for (int s = 0; s < n_part; ++s) g += grad[(size_t)s * part_stride + i];
        float th = theta[i];
        g += l2 * th;
        float a = s0[i];
        float b = (KIND == DCTR_OPT_ADAM || KIND == DCTR_OPT_FTRL) ? s1[i] : 0.f;
        opt_update<true>(KIND, h, th, a, b, g);
        theta[i] = th;
        s0[i] = a;
        if (KIND == DCTR_OPT_ADAM || KIND == DCTR_OPT_FTRL) s1[i] = b;
    }
}

int opt_dense_flat(int kind, const Hyper* hdev, const Hyper& hval, float* theta, float* s0, float* s1, const float* grad,
                   int n_part, int64_t part_stride, int64_t n, float l2, hipStream_t st) {
    if (n <= 0) return DCTR_OK;
    const int grid = (int)std::min<int64_t>(ceil_div(n, 256), 4096);
    switch (kind) {
#define DCTR_K(KD) case KD: opt_flat_kernel<KD><<<grid, 256, 0, st>>>(hdev, hval, theta, s0, s1, grad, n_part, part_stride, n, l2); break
        DCTR_K(DCTR_OPT_ADAM); DCTR_K(DCTR_OPT_ADAGRAD); DCTR_K(DCTR_OPT_MOMENTUM); DCTR_K(DCTR_OPT_FTRL);
#undef DCTR_K
        default: set_error("unknown optimizer kind %d", kind); return DCTR_ERR_INVALID_ARG;
    }
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

// tables.  DENSE: stream every row; grad = l2*theta + (slot[r] ? compact[slot[r]-1] : 0)  -- what TF does when the
// IndexedSlices gradient meets the dense l2_loss gradient (DeepFM.py:189-190,213).  Also accumulates
// sum(theta_old^2) so the l2 part of the reported loss is free.  TOUCHED: only rows uniq[0:U).
__device__ __forceinline__ float4 ntload4(const float4* p) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    const v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void ntstore4(float4* p, float4 x) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    v4 v; v.x = x.x; v.y = x.y; v.z = x.z; v.w = x.w;
    __builtin_nontemporal_store(v, reinterpret_cast<v4*>(p));
}

template <int KIND, int KQ, bool DENSE>
__global__ __launch_bounds__(256) void opt_table_kernel(const Hyper* __restrict__ hdev, Hyper hval, int64_t rows,
                                                       float4* __restrict__ emb, float4* __restrict__ s0,
                                                       float4* __restrict__ s1, float* __restrict__ lin,
                                                       float* __restrict__ l0, float* __restrict__ l1,
                                                       const int32_t* __restrict__ slot, const int32_t* __restrict__ uniq,
                                                       const int32_t* __restrict__ counters, const float4* __restrict__ gemb,
                                                       const float* __restrict__ glin, float l2, float* __restrict__ sumsq_emb,
                                                       float* __restrict__ sumsq_lin, int nt, int untouched_only, int ld4, int lin_ld) {
    // KQ lanes per row (one float4 each); lane kq == 0 also steps the row's linear weight (same slot word, one launch)
    const Hyper h = load_hyper(hdev, hval);
    const int64_t n_items = DENSE ? rows : (int64_t)counters[0];
    float sq = 0.f, sql = 0.f;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_items * KQ;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t item = t / KQ;
        const int kq = (int)(t - item * KQ);
        int64_t r;
        int u;
        if (DENSE) { r = item; u = slot[r] - 1; } else { u = (int)item; r = uniq[u]; }
        if (DENSE && untouched_only && u >= 0) continue;    // the batch's rows are stepped later, once their gradients exist
        const size_t i4 = (size_t)r * ld4 + kq, il = (size_t)r * lin_ld;
        float4 th, a, b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (nt) {
            th = ntload4(emb + i4); a = ntload4(s0 + i4);
            if (KIND == DCTR_OPT_ADAM || KIND == DCTR_OPT_FTRL) b = ntload4(s1 + i4);
        } else {
            th = emb[i4]; a = s0[i4];
            if (KIND == DCTR_OPT_ADAM || KIND == DCTR_OPT_FTRL) b = s1[i4];
        }
        sq += th.x * th.x + th.y * th.y + th.z * th.z + th.w * th.w;
        float4 g = make_float4(l2 * th.x, l2 * th.y, l2 * th.z, l2 * th.w);
        if (u >= 0) {
            const float4 q = gemb[(size_t)u * KQ + kq];
            g.x += q.x; g.y += q.y; g.z += q.z; g.w += q.w;
        }
        opt_update(KIND, h, th.x, a.x, b.x, g.x);
        opt_update(KIND, h, th.y, a.y, b.y, g.y);
        opt_update(KIND, h, th.z, a.z, b.z, g.z);
        opt_update(KIND, h, th.w, a.w, b.w, g.w);
        if (nt) {
            ntstore4(emb + i4, th); ntstore4(s0 + i4, a);
            if (KIND == DCTR_OPT_ADAM || KIND == DCTR_OPT_FTRL) ntstore4(s1 + i4, b);
        } else {
            emb[i4] = th; s0[i4] = a;
            if (KIND == DCTR_OPT_ADAM || KIND == DCTR_OPT_FTRL) s1[i4] = b;
        }
        if (kq == 0 && lin != nullptr) {
            float lt = lin[il];
            float la = l0[il];
            float lb = (KIND == DCTR_OPT_ADAM || KIND == DCTR_OPT_FTRL) ? l1[il] : 0.f;
            sql += lt * lt;
            float lg = l2 * lt;
            if (u >= 0) lg += glin[u];
            opt_update(KIND, h, lt, la, lb, lg);
            lin[il] = lt;
            l0[il] = la;
            if (KIND == DCTR_OPT_ADAM || KIND == DCTR_OPT_FTRL) l1[il] = lb;
        }
    }
    if (sumsq_emb != nullptr) {
        __shared__ float red[2][4];
        sq = wave_sum(sq);
        sql = wave_sum(sql);
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sq; red[1][threadIdx.x >> 6] = sql; }
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(sumsq_emb + (blockIdx.x & (SUMSQ_SHARDS - 1)), red[0][0] + red[0][1] + red[0][2] + red[0][3]);
            if (sumsq_lin != nullptr) atomicAdd(sumsq_lin + (blockIdx.x & (SUMSQ_SHARDS - 1)), red[1][0] + red[1][1] + red[1][2] + red[1][3]);
        }
    }
}

// Background pass over the rows the batch does NOT touch (gradient = l2*theta), meant to run UNDER the MLP GEMMs: a small grid
// (a wave or two per SIMD, so the GEMM blocks still find their wave slots) with UNR independent row pieces in flight per lane
// to keep the HBM pipe full from few waves.
template <int KIND, int KQ, int UNR>
// (5 waves per SIMD asked for = at most 96 VGPRs: two of its waves and one block of the direct GEMM kernels, up to 320 VGPRs, then
// share a SIMD -- a GEMM block that finds no room beside this background pass has to wait for its blocks to END)
__global__ __launch_bounds__(256, 5) void opt_table_untouched_kernel(const Hyper* __restrict__ hdev, Hyper hval, int64_t rows,
                                                                 float4* __restrict__ emb, float4* __restrict__ s0,
                                                                 float4* __restrict__ s1, const int32_t* __restrict__ slot, float l2,
                                                                 float* __restrict__ sumsq_emb) {
    const Hyper h = load_hyper(hdev, hval);
    constexpr bool TWO = (KIND == DCTR_OPT_ADAM || KIND == DCTR_OPT_FTRL);
    const int64_t n = rows * KQ;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float sq = 0.f;
    for (int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t0 < n; t0 += stride * UNR) {
        float4 th[UNR], a[UNR], b[UNR];
        bool live[UNR];
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            const int64_t t = t0 + j * stride;
            live[j] = t < n && slot[t / KQ] == 0;
            if (live[j]) {
                th[j] = emb[t]; a[j] = s0[t];
                b[j] = TWO ? s1[t] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            if (!live[j]) continue;
            const int64_t t = t0 + j * stride;
            sq += th[j].x * th[j].x + th[j].y * th[j].y + th[j].z * th[j].z + th[j].w * th[j].w;
            opt_update(KIND, h, th[j].x, a[j].x, b[j].x, l2 * th[j].x);
            opt_update(KIND, h, th[j].y, a[j].y, b[j].y, l2 * th[j].y);
            opt_update(KIND, h, th[j].z, a[j].z, b[j].z, l2 * th[j].z);
            opt_update(KIND, h, th[j].w, a[j].w, b[j].w, l2 * th[j].w);
            emb[t] = th[j]; s0[t] = a[j];
            if (TWO) s1[t] = b[j];
        }
    }
    if (sumsq_emb != nullptr) {
        __shared__ float red[4];
        sq = wave_sum(sq);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(sumsq_emb + (blockIdx.x & (SUMSQ_SHARDS - 1)), red[0] + red[1] + red[2] + red[3]);
    }
}

// dense-exact step of the [rows] linear table, 4 rows per thread (16-byte accesses on all seven streams)
template <int KIND>
__global__ __launch_bounds__(256) void opt_lin_dense_kernel(const Hyper* __restrict__ hdev, Hyper hval, int64_t rows,
                                                           float* __restrict__ lin, float* __restrict__ l0, float* __restrict__ l1,
                                                           const int32_t* __restrict__ slot, const float* __restrict__ glin, float l2,
                                                           float* __restrict__ sumsq, int untouched_only) {
    const Hyper h = load_hyper(hdev, hval);
    const int64_t n4 = rows / 4;
    float sq = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 th = reinterpret_cast<float4*>(lin)[i];
        float4 a = reinterpret_cast<float4*>(l0)[i];
        float4 b = (KIND == DCTR_OPT_ADAM || KIND == DCTR_OPT_FTRL) ? reinterpret_cast<float4*>(l1)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        const int4 sl = reinterpret_cast<const int4*>(slot)[i];
        if (untouched_only) {
            // rows of the batch keep their values (written back unchanged: the touched pass is ordered after this one)
            if (!sl.x) { sq += th.x * th.x; opt_update(KIND, h, th.x, a.x, b.x, l2 * th.x); }
            if (!sl.y) { sq += th.y * th.y; opt_update(KIND, h, th.y, a.y, b.y, l2 * th.y); }
            if (!sl.z) { sq += th.z * th.z; opt_update(KIND, h, th.z, a.z, b.z, l2 * th.z); }
            if (!sl.w) { sq += th.w * th.w; opt_update(KIND, h, th.w, a.w, b.w, l2 * th.w); }
        } else {
            sq += th.x * th.x + th.y * th.y + th.z * th.z + th.w * th.w;
            float4 g = make_float4(l2 * th.x, l2 * th.y, l2 * th.z, l2 * th.w);
            if (sl.x) g.x += glin[sl.x - 1];
            if (sl.y) g.y += glin[sl.y - 1];
            if (sl.z) g.z += glin[sl.z - 1];
            if (sl.w) g.w += glin[sl.w - 1];
            opt_update(KIND, h, th.x, a.x, b.x, g.x);
            opt_update(KIND, h, th.y, a.y, b.y, g.y);
            opt_update(KIND, h, th.z, a.z, b.z, g.z);
            opt_update(KIND, h, th.w, a.w, b.w, g.w);
        }
        reinterpret_cast<float4*>(lin)[i] = th;
        reinterpret_cast<float4*>(l0)[i] = a;
        if (KIND == DCTR_OPT_ADAM || KIND == DCTR_OPT_FTRL) reinterpret_cast<float4*>(l1)[i] = b;
    }
    // the rows % 4 tail
    const int64_t r = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) {
        const int u = slot[r] - 1;
        if (!(untouched_only && u >= 0)) {
            float th = lin[r], a = l0[r], b = (KIND == DCTR_OPT_ADAM || KIND == DCTR_OPT_FTRL) ? l1[r] : 0.f;
            sq += th * th;
            float g = l2 * th;
            if (u >= 0) g += glin[u];
            opt_update(KIND, h, th, a, b, g);
            lin[r] = th; l0[r] = a;
            if (KIND == DCTR_OPT_ADAM || KIND == DCTR_OPT_FTRL) l1[r] = b;
        }
    }
    if (sumsq != nullptr) {
        __shared__ float red[4];
        sq = wave_sum(sq);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(sumsq + (blockIdx.x & (SUMSQ_SHARDS - 1)), red[0] + red[1] + red[2] + red[3]);
    }
}

template <int KIND, bool DENSE>
static int launch_table(const Hyper* hdev, const Hyper& hval, int64_t rows, int K, float* emb, float* e0, float* e1,
                        float* lin, float* l0, float* l1, const int32_t* slot, const int32_t* uniq,
                        const int32_t* counters, int64_t max_entries, const float* gemb, const float* glin, float l2,
                        float* sumsq_emb, float* sumsq_lin, hipStream_t st, hipStream_t st_lin, int untouched_only, int tab_ld, int tab_lin_ld) {
    const int KQ = K / 4;
    // row records (engine.h): the table and its slots interleaved -- the generic kernel with row strides, linear weights in the same launch
    const bool strided = (tab_ld != 0 && tab_ld != K) || tab_lin_ld != 1;
    const int ld4 = tab_ld != 0 ? tab_ld / 4 : KQ;
    // dense mode: the linear table gets its own vectorised kernel (on st_lin, beside the embedding pass); touched-rows
    // mode keeps it fused (lane kq == 0 of each visited row)
    float* lin_f = lin; float* l0_f = l0; float* l1_f = l1;
    const bool split_lin = !strided && DENSE && lin != nullptr && ((reinterpret_cast<uintptr_t>(lin) | reinterpret_cast<uintptr_t>(l0) |
                                                       reinterpret_cast<uintptr_t>(l1) | reinterpret_cast<uintptr_t>(slot)) & 15) == 0;
    if (split_lin) { lin = nullptr; l0 = nullptr; l1 = nullptr; }
    const int64_t items = DENSE ? rows : max_entries;
    static const int gmul = getenv("DCTR_OPT_GRID") ? atoi(getenv("DCTR_OPT_GRID")) : 8;
    static const int nt = getenv("DCTR_OPT_NT") ? atoi(getenv("DCTR_OPT_NT")) : 0;
    const int grid = (int)std::min<int64_t>(ceil_div(items * KQ, 256), 256 * gmul);
    float4* e4 = reinterpret_cast<float4*>(emb);
    float4* a4 = reinterpret_cast<float4*>(e0);
    float4* b4 = reinterpret_cast<float4*>(e1);
    const float4* g4 = reinterpret_cast<const float4*>(gemb);
    if (DENSE && untouched_only && !strided) {
        static const int bg = getenv("DCTR_OPT_BG_GRID") ? atoi(getenv("DCTR_OPT_BG_GRID")) : 2;     // blocks per CU of the background pass
        const int g2 = (int)std::min<int64_t>(ceil_div(items * KQ, 256 * 4), 256 * bg);
        switch (KQ) {
#define DCTR_U(Q) case Q: opt_table_untouched_kernel<KIND, Q, 4><<<g2, 256, 0, st>>>(hdev, hval, rows, e4, a4, b4, slot, l2, sumsq_emb); break
            DCTR_U(1); DCTR_U(2); DCTR_U(4); DCTR_U(8); DCTR_U(16); DCTR_U(32); DCTR_U(64);
#undef DCTR_U
            default: set_error("opt_table: K=%d unsupported", K); return DCTR_ERR_UNSUPPORTED;
        }
        if (lin_f != nullptr) {
            if (split_lin) {
                const int gl = (int)std::min<int64_t>(ceil_div(rows / 4 + 1, 256), 256 * bg);
                opt_lin_dense_kernel<KIND><<<gl, 256, 0, st>>>(hdev, hval, rows, lin_f, l0_f, l1_f, slot, glin, l2, sumsq_lin, 1);
            } else {
                // (unaligned linear table: the generic kernel, embedding side disabled, is not available -- fall through)
                set_error("opt_table: unaligned linear table in the split dense pass");
                return DCTR_ERR_UNSUPPORTED;
            }
        }
        DCTR_LAUNCH_CHECK();
        return DCTR_OK;
    }
    switch (KQ) {
#define DCTR_T(Q) case Q: opt_table_kernel<KIND, Q, DENSE><<<grid, 256, 0, st>>>(hdev, hval, rows, e4, a4, b4, lin, l0, l1, slot, uniq, counters, g4, glin, l2, sumsq_emb, sumsq_lin, nt, untouched_only, ld4, tab_lin_ld); break
        DCTR_T(1); DCTR_T(2); DCTR_T(4); DCTR_T(8); DCTR_T(16); DCTR_T(32); DCTR_T(64);
#undef DCTR_T
        default: set_error("opt_table: K=%d unsupported", K); return DCTR_ERR_UNSUPPORTED;
    }
    if (split_lin) {
        const int gl = (int)std::min<int64_t>(ceil_div(rows / 4 + 1, 256), 2048);
        opt_lin_dense_kernel<KIND><<<gl, 256, 0, st_lin>>>(hdev, hval, rows, lin_f, l0_f, l1_f, slot, glin, l2, sumsq_lin, untouched_only);
    }
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

int opt_table(int kind, const Hyper* hdev, const Hyper& hval, int table_mode, int64_t rows, int K, float* emb, float* e0,
              float* e1, float* lin, float* l0, float* l1, const int32_t* slot, const int32_t* uniq,
              const int32_t* counters, int64_t max_entries, const float* gemb, const float* glin, float l2,
              float* sumsq_emb, float* sumsq_lin, hipStream_t st, hipStream_t st_lin, int pass, int tab_ld, int tab_lin_ld) {
    DCTR_REQUIRE(tab_ld % 4 == 0 && (tab_ld == 0 || tab_ld >= K) && tab_lin_ld >= 1, "opt_table: row strides %d / %d", tab_ld, tab_lin_ld);
    // pass (dense-exact only): OPT_PASS_ALL = every row in one sweep; OPT_PASS_UNTOUCHED = rows the batch does not touch
    // (gradient = l2*theta: needs the grouping, not the backward pass); OPT_PASS_TOUCHED = the batch's distinct rows (the
    // touched-rows kernel computes the same l2*theta + segment sum).  UNTOUCHED then TOUCHED == ALL, row for row.
    const bool dense = table_mode == DCTR_TABLE_DENSE_EXACT && pass != OPT_PASS_TOUCHED;
    const int untouched_only = (dense && pass == OPT_PASS_UNTOUCHED) ? 1 : 0;
    if (table_mode != DCTR_TABLE_DENSE_EXACT && pass == OPT_PASS_UNTOUCHED) return DCTR_OK;
    if (st_lin == nullptr) st_lin = st;
#define DCTR_K(KD)                                                                                                      \
    case KD:                                                                                                            \
        return dense ? launch_table<KD, true>(hdev, hval, rows, K, emb, e0, e1, lin, l0, l1, slot, uniq, counters,      \
                                              max_entries, gemb, glin, l2, sumsq_emb, sumsq_lin, st, st_lin, untouched_only, tab_ld, tab_lin_ld) \
                     : launch_table<KD, false>(hdev, hval, rows, K, emb, e0, e1, lin, l0, l1, slot, uniq, counters,     \
                                               max_entries, gemb, glin, l2, sumsq_emb, sumsq_lin, st, st_lin, 0, tab_ld, tab_lin_ld)
    switch (kind) {
        DCTR_K(DCTR_OPT_ADAM); DCTR_K(DCTR_OPT_ADAGRAD); DCTR_K(DCTR_OPT_MOMENTUM); DCTR_K(DCTR_OPT_FTRL);
        default: set_error("unknown optimizer kind %d", kind); return DCTR_ERR_INVALID_ARG;
    }
#undef DCTR_K
}

// ---- per-step device state: global_step, Adam's lr_t, the dropout seed of this step.  Lives in device memory
// so that a captured hipGraph sees fresh values on every replay.
__global__ void step_state_kernel(StepState* s, float* zero, int n_zero, uint64_t row0) {
    for (int i = threadIdx.x; i < n_zero; i += blockDim.x) zero[i] = 0.f;     // the step's loss scalars
    if (threadIdx.x != 0) return;
    s->t += 1;
    s->row0 = row0;
    const double t = (double)s->t;
    for (Hyper* hp : {&s->hyper, &s->hyper_lin}) {
        Hyper& h = *hp;
        h.lr_t = (float)((double)h.lr * sqrt(1.0 - pow((double)h.beta2, t)) / (1.0 - pow((double)h.beta1, t)));
    }
    s->seed_t = s->seed ^ ((uint64_t)s->t * STEP_SEED_MULT);
    s->lr_hist[s->t & (LR_HIST - 1)] = s->hyper.lr_t;
}

// the NEXT step's state from the current one, into a second StepState (and a second set of loss scalars, zeroed): launched under
// the tail of the step in flight, so that the next step starts with its state ready instead of with this 5 us kernel
__global__ void step_state_next_kernel(const StepState* __restrict__ cur, StepState* __restrict__ nxt, float* __restrict__ zero, int n_zero) {
    for (int i = threadIdx.x; i < n_zero; i += blockDim.x) zero[i] = 0.f;
    if (threadIdx.x != 0) return;
    StepState s = *cur;
    // (prepared EARLY in the step: a replay later in that step sets lag_overflow in `cur` after this copy was made -- it then sits in the
    //  state this kernel overwrites one step later: keep what is there)
    s.lag_overflow |= nxt->lag_overflow;
    s.t += 1;
    const double t = (double)s.t;
    for (Hyper* hp : {&s.hyper, &s.hyper_lin}) {
        Hyper& h = *hp;
        h.lr_t = (float)((double)h.lr * sqrt(1.0 - pow((double)h.beta2, t)) / (1.0 - pow((double)h.beta1, t)));
    }
    s.seed_t = s.seed ^ ((uint64_t)s.t * STEP_SEED_MULT);
    s.lr_hist[s.t & (LR_HIST - 1)] = s.hyper.lr_t;
    *nxt = s;
}

int step_state_next(const StepState* cur, StepState* nxt, float* zero, int n_zero, hipStream_t st) {
    step_state_next_kernel<<<1, 256, 0, st>>>(cur, nxt, zero, n_zero);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

int step_state_advance(StepState* s, float* zero, int n_zero, hipStream_t st, uint64_t row0) {
    step_state_kernel<<<1, 256, 0, st>>>(s, zero, n_zero, row0);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

// ---- K10 tf.metrics.auc: 200 thresholds, counts tp/fn/tn/fp with pred > thr [TF-1.4] ---------
__global__ __launch_bounds__(256) void auc_update_kernel(const float* __restrict__ labels, const float* __restrict__ prob,
                                                        int B, unsigned long long* __restrict__ counts) {
    // one thread per threshold (200 of the 256 lanes), each walks the batch: B*200 compares, trivial
    const int k = threadIdx.x;
    if (k >= 200) return;
    const float eps = 1e-7f;
    float thr;
    if (k == 0) thr = 0.0f - eps;
    else if (k == 199) thr = 1.0f + eps;
    else thr = (float)k * 1.0f / 199.0f;
    const int chunk = (B + gridDim.x - 1) / gridDim.x;
    const int b0 = blockIdx.x * chunk, b1 = min(B, b0 + chunk);
    unsigned long long tp = 0, fn = 0, tn = 0, fp = 0;
    for (int b = b0; b < b1; ++b) {
        const bool pos = labels[b] != 0.f;
        const bool gt = prob[b] > thr;
        tp += (pos && gt); fn += (pos && !gt); fp += (!pos && gt); tn += (!pos && !gt);
    }
    if (b1 > b0) {
        atomicAdd(&counts[0 * 200 + k], tp);
        atomicAdd(&counts[1 * 200 + k], fn);
        atomicAdd(&counts[2 * 200 + k], tn);
        atomicAdd(&counts[3 * 200 + k], fp);
    }
}

}  // namespace dctr

using namespace dctr;

static Hyper hyper_from_host(int kind, const float* hyper) {
    Hyper h{};
    h.lr = hyper[0];
    h.beta1 = 0.9f; h.beta2 = 0.999f; h.eps = 1e-8f; h.momentum = 0.95f; h.lr_t = h.lr;
    if (kind == DCTR_OPT_ADAM) {
        h.beta1 = hyper[1]; h.beta2 = hyper[2]; h.eps = hyper[3];
        const double t = (double)hyper[4];
        h.lr_t = (float)((double)h.lr * sqrt(1.0 - pow((double)h.beta2, t)) / (1.0 - pow((double)h.beta1, t)));
    } else if (kind == DCTR_OPT_MOMENTUM) {
        h.momentum = hyper[1];
    }
    return h;
}

extern "C" {

int dctr_loss_head(const float* d_bias, const float* d_yw, const float* d_yv, const float* d_yd, const float* d_labels,
                   int B, float inv_batch, float* d_y, float* d_prob, float* d_dy, float* d_loss_sum, void* stream) {
    return loss_head(d_bias, d_yw, d_yv, d_yd, d_labels, B, inv_batch, d_y, d_prob, d_dy, d_loss_sum, as_stream(stream));
}

int dctr_opt_dense(int kind, const float* hyper, float* d_theta, float* d_slot0, float* d_slot1, const float* d_grad,
                   int n_partials, int64_t partial_stride, int64_t n, float l2, void* stream) {
    DCTR_REQUIRE(hyper != nullptr, "hyper required");
    const Hyper h = hyper_from_host(kind, hyper);
    return opt_dense_flat(kind, nullptr, h, d_theta, d_slot0, d_slot1, d_grad, n_partials, partial_stride, n, l2, as_stream(stream));
}

int dctr_opt_table(int kind, const float* hyper, int table_mode, int64_t rows, int K, float* d_emb, float* d_emb_s0,
                   float* d_emb_s1, float* d_lin, float* d_lin_s0, float* d_lin_s1, dctr_group_t g, float l2,
                   float* d_sumsq, void* stream) {
    DCTR_REQUIRE(hyper != nullptr && g != nullptr, "hyper and group required");
    const Hyper h = hyper_from_host(kind, hyper);
    const int32_t *uniq, *slot, *counters;
    float *gemb, *glin;
    DCTR_TRY(dctr_group_buffers(g, &uniq, nullptr, nullptr, nullptr, &slot, &counters, &gemb, &glin));
    const int64_t max_entries = group_capacity(reinterpret_cast<Group*>(g));
    return opt_table(kind, nullptr, h, table_mode, rows, K, d_emb, d_emb_s0, d_emb_s1, d_lin, d_lin_s0, d_lin_s1, slot,
                     uniq, counters, max_entries, gemb, glin, l2, d_sumsq, d_sumsq ? d_sumsq + SUMSQ_SHARDS : nullptr, as_stream(stream), nullptr);
}

int dctr_auc_update(const float* d_labels, const float* d_prob, int B, int64_t* d_counts, void* stream) {
    if (B <= 0) return DCTR_OK;
    const int grid = std::max(1, std::min(256, B / 64));
    auc_update_kernel<<<grid, 256, 0, as_stream(stream)>>>(d_labels, d_prob, B, reinterpret_cast<unsigned long long*>(d_counts));
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

int dctr_auc_result(const int64_t* d_counts, float* h_auc, void* stream) {
    DCTR_REQUIRE(h_auc != nullptr, "null out pointer");
    int64_t c[800];
    DCTR_HIP_CHECK(hipMemcpyAsync(c, d_counts, sizeof(c), hipMemcpyDeviceToHost, as_stream(stream)));
    DCTR_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
    // trapezoid over the ROC points, float32 arithmetic as tf.metrics.auc does [TF-1.4]
    const float eps = 1e-6f;
    float tpr[200], fpr[200];
    for (int k = 0; k < 200; ++k) {
        const float tp = (float)c[k], fn = (float)c[200 + k], tn = (float)c[400 + k], fp = (float)c[600 + k];
        tpr[k] = (tp + eps) / (tp + fn + eps);
        fpr[k] = fp / (fp + tn + eps);
    }
    float auc = 0.f;
    for (int k = 0; k < 199; ++k) auc += (fpr[k] - fpr[k + 1]) * (tpr[k] + tpr[k + 1]) / 2.0f;
    *h_auc = auc;
    return DCTR_OK;
}

}  // extern "C"
