// contrib.layers.batch_norm(decay, center=True, scale=True, epsilon=1e-3, updates_collections=None) as the reference's
// batch_norm_layer applies it AFTER the ReLU of every hidden layer (DeepFM.py:159-160, 231-235), followed by dropout
// (DeepFM.py:161-162).  TRAIN: batch statistics (biased variance) + in-place moving-average update
// moving <- decay*moving + (1-decay)*batch [TF-1.4]; otherwise the moving statistics.  The moving VARIANCE is fed the
// Bessel-corrected batch variance var * B/(B-1), as TF-1.4's fused batch-norm kernel does (the path contrib.layers.batch_norm
// takes for rank-2 inputs); dctr_config.batch_norm_biased_moving_variance = 1 feeds it the biased one.
// HBM-bound column reductions over [B, H]: S row-splits of partial sums per 64-column group, then a tiny finalize.
#include "ops.h"

namespace dctr {

constexpr int BN_SPLITS = 64;

// part[s][0][c] = sum_{r in split s} y[r,c],  part[s][1][c] = sum y^2
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ y, int ldy, int B, int H, int rows_per_split,
                                                              float* __restrict__ part) {
    __shared__ float red[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int rbeg = blockIdx.y * rows_per_split, rend = min(B, rbeg + rows_per_split);
    float s = 0.f, q = 0.f;
    if (c < H)
        for (int r = rbeg + rl; r < rend; r += 4) { const float v = y[(size_t)r * ldy + c]; s += v; q += v * v; }
    red[0][rl][threadIdx.x & 63] = s; red[1][rl][threadIdx.x & 63] = q;
    __syncthreads();
    if (rl == 0 && c < H) {
        const int x = threadIdx.x;
        part[((size_t)blockIdx.y * 2 + 0) * H + c] = red[0][0][x] + red[0][1][x] + red[0][2][x] + red[0][3][x];
        part[((size_t)blockIdx.y * 2 + 1) * H + c] = red[1][0][x] + red[1][1][x] + red[1][2][x] + red[1][3][x];
    }
}

// mean / invstd of the batch (stats[0][c], stats[1][c]) and the moving-average update
__global__ void bn_stats_finalize_kernel(const float* __restrict__ part, int S, int64_t B, int H, float eps, float decay,
                                         float* __restrict__ stats, float* __restrict__ mm, float* __restrict__ mv, int bessel) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= H) return;
    double s = 0.0, q = 0.0;
    for (int k = 0; k < S; ++k) { s += part[((size_t)k * 2 + 0) * H + c]; q += part[((size_t)k * 2 + 1) * H + c]; }
    const double mean = s / B;
    double var = q / B - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[c] = (float)mean;
    stats[H + c] = (float)(1.0 / sqrt(var + (double)eps));
    mm[c] = decay * mm[c] + (1.0f - decay) * (float)mean;
    // fused_batch_norm_op.cc [TF-1.4]: rest_size_adjust = rest_size / max(rest_size - 1, 1) on the variance handed to the moving average
    const double adj = bessel ? (double)B / (double)(B > 1 ? B - 1 : 1) : 1.0;
    mv[c] = decay * mv[c] + (1.0f - decay) * (float)(var * adj);
}

// inference statistics: mean = moving_mean, invstd = 1/sqrt(moving_variance + eps)
__global__ void bn_stats_moving_kernel(const float* __restrict__ mm, const float* __restrict__ mv, int H, float eps,
                                       float* __restrict__ stats) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= H) return;
    stats[c] = mm[c];
    stats[H + c] = 1.0f / sqrtf(mv[c] + eps);
}

// out = dropout((y - mean) * invstd * gamma + beta)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ y, int ldy, const float* __restrict__ stats,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, int B, int H,
                                                      float keep, const uint64_t* __restrict__ seed_ptr, uint64_t salt,
                                                      float* __restrict__ out, int ldo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * H) return;
    const int r = (int)(i / H), c = (int)(i % H);
    float z = (y[(size_t)r * ldy + c] - stats[c]) * stats[H + c] * gamma[c] + beta[c];
    if (keep < 1.0f) z *= dropout_scale(*seed_ptr ^ salt, dropout_row0(seed_ptr) * (uint64_t)H + (uint64_t)i, keep);
    out[(size_t)r * ldo + c] = z;
}

// part[s][0][c] = sum dz, part[s][1][c] = sum dz*xhat with dz = dout * dropout mask, xhat = (y-mean)*invstd
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ dout, int ldd, const float* __restrict__ y,
                                                            int ldy, const float* __restrict__ stats, int B, int H, int rows_per_split,
                                                            float keep, const uint64_t* __restrict__ seed_ptr, uint64_t salt,
                                                            float* __restrict__ part) {
    __shared__ float red[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int rbeg = blockIdx.y * rows_per_split, rend = min(B, rbeg + rows_per_split);
    float s = 0.f, q = 0.f;
    if (c < H) {
        const float mean = stats[c], inv = stats[H + c];
        const uint64_t seed = keep < 1.0f ? (*seed_ptr ^ salt) : 0ull;
        for (int r = rbeg + rl; r < rend; r += 4) {
            float dz = dout[(size_t)r * ldd + c];
            if (keep < 1.0f) dz *= dropout_scale(seed, (dropout_row0(seed_ptr) + (uint64_t)r) * H + c, keep);
            s += dz;
            q += dz * (y[(size_t)r * ldy + c] - mean) * inv;
        }
    }
    red[0][rl][threadIdx.x & 63] = s; red[1][rl][threadIdx.x & 63] = q;
    __syncthreads();
    if (rl == 0 && c < H) {
        const int x = threadIdx.x;
        part[((size_t)blockIdx.y * 2 + 0) * H + c] = red[0][0][x] + red[0][1][x] + red[0][2][x] + red[0][3][x];
        part[((size_t)blockIdx.y * 2 + 1) * H + c] = red[1][0][x] + red[1][1][x] + red[1][2][x] + red[1][3][x];
    }
}

// dbeta / dgamma (written to their gradient slabs and to sums[0..H), sums[H..2H) for the apply kernel).  gscale: 1, or 1 / world
// when the sums are already global (synchronised statistics): the dense all-reduce that follows sums the slabs of all ranks.
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ part, int S, int H, float gscale, float* __restrict__ sums,
                                       float* __restrict__ dbeta, float* __restrict__ dgamma) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= H) return;
    double s = 0.0, q = 0.0;
    for (int k = 0; k < S; ++k) { s += part[((size_t)k * 2 + 0) * H + c]; q += part[((size_t)k * 2 + 1) * H + c]; }
    sums[c] = (float)s; sums[H + c] = (float)q;
    dbeta[c] = (float)s * gscale; dgamma[c] = (float)q * gscale;
}

// out[0][c], out[1][c] = the S partial sums folded (what goes through the cross-rank sum)
__global__ void bn_fold_kernel(const float* __restrict__ part, int S, int H, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= H) return;
    double s = 0.0, q = 0.0;
    for (int k = 0; k < S; ++k) { s += part[((size_t)k * 2 + 0) * H + c]; q += part[((size_t)k * 2 + 1) * H + c]; }
    out[c] = (float)s; out[H + c] = (float)q;
}

// dy = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)), then the ReLU that produced y: dpre = dy * (y > 0)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dout, int ldd, const float* __restrict__ y, int ldy,
                                                          const float* __restrict__ stats, const float* __restrict__ gamma,
                                                          const float* __restrict__ sums, int B, int64_t Bstat, int H, float keep,
                                                          const uint64_t* __restrict__ seed_ptr, uint64_t salt,
                                                          float* __restrict__ dpre, int ldp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * H) return;
    const int r = (int)(i / H), c = (int)(i % H);
    const float yv = y[(size_t)r * ldy + c];
    float dz = dout[(size_t)r * ldd + c];
    if (keep < 1.0f) dz *= dropout_scale(*seed_ptr ^ salt, dropout_row0(seed_ptr) * (uint64_t)H + (uint64_t)i, keep);
    const float inv = stats[H + c];
    const float xhat = (yv - stats[c]) * inv;
    const float invB = 1.0f / (float)Bstat;
    const float dy = gamma[c] * inv * (dz - sums[c] * invB - xhat * sums[H + c] * invB);
    dpre[(size_t)r * ldp + c] = yv > 0.f ? dy : 0.f;
}

// sync (optional): data-parallel ranks, each with B rows of ONE global batch of B * world -- the column sums go through a
// cross-rank sum so that mean / variance (and, backward, the two gradient sums) are the global batch's: N ranks == 1 rank.
int bn_forward(const float* y, int ldy, int B, int H, bool train, float eps, float decay, const float* gamma, const float* beta,
               float* mm, float* mv, float keep, const uint64_t* seed_ptr, uint64_t salt, float* stats, float* scratch, float* out,
               int ldo, hipStream_t st, const BnSync* sync, bool bessel) {
    if (train) {
        const int S = BN_SPLITS;
        bn_stats_partial_kernel<<<dim3(ceil_div(H, 64), S), 256, 0, st>>>(y, ldy, B, H, ceil_div(B, S), scratch);
        if (sync != nullptr && sync->world > 1) {
            float* folded = scratch + (size_t)S * 2 * H;
            bn_fold_kernel<<<ceil_div(H, 256), 256, 0, st>>>(scratch, S, H, folded);
            DCTR_LAUNCH_CHECK();
            DCTR_TRY(sync->all_reduce(sync->ctx, 0, folded, 2 * (int64_t)H, st));
            bn_stats_finalize_kernel<<<ceil_div(H, 256), 256, 0, st>>>(folded, 1, (int64_t)B * sync->world, H, eps, decay, stats, mm, mv, bessel ? 1 : 0);
        } else {
            bn_stats_finalize_kernel<<<ceil_div(H, 256), 256, 0, st>>>(scratch, S, B, H, eps, decay, stats, mm, mv, bessel ? 1 : 0);
        }
    } else {
        bn_stats_moving_kernel<<<ceil_div(H, 256), 256, 0, st>>>(mm, mv, H, eps, stats);
    }
    bn_apply_kernel<<<ceil_div((int64_t)B * H, 256), 256, 0, st>>>(y, ldy, stats, gamma, beta, B, H, train ? keep : 1.0f, seed_ptr, salt,
                                                                   out, ldo);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

int bn_backward(const float* dout, int ldd, const float* y, int ldy, int B, int H, const float* stats, const float* gamma, float keep,
                const uint64_t* seed_ptr, uint64_t salt, float* scratch, float* dbeta, float* dgamma, float* dpre, int ldp,
                hipStream_t st, const BnSync* sync) {
    const int S = BN_SPLITS;
    float* sums = scratch + (size_t)S * 2 * H;
    bn_bwd_partial_kernel<<<dim3(ceil_div(H, 64), S), 256, 0, st>>>(dout, ldd, y, ldy, stats, B, H, ceil_div(B, S), keep, seed_ptr, salt,
                                                                    scratch);
    int64_t Bstat = B;
    if (sync != nullptr && sync->world > 1) {
        bn_fold_kernel<<<ceil_div(H, 256), 256, 0, st>>>(scratch, S, H, sums);
        DCTR_LAUNCH_CHECK();
        DCTR_TRY(sync->all_reduce(sync->ctx, 0, sums, 2 * (int64_t)H, st));
        bn_bwd_finalize_kernel<<<ceil_div(H, 256), 256, 0, st>>>(sums, 1, H, 1.0f / (float)sync->world, sums, dbeta, dgamma);
        Bstat = (int64_t)B * sync->world;
    } else {
        bn_bwd_finalize_kernel<<<ceil_div(H, 256), 256, 0, st>>>(scratch, S, H, 1.0f, sums, dbeta, dgamma);
    }
    bn_bwd_apply_kernel<<<ceil_div((int64_t)B * H, 256), 256, 0, st>>>(dout, ldd, y, ldy, stats, gamma, sums, B, Bstat, H, keep, seed_ptr, salt,
                                                                       dpre, ldp);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

int bn_scratch_floats(int H) { return (BN_SPLITS * 2 + 2) * H; }

}  // namespace dctr
