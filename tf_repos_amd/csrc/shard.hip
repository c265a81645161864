// Row-sharded tables across the GPUs of one node: owner(id) = id % world, local row = id / world.
// These kernels only bucketise / permute indices and rows; the exchange itself is an RCCL all-to-all issued by the
// host (tf_repos_amd/distributed.py).  The batch's ids are de-duplicated first (group.hip), so each distinct id
// crosses xGMI once per direction regardless of how often the batch repeats it (Criteo's 13 numeric ids are hit by
// every example).
#include "common.h"
#include "ops.h"
#include "lag.h"

namespace dctr {

// counts[d] = number of distinct ids owned by rank d.  Per-wave ballots, summed per block in LDS: one atomic per (block,
// destination) -- thousands of waves adding to `world` words would serialise at ~90 atomics/us per word
constexpr int ROUTE_WAVES = 4;      // 256-thread blocks
constexpr int ROUTE_MAX_WORLD = 64;

__global__ __launch_bounds__(256) void route_count_kernel(const int32_t* __restrict__ uniq, const int32_t* __restrict__ counters,
                                                         int world, int32_t* __restrict__ counts) {
    __shared__ int32_t wc[ROUTE_WAVES][ROUTE_MAX_WORLD];
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    const int U = counters[0];
    if ((int)(blockIdx.x * blockDim.x) >= U) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int dest = (u < U) ? uniq[u] % world : -1;
    for (int d = 0; d < world; ++d) {
        const unsigned long long m = __ballot(dest == d);
        if (lane == 0) wc[wave][d] = __popcll(m);
    }
    __syncthreads();
    if ((int)threadIdx.x < world) {
        int tot = 0;
#pragma unroll
        for (int w = 0; w < ROUTE_WAVES; ++w) tot += wc[w][threadIdx.x];
        if (tot) atomicAdd(&counts[threadIdx.x], tot);
    }
}

// send_rows grouped by destination; upos[u] = position of distinct id u in the send buffer
__global__ __launch_bounds__(256) void route_fill_kernel(const int32_t* __restrict__ uniq, const int32_t* __restrict__ counters,
                                                        int world, const int32_t* __restrict__ counts, int32_t* __restrict__ cursor,
                                                        int32_t* __restrict__ send_rows, int32_t* __restrict__ upos) {
    __shared__ int32_t wc[ROUTE_WAVES][ROUTE_MAX_WORLD];    // per-wave counts, then per-wave bases in the send buffer
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    const int U = counters[0];
    if ((int)(blockIdx.x * blockDim.x) >= U) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int id = (u < U) ? uniq[u] : -1;
    const int dest = (u < U) ? id % world : -1;
    unsigned long long mine = 0ull;
    for (int d = 0; d < world; ++d) {
        const unsigned long long m = __ballot(dest == d);
        if (dest == d) mine = m;
        if (lane == 0) wc[wave][d] = __popcll(m);
    }
    __syncthreads();
    if ((int)threadIdx.x < world) {
        const int d = threadIdx.x;
        int tot = 0;
#pragma unroll
        for (int w = 0; w < ROUTE_WAVES; ++w) tot += wc[w][d];
        int base = tot ? atomicAdd(&cursor[d], tot) : 0;
        for (int j = 0; j < d; ++j) base += counts[j];      // start of destination d's range
#pragma unroll
        for (int w = 0; w < ROUTE_WAVES; ++w) { const int c = wc[w][d]; wc[w][d] = base; base += c; }
    }
    __syncthreads();
    if (dest >= 0) {
        const int pos = wc[wave][dest] + __popcll(mine & ((1ull << lane) - 1ull));
        send_rows[pos] = id / world;
        upos[u] = pos;
    }
}

__global__ __launch_bounds__(256) void entry_index_kernel(const int32_t* __restrict__ ids, int n, int64_t rows,
                                                         const int32_t* __restrict__ slot, const int32_t* __restrict__ upos,
                                                         int32_t* __restrict__ idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int id = ids[i];
    idx[i] = (id >= 0 && (int64_t)id < rows) ? upos[slot[id] - 1] : -1;     // -1 trips the gather's range check
}

// Rows cross the fabric PACKED: one [K+4]-float record per row = K embedding floats | linear weight | 3 pad floats, so each
// direction of the exchange is ONE all-to-all of 16-byte-aligned records (K=16: 80 B) instead of one per table.

// owner side: out[i] = { emb[rows_idx[i], :], lin[rows_idx[i]], 0, 0, 0 }
__global__ __launch_bounds__(256) void pack_table_rows_kernel(const float4* __restrict__ emb, const float* __restrict__ lin,
                                                             int64_t rows, const int32_t* __restrict__ rows_idx, int n, int KQ,
                                                             float4* __restrict__ out, int32_t* __restrict__ status, int ld4, int lin_ld) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int Q = KQ + 1;
    const int i = (int)(t / Q), q = (int)(t % Q);
    if (i >= n) return;
    const int r = rows_idx[i];
    const bool ok = r >= 0 && (int64_t)r < rows;
    if (!ok && q == 0) { atomicExch(&status[1], r); atomicExch(&status[0], 1); }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) {
        if (q < KQ) v = emb[(size_t)r * ld4 + q];
        else if (lin != nullptr) v.x = lin[(size_t)r * lin_ld];
    }
    out[(size_t)i * Q + q] = v;
}

// the same from a table whose rows may lag (lag.h): the row is advanced to the present (state->t) in registers before it is
// shipped; nothing is written back (the touched-rows step of the batch redoes it and stores)
__global__ __launch_bounds__(256) void pack_table_rows_lag_kernel(const float4* __restrict__ emb, const float* __restrict__ lin,
                                                                 int64_t rows, const int32_t* __restrict__ rows_idx, int n, int KQ,
                                                                 float4* __restrict__ out, int32_t* __restrict__ status, LagView L) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int Q = KQ + 1;
    const int i = (int)(t / Q), q = (int)(t % Q);
    if (i >= n) return;
    const int r = rows_idx[i];
    const bool ok = r >= 0 && (int64_t)r < rows;
    if (!ok && q == 0) { atomicExch(&status[1], r); atomicExch(&status[0], 1); }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) {
        const int64_t T = L.state->t;
        const int nl = lag_behind(T, L.ts[r]);
        const Hyper h = L.state->hyper;
        if (q < KQ) {
            v = emb[(size_t)r * L.ld4 + q];
            if (nl > 0) {
                float4 m = L.s0[(size_t)r * L.ld4 + q], vv = L.s1[(size_t)r * L.ld4 + q];
                lag_catch_up4(L.state, h, L.l2, T - nl + 1, nl, v, m, vv);
            }
        } else if (lin != nullptr) {
            v.x = lin[(size_t)r * L.lin_ld];
            if (nl > 0) {
                float m = L.l0[(size_t)r * L.lin_ld], vv = L.l1[(size_t)r * L.lin_ld];
                lag_catch_up1(L.state, h, L.l2, T - nl + 1, nl, v.x, m, vv);
            }
        }
    }
    out[(size_t)i * Q + q] = v;
}

// requester side: out[upos[u]] = { gemb[u, :], glin[u], 0, 0, 0 } for u < U (U read from device memory)
__global__ __launch_bounds__(256) void pack_unique_grads_kernel(const float4* __restrict__ gemb, const float* __restrict__ glin,
                                                               const int32_t* __restrict__ upos, const int32_t* __restrict__ counters,
                                                               int KQ, float4* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int Q = KQ + 1;
    const int u = (int)(t / Q), q = (int)(t % Q);
    if (u >= counters[0]) return;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < KQ) v = gemb[(size_t)u * KQ + q];
    else if (glin != nullptr) v.x = glin[u];
    out[(size_t)upos[u] * Q + q] = v;
}

int pack_table_rows(const float* emb, const float* lin, int64_t rows, int K, const int32_t* rows_idx, int n, float* out,
                    int32_t* status, hipStream_t st, const LagView* lag, int tab_ld, int tab_lin_ld) {
    if (n <= 0) return DCTR_OK;
    const int KQ = K / 4;
    const int ld4 = tab_ld > 0 ? tab_ld / 4 : KQ;
    if (lag != nullptr) {
        LagView L = *lag;       // (the slots share the table's row strides)
        L.ld4 = ld4; L.lin_ld = tab_lin_ld;
        pack_table_rows_lag_kernel<<<ceil_div((int64_t)n * (KQ + 1), 256), 256, 0, st>>>(reinterpret_cast<const float4*>(emb), lin, rows, rows_idx,
                                                                                       n, KQ, reinterpret_cast<float4*>(out), status, L);
        DCTR_LAUNCH_CHECK();
        return DCTR_OK;
    }
    pack_table_rows_kernel<<<ceil_div((int64_t)n * (KQ + 1), 256), 256, 0, st>>>(reinterpret_cast<const float4*>(emb), lin, rows, rows_idx,
                                                                               n, KQ, reinterpret_cast<float4*>(out), status, ld4, tab_lin_ld);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

int pack_unique_grads(const Group* g, const float* glin, const int32_t* upos, float* out, hipStream_t st) {
    const int KQ = g->K / 4;
    pack_unique_grads_kernel<<<ceil_div(g->max_entries * (KQ + 1), 256), 256, 0, st>>>(reinterpret_cast<const float4*>(g->gemb), glin, upos,
                                                                                      g->counters, KQ, reinterpret_cast<float4*>(out));
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

}  // namespace dctr

using namespace dctr;

extern "C" {

int dctr_route_unique(dctr_group_t g, int world, int32_t* d_send_rows, int32_t* d_upos, int32_t* d_counts, void* stream) {
    DCTR_REQUIRE(g != nullptr && world >= 1 && world <= 64 && d_send_rows && d_upos && d_counts, "bad argument");
    Group* p = reinterpret_cast<Group*>(g);
    hipStream_t st = as_stream(stream);
    // d_counts: int32[2*world] = counts then cursors
    DCTR_HIP_CHECK(hipMemsetAsync(d_counts, 0, sizeof(int32_t) * 2 * world, st));
    const int nb = ceil_div(p->max_entries, 256);
    route_count_kernel<<<nb, 256, 0, st>>>(p->uniq, p->counters, world, d_counts);
    route_fill_kernel<<<nb, 256, 0, st>>>(p->uniq, p->counters, world, d_counts, d_counts + world, d_send_rows, d_upos);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

int dctr_entry_index(dctr_group_t g, const int32_t* d_ids, int n, const int32_t* d_upos, int32_t* d_idx, void* stream) {
    DCTR_REQUIRE(g != nullptr && d_ids && d_upos && d_idx, "bad argument");
    Group* p = reinterpret_cast<Group*>(g);
    if (n <= 0) return DCTR_OK;
    entry_index_kernel<<<ceil_div(n, 256), 256, 0, as_stream(stream)>>>(d_ids, n, p->rows, p->slot, d_upos, d_idx);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

}  // extern "C"
