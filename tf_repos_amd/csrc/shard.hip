// Row-sharded tables across the GPUs of one node: owner(id) = id % world, local row = id / world.
// These kernels only bucketise / permute indices and rows; the exchange itself is an RCCL all-to-all issued by the
// host (tf_repos_amd/distributed.py).  The batch's ids are de-duplicated first (group.hip), so each distinct id
// crosses xGMI once per direction regardless of how often the batch repeats it (Criteo's 13 numeric ids are hit by
// every example).
#include "common.h"
#include "ops.h"

namespace dctr {

// counts[d] = number of distinct ids owned by rank d; one atomic per (wave, destination)
__global__ __launch_bounds__(256) void route_count_kernel(const int32_t* __restrict__ uniq, const int32_t* __restrict__ counters,
                                                         int world, int32_t* __restrict__ counts) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    const int U = counters[0];
    const int lane = threadIdx.x & 63;
    const int dest = (u < U) ? uniq[u] % world : -1;
    for (int d = 0; d < world; ++d) {
        const unsigned long long m = __ballot(dest == d);
        if (m != 0ull && lane == __ffsll((long long)m) - 1) atomicAdd(&counts[d], __popcll(m));
    }
}

// send_rows grouped by destination; upos[u] = position of distinct id u in the send buffer
__global__ __launch_bounds__(256) void route_fill_kernel(const int32_t* __restrict__ uniq, const int32_t* __restrict__ counters,
                                                        int world, const int32_t* __restrict__ counts, int32_t* __restrict__ cursor,
                                                        int32_t* __restrict__ send_rows, int32_t* __restrict__ upos) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    const int U = counters[0];
    const int lane = threadIdx.x & 63;
    const int id = (u < U) ? uniq[u] : -1;
    const int dest = (u < U) ? id % world : -1;
    for (int d = 0; d < world; ++d) {
        const unsigned long long m = __ballot(dest == d);
        if (m == 0ull) continue;
        const int head = __ffsll((long long)m) - 1;
        int base = 0;
        if (lane == head) base = atomicAdd(&cursor[d], __popcll(m));
        base = __shfl(base, head);
        if (dest == d) {
            int off = 0;
            for (int j = 0; j < d; ++j) off += counts[j];
            const int pos = off + base + __popcll(m & ((1ull << lane) - 1ull));
            send_rows[pos] = id / world;
            upos[u] = pos;
        }
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

// dst[pos[u], :] = src[u, :] for u < U (U read from device memory)
__global__ __launch_bounds__(256) void permute_rows_kernel(const float4* __restrict__ src, const int32_t* __restrict__ pos,
                                                          const int32_t* __restrict__ counters, int kq_per_row, float4* __restrict__ dst) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int u = (int)(t / kq_per_row), kq = (int)(t % kq_per_row);
    if (u >= counters[0]) return;
    dst[(size_t)pos[u] * kq_per_row + kq] = src[(size_t)u * kq_per_row + kq];
}
__global__ __launch_bounds__(256) void permute_scalars_kernel(const float* __restrict__ src, const int32_t* __restrict__ pos,
                                                             const int32_t* __restrict__ counters, float* __restrict__ dst) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= counters[0]) return;
    dst[pos[u]] = src[u];
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

int dctr_permute_unique_rows(dctr_group_t g, const float* d_src, const int32_t* d_pos, int K, float* d_dst, void* stream) {
    DCTR_REQUIRE(g != nullptr && d_src && d_pos && d_dst && (K == 1 || K % 4 == 0), "bad argument");
    Group* p = reinterpret_cast<Group*>(g);
    hipStream_t st = as_stream(stream);
    if (K == 1) {
        permute_scalars_kernel<<<ceil_div(p->max_entries, 256), 256, 0, st>>>(d_src, d_pos, p->counters, d_dst);
    } else {
        const int kq = K / 4;
        permute_rows_kernel<<<ceil_div(p->max_entries * kq, 256), 256, 0, st>>>(reinterpret_cast<const float4*>(d_src), d_pos, p->counters,
                                                                                kq, reinterpret_cast<float4*>(d_dst));
    }
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

}  // extern "C"
