// HBM copy-bandwidth probe: a few float4 copy kernel shapes + hipMemcpyAsync, 1 GiB buffers.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_one(const float4* __restrict__ s, float4* __restrict__ d, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = s[i];
}
template <int U>
__global__ __launch_bounds__(256) void k_chunk(const float4* __restrict__ s, float4* __restrict__ d, size_t n) {
    // each block owns a contiguous chunk of U*256 float4
    const size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) v[u] = s[base + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) d[base + u * 256] = v[u];
}
template <int U>
__global__ __launch_bounds__(256) void k_stride(const float4* __restrict__ s, float4* __restrict__ d, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = s[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) d[i + u * stride] = v[u];
    }
    for (; i < n; i += stride) d[i] = s[i];
}
template <int U>
__global__ __launch_bounds__(256) void k_nt(const float4* __restrict__ s, float4* __restrict__ d, size_t n) {
    const size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) { const float* p = (const float*)(s + base + u * 256);
        v[u].x = __builtin_nontemporal_load(p); v[u].y = __builtin_nontemporal_load(p + 1); v[u].z = __builtin_nontemporal_load(p + 2); v[u].w = __builtin_nontemporal_load(p + 3); }
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) { float* p = (float*)(d + base + u * 256);
        __builtin_nontemporal_store(v[u].x, p); __builtin_nontemporal_store(v[u].y, p + 1); __builtin_nontemporal_store(v[u].z, p + 2); __builtin_nontemporal_store(v[u].w, p + 3); }
}

int main() {
    const size_t bytes = 1ull << 30, n = bytes / 16;
    float4 *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char* name, auto launch) {
        launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %8.1f GB/s (read+write)\n", name, 2.0 * bytes * 20 / (ms * 1e6));
    };
    time("one float4/thread", [&] { k_one<<<(unsigned)((n + 255) / 256), 256>>>(a, b, n); });
    time("chunk U=4", [&] { k_chunk<4><<<(unsigned)((n + 1023) / 1024), 256>>>(a, b, n); });
    time("chunk U=8", [&] { k_chunk<8><<<(unsigned)((n + 2047) / 2048), 256>>>(a, b, n); });
    time("grid-stride 2048 blk U=4", [&] { k_stride<4><<<2048, 256>>>(a, b, n); });
    time("grid-stride 4096 blk U=8", [&] { k_stride<8><<<4096, 256>>>(a, b, n); });
    time("grid-stride 1024 blk U=8", [&] { k_stride<8><<<1024, 256>>>(a, b, n); });
    time("nontemporal chunk U=4", [&] { k_nt<4><<<(unsigned)((n + 1023) / 1024), 256>>>(a, b, n); });
    time("hipMemcpyAsync D2D", [&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
    return 0;
}
