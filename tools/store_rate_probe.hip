// Round 6 probe: how fast can 3.1 GB be WRITTEN -- contiguous 16 bytes per lane (a wave instruction = 1 KB contiguous), against the tall gate
// kernel's epilogue pattern (a wave instruction = 16 rows x 64 contiguous bytes, the rest of each 1-KB row by 15 more instructions).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/_bin/store_rate_probe tools/store_rate_probe.hip && tools/_bin/store_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// pattern 0: lane-contiguous float4, grid-stride.  pattern 1: per wave 64 rows x 256 columns written as (row tile i, column tile tt): lane (c, q)
// stores 16 bytes at row 16 i + c, column 16 tt + 4 q -- the MFMA accumulator layout of gemm_ts_kernel.
template <int PATTERN>
__global__ __launch_bounds__(256) void store_kernel(float* out, long long rows, int ld) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;
    const f32x4 v = {1.f, 2.f, 3.f, (float)lane};
    if (PATTERN == 0) {
        const long long n4 = rows * ld / 4;
        for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += gridDim.x * 256ll) reinterpret_cast<f32x4*>(out)[i] = v;
    } else {
        for (long long t = blockIdx.x; t * 256 < rows; t += gridDim.x) {
            const long long m0 = t * 256 + 64 * w;
#pragma unroll
            for (int tt = 0; tt < 16; ++tt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const long long row = m0 + 16 * i + c;
                    if (row < rows) *reinterpret_cast<f32x4*>(out + row * ld + 16 * tt + 4 * q) = v;
                }
        }
    }
}
int main() {
    const long long rows = 4096ll * 741; const int ld = 256;
    float* out; CK(hipMalloc(&out, rows * ld * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int pat = 0; pat < 2; ++pat)
        for (int grid : {256, 512, 1024, 4096}) {
            float best = 1e9f;
            for (int r = 0; r < 5; ++r) {
                CK(hipEventRecord(e0));
                if (pat == 0) store_kernel<0><<<grid, 256>>>(out, rows, ld); else store_kernel<1><<<grid, 256>>>(out, rows, ld);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best = best < ms ? best : ms;
            }
            printf("%s, %4d blocks: %.3f ms for %.2f GB = %.2f TB/s\n", pat == 0 ? "contiguous 16 B per lane        " : "accumulator layout (16 x 64 B)  ", grid, best, rows * ld * 4e-9, rows * ld * 4e-9 / best);
        }
    return 0;
}
