// f32 MFMA issue-rate probe: NCH independent accumulator chains per wave, `waves` waves per SIMD, all 256 CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NCH, bool SMALL>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float a, float b) {
    f32x16 acc[NCH]; f32x4 acs[NCH];
    for (int c = 0; c < NCH; ++c) { for (int i = 0; i < 16; ++i) acc[c][i] = 0.f; for (int i = 0; i < 4; ++i) acs[c][i] = 0.f; }
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (SMALL) acs[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acs[c], 0, 0, 0);
            else acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        }
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int c = 0; c < NCH; ++c) { for (int i = 0; i < 16; ++i) s += acc[c][i]; for (int i = 0; i < 4; ++i) s += acs[c][i]; }
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = s; out[1] = (float)(t1 - t0); }
}
template <int NCH, bool SMALL>
void run(int blocks, int iters) {
    float* d; hipMalloc(&d, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NCH, SMALL><<<blocks, 256>>>(d, iters, 1.0f, 0.5f); hipDeviceSynchronize();
    hipEventRecord(e0); probe<NCH, SMALL><<<blocks, 256>>>(d, iters, 1.0f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    double n = (double)iters * NCH;
    double flop = (SMALL ? 2.0 * 16 * 16 * 4 : 2.0 * 32 * 32 * 2) * n * blocks * 4;
    printf("%s NCH=%d blocks=%d (%.2f waves/SIMD): %.1f clk/MFMA/wave, %.3f ms, %.1f TF, clock %.2f GHz\n", SMALL ? "16x16x4" : "32x32x2", NCH, blocks,
           blocks / 256.0, h[1] / n, ms, flop / ms / 1e9, h[1] / (ms * 1e6));
}
int main() {
    int it = 4000;
    run<1, false>(256, it); run<2, false>(256, it); run<4, false>(256, it);
    run<1, false>(512, it); run<2, false>(512, it); run<4, false>(512, it);
    run<1, false>(1024, it); run<1, false>(448, it); run<4, false>(448, it);
    run<1, true>(256, it); run<2, true>(256, it); run<4, true>(256, it); run<4, true>(512, it);
    return 0;
}
