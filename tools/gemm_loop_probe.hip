// What does one k-step of the 64x64x16 f32 MFMA GEMM cost, piece by piece?  4 waves per block, each: [LDS writes] [barrier]
// [8 ds_read2 fragment reads] [8 dependent v_mfma_f32_32x32x2_f32].  Knobs are template flags; prints cycles per step.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool WR, bool BAR, bool RD, bool MF, bool SB>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float a, float b) {
    __shared__ __attribute__((aligned(16))) float smem[4 * 16 * 68];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float fa[8], fb[8];
    for (int j = 0; j < 8; ++j) { fa[j] = a + j; fb[j] = b - j; }
    for (int i = t; i < 4 * 16 * 68; i += 256) smem[i] = (float)i * 1e-6f;
    __syncthreads();
    const int rdA = (lane >> 5) * 68 + (wave >> 1) * 32 + (lane & 31);
    const int rdB = (lane >> 5) * 68 + (wave & 1) * 32 + (lane & 31);
    float4 v = make_float4(a, b, a, b);
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        float* As = smem + (it & 1) * 16 * 68;
        float* Bs = smem + (2 + (it & 1)) * 16 * 68;
        if (WR) {
            const int mn = t >> 2, k = (t & 3) * 4;
            As[(k + 0) * 68 + mn] = v.x; As[(k + 1) * 68 + mn] = v.y; As[(k + 2) * 68 + mn] = v.z; As[(k + 3) * 68 + mn] = v.w;
            *reinterpret_cast<float4*>(&Bs[(t >> 4) * 68 + (t & 15) * 4]) = v;
        }
        if (BAR) __syncthreads();
        float na[8], nb[8];
        if (RD) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { na[j] = As[rdA + 2 * j * 68]; nb[j] = Bs[rdB + 2 * j * 68]; }
        }
        if (SB) __builtin_amdgcn_sched_barrier(0);
        if (MF) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], fb[j], acc, 0, 0, 0);
        }
        if (SB) __builtin_amdgcn_sched_barrier(0);
        if (RD) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { fa[j] = na[j]; fb[j] = nb[j]; }
        }
        v.x += 1e-9f;
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int j = 0; j < 8; ++j) s += fa[j] + fb[j];
    if (t == 0 && blockIdx.x == 0) { out[0] = s; out[1] = (float)(t1 - t0); }
    if (s == 12345.678f) out[2] = s;
}

template <bool WR, bool BAR, bool RD, bool MF, bool SB>
void run(const char* name, int blocks, int iters) {
    float* d; hipMalloc(&d, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<WR, BAR, RD, MF, SB><<<blocks, 256>>>(d, iters, 1.0f, 0.5f); hipDeviceSynchronize();
    hipEventRecord(e0); probe<WR, BAR, RD, MF, SB><<<blocks, 256>>>(d, iters, 1.0f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("%-38s blocks=%4d: %7.1f clk/step (wave 0), %.3f ms total = %.0f ns/step\n", name, blocks, h[1] / iters, ms, ms * 1e6 / iters);
    hipFree(d);
}

int main() {
    const int it = 2000;
    for (int blocks : {256, 512}) {
        run<false, false, false, true, false>("mfma only", blocks, it);
        run<false, false, true, true, false>("reads + mfma", blocks, it);
        run<false, false, true, true, true>("reads + mfma (sched_barrier)", blocks, it);
        run<false, true, true, true, false>("barrier + reads + mfma", blocks, it);
        run<true, true, true, true, false>("writes + barrier + reads + mfma", blocks, it);
        run<true, true, true, true, true>("writes + barrier + reads + mfma (sb)", blocks, it);
        run<true, true, true, false, false>("writes + barrier + reads (no mfma)", blocks, it);
        run<true, true, false, true, false>("writes + barrier + mfma (no reads)", blocks, it);
    }
    return 0;
}
