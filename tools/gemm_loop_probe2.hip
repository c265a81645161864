// Second probe: the GEMM k-loop unrolled by two with alternating fragment sets (no register copies), optional global loads
// feeding the LDS writes.  VARIANT 0: loads after the barrier (old kernel); 1: loads at the top of the step into a second
// staging set; 2: like 1 with sched_barriers around the MFMA group.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define LDS_W(As, Bs, VV, WW)                                                                              \
    { const int mn = t >> 2, k = (t & 3) * 4;                                                             \
      (As)[(k + 0) * 68 + mn] = (VV).x; (As)[(k + 1) * 68 + mn] = (VV).y; (As)[(k + 2) * 68 + mn] = (VV).z; (As)[(k + 3) * 68 + mn] = (VV).w; \
      *reinterpret_cast<float4*>(&(Bs)[(t >> 4) * 68 + (t & 15) * 4]) = (WW); }
#define LDS_R(As, Bs, fa, fb) _Pragma("unroll") for (int j = 0; j < 8; ++j) { fa[j] = (As)[rdA + 2 * j * 68]; fb[j] = (Bs)[rdB + 2 * j * 68]; }
#define MFMA8(fa, fb) _Pragma("unroll") for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], fb[j], acc, 0, 0, 0);

template <int VARIANT, bool GL>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ ga, const float* __restrict__ gb, float* out, int iters, int stride) {
    __shared__ __attribute__((aligned(16))) float smem[4 * 16 * 68];
    float* As0 = smem; float* As1 = smem + 16 * 68; float* Bs0 = smem + 2 * 16 * 68; float* Bs1 = smem + 3 * 16 * 68;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float f0a[8], f0b[8], f1a[8], f1b[8];
    for (int i = t; i < 4 * 16 * 68; i += 256) smem[i] = (float)i * 1e-6f;
    __syncthreads();
    const int rdA = (lane >> 5) * 68 + (wave >> 1) * 32 + (lane & 31);
    const int rdB = (lane >> 5) * 68 + (wave & 1) * 32 + (lane & 31);
    const float* pa = ga + (size_t)(blockIdx.x % 64) * 4096 + (t >> 2) * 16 + (t & 3) * 4;
    const float* pb = gb + (size_t)(blockIdx.x % 64) * 4096 + t * 4;
    float4 a0 = make_float4(1, 2, 3, 4), b0 = a0, a1 = a0, b1 = a0;
    LDS_R(As0, Bs0, f0a, f0b);
    long long t0 = clock64();
    for (int it = 0; it < iters; it += 2) {
        if (VARIANT == 0) {
            LDS_W(As1, Bs1, a0, b0); __syncthreads();
            if (GL) { a0 = *reinterpret_cast<const float4*>(pa + (size_t)(it & 15) * stride); b0 = *reinterpret_cast<const float4*>(pb + (size_t)(it & 15) * stride); }
            LDS_R(As1, Bs1, f1a, f1b); MFMA8(f0a, f0b);
            LDS_W(As0, Bs0, a0, b0); __syncthreads();
            if (GL) { a0 = *reinterpret_cast<const float4*>(pa + (size_t)((it + 1) & 15) * stride); b0 = *reinterpret_cast<const float4*>(pb + (size_t)((it + 1) & 15) * stride); }
            LDS_R(As0, Bs0, f0a, f0b); MFMA8(f1a, f1b);
        } else {
            if (GL) { a0 = *reinterpret_cast<const float4*>(pa + (size_t)(it & 15) * stride); b0 = *reinterpret_cast<const float4*>(pb + (size_t)(it & 15) * stride); }
            LDS_W(As1, Bs1, a1, b1); __syncthreads();
            LDS_R(As1, Bs1, f1a, f1b);
            if (VARIANT == 2) __builtin_amdgcn_sched_barrier(0);
            MFMA8(f0a, f0b);
            if (VARIANT == 2) __builtin_amdgcn_sched_barrier(0);
            if (GL) { a1 = *reinterpret_cast<const float4*>(pa + (size_t)((it + 1) & 15) * stride); b1 = *reinterpret_cast<const float4*>(pb + (size_t)((it + 1) & 15) * stride); }
            LDS_W(As0, Bs0, a0, b0); __syncthreads();
            LDS_R(As0, Bs0, f0a, f0b);
            if (VARIANT == 2) __builtin_amdgcn_sched_barrier(0);
            MFMA8(f1a, f1b);
            if (VARIANT == 2) __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (t == 0 && blockIdx.x == 0) { out[0] = s; out[1] = (float)(t1 - t0); }
    if (s == 12345.678f) out[2] = s + a0.x + b1.y;
}

template <int VARIANT, bool GL>
void run(const char* name, int blocks, int iters, const float* ga, const float* gb) {
    float* d; hipMalloc(&d, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<VARIANT, GL><<<blocks, 256>>>(ga, gb, d, iters, 256); hipDeviceSynchronize();
    hipEventRecord(e0); probe<VARIANT, GL><<<blocks, 256>>>(ga, gb, d, iters, 256); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("%-44s blocks=%4d: %7.1f clk/step (wave 0), %.0f ns/step\n", name, blocks, h[1] / iters, ms * 1e6 / iters);
    hipFree(d);
}

int main() {
    const int it = 2000;
    float *ga, *gb; hipMalloc(&ga, 64 * 4096 * 4 + 65536); hipMalloc(&gb, 64 * 4096 * 4 + 65536);
    hipMemset(ga, 0, 64 * 4096 * 4 + 65536); hipMemset(gb, 0, 64 * 4096 * 4 + 65536);
    for (int blocks : {256, 448, 512}) {
        run<0, false>("unroll2, no global loads", blocks, it, ga, gb);
        run<0, true>("unroll2, loads after barrier", blocks, it, ga, gb);
        run<1, true>("unroll2, loads at top (2 staging sets)", blocks, it, ga, gb);
        run<2, true>("unroll2, loads at top + sched_barrier", blocks, it, ga, gb);
    }
    return 0;
}
