// Issue-model probe for v_mfma_f32_32x32x16_bf16 (round 6): cycles per MFMA by (a) the distance between two MFMAs on the same accumulator
// and (b) the number of split-chain VALU ops (v_cvt_pk_bf16_f32, shift, and, v_sub_f32) between two MFMAs, one wave per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o tools/_bin/mfma32_mix tools/mfma32_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned pk(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2)); }
__device__ __forceinline__ float fsub(float a, float b) { float r; asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// NACC accumulators in rotation (distance between two MFMAs on the same accumulator), NP split pairs (11 VALU each, independent of each
// other) issued per group of 4 MFMAs, CHAINS = how many of the pairs are interleaved instruction by instruction (1: one dependent chain)
template <int NACC, int NP>
__global__ __launch_bounds__(256, 1) void probe(float* out, const float* in, int iters) {
    f32x16 acc[NACC];
    for (int c = 0; c < NACC; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
    float x[2 * (NP > 0 ? NP : 1)];
    for (int i = 0; i < 2 * NP; ++i) x[i] = in[threadIdx.x + 256 * i];
    unsigned sink = 0;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {          // 8 groups of 4 MFMAs
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = (4 * g + k) % NACC;
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[c], 0, 0, 0);
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                float x0 = x[2 * p], x1 = x[2 * p + 1];
                unsigned h = pk(x0, x1);
                float r0 = fsub(x0, __uint_as_float(h << 16)), r1 = fsub(x1, __uint_as_float(h & 0xffff0000u));
                unsigned m = pk(r0, r1);
                float s0 = fsub(r0, __uint_as_float(m << 16)), s1 = fsub(r1, __uint_as_float(m & 0xffff0000u));
                unsigned l = pk(s0, s1);
                sink ^= h ^ m ^ l;
                x[2 * p] = s0; x[2 * p + 1] = s1;
            }
            // spread: one MFMA, then NP * 14 / 4 VALU
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                if (NP > 0) __builtin_amdgcn_sched_group_barrier(0x2, (NP * 14 + 3) / 4, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int c = 0; c < NACC; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
    for (int i = 0; i < 2 * NP; ++i) s += x[i];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = s + sink; out[1] = (float)(t1 - t0); }
}
template <int NACC, int NP>
void run(int iters) {
    float *d, *in; hipMalloc(&d, 64); hipMalloc(&in, 256 * 64 * 4); hipMemset(in, 0, 256 * 64 * 4);
    probe<NACC, NP><<<256, 256>>>(d, in, 10);
    hipDeviceSynchronize();
    probe<NACC, NP><<<256, 256>>>(d, in, iters);
    hipDeviceSynchronize();
    float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("accumulators in rotation %d, VALU per MFMA %.1f: %.1f clk per MFMA (one wave per SIMD)\n", NACC, NP * 14 / 4.0, h[1] / (iters * 32.0));
    hipFree(d); hipFree(in);
}
int main() {
    run<1, 0>(2000); run<2, 0>(2000); run<4, 0>(2000); run<8, 0>(2000);
    run<2, 1>(2000); run<2, 2>(2000); run<2, 3>(2000); run<2, 4>(2000);
    run<4, 1>(2000); run<4, 2>(2000); run<4, 3>(2000); run<4, 4>(2000);
    run<8, 1>(2000); run<8, 2>(2000); run<8, 3>(2000); run<8, 4>(2000);
    return 0;
}
