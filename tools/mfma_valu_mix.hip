// Issue-model probe for the split-precision GEMM (gemm_dr3_kernel): how many VALU ops of the split (v_cvt_pk_bf16_f32, shifts, ands,
// v_sub_f32) hide beside one v_mfma_f32_16x16x32_bf16, with one and with two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o tools/_bin/mfma_valu_mix tools/mfma_valu_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned pk(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2)); }

// NF = split "pairs" (11 VALU each) per 4 MFMAs; KIND 0: the real split chain, 1: independent v_sub only (same count), 2: v_and only
template <int NF, int KIND, int W>
__global__ __launch_bounds__(256, W) void probe(float* out, const float* in, int iters) {
    f32x4 acc[8];
    for (int c = 0; c < 8; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
    float x[2 * (NF > 0 ? NF : 1)];
    for (int i = 0; i < 2 * NF; ++i) x[i] = in[threadIdx.x + 256 * i];
    unsigned sink = 0;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 8; c += 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[c + k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[c + k], 0, 0, 0);
#pragma unroll
            for (int p = 0; p < NF; ++p) {
                float x0 = x[2 * p], x1 = x[2 * p + 1];
                if (KIND == 0) {
                    unsigned h = pk(x0, x1);
                    float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
                    unsigned m = pk(r0, r1);
                    float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
                    unsigned l = pk(s0, s1);
                    sink ^= h ^ m ^ l;                  // (+3 xors: 14 per pair)
                    x[2 * p] = x0 + 1.0f; x[2 * p + 1] = x1 + 1.0f;   // (+2)
                } else if (KIND == 1) {
                    for (int r = 0; r < 8; ++r) { x0 = x0 - 1.5f; x1 = x1 - 0.5f; }
                    x[2 * p] = x0; x[2 * p + 1] = x1;
                } else {
                    unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
                    for (int r = 0; r < 8; ++r) { u0 = (u0 & 0xfffffff0u) + 3u; u1 = (u1 << 1) ^ u0; }
                    x[2 * p] = __uint_as_float(u0); x[2 * p + 1] = __uint_as_float(u1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int c = 0; c < 8; ++c) for (int i = 0; i < 4; ++i) s += acc[c][i];
    for (int i = 0; i < 2 * NF; ++i) s += x[i];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = s + sink; out[1] = (float)(t1 - t0); }
}
template <int NF, int KIND, int W>
void run(int iters) {
    float *d, *in; hipMalloc(&d, 64); hipMalloc(&in, 256 * 64 * 4); hipMemset(in, 0, 256 * 64 * 4);
    const int blocks = 256 * W;
    probe<NF, KIND, W><<<blocks, 256>>>(d, in, iters); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); probe<NF, KIND, W><<<blocks, 256>>>(d, in, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    const double nm = (double)iters * 8;
    const int valu = KIND == 0 ? 16 * NF : 16 * NF;
    printf("kind %d  waves/SIMD %d  VALU per MFMA %.2f: %.1f clk per MFMA per wave  -> %.1f clk per MFMA per SIMD  (%.3f ms, clock %.2f GHz)\n", KIND, W, valu / 4.0,
           h[1] / nm, h[1] / nm / W, ms, h[1] / (ms * 1e6));
    hipFree(d); hipFree(in);
}
int main() {
    const int it = 20000;
    run<0, 0, 1>(it); run<1, 0, 1>(it); run<2, 0, 1>(it); run<3, 0, 1>(it); run<4, 0, 1>(it);
    run<0, 0, 2>(it); run<1, 0, 2>(it); run<2, 0, 2>(it); run<3, 0, 2>(it); run<4, 0, 2>(it);
    run<1, 1, 1>(it); run<2, 1, 1>(it); run<3, 1, 1>(it);
    run<1, 1, 2>(it); run<2, 1, 2>(it); run<3, 1, 2>(it);
    run<1, 2, 1>(it); run<2, 2, 1>(it); run<3, 2, 1>(it);
    run<1, 2, 2>(it); run<2, 2, 2>(it); run<3, 2, 2>(it);
    return 0;
}
