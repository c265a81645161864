// Where does the dispatcher put the workgroups of a 448-block / 256-thread / 17 KB-LDS grid?  (speed experiment only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void census(unsigned* out, int spin) {
    __shared__ float pad[4352];
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
    long long t0 = clock64();
    float acc = pad[(threadIdx.x * 7) & 255];
    while (clock64() - t0 < spin) acc = acc * 1.0001f + 0.5f;
    if (threadIdx.x == 0) { out[blockIdx.x * 3] = hw; out[blockIdx.x * 3 + 1] = xcc; out[blockIdx.x * 3 + 2] = 0; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) atomicOr(&out[blockIdx.x * 3 + 2], (((hw >> 4) & 3) << (4 * (threadIdx.x >> 6))) | (acc > 1e30f ? 1u << 31 : 0));
}
int main(int argc, char** argv) {
    int nb = argc > 1 ? atoi(argv[1]) : 448, spin = argc > 2 ? atoi(argv[2]) : 40000;
    unsigned* d; hipMalloc(&d, nb * 12);
    std::vector<unsigned> h(nb * 3);
    for (int rep = 0; rep < 2; ++rep) {
        census<<<nb, 256>>>(d, spin); hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d, nb * 12, hipMemcpyDeviceToHost);
    std::map<unsigned, int> percu;
    for (int b = 0; b < nb; ++b) {
        unsigned hw = h[b * 3], xcc = h[b * 3 + 1] & 0xf;
        unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        percu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
        if (b < 16) printf("block %d: xcc %u se %u sh %u cu %u simd %u\n", b, xcc, se, sh, cu, (hw >> 4) & 3);
    }
    std::map<int, int> hist;
    for (auto& kv : percu) hist[kv.second]++;
    std::map<unsigned,int> pat; for (int b = 0; b < nb; ++b) pat[h[b*3+2] & 0xffff]++;
    for (auto& kv : pat) printf("  simd pattern (wave3..wave0) %04x : %d blocks\n", kv.first, kv.second);
    printf("distinct CUs used: %zu\n", percu.size());
    for (auto& kv : hist) printf("  CUs with %d blocks: %d\n", kv.first, kv.second);
    return 0;
}
