// How many workgroups of 256 threads does a CU of this GPU really hold at once, as a function of the dynamic LDS size and the
// register count?  (hipOccupancyMaxActiveBlocksPerMultiprocessor answers 2 for k_train_fwd_b6 - 237 registers, 66 704 bytes of LDS -
// and the time stamps of its workgroups say 1.)  Every workgroup of a 512-workgroup launch stamps the 100 MHz clock, spins ~40 us and
// stamps again: with 2 resident per CU all 512 start together, with 1 the second half starts when the first ends.
//   hipcc -O3 --offload-arch=gfx950 occupancy_probe.hip -o bin/occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__device__ unsigned long long g_t[2 * 4096];

template <int NREG, int MINB>
__global__ void __launch_bounds__(256, MINB) k_probe(float *out, int spin) {
    extern __shared__ float smem[];
    if (threadIdx.x == 0) g_t[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    float r[NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i) r[i] = (float)(threadIdx.x + i);
    smem[threadIdx.x] = 1.0f;
    __syncthreads();
    for (int it = 0; it < spin; ++it) {
#pragma unroll
        for (int i = 0; i < NREG; ++i) r[i] = fmaf(r[i], 1.0000001f, smem[(threadIdx.x + i) & 255]);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NREG; ++i) s += r[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
    if (threadIdx.x == 0) g_t[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
}

template <int NREG, int MINB>
void probe(const char *name, size_t lds, float *out) {
    const int grid = 512;
    CK(hipFuncSetAttribute((const void *)k_probe<NREG, MINB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int nb = -1; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_probe<NREG, MINB>, 256, lds));
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, (const void *)k_probe<NREG, MINB>));
    const int spin = 20000 / NREG;
    k_probe<NREG, MINB><<<grid, 256, lds>>>(out, spin); CK(hipDeviceSynchronize());
    k_probe<NREG, MINB><<<grid, 256, lds>>>(out, spin); CK(hipDeviceSynchronize());
    std::vector<unsigned long long> t(2 * 4096); CK(hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_t), t.size() * 8));
    unsigned long long t0 = ~0ull; for (int b = 0; b < grid; ++b) t0 = std::min(t0, t[2 * b]);
    double dur = 0; for (int b = 0; b < grid; ++b) dur += (t[2 * b + 1] - t[2 * b]) * 0.01 / grid;
    int early = 0; for (int b = 0; b < grid; ++b) if ((t[2 * b] - t0) * 0.01 < 0.5 * dur) ++early;
    printf("%-22s registers %3d, dynamic LDS %6zu: API says %d per CU; %3d of %d workgroups start at once (a workgroup lasts %.0f us) -> %s\n", name, fa.numRegs, lds, nb,
           early, grid, dur, early > 400 ? "2 per CU" : "1 per CU");
}

int main() {
    float *out; CK(hipMalloc(&out, 4096));
    for (size_t lds : {(size_t)1024, (size_t)32768, (size_t)49152, (size_t)65536 - 64, (size_t)65536 + 1024, (size_t)81920 - 64, (size_t)81920 + 1024}) {
        probe<64, 2>("64 values a lane", lds, out);
        probe<200, 2>("200 values a lane", lds, out);
    }
    return 0;
}
