// Peak rate of the exact-f32 MFMA instructions on this chip, no memory traffic: every wave runs a chain-free loop of
// v_mfma_f32_16x16x4_f32 (or 32x32x2) on NACC independent accumulators.  Build + run:
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
// Prints TFLOP/s for 1, 2, 4 waves per SIMD and the implied clock if an MFMA 16x16x4 takes 32 cycles (8 passes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) k16(float *out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}

template <int NACC>
__global__ void __launch_bounds__(256) k32(float *out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][5];
    if (s == 12345.678f) out[0] = s;
}

template <typename K>
double run(K kern, int blocks, int iters, float *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<blocks, 256>>>(out, iters, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<blocks, 256>>>(out, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s: %d CUs, clockRate %d kHz\n", p.name, cus, p.clockRate);
    float *out;
    hipMalloc(&out, 4);
    const int iters = 200000;
    for (int wps = 1; wps <= 4; wps *= 2) {                       // waves per SIMD = workgroups (4 waves) per CU
        const int blocks = cus * wps;
        {
            const double s = run(k16<4>, blocks, iters, out);
            const double flop = (double)blocks * 4 * iters * 4 * 2048.0;
            const double per_simd = (double)wps * iters * 4;         // MFMAs per SIMD
            printf("16x16x4 f32, 4 accumulators, %d wave(s)/SIMD: %.1f TFLOP/s, %.3f ms, %.0f MHz if 32 cycles each\n", wps, flop / s * 1e-12,
                   s * 1e3, per_simd * 32 / s * 1e-6);
        }
        {
            const double s = run(k32<2>, blocks, iters, out);
            const double flop = (double)blocks * 4 * iters * 2 * 4096.0;
            printf("32x32x2 f32, 2 accumulators, %d wave(s)/SIMD: %.1f TFLOP/s, %.3f ms\n", wps, flop / s * 1e-12, s * 1e3);
        }
    }
    // a long run (about a second): does the rate hold?
    {
        const int blocks = cus * 2;
        const double s = run(k16<4>, blocks, iters * 20, out);
        printf("16x16x4 f32 sustained over %.0f ms: %.1f TFLOP/s\n", s * 1e3, (double)blocks * 4 * iters * 20 * 4 * 2048.0 / s * 1e-12);
    }
    return 0;
}
