// Do v_mfma_f32_16x16x32_bf16 (16 cycles of the matrix pipe each) and ordinary VALU instructions overlap on a gfx950 SIMD
//   (a) when they alternate inside ONE wave's instruction stream, (b) when one wave of the SIMD issues only MFMAs and the other only VALU?
// One workgroup of 512 threads per CU (waves w and w + 4 share SIMD w), every wave does ITERS iterations of its role:
//   mode 0: 24 MFMAs                                  (matrix pipe alone)
//   mode 1: 96 v_fma_f32                              (VALU alone; 4 cycles each)
//   mode 2: 24 MFMAs, then 96 v_fma                   (phases, as k_train_fwd_b6 runs them)
//   mode 3: (1 MFMA, 4 v_fma) x 24                    (interleaved in the stream)
//   mode 4: waves 0-3: 48 MFMAs, waves 4-7: 192 v_fma (specialised waves; the same work per SIMD as modes 2 / 3)
// Times are wall clock (hipEvents over the launch), reported as ns per iteration-pair of a SIMD and as a fraction of the sum of the two
// pipes' times alone.   hipcc -O3 --offload-arch=gfx950 mfma_bf16_valu_overlap.hip -o bin/mfma_bf16_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ void __launch_bounds__(512, 2) k(float *out, int iters, float a0) {
    const int wave = threadIdx.x >> 6;
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a0 + i + threadIdx.x;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(a0 + i); b[i] = (__bf16)(a0 - i); }
#define MF(i_) acc[(i_) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[(i_) & 3], 0, 0, 0)
#define VF(i_) v[(i_) & 7] = __builtin_fmaf(v[(i_) & 7], 1.0001f, 0.5f)
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 24; ++i) MF(i);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 96; ++i) VF(i);
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 24; ++i) MF(i);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 96; ++i) VF(i);
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                MF(i);
                __builtin_amdgcn_sched_barrier(0);
                VF(4 * i); VF(4 * i + 1); VF(4 * i + 2); VF(4 * i + 3);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            if (wave < 4) {
#pragma unroll
                for (int i = 0; i < 48; ++i) MF(i);
            } else {
#pragma unroll
                for (int i = 0; i < 192; ++i) VF(i);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
double run(float *out, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<MODE><<<256, 512>>>(out, iters, 1.0f);
    CK(hipEventRecord(e0));
    k<MODE><<<256, 512>>>(out, iters, 1.0f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e6 / iters;            // ns per iteration
}

int main() {
    float *out; CK(hipMalloc(&out, 64));
    const int iters = 20000;
    const double m = run<0>(out, iters), v = run<1>(out, iters), ph = run<2>(out, iters), il = run<3>(out, iters), sp = run<4>(out, iters);
    printf("per iteration, two waves per SIMD (each SIMD: 48 MFMAs of 16 cycles = 768 cycles, 192 v_fma of 4 cycles = 768 cycles)\n");
    printf("  MFMAs alone        %8.1f ns\n  v_fma alone        %8.1f ns\n  phases in a wave   %8.1f ns  (sum of the two alone: %.1f)\n  interleaved        %8.1f ns\n  specialised waves  %8.1f ns\n",
           m, v, ph, m + v, il, sp);
    return 0;
}
