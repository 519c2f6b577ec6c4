// Grid barrier + row exchange latency between a few resident workgroups: through memory (agent scope, sc1: what the persistent kernels
// of kernels_train_small.hpp did first) against through the L2 of ONE XCD (workgroups 0, 8, 16, .. of a launch land on XCD 0; the other
// workgroups exit at once).  hipcc --offload-arch=gfx950 -O3 gridbar_bench.hip -o gridbar_bench && ./gridbar_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void k_where(int *xcc) { if (threadIdx.x == 0) xcc[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (31 << 11)) & 15; }

template <bool L2>
__global__ void __launch_bounds__(256) k_bar(unsigned long long *bar, float *rows, int n_act, int stride, int iters, int *xcc, long long *ticks, int *bad) {
    if (blockIdx.x % stride != 0) return;
    const int wg = blockIdx.x / stride, tid = threadIdx.x;
    if (tid == 0) xcc[wg] = __builtin_amdgcn_s_getreg(20 | (31 << 11)) & 15;
    __shared__ int dummy;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(rows, 0, 0x7FFFFFF0, 0x00020000);
    int errors = 0, break_all = 0;
    const long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        // every thread publishes 16 bytes of "its row" of this iteration (a fresh address per iteration, as the tape of states is)
        const unsigned off = (((unsigned)it * n_act + wg) * 256u + tid) * 16u;
        const u32x4 v = {(unsigned)it, (unsigned)wg, (unsigned)tid, 7u};
        if (L2) __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 0); else __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            unsigned long long *ctr = bar + (it & 1);
            const unsigned target = (unsigned)(it / 2 + 1) * n_act;
            if (L2) {
                __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                int spin = 0;
                while ((unsigned)__hip_atomic_fetch_add(ctr, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target && ++spin < (1 << 16)) __builtin_amdgcn_s_sleep(1);
                if (spin >= (1 << 16)) { atomicAdd(bad, 1 << 20); break_all = 1; }
            } else {
                __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int spin = 0;
                while ((unsigned)__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spin < (1 << 16)) __builtin_amdgcn_s_sleep(1);
                if (spin >= (1 << 16)) { atomicAdd(bad, 1 << 20); break_all = 1; }
            }
            dummy = break_all;
        }
        __syncthreads();
        if (dummy) break;
        // read the row of the next workgroup
        const int other = (wg + 1) % n_act;
        const unsigned off2 = (((unsigned)it * n_act + other) * 256u + tid) * 16u;
        const u32x4 g = L2 ? __builtin_amdgcn_raw_buffer_load_b128(r, off2, 0, 0) : __builtin_amdgcn_raw_buffer_load_b128(r, off2, 0, 16);
        if (g[0] != (unsigned)it || g[1] != (unsigned)other || g[2] != (unsigned)tid) ++errors;
    }
    const long long t1 = wall_clock64();
    if (tid == 0) ticks[wg] = t1 - t0;
    if (errors) atomicAdd(bad, errors);
}

int main(int argc, char **argv) {
    const int iters = 2000;
    unsigned long long *bar; float *rows; int *xcc, *bad; long long *ticks;
    CK(hipMalloc(&bar, 16)); CK(hipMalloc(&rows, (size_t)iters * 32 * 256 * 16)); CK(hipMalloc(&xcc, 4 * 64)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&ticks, 8 * 64));
    {
        k_where<<<64, 64>>>(xcc);
        CK(hipDeviceSynchronize());
        std::vector<int> hx(64);
        CK(hipMemcpy(hx.data(), xcc, 4 * 64, hipMemcpyDeviceToHost));
        printf("XCC id of workgroups 0 .. 63 of a launch:");
        for (int i = 0; i < 64; ++i) printf(" %d", hx[i]);
        printf("\n"); fflush(stdout);
    }
    for (int n_act : {2, 8, 16, 32}) {
        for (int mode = 0; mode < 3; ++mode) {      // 0: memory path, workgroups on all XCDs; 1: memory path, one XCD; 2: L2 path, one XCD
            const int stride = mode == 0 ? 1 : 8;
            CK(hipMemset(bar, 0, 16)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(rows, 0xff, (size_t)iters * 32 * 256 * 16));
            const size_t lds = 100 * 1024;
            if (mode == 2) { CK(hipFuncSetAttribute((const void *)&k_bar<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                             k_bar<true><<<n_act * stride, 256, lds>>>(bar, rows, n_act, stride, iters, xcc, ticks, bad); }
            else { CK(hipFuncSetAttribute((const void *)&k_bar<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                   k_bar<false><<<n_act * stride, 256, lds>>>(bar, rows, n_act, stride, iters, xcc, ticks, bad); }
            CK(hipDeviceSynchronize());
            std::vector<int> hx(64); std::vector<long long> ht(64); int hb = 0;
            CK(hipMemcpy(hx.data(), xcc, 4 * 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(ht.data(), ticks, 8 * 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
            printf("workgroups %2d  %-28s  %.2f us per barrier + exchange   wrong values %d   XCC ids:", n_act,
                   mode == 0 ? "memory path, all XCDs" : mode == 1 ? "memory path, one XCD" : "L2 path, one XCD", ht[0] / 100.0 / iters, hb);
            for (int i = 0; i < n_act && i < 16; ++i) printf(" %d", hx[i]);
            printf("\n"); fflush(stdout);
        }
    }
    return 0;
}
