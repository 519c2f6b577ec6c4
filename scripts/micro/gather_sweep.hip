// Sweep of the bare C4 access pattern (see gather_ceiling.hip) over gather depth (rows in flight per lane group) and
// occupancy (waves per SIMD, capped with a dynamic-LDS allocation): which shape does the memory system like?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int DEPTH>
__global__ void __launch_bounds__(256) k_gather(int n, const int *__restrict__ rowptr, const int *__restrict__ src,
                                                const float4 *__restrict__ S, const float4 *__restrict__ C,
                                                float4 *__restrict__ out) {
    extern __shared__ char pad[];
    const int lane = threadIdx.x & 15;
    const int groups = (blockDim.x >> 4) * gridDim.x;
    for (int j = blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4); j < n; j += groups) {
        const int beg = rowptr[j], end = rowptr[j + 1];
        float4 acc = C[(size_t)j * 16 + lane];
        float4 own = S[(size_t)j * 16 + lane];
        acc.x += own.x; acc.y += own.y; acc.z += own.z; acc.w += own.w;
        for (int e = beg; e < end; e += DEPTH) {
            float4 v[DEPTH];
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) {
                const bool ok = e + i < end;
                const int s = ok ? src[e + i] : 0;
                v[i] = ok ? S[(size_t)s * 16 + lane] : make_float4(0, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
        }
        out[(size_t)j * 16 + lane] = acc;
    }
}

int main(int argc, char **argv) {
    const int N = 1000000; const long E = 10000000;
    std::mt19937_64 rng(1234);
    std::vector<int> dst(E), srcv(E), rowptr(N + 1, 0);
    for (long e = 0; e < E; ++e) { dst[e] = rng() % N; srcv[e] = rng() % N; rowptr[dst[e] + 1]++; }
    for (int j = 0; j < N; ++j) rowptr[j + 1] += rowptr[j];
    std::vector<int> fill(rowptr.begin(), rowptr.end() - 1), src(E);
    for (long e = 0; e < E; ++e) src[fill[dst[e]]++] = srcv[e];
    int *d_rowptr, *d_src; float4 *S, *C, *O;
    CK(hipMalloc(&d_rowptr, (N + 1) * 4)); CK(hipMalloc(&d_src, E * 4));
    CK(hipMalloc(&S, (size_t)N * 256)); CK(hipMalloc(&C, (size_t)N * 256)); CK(hipMalloc(&O, (size_t)N * 256));
    CK(hipMemcpy(d_rowptr, rowptr.data(), (N + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_src, src.data(), E * 4, hipMemcpyHostToDevice));
    CK(hipMemset(S, 0, (size_t)N * 256)); CK(hipMemset(C, 0, (size_t)N * 256));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const double bytes = (double)E * 260 + (double)N * (4 + 3 * 256);
    printf("waves/SIMD  depth   us/iter   TB/s\n");
    for (int occ : {2, 3, 4, 6, 8}) {
        const int lds = 160 * 1024 / occ - 512;
        auto run = [&](int depth, auto kern) {
            CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            const int grid = 256 * occ;
            for (int i = 0; i < 3; ++i) kern<<<grid, 256, lds>>>(N, d_rowptr, d_src, S, C, O);
            CK(hipEventRecord(a));
            for (int i = 0; i < 20; ++i) kern<<<grid, 256, lds>>>(N, d_rowptr, d_src, S, C, O);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("%6d %8d %10.1f %7.2f\n", occ, depth, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
        };
        run(2, k_gather<2>); run(4, k_gather<4>); run(8, k_gather<8>); run(12, k_gather<12>); run(16, k_gather<16>);
    }
    CK(hipGetLastError());
    return 0;
}
