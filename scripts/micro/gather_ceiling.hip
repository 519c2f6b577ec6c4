// Micro-benchmark: what does the bare access pattern of one C4 iteration cost on this GPU, with no arithmetic beyond
// the row sum?  Per destination node: read rowptr, read its source ids, gather the 256-B source rows, read own row
// (256 B) and the constant row (256 B), write one 256-B row.  Same algorithmic bytes as the fused iteration kernel
// (SURVEY.md 8d): E*(4+256) + N*(4+3*256).   Build: hipcc -O3 --offload-arch=gfx950 gather_ceiling.hip -o gather_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int DEPTH>
__global__ void __launch_bounds__(256) k_gather(int n, const int *__restrict__ rowptr, const int *__restrict__ src,
                                                const float4 *__restrict__ S, const float4 *__restrict__ C,
                                                float4 *__restrict__ out) {
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

__global__ void k_stream(size_t n4, const float4 *__restrict__ a, float4 *__restrict__ b) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 1000000;
    const long E = argc > 2 ? atol(argv[2]) : 10000000;
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
    auto run = [&](const char *name, auto launch, double by) {
        for (int i = 0; i < 5; ++i) launch();
        CK(hipEventRecord(a));
        const int reps = 30;
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%-28s %8.1f us   %6.2f TB/s\n", name, ms / reps * 1e3, by / (ms / reps * 1e-3) / 1e12);
    };
    for (int blocks : {2048, 4096, 8192}) {
        char nm[64];
        snprintf(nm, 64, "gather depth 8  grid %d", blocks);  run(nm, [&] { k_gather<8><<<blocks, 256>>>(N, d_rowptr, d_src, S, C, O); }, bytes);
        snprintf(nm, 64, "gather depth 16 grid %d", blocks);  run(nm, [&] { k_gather<16><<<blocks, 256>>>(N, d_rowptr, d_src, S, C, O); }, bytes);
    }
    run("stream copy 256 MB", [&] { k_stream<<<4096, 256>>>((size_t)N * 16, S, O); }, 2.0 * N * 256);
    CK(hipGetLastError());
    return 0;
}
