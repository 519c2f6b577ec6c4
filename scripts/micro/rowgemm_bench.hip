// Micro-benchmark of the row-streaming dense kernels of gnnkeras_amd/csrc/kernels_train_big.hpp (k_train_fwd, k_train_bwd_dx) at
// C4 size: 1 M rows, S = 64.  Build one binary per ablation:  hipcc -O3 --offload-arch=gfx950 -DTB_ABL=<bits> rowgemm_bench.hip
// (1 no MFMAs, 2 no predicate loads, 4 no stores, 8 no statistics).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../../gnnkeras_amd/csrc/kernels_train_big.hpp"

__global__ void k_fill(float *p, size_t n, unsigned seed, float scale, float shift) {       // uniform(-1, 1) * scale + shift (real bits in the registers: all-zero operands draw less power)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        p[i] = ((float)(h >> 8) * (2.0f / 16777216.0f) - 1.0f) * scale + shift;
    }
}
__global__ void k_calib(unsigned long long *out) {      // ticks of s_memtime against the 100 MHz s_memrealtime
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float x = (float)threadIdx.x;
    for (int i = 0; i < 2000000; ++i) x = fmaf(x, 1.0000001f, 1e-9f);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long)x; }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 1000000;
    const int grid = argc > 2 ? atoi(argv[2]) : 512;
    const int S = 64, K = 159;
    float *state, *agg, *xc, *Wf, *bf, *Y, *part, *dx, *rs, *stats; int *flag;
    CK(hipMalloc(&state, (size_t)M * S * 4)); CK(hipMalloc(&agg, (size_t)M * S * 4)); CK(hipMalloc(&xc, (size_t)M * 32 * 4));
    CK(hipMalloc(&Y, (size_t)M * S * 4)); CK(hipMalloc(&Wf, K * S * 4)); CK(hipMalloc(&bf, S * 4)); CK(hipMalloc(&part, 2048 * 2 * S * 4));
    CK(hipMalloc(&dx, (size_t)M * 2 * S * 4)); CK(hipMalloc(&rs, (size_t)M * 4)); CK(hipMalloc(&stats, 8 * K * 4)); CK(hipMalloc(&flag, 64));
    CK(hipMemset(state, 0, (size_t)M * S * 4)); CK(hipMemset(agg, 0, (size_t)M * S * 4)); CK(hipMemset(xc, 0, (size_t)M * 32 * 4));
    CK(hipMemset(Wf, 0, K * S * 4)); CK(hipMemset(bf, 0, S * 4)); CK(hipMemset(rs, 0, (size_t)M * 4)); CK(hipMemset(stats, 0, 8 * K * 4)); CK(hipMemset(flag, 0, 64));
    if (!getenv("ROWGEMM_ZEROS")) {
        k_fill<<<2048, 256>>>(state, (size_t)M * S, 1u, 0.5f, 0.f); k_fill<<<2048, 256>>>(agg, (size_t)M * S, 2u, 0.5f, 0.f); k_fill<<<2048, 256>>>(xc, (size_t)M * 32, 3u, 1.f, 0.f);
        k_fill<<<64, 256>>>(Wf, (size_t)K * S, 4u, 0.1f, 0.f); k_fill<<<1, 64>>>(bf, S, 5u, 0.1f, 0.f); k_fill<<<64, 256>>>(rs, (size_t)M, 6u, 0.05f, 0.1f);
        k_fill<<<8, 256>>>(stats, (size_t)8 * K, 7u, 0.2f, 1.0f);
        CK(hipDeviceSynchronize());
    }
    gnn::TrainFwdArgs fa; memset(&fa, 0, sizeof(fa));
    fa.M = M; fa.state = state; fa.ld_state = S; fa.agg = agg; fa.ld_agg = S; fa.xc = xc; fa.Wf = Wf; fa.bf = bf; fa.H = S;
    fa.wrow_state = 0; fa.wrow_agg = S + 14; fa.cs.n = 3; fa.cs.width[0] = 14; fa.cs.wrow[0] = S; fa.cs.width[1] = 14; fa.cs.wrow[1] = 2 * S + 14;
    fa.cs.width[2] = 3; fa.cs.wrow[2] = 2 * S + 28; fa.act = 2; fa.Y = Y; fa.ldy = S; fa.thr = 0.f; fa.pred_flag = flag; fa.stat_part = part;
    gnn::TrainBwdArgs ba; memset(&ba, 0, sizeof(ba));
    ba.M = M; ba.dZ = Y; ba.ldz = S; ba.W = Wf; ba.ldw = S; ba.H = S; ba.S = S; ba.wrow_state = 0; ba.wrow_agg = S + 14;
    ba.state = state; ba.ld_state = S; ba.agg = agg; ba.ld_agg = S; ba.gamma = stats; ba.mean = stats + K; ba.var = stats + 2 * K; ba.m1 = stats + 3 * K;
    ba.m2 = stats + 4 * K; ba.eps = 1e-3f; ba.agg_row_scale = rs; ba.dx = dx; ba.ld_dx = 2 * S;
    { hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
      printf("device: %d CUs, LDS per CU %zu, per workgroup %zu, registers per CU %d\n", pr.multiProcessorCount, pr.maxSharedMemoryPerMultiProcessor, pr.sharedMemPerBlock, pr.regsPerMultiprocessor);
      CK(hipFuncSetAttribute((const void *)gnn::k_train_fwd_b6<4, GNN_ACT_SELU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gnn::train_fwd_b6_lds<4>()));
      CK(hipFuncSetAttribute((const void *)gnn::k_train_bwd_dx_b6<4, GNN_ACT_SELU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gnn::train_bwd_b6_lds<4>()));
      int nb = -1;
      for (size_t l : {(size_t)0, (size_t)16384, (size_t)32768, (size_t)49152, (size_t)60000, gnn::train_fwd_b6_lds<4>(), (size_t)80000}) {
          CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)gnn::k_train_fwd_b6<4, GNN_ACT_SELU>, 256, l)); printf("k_train_fwd_b6<4,selu>: dynamic LDS %zu -> %d workgroups per CU\n", l, nb); }
      CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)gnn::k_train_bwd_dx_b6<4, GNN_ACT_SELU>, 256, gnn::train_bwd_b6_lds<4>())); printf("k_train_bwd_dx_b6<4,selu>: LDS %zu -> %d per CU\n", gnn::train_bwd_b6_lds<4>(), nb);
      CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)gnn::k_train_wgrad32<2, GNN_ACT_SELU>, 256, 0)); printf("k_train_wgrad32<2,selu>: %d per CU\n", nb); }
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](const char *name, auto launch, double by) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(a));
        const int reps = 20;
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("ABL %d grid %4d  %-18s %8.1f us   %6.2f TB/s\n", TB_ABL, grid, name, ms / reps * 1e3, by / (ms / reps * 1e-3) / 1e12);
        CK(hipGetLastError());
    };
    run("k_train_fwd<4,4>", [&] { gnn::k_train_fwd<4, 4><<<grid, 64 * gnn::TB_WAVES, gnn::train_fwd_lds<4, 4>()>>>(fa); }, (double)M * (3.0 * S + 32) * 4);
    run("k_train_fwd_b6<4,selu>", [&] { gnn::k_train_fwd_b6<4, GNN_ACT_SELU><<<grid, 64 * gnn::TB_WAVES, gnn::train_fwd_b6_lds<4>()>>>(fa); }, (double)M * (3.0 * S + 32) * 4);
#ifdef TB_STAMPS
    { unsigned long long st[256]; CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(gnn::g_tb_stamps), sizeof(st)));
      unsigned long long *cal; CK(hipMalloc(&cal, 32)); k_calib<<<1, 64>>>(cal); unsigned long long hc[3]; CK(hipMemcpy(hc, cal, 24, hipMemcpyDeviceToHost));
      printf("s_memtime: %llu ticks in %llu ticks of the 100 MHz clock = %.1f MHz; block %d: set-up %llu, loop %llu, tail %llu ticks\n", hc[0], hc[1], 100.0 * hc[0] / hc[1], TB_STAMP_BLOCK, st[241] - st[240], st[242] - st[241], st[243] - st[242]);
      { std::vector<unsigned long long> bt(2 * 4096); CK(hipMemcpyFromSymbol(bt.data(), HIP_SYMBOL(gnn::g_tb_blocks), bt.size() * 8));
        unsigned long long t0 = ~0ull; for (int b = 0; b < grid; ++b) if (bt[2 * b] < t0) t0 = bt[2 * b];
        std::vector<double> st_(grid), en_(grid); for (int b = 0; b < grid; ++b) { st_[b] = (bt[2 * b] - t0) * 0.01; en_[b] = (bt[2 * b + 1] - t0) * 0.01; }
        std::vector<double> s2 = st_, e2 = en_; std::sort(s2.begin(), s2.end()); std::sort(e2.begin(), e2.end());
        printf("workgroup starts (us after the first): p10 %.1f p50 %.1f p90 %.1f max %.1f;  ends: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f\n",
               s2[grid / 10], s2[grid / 2], s2[grid * 9 / 10], s2[grid - 1], e2[0], e2[grid / 10], e2[grid / 2], e2[grid * 9 / 10], e2[grid - 1]); }
      { std::vector<unsigned long long> bt(2 * 4096); CK(hipMemcpyFromSymbol(bt.data(), HIP_SYMBOL(gnn::g_tb_blocks), bt.size() * 8));
        std::vector<unsigned> hw(2 * 4096); CK(hipMemcpyFromSymbol(hw.data(), HIP_SYMBOL(gnn::g_tb_hwid), hw.size() * 4));
        unsigned long long t0 = ~0ull; for (int b = 0; b < grid; ++b) if (bt[2 * b] < t0) t0 = bt[2 * b];
        std::vector<unsigned> early, late;
        for (int b = 0; b < grid; ++b) { const unsigned id = ((hw[2 * b + 1] & 0xF) << 16) | (hw[2 * b] & 0xFF00); ((bt[2 * b] - t0) < 2000 ? early : late).push_back(id); }
        auto distinct = [](std::vector<unsigned> v) { std::sort(v.begin(), v.end()); return (int)(std::unique(v.begin(), v.end()) - v.begin()); };
        printf("workgroups that start in the first 20 us: %zu on %d distinct (xcc, se, sh, cu); later: %zu on %d;  HW_ID / XCC_ID of blocks 0..7:", early.size(), distinct(early), late.size(), distinct(late));
        for (int b = 0; b < 8; ++b) printf(" %08x/%x", hw[2 * b], hw[2 * b + 1]);
        printf("\n"); }
      for (int i = 2; i < 12; ++i) printf("trip %2d: head->ready %6llu  mfma %6llu  epilogue %6llu  next head %6llu\n", i, st[4*i+1]-st[4*i], st[4*i+2]-st[4*i+1], st[4*i+3]-st[4*i+2], st[4*i+4]-st[4*i+3]); }
#endif
    run("k_train_bwd_dx<4,8>", [&] { gnn::k_train_bwd_dx<4, 8><<<grid, 64 * gnn::TB_WAVES, gnn::train_bwd_lds<4, 8>()>>>(ba); }, (double)M * (5.0 * S) * 4);
    { gnn::TrainWgradArgs wa; memset(&wa, 0, sizeof(wa));
      const int n_wg = grid; wa.M = M; wa.rows_per_wg = ((M + n_wg - 1) / n_wg + 15) / 16 * 16; const int wgrid = (M + wa.rows_per_wg - 1) / wa.rows_per_wg;
      float *wpart; CK(hipMalloc(&wpart, (size_t)std::max(wgrid, 512) * (K * S + S) * 4));
      wa.G = Y; wa.Y = state; wa.act = GNN_ACT_SELU; wa.state = state; wa.agg = agg; wa.xc = xc; wa.K = K; wa.wrow_state = 0; wa.wrow_agg = S + 14; wa.Kc = 31; wa.cs = fa.cs; wa.part = wpart;
      run("k_train_wgrad<4>", [&] { gnn::k_train_wgrad<4><<<wgrid, 256>>>(wa); }, (double)M * (4.0 * S + 32) * 4);
      run("k_train_wgrad32<2>", [&] { gnn::k_train_wgrad32<2, GNN_ACT_SELU><<<wgrid, 256>>>(wa); }, (double)M * (4.0 * S + 32) * 4);
      wa.Y = nullptr; wa.act = GNN_ACT_LINEAR;
      run("wgrad32<2,lin> dz", [&] { gnn::k_train_wgrad32<2, GNN_ACT_LINEAR><<<wgrid, 256>>>(wa); }, (double)M * (3.0 * S + 32) * 4);
      for (int g6 : {256, 512}) {
          wa.rows_per_wg = ((M + g6 - 1) / g6 + 63) / 64 * 64; const int grid6 = (M + wa.rows_per_wg - 1) / wa.rows_per_wg;
          char nm[64]; snprintf(nm, sizeof nm, "wgrad_b6<2,lin> g%d", grid6);
          CK(hipFuncSetAttribute((const void *)gnn::k_train_wgrad_b6<2, GNN_ACT_LINEAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gnn::train_wgrad_b6_lds<2, GNN_ACT_LINEAR>()));
          run(nm, [&] { gnn::k_train_wgrad_b6<2, GNN_ACT_LINEAR><<<grid6, 256, gnn::train_wgrad_b6_lds<2, GNN_ACT_LINEAR>()>>>(wa); }, (double)M * (3.0 * S + 32) * 4);
      } }
    ba.Y = state; ba.act = GNN_ACT_SELU;
    run("bwd_dx<4,8> with Y", [&] { gnn::k_train_bwd_dx<4, 8><<<grid, 64 * gnn::TB_WAVES, gnn::train_bwd_lds<4, 8>()>>>(ba); }, (double)M * (6.0 * S) * 4);
    run("bwd_dx_b6<4,selu>", [&] { gnn::k_train_bwd_dx_b6<4, GNN_ACT_SELU><<<grid, 256, gnn::train_bwd_b6_lds<4>()>>>(ba); }, (double)M * (6.0 * S) * 4);
    // the dZ form of the backward pass (train_loop.hpp, GNN_TRAIN_DZ): dZ arrives with the activation's derivative applied, the state half's
    // BatchNormalization term is left to k_aggregate_dz
    ba.Y = nullptr; ba.act = GNN_ACT_LINEAR; ba.defer_state_bn = 1;
    run("bwd_dx_b6<4,lin> dz", [&] { gnn::k_train_bwd_dx_b6<4, GNN_ACT_LINEAR><<<grid, 256, gnn::train_bwd_b6_lds<4>()>>>(ba); }, (double)M * (4.0 * S) * 4);
    {   // weight gradient and input gradient in one pass (the dZ form)
        gnn::TrainWgradArgs wa; memset(&wa, 0, sizeof(wa));
        const int g6 = 256; wa.M = M; wa.rows_per_wg = ((M + g6 - 1) / g6 + 63) / 64 * 64; const int grid6 = (M + wa.rows_per_wg - 1) / wa.rows_per_wg;
        float *wpart; CK(hipMalloc(&wpart, (size_t)grid6 * (K * S + S) * 4));
        wa.G = Y; wa.Y = nullptr; wa.act = GNN_ACT_LINEAR; wa.state = state; wa.agg = agg; wa.xc = xc; wa.K = K; wa.wrow_state = 0; wa.wrow_agg = S + 14; wa.Kc = 31; wa.cs = fa.cs; wa.part = wpart;
        CK(hipFuncSetAttribute((const void *)gnn::k_train_wgrad_dx_b6<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gnn::train_wgrad_dx_b6_lds<2>()));
        run("wgrad_dx_b6<2> fused", [&] { gnn::k_train_wgrad_dx_b6<2><<<grid6, 256, gnn::train_wgrad_dx_b6_lds<2>()>>>(wa, ba); }, (double)M * (5.0 * S + 32) * 4);
    }
    return 0;
}
