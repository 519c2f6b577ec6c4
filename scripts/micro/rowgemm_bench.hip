// Micro-benchmark of the row-streaming dense kernels of gnnkeras_amd/csrc/kernels_train_big.hpp (k_train_fwd, k_train_bwd_dx) at
// C4 size: 1 M rows, S = 64.  Build one binary per ablation:  hipcc -O3 --offload-arch=gfx950 -DTB_ABL=<bits> rowgemm_bench.hip
// (1 no MFMAs, 2 no predicate loads, 4 no stores, 8 no statistics).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../gnnkeras_amd/csrc/kernels_train_big.hpp"

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
    gnn::TrainFwdArgs fa; memset(&fa, 0, sizeof(fa));
    fa.M = M; fa.state = state; fa.ld_state = S; fa.agg = agg; fa.ld_agg = S; fa.xc = xc; fa.Wf = Wf; fa.bf = bf; fa.H = S;
    fa.wrow_state = 0; fa.wrow_agg = S + 14; fa.cs.n = 3; fa.cs.width[0] = 14; fa.cs.wrow[0] = S; fa.cs.width[1] = 14; fa.cs.wrow[1] = 2 * S + 14;
    fa.cs.width[2] = 3; fa.cs.wrow[2] = 2 * S + 28; fa.act = 2; fa.Y = Y; fa.ldy = S; fa.thr = 0.f; fa.pred_flag = flag; fa.stat_part = part;
    gnn::TrainBwdArgs ba; memset(&ba, 0, sizeof(ba));
    ba.M = M; ba.dZ = Y; ba.ldz = S; ba.W = Wf; ba.ldw = S; ba.H = S; ba.S = S; ba.wrow_state = 0; ba.wrow_agg = S + 14;
    ba.state = state; ba.ld_state = S; ba.agg = agg; ba.ld_agg = S; ba.gamma = stats; ba.mean = stats + K; ba.var = stats + 2 * K; ba.m1 = stats + 3 * K;
    ba.m2 = stats + 4 * K; ba.eps = 1e-3f; ba.agg_row_scale = rs; ba.dx = dx; ba.ld_dx = 2 * S;
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
      for (int i = 2; i < 12; ++i) printf("trip %2d: head->ready %6llu  mfma %6llu  epilogue %6llu  next head %6llu\n", i, st[4*i+1]-st[4*i], st[4*i+2]-st[4*i+1], st[4*i+3]-st[4*i+2], st[4*i+4]-st[4*i+3]); }
#endif
    run("k_train_bwd_dx<4,8>", [&] { gnn::k_train_bwd_dx<4, 8><<<grid, 64 * gnn::TB_WAVES, gnn::train_bwd_lds<4, 8>()>>>(ba); }, (double)M * (5.0 * S) * 4);
    { gnn::TrainWgradArgs wa; memset(&wa, 0, sizeof(wa));
      const int n_wg = grid; wa.M = M; wa.rows_per_wg = ((M + n_wg - 1) / n_wg + 15) / 16 * 16; const int wgrid = (M + wa.rows_per_wg - 1) / wa.rows_per_wg;
      float *wpart; CK(hipMalloc(&wpart, (size_t)wgrid * (K * S + S) * 4));
      wa.G = Y; wa.Y = state; wa.act = GNN_ACT_SELU; wa.state = state; wa.agg = agg; wa.xc = xc; wa.K = K; wa.wrow_state = 0; wa.wrow_agg = S + 14; wa.Kc = 31; wa.cs = fa.cs; wa.part = wpart;
      run("k_train_wgrad<4>", [&] { gnn::k_train_wgrad<4><<<wgrid, 256>>>(wa); }, (double)M * (4.0 * S + 32) * 4);
      run("k_train_wgrad32<2>", [&] { gnn::k_train_wgrad32<2, GNN_ACT_SELU><<<wgrid, 256>>>(wa); }, (double)M * (4.0 * S + 32) * 4); }
    ba.Y = state; ba.act = GNN_ACT_SELU;
    run("bwd_dx<4,8> with Y", [&] { gnn::k_train_bwd_dx<4, 8><<<grid, 64 * gnn::TB_WAVES, gnn::train_bwd_lds<4, 8>()>>>(ba); }, (double)M * (6.0 * S) * 4);
    run("bwd_dx_b6<4,selu>", [&] { gnn::k_train_bwd_dx_b6<4, GNN_ACT_SELU><<<grid, 256, gnn::train_bwd_b6_lds<4>()>>>(ba); }, (double)M * (6.0 * S) * 4);
    return 0;
}
