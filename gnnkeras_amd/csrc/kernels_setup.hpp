// Everything a small homogeneous graph needs before its first iteration, in ONE launch (GNN/Models/GNN.py:249-262: the
// ArcNode scatter-add of the arc labels, the neighbour-label aggregate, state_0's predicate) plus what this library adds in
// front of the loop (BatchNormalization folded into the first Dense, the per-node constant C of DESIGN.md section 3).
// The general path spends five launches on these (k_fold_bn, k_aggregate x2, k_segdense, k_converge); a merged MUTAG batch
// has ~1 k nodes and each of those launches is a few microseconds of host and device time around almost no work, so the
// whole-loop kernel (kernel_state_small.hpp) is preceded by this one launch instead.
//
//   workgroup t < n_tiles : 64 nodes.  a) BN affine of the constant rows and the folded bias in LDS, b) the constant
//                           inputs [labels | agg_nodes | agg_arcs] of its nodes in LDS (CSR walks in arc order, the same
//                           per-column fmaf chain as k_aggregate), c) C = inputs . Wf[const rows] + bf, d) its nodes'
//                           share of the state_0 predicate into pred0[t] (a plain store: no word to zero beforehand).
//   workgroup n_tiles     : folds both networks' first layers to global memory (the loop and output kernels read them)
//                           and zeroes the loop words, the barrier counters and k.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_general.hpp"

namespace gnn {

struct SetupCsr { const int *rowptr, *src; const float *w, *row_scale; };

struct SetupArgs {
    FoldJob net, out;                 // first layers of the state network and (W != nullptr) the output network
    int N, n_tiles;
    GroupTab groups;                  // n >= 1: tile t covers 64 nodes of ONE group (a merged graph without groups is one group)
    // constant inputs of the state network's first layer and the rows of W they meet
    const float *nodes; int ld_nodes; int L;                  // L = 0: the model has no label columns (state_dim == 0)
    const float *nodes_src; int ld_nodes_src; SetupCsr adj;   // agg_nodes = Adjacency^T . nodes_src
    const float *arc_labels; int ld_arcs; int A; SetupCsr arcnode;
    int row_nodes, row_aggn, row_agga;
    float *C; int ldC;
    // predicate of state_0 against ones (GNN.py:261)
    const float *state0; int ld_s0; int S; float thr; int *pred0;
    int *zero_a; int n_a; float *zero_b; int n_b;
};

__device__ __forceinline__ void setup_fold_global(const FoldJob &jb, float *part /* [256] */) {
    const int K = jb.K, H = jb.H, tid = threadIdx.x;
    for (int i = tid; i < K * H; i += 256) {
        const int k = i / H;
        const float inv = jb.gamma ? jb.gamma[k] / sqrtf(jb.var[k] + jb.eps) : 1.0f;
        jb.Wf[i] = jb.W[i] * inv;
    }
    // bf[h] = b[h] + sum_k shift[k] W[k][h]: 256 / H' row groups per column, partials meet in LDS in group order
    for (int h0 = 0; h0 < H; h0 += 256) {
        const int Hc = min(H - h0, 256);
        int G = 1;
        while (G * 2 * Hc <= 256) G *= 2;
        const int h = tid % Hc, grp = tid / Hc;
        float acc = 0.0f;
        if (jb.gamma && grp < G)
            for (int k = grp; k < K; k += G) {
                const float inv = jb.gamma[k] / sqrtf(jb.var[k] + jb.eps);
                acc = fmaf(jb.beta[k] - jb.mean[k] * inv, jb.W[(size_t)k * H + h0 + h], acc);
            }
        __syncthreads();
        part[tid] = acc;
        __syncthreads();
        if (tid < Hc) {
            float s = jb.b ? jb.b[h0 + tid] : 0.0f;
            for (int g2 = 0; g2 < G; ++g2) s += part[g2 * Hc + tid];
            jb.bf[h0 + tid] = s;
        }
    }
}

// dynamic LDS (floats): bfs[H] | Wc[Kc][H] | Xc[64][Kc + 1] | part[256]
__global__ void __launch_bounds__(256) k_setup_small(SetupArgs sa) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x;
    const int H = sa.net.H, K = sa.net.K, L = sa.L, A = sa.A;
    const int Kc = 2 * L + A, LDXC = Kc + 1;
    float *bfs = sm, *Wc = bfs + H, *Xc = Wc + Kc * H, *part = Xc + 64 * LDXC;

    if ((int)blockIdx.x == sa.n_tiles) {
        for (int i = tid; i < sa.n_a; i += 256) sa.zero_a[i] = 0;
        for (int i = tid; i < sa.n_b; i += 256) sa.zero_b[i] = 0.0f;
        setup_fold_global(sa.net, part);
        if (sa.out.W) setup_fold_global(sa.out, part);
        return;
    }
    const FoldJob &jb = sa.net;
    const GroupOfTile got = group_of_tile(sa.groups, blockIdx.x);
    const int jbase = got.node0 + ((int)blockIdx.x - got.tile0) * 64, jend = got.node_end;

    // ---- a. folded bias (every tile needs it; K x H multiply-adds) and the constant rows of Wf ------------------------
    {
        int G = 1;
        while (G * 2 * H <= 256) G *= 2;              // H <= 128 (checked by the launcher)
        const int h = tid % H, grp = tid / H;
        float acc = 0.0f;
        if (jb.gamma && grp < G)
            for (int k = grp; k < K; k += G) {
                const float inv = jb.gamma[k] / sqrtf(jb.var[k] + jb.eps);
                acc = fmaf(jb.beta[k] - jb.mean[k] * inv, jb.W[(size_t)k * H + h], acc);
            }
        part[tid] = acc;
        __syncthreads();
        if (tid < H) {
            float s = jb.b ? jb.b[tid] : 0.0f;
            for (int g2 = 0; g2 < G; ++g2) s += part[g2 * H + tid];
            bfs[tid] = s;
        }
        for (int i = tid; i < Kc * H; i += 256) {
            const int kc = i / H, hh = i % H;
            const int k = kc < L ? sa.row_nodes + kc : (kc < 2 * L ? sa.row_aggn + (kc - L) : sa.row_agga + (kc - 2 * L));
            const float inv = jb.gamma ? jb.gamma[k] / sqrtf(jb.var[k] + jb.eps) : 1.0f;
            Wc[i] = jb.W[(size_t)k * H + hh] * inv;
        }
    }

    // ---- b. constant inputs of this tile's nodes: 4 lanes per node, lane c owns columns c, c + 4, ... ----------------
    {
        const int m = tid >> 2, c = tid & 3;
        const int j = jbase + m;
        float *xr = Xc + m * LDXC;
        if (j < jend) {
            for (int f = c; f < L; f += 4) xr[f] = sa.nodes[(size_t)j * sa.ld_nodes + f];
            if (L > 0) {
                const int beg = sa.adj.rowptr[j], end = sa.adj.rowptr[j + 1];
                const float scale = sa.adj.row_scale ? sa.adj.row_scale[j] : 1.0f;
                for (int f = c; f < L; f += 4) {
                    float acc = 0.0f;
                    for (int e = beg; e < end; ++e) {
                        const float x = sa.nodes_src[(size_t)sa.adj.src[e] * sa.ld_nodes_src + f];
                        acc = sa.adj.w ? fmaf(sa.adj.w[e], x, acc) : acc + x;
                    }
                    xr[L + f] = acc * scale;
                }
            }
            if (A > 0) {
                const int beg = sa.arcnode.rowptr[j], end = sa.arcnode.rowptr[j + 1];
                const float scale = sa.arcnode.row_scale ? sa.arcnode.row_scale[j] : 1.0f;
                for (int f = c; f < A; f += 4) {
                    float acc = 0.0f;
                    for (int e = beg; e < end; ++e) {
                        const float x = sa.arc_labels[(size_t)sa.arcnode.src[e] * sa.ld_arcs + f];
                        acc = sa.arcnode.w ? fmaf(sa.arcnode.w[e], x, acc) : acc + x;
                    }
                    xr[2 * L + f] = acc * scale;
                }
            }
        } else {
            for (int f = c; f < Kc; f += 4) xr[f] = 0.0f;
        }
    }
    __syncthreads();

    // ---- c. C[j][h] = bf[h] + sum_kc X[j][kc] Wc[kc][h] ------------------------------------------------------------------
    for (int i = tid; i < 64 * H; i += 256) {
        const int m = i / H, h = i % H;
        const int j = jbase + m;
        float acc = bfs[h];
        const float *xr = Xc + m * LDXC;
        for (int kc = 0; kc < Kc; ++kc) acc = fmaf(xr[kc], Wc[kc * H + h], acc);
        if (j < jend) sa.C[(size_t)j * sa.ldC + h] = acc;
    }

    // ---- d. does any node of this tile still move between ones and state_0?  (16 lanes per node, as k_converge) -------
    int any = 0;
    {
        const int lane = tid & 15;
        for (int m = tid >> 4; m < 64; m += 16) {
            const int j = jbase + m;
            float d2 = 0.0f, n2 = 0.0f;
            if (j < jend)
                for (int f = lane; f < sa.S; f += 16) {
                    const float d = sa.state0[(size_t)j * sa.ld_s0 + f] - 1.0f;
                    d2 = fmaf(d, d, d2);
                    n2 = fmaf(1.0f, 1.0f, n2);
                }
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) {
                d2 += __shfl_xor(d2, off, 16);
                n2 += __shfl_xor(n2, off, 16);
            }
            if (j < jend && sqrtf(d2) > sa.thr * sqrtf(n2)) any = 1;
        }
    }
    any = __syncthreads_or(any);
    if (tid == 0) sa.pred0[blockIdx.x] = any;
}

inline size_t setup_small_lds(int H, int Kc) { return sizeof(float) * ((size_t)H + (size_t)Kc * H + 64 * (size_t)(Kc + 1) + 256); }

}  // namespace gnn
