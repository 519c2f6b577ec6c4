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

// BN affine of one network's inputs into LDS: x_bn = x * inv + shift (ones / zeros without BatchNormalization); one pass of
// independent loads instead of re-deriving them inside the dependent sums below
__device__ __forceinline__ void setup_bn_affine(const FoldJob &jb, float *inv, float *shift) {
    for (int k = threadIdx.x; k < jb.K; k += 256) {
        float a = 1.0f, c = 0.0f;
        if (jb.gamma) { a = jb.gamma[k] / sqrtf(jb.var[k] + jb.eps); c = jb.beta[k] - jb.mean[k] * a; }
        inv[k] = a; shift[k] = c;
    }
    __syncthreads();
}

// bf[h] = b[h] + sum_k shift[k] W[k][h] for h in [h0, h0 + Hc): G row groups per column, partials meet in LDS in group order
__device__ __forceinline__ float setup_bias_column(const FoldJob &jb, const float *shift, float *part, int h0, int Hc) {
    const int tid = threadIdx.x, K = jb.K, H = jb.H;
    int G = 1;
    while (G * 2 * Hc <= 256) G *= 2;
    const int h = tid % Hc, grp = tid / Hc;
    float acc = 0.0f;
    if (jb.gamma && grp < G) {
        int k = grp;
        for (; k + 7 * G < K; k += 8 * G) {               // 8 independent loads in flight, summed in k order
            float w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = jb.W[(size_t)(k + u * G) * H + h0 + h];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fmaf(shift[k + u * G], w[u], acc);
        }
        for (; k < K; k += G) acc = fmaf(shift[k], jb.W[(size_t)k * H + h0 + h], acc);
    }
    __syncthreads();
    part[tid] = acc;
    __syncthreads();
    float s = 0.0f;
    if (tid < Hc) {
        s = jb.b ? jb.b[h0 + tid] : 0.0f;
        for (int g2 = 0; g2 < G; ++g2) s += part[g2 * Hc + tid];
    }
    return s;                                              // valid for tid < Hc
}

__device__ __forceinline__ void setup_fold_global(const FoldJob &jb, float *inv, float *shift, float *part /* [256] */) {
    const int K = jb.K, H = jb.H, tid = threadIdx.x;
    setup_bn_affine(jb, inv, shift);
    for (int i = tid; i < K * H; i += 256) jb.Wf[i] = jb.W[i] * inv[i / H];
    for (int h0 = 0; h0 < H; h0 += 256) {
        const int Hc = min(H - h0, 256);
        const float s = setup_bias_column(jb, shift, part, h0, Hc);
        if (tid < Hc) jb.bf[h0 + tid] = s;
    }
    __syncthreads();
}

// out[f] (f = c, c + 4, ... < F) = row_scale * sum_e w_e X[src_e][f] over the arcs [beg, end) of one destination, in arc order
// (the per-column fmaf chain of k_aggregate).  The first 8 source ids / weights are fetched at once and each column's 8 rows
// are independent loads: the chain  id -> row -> add  is paid once per column, not once per arc.
__device__ __forceinline__ void setup_walk(const SetupCsr &csr, int j, const float *__restrict__ X, int ldx, int F, int c, float *out) {
    const int beg = csr.rowptr[j], end = csr.rowptr[j + 1];
    const float scale = csr.row_scale ? csr.row_scale[j] : 1.0f;
    int ids[8]; float ws[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool on = beg + i < end;
        ids[i] = on ? csr.src[beg + i] : 0;
        ws[i] = (on && csr.w) ? csr.w[beg + i] : 1.0f;
    }
    for (int f = c; f < F; f += 4) {
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = beg + i < end ? X[(size_t)ids[i] * ldx + f] : 0.0f;
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (beg + i < end) acc = csr.w ? fmaf(ws[i], x[i], acc) : acc + x[i];
        for (int e = beg + 8; e < end; ++e) {
            const float xv = X[(size_t)csr.src[e] * ldx + f];
            acc = csr.w ? fmaf(csr.w[e], xv, acc) : acc + xv;
        }
        out[f] = acc * scale;
    }
}

// dynamic LDS (floats): bfs[H] | Wc[Kc][H] | Xc[64][Kc + 1] | part[256] | inv[Kmax] | shift[Kmax]   (Kmax = widest first layer)
__global__ void __launch_bounds__(256) k_setup_small(SetupArgs sa) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x;
    const int H = sa.net.H, K = sa.net.K, L = sa.L, A = sa.A;
    const int Kc = 2 * L + A, LDXC = Kc + 1;
    float *bfs = sm, *Wc = bfs + H, *Xc = Wc + Kc * H, *part = Xc + 64 * LDXC;
    float *inv = part + 256, *shift = inv + max(K, sa.out.W ? sa.out.K : 0);

    if ((int)blockIdx.x == sa.n_tiles) {
        for (int i = tid; i < sa.n_a; i += 256) sa.zero_a[i] = 0;
        for (int i = tid; i < sa.n_b; i += 256) sa.zero_b[i] = 0.0f;
        setup_fold_global(sa.net, inv, shift, part);
        if (sa.out.W) setup_fold_global(sa.out, inv, shift, part);
        return;
    }
    const FoldJob &jb = sa.net;
    const GroupOfTile got = group_of_tile(sa.groups, blockIdx.x);
    const int jbase = got.node0 + ((int)blockIdx.x - got.tile0) * 64, jend = got.node_end;

    // ---- a. folded bias (every tile needs it; K x H multiply-adds) and the constant rows of Wf ------------------------
    setup_bn_affine(jb, inv, shift);
    {
        const float s = setup_bias_column(jb, shift, part, 0, H);          // H <= 128 (checked by the launcher)
        if (tid < H) bfs[tid] = s;
        for (int i = tid; i < Kc * H; i += 256) {
            const int kc = i / H, hh = i % H;
            const int k = kc < L ? sa.row_nodes + kc : (kc < 2 * L ? sa.row_aggn + (kc - L) : sa.row_agga + (kc - 2 * L));
            Wc[i] = jb.W[(size_t)k * H + hh] * inv[k];
        }
    }

    // ---- b. constant inputs of this tile's nodes: 4 lanes per node, lane c owns columns c, c + 4, ... ----------------
    {
        const int m = tid >> 2, c = tid & 3;
        const int j = jbase + m;
        float *xr = Xc + m * LDXC;
        if (j < jend) {
            for (int f = c; f < L; f += 4) xr[f] = sa.nodes[(size_t)j * sa.ld_nodes + f];
            if (L > 0) setup_walk(sa.adj, j, sa.nodes_src, sa.ld_nodes_src, L, c, xr + L);
            if (A > 0) setup_walk(sa.arcnode, j, sa.arc_labels, sa.ld_arcs, A, c, xr + 2 * L);
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

inline size_t setup_small_lds(int H, int Kc, int Kmax) { return sizeof(float) * ((size_t)H + (size_t)Kc * H + 64 * (size_t)(Kc + 1) + 256 + 2 * (size_t)Kmax); }

}  // namespace gnn
