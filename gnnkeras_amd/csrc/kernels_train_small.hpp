// Training on SMALL graphs (a merged MUTAG batch: ~1 k nodes): the K training-mode iterations of the forward pass - and the k
// iterations of back-propagation through them - as ONE persistent launch each (reference GNN/Models/GNN.py:277-306: `self(x,
// training=True)` and `tape.gradient` through the unrolled loop).
//
// Round 2 ran such a step as ~450 dependent launches of ~8 us (4.3 ms at d = 32 x 50 iterations): nine per iteration pair, each
// keeping 15 of the 256 CUs busy for a few microseconds.  Here one workgroup owns one tile of <= 64 nodes for the whole loop, and what
// an iteration needs from the other tiles crosses a grid barrier (one 64-bit arrival counter per parity, agent-scope add + poll):
//
//   forward iteration t   [own | neighbour sum] of the tile -> LDS, neighbour sums -> the tape; BatchNormalization in training mode
//                         needs the statistics of ALL nodes: the tile's column sums / squares of [state | agg] go to a partial slot,
//                         BARRIER, every workgroup adds the partials in workgroup order (the same bits everywhere); the normalisation
//                         a x + c is applied as the rows are read from LDS, [state | agg] . W on the matrix cores (operands swapped:
//                         the result is row-major), activation, predicate, rows -> states[t + 1].
//   backward iteration t  dZ = G (.) act'(states[t + 1]); P_wg = X^T dZ of the tile on the matrix cores - kept LOCAL: the weight
//                         gradient a (.) P + c q^T is linear in P, so every workgroup accumulates its own share in registers over
//                         all iterations and the shares are summed once at the end; what the BatchNorm input gradient needs of the
//                         other tiles is only q = colsum(dZ) and S2_k = sum_h W[k, h] P[k, h] - 127 floats per workgroup: partial
//                         slot, BARRIER, summed in order; dx = BN-gradient(dZ . W^T) on the matrix cores, the agg half scaled once per
//                         row; G = dx_state + Adj . dx_agg gathered by source.
// Two forms (template parameter LOCAL):
//   general   tile b = nodes [64 b, 64 b + 64); a neighbour may live in another tile: rows are exchanged through memory (`sc1`) behind
//             a second barrier per iteration (which also carries "some node still moves").
//   LOCAL     the batch is block-diagonal (a merge of small graphs) and the caller hands over tiles cut at graph boundaries (TileTab):
//             no arc leaves a tile, the state of a tile never leaves LDS, and the only thing tiles share is the BatchNorm statistics and
//             the `reduce_any` of the loop condition - ONE barrier per forward iteration (the flag of iteration t - 1 travels with the
//             statistics of iteration t), one per backward iteration with BatchNormalization, none without.
// Every workgroup must be resident (grid <= CUs, one workgroup per CU); polls sleep and are bounded: an expired wait poisons k / the
// gradients.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_general.hpp"
#include "kernel_state_fused4.hpp"      // activate4
#include "kernel_state_lds.hpp"         // row16_sum_to_lane15
#include "kernels_train.hpp"            // activate_grad_from_output
#include "buffer_ops.hpp"

namespace gnn {

#ifdef GNN_TS_TIMELINE
// phase times of workgroup 0 (wall_clock64 ticks of 10 ns, summed over the iterations): [0] forward, [1] backward; scripts/ts_timeline.py
__device__ unsigned long long g_ts_phase[2][16];
#define TS_CLOCK() long long ts_last = wall_clock64(); unsigned long long ts_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define TS_STAMP(i) do { const long long now = wall_clock64(); ts_acc[i] += (unsigned long long)(now - ts_last); ts_last = now; } while (0)
#define TS_WRITE(w) do { if (blockIdx.x == 0 && threadIdx.x == 0) for (int i_ = 0; i_ < 10; ++i_) g_ts_phase[w][i_] = ts_acc[i_]; } while (0)
#else
#define TS_CLOCK() do { } while (0)
#define TS_STAMP(i) do { } while (0)
#define TS_WRITE(w) do { } while (0)
#endif

constexpr int TS_NT = 256;               // threads per workgroup: 4 waves, wave w owns rows 16 w .. 16 w + 15 of the 64-node tile
constexpr size_t TS_LDS = 96 * 1024;     // > half of a CU's LDS: one workgroup per CU

// the grid barrier of kernel_state_small.hpp as a function: every thread calls it; `any` (workgroup-uniform or not) is OR-ed over the
// workgroup and travels in the high word.  Returns whether ANY workgroup reported any != 0 at this barrier.
struct GridBar {
    unsigned long long *bar;      // two counters (parity of the barrier index), zero before the launch
    unsigned n_wg;
    int bi;                       // barriers passed so far
    unsigned moved_seen0, moved_seen1;
    int timed_out;
    unsigned long long wait_ticks;   // bound of a barrier wait (buffer_ops.hpp: wait_until)
};
__device__ __forceinline__ bool grid_barrier(GridBar &gb, int any, int *cont_lds) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's sc1 stores have left
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) {
        unsigned long long *ctr = gb.bar + (gb.bi & 1);
        __hip_atomic_fetch_add(ctr, 1ull + ((unsigned long long)(any ? 1u : 0u) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (unsigned)(gb.bi / 2 + 1) * gb.n_wg;
        unsigned long long v = 0;
        // (once a wait has expired the launch's results are void: the remaining barriers do not wait out the bound again)
        if (gb.timed_out || !wait_until(gb.wait_ticks, [&]() { v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return (unsigned)v >= target; }))
            gb.timed_out = 1;
        const unsigned moved = (unsigned)(v >> 32);
        const bool odd = (gb.bi & 1) != 0;
        *cont_lds = (moved != (odd ? gb.moved_seen1 : gb.moved_seen0)) ? 1 : 0;
        if (odd) gb.moved_seen1 = moved; else gb.moved_seen0 = moved;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();
    gb.bi += 1;
    return *cont_lds != 0;
}

// ---- constant part of the first layer, once per step ------------------------------------------------------------------------------------
//   Cc[n, h] = b[h] + sum over the constant input columns k of (a_k (x[n, k] - mean_k) + beta_k) W[k, h],   a = gamma rstd
// (a = 1, mean = beta = 0 without BatchNormalization; the constants' batch statistics do not change between iterations).  The mean leaves
// the value FIRST: a x + (beta - mean a) cancels mean a against a x afterwards and loses digits in proportion to mean / sigma (raw labels).
struct ConstSegs { const float *ptr[3]; int ld[3], width[3], wrow[3]; int n; };
__global__ void __launch_bounds__(256)
k_train_small_const(int N, int S, int Sw, ConstSegs cs, const float *__restrict__ W, const float *__restrict__ b, const float *gamma, const float *beta,
                    const float *mean, const float *var, float eps, float *__restrict__ Cc, const int *__restrict__ rows = nullptr) {
    // S = the kernels' padded state width (16 / 32 / 64), Sw <= S the network's: W is [in_dim][Sw]; pad columns of Cc are zero.
    // `rows` (heterogeneous models: the node ids of one type, N of them): row m of Cc belongs to node rows[m] of the constant segments
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * S) return;
    const int h = i % S, n = rows ? rows[i / S] : i / S;
    if (h >= Sw) { Cc[i] = 0.0f; return; }
    float acc = b[h];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        if (s >= cs.n) break;
        for (int j = 0; j < cs.width[s]; ++j) {
            const int k = cs.wrow[s] + j;
            float x = cs.ptr[s][(size_t)n * cs.ld[s] + j];
            if (gamma) { const float a = gamma[k] / sqrtf(var[k] + eps); x = fmaf(x - mean[k], a, beta[k]); }
            acc = fmaf(x, W[(size_t)k * Sw + h], acc);
        }
    }
    Cc[i] = acc;
}

// ---- tiles ------------------------------------------------------------------------------------------------------------------------------
struct TileTab { int n; int begin[257]; };      // n == 0: tile b = nodes [64 b, 64 b + 64); else tile b = nodes [begin[b], begin[b + 1]), <= 64 of them
                                                // (LOCAL kernels: cut at graph boundaries, no arc leaves a tile; heterogeneous models: cut at TYPE boundaries)

// ---- heterogeneous models (reference CompositeGNN.py:223-232: one state network per node type, applied to the rows of its type) -----------
// The persistent kernels walk the nodes in TYPE ORDER (position i = node perm[i] of the caller's graph; the caller's type lists ARE that
// permutation) in tiles that never straddle two types: a workgroup then needs ONE network - its weights, its BatchNormalization
// parameters and statistics (taken over the tiles of ITS type only), its share of ITS gradients - and the inner loops are those of the
// homogeneous kernels.  What crosses types is what crosses tiles anyway: neighbour rows (through memory, behind the grid barrier) and
// the loop condition.  n == 0: a homogeneous model (the scalar fields of the kernel arguments apply).
struct TypeTab {
    int n;
    int wg_begin[GNN_MAX_TYPES + 1];             // the tiles (= workgroups) of type t: [wg_begin[t], wg_begin[t + 1])
    int count[GNN_MAX_TYPES];                    // nodes of the type
    const float *W[GNN_MAX_TYPES], *gamma[GNN_MAX_TYPES], *beta[GNN_MAX_TYPES];
    float *stats[GNN_MAX_TYPES];                 // [K][2 in_s[t]]
    int in_s[GNN_MAX_TYPES], off_state[GNN_MAX_TYPES], off_agg[GNN_MAX_TYPES], act[GNN_MAX_TYPES];
    float *partW[GNN_MAX_TYPES], *partBN[GNN_MAX_TYPES];     // backward: the type's [tiles][in_s S + S] / [tiles][2 in_s] shares
    const int *perm, *inv;
};
struct TypeConsts { ConstSegs cs[GNN_MAX_TYPES]; };          // backward: the constant input segments of every type's network

// The CSR rows of the tile's nodes as the gather walks them: thread (q, l4) = lane l4 of the LPR lanes that fetch 16-byte pieces of
// row q of a pass; the first 16 source ids / weights of every row stay in registers for all iterations.
template <int SQ, bool HAS_W, bool LOCAL>
struct TileCsr {
    static constexpr int S = 16 * SQ, LPR = S / 4, NPP = TS_NT / LPR, NPASS = 64 / NPP, IPL = 16 / LPR, PP = NPASS < 2 ? NPASS : 2;
    int node[NPASS], beg[NPASS], end[NPASS], ids[NPASS][IPL];
    float wts[NPASS][IPL], scl[NPASS];
    const int *src; const float *w; const int *inv;
    int n0, nt, q, l4, bad;

    // `node` = the row's POSITION in the tape (what the kernels address); heterogeneous models walk the nodes in type order: `perm[position]`
    // = the node's id in the caller's operators (row pointers, scales), `inv[id]` = its position (source ids)
    __device__ __forceinline__ int local_id(int s) {        // LOCAL: position inside the tile; an arc that leaves the tile is an error
        if (!LOCAL) return inv ? inv[s] : s;
        const int r = s - n0;
        if (r < 0 || r >= nt) { bad = 1; return 0; }
        return r;
    }
    __device__ __forceinline__ void load(int n0_, int nt_, const int *rowptr, const int *src_, const float *w_, const float *row_scale,
                                         const int *perm = nullptr, const int *inv_ = nullptr) {
        n0 = n0_; nt = nt_; src = src_; w = w_; bad = 0; inv = inv_;
        q = threadIdx.x / LPR; l4 = threadIdx.x % LPR;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int r = p * NPP + q;
            node[p] = r < nt ? n0 + r : -1;
            beg[p] = end[p] = 0; scl[p] = 1.0f;
            if (node[p] >= 0) { const int o = perm ? perm[node[p]] : node[p]; beg[p] = rowptr[o]; end[p] = rowptr[o + 1]; if (row_scale) scl[p] = row_scale[o]; }
#pragma unroll
            for (int u = 0; u < IPL; ++u) {
                const int e = beg[p] + u * LPR + l4;
                ids[p][u] = e < end[p] ? local_id(src[e]) : 0;
                wts[p][u] = (HAS_W && e < end[p]) ? w[e] : 0.0f;
            }
        }
    }
    // acc[p] = scale_p * sum over the arcs of row p of w_e * X[src_e]   (this thread's 16-byte piece).  FROM_LDS: X = rows of the tile
    // in LDS (row stride ld floats); else rows of a matrix in memory read `sc1` (written by other workgroups during this launch).
    template <bool FROM_LDS>
    __device__ __forceinline__ void gather(f32x4 (&acc)[NPASS], const float *lds, int ld, __amdgpu_buffer_rsrc_t rs) {
#pragma unroll
        for (int p0 = 0; p0 < NPASS; p0 += PP) {
            f32x4 v[PP][16];
#pragma unroll
            for (int pp = 0; pp < PP; ++pp) {
                const int p = p0 + pp, rem = end[p] - beg[p];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int id = __shfl(ids[p][i / LPR], i % LPR, LPR);
                    if (FROM_LDS) {
                        v[pp][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (__any(i < rem)) { const f32x4 t = *reinterpret_cast<const f32x4 *>(lds + id * ld + 4 * l4); if (i < rem) v[pp][i] = t; }
                    } else v[pp][i] = buf_ld_sc1(rs, i < rem ? (unsigned)id * (unsigned)(S * 4) + 16u * l4 : BUF_OFF);
                }
            }
#pragma unroll
            for (int pp = 0; pp < PP; ++pp) {
                const int p = p0 + pp;
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (HAS_W) t += __shfl(wts[p][i / LPR], i % LPR, LPR) * v[pp][i];
                    else t += v[pp][i];
                }
                acc[p] = t;
            }
        }
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            int rem = end[p] - beg[p] - 16, eb = beg[p] + 16;
#pragma unroll 1
            while (__any(rem > 0)) {                         // rows with more than 16 arcs: 16 more per trip
                int idc[IPL]; float wsc[IPL];
#pragma unroll
                for (int u = 0; u < IPL; ++u) {
                    const int e = eb + u * LPR + l4;
                    idc[u] = e < end[p] ? local_id(src[e]) : 0;
                    wsc[u] = (HAS_W && e < end[p]) ? w[e] : 0.0f;
                }
                f32x4 v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int id = __shfl(idc[i / LPR], i % LPR, LPR);
                    if (FROM_LDS) { const f32x4 t = *reinterpret_cast<const f32x4 *>(lds + id * ld + 4 * l4); v[i] = i < rem ? t : (f32x4){0.f, 0.f, 0.f, 0.f}; }
                    else v[i] = buf_ld_sc1(rs, i < rem ? (unsigned)id * (unsigned)(S * 4) + 16u * l4 : BUF_OFF);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (HAS_W) acc[p] += __shfl(wsc[i / LPR], i % LPR, LPR) * v[i];
                    else acc[p] += v[i];
                }
                rem -= 16; eb += 16;
            }
            acc[p] *= scl[p];
        }
    }
};

// partial slots of the other workgroups: `n` floats per workgroup, summed in workgroup order with 16 loads in flight (double accumulator)
__device__ __forceinline__ double sum_partials(const float *part, unsigned wg_lo, unsigned n_wg, int n, int i) {      // the workgroups [wg_lo, n_wg)
    const __amdgpu_buffer_rsrc_t r_p = buf_rsrc(part);
    double t = 0.0;
    for (unsigned wg = wg_lo; wg < n_wg; wg += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u)          // (absent slots read 0)
            v[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_p, wg + u < n_wg ? (int)(((wg + u) * n + i) * 4) : (int)BUF_OFF, 0, 16));
#pragma unroll
        for (int u = 0; u < 16; ++u) t += (double)v[u];
    }
    return t;
}

// Column statistics of ALL nodes from the tiles' shares, in workgroup order and double precision (the same bits in every workgroup).  A
// tile's share is taken in float32 around a pivot of ITS OWN (the tile's first row: nothing of the mean's size in the sums, whatever the
// data - raw labels 30 sigma from zero, a state whose distribution jumps between iterations): slot = (s1 = sum (x - pv) | s2 = sum (x - pv)^2
// | pv), `w` columns each.  The shares are moved to the origin and added in DOUBLE - sum x = s1 + n pv, sum x^2 = s2 + 2 pv s1 + n pv^2 (every
// product exact) - where E[x^2] - mean^2 loses (mean / sigma)^2 of 2^-53, not of 2^-24.  `n_of(j)` = rows of tile j.
template <typename NOf>
__device__ __forceinline__ void merge_tile_stats(const float *part, unsigned wg_lo, unsigned n_wg, int w, int col, NOf n_of, double inv_n, float &mean_out,
                                                 float &var_out) {      // the tiles [wg_lo, n_wg)
    const __amdgpu_buffer_rsrc_t r_p = buf_rsrc(part);
    double S1 = 0.0, S2 = 0.0;
    for (unsigned wg = wg_lo; wg < n_wg; wg += 8) {
        float s1[8], s2[8], pv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {          // (absent slots read 0)
            const bool in = wg + u < n_wg;
            const int base = (int)(((wg + u) * 3 * w + col) * 4);
            s1[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_p, in ? base : (int)BUF_OFF, 0, 16));
            s2[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_p, in ? base + w * 4 : (int)BUF_OFF, 0, 16));
            pv[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_p, in ? base + 2 * w * 4 : (int)BUF_OFF, 0, 16));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double nj = wg + u < n_wg ? (double)n_of((int)(wg + u)) : 0.0, p_ = (double)pv[u], a1 = (double)s1[u];
            S1 += a1 + nj * p_;
            S2 += (double)s2[u] + p_ * (2.0 * a1 + nj * p_);
        }
    }
    const double mean = S1 * inv_n;
    mean_out = (float)mean;
    var_out = (float)fmax(S2 * inv_n - mean * mean, 0.0);
}

// ---- forward ---------------------------------------------------------------------------------------------------------------------------------
struct TrainSmallFwd {
    int N, S, K;
    int Sw;                      // the state's real width (<= the template's S = 16 SQ): rows of the tape are S floats, pad columns zero
    const int *rowptr, *src; const float *w, *row_scale;       // adjacency by destination
    float *states;               // [K + 1][N][S]; [0] = state_0 (the caller's)
    float *agg;                  // [K][N][S] neighbour sums, kept for the backward pass
    float *stats;                // [K][2 in_s] (mean | var): constant columns pre-filled, the state / agg columns written here
    int in_s, off_agg;
    const float *W;              // first-layer kernel [in_s][S]
    const float *gamma, *beta; float eps;                       // BatchNormalization or NULL
    int act;
    const float *Cc;             // [N][S] constant part (k_train_small_const)
    float thr;
    const int *flag0;            // predicate of state_0 (one word)
    unsigned long long *bar;     // two arrival counters, zero
    float *part;                 // [2][n_wg][6 S] statistics shares (s1 | s2 | pivot of the tile, 2 S columns each; parity of the iteration)
    float *k_out;                // [0] = iterations executed (-1e9: a barrier timed out), [1] = 1 when an arc leaves its tile (LOCAL)
    unsigned long long wait_ticks;   // bound of a barrier wait (buffer_ops.hpp: wait_until)
};

template <int SQ, bool HAS_W, bool LOCAL>
__global__ void __launch_bounds__(TS_NT, 1) k_train_small_fwd(TrainSmallFwd a, TileTab tt, TypeTab yt) {
    using Csr = TileCsr<SQ, HAS_W, LOCAL>;
    constexpr int S = 16 * SQ, NPP = Csr::NPP, NPASS = Csr::NPASS;
    constexpr int LDX = 2 * S + 4;        // row stride of the [own | agg] tile: == 4 (mod 32) dwords, 16-B chunks conflict-free
    constexpr int LDW = S + 4;            // weight rows: the A-operand read takes rows 4 g + e of a 16-row block, == 4 (mod 32) spreads them
    extern __shared__ __attribute__((aligned(16))) float ts_smem[];
    float *Xs = ts_smem;                  // [64][LDX]
    float *W0 = Xs + 64 * LDX;            // [2 S][LDW] kernel rows of the state / agg columns
    float *st_a = W0 + 2 * S * LDW;       // [2 S] a_k, [2 S] beta_k, [2 S] column means of this iteration, reduction scratch [512]
    float *st_c = st_a + 2 * S, *piv = st_c + 2 * S, *red = piv + 2 * S;
    __shared__ int cont;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const bool tab = LOCAL || tt.n > 0;
    const int n0 = tab ? tt.begin[blockIdx.x] : (int)blockIdx.x * 64;
    const int nt = tab ? tt.begin[blockIdx.x + 1] - n0 : min(64, a.N - n0);
    // heterogeneous models: this tile's node type selects the network (TypeTab); the statistics span the tiles [wg_lo, wg_hi) of the type
    int ty = 0;
    if (yt.n > 0) { while (ty + 1 < yt.n && (int)blockIdx.x >= yt.wg_begin[ty + 1]) ++ty; }
    const float *Wk = yt.n > 0 ? yt.W[ty] : a.W, *gamma_ = yt.n > 0 ? yt.gamma[ty] : a.gamma, *beta_ = yt.n > 0 ? yt.beta[ty] : a.beta;
    float *stats_ = yt.n > 0 ? yt.stats[ty] : a.stats;
    const int in_s = yt.n > 0 ? yt.in_s[ty] : a.in_s, off_state = yt.n > 0 ? yt.off_state[ty] : 0, off_agg = yt.n > 0 ? yt.off_agg[ty] : a.off_agg;
    const int act = yt.n > 0 ? yt.act[ty] : a.act;
    const unsigned wg_lo = yt.n > 0 ? (unsigned)yt.wg_begin[ty] : 0u, wg_hi = yt.n > 0 ? (unsigned)yt.wg_begin[ty + 1] : gridDim.x;
    const double inv_rows = 1.0 / (double)(yt.n > 0 ? yt.count[ty] : a.N);
    const bool bn = gamma_ != nullptr;
    const size_t NS = (size_t)a.N * S;
    GridBar gb{a.bar, gridDim.x, 0, 0u, 0u, 0, a.wait_ticks};

    for (int i = tid; i < 2 * S * S; i += TS_NT) {
        const int k = i / S, h = i % S;
        const int kk = k < S ? k : k - S;
        W0[k * LDW + h] = (kk < a.Sw && h < a.Sw) ? Wk[(size_t)((k < S ? off_state : off_agg) + kk) * a.Sw + h] : 0.0f;
    }
    if (tid < 2 * S) { piv[tid] = 0.0f; st_a[tid] = 1.0f; st_c[tid] = 0.0f; }
    Csr csr;
    csr.load(n0, nt, a.rowptr, a.src, a.w, a.row_scale, yt.perm, yt.inv);
    const int q = csr.q, l4 = csr.l4;
    // the constant part of this lane's output chunks (row 16 wave + c, columns 16 ct + 4 g ..)
    const bool oin = 16 * wave + c < nt;
    const int orow = n0 + 16 * wave + c;
    f32x4 cc[SQ];
#pragma unroll
    for (int ct = 0; ct < SQ; ++ct)
        cc[ct] = oin ? *reinterpret_cast<const f32x4 *>(a.Cc + (size_t)orow * S + 16 * ct + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
    if (LOCAL) {                          // the tile's rows of state_0: from here on the state lives in the own half of Xs
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            f32x4 own = {0.f, 0.f, 0.f, 0.f};
            if (csr.node[p] >= 0) own = *reinterpret_cast<const f32x4 *>(a.states + (size_t)csr.node[p] * S + 4 * l4);
            *reinterpret_cast<f32x4 *>(Xs + (p * NPP + q) * LDX + 4 * l4) = own;
        }
    }
    const bool run = __hip_atomic_load(a.flag0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    __syncthreads();

    // the kernel rows as this lane's A operands (k-step (qq, e) of lane group g = input column 16 qq + 4 g + e), in registers for the whole loop
    // (widths 16 / 32: 8 / 32 registers; at 64 they would be 128 and spill: that width keeps reading them from LDS)
    constexpr bool WREG = SQ <= 2;
    float wreg[WREG ? 2 * S / 16 : 1][4][SQ];
    if (WREG) {
#pragma unroll
        for (int qq = 0; qq < 2 * S / 16; ++qq)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ct = 0; ct < SQ; ++ct) wreg[WREG ? qq : 0][e][ct] = W0[(16 * qq + 4 * g + e) * LDW + c + 16 * ct];
    }
    int k_done = 0, any_prev = 0;
    TS_CLOCK();
    for (int it = 0; run && it < a.K; ++it) {
        const __amdgpu_buffer_rsrc_t r_in = buf_rsrc(a.states + (size_t)it * NS), r_out = buf_rsrc(a.states + (size_t)(it + 1) * NS);
        float *agg_t = a.agg + (size_t)it * NS;
        // ---- A. own row and neighbour sum of the tile's nodes -> LDS; the sum also goes to the tape ------------------------------------------
        {
            f32x4 own[NPASS], acc[NPASS];
            if (!LOCAL) {
#pragma unroll
                for (int p = 0; p < NPASS; ++p)
                    own[p] = buf_ld_sc1(r_in, csr.node[p] >= 0 ? (unsigned)csr.node[p] * (unsigned)(S * 4) + 16u * l4 : BUF_OFF);
            }
            if (LOCAL) csr.template gather<true>(acc, Xs, LDX, r_in); else csr.template gather<false>(acc, nullptr, 0, r_in);
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                float *xr = Xs + (p * NPP + q) * LDX + 4 * l4;
                if (!LOCAL) *reinterpret_cast<f32x4 *>(xr) = own[p];
                *reinterpret_cast<f32x4 *>(xr + S) = acc[p];
                if (csr.node[p] >= 0) *reinterpret_cast<f32x4 *>(agg_t + (size_t)csr.node[p] * S + 4 * l4) = acc[p];
            }
        }
        __syncthreads();
        TS_STAMP(0);
        if (bn) {
            // ---- B. column sums / squares of [state | agg] over the tile's rows -> partial slot ---------------------------------------------
            const int col = tid & (2 * S - 1), part_i = tid / (2 * S);          // 256 / (2 S) row groups (S = 64: 2, 32: 4, 16: 8)
            constexpr int NG = TS_NT / (2 * S), RPG = 64 / NG;
            // sums of (x - pivot) around the tile's OWN first row (merge_tile_stats: nothing of the mean's size to cancel, whatever the data)
            float s1 = 0.0f, s2 = 0.0f;
            const float pv = Xs[col];
            float xr[RPG];                                     // every read of the column piece issued before the first add (rows behind nt are zero)
#pragma unroll
            for (int u = 0; u < RPG; ++u) xr[u] = Xs[(part_i * RPG + u) * LDX + col];
#pragma unroll
            for (int u = 0; u < RPG; ++u) {
                const float x = part_i * RPG + u < nt ? xr[u] - pv : 0.0f;
                s1 += x; s2 = fmaf(x, x, s2);
            }
            red[part_i * 4 * S + col] = s1; red[part_i * 4 * S + 2 * S + col] = s2;
            __syncthreads();
            if (tid < 4 * S) {
                float t = 0.0f;
#pragma unroll
                for (int gq = 0; gq < NG; ++gq) t += red[gq * 4 * S + tid];
                const __amdgpu_buffer_rsrc_t r_p = buf_rsrc(a.part + ((size_t)(it & 1) * gridDim.x + blockIdx.x) * 6 * S);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(t), r_p, tid * 4, 0, 16);
                if (tid < 2 * S) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(Xs[tid]), r_p, (4 * S + tid) * 4, 0, 16);      // the pivots
            }
        }
        TS_STAMP(1);
        if (LOCAL) {
            if (it > 0 || bn) {           // one barrier: the statistics partials, and whether any node moved in iteration it - 1
                const bool moving = grid_barrier(gb, any_prev, &cont);
                if (it > 0 && !moving) break;
            }
        } else if (bn) grid_barrier(gb, 0, &cont);
        TS_STAMP(2);
        if (bn) {
            if (tid < 2 * S) {            // this column's statistics over ALL nodes: the tiles' shares in workgroup order
                const int kk = tid < S ? tid : tid - S;
                float ak = 0.0f, ck = 0.0f, mu = 0.0f;             // (pad columns: zero in, zero out)
                if (kk < a.Sw) {
                    const int k = (tid < S ? off_state : off_agg) + kk;               // BatchNorm column of this input column
                    float va;
                    merge_tile_stats(a.part + (size_t)(it & 1) * gridDim.x * 6 * S, wg_lo, wg_hi, 2 * S, tid,
                                     [&](int j) { return tab ? tt.begin[j + 1] - tt.begin[j] : min(64, a.N - 64 * j); }, inv_rows, mu, va);
                    ak = gamma_[k] / sqrtf(va + a.eps); ck = beta_[k];
                    if (blockIdx.x == wg_lo) { stats_[(size_t)it * 2 * in_s + k] = mu; stats_[(size_t)it * 2 * in_s + in_s + k] = va; }
                }
                st_a[tid] = ak; st_c[tid] = ck; piv[tid] = mu;
            }
            __syncthreads();
        }
        TS_STAMP(3);
        // ---- C. (a x + c) . W (+ constant part) on the matrix cores, operands swapped: lane (c, g) gets columns 16 ct + 4 g .. of row c -----
        f32x4 acc[SQ];
#pragma unroll
        for (int ct = 0; ct < SQ; ++ct) acc[ct] = cc[ct];
        // k-step (qq, e) takes input column 16 qq + 4 g + e from lane group g: one 16-byte LDS read feeds four steps
        const float *xrow = Xs + (16 * wave + c) * LDX + 4 * g;
#pragma unroll
        for (int qq = 0; qq < 2 * S / 16; ++qq) {
            f32x4 xv = *reinterpret_cast<const f32x4 *>(xrow + 16 * qq);
            if (bn) {         // a (x - mean) + beta: the column mean leaves the value first
                const f32x4 av = *reinterpret_cast<const f32x4 *>(st_a + 16 * qq + 4 * g), cv = *reinterpret_cast<const f32x4 *>(st_c + 16 * qq + 4 * g);
                const f32x4 mv = *reinterpret_cast<const f32x4 *>(piv + 16 * qq + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) xv[e] = fmaf(xv[e] - mv[e], av[e], cv[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float *wr = W0 + (16 * qq + 4 * g + e) * LDW + c;
#pragma unroll
                for (int ct = 0; ct < SQ; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(WREG ? wreg[WREG ? qq : 0][e][ct] : wr[16 * ct], xv[e], acc[ct], 0, 0, 0);
            }
        }
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");          // (MFMA results consumed behind a branch: see kernels_train_big.hpp)
        float d2 = 0.0f, n2 = 0.0f;
        float *orow_lds = Xs + (16 * wave + c) * LDX + 4 * g;
#pragma unroll
        for (int ct = 0; ct < SQ; ++ct) {
            f32x4 v = acc[ct];
            activate4(act, v);
            const f32x4 o = *reinterpret_cast<const f32x4 *>(orow_lds + 16 * ct);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = (oin && 16 * ct + 4 * g + e < a.Sw) ? v[e] : 0.0f; const float d = v[e] - o[e]; d2 = fmaf(d, d, d2); n2 = fmaf(o[e], o[e], n2); }
            if (LOCAL) {                  // (this wave's rows: no other wave reads them before the barrier below)
                *reinterpret_cast<f32x4 *>(orow_lds + 16 * ct) = v;
                if (oin) *reinterpret_cast<f32x4 *>(a.states + (size_t)(it + 1) * NS + (size_t)orow * S + 16 * ct + 4 * g) = v;
            } else buf_st_sc1(r_out, oin ? ((unsigned)orow * (unsigned)S + 16u * ct + 4u * g) * 4u : BUF_OFF, v);
        }
        d2 += __shfl_xor(d2, 16, 64); d2 += __shfl_xor(d2, 32, 64);
        n2 += __shfl_xor(n2, 16, 64); n2 += __shfl_xor(n2, 32, 64);
        const int any = (oin && sqrtf(d2) > a.thr * sqrtf(n2)) ? 1 : 0;
        k_done = it + 1;
        TS_STAMP(4);
        if (LOCAL) { any_prev = any; __syncthreads(); }
        else if (!grid_barrier(gb, any, &cont)) break;
        TS_STAMP(5);
    }
    TS_WRITE(0);
    if (LOCAL && __syncthreads_or(csr.bad) && tid == 0) a.k_out[1] = 1.0f;
    if (tid == 0 && blockIdx.x == 0) a.k_out[0] = (float)k_done;
    if (tid == 0 && gb.timed_out) a.k_out[1] = 2.0f;
}

template <int SQ>
inline size_t train_small_fwd_lds() {
    constexpr int S = 16 * SQ;
    return sizeof(float) * (64 * (2 * S + 4) + (2 * S) * (S + 4) + 3 * 2 * S + 512);
}

// ---- backward ----------------------------------------------------------------------------------------------------------------------------------
// BatchNormalization in terms of xhat = (x - mean) rstd (what the tile keeps in LDS):  with  Phat = xhat^T dZ,  q = colsum(dZ),
//   S1_k = sum_h W[k, h] q[h] (= d beta_k),  S2_k = sum_h W[k, h] Phat[k, h] (= d gamma_k),
//   dW[k, h] = gamma_k Phat[k, h] + beta_k q[h],       dx = gamma rstd (dy - S1 / N - xhat S2 / N),   dy = dZ . W^T.
// gamma and beta do not depend on the iteration, so sum_t Phat_t is accumulated as it is (one register add per tile and iteration) and
// scaled once at the end; the constant input columns have an iteration-invariant xhat, so their products are ONE pass over sum_t dZ_t at
// the end.  Every sum over rows is linear: a workgroup keeps the shares of its own tile (P, q, S1, S2 summed over the iterations) and the
// caller adds the shares in workgroup order.  Only the per-iteration TOTALS of S1 and S2 over the state / agg columns have to cross tiles
// (the BatchNorm input gradient): 4 S floats per workgroup and iteration.
struct TrainSmallBwd {
    int N, S, k;                 // k = iterations the forward pass executed
    int Sw;                      // the state's real width (<= 16 SQ): W is [in_s][Sw], G0 [N][Sw], the shares [in_s * Sw + Sw]
    const int *rowptr_s, *src_s; const float *w_s, *row_scale_s; // adjacency by SOURCE (transposed aggregate); w_s NULL = unit weights
    const float *row_scale;      // [N] scale of the by-destination operator when its entries depend on the destination only (then w_s = NULL)
    const float *states, *agg, *stats;                           // the forward tape
    int in_s, off_agg;
    ConstSegs cs;                // constant input segments: labels, aggregated labels, aggregated arc labels
    const float *W;              // [in_s][S]
    const float *gamma, *beta; float eps;
    int act;
    const float *G0;             // [N][S] d loss / d states[k] (from the output network)
    float *dxa;                  // [N][S] exchange buffer: d loss / d agg rows of the current iteration (general form)
    unsigned long long *bar;
    float *part;                 // [2][n_wg][4 S]  (S1 | S2) partials of the state / agg columns
    float *partW;                // [n_wg][in_s * S + S] every workgroup's share of the kernel and bias gradients (summed by the caller)
    float *partBN;               // [n_wg][2 in_s] every workgroup's share of d gamma | d beta (BatchNormalization)
    float inv_n;                 // 1 / N
    float *err;                  // [1] = 2 when a barrier timed out
    unsigned long long wait_ticks;   // bound of a barrier wait (buffer_ops.hpp: wait_until)
};

template <int SQ, bool HAS_W, bool LOCAL>
__global__ void __launch_bounds__(TS_NT, 1) k_train_small_bwd(TrainSmallBwd a, TileTab tt, TypeTab yt, TypeConsts yc) {
    using Csr = TileCsr<SQ, HAS_W, LOCAL>;
    constexpr int S = 16 * SQ, LPR = S / 4, NPP = Csr::NPP, NPASS = Csr::NPASS;
    constexpr int LDX = 2 * S + 4;        // == 4 (mod 32)
    constexpr int LDZ = S + 4, LDW = S + 4, LDC = 36;
    constexpr int NT_D = 2 * SQ * SQ;     // 16 x 16 tiles of Phat over the state / agg columns: (kt, ht), kt < 2 SQ
    constexpr int TPW = (NT_D + 3) / 4;   // ... per wave
    constexpr int NT_C = 2 * SQ, TPC = (NT_C + 3) / 4;        // tiles of the constants' product (32 columns)
    constexpr int NCH = 64 * LPR / TS_NT; // 16-byte row pieces per thread
    extern __shared__ __attribute__((aligned(16))) float ts_smem[];
    float *Xs = ts_smem;                  // [64][LDX]  xhat of [state_t | agg_t]; at the end [64][LDC] xhat of the constants
    float *Zs = Xs + 64 * LDX;            // [64][LDZ]  dZ
    float *Gs = Zs + 64 * LDZ;            // [64][LDZ]  G (d loss / d state of the tile)
    float *Da = Gs + 64 * LDZ;            // [64][LDZ]  d loss / d agg rows of the tile (LOCAL)
    float *Wr = Da + (LOCAL ? 64 * LDZ : 0);       // [2 S][LDW] kernel rows of the state / agg columns
    float *vec = Wr + 2 * S * LDW;
    float *ql_s = vec, *S1_s = ql_s + S, *S2_s = S1_s + 2 * S, *cfA = S2_s + 2 * S, *cfC = cfA + 2 * S, *cfB = cfC + 2 * S,
          *rs_s = cfB + 2 * S, *sh_s = rs_s + 2 * S, *red = sh_s + 2 * S;      // red: [4][2 S] / [4][S] scratch
    __shared__ int cont;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const bool tab = LOCAL || tt.n > 0;
    const int n0 = tab ? tt.begin[blockIdx.x] : (int)blockIdx.x * 64;
    const int nt = tab ? tt.begin[blockIdx.x + 1] - n0 : min(64, a.N - n0);
    // heterogeneous models: this tile's node type selects the network and where its shares go (TypeTab); sums over "all rows" of a
    // BatchNormalization span the tiles [wg_lo, wg_hi) of the type
    int ty = 0;
    if (yt.n > 0) { while (ty + 1 < yt.n && (int)blockIdx.x >= yt.wg_begin[ty + 1]) ++ty; }
    const float *Wk = yt.n > 0 ? yt.W[ty] : a.W, *gamma_ = yt.n > 0 ? yt.gamma[ty] : a.gamma, *beta_ = yt.n > 0 ? yt.beta[ty] : a.beta;
    const float *stats_ = yt.n > 0 ? yt.stats[ty] : a.stats;
    const int in_s = yt.n > 0 ? yt.in_s[ty] : a.in_s, off_state = yt.n > 0 ? yt.off_state[ty] : 0, off_agg = yt.n > 0 ? yt.off_agg[ty] : a.off_agg;
    const int act = yt.n > 0 ? yt.act[ty] : a.act;
    const unsigned wg_lo = yt.n > 0 ? (unsigned)yt.wg_begin[ty] : 0u, wg_hi = yt.n > 0 ? (unsigned)yt.wg_begin[ty + 1] : gridDim.x;
    const float inv_n = yt.n > 0 ? 1.0f / (float)yt.count[ty] : a.inv_n;
    const ConstSegs &cs = yt.n > 0 ? yc.cs[ty] : a.cs;
    const int *perm = yt.perm;
    const bool bn = gamma_ != nullptr;
    const size_t NS = (size_t)a.N * S;
    GridBar gb{a.bar, gridDim.x, 0, 0u, 0u, 0, a.wait_ticks};
    auto wrow_dyn = [&](int j) { return j < S ? off_state + j : off_agg + (j - S); };      // weight row / BatchNorm column of tile column j (valid j only)
    auto valid_dyn = [&](int j) { return (j < S ? j : j - S) < a.Sw; };           // pad columns of the padded state width carry zeros

    for (int i = tid; i < 2 * S * S; i += TS_NT) {
        const int j = i / S, h = i % S;
        Wr[j * LDW + h] = (valid_dyn(j) && h < a.Sw) ? Wk[(size_t)wrow_dyn(j) * a.Sw + h] : 0.0f;
    }
    for (int i = tid; i < 64 * S; i += TS_NT) {          // (G0 is in the caller's node order)
        const int rr = i / S, h = i % S;
        Gs[rr * LDZ + h] = (rr < nt && h < a.Sw) ? a.G0[(size_t)(perm ? perm[n0 + rr] : n0 + rr) * a.Sw + h] : 0.0f;
    }
    // xhat = (x - sh) rs of iteration t (rs = rstd, sh = mean; 1 / 0 without BatchNormalization): centred first, then scaled
    auto coefficients = [&](int t) {
        if (tid < 2 * S) {
            float r_ = 1.0f, s_ = 0.0f;
            if (bn && valid_dyn(tid)) { const float *st = stats_ + (size_t)t * 2 * in_s; const int k = wrow_dyn(tid); r_ = 1.0f / sqrtf(st[in_s + k] + a.eps); s_ = st[k]; }
            rs_s[tid] = r_; sh_s[tid] = s_;
        }
    };
    Csr csr;
    csr.load(n0, nt, a.rowptr_s, a.src_s, a.w_s, a.row_scale_s, yt.perm, yt.inv);
    const int q = csr.q, l4 = csr.l4;
    f32x4 accP[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) accP[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 dzsum[NCH];
#pragma unroll
    for (int u = 0; u < NCH; ++u) dzsum[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float qsum = 0.0f, S1sum = 0.0f, S2sum = 0.0f;                   // thread h < S / thread j < 2 S: this tile's shares over all iterations
    const bool oin = 16 * wave + c < nt;
    const int orow = n0 + 16 * wave + c;
    const float rs = (a.row_scale && oin) ? a.row_scale[perm ? perm[orow] : orow] : 1.0f;
    const __amdgpu_buffer_rsrc_t r_dxa = buf_rsrc(a.dxa);
    // this thread's row pieces: rows (tid + 256 u) / LPR, chunk tid % LPR
    const int ch = tid % LPR;
    f32x4 xs[NCH], xa[NCH], y[NCH];
    auto fetch = [&](int t, f32x4 (&fs)[NCH], f32x4 (&fa)[NCH]) {
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const int rr = (tid + u * TS_NT) / LPR;
            fs[u] = fa[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (rr < nt && t >= 0) {
                const size_t o = (size_t)t * NS + (size_t)(n0 + rr) * S + 4 * ch;
                fs[u] = *reinterpret_cast<const f32x4 *>(a.states + o);
                fa[u] = *reinterpret_cast<const f32x4 *>(a.agg + o);
            }
        }
    };
    if (a.k > 0) {
        fetch(a.k - 1, xs, xa);
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const int rr = (tid + u * TS_NT) / LPR;
            y[u] = rr < nt ? *reinterpret_cast<const f32x4 *>(a.states + (size_t)a.k * NS + (size_t)(n0 + rr) * S + 4 * ch) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        coefficients(a.k - 1);
    }
    __syncthreads();

    TS_CLOCK();
    for (int t = a.k - 1; t >= 0; --t) {
        // ---- A. xhat and dZ = G (.) act'(output) of the tile into LDS; the rows of iteration t - 1 are requested now -----------------------------
        f32x4 xs_n[NCH], xa_n[NCH];
        fetch(t - 1, xs_n, xa_n);
        {
            f32x4 qp = {0.f, 0.f, 0.f, 0.f};
            const f32x4 r0 = *reinterpret_cast<const f32x4 *>(rs_s + 4 * ch), s0 = *reinterpret_cast<const f32x4 *>(sh_s + 4 * ch);
            const f32x4 r1 = *reinterpret_cast<const f32x4 *>(rs_s + S + 4 * ch), s1 = *reinterpret_cast<const f32x4 *>(sh_s + S + 4 * ch);
#pragma unroll
            for (int u = 0; u < NCH; ++u) {
                const int rr = (tid + u * TS_NT) / LPR;
                f32x4 hs, ha, gz = *reinterpret_cast<const f32x4 *>(Gs + rr * LDZ + 4 * ch);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    hs[e] = rr < nt ? (xs[u][e] - s0[e]) * r0[e] : 0.0f;
                    ha[e] = rr < nt ? (xa[u][e] - s1[e]) * r1[e] : 0.0f;
                    gz[e] *= activate_grad_from_output(act, y[u][e]);
                }
                *reinterpret_cast<f32x4 *>(Xs + rr * LDX + 4 * ch) = hs;
                *reinterpret_cast<f32x4 *>(Xs + rr * LDX + S + 4 * ch) = ha;
                *reinterpret_cast<f32x4 *>(Zs + rr * LDZ + 4 * ch) = gz;
                dzsum[u] += gz; qp += gz;
            }
            // q of the tile: this thread's rows, then the lanes with the same chunk (stride LPR), then the four waves (in order)
#pragma unroll
            for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
                for (int e = 0; e < 4; ++e) qp[e] += __shfl_xor(qp[e], off, 64);
            if (lane < LPR) *reinterpret_cast<f32x4 *>(red + wave * S + 4 * ch) = qp;
        }
        __syncthreads();
        TS_STAMP(0);
        if (tid < S) { const float t3 = ((red[tid] + red[S + tid]) + red[2 * S + tid]) + red[3 * S + tid]; ql_s[tid] = t3; qsum += t3; }
        // ---- B. dy = dZ . W^T on the matrix cores (operands swapped: lane (c, g) holds columns j0 .. j0 + 3 of row c of its wave) ------------
        f32x4 dy[2][SQ];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int u = 0; u < SQ; ++u) dy[half][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float *zrow = Zs + (16 * wave + c) * LDZ + 4 * g;
#pragma unroll 2
            for (int qq = 0; qq < S / 16; ++qq) {                  // k-step (qq, e) = dZ column 16 qq + 4 g + e: both operands are 16-byte reads
                const f32x4 zv = *reinterpret_cast<const f32x4 *>(zrow + 16 * qq);
#pragma unroll
                for (int u = 0; u < SQ; ++u) {
                    const f32x4 wv = *reinterpret_cast<const f32x4 *>(Wr + (half * S + 16 * u + c) * LDW + 16 * qq + 4 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dy[half][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[e], zv[e], dy[half][u], 0, 0, 0);
                }
            }
        }
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
        TS_STAMP(1);
        // Phat_wg = xhat^T dZ (kept local, summed over the iterations in registers): issued here, behind the partial store when
        // BatchNormalization makes the tiles wait for each other
        auto phat = [&]() {
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int ti = wave + 4 * i;
                if (ti < NT_D) {
                    const int kt = ti / SQ, ht = ti % SQ;
                    const float *xp = Xs + g * LDX + 16 * kt + c, *zp = Zs + g * LDZ + 16 * ht + c;
#pragma unroll 4
                    for (int ms = 0; ms < 16; ++ms)
                        accP[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(xp[4 * ms * LDX], zp[4 * ms * LDZ], accP[i], 0, 0, 0);
                }
            }
        };
        if (bn) {
            // S1_j = colsum(dy)_j (= (W q)_j) and S2_j = colsum(dy (.) xhat)_j over the tile's rows: the 16 rows of a wave by lane
            // exchange, the four waves in wave order
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int u = 0; u < SQ; ++u) {
                    const int j0 = half * S + 16 * u + 4 * g;
                    const f32x4 x = *reinterpret_cast<const f32x4 *>(Xs + (16 * wave + c) * LDX + j0);
                    f32x4 s1 = dy[half][u], s2 = dy[half][u] * x;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { s1[e] = row16_sum_to_lane15(s1[e]); s2[e] = row16_sum_to_lane15(s2[e]); }      // (VALU only)
                    if (c == 15) {
                        *reinterpret_cast<f32x4 *>(red + 4 * S + wave * 4 * S + j0) = s1;
                        *reinterpret_cast<f32x4 *>(red + 4 * S + wave * 4 * S + 2 * S + j0) = s2;
                    }
                }
            __syncthreads();
            if (tid < 4 * S) {
                const float *rp = red + 4 * S + tid;
                const float v = ((rp[0] + rp[4 * S]) + rp[8 * S]) + rp[12 * S];
                if (tid < 2 * S) S1sum += v; else S2sum += v;
                const __amdgpu_buffer_rsrc_t r_p = buf_rsrc(a.part + ((size_t)(t & 1) * gridDim.x + blockIdx.x) * 4 * S);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r_p, tid * 4, 0, 16);
            }
            TS_STAMP(2);
            phat();
            asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
            grid_barrier(gb, 0, &cont);
            TS_STAMP(3);
            if (tid < 4 * S) {
                const float tt2 = (float)sum_partials(a.part + (size_t)(t & 1) * gridDim.x * 4 * S, wg_lo, wg_hi, 4 * S, tid);
                if (tid < 2 * S) S1_s[tid] = tt2; else S2_s[tid - 2 * S] = tt2;
            }
            __syncthreads();
            if (tid < 2 * S) {
                const float Ac = valid_dyn(tid) ? gamma_[wrow_dyn(tid)] * rs_s[tid] : 0.0f;
                cfA[tid] = Ac; cfC[tid] = -Ac * S2_s[tid] * inv_n; cfB[tid] = -Ac * S1_s[tid] * inv_n;
            }
            __syncthreads();
        } else {
            phat();
            asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
        }
        TS_STAMP(4);
        // ---- D. dx = BN-gradient(dy); state half -> Gs, agg half (scaled once per row) -> Da / memory ------------------------------------
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int u = 0; u < SQ; ++u) {
                const int j0 = half * S + 16 * u + 4 * g;
                f32x4 v = dy[half][u];
                if (bn) {
                    const f32x4 x = *reinterpret_cast<const f32x4 *>(Xs + (16 * wave + c) * LDX + j0);
                    const f32x4 A_ = *reinterpret_cast<const f32x4 *>(cfA + j0), C_ = *reinterpret_cast<const f32x4 *>(cfC + j0), B_ = *reinterpret_cast<const f32x4 *>(cfB + j0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(A_[e], v[e], fmaf(C_[e], x[e], B_[e]));
                }
                if (!oin) v = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (half == 0) {
                    *reinterpret_cast<f32x4 *>(Gs + (16 * wave + c) * LDZ + 16 * u + 4 * g) = v;      // dx_state: the own part of the next G
                } else {
                    v *= rs;
                    if (LOCAL) *reinterpret_cast<f32x4 *>(Da + (16 * wave + c) * LDZ + 16 * u + 4 * g) = v;
                    else buf_st_sc1(r_dxa, oin ? ((unsigned)orow * (unsigned)S + 16u * u + 4u * g) * 4u : BUF_OFF, v);
                }
            }
        TS_STAMP(5);
        if (LOCAL) __syncthreads(); else grid_barrier(gb, 0, &cont);
        TS_STAMP(6);
        // ---- E. G = dx_state + Adj . dx_agg: gather by source; the coefficients of the next iteration -----------------------------------------------
        {
            f32x4 acc[NPASS];
            if (LOCAL) csr.template gather<true>(acc, Da, LDZ, r_dxa); else csr.template gather<false>(acc, nullptr, 0, r_dxa);
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                float *gr = Gs + (p * NPP + q) * LDZ + 4 * l4;
                f32x4 gv = *reinterpret_cast<const f32x4 *>(gr);
                gv += acc[p];
                if (csr.node[p] < 0) gv = (f32x4){0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4 *>(gr) = gv;
            }
        }
        if (t > 0) coefficients(t - 1);
#pragma unroll
        for (int u = 0; u < NCH; ++u) { y[u] = xs[u]; xs[u] = xs_n[u]; xa[u] = xa_n[u]; }       // states[t] is the output of iteration t - 1
        __syncthreads();
        TS_STAMP(7);
    }
    TS_WRITE(1);
    // ---- the shares of this workgroup (heterogeneous models: the type's own arrays, one slot per tile of the type) ------------------------------
    float *pw = yt.n > 0 ? yt.partW[ty] + (size_t)(blockIdx.x - wg_lo) * ((size_t)in_s * a.Sw + a.Sw) : a.partW + (size_t)blockIdx.x * ((size_t)in_s * a.Sw + a.Sw);
    float *pb = !bn ? nullptr : yt.n > 0 ? yt.partBN[ty] + (size_t)(blockIdx.x - wg_lo) * 2 * in_s : a.partBN + (size_t)blockIdx.x * 2 * in_s;
    if (tid < S) { if (tid < a.Sw) pw[(size_t)in_s * a.Sw + tid] = qsum; ql_s[tid] = qsum; }
    __syncthreads();
    // kernel rows of the state / agg columns: gamma_k sum_t Phat + beta_k sum_t q
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int ti = wave + 4 * i;
        if (ti < NT_D) {
            const int kt = ti / SQ, ht = ti % SQ;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int j = 16 * kt + 4 * g + reg, k = wrow_dyn(j), h = 16 * ht + c;
                if (!valid_dyn(j) || h >= a.Sw) continue;
                float v = accP[i][reg];
                if (bn) v = fmaf(gamma_[k], v, beta_[k] * ql_s[h]);
                pw[(size_t)k * a.Sw + h] = v;
            }
        }
    }
    if (bn && tid < 2 * S) { if (valid_dyn(tid)) pb[in_s + wrow_dyn(tid)] = S1sum; }              // d beta share
    else if (bn && tid < 4 * S) { if (valid_dyn(tid - 2 * S)) pb[wrow_dyn(tid - 2 * S)] = S2sum; }  // d gamma share
    // the constant input columns: xhat is the same in every iteration, so Phat_c = xhat_c^T (sum_t dZ_t), once - in blocks of 32 columns
    // (a homogeneous model has 2 L + A <= 32 of them on this path; a type's network of a heterogeneous one d_t + sum d + A)
    const int Kc = (cs.n > 0 ? cs.width[0] : 0) + (cs.n > 1 ? cs.width[1] : 0) + (cs.n > 2 ? cs.width[2] : 0);
    if (Kc > 0) {
#pragma unroll
        for (int u = 0; u < NCH; ++u) *reinterpret_cast<f32x4 *>(Zs + ((tid + u * TS_NT) / LPR) * LDZ + 4 * ch) = dzsum[u];
    }
    for (int cb = 0; cb < Kc; cb += 32) {
        int *wrow_c = reinterpret_cast<int *>(red);              // [32] weight row (= BatchNorm column) of constant column cb + jj, -1 = padding
        __syncthreads();                                         // (the previous block's readers of Xs / red are done; Zs is staged)
        if (tid < 32) {
            int r_ = -1, b0 = 0;
            const int jc = cb + tid;
#pragma unroll
            for (int sg = 0; sg < 3; ++sg) { if (sg < cs.n && jc >= b0 && jc < b0 + cs.width[sg]) r_ = cs.wrow[sg] + (jc - b0); if (sg < cs.n) b0 += cs.width[sg]; }
            wrow_c[tid] = r_;
        }
        __syncthreads();
        for (int i = tid; i < 64 * 32; i += TS_NT) {
            const int rr = i >> 5, jj = i & 31, jc = cb + jj;
            float v = 0.0f;
            if (rr < nt && jc < Kc) {
                int b0 = 0;
                const size_t orig = (size_t)(perm ? perm[n0 + rr] : n0 + rr);        // (the constants are in the caller's node order)
#pragma unroll
                for (int sg = 0; sg < 3; ++sg) { if (sg < cs.n && jc >= b0 && jc < b0 + cs.width[sg]) v = cs.ptr[sg][orig * cs.ld[sg] + (jc - b0)]; if (sg < cs.n) b0 += cs.width[sg]; }
                if (bn) { const int k = wrow_c[jj]; const float r_ = 1.0f / sqrtf(stats_[in_s + k] + a.eps); v = (v - stats_[k]) * r_; }      // (constants: the statistics of any iteration)
            }
            Xs[rr * LDC + jj] = v;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TPC; ++i) {
            const int ti = wave + 4 * i;
            if (ti < NT_C) {
                const int kt = ti / SQ, ht = ti % SQ;
                f32x4 Pc = {0.f, 0.f, 0.f, 0.f};
                const float *xp = Xs + g * LDC + 16 * kt + c, *zp = Zs + g * LDZ + 16 * ht + c;
#pragma unroll 4
                for (int ms = 0; ms < 16; ++ms) Pc = __builtin_amdgcn_mfma_f32_16x16x4f32(xp[4 * ms * LDC], zp[4 * ms * LDZ], Pc, 0, 0, 0);
                asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int jj = 16 * kt + 4 * g + reg, k = cb + jj < Kc ? wrow_c[jj] : -1;
                    const bool hv = 16 * ht + c < a.Sw;
                    const float w = (k >= 0 && hv) ? Wk[(size_t)k * a.Sw + 16 * ht + c] : 0.0f;
                    float s2 = w * Pc[reg], s1 = w * ql_s[16 * ht + c];
#pragma unroll
                    for (int off = 1; off < 16; off <<= 1) { s2 += __shfl_xor(s2, off, 64); s1 += __shfl_xor(s1, off, 64); }
                    if (k >= 0) {
                        if (hv) pw[(size_t)k * a.Sw + 16 * ht + c] = bn ? fmaf(gamma_[k], Pc[reg], beta_[k] * ql_s[16 * ht + c]) : Pc[reg];
                        if (bn && c == 0) { red[64 + ht * 32 + jj] = s2; red[64 + 4 * 32 + ht * 32 + jj] = s1; }
                    }
                }
            }
        }
        __syncthreads();
        if (bn && tid < 32 && cb + tid < Kc) {
            float s2 = 0.0f, s1 = 0.0f;
#pragma unroll
            for (int ht = 0; ht < SQ; ++ht) { s2 += red[64 + ht * 32 + tid]; s1 += red[64 + 4 * 32 + ht * 32 + tid]; }
            pb[wrow_c[tid]] = s2; pb[in_s + wrow_c[tid]] = s1;
        }
    }
    if (tid == 0 && gb.timed_out) a.err[1] = 2.0f;
    if (__syncthreads_or(gb.timed_out || (LOCAL && csr.bad)) && tid == 0) pw[0] = __builtin_nanf("");      // loud: the gradients are invalid
}

template <int SQ>
inline size_t train_small_bwd_lds(bool local) {
    constexpr int S = 16 * SQ;
    return sizeof(float) * (64 * (2 * S + 4) + (local ? 3 : 2) * 64 * (S + 4) + 2 * S * (S + 4) + (S + 16 * S) + 4 * S + 16 * S + 512);
}

}  // namespace gnn
