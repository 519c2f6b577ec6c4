// Training on SMALL graphs (a merged MUTAG batch: ~1 k nodes): the K training-mode iterations of the forward pass - and the k
// iterations of back-propagation through them - as ONE persistent launch each (reference GNN/Models/GNN.py:277-306: `self(x,
// training=True)` and `tape.gradient` through the unrolled loop).
//
// Round 2 ran such a step as ~450 dependent launches of ~8 us (4.3 ms at d = 32 x 50 iterations): nine per iteration pair, each
// keeping 15 of the 256 CUs busy for a few microseconds.  Here one workgroup owns one 64-node tile for the whole loop, as in
// kernel_state_small.hpp, and what an iteration needs from the other tiles crosses a grid barrier (one 64-bit arrival counter per
// parity, agent-scope add + poll, rows written and read `sc1`):
//
//   forward iteration t   gather [own | neighbour sum] of the tile from states[t] -> LDS, neighbour sums -> the tape;
//                         BatchNormalization in training mode needs the statistics of ALL nodes: the tile's column sums / squares of
//                         [state | agg] go to a partial slot, BARRIER, every workgroup adds the partials in workgroup order (the same
//                         bits everywhere), folds mean / variance into its copy of the weights; [state | agg] . W on the matrix cores
//                         from LDS (operands swapped: the result is row-major), activation, predicate, rows -> states[t + 1]; BARRIER
//                         (its counter also carries "some node still moves": every workgroup leaves the loop after the same k).
//   backward iteration t  dZ = G (.) act'(states[t + 1]); P_wg = X^T dZ of the tile on the matrix cores - kept LOCAL: the weight
//                         gradient a (.) P + c q^T is linear in P, so every workgroup accumulates its own share in registers over
//                         all iterations and the shares are summed once at the end; what the BatchNorm input gradient needs of the
//                         other tiles is only q = colsum(dZ) and S2_k = sum_h W[k, h] P[k, h] - 127 floats per workgroup: partial
//                         slot, BARRIER, summed in order; dx = BN-gradient(dZ . W^T) on the matrix cores, the agg half scaled once per
//                         row and written `sc1`, BARRIER; G = dx_state + Adj . dx_agg gathered by source.
// Two barriers per iteration each way (one forward without BatchNormalization), ~1.5 us each.  Every workgroup must be resident
// (grid <= CUs, one workgroup per CU); polls sleep and are bounded: an expired wait raises the error word, results are invalid.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_general.hpp"
#include "kernel_state_fused4.hpp"      // activate4
#include "kernels_train.hpp"            // activate_grad_from_output
#include "buffer_ops.hpp"

namespace gnn {

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
};
__device__ __forceinline__ bool grid_barrier(GridBar &gb, int any, int *cont_lds) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's sc1 stores have left
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) {
        unsigned long long *ctr = gb.bar + (gb.bi & 1);
        __hip_atomic_fetch_add(ctr, 1ull + ((unsigned long long)(any ? 1u : 0u) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (unsigned)(gb.bi / 2 + 1) * gb.n_wg;
        unsigned long long v = 0;
        int spin = 0;
        for (; spin < (1 << 22); ++spin) {
            v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)v >= target) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (spin == (1 << 22)) gb.timed_out = 1;
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
//   Cc[n, h] = b[h] + sum over the constant input columns k of (a_k x[n, k] + c_k) W[k, h],   a = gamma rstd, c = beta - mean a
// (a = 1, c = 0 without BatchNormalization; the constants' batch statistics do not change between iterations).
struct ConstSegs { const float *ptr[3]; int ld[3], width[3], wrow[3]; int n; };
__global__ void __launch_bounds__(256)
k_train_small_const(int N, int S, ConstSegs cs, const float *__restrict__ W, const float *__restrict__ b, const float *gamma, const float *beta,
                    const float *mean, const float *var, float eps, float *__restrict__ Cc) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * S) return;
    const int n = i / S, h = i % S;
    float acc = b[h];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        if (s >= cs.n) break;
        for (int j = 0; j < cs.width[s]; ++j) {
            const int k = cs.wrow[s] + j;
            float x = cs.ptr[s][(size_t)n * cs.ld[s] + j];
            if (gamma) { const float a = gamma[k] / sqrtf(var[k] + eps); x = fmaf(x, a, beta[k] - mean[k] * a); }
            acc = fmaf(x, W[(size_t)k * S + h], acc);
        }
    }
    Cc[i] = acc;
}

// ---- forward ---------------------------------------------------------------------------------------------------------------------------------
struct TrainSmallFwd {
    int N, S, K;
    const int *rowptr, *src; const float *w, *row_scale;       // adjacency by destination
    float *states;               // [K + 1][N][S]; [0] = state_0 (the caller's)
    float *agg;                  // [K][N][S] neighbour sums, kept for the backward pass
    float *stats;                // [K][2 in_s] (mean | var): constant columns pre-filled, the state / agg columns written here
    int in_s, off_agg;
    const float *W;              // first-layer kernel [in_s][S]
    const float *gamma, *beta; float eps;                       // BatchNormalization or NULL
    int act;
    const float *Cc;             // [N][S] constant part (k_train_small_const)
    float thr; int no_exit;
    const int *flag0;            // predicate of state_0 (one word)
    unsigned long long *bar;     // two arrival counters, zero
    float *part;                 // [2][n_wg][4 S] statistics partials (parity of the iteration)
    float *k_out; int *err;
};

template <int SQ, bool HAS_W>
__global__ void __launch_bounds__(TS_NT, 1) k_train_small_fwd(TrainSmallFwd a) {
    constexpr int S = 16 * SQ, LPR = S / 4, NPP = TS_NT / LPR, NPASS = 64 / NPP, IPL = 16 / LPR;
    constexpr int LDX = 2 * S + 4;        // row stride of the [own | agg] tile: == 4 (mod 32) dwords, 16-B chunks conflict-free
    constexpr int LDW = S + 4;            // weight rows: the A-operand read takes rows 4 g + e of a 16-row block, == 4 (mod 32) spreads them
    extern __shared__ __attribute__((aligned(16))) float ts_smem[];
    float *Xs = ts_smem;                  // [64][LDX]
    float *W0 = Xs + 64 * LDX;            // [2 S][LDW] kernel rows of the state / agg columns
    float *Wsc = W0 + 2 * S * LDW;        // [2 S][LDW] the same with the BatchNorm scale of this iteration folded in
    float *st_a = Wsc + 2 * S * LDW;      // [2 S] a_k, then [2 S] c_k, [S] bias_dyn, [2 S] pivots of the statistics, reduction scratch [512]
    float *st_c = st_a + 2 * S, *bias_dyn = st_c + 2 * S, *piv = bias_dyn + S, *red = piv + 2 * S;
    __shared__ int cont;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int tile0 = blockIdx.x * 64;
    const bool bn = a.gamma != nullptr;
    const size_t NS = (size_t)a.N * S;
    GridBar gb{a.bar, gridDim.x, 0, 0u, 0u, 0};

    for (int i = tid; i < 2 * S * S; i += TS_NT) {
        const int k = i / S, h = i % S;
        const float v = a.W[(size_t)((k < S ? 0 : a.off_agg - S) + k) * S + h];
        W0[k * LDW + h] = v; Wsc[k * LDW + h] = v;
    }
    if (tid < S) bias_dyn[tid] = 0.0f;
    if (tid < 2 * S) piv[tid] = 0.0f;
    // ---- iteration-invariant: the CSR rows of this lane group's nodes (first 16 source ids in registers) ----------------------------
    const int q = tid / LPR, l4 = tid % LPR;
    int jn[NPASS], beg[NPASS], end[NPASS], ids[NPASS][IPL];
    float wts[NPASS][IPL], scl[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int m = tile0 + p * NPP + q;
        jn[p] = m < a.N ? m : -1;
        beg[p] = end[p] = 0; scl[p] = 1.0f;
        if (jn[p] >= 0) { beg[p] = a.rowptr[m]; end[p] = a.rowptr[m + 1]; if (a.row_scale) scl[p] = a.row_scale[m]; }
#pragma unroll
        for (int u = 0; u < IPL; ++u) {
            const int e = beg[p] + u * LPR + l4;
            ids[p][u] = e < end[p] ? a.src[e] : 0;
            wts[p][u] = (HAS_W && e < end[p]) ? a.w[e] : 0.0f;
        }
    }
    // the constant part of this lane's output chunks (row 16 wave + c, columns 16 ct + 4 g ..)
    const int orow = tile0 + 16 * wave + c;
    const bool oin = orow < a.N;
    f32x4 cc[SQ];
#pragma unroll
    for (int ct = 0; ct < SQ; ++ct)
        cc[ct] = oin ? *reinterpret_cast<const f32x4 *>(a.Cc + (size_t)orow * S + 16 * ct + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
    bool run = a.no_exit != 0 || __hip_atomic_load(a.flag0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    __syncthreads();

    int k_done = 0;
    for (int it = 0; run && it < a.K; ++it) {
        const __amdgpu_buffer_rsrc_t r_in = buf_rsrc(a.states + (size_t)it * NS), r_out = buf_rsrc(a.states + (size_t)(it + 1) * NS);
        float *agg_t = a.agg + (size_t)it * NS;
        // ---- A. gather: own row and neighbour sum of the tile's nodes -> LDS; the sum also goes to the tape ------------------------------
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            f32x4 own = buf_ld_sc1(r_in, jn[p] >= 0 ? (unsigned)jn[p] * (unsigned)(S * 4) + 16u * l4 : BUF_OFF);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            int idc[IPL]; float wsc[IPL];
#pragma unroll
            for (int u = 0; u < IPL; ++u) { idc[u] = ids[p][u]; wsc[u] = wts[p][u]; }
            int rem = end[p] - beg[p], eb = beg[p];
#pragma unroll 1
            while (true) {
                f32x4 v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const unsigned off = (unsigned)__shfl(idc[i / LPR], i % LPR, LPR) * (unsigned)(S * 4) + 16u * l4;
                    v[i] = buf_ld_sc1(r_in, i < rem ? off : BUF_OFF);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (HAS_W) acc += __shfl(wsc[i / LPR], i % LPR, LPR) * v[i];
                    else acc += v[i];
                }
                rem -= 16; eb += 16;
                if (!__any(rem > 0)) break;
#pragma unroll
                for (int u = 0; u < IPL; ++u) {
                    const int e = eb + u * LPR + l4;
                    idc[u] = e < end[p] ? a.src[e] : 0;
                    wsc[u] = (HAS_W && e < end[p]) ? a.w[e] : 0.0f;
                }
            }
            acc *= scl[p];
            float *xr = Xs + (p * NPP + q) * LDX + 4 * l4;
            *reinterpret_cast<f32x4 *>(xr) = own;
            *reinterpret_cast<f32x4 *>(xr + S) = acc;
            if (jn[p] >= 0) *reinterpret_cast<f32x4 *>(agg_t + (size_t)jn[p] * S + 4 * l4) = acc;
        }
        __syncthreads();
        if (bn) {
            // ---- B. column sums / squares of [state | agg] over the tile's rows -> partial slot -> BARRIER -> batch statistics -----------
            {
                const int col = tid & (2 * S - 1), part_i = tid / (2 * S);          // 256 / (2 S) row groups (S = 64: 2, 32: 4, 16: 8)
                constexpr int NG = TS_NT / (2 * S), RPG = 64 / NG;
                // sums of (x - pivot), pivot = the previous iteration's mean (the same bits in every workgroup): E[d^2] - E[d]^2 does
                // not cancel once the states settle
                float s1 = 0.0f, s2 = 0.0f;
                const float pv = piv[col];
                for (int rr = part_i * RPG; rr < (part_i + 1) * RPG; ++rr) {
                    const float x = (tile0 + rr < a.N) ? Xs[rr * LDX + col] - pv : 0.0f;
                    s1 += x; s2 = fmaf(x, x, s2);
                }
                red[part_i * 4 * S + col] = s1; red[part_i * 4 * S + 2 * S + col] = s2;
                __syncthreads();
                if (tid < 4 * S) {
                    float t = 0.0f;
                    for (int gq = 0; gq < NG; ++gq) t += red[gq * 4 * S + tid];
                    const __amdgpu_buffer_rsrc_t r_p = buf_rsrc(a.part + ((size_t)(it & 1) * gridDim.x + blockIdx.x) * 4 * S);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(t), r_p, tid * 4, 0, 16);
                }
            }
            grid_barrier(gb, 0, &cont);
            if (tid < 4 * S) {                                     // (sum | square, column): partials in workgroup order, double accumulator
                const __amdgpu_buffer_rsrc_t r_p = buf_rsrc(a.part + (size_t)(it & 1) * gridDim.x * 4 * S);
                double t = 0.0;
                for (unsigned wg = 0; wg < gridDim.x; ++wg)
                    t += (double)__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_p, (int)((wg * 4 * S + tid) * 4), 0, 16));
                red[tid] = (float)(t / (double)a.N);               // E[x - pivot] (first 2 S), E[(x - pivot)^2] (next 2 S)
            }
            __syncthreads();
            if (tid < 2 * S) {
                const int k = (tid < S ? 0 : a.off_agg - S) + tid;                    // BatchNorm column of this input column
                const float dm = red[tid], va = fmaxf(red[2 * S + tid] - dm * dm, 0.0f), mu = piv[tid] + dm;
                piv[tid] = mu;
                const float ak = a.gamma[k] / sqrtf(va + a.eps);
                st_a[tid] = ak; st_c[tid] = a.beta[k] - mu * ak;
                if (blockIdx.x == 0) { a.stats[(size_t)it * 2 * a.in_s + k] = mu; a.stats[(size_t)it * 2 * a.in_s + a.in_s + k] = va; }
            }
            __syncthreads();
            for (int i = tid; i < 2 * S * S; i += TS_NT) { const int k = i / S, h = i % S; Wsc[k * LDW + h] = st_a[k] * W0[k * LDW + h]; }
            if (tid < S) {
                float t = 0.0f;
                for (int k = 0; k < 2 * S; ++k) t = fmaf(st_c[k], W0[k * LDW + tid], t);
                bias_dyn[tid] = t;
            }
            __syncthreads();
        }
        // ---- C. [state | agg] . W (+ constant part) on the matrix cores, operands swapped: lane (c, g) gets columns 16 ct + 4 g .. of row c ---
        f32x4 acc[SQ];
#pragma unroll
        for (int ct = 0; ct < SQ; ++ct) acc[ct] = cc[ct] + *reinterpret_cast<const f32x4 *>(bias_dyn + 16 * ct + 4 * g);
        // k-step (qq, e) takes input column 16 qq + 4 g + e from lane group g: one 16-byte LDS read feeds four steps
        const float *xrow = Xs + (16 * wave + c) * LDX + 4 * g;
#pragma unroll 2
        for (int qq = 0; qq < 2 * S / 16; ++qq) {
            const f32x4 xv = *reinterpret_cast<const f32x4 *>(xrow + 16 * qq);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float *wr = Wsc + (16 * qq + 4 * g + e) * LDW + c;
#pragma unroll
                for (int ct = 0; ct < SQ; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[16 * ct], xv[e], acc[ct], 0, 0, 0);
            }
        }
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");          // (MFMA results consumed behind a branch: see kernels_train_big.hpp)
        float d2 = 0.0f, n2 = 0.0f;
#pragma unroll
        for (int ct = 0; ct < SQ; ++ct) {
            f32x4 v = acc[ct];
            activate4(a.act, v);
            const f32x4 o = *reinterpret_cast<const f32x4 *>(Xs + (16 * wave + c) * LDX + 16 * ct + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = oin ? v[e] : 0.0f; const float d = v[e] - o[e]; d2 = fmaf(d, d, d2); n2 = fmaf(o[e], o[e], n2); }
            buf_st_sc1(r_out, oin ? ((unsigned)orow * (unsigned)S + 16u * ct + 4u * g) * 4u : BUF_OFF, v);
        }
        d2 += __shfl_xor(d2, 16, 64); d2 += __shfl_xor(d2, 32, 64);
        n2 += __shfl_xor(n2, 16, 64); n2 += __shfl_xor(n2, 32, 64);
        const int any = (oin && sqrtf(d2) > a.thr * sqrtf(n2)) ? 1 : 0;
        const bool moving = grid_barrier(gb, any, &cont);
        k_done = it + 1;
        if (!a.no_exit && !moving) break;
    }
    if (tid == 0) {
        if (gb.timed_out && a.err) atomicOr(a.err, 1);
        if (blockIdx.x == 0) *a.k_out = gb.timed_out ? -1.0e9f : (float)k_done;
    }
}

template <int SQ>
inline size_t train_small_fwd_lds() {
    constexpr int S = 16 * SQ;
    return sizeof(float) * (64 * (2 * S + 4) + 2 * (2 * S) * (S + 4) + 2 * S + 2 * S + S + 2 * S + 512);
}

// ---- backward ----------------------------------------------------------------------------------------------------------------------------------
struct TrainSmallBwd {
    int N, S, k;                 // k = iterations the forward pass executed
    const int *rowptr_s, *src_s; const float *w_s, *row_scale_s; // adjacency by SOURCE (transposed aggregate); w_s NULL = unit weights
    const float *row_scale;      // [N] scale of the by-destination operator when its entries depend on the destination only (then w_s = NULL)
    const float *states, *agg, *stats;                           // the forward tape
    int in_s, off_agg;
    ConstSegs cs;                // constant input segments (for P = X^T dZ): labels, aggregated labels, aggregated arc labels
    const float *W;              // [in_s][S]
    const float *gamma, *beta; float eps;
    int act;
    const float *G0;             // [N][S] d loss / d states[k] (from the output network)
    float *dxa;                  // [N][S] exchange buffer: d loss / d agg rows of the current iteration
    unsigned long long *bar;
    float *part;                 // [2][n_wg][S + in_s]  (q | S2) partials
    float *partW;                // [n_wg][in_s * S] every workgroup's share of the kernel gradient (summed by the caller)
    float *db, *dgamma, *dbeta;  // [S], [in_s], [in_s] written by workgroup 0 (complete)
    float inv_n;                 // 1 / N
    int *err;
};

template <int SQ, bool HAS_W>
__global__ void __launch_bounds__(TS_NT, 1) k_train_small_bwd(TrainSmallBwd a) {
    constexpr int S = 16 * SQ, LPR = S / 4, NPP = TS_NT / LPR, NPASS = 64 / NPP, IPL = 16 / LPR;
    constexpr int KMAX = 2 * S + 32;      // input columns: state, agg, up to 32 constant columns (in tile order: state | agg | constants)
    constexpr int LDX = KMAX + 4;         // == 4 (mod 32)
    constexpr int LDZ = S + 4, LDW = S + 4;
    constexpr int NKT = KMAX / 16;        // 16-row tiles of P
    constexpr int TPW = (NKT * SQ + 3) / 4;   // P tiles per wave
    extern __shared__ __attribute__((aligned(16))) float ts_smem[];
    float *Xs = ts_smem;                  // [64][LDX]  inputs of the tile: [state_t | agg_t | constants]
    float *Zs = Xs + 64 * LDX;            // [64][LDZ]  dZ
    float *Gs = Zs + 64 * LDZ;            // [64][LDZ]  G (d loss / d state of the tile)
    float *Wr = Gs + 64 * LDZ;            // [KMAX][LDW] kernel rows in tile-column order (state, agg, constants)
    float *vec = Wr + KMAX * LDW;         // q [S] | S2 [KMAX] | S1 [KMAX] | a [KMAX] | c [KMAX] | coefA, coefC, coefB [2 S each] | scratch
    float *q_s = vec, *S2_s = q_s + S, *S1_s = S2_s + KMAX, *a_s = S1_s + KMAX, *c_s = a_s + KMAX, *cfA = c_s + KMAX, *cfC = cfA + 2 * S,
          *cfB = cfC + 2 * S, *red = cfB + 2 * S;       // red: [4][S] / [4][KMAX] scratch
    int *wrow_of = reinterpret_cast<int *>(red + 4 * KMAX);     // [KMAX] weight row (= BatchNorm column) of tile column j, -1 = padding
    __shared__ int cont;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int tile0 = blockIdx.x * 64;
    const bool bn = a.gamma != nullptr;
    const size_t NS = (size_t)a.N * S;
    GridBar gb{a.bar, gridDim.x, 0, 0u, 0u, 0};
    const int Kc = (a.cs.n > 0 ? a.cs.width[0] : 0) + (a.cs.n > 1 ? a.cs.width[1] : 0) + (a.cs.n > 2 ? a.cs.width[2] : 0);
    const int KU = 2 * S + Kc;            // used tile columns

    for (int j = tid; j < KMAX; j += TS_NT) {
        int r = -1;
        if (j < S) r = j;
        else if (j < 2 * S) r = a.off_agg + (j - S);
        else {
            int jj = j - 2 * S, b0 = 0;
#pragma unroll
            for (int s = 0; s < 3; ++s) { if (s < a.cs.n && jj >= b0 && jj < b0 + a.cs.width[s]) r = a.cs.wrow[s] + (jj - b0); if (s < a.cs.n) b0 += a.cs.width[s]; }
        }
        wrow_of[j] = r;
    }
    __syncthreads();
    for (int i = tid; i < KMAX * S; i += TS_NT) {
        const int j = i / S, h = i % S;
        Wr[j * LDW + h] = wrow_of[j] >= 0 ? a.W[(size_t)wrow_of[j] * S + h] : 0.0f;
    }
    // the tile's constant inputs and its first G
    for (int i = tid; i < 64 * 32; i += TS_NT) {
        const int rr = i >> 5, jj = i & 31, n = tile0 + rr;
        float v = 0.0f;
        if (n < a.N && jj < Kc) {
            int b0 = 0;
#pragma unroll
            for (int s = 0; s < 3; ++s) { if (s < a.cs.n && jj >= b0 && jj < b0 + a.cs.width[s]) v = a.cs.ptr[s][(size_t)n * a.cs.ld[s] + (jj - b0)]; if (s < a.cs.n) b0 += a.cs.width[s]; }
        }
        Xs[rr * LDX + 2 * S + jj] = v;
    }
    for (int i = tid; i < 64 * S; i += TS_NT) {
        const int rr = i / S, h = i % S, n = tile0 + rr;
        Gs[rr * LDZ + h] = n < a.N ? a.G0[(size_t)n * S + h] : 0.0f;
    }
    // by-source CSR rows of this lane group's nodes
    const int q = tid / LPR, l4 = tid % LPR;
    int jn[NPASS], beg[NPASS], end[NPASS], ids[NPASS][IPL];
    float wts[NPASS][IPL], scl[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int m = tile0 + p * NPP + q;
        jn[p] = m < a.N ? m : -1;
        beg[p] = end[p] = 0; scl[p] = 1.0f;
        if (jn[p] >= 0) { beg[p] = a.rowptr_s[m]; end[p] = a.rowptr_s[m + 1]; if (a.row_scale_s) scl[p] = a.row_scale_s[m]; }
#pragma unroll
        for (int u = 0; u < IPL; ++u) {
            const int e = beg[p] + u * LPR + l4;
            ids[p][u] = e < end[p] ? a.src_s[e] : 0;
            wts[p][u] = (HAS_W && e < end[p]) ? a.w_s[e] : 0.0f;
        }
    }
    // this wave's tiles of P: (kt, ht) = tile index wave + 4 i  ->  accumulated kernel-gradient share, over all iterations
    f32x4 accW[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) accW[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float db_acc = 0.0f, dg_acc = 0.0f, dbt_acc = 0.0f;              // workgroup 0: thread h < S / thread j < KU
    const int orow = tile0 + 16 * wave + c;
    const bool oin = orow < a.N;
    const float rs = (a.row_scale && oin) ? a.row_scale[orow] : 1.0f;
    const __amdgpu_buffer_rsrc_t r_dxa = buf_rsrc(a.dxa);
    __syncthreads();

    for (int t = a.k - 1; t >= 0; --t) {
        const float *s_t = a.states + (size_t)t * NS, *s_n = a.states + (size_t)(t + 1) * NS, *agg_t = a.agg + (size_t)t * NS;
        const float *stats = a.stats + (size_t)t * 2 * a.in_s;
        // ---- A. the tile's inputs and dZ = G (.) act'(output) into LDS -----------------------------------------------------------------------
        for (int i = tid; i < 64 * LPR; i += TS_NT) {
            const int rr = i / LPR, ch = i % LPR, n = tile0 + rr;
            f32x4 xs = {0.f, 0.f, 0.f, 0.f}, xa = xs, y = xs;
            if (n < a.N) {
                xs = *reinterpret_cast<const f32x4 *>(s_t + (size_t)n * S + 4 * ch);
                xa = *reinterpret_cast<const f32x4 *>(agg_t + (size_t)n * S + 4 * ch);
                y = *reinterpret_cast<const f32x4 *>(s_n + (size_t)n * S + 4 * ch);
            }
            *reinterpret_cast<f32x4 *>(Xs + rr * LDX + 4 * ch) = xs;
            *reinterpret_cast<f32x4 *>(Xs + rr * LDX + S + 4 * ch) = xa;
            f32x4 gz = *reinterpret_cast<const f32x4 *>(Gs + rr * LDZ + 4 * ch);
#pragma unroll
            for (int e = 0; e < 4; ++e) gz[e] *= activate_grad_from_output(a.act, y[e]);
            *reinterpret_cast<f32x4 *>(Zs + rr * LDZ + 4 * ch) = gz;
        }
        if (tid < KU) {                                            // BatchNorm scale / shift of this iteration per tile column
            float ak = 1.0f, ck = 0.0f;
            if (bn) { const int k = wrow_of[tid]; ak = a.gamma[k] / sqrtf(stats[a.in_s + k] + a.eps); ck = a.beta[k] - stats[k] * ak; }
            a_s[tid] = ak; c_s[tid] = ck;
        }
        __syncthreads();
        // ---- B. P_wg = X^T dZ on the matrix cores (kept local), q and S2 partials -----------------------------------------------------------
        f32x4 P[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            P[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int ti = wave + 4 * i;
            if (ti < NKT * SQ) {
                const int kt = ti / SQ, ht = ti % SQ;
                const float *xa = Xs + g * LDX + 16 * kt + c, *za = Zs + g * LDZ + 16 * ht + c;
#pragma unroll 4
                for (int ms = 0; ms < 16; ++ms)
                    P[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[4 * ms * LDX], za[4 * ms * LDZ], P[i], 0, 0, 0);
            }
        }
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
        // P[i][reg] = P_wg[16 kt + 4 g + reg][16 ht + c]; S2 partial: sum over h of W[k][h] P[k][h] -> LDS scratch (per h-tile, per c lane)
        if (tid < S) {
            float t2 = 0.0f;
            for (int rr = 0; rr < 64; ++rr) t2 += Zs[rr * LDZ + tid];
            q_s[tid] = t2;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int ti = wave + 4 * i;
            if (ti < NKT * SQ) {
                const int kt = ti / SQ, ht = ti % SQ;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int j = 16 * kt + 4 * g + reg;
                    float v = Wr[j * LDW + 16 * ht + c] * P[i][reg];
                    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);   // over the 16 columns
                    if (c == 0) red[ht * KMAX + j] = v;
                }
            }
        }
        __syncthreads();
        if (tid < S + KMAX) {
            float v = 0.0f;
            if (tid < S) v = q_s[tid];
            else for (int ht = 0; ht < SQ; ++ht) v += red[ht * KMAX + tid - S];            // in column-tile order
            const __amdgpu_buffer_rsrc_t r_p = buf_rsrc(a.part + ((size_t)(t & 1) * gridDim.x + blockIdx.x) * (S + KMAX));
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r_p, tid * 4, 0, 16);
        }
        grid_barrier(gb, 0, &cont);
        if (tid < S + KMAX) {
            const __amdgpu_buffer_rsrc_t r_p = buf_rsrc(a.part + (size_t)(t & 1) * gridDim.x * (S + KMAX));
            double tt = 0.0;
            for (unsigned wg = 0; wg < gridDim.x; ++wg)
                tt += (double)__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_p, (int)((wg * (S + KMAX) + tid) * 4), 0, 16));
            if (tid < S) q_s[tid] = (float)tt; else S2_s[tid - S] = (float)tt;
        }
        __syncthreads();
        if (tid < KU) {                                            // S1_k = (W q)_k
            float t1 = 0.0f;
            for (int h = 0; h < S; ++h) t1 = fmaf(Wr[tid * LDW + h], q_s[h], t1);
            S1_s[tid] = t1;
        }
        __syncthreads();
        // ---- C. parameter-gradient shares (registers), BatchNorm input-gradient coefficients -------------------------------------------------
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int ti = wave + 4 * i;
            if (ti < NKT * SQ) {
                const int kt = ti / SQ, ht = ti % SQ;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int j = 16 * kt + 4 * g + reg;
                    float v = j < KU ? a_s[j] * P[i][reg] : 0.0f;
                    if (blockIdx.x == 0 && j < KU) v = fmaf(c_s[j], q_s[16 * ht + c], v);      // the c q^T term once (q is the total)
                    accW[i][reg] += v;
                }
            }
        }
        if (blockIdx.x == 0) {
            if (tid < S) db_acc += q_s[tid];
            if (bn && tid < KU) {
                const int k = wrow_of[tid];
                const float rstd = 1.0f / sqrtf(stats[a.in_s + k] + a.eps);
                dg_acc += rstd * (S2_s[tid] - stats[k] * S1_s[tid]); dbt_acc += S1_s[tid];
            }
        }
        if (tid < 2 * S) {
            float Ac = 1.0f, Cc = 0.0f, Bc = 0.0f;
            if (bn) {
                const int k = wrow_of[tid];
                const float rstd = 1.0f / sqrtf(stats[a.in_s + k] + a.eps), mu = stats[k];
                const float m1 = S1_s[tid] * a.inv_n, m2 = rstd * (S2_s[tid] - mu * S1_s[tid]) * a.inv_n;
                Ac = a.gamma[k] * rstd; Cc = -Ac * rstd * m2; Bc = -Ac * m1 - Cc * mu;
            }
            cfA[tid] = Ac; cfC[tid] = Cc; cfB[tid] = Bc;
        }
        __syncthreads();
        // ---- D. dx = BN-gradient(dZ . W^T) on the matrix cores (operands swapped: row-major); agg half -> exchange buffer, state half -> Gs ----
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 acc[SQ];
#pragma unroll
            for (int u = 0; u < SQ; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float *zrow = Zs + (16 * wave + c) * LDZ + 4 * g;
#pragma unroll 2
            for (int qq = 0; qq < S / 16; ++qq) {                  // k-step (qq, e) = dZ column 16 qq + 4 g + e: both operands are 16-byte reads
                const f32x4 zv = *reinterpret_cast<const f32x4 *>(zrow + 16 * qq);
#pragma unroll
                for (int u = 0; u < SQ; ++u) {
                    const f32x4 wv = *reinterpret_cast<const f32x4 *>(Wr + (half * S + 16 * u + c) * LDW + 16 * qq + 4 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[e], zv[e], acc[u], 0, 0, 0);
                }
            }
            asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
            for (int u = 0; u < SQ; ++u) {
                const int j0 = half * S + 16 * u + 4 * g;
                f32x4 v = acc[u];
                if (bn) {
                    const f32x4 x = *reinterpret_cast<const f32x4 *>(Xs + (16 * wave + c) * LDX + j0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(cfA[j0 + e], v[e], fmaf(cfC[j0 + e], x[e], cfB[j0 + e]));
                }
                if (half == 0) {
                    *reinterpret_cast<f32x4 *>(Gs + (16 * wave + c) * LDZ + 16 * u + 4 * g) = v;      // dx_state: the own part of the next G
                } else {
                    v *= rs;
                    buf_st_sc1(r_dxa, oin ? ((unsigned)orow * (unsigned)S + 16u * u + 4u * g) * 4u : BUF_OFF, v);
                }
            }
        }
        grid_barrier(gb, 0, &cont);
        // ---- E. G = dx_state + Adj . dx_agg: gather by source -----------------------------------------------------------------------------------
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            int idc[IPL]; float wsc[IPL];
#pragma unroll
            for (int u = 0; u < IPL; ++u) { idc[u] = ids[p][u]; wsc[u] = wts[p][u]; }
            int rem = end[p] - beg[p], eb = beg[p];
#pragma unroll 1
            while (true) {
                f32x4 v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const unsigned off = (unsigned)__shfl(idc[i / LPR], i % LPR, LPR) * (unsigned)(S * 4) + 16u * l4;
                    v[i] = buf_ld_sc1(r_dxa, i < rem ? off : BUF_OFF);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (HAS_W) acc += __shfl(wsc[i / LPR], i % LPR, LPR) * v[i];
                    else acc += v[i];
                }
                rem -= 16; eb += 16;
                if (!__any(rem > 0)) break;
#pragma unroll
                for (int u = 0; u < IPL; ++u) {
                    const int e = eb + u * LPR + l4;
                    idc[u] = e < end[p] ? a.src_s[e] : 0;
                    wsc[u] = (HAS_W && e < end[p]) ? a.w_s[e] : 0.0f;
                }
            }
            float *gr = Gs + (p * NPP + q) * LDZ + 4 * l4;
            f32x4 gv = *reinterpret_cast<const f32x4 *>(gr);
            gv += acc * scl[p];
            if (jn[p] < 0) gv = (f32x4){0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4 *>(gr) = gv;
        }
        __syncthreads();
    }
    // ---- the kernel-gradient share of this workgroup; bias / BatchNorm gradients from workgroup 0 ----------------------------------------------
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int ti = wave + 4 * i;
        if (ti < NKT * SQ) {
            const int kt = ti / SQ, ht = ti % SQ;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int j = 16 * kt + 4 * g + reg;
                if (j < KU) a.partW[(size_t)blockIdx.x * a.in_s * S + (size_t)wrow_of[j] * S + 16 * ht + c] = accW[i][reg];
            }
        }
    }
    if (blockIdx.x == 0) {
        if (tid < S) a.db[tid] = db_acc;
        if (bn && tid < KU) { a.dgamma[wrow_of[tid]] = dg_acc; a.dbeta[wrow_of[tid]] = dbt_acc; }
    }
    if (tid == 0 && gb.timed_out) { if (a.err) atomicOr(a.err, 1); a.db[0] = __builtin_nanf(""); }      // loud: the gradients are invalid
}

template <int SQ>
inline size_t train_small_bwd_lds() {
    constexpr int S = 16 * SQ, KMAX = 2 * S + 32;
    return sizeof(float) * (64 * (KMAX + 4) + 2 * 64 * (S + 4) + KMAX * (S + 4) + (S + 4 * KMAX + 6 * S) + 4 * KMAX + KMAX);
}

}  // namespace gnn
