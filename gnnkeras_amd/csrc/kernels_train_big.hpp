// Training at large M (reference GNN/Models/GNN.py:277-306 on graphs of 10^5 .. 10^7 nodes): the per-iteration kernels of
// gnn_train_step when the step is bandwidth work, not launch count.
//
// Round 2 ran a training iteration on the general kernels: aggregate (401 us at C4 size) + four column-statistics passes
// (4 x 115 us) + the staged dense kernel k_segdense<1> (378 us, 1.5-2.4 TB/s: its waves are parked 67 % of the time at the
// workgroup barriers of its LDS staging) forward; activation gradient + weight gradient + two more k_segdense calls + the
// BatchNorm input gradient + the transposed aggregate backward.  Here:
//
//   * k_aggregate_stats   - the neighbour sum of an iteration (kept on the tape) that also leaves the per-column sum and sum of
//                           squares of what it wrote: the training-mode BatchNormalization statistics of the aggregated-state
//                           columns (reference MLP.py:67-70) cost no extra pass;
//   * k_train_fwd<SQ,NCT> - the first Dense of the state network over [state | agg | constant inputs] with the batch statistics
//                           folded into its weights: rows go from global memory STRAIGHT into MFMA A-fragment registers (16-byte
//                           loads, permuted k order, as kernel_state_wide.hpp / kernel_state_lds.hpp do), the folded weights sit
//                           in LDS in the matching order - no staging tile, no barrier inside the row loop; activation, the
//                           convergence predicate (GNN.py:196-212) and the column statistics of the NEW state (next iteration's
//                           BatchNorm input) in the epilogue;
//   * k_train_bwd_dx<HQ,NCT> - d loss / d [state | agg] = BN-input-gradient(dZ . W1^T): the same row-streaming MFMA loop, the
//                           BatchNormalization input gradient (three coefficients per column) and the row scale of 'average'
//                           aggregation in the epilogue - replaces two dense launches and the BN-gradient pass.
// Round 4: the forward Dense and dZ . W^T run on the bf16 matrix cores with every f32 operand split into three bf16 terms
// (k_train_fwd_b6, k_train_bwd_dx_b6: f32-chain accuracy, not its bits; GNN_TRAIN_BF16X6=0 selects the f32 kernels), the weight
// gradient on v_mfma_f32_32x32x2_f32 (k_train_wgrad32).  The f32-input MFMA kernels below stay as the exact path and for S = 16.
// Exact float32 on v_mfma_f32_16x16x4_f32 like every other dense kernel here (k_train_fwd, k_train_bwd_dx, k_train_wgrad); deterministic (fixed tile -> wave assignment,
// per-workgroup partial statistics summed in workgroup order by k_stats_finish).
#pragma once
#include "bf16_split.hpp"
#include <hip/hip_runtime.h>
#include "kernels_general.hpp"
#include "kernel_state_fused4.hpp"      // activate4
#include <type_traits>
#include "kernel_state_lds.hpp"         // row16_sum_to_lane15
#include "kernels_train.hpp"            // activate_grad_from_output
#include "buffer_ops.hpp"

namespace gnn {

#ifndef TB_STREAM_AUX
#define TB_STREAM_AUX 0                  // cache-policy bits of the row loads of k_train_fwd_b6 / k_train_bwd_dx_b6 (experiment: scripts/micro/rowgemm_bench.hip)
#endif
#ifndef TB_RING_AUX
#define TB_RING_AUX 2                    // cache-policy bits of the ring's LDS-DMA loads: 2 = nt.  k_train_wgrad_b6 at 1 M rows: 181.5 us default policy, 172.4 nt (183.5 / 177.6
                                         // with sc0 / sc0 + nt); inside the step 188 -> 171 us.  (nt on the register loads of k_train_fwd_b6 / k_train_bwd_dx_b6: 199 -> 202, 196 -> 207.)
#endif
#ifndef TB_ABL
#define TB_ABL 0                         // ablation switches of scripts/micro/rowgemm_bench.hip (1 no MFMAs, 2 no predicate, 4 no stores, 8 no statistics); 0 in the library
#endif
// Results of the last MFMAs of a tile are consumed behind a branch (the activation switch, `if (gamma)`): hipcc 7.2's hazard
// recognizer does not carry the "XDL write -> VALU / VMEM read" wait states (up to 18 for this shape) across the block boundary -
// k_train_bwd_dx<1, 2> without BatchNormalization read its accumulators two instructions after the MFMA that wrote them and
// returned stale registers (scripts/micro/rowgemm_check.hip reproduces it).  Explicit wait states after the MFMA loop.
#define TB_MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 3" ::: "memory")
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int TB_WAVES = 8;              // waves per workgroup (512 threads; launch bound 4 waves per SIMD = 2 workgroups per CU at <= 128 VGPRs)

// ---- neighbour sum + column statistics of the result -------------------------------------------------------------------------------
// k_aggregate_vec (kernels_general.hpp) with one addition: every thread keeps the sum and the sum of squares of the float4 column
// chunk it writes; a workgroup folds its threads' partials in a fixed order into stat_part[blockIdx.x][2 F] (sums, then squares).
template <int LPR, bool HAS_W, bool BUF = false>
__global__ void __launch_bounds__(256, 8)              // (8 waves per SIMD = 64 VGPRs: with the shift in registers hipcc took 66, and the gather lost 18 %)
k_aggregate_stats(const int *gate, int n_dst, const int *__restrict__ rowptr, const int *__restrict__ src,
                  const float *__restrict__ w, const float *__restrict__ row_scale, const float *__restrict__ X, int ldx,
                  float *__restrict__ out, int ldo, float *__restrict__ stat_part, const float *__restrict__ shift) {
    if (gate_closed(gate)) return;
    __shared__ f32x4 red[2][256];
    const int l4 = threadIdx.x % LPR;
    constexpr int groups = 256 / LPR;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    // one-pass moments around `shift` (a value near the column mean: the previous iteration's mean): sum (x - c), sum (x - c)^2 -
    // E[x^2] - mean^2 on raw sums loses (1 + 2 mean^2 / var) digits, and a neighbour AVERAGE of relu / sigmoid states has var << mean^2
    const f32x4 sh = shift ? *reinterpret_cast<const f32x4 *>(shift + 4 * l4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int j = blockIdx.x * groups + threadIdx.x / LPR; j < n_dst; j += gridDim.x * groups) {
        const int beg = rowptr[j], end = rowptr[j + 1];
        f32x4 acc = gather_sum8<HAS_W, BUF>(beg, end, src, w, X, ldx, 4 * l4);                   // summed in arc order
        if (row_scale) acc *= row_scale[j];
        *reinterpret_cast<f32x4 *>(out + (size_t)j * ldo + 4 * l4) = acc;
        acc -= sh;
        s1 += acc; s2 += acc * acc;
    }
    red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < 2 * LPR) {                                // one thread per (sum | square, column chunk): groups in order
        const int which = threadIdx.x / LPR, c4 = threadIdx.x % LPR;
        f32x4 t = red[which][c4];
        for (int gq = 1; gq < groups; ++gq) t += red[which][gq * LPR + c4];
        *reinterpret_cast<f32x4 *>(stat_part + (size_t)blockIdx.x * (8 * LPR) + which * (4 * LPR) + 4 * c4) = t;
    }
}

// One pass over a row-major matrix [M][4 LPR] (leading dimension ld): per-workgroup column sums and sums of squares in the partial layout
// of k_aggregate_stats (the statistics of state_0 and of the packed constants line: two passes per segment before, 12 launches per step).
template <int LPR>
__global__ void __launch_bounds__(256)
k_rows_stats(const int *gate, int M, const float *__restrict__ X, int ld, float *__restrict__ stat_part) {
    if (gate_closed(gate)) return;
    __shared__ f32x4 red[2][256];
    const int l4 = threadIdx.x % LPR;
    constexpr int groups = 256 / LPR;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 sh = *reinterpret_cast<const f32x4 *>(X + 4 * l4);          // moments around ROW 0 (k_stats_finish adds it back: pass X as its shift)
    for (int j = blockIdx.x * groups + threadIdx.x / LPR; j < M; j += gridDim.x * groups * 4) {       // four rows in flight
        f32x4 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = j + u * gridDim.x * groups;
            x[u] = r < M ? *reinterpret_cast<const f32x4 *>(X + (size_t)r * ld + 4 * l4) - sh : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { s1 += x[u]; s2 += x[u] * x[u]; }
    }
    red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < 2 * LPR) {
        const int which = threadIdx.x / LPR, c4 = threadIdx.x % LPR;
        f32x4 t = red[which][c4];
        for (int gq = 1; gq < groups; ++gq) t += red[which][gq * LPR + c4];
        *reinterpret_cast<f32x4 *>(stat_part + (size_t)blockIdx.x * (8 * LPR) + which * (4 * LPR) + 4 * c4) = t;
    }
}

// mean[c] = sum / M, var[c] = max(sumsq / M - mean^2, 0) from n_part partials of [2 F] (sums | squares).  One 256-thread workgroup
// per column: thread i sums partials i, i + 256, .. in order, then a fixed LDS tree - deterministic.  (One-pass moments: the columns
// are activations / their neighbour averages, |mean| and sigma of the same order, the tail in double; BatchNormalization adds
// eps = 1e-3 to the variance before the square root.)
__device__ __forceinline__ void stats_finish_body(const float *__restrict__ part, int n_part, int F, float inv_m, float *__restrict__ mean, float *__restrict__ var,
                                                  const float *__restrict__ shift) {
    __shared__ double sh[2][256];
    const int c = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int p = threadIdx.x; p < n_part; p += 256) { s += (double)part[(size_t)p * 2 * F + c]; q += (double)part[(size_t)p * 2 * F + F + c]; }
    sh[0][threadIdx.x] = s; sh[1][threadIdx.x] = q;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (threadIdx.x < off) { sh[0][threadIdx.x] += sh[0][threadIdx.x + off]; sh[1][threadIdx.x] += sh[1][threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mu = sh[0][0] * (double)inv_m;
        mean[c] = (float)(mu + (shift ? (double)shift[c] : 0.0));
        var[c] = (float)fmax(sh[1][0] * (double)inv_m - mu * mu, 0.0);
    }
}
__global__ void __launch_bounds__(256)
k_stats_finish(const int *gate, const float *__restrict__ part, int n_part, int F, float inv_m, float *__restrict__ mean, float *__restrict__ var,
               const float *__restrict__ shift) {      // shift[c]: what the producer subtracted from column c (NULL: nothing)
    if (gate_closed(gate)) return;
    stats_finish_body(part, n_part, F, inv_m, mean, var, shift);
}
// ... of up to 2 statistics per node type in one launch (heterogeneous models, train_composite_big.hpp): blockIdx.y = the job
struct StatsFinishT { const float *part; int n_part; float inv_m; float *mean, *var; const float *shift; };
struct StatsFinishJobs { StatsFinishT t[2 * GNN_MAX_TYPES]; };
__global__ void __launch_bounds__(256) k_stats_finish_jobs(const int *gate, StatsFinishJobs m, int F) {
    if (gate_closed(gate)) return;
    StatsFinishT a = m.t[0];
#pragma unroll
    for (int t = 1; t < 2 * GNN_MAX_TYPES; ++t) if ((int)blockIdx.y == t) a = m.t[t];
    if (!a.part) return;
    stats_finish_body(a.part, a.n_part, F, a.inv_m, a.mean, a.var, a.shift);
}

// ---- forward: first Dense of the state network in training mode ----------------------------------------------------------------------
struct ConstCols { int width[3], wrow[3], n; };       // constant input columns of xc: segment s covers `width[s]` columns, weight rows wrow[s]..

// Heterogeneous models (train_composite_big.hpp): ONE launch of a row-streaming kernel for the rows of every node type - the workgroups
// [blk_begin[t], blk_begin[t + 1]) run the kernel's body on type t's arguments (its row range, weights, statistics, partials) as workgroups
// 0 .. of a launch of their own.  A third of the launches, and a workgroup's set-up (weights into LDS, the epilogue's exchange) is paid for
// three times the rows.  Static selection (a runtime-indexed kernel-argument array would be copied to scratch memory).
template <typename A> struct TypeLaunch { A t[GNN_MAX_TYPES]; int blk_begin[GNN_MAX_TYPES + 1]; int n; };
template <typename A> __device__ __forceinline__ A select_type(const TypeLaunch<A> &m, int &bid, int &nblk) {
    int ty = 0;
#pragma unroll
    for (int t = 1; t < GNN_MAX_TYPES; ++t) if (t < m.n && (int)blockIdx.x >= m.blk_begin[t]) ty = t;
    A a = m.t[0];
    int b0 = m.blk_begin[0], b1 = m.blk_begin[1];
#pragma unroll
    for (int t = 1; t < GNN_MAX_TYPES; ++t) if (ty == t) { a = m.t[t]; b0 = m.blk_begin[t]; b1 = m.blk_begin[t + 1]; }
    bid = (int)blockIdx.x - b0; nblk = b1 - b0;
    return a;
}

struct TrainFwdArgs {
    const int *gate;
    int M;
    const float *state; int ld_state;     // [M, 16 SQ]
    const float *agg; int ld_agg;         // [M, 16 SQ]
    const float *xc;                      // [M, 32] constant inputs (k_pack_xc layout: segments, then a 1, then zeros) or NULL
    const float *Wf, *bf; int H;          // folded first layer [in_dim][H] (BatchNorm of THIS iteration folded in), bias [H]
    int wrow_state, wrow_agg;
    ConstCols cs;
    int act;
    float *Y; int ldy;                    // [M, H]
    float thr; int *pred_flag; float *pred_k; float pred_kval;      // predicate of Y against `state` (H == 16 SQ), optional
    float *stat_part;                     // [gridDim.x][2 * 16 NCT] column sums / squares of Y - stat_shift, optional
    const float *stat_shift;              // [H] or NULL: subtracted before the sums (the input state's column means: see k_aggregate_stats)
    const float *in_mean;                 // [in_dim] (by weight row) or NULL: subtracted from the input rows as they arrive; (Wf, bf) are then the
                                          // fold WITHOUT the -mean a term (FoldJob::centred)
    const float *addend; int ld_add;      // k_train_fwd_b6<.., ADD = true> only: [M, H] added to the pre-activations INSTEAD of the constants line's
                                          // product - the constant inputs' share of the first layer, which does not change between the iterations
                                          // of a step (their batch statistics do not: train_composite_big.hpp; constant inputs wider than the line)
};

// mean of virtual input column k of [state | agg | constants line] (chunks of 16), 0 where there is none
__device__ __forceinline__ float fwd_in_mean(const TrainFwdArgs &a, int SQ, int k) {
    if (!a.in_mean) return 0.0f;
    if (k < 16 * SQ) return a.in_mean[a.wrow_state + k];
    if (k < 32 * SQ) return a.in_mean[a.wrow_agg + (k - 16 * SQ)];
    int jj = k - 32 * SQ, beg = 0, row = -1;
#pragma unroll
    for (int sg = 0; sg < 3; ++sg) {
        if (sg < a.cs.n && jj >= beg && jj < beg + a.cs.width[sg]) row = a.cs.wrow[sg] + (jj - beg);
        if (sg < a.cs.n) beg += a.cs.width[sg];
    }
    return (a.xc && row >= 0) ? a.in_mean[row] : 0.0f;
}

template <int NCT> struct BFrag;
template <> struct BFrag<1> { float v[1]; __device__ __forceinline__ void load(const float *p) { v[0] = *p; } };
template <> struct BFrag<2> { float v[2]; __device__ __forceinline__ void load(const float *p) { const float2 t = *reinterpret_cast<const float2 *>(p); v[0] = t.x; v[1] = t.y; } };
template <> struct BFrag<3> { float v[3]; __device__ __forceinline__ void load(const float *p) { v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; } };
template <> struct BFrag<4> { float v[4]; __device__ __forceinline__ void load(const float *p) { const f32x4 t = *reinterpret_cast<const f32x4 *>(p); v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; } };
template <> struct BFrag<8> { float v[8]; __device__ __forceinline__ void load(const float *p) {
    const f32x4 t = *reinterpret_cast<const f32x4 *>(p), u = *reinterpret_cast<const f32x4 *>(p + 4);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; v[4] = u[0]; v[5] = u[1]; v[6] = u[2]; v[7] = u[3]; } };

// The matrix core is fed TRANSPOSED: the weights are the A operand (A[i][k] = W[k][16 ct + i]), the input rows the B operand
// (B[k][j] = X[row j][k]) - for v_mfma_f32_16x16x4_f32 both operands have the same register layout (lane l supplies element
// (l % 16, l / 16)), so the 16-byte row chunks a lane loaded serve as they are - and the result D[i][j] = Y[row j][16 ct + i] leaves
// lane (c, g) holding Y[row c][16 ct + 4 g .. + 3]: four CONSECUTIVE columns of its own row.  Output, old state (the A chunks of the
// state segment) and stores all share the row-major 16-byte layout: no transposition, no 4-byte accesses, no loads in the epilogue.
#ifndef TB_PREFETCH
#define TB_PREFETCH 0                  // 1 (experiment): the next tile's rows are requested before this tile's MFMAs.  162 VGPRs = 3 waves per SIMD
                                       // instead of 4: SLOWER, 227 -> 247 us per 1 M rows (scripts/micro/rowgemm_bench.hip): the waves hide more than the prefetch
#endif
#ifndef TB_FWD_MIN_WAVES
#define TB_FWD_MIN_WAVES 4             // waves per SIMD the register allocation is held to (experiment knob of scripts/micro/rowgemm_bench.hip)
#endif
template <int SQ, int NCT>
__global__ void __launch_bounds__(64 * TB_WAVES, TB_PREFETCH ? 2 : TB_FWD_MIN_WAVES) k_train_fwd(TrainFwdArgs a) {
    if (gate_closed(a.gate)) return;
    constexpr int NQ = 2 * SQ + 2;                    // 16-column chunks of an input row: state, agg, constant inputs (32 columns)
    constexpr int HP = 16 * NCT;
    extern __shared__ __attribute__((aligned(16))) float tb_smem[];
    float *Wl = tb_smem;                               // [4 NQ k-steps][4 g][16 c][NCT]
    float *bias_l = tb_smem + 16 * NQ * HP;            // [HP]
    float *red = bias_l + HP;                          // [TB_WAVES][2 HP] statistics hand-over
    float *shift_l = red + TB_WAVES * 2 * HP;          // [HP] what the statistics are taken around
    float *mean_l = shift_l + HP;                      // [16 NQ] the input columns' means (a.in_mean), 0 where there is none
    __shared__ int any_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    if (tid == 0) any_s = 0;
    for (int h = tid; h < HP; h += 64 * TB_WAVES) shift_l[h] = (a.stat_shift && h < a.H) ? a.stat_shift[h] : 0.0f;
    for (int k = tid; k < 16 * NQ; k += 64 * TB_WAVES) mean_l[k] = fwd_in_mean(a, SQ, k);
    // ---- folded weights into LDS in fragment order: k-step (q, e) of lane group g multiplies virtual column 16 q + 4 g + e ------
    for (int i = tid; i < 16 * NQ * HP; i += 64 * TB_WAVES) {
        const int k = i / HP, h = i % HP;
        int row = -1;
        if (k < 16 * SQ) row = a.wrow_state + k;
        else if (k < 32 * SQ) row = a.wrow_agg + (k - 16 * SQ);
        else if (a.xc) {
            int j = k - 32 * SQ, beg = 0;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                if (s < a.cs.n && j >= beg && j < beg + a.cs.width[s]) row = a.cs.wrow[s] + (j - beg);
                if (s < a.cs.n) beg += a.cs.width[s];
            }
        }
        const float v = (row >= 0 && h < a.H) ? a.Wf[(size_t)row * a.H + h] : 0.0f;
        const int q = k >> 4, rem = k & 15, gg = rem >> 2, e = rem & 3;
        Wl[(((4 * q + e) * 4 + gg) * 16 + (h & 15)) * NCT + (h >> 4)] = v;
    }
    for (int h = tid; h < HP; h += 64 * TB_WAVES) bias_l[h] = h < a.H ? a.bf[h] : 0.0f;
    __syncthreads();

    const __amdgpu_buffer_rsrc_t r_s = buf_rsrc(a.state), r_a = buf_rsrc(a.agg), r_x = buf_rsrc(a.xc), r_y = buf_rsrc(a.Y);
    const int n_tiles = (a.M + 15) >> 4;
    f32x4 cs1[NCT], cs2[NCT];                           // column sums / squares of this lane's four columns per tile
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) { cs1[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; cs2[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    int any = 0;
    auto fetch = [&](int t, f32x4 (&A_)[NQ]) {          // this lane's 16-byte pieces of row 16 t + c: state, agg, constants line
        const int row_ = 16 * t + c;
        const bool in_ = t < n_tiles && row_ < a.M;
#pragma unroll
        for (int q = 0; q < SQ; ++q) {
            A_[q] = buf_ld_f32x4(r_s, in_ ? ((unsigned)row_ * (unsigned)a.ld_state + 16u * q + 4u * g) * 4u : BUF_OFF);
            A_[SQ + q] = buf_ld_f32x4(r_a, in_ ? ((unsigned)row_ * (unsigned)a.ld_agg + 16u * q + 4u * g) * 4u : BUF_OFF);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) A_[2 * SQ + q] = buf_ld_f32x4(r_x, in_ ? ((unsigned)row_ * 32u + 16u * q + 4u * g) * 4u : BUF_OFF);
    };
    const int t_step = gridDim.x * TB_WAVES;
    f32x4 A[NQ];
#if TB_PREFETCH
    fetch(blockIdx.x * TB_WAVES + wave, A);
#endif
#pragma unroll 1
    for (int t = blockIdx.x * TB_WAVES + wave; t < n_tiles; t += t_step) {
        const int row = 16 * t + c;                     // this lane's row: input chunks, output chunks, old state
        const bool in = row < a.M;
#if TB_PREFETCH
        f32x4 An[NQ];
        fetch(t + t_step, An);                          // (in flight while this tile multiplies)
#else
        fetch(t, A);
#endif
        f32x4 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = *reinterpret_cast<const f32x4 *>(bias_l + 16 * ct + 4 * g);
#if (TB_ABL & 1)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q % NCT] += A[q];
#else
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const f32x4 xq = A[q] - *reinterpret_cast<const f32x4 *>(mean_l + 16 * q + 4 * g);      // (centred: see TrainFwdArgs::in_mean)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                BFrag<NCT> w;
                w.load(Wl + (((4 * q + e) * 4 + g) * 16 + c) * NCT);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.v[ct], xq[e], acc[ct], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);         // keep the fragment reads of later chunks from being hoisted (register budget)
        }
        TB_MFMA_DRAIN();
#endif
        // ---- epilogue: acc[ct] = Y[row][16 ct + 4 g .. + 3] ----------------------------------------------------------------------------
        float d2 = 0.0f, n2 = 0.0f;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            f32x4 v = acc[ct];
            activate4(a.act, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (in && 16 * ct + 4 * g + e < a.H) ? v[e] : 0.0f;
            const u32x4 bits = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
#if (TB_ABL & 4)
            __builtin_amdgcn_raw_buffer_store_b128(bits, r_y, (in && v[0] == 1.2345e30f) ? (int)(((unsigned)row * (unsigned)a.ldy + 16u * ct + 4u * g) * 4u) : (int)BUF_OFF, 0, 0);
#else
            __builtin_amdgcn_raw_buffer_store_b128(bits, r_y, (in && 16 * ct + 4 * g < a.H) ? (int)(((unsigned)row * (unsigned)a.ldy + 16u * ct + 4u * g) * 4u) : (int)BUF_OFF, 0, 0);
#endif
#if !(TB_ABL & 8)
            {
                const f32x4 sh = *reinterpret_cast<const f32x4 *>(shift_l + 16 * ct + 4 * g);
                f32x4 dv;
#pragma unroll
                for (int e = 0; e < 4; ++e) dv[e] = in ? v[e] - sh[e] : 0.0f;
                cs1[ct] += dv; cs2[ct] += dv * dv;
            }
#endif
            if (a.pred_flag && !(TB_ABL & 2) && ct < SQ) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float o = A[ct][e], d = v[e] - o; d2 = fmaf(d, d, d2); n2 = fmaf(o, o, n2); }
            }
        }
        if (a.pred_flag && !(TB_ABL & 2)) {             // the row's four lane groups
            d2 += __shfl_xor(d2, 16, 64); d2 += __shfl_xor(d2, 32, 64);
            n2 += __shfl_xor(n2, 16, 64); n2 += __shfl_xor(n2, 32, 64);
            if (in && sqrtf(d2) > a.thr * sqrtf(n2)) any = 1;
        }
#if TB_PREFETCH
#pragma unroll
        for (int q = 0; q < NQ; ++q) A[q] = An[q];
#endif
    }
    // ---- predicate flag, k, statistics partial of this workgroup ---------------------------------------------------------------------
    if (a.pred_flag && __any(any) && lane == 0) any_s = 1;             // benign race: every writer stores 1
    if (a.stat_part) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) {                                 // fold the 16 rows of the tile layout (lanes c = 0 .. 15 of a group)
                const float s1 = row16_sum_to_lane15(cs1[ct][e]), s2 = row16_sum_to_lane15(cs2[ct][e]);
                if (c == 15) { red[wave * 2 * HP + 16 * ct + 4 * g + e] = s1; red[wave * 2 * HP + HP + 16 * ct + 4 * g + e] = s2; }
            }
    }
    __syncthreads();
    if (a.stat_part && tid < 2 * HP) {
        float t = 0.0f;
        for (int w = 0; w < TB_WAVES; ++w) t += red[w * 2 * HP + tid];    // waves in order
        a.stat_part[(size_t)blockIdx.x * 2 * HP + tid] = t;
    }
    if (a.pred_flag && tid == 0) {
        if (any_s) atomicOr(a.pred_flag, 1);
        if (blockIdx.x == 0 && a.pred_k) *a.pred_k = a.pred_kval;
    }
}

#ifndef TB_STAMP_BLOCK
#define TB_STAMP_BLOCK 0
#endif
#ifdef TB_STAMPS                        // debug build of scripts/micro: clock stamps of one wave (block 0, wave 0), 4 per trip
__device__ unsigned long long g_tb_stamps[4 * 64];
__device__ unsigned long long g_tb_blocks[2 * 4096];      // s_memrealtime (100 MHz, one base for the whole device) at the start / end of every workgroup
__device__ unsigned g_tb_hwid[2 * 4096];                  // HW_ID and XCC_ID of wave 0 of every workgroup
#define TB_BLOCK_TIME(i_) do { if (threadIdx.x == 0 && blockIdx.x < 4096) { g_tb_blocks[2 * blockIdx.x + (i_)] = __builtin_amdgcn_s_memrealtime(); \
    g_tb_hwid[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)); g_tb_hwid[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)); } } while (0)
#define TB_STAMP(i_) do { if (blockIdx.x == TB_STAMP_BLOCK && wave == 0 && trip_ < 60) { const unsigned long long c_ = __builtin_amdgcn_s_memtime(); if (lane == 0) g_tb_stamps[4 * trip_ + (i_)] = c_; } } while (0)
#define TB_STAMP_DEP(i_, r_) do { asm volatile("" : "+v"(r_)); TB_STAMP(i_); } while (0)
#define TB_MARK(i_) do { if (blockIdx.x == TB_STAMP_BLOCK && threadIdx.x == 0) g_tb_stamps[240 + (i_)] = __builtin_amdgcn_s_memtime(); } while (0)      // kernel phases
#define TB_TRIP_END() (++trip_)
#else
#define TB_MARK(i_) do {} while (0)
#define TB_BLOCK_TIME(i_) do {} while (0)
#define TB_STAMP(i_) do {} while (0)
#define TB_STAMP_DEP(i_, r_) do {} while (0)
#define TB_TRIP_END() do {} while (0)
#endif

// the activation as a template parameter: no branch inside an instruction stream that is to be interleaved (a basic-block boundary is
// a scheduling boundary for hipcc)
template <int ACT> __device__ __forceinline__ float activate1(float v) {
    if (ACT == GNN_ACT_RELU) return fmaxf(v, 0.0f);
    // (selu / elu as a sum of their two arms - one of them is an exact zero - so that hipcc cannot put the exponential behind a branch)
    if (ACT == GNN_ACT_SELU) return 1.0507009873554805f * fmaxf(v, 0.0f) + (1.0507009873554805f * 1.6732632423543772f) * (expf(fminf(v, 0.0f)) - 1.0f);
    if (ACT == GNN_ACT_TANH) return tanhf(v);
    if (ACT == GNN_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    if (ACT == GNN_ACT_ELU) return fmaxf(v, 0.0f) + (expf(fminf(v, 0.0f)) - 1.0f);
    if (ACT == GNN_ACT_SOFTPLUS) return v > 20.0f ? v : log1pf(expf(v));
    return v;
}

// ---- ... and on the bf16 matrix cores, every f32 operand split into three bf16 terms -------------------------------------------------
// scripts/micro/mfma_valu_overlap.hip: the f32-input MFMAs run on the SIMD's f32 FMA lanes - a wave's VALU instructions do not issue in
// their shadow (2 MFMAs + 16 v_fma per loop trip: 208 cycles against 128 / 76 alone), so the three dense training kernels cost MFMA time
// PLUS epilogue time whichever way they are scheduled (tried this round, profiles/r04_notes.txt: 32-row tiles on the 32x32x2 form 237 us
// per 1 M rows, the same with the epilogue of tile t - 1 interleaved into tile t's MFMA stream 229, against 225 for k_train_fwd).
// The bf16 matrix cores are a separate pipe at 16x the rate.  x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi),
// lo = bf16(x - hi - mid): exact (8 + 8 + 8 significand bits, signed), and  x w = hi wh + (hi wm + mid wh) + (mid wm + hi wl + lo wh) + O(2^-24 |x w|):
// six v_mfma_f32_32x32x16_bf16 per 16 columns of k, each product exact in f32, f32 accumulation - the accuracy of an f32 product chain
// (not its bits; scripts/micro/rowgemm_check.hip: max error against a float64 loop 0.5-1.4e-6 for both forms).
// Round to NEAREST at both levels (v_cvt_pk_bf16_f32): with truncation every dropped term has the sign of its product, a relative bias of
// ~3e-8 that the column statistics and the BatchNormalization gradients add up coherently over 10^4 .. 10^6 rows (state-network
// gradients of the 40 000-node test against float64: 3e-4 .. 7e-3 with truncation; the f32 chain 1e-5 .. 4e-4).
__device__ __forceinline__ void split3_pair(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
    const bf16x2 hv = {(__bf16)a, (__bf16)b};
    h = __builtin_bit_cast(unsigned, hv);
    const float a1 = a - __uint_as_float(h << 16), b1 = b - __uint_as_float(h & 0xFFFF0000u);
    const bf16x2 mv = {(__bf16)a1, (__bf16)b1};
    m = __builtin_bit_cast(unsigned, mv);
    const float a2 = a1 - __uint_as_float(m << 16), b2 = b1 - __uint_as_float(m & 0xFFFF0000u);
    const bf16x2 lv = {(__bf16)a2, (__bf16)b2};
    l = __builtin_bit_cast(unsigned, lv);
}
__device__ __forceinline__ void split3_x8(const f32x4 &x0, const f32x4 &x1, u32x4 &h, u32x4 &m, u32x4 &l) {
    unsigned hh[4], mm[4], ll[4];
    split3_pair(x0[0], x0[1], hh[0], mm[0], ll[0]); split3_pair(x0[2], x0[3], hh[1], mm[1], ll[1]);
    split3_pair(x1[0], x1[1], hh[2], mm[2], ll[2]); split3_pair(x1[2], x1[3], hh[3], mm[3], ll[3]);
    h = (u32x4){hh[0], hh[1], hh[2], hh[3]}; m = (u32x4){mm[0], mm[1], mm[2], mm[3]}; l = (u32x4){ll[0], ll[1], ll[2], ll[3]};
}
// the activation of the bf16-split kernels: selu / elu with the exponential as ONE v_exp_f32 (2^(x log2 e): the argument's rounding moves
// e^x by |x| 6e-8 relative - x e^x <= 0.37: 2e-8 absolute - where expf()'s range reduction and its two range checks cost 12 instructions
// a value, 190 of a tile's 780)
template <int ACT> __device__ __forceinline__ float activate_b6(float v) {
    if (ACT == GNN_ACT_SELU) {
        const float e = __builtin_amdgcn_exp2f(fminf(v, 0.0f) * 1.4426950408889634f);
        constexpr float SC = 1.0507009873554805f, SA = 1.0507009873554805f * 1.6732632423543772f;
        return fmaf(SA, e, fmaf(SC, fmaxf(v, 0.0f), -SA));
    }
    if (ACT == GNN_ACT_ELU) return fmaxf(v, 0.0f) + (__builtin_amdgcn_exp2f(fminf(v, 0.0f) * 1.4426950408889634f) - 1.0f);
    return activate1<ACT>(v);
}
__device__ __forceinline__ f32x4 fma4(const f32x4 &a, const f32x4 &b, const f32x4 &c) { return __builtin_elementwise_fma(a, b, c); }

// one f32 weight into its three bf16 planes in LDS (16-bit stores; once per workgroup)
__device__ __forceinline__ void split3_store(unsigned short *base, int plane_stride, int idx, float v) {
    const __bf16 vh = (__bf16)v;
    const float r1 = v - (float)vh;
    const __bf16 vm = (__bf16)r1;
    const float r2 = r1 - (float)vm;
    const __bf16 vl = (__bf16)r2;
    base[idx] = __builtin_bit_cast(unsigned short, vh); base[plane_stride + idx] = __builtin_bit_cast(unsigned short, vm);
    base[2 * plane_stride + idx] = __builtin_bit_cast(unsigned short, vl);
}

__device__ __forceinline__ f32x4 mfma_b6_16(const u32x4 &wh, const u32x4 &wm, const u32x4 &wl, const u32x4 &xh, const u32x4 &xm, const u32x4 &xl, f32x4 acc) {
#define B8(v_) __builtin_bit_cast(bf16x8, v_)
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B8(wl), B8(xh), acc, 0, 0, 0);      // small terms first
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B8(wh), B8(xl), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B8(wm), B8(xm), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B8(wm), B8(xh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B8(wh), B8(xm), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B8(wh), B8(xh), acc, 0, 0, 0);
#undef B8
    return acc;
}

// k_train_fwd's tile (16 rows a wave, lane (c, g) = row c, 16-byte pieces at columns 16 q + 4 g) on v_mfma_f32_16x16x32_bf16: a k block is
// two chunks (the lane's 8 values = its pieces of chunks 2 kb and 2 kb + 1; the weight planes are laid out in the same order).  The row
// registers are refilled with the next tile's chunks as soon as they have been split (the state chunks after the epilogue, which compares
// with them), the activation is a template parameter and every select that hipcc could turn into a branch is arithmetic.
#ifndef TB_B6_WAVES
#define TB_B6_WAVES 2
#endif
// One wait per trip.  hipcc counts a wave's loads and stores in one in-order counter and, with a store pending, waits for everything
// issued before it - so a tile's stores directly in front of the next tile's first wait cost their whole round trip every trip (the
// kernels above).  Here a trip is: wait for the tile's rows (requested a whole trip ago) -> split ALL of them to bf16 (60 registers)
// and keep the state chunks for the predicate -> request the next tile's rows -> store the PREVIOUS tile's output (its activated values
// waited in 16 registers) -> products -> activation, statistics, predicate.  Nothing the next wait covers is younger than most of a trip.
template <int SQ, int ACT, bool ADD>
__device__ __forceinline__ void train_fwd_b6_body(const TrainFwdArgs &a, const int bid, const int nblk) {
    TB_MARK(0); TB_BLOCK_TIME(0);
    if (gate_closed(a.gate)) return;
    constexpr int XQ = ADD ? 0 : 2;                     // 16-column chunks of the constants line (ADD: none - TrainFwdArgs::addend stands in for their product)
    constexpr int NCT = SQ, HP = 16 * NCT, NQ = 2 * SQ + XQ, NKB = NQ / 2;
    constexpr int PLANE = NKB * NCT * 64 * 8;                                                   // bf16 elements of one weight plane
    extern __shared__ __attribute__((aligned(16))) float tb_smem[];
    unsigned short *Wl = reinterpret_cast<unsigned short *>(tb_smem);                           // [3 planes][NKB][NCT][64 lanes][8]
    float *bias_l = tb_smem + 3 * PLANE / 2;           // [HP]
    float *red = bias_l + HP;                          // [TB_WAVES][2 HP]
    float *shift_l = red + TB_WAVES * 2 * HP;          // [HP] what the statistics are taken around
    float *mean_l = shift_l + HP;                      // [16 NQ] the input columns' means (a.in_mean), 0 where there is none
    __shared__ int any_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    if (tid == 0) any_s = 0;
    for (int h = tid; h < HP; h += 64 * TB_WAVES) shift_l[h] = (a.stat_shift && h < a.H) ? a.stat_shift[h] : 0.0f;
    for (int k = tid; k < 16 * NQ; k += 64 * TB_WAVES) mean_l[k] = fwd_in_mean(a, SQ, k);
    // the folded weights -> three bf16 planes.  Weight row of every virtual input column first (a table in `red`, free until the end), then
    // the elements in batches of independent loads (one dependent load per trip of a 20-trip loop was 10 us of every launch)
    int *rowtab = reinterpret_cast<int *>(red);
    for (int k = tid; k < 16 * NQ; k += 64 * TB_WAVES) {
        int row = -1;
        if (k < 16 * SQ) row = a.wrow_state + k;
        else if (k < 32 * SQ) row = a.wrow_agg + (k - 16 * SQ);
        else if (!ADD && a.xc) {
            int jj = k - 32 * SQ, beg = 0;
#pragma unroll
            for (int sg = 0; sg < 3; ++sg) {
                if (sg < a.cs.n && jj >= beg && jj < beg + a.cs.width[sg]) row = a.cs.wrow[sg] + (jj - beg);
                if (sg < a.cs.n) beg += a.cs.width[sg];
            }
        }
        rowtab[k] = row;
    }
    __syncthreads();
    {
        const __amdgpu_buffer_rsrc_t r_w = buf_rsrc(a.Wf);
        constexpr int PER = (16 * NQ * HP) / (64 * TB_WAVES), BATCH = PER % 5 == 0 ? 5 : PER % 4 == 0 ? 4 : PER % 3 == 0 ? 3 : PER % 2 == 0 ? 2 : 1;       // (64 TB_WAVES is a multiple of HP: a thread keeps its column h)
        static_assert((16 * NQ * HP) % (64 * TB_WAVES) == 0 && (64 * TB_WAVES) % HP == 0 && PER % BATCH == 0, "weight staging: whole batches");
        const int h = tid % HP, k0 = tid / HP;
#pragma unroll 1
        for (int it0 = 0; it0 < PER; it0 += BATCH) {
            float v[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int k = k0 + (it0 + u) * ((64 * TB_WAVES) / HP), row = rowtab[k];
                v[u] = buf_ld_f32(r_w, (row >= 0 && h < a.H) ? ((unsigned)row * (unsigned)a.H + (unsigned)h) * 4u : BUF_OFF);
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int k = k0 + (it0 + u) * ((64 * TB_WAVES) / HP);
                const int q = k >> 4, gg = (k >> 2) & 3, e = k & 3;
                split3_store(Wl, PLANE, ((((q >> 1) * NCT + (h >> 4)) * 64 + 16 * gg + (h & 15)) * 8) + 4 * (q & 1) + e, v[u]);
            }
        }
    }
    __syncthreads();                                     // (rowtab lives in `red`)
    for (int h = tid; h < HP; h += 64 * TB_WAVES) bias_l[h] = h < a.H ? a.bf[h] : 0.0f;
    __syncthreads();
    TB_MARK(1);

    // windows of exactly the arrays' sizes: the rows past M of the last tile are out of range by themselves (they read 0), no select per load
    const __amdgpu_buffer_rsrc_t r_s = buf_rsrc_n(a.state, (unsigned)a.M * (unsigned)a.ld_state * 4u), r_a = buf_rsrc_n(a.agg, (unsigned)a.M * (unsigned)a.ld_agg * 4u),
                                 r_x = ADD ? buf_rsrc_n(uniform_ptr(a.addend), (unsigned)a.M * (unsigned)a.ld_add * 4u) : buf_rsrc_n(uniform_ptr(a.xc), (unsigned)a.M * 128u),
                                 r_y = buf_rsrc(a.Y);
    const int n_tiles = (a.M + 15) >> 4;
    f32x4 cs1[NCT], cs2[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) { cs1[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; cs2[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    int any = 0;
    f32x4 A[NQ];
    f32x4 Cn[ADD ? NCT : 1];                            // ADD: this lane's pieces of the addend's row (the accumulators' layout: columns 16 ct + 4 g)
    const unsigned o_s = (unsigned)c * (unsigned)a.ld_state * 4u + 16u * g, o_a = (unsigned)c * (unsigned)a.ld_agg * 4u + 16u * g,
                   o_x = ADD ? (unsigned)c * (unsigned)a.ld_add * 4u + 16u * g : (unsigned)c * 128u + 16u * g;
    auto fetch = [&](int t) {                           // this lane's 16-byte pieces of row 16 t + c: state, agg, constants line (or the addend)
        const unsigned b_s = 64u * (unsigned)t * (unsigned)a.ld_state + o_s, b_a = 64u * (unsigned)t * (unsigned)a.ld_agg + o_a,
                       b_x = ADD ? 64u * (unsigned)t * (unsigned)a.ld_add + o_x : 2048u * (unsigned)t + o_x;
#pragma unroll
        for (int q = 0; q < SQ; ++q) {
            A[q] = buf_ld_f32x4_aux<TB_STREAM_AUX>(r_s, b_s + 64u * q);
            A[SQ + q] = buf_ld_f32x4_aux<TB_STREAM_AUX>(r_a, b_a + 64u * q);
        }
        if constexpr (ADD) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) Cn[ct] = buf_ld_f32x4_aux<TB_STREAM_AUX>(r_x, b_x + 64u * ct);
        } else {
#pragma unroll
            for (int q = 0; q < XQ; ++q) A[2 * SQ + q] = buf_ld_f32x4_aux<TB_STREAM_AUX>(r_x, b_x + 64u * q);
        }
    };
    const int t_step = nblk * TB_WAVES;
    int trip_ = 0; (void)trip_;
    fetch(bid * TB_WAVES + wave);
    const u32x4 *Wv = reinterpret_cast<const u32x4 *>(Wl) + lane;
    f32x4 vP[NCT];                                      // the previous tile's output, stored one trip late
    int offP = 0; bool inP = false;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) vP[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int t = bid * TB_WAVES + wave; t < n_tiles; t += t_step) {
        const int row = 16 * t + c;
        const bool in = row < a.M;
        TB_STAMP(0);
        u32x4 xh[NKB], xm[NKB], xl[NKB];
        f32x4 old[SQ];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)             // (centred: see TrainFwdArgs::in_mean; the predicate below compares with the raw state)
            split3_x8pk(A[2 * kb] - *reinterpret_cast<const f32x4 *>(mean_l + 32 * kb + 4 * g), A[2 * kb + 1] - *reinterpret_cast<const f32x4 *>(mean_l + 32 * kb + 16 + 4 * g),
                        xh[kb], xm[kb], xl[kb]);
#pragma unroll
        for (int q = 0; q < SQ; ++q) old[q] = A[q];
        f32x4 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            acc[ct] = *reinterpret_cast<const f32x4 *>(bias_l + 16 * ct + 4 * g);
            if constexpr (ADD) acc[ct] += Cn[ct];          // (before the next tile's pieces take the registers)
        }
        __builtin_amdgcn_sched_barrier(0);
        fetch(t + t_step);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const u32x4 bits = {__float_as_uint(vP[ct][0]), __float_as_uint(vP[ct][1]), __float_as_uint(vP[ct][2]), __float_as_uint(vP[ct][3])};
            __builtin_amdgcn_raw_buffer_store_b128(bits, r_y, inP ? offP + (int)((16u * ct + 4u * g) * 4u) : (int)BUF_OFF, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        TB_STAMP_DEP(1, xh[NKB - 1]);
        // products in stages of (k block, CTG column tiles): a stage's weight fragments (3 planes x CTG) are read from LDS one stage ahead,
        // and inside a stage consecutive MFMAs go to different accumulators (left alone hipcc builds chains of six dependent MFMAs with
        // an LDS round trip in front of each: 14 000 cycles per tile)
        constexpr int CTG = NCT < 2 ? NCT : 2, SPK = NCT / CTG, NS = NKB * SPK;
        u32x4 W[2][3][CTG];
        auto load_w = [&](int st, u32x4 (&w)[3][CTG]) {
            const int kb = st / SPK, ct0 = (st % SPK) * CTG;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int u = 0; u < CTG; ++u) w[pl][u] = Wv[pl * (PLANE / 8) + (kb * NCT + ct0 + u) * 64];
        };
        load_w(0, W[0]);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            const int kb = st / SPK, ct0 = (st % SPK) * CTG;
            if (st + 1 < NS) load_w(st + 1, W[(st + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#define B8(v_) __builtin_bit_cast(bf16x8, v_)
#define MF(pl_, x_) _Pragma("unroll") for (int u = 0; u < CTG; ++u) acc[ct0 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B8(W[st & 1][pl_][u]), B8(x_), acc[ct0 + u], 0, 0, 0)
            MF(2, xh[kb]); MF(0, xl[kb]); MF(1, xm[kb]); MF(1, xh[kb]); MF(0, xm[kb]); MF(0, xh[kb]);     // small terms first
#undef MF
#undef B8
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
        TB_STAMP_DEP(2, acc[NCT - 1]);
        // (packed f32 arithmetic: two values an instruction; the rows past M of the last tile read zeros - their outputs are not stored and
        // `mask` / `in` keep them out of the statistics and the predicate)
        f32x4 d2v = {0.f, 0.f, 0.f, 0.f}, n2v = {0.f, 0.f, 0.f, 0.f};
        const float mask = in ? 1.0f : 0.0f;
        const f32x4 mask4 = {mask, mask, mask, mask};
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            f32x4 v = acc[ct];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (TB_ABL & 16) ? v[e] : activate_b6<ACT>(v[e]);       // (H == 16 SQ: the launcher checks; a column mask here becomes a branch per element)
            vP[ct] = v;
            { const f32x4 dv = (v - *reinterpret_cast<const f32x4 *>(shift_l + 16 * ct + 4 * g)) * mask4; cs1[ct] += dv; cs2[ct] = fma4(dv, dv, cs2[ct]); }
            { const f32x4 dd = v - old[ct]; d2v = fma4(dd, dd, d2v); n2v = fma4(old[ct], old[ct], n2v); }
        }
        float d2 = (d2v[0] + d2v[1]) + (d2v[2] + d2v[3]), n2 = (n2v[0] + n2v[1]) + (n2v[2] + n2v[3]);
        d2 += __shfl_xor(d2, 16, 64); d2 += __shfl_xor(d2, 32, 64);
        n2 += __shfl_xor(n2, 16, 64); n2 += __shfl_xor(n2, 32, 64);
        any |= (in && sqrtf(d2) > a.thr * sqrtf(n2)) ? 1 : 0;
        offP = (int)((unsigned)row * (unsigned)a.ldy * 4u); inP = in;
        TB_STAMP_DEP(3, vP[NCT - 1]);
        TB_TRIP_END();
    }
    TB_MARK(2);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {                 // the last tile's output
        const u32x4 bits = {__float_as_uint(vP[ct][0]), __float_as_uint(vP[ct][1]), __float_as_uint(vP[ct][2]), __float_as_uint(vP[ct][3])};
        __builtin_amdgcn_raw_buffer_store_b128(bits, r_y, inP ? offP + (int)((16u * ct + 4u * g) * 4u) : (int)BUF_OFF, 0, 0);
    }
    if (a.pred_flag && __any(any) && lane == 0) any_s = 1;
    if (a.stat_part) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float s1 = row16_sum_to_lane15(cs1[ct][e]), s2 = row16_sum_to_lane15(cs2[ct][e]);
                if (c == 15) { red[wave * 2 * HP + 16 * ct + 4 * g + e] = s1; red[wave * 2 * HP + HP + 16 * ct + 4 * g + e] = s2; }
            }
    }
    __syncthreads();
    if (a.stat_part && tid < 2 * HP) {
        float tsum = 0.0f;
        for (int w_ = 0; w_ < TB_WAVES; ++w_) tsum += red[w_ * 2 * HP + tid];
        a.stat_part[(size_t)bid * 2 * HP + tid] = tsum;
    }
    TB_MARK(3); TB_BLOCK_TIME(1);
    if (a.pred_flag && tid == 0) {
        if (any_s) atomicOr(a.pred_flag, 1);
        if (bid == 0 && a.pred_k) *a.pred_k = a.pred_kval;
    }
}

template <int SQ, int ACT, bool ADD = false>
__global__ void __launch_bounds__(64 * TB_WAVES, TB_B6_WAVES) k_train_fwd_b6(TrainFwdArgs a) { train_fwd_b6_body<SQ, ACT, ADD>(a, blockIdx.x, gridDim.x); }
template <int SQ, int ACT, bool ADD>         // every node type's rows in one launch (TypeLaunch)
__global__ void __launch_bounds__(64 * TB_WAVES, TB_B6_WAVES) k_train_fwd_b6_types(TypeLaunch<TrainFwdArgs> m) {
    int bid, nblk;
    const TrainFwdArgs a = select_type(m, bid, nblk);
    train_fwd_b6_body<SQ, ACT, ADD>(a, bid, nblk);
}

template <int SQ, bool ADD = false>
inline size_t train_fwd_b6_lds() {
    constexpr int NQ = 2 * SQ + (ADD ? 0 : 2);
    return (size_t)(3 * (NQ / 2) * SQ * 64 * 8 * 2) + (size_t)(16 * SQ + TB_WAVES * 2 * 16 * SQ + 16 * SQ + 16 * NQ) * sizeof(float);
}

template <int SQ, int NCT>
inline size_t train_fwd_lds() { return (size_t)(16 * (2 * SQ + 2) * 16 * NCT + 16 * NCT + TB_WAVES * 2 * 16 * NCT + 16 * NCT + 16 * (2 * SQ + 2)) * sizeof(float); }

// ---- backward: d loss / d [state | agg] through the first Dense and its training-mode BatchNormalization ------------------------
//   dy[m, j]  = sum_h dZ[m, h] W[row_j, h]                      (j < S: state column j, else agg column j - S)
//   dx[m, j]  = gamma_k rstd_k (dy - m1_k - xhat m2_k)          xhat = (x[m, j] - mean_k) rstd_k,  k = row_j      (reference: the
//               gradient autograd takes through tf.keras BatchNormalization(training=True); kernels_train.hpp k_bn_input_grad)
//             = Ac_j (dy - m1_k) + Cc_j (x - mean_k)      with  Ac = gamma rstd,  Cc = -Ac rstd m2
//   agg half optionally times row_scale[m]: the transposed aggregate then walks unit weights ('average': w_e = 1 / in-degree(dst_e))
struct TrainBwdArgs {
    int M;
    const float *dZ; int ldz;             // [M, H], H = 16 HQ
    const float *Y; int act;              // optional: `dZ` holds G = d loss / d output and dZ = G (.) act'(Y) is formed as the rows arrive
    const float *W; int ldw;              // first-layer kernel [in_dim][ldw] (NOT folded: BatchNorm enters through the coefficients)
    int H, S, wrow_state, wrow_agg;
    const float *state; int ld_state; const float *agg; int ld_agg;   // the layer's inputs x (for xhat)
    const float *gamma, *mean, *var, *m1, *m2; float eps;               // NULL gamma: no BatchNormalization
    const float *agg_row_scale;           // [M] or NULL
    float *dx; int ld_dx;                 // [M, 2 S]
    int defer_state_bn;                   // != 0 (with BatchNormalization): the STATE half leaves as Ac dy only - the rest of its BatchNorm input
                                          // gradient, Cc (x - mean) - Ac m1, is added where the rows of x = state_t are read anyway: by
                                          // k_aggregate_dz, which turns this iteration's dx into the previous iteration's dZ (state_t is that
                                          // iteration's output).  The kernel then does not read the state rows at all (256 of 1536 bytes per row).
};

template <int HQ, int NCT>                 // NCT = 2 S / 16 output column tiles
__global__ void __launch_bounds__(64 * TB_WAVES, 4) k_train_bwd_dx(TrainBwdArgs a) {
    constexpr int HP = 16 * NCT;           // = 2 S
    constexpr int SQ = NCT / 2;            // 16-column tiles of one half (state | agg)
    extern __shared__ __attribute__((aligned(16))) float tb_smem[];
    float *Wl = tb_smem;                    // [4 HQ k-steps][4 g][16 c][NCT]
    float *coef = tb_smem + 16 * HQ * HP;   // [4][HP]: Ac, Cc, m1, mean
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int S = a.S;                      // == 8 NCT: the halves are whole 16-column tiles
    for (int i = tid; i < 16 * HQ * HP; i += 64 * TB_WAVES) {
        const int k = i / HP, j = i % HP;                                   // k = dZ column h, j = output column
        const int row = j < S ? a.wrow_state + j : a.wrow_agg + (j - S);
        const float v = k < a.H ? a.W[(size_t)row * a.ldw + k] : 0.0f;
        const int q = k >> 4, rem = k & 15, gg = rem >> 2, e = rem & 3;
        Wl[(((4 * q + e) * 4 + gg) * 16 + (j & 15)) * NCT + (j >> 4)] = v;
    }
    for (int j = tid; j < HP; j += 64 * TB_WAVES) {
        float Ac = 1.0f, Cc = 0.0f, M1 = 0.0f, Mu = 0.0f;      // dx = Ac (dy - m1) + Cc (x - mean): the centred form - an Ac dy + Cc x + Bc with
        if (a.gamma) {                                          // Bc = -Ac m1 - Cc mean rounds ONE constant per column whose error every row shares
            const int k = j < S ? a.wrow_state + j : a.wrow_agg + (j - S);
            const float rstd = 1.0f / sqrtf(a.var[k] + a.eps);
            Ac = a.gamma[k] * rstd; Cc = -Ac * rstd * a.m2[k]; M1 = a.m1[k]; Mu = a.mean[k];
            if (a.defer_state_bn && j < S) { Cc = 0.0f; M1 = 0.0f; Mu = 0.0f; }       // (see TrainBwdArgs::defer_state_bn)
        }
        coef[j] = Ac; coef[HP + j] = Cc; coef[2 * HP + j] = M1; coef[3 * HP + j] = Mu;
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r_z = buf_rsrc(a.dZ), r_y = buf_rsrc(a.Y), r_s = buf_rsrc(a.defer_state_bn ? nullptr : a.state), r_a = buf_rsrc(a.agg), r_o = buf_rsrc(a.dx),
                                 r_rs = buf_rsrc(a.agg_row_scale);
    const int n_tiles = (a.M + 15) >> 4;
#pragma unroll 1
    for (int t = blockIdx.x * TB_WAVES + wave; t < n_tiles; t += gridDim.x * TB_WAVES) {
        const int row = 16 * t + c;
        const bool in = row < a.M;
        // every load of the tile is issued here: the dZ row and - for the BatchNorm term - the layer's inputs x = [state | agg] of the
        // same row, all as 16-byte chunks (columns 16 q + 4 g ..): the transposed product (see k_train_fwd) returns dy in that layout
        f32x4 A[HQ], X[NCT], Yv[HQ];
#pragma unroll
        for (int q = 0; q < HQ; ++q) A[q] = buf_ld_f32x4(r_z, in ? ((unsigned)row * (unsigned)a.ldz + 16u * q + 4u * g) * 4u : BUF_OFF);
        if (a.Y) {
#pragma unroll
            for (int q = 0; q < HQ; ++q) Yv[q] = buf_ld_f32x4(r_y, in ? ((unsigned)row * (unsigned)a.ldz + 16u * q + 4u * g) * 4u : BUF_OFF);
        }
        if (a.gamma) {
#pragma unroll
            for (int q = 0; q < SQ; ++q) {
                X[q] = buf_ld_f32x4(r_s, in ? ((unsigned)row * (unsigned)a.ld_state + 16u * q + 4u * g) * 4u : BUF_OFF);
                X[SQ + q] = buf_ld_f32x4(r_a, in ? ((unsigned)row * (unsigned)a.ld_agg + 16u * q + 4u * g) * 4u : BUF_OFF);
            }
        }
        const float rs = a.agg_row_scale ? buf_ld_f32(r_rs, in ? (unsigned)row * 4u : BUF_OFF) : 1.0f;
        if (a.Y) {
#pragma unroll
            for (int q = 0; q < HQ; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) A[q][e] *= activate_grad_from_output(a.act, Yv[q][e]);
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {                       // state half, then agg half (16 accumulator registers at a time)
            f32x4 acc[SQ];
#pragma unroll
            for (int u = 0; u < SQ; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < HQ; ++q) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    BFrag<NCT> w;
                    w.load(Wl + (((4 * q + e) * 4 + g) * 16 + c) * NCT);
#pragma unroll
                    for (int u = 0; u < SQ; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.v[half * SQ + u], A[q][e], acc[u], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            TB_MFMA_DRAIN();
#pragma unroll
            for (int u = 0; u < SQ; ++u) {
                const int j0 = 16 * (half * SQ + u) + 4 * g;            // this lane's four output columns
                f32x4 v = acc[u];
                if (a.gamma) {
                    const f32x4 Ac = *reinterpret_cast<const f32x4 *>(coef + j0), Cc = *reinterpret_cast<const f32x4 *>(coef + HP + j0),
                                M1 = *reinterpret_cast<const f32x4 *>(coef + 2 * HP + j0), Mu = *reinterpret_cast<const f32x4 *>(coef + 3 * HP + j0);
                    const f32x4 x = X[half * SQ + u];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(Ac[e], v[e] - M1[e], Cc[e] * (x[e] - Mu[e]));
                }
                if (half == 1) v *= rs;
                const u32x4 bits = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                __builtin_amdgcn_raw_buffer_store_b128(bits, r_o, in ? (int)(((unsigned)row * (unsigned)a.ld_dx + (unsigned)j0) * 4u) : (int)BUF_OFF, 0, 0);
            }
        }
    }
}

template <int ACT> __device__ __forceinline__ float activate_grad1(float y) {          // act'(z) from y = act(z), as kernels_train.hpp's switch
    if (ACT == GNN_ACT_RELU) return y > 0.0f ? 1.0f : 0.0f;
    if (ACT == GNN_ACT_SELU) return y > 0.0f ? 1.0507009873554805f : y + 1.0507009873554805f * 1.6732632423543772f;
    if (ACT == GNN_ACT_TANH) return 1.0f - y * y;
    if (ACT == GNN_ACT_SIGMOID) return y * (1.0f - y);
    if (ACT == GNN_ACT_ELU) return y > 0.0f ? 1.0f : y + 1.0f;
    if (ACT == GNN_ACT_SOFTPLUS) return 1.0f - expf(-y);
    return 1.0f;
}

// k_train_bwd_dx on the bf16 matrix cores (the three-term split of k_train_fwd_b6: dZ rows are split as they arrive, W^T sits in LDS as
// three bf16 planes in fragment order).  256-thread workgroups, two per CU; every row register is refilled with the NEXT tile's piece as
// soon as its last reader has issued (dZ and Y after the split, the layer inputs x after the BatchNorm term of their column tile), so a
// wave keeps 16 KB requested for a whole trip.  `a.Y == NULL` (dZ already formed): instantiate with ACT = LINEAR - the loads of Y then
// fall out of range and return zeros, act'(0) = 1; no BatchNormalization: the x loads fall out of range the same way (Cc = m1 = mean = 0).
template <int HQ, int ACT>
__device__ __forceinline__ void train_bwd_dx_b6_body(const TrainBwdArgs &a, const int bid, const int nblk) {
    constexpr int NCT = 2 * HQ, HP = 16 * NCT, SQ = HQ, NKB = (HQ + 1) / 2, NW = 4;
    constexpr int PLANE = NKB * NCT * 64 * 8;
    extern __shared__ __attribute__((aligned(16))) float tb_smem[];
    unsigned short *Wl = reinterpret_cast<unsigned short *>(tb_smem);        // [3 planes][NKB][NCT][64 lanes][8]
    float *coef = tb_smem + 3 * PLANE / 2;                                    // [4][HP]: Ac, Cc, m1, mean
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int S = a.S;                                                        // == 16 HQ == H (the launcher checks)
    {   // W^T -> three bf16 planes, eight independent loads a batch (one dependent load per trip of a 32-trip loop was a tenth of the launch)
        const __amdgpu_buffer_rsrc_t r_w = buf_rsrc(a.W);
        constexpr int TOT = 32 * NKB * HP, BATCH = (TOT / (64 * NW)) % 8 == 0 ? 8 : 4;
        static_assert(TOT % (64 * NW * BATCH) == 0, "weight staging: whole batches");
#pragma unroll 1
        for (int i0 = tid; i0 < TOT; i0 += 64 * NW * BATCH) {
            float v[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int i = i0 + u * 64 * NW, k = i / HP, j = i % HP;       // k = dZ column h, j = output column
                const int row = j < S ? a.wrow_state + j : a.wrow_agg + (j - S);
                v[u] = buf_ld_f32(r_w, k < a.H ? ((unsigned)row * (unsigned)a.ldw + (unsigned)k) * 4u : BUF_OFF);
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int i = i0 + u * 64 * NW, k = i / HP, j = i % HP;
                const int q = k >> 4, gg = (k >> 2) & 3, e = k & 3;
                split3_store(Wl, PLANE, ((((q >> 1) * NCT + (j >> 4)) * 64 + 16 * gg + (j & 15)) * 8) + 4 * (q & 1) + e, v[u]);
            }
        }
    }
    for (int j = tid; j < HP; j += 64 * NW) {
        float Ac = 1.0f, Cc = 0.0f, M1 = 0.0f, Mu = 0.0f;      // (the centred form: see k_train_bwd_dx)
        if (a.gamma) {
            const int k = j < S ? a.wrow_state + j : a.wrow_agg + (j - S);
            const float rstd = 1.0f / sqrtf(a.var[k] + a.eps);
            Ac = a.gamma[k] * rstd; Cc = -Ac * rstd * a.m2[k]; M1 = a.m1[k]; Mu = a.mean[k];
            if (a.defer_state_bn && j < S) { Cc = 0.0f; M1 = 0.0f; Mu = 0.0f; }       // (see TrainBwdArgs::defer_state_bn)
        }
        coef[j] = Ac; coef[HP + j] = Cc; coef[2 * HP + j] = M1; coef[3 * HP + j] = Mu;
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r_z = buf_rsrc(a.dZ), r_y = buf_rsrc(a.Y), r_s = buf_rsrc((a.gamma && !a.defer_state_bn) ? a.state : nullptr), r_a = buf_rsrc(a.gamma ? a.agg : nullptr),
                                 r_o = buf_rsrc(a.dx), r_rs = buf_rsrc(a.agg_row_scale);
    const int n_tiles = (a.M + 15) >> 4;
    const int t_step = nblk * NW;
    f32x4 A[HQ], Yv[HQ], X[NCT];
    float rs;
    auto off_row = [&](int t, bool &in_) { const int row_ = 16 * t + c; in_ = t < n_tiles && row_ < a.M; return (unsigned)row_; };
    auto fetch_zy = [&](int t) {
        bool in_; const unsigned r = off_row(t, in_);
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
            A[q] = buf_ld_f32x4_aux<TB_STREAM_AUX>(r_z, in_ ? (r * (unsigned)a.ldz + 16u * q + 4u * g) * 4u : BUF_OFF);
            Yv[q] = buf_ld_f32x4_aux<TB_STREAM_AUX>(r_y, in_ ? (r * (unsigned)a.ldz + 16u * q + 4u * g) * 4u : BUF_OFF);
        }
    };
    auto fetch_x = [&](int t, int ct) {
        bool in_; const unsigned r = off_row(t, in_);
        if (ct < SQ) X[ct] = buf_ld_f32x4_aux<TB_STREAM_AUX>(r_s, in_ ? (r * (unsigned)a.ld_state + 16u * ct + 4u * g) * 4u : BUF_OFF);
        else X[ct] = buf_ld_f32x4_aux<TB_STREAM_AUX>(r_a, in_ ? (r * (unsigned)a.ld_agg + 16u * (ct - SQ) + 4u * g) * 4u : BUF_OFF);
    };
    auto fetch_rs = [&](int t) { bool in_; const unsigned r = off_row(t, in_); rs = buf_ld_f32(r_rs, in_ ? r * 4u : BUF_OFF); };
    {
        const int t0 = bid * NW + wave;
        fetch_zy(t0); fetch_rs(t0);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) fetch_x(t0, ct);
    }
    const u32x4 *Wv = reinterpret_cast<const u32x4 *>(Wl) + lane;
    const bool has_rs = a.agg_row_scale != nullptr;
#pragma unroll 1
    for (int t = bid * NW + wave; t < n_tiles; t += t_step) {
        const int row = 16 * t + c;
        const bool in = row < a.M;
        u32x4 xh[NKB], xm[NKB], xl[NKB];
#pragma unroll
        for (int q = 0; q < HQ; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) A[q][e] *= activate_grad1<ACT>(Yv[q][e]);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            split3_x8(A[2 * kb], 2 * kb + 1 < HQ ? A[2 * kb + 1 < HQ ? 2 * kb + 1 : 0] : z4, xh[kb], xm[kb], xl[kb]);
        }
        const float rs_t = has_rs ? rs : 1.0f;
        __builtin_amdgcn_sched_barrier(0);
        fetch_zy(t + t_step); fetch_rs(t + t_step);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        constexpr int CTG = 2, SPK = NCT / CTG, NS = NKB * SPK;
        u32x4 W[2][3][CTG];
        auto load_w = [&](int st, u32x4 (&w)[3][CTG]) {
            const int kb = st / SPK, ct0 = (st % SPK) * CTG;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int u = 0; u < CTG; ++u) w[pl][u] = Wv[pl * (PLANE / 8) + (kb * NCT + ct0 + u) * 64];
        };
        load_w(0, W[0]);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            const int kb = st / SPK, ct0 = (st % SPK) * CTG;
            if (st + 1 < NS) load_w(st + 1, W[(st + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#define B8(v_) __builtin_bit_cast(bf16x8, v_)
#define MF(pl_, x_) _Pragma("unroll") for (int u = 0; u < CTG; ++u) acc[ct0 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B8(W[st & 1][pl_][u]), B8(x_), acc[ct0 + u], 0, 0, 0)
            MF(2, xh[kb]); MF(0, xl[kb]); MF(1, xm[kb]); MF(1, xh[kb]); MF(0, xm[kb]); MF(0, xh[kb]);
#undef MF
#undef B8
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const int j0 = 16 * ct + 4 * g;
            const f32x4 Ac = *reinterpret_cast<const f32x4 *>(coef + j0), Cc = *reinterpret_cast<const f32x4 *>(coef + HP + j0),
                        M1 = *reinterpret_cast<const f32x4 *>(coef + 2 * HP + j0), Mu = *reinterpret_cast<const f32x4 *>(coef + 3 * HP + j0);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(Ac[e], acc[ct][e] - M1[e], Cc[e] * (X[ct][e] - Mu[e]));
            if (ct >= SQ) v *= rs_t;
            const u32x4 bits = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
            __builtin_amdgcn_raw_buffer_store_b128(bits, r_o, in ? (int)(((unsigned)row * (unsigned)a.ld_dx + (unsigned)j0) * 4u) : (int)BUF_OFF, 0, 0);
            fetch_x(t + t_step, ct);
        }
    }
}

template <int HQ, int ACT>
__global__ void __launch_bounds__(256, 2) k_train_bwd_dx_b6(TrainBwdArgs a) { train_bwd_dx_b6_body<HQ, ACT>(a, blockIdx.x, gridDim.x); }
template <int HQ, int ACT>                   // every node type's rows in one launch (TypeLaunch)
__global__ void __launch_bounds__(256, 2) k_train_bwd_dx_b6_types(TypeLaunch<TrainBwdArgs> m) {
    int bid, nblk;
    const TrainBwdArgs a = select_type(m, bid, nblk);
    train_bwd_dx_b6_body<HQ, ACT>(a, bid, nblk);
}

template <int HQ>
inline size_t train_bwd_b6_lds() { return (size_t)(3 * ((HQ + 1) / 2) * 2 * HQ * 64 * 8 * 2) + (size_t)(4 * 32 * HQ) * sizeof(float); }

template <int HQ, int NCT>
inline size_t train_bwd_lds() { return (size_t)(16 * HQ * 16 * NCT + 4 * 16 * NCT) * sizeof(float); }

// ---- weight gradient of the first Dense at large M: P = X^T dZ on the matrix cores, straight from memory ------------------------------------
// X = [state | agg | constants] (the virtual concatenation, never materialised), dZ = G (.) act'(Y) formed as the rows arrive.  The
// contraction runs over ROWS: for v_mfma_f32_16x16x4_f32 lane (c, g) supplies A[c][g] and B[g][c], i.e. one element of row r0 + g of each
// operand - so a lane loads the SQ consecutive floats  X[r0 + g][SQ c ..]  (a wave's load = four whole rows, contiguous) and uses value e
// of the piece as the A operand of row tile e: tile row m is input column SQ m + e, a permutation of P's rows that costs nothing.  dZ
// pieces serve as B operands the same way.  No LDS on the way in, every byte of every row is read exactly once; each wave keeps ALL
// (2 SQ + 2) x SQ accumulator tiles (160 registers at S = 64) for its share of the rows, PD steps of loads in flight; the four waves' tiles
// meet in LDS in wave order at the end and leave as one partial per workgroup in the layout of k_dense_grad_allk (k_reduce_partials /
// k_first_layer_param_grads take it from there).  The constants line carries a 1 behind its Kc < 32 columns: its row of P is q = colsum(dZ).
// 1 M rows, S = 64: k_dense_grad_allk 479 us + k_act_grad 127 us -> this kernel (profiles/r03_notes.txt).
struct TrainWgradArgs {
    int M, rows_per_wg;
    const float *G, *Y; int act;          // [M, S] each
    const float *state, *agg;             // [M, S]
    const float *xc;                      // [M, 32] (k_pack_xc layout)
    int K, wrow_state, wrow_agg, Kc; ConstCols cs;
    float *part;                          // [gridDim.x][K * S + S]
    const float *mean;                    // [K] column means of this iteration's BatchNormalization, or NULL: X is centred as it arrives (P - mean q^T)
};

// weight row (= BatchNorm column) of virtual input column kv of [state | agg | constants line], -1: none (the line's 1 and its padding)
__device__ __forceinline__ int wgrad_wrow(const TrainWgradArgs &a, int S, int kv) {
    if (kv < S) return a.wrow_state + kv;
    if (kv < 2 * S) return a.wrow_agg + (kv - S);
    int jj = kv - 2 * S, b0 = 0, wrow = -1;
#pragma unroll
    for (int sg = 0; sg < 3; ++sg) { if (sg < a.cs.n && jj >= b0 && jj < b0 + a.cs.width[sg]) wrow = a.cs.wrow[sg] + (jj - b0); if (sg < a.cs.n) b0 += a.cs.width[sg]; }
    return wrow;
}

template <int SQ> struct Piece { float v[SQ]; };
template <int SQ> __device__ __forceinline__ Piece<SQ> ld_piece(__amdgpu_buffer_rsrc_t r, unsigned off);
template <> __device__ __forceinline__ Piece<1> ld_piece<1>(__amdgpu_buffer_rsrc_t r, unsigned off) {
    Piece<1> p; p.v[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0)); return p; }
template <> __device__ __forceinline__ Piece<2> ld_piece<2>(__amdgpu_buffer_rsrc_t r, unsigned off) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0);
    Piece<2> p; p.v[0] = __uint_as_float(t[0]); p.v[1] = __uint_as_float(t[1]); return p; }
template <> __device__ __forceinline__ Piece<4> ld_piece<4>(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    Piece<4> p; p.v[0] = __uint_as_float(t[0]); p.v[1] = __uint_as_float(t[1]); p.v[2] = __uint_as_float(t[2]); p.v[3] = __uint_as_float(t[3]); return p; }

template <int SQ>
__global__ void __launch_bounds__(256, 2) k_train_wgrad(TrainWgradArgs a) {
    constexpr int S = 16 * SQ, RT = 2 * SQ + 2, KV = 2 * S + 32, PD = 3;
    __shared__ float Ps[KV * S];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int m_beg = blockIdx.x * a.rows_per_wg, m_end = min(a.M, m_beg + a.rows_per_wg);
    const int n_steps = (max(m_end - m_beg, 0) + 15) >> 4;           // a workgroup step = 16 rows: 4 per wave
    const __amdgpu_buffer_rsrc_t r_g = buf_rsrc(a.G), r_y = buf_rsrc(a.Y), r_s = buf_rsrc(a.state), r_a = buf_rsrc(a.agg), r_c = buf_rsrc(a.xc);
    f32x4 acc[RT][SQ];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < SQ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mu_s[SQ], mu_a[SQ], mu_c[2];                 // the means of this lane's columns (0 without BatchNormalization, for the line's 1 and its padding)
#pragma unroll
    for (int e = 0; e < SQ; ++e) { mu_s[e] = a.mean ? a.mean[a.wrow_state + SQ * c + e] : 0.0f; mu_a[e] = a.mean ? a.mean[a.wrow_agg + SQ * c + e] : 0.0f; }
#pragma unroll
    for (int e = 0; e < 2; ++e) { const int wr = wgrad_wrow(a, S, 2 * S + 2 * c + e); mu_c[e] = (a.mean && wr >= 0) ? a.mean[wr] : 0.0f; }
    struct Step { Piece<SQ> xs, xa, gz, y; Piece<2> xc; };
    Step buf[PD];
    auto fetch = [&](Step &b, int s) {
        const int row = m_beg + 16 * s + 4 * wave + g;
        const bool ok = s < n_steps && row < m_end;
        const unsigned off = ok ? ((unsigned)row * (unsigned)S + (unsigned)(SQ * c)) * 4u : BUF_OFF;
        b.gz = ld_piece<SQ>(r_g, off); b.y = ld_piece<SQ>(r_y, off);
        b.xs = ld_piece<SQ>(r_s, off); b.xa = ld_piece<SQ>(r_a, off);
        b.xc = ld_piece<2>(r_c, ok ? ((unsigned)row * 32u + 2u * c) * 4u : BUF_OFF);
    };
#pragma unroll
    for (int u = 0; u < PD; ++u) fetch(buf[u], u);
#pragma unroll 1
    for (int s0 = 0; s0 < n_steps; s0 += PD) {
#pragma unroll
        for (int u = 0; u < PD; ++u) {
            if (s0 + u < n_steps) {
                Step &b = buf[u];
                float dz[SQ];
#pragma unroll
                for (int j = 0; j < SQ; ++j) dz[j] = b.gz.v[j] * activate_grad_from_output(a.act, b.y.v[j]);
#pragma unroll
                for (int e = 0; e < SQ; ++e)
#pragma unroll
                    for (int j = 0; j < SQ; ++j) {
                        acc[e][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.xs.v[e] - mu_s[e], dz[j], acc[e][j], 0, 0, 0);
                        acc[SQ + e][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.xa.v[e] - mu_a[e], dz[j], acc[SQ + e][j], 0, 0, 0);
                    }
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int j = 0; j < SQ; ++j) acc[2 * SQ + e][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.xc.v[e] - mu_c[e], dz[j], acc[2 * SQ + e][j], 0, 0, 0);
                fetch(b, s0 + u + PD);
            }
        }
    }
    TB_MFMA_DRAIN();
    // acc[rt][j][i] = P[virtual column kv(rt, 4 g + i)][SQ c + j]; the waves add their tiles in wave order
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = 4 * g + i;
                    const int kv = rt < SQ ? SQ * m + rt : rt < 2 * SQ ? S + SQ * m + (rt - SQ) : 2 * S + 2 * m + (rt - 2 * SQ);
                    float *dst = Ps + kv * S + SQ * c;
#pragma unroll
                    for (int j = 0; j < SQ; ++j) dst[j] = w == 0 ? acc[rt][j][i] : dst[j] + acc[rt][j][i];
                }
        }
        __syncthreads();
    }
    float *Pp = a.part + (size_t)blockIdx.x * ((size_t)a.K * S + S);
    for (int i = tid; i < KV * S; i += 256) {
        const int kv = i / S, h = i % S;
        int wrow = -1;
        if (kv < S) wrow = a.wrow_state + kv;
        else if (kv < 2 * S) wrow = a.wrow_agg + (kv - S);
        else {
            int jj = kv - 2 * S, b0 = 0;
#pragma unroll
            for (int sg = 0; sg < 3; ++sg) { if (sg < a.cs.n && jj >= b0 && jj < b0 + a.cs.width[sg]) wrow = a.cs.wrow[sg] + (jj - b0); if (sg < a.cs.n) b0 += a.cs.width[sg]; }
            if (jj == a.Kc) wrow = a.K;                              // the line's 1: q
        }
        if (wrow >= 0) Pp[(size_t)wrow * S + h] = Ps[i];
    }
}

// The 32x32 tiles of a workgroup's four waves -> its partial P (LDS, waves in order) -> a.part, by weight row (shared by the f32 and the
// bf16-split forms of the kernel: v_mfma_f32_32x32x2_f32 and v_mfma_f32_32x32x16_bf16 leave their results in the same registers).
template <int NB, int XT = 1>                 // XT: 32-column tiles of the constants line (1: the 128-byte line; 2: 256 bytes, k_train_wgrad_b6<.., XT = 2>)
__device__ __forceinline__ void wgrad32_store(const TrainWgradArgs &a, f32x16 (&acc)[2 * NB + XT][NB], float *Ps, const int bid) {
    constexpr int S = 32 * NB, RT = 2 * NB + XT, KV = 2 * S + 32 * XT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kk = lane >> 5;
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    // acc[rt][f][v] = P[virtual column kv(rt, 8 (v / 4) + 4 kk + v % 4)][NB i + f]; the waves add their tiles in wave order
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int m = 8 * (v >> 2) + 4 * kk + (v & 3);
                    const int kv = rt < NB ? NB * m + rt : rt < 2 * NB ? S + NB * m + (rt - NB) : 2 * S + 32 * (rt - 2 * NB) + m;
                    float *dst = Ps + kv * S + NB * i;
#pragma unroll
                    for (int f = 0; f < NB; ++f) dst[f] = w == 0 ? acc[rt][f][v] : dst[f] + acc[rt][f][v];
                }
        }
        __syncthreads();
    }
    float *Pp = a.part + (size_t)bid * ((size_t)a.K * S + S);
    for (int idx = tid; idx < KV * S; idx += 256) {
        const int kv = idx / S, h = idx % S;
        int wrow = -1;
        if (kv < S) wrow = a.wrow_state + kv;
        else if (kv < 2 * S) wrow = a.wrow_agg + (kv - S);
        else {
            int jj = kv - 2 * S, b0 = 0;
#pragma unroll
            for (int sg = 0; sg < 3; ++sg) { if (sg < a.cs.n && jj >= b0 && jj < b0 + a.cs.width[sg]) wrow = a.cs.wrow[sg] + (jj - b0); if (sg < a.cs.n) b0 += a.cs.width[sg]; }
            if (jj == a.Kc) wrow = a.K;                              // the line's 1: q
        }
        if (wrow >= 0) Pp[(size_t)wrow * S + h] = Ps[idx];
    }
}

// The same contraction on v_mfma_f32_32x32x2_f32 (S = 32 NB): 156 TFLOP/s sustained against 104 .. 126 for the 16x16x4 form
// (scripts/micro/mfma_peak.hip), and the weight gradient is the one dense training kernel that is almost all MFMA.  Lane (i = lane % 32,
// kk = lane / 32) supplies one element of row r0 + kk of each operand: it loads the NB consecutive floats  X[r0 + kk][NB i ..]  (a
// wave's load = two whole rows) and uses value e as the A operand of row tile e (tile row i is input column NB i + e); dZ pieces serve
// as B operands the same way (tile column j is dZ column NB j + f).  A workgroup step is 8 rows, PD steps of loads in flight.
#ifndef TB_WG32_PD
#define TB_WG32_PD 6
#define TB_WG32_WAVES 2
#endif
template <int NB, int ACT>
__global__ void __launch_bounds__(256, TB_WG32_WAVES) k_train_wgrad32(TrainWgradArgs a) {
    constexpr int S = 32 * NB, RT = 2 * NB + 1, KV = 2 * S + 32, PD = TB_WG32_PD;
    __shared__ float Ps[KV * S];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kk = lane >> 5;
    const int m_beg = blockIdx.x * a.rows_per_wg, m_end = min(a.M, m_beg + a.rows_per_wg);
    const int n_steps = (max(m_end - m_beg, 0) + 7) >> 3;            // a workgroup step = 8 rows: 2 per wave
    const __amdgpu_buffer_rsrc_t r_g = buf_rsrc(a.G), r_y = buf_rsrc(a.Y), r_s = buf_rsrc(a.state), r_a = buf_rsrc(a.agg), r_c = buf_rsrc(a.xc);
    f32x16 acc[RT][NB];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int f = 0; f < NB; ++f)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[rt][f][v] = 0.0f;
    float mu_s[NB], mu_a[NB], mu_c;                     // the means of this lane's columns (0 without BatchNormalization, for the line's 1 and its padding)
#pragma unroll
    for (int e = 0; e < NB; ++e) { mu_s[e] = a.mean ? a.mean[a.wrow_state + NB * i + e] : 0.0f; mu_a[e] = a.mean ? a.mean[a.wrow_agg + NB * i + e] : 0.0f; }
    { const int wr = wgrad_wrow(a, S, 2 * S + i); mu_c = (a.mean && wr >= 0) ? a.mean[wr] : 0.0f; }
    struct Step { Piece<NB> xs, xa, gz, y; Piece<1> xc; };
    Step buf[PD];
    auto fetch = [&](Step &b, int s) {
        const int row = m_beg + 8 * s + 2 * wave + kk;
        const bool ok = s < n_steps && row < m_end;
        const unsigned off = ok ? ((unsigned)row * (unsigned)S + (unsigned)(NB * i)) * 4u : BUF_OFF;
        b.gz = ld_piece<NB>(r_g, off); b.y = ld_piece<NB>(r_y, off);
        b.xs = ld_piece<NB>(r_s, off); b.xa = ld_piece<NB>(r_a, off);
        b.xc = ld_piece<1>(r_c, ok ? ((unsigned)row * 32u + (unsigned)i) * 4u : BUF_OFF);
    };
#pragma unroll
    for (int u = 0; u < PD; ++u) fetch(buf[u], u);
#pragma unroll 1
    for (int s0 = 0; s0 < n_steps; s0 += PD) {
#pragma unroll
        for (int u = 0; u < PD; ++u) {
            {   // (no branch on s0 + u < n_steps, no switch on the activation: rows past the end load zeros, and hipcc drains the whole
                //  load queue - s_waitcnt vmcnt(0) - wherever two blocks of the loop body meet)
                Step &b = buf[u];
                float dz[NB];
#pragma unroll
                for (int f = 0; f < NB; ++f) dz[f] = b.gz.v[f] * activate_grad1<ACT>(b.y.v[f]);
#pragma unroll
                for (int e = 0; e < NB; ++e)
#pragma unroll
                    for (int f = 0; f < NB; ++f) {
                        acc[e][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.xs.v[e] - mu_s[e], dz[f], acc[e][f], 0, 0, 0);
                        acc[NB + e][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.xa.v[e] - mu_a[e], dz[f], acc[NB + e][f], 0, 0, 0);
                    }
#pragma unroll
                for (int f = 0; f < NB; ++f) acc[2 * NB][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.xc.v[0] - mu_c, dz[f], acc[2 * NB][f], 0, 0, 0);
                fetch(b, s0 + u + PD);
            }
        }
    }
    wgrad32_store<NB>(a, acc, Ps, blockIdx.x);
}

template <int J0, int J1, typename F> __device__ __forceinline__ void static_for(F &&f) {       // f(integral_constant<int, J0>) ... : loop indices that are constant expressions
    if constexpr (J0 < J1) { f(std::integral_constant<int, J0>{}); static_for<J0 + 1, J1>(f); }
}
// NB floats at LDS byte address addr + OFF, as inline assembly (hipcc neither sees the read nor waits for it: the caller does)
template <int NB, int OFF> __device__ __forceinline__ void lds_read_piece(Piece<NB> &p, unsigned addr) {
    if constexpr (NB == 2) { f32x2 v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF)); p.v[0] = v[0]; p.v[1] = v[1]; }
    else { float v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF)); p.v[0] = v; }
}

template <int OFF, typename V4> __device__ __forceinline__ void lds_read_b128(V4 &v, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF)); }
template <typename T> __device__ __forceinline__ void asm_tie(T &x) { asm volatile("" : "+v"(x)); }       // no instruction: orders the uses of x behind the volatile statements before it
template <int OFF> __device__ __forceinline__ void lds_read_f32(float &v, unsigned addr) { asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF)); }

// ... and on the bf16 matrix cores (round 5).  scripts/micro/mfma_bf16_valu_overlap.hip: on gfx950 a SIMD issues EITHER a matrix instruction
// OR a VALU instruction - a wave's own VALU work does not run in the shadow of its MFMAs, nor does the other wave's (24 MFMAs + 96 v_fma a
// trip: 516 ns in phases, 533 interleaved, 589 on specialised waves, against 336 + 212 alone) - so a dense kernel costs the SUM of its
// matrix and VALU cycles, and k_train_wgrad32 is 64 cycles of v_mfma_f32_32x32x2_f32 for every 2 rows of every 32 x 32 tile: 5 120 cycles
// per 16 rows of S = 64.  v_mfma_f32_32x32x16_bf16 contracts 16 rows in 32 cycles: with both operands split into three bf16 terms
// (split3_pk: exact, the products' sum is an f32 chain's to 2^-24) the same 16 rows cost 60 x 32 = 1 920 matrix cycles + ~1 400 of VALU.
// Lane (i = lane % 32, kg = lane / 32) supplies 8 consecutive k of a tile row: the NB floats X[r0 + 8 kg + j][NB i ..] of EIGHT rows (j = 0
// .. 7), paired along the rows; dZ pieces are B operands the same way, and the results land where v_mfma_f32_32x32x2_f32 leaves them.
//
// The rows reach the wave through a ring in LDS that the memory system fills (buffer_load_dwordx4 ... lds: no destination registers).  A
// wave's step is 16 rows = one contiguous 2 NB KB piece of each [M, S] array: 2 NB wave-loads of 1 KB land it in LDS as it lies in
// memory, the operands are then read with ds_read_b64.  Two slots a wave (rows and constants line in rings of their own): at the top of
// step s the wave waits for slot s % 2 (filled two steps ago), reads it into registers, hands the slot back to the loads of step s + 2 and
// computes - 28 KB a wave in flight for two whole steps, and nothing that is in flight lives in a register across the loop's back edge (the first form of this kernel kept
// two steps of rows in registers: hipcc loaded them into other registers than the loop carries, copied them at the loop's end and
// waited for every load there - 177 us = 98 of loads + 39 of splits + 40 of products, one after the other).  The waits are counted by
// hand (s_waitcnt vmcnt(NG): every step issues exactly NG loads, past the end of the workgroup's rows too - those touch no memory and
// fill zeros), the LDS reads are inline assembly so that hipcc does not drain the ring in front of them.
// 160 accumulator registers: one wave per SIMD, one workgroup per CU.
template <int NB, int ACT, int XT>           // XT = 2: a constants line of 64 floats (k_pack_xc_pos<64>: 32 .. 63 constant inputs - heterogeneous models)
__device__ __forceinline__ void train_wgrad_b6_body(const TrainWgradArgs &a, const int bid) {
    constexpr int S = 32 * NB, RT = 2 * NB + XT;
    constexpr bool HAS_Y = ACT != GNN_ACT_LINEAR;
    constexpr int ARR = 16 * S * 4, NA = HAS_Y ? 4 : 3;        // bytes of 16 rows of an [M, S] array; arrays in a slot: dZ | state | agg | (Y)
    constexpr int LINE_B = 128 * XT;                           // bytes of a row of the constants line
    constexpr int MAIN = NA * ARR, NGM = MAIN / 1024, XCB = 16 * LINE_B, NGX = XCB / 1024;      // a slot of rows and its LDS-DMA instructions; 16 rows of the constants line
    constexpr int D = 2, XR = 2;                               // ring depths: slots of rows / of the line a wave.  (D = 3 - the CU's whole 160 KB at S = 64 - measured
                                                               //  185 us against 177 - 183 at 1 M rows: the LDS-DMA stream is not short of bytes in flight)
    constexpr int WAVE_B = D * MAIN + XR * XCB;
    extern __shared__ __attribute__((aligned(16))) float tb_smem[];     // [4 waves][D slots | XR line slots]; the workgroup's partial P afterwards
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // (uniform: the ring's addresses stay in scalar registers)
    const int i = lane & 31, kg = lane >> 5;
    const int m_beg = bid * a.rows_per_wg, m_end = min(a.M, m_beg + a.rows_per_wg);
    const int rows = max(m_end - m_beg, 0);
    const int n_steps = (rows + 63) >> 6;               // a workgroup step = 64 rows: 16 per wave
    // windows of exactly this workgroup's rows: what lies past them reads 0
    const size_t o_rows = (size_t)m_beg * S;
    const unsigned win = (unsigned)rows * (unsigned)S * 4u;
    const __amdgpu_buffer_rsrc_t r_g = buf_rsrc_n(a.G + o_rows, win), r_y = buf_rsrc_n(a.Y ? a.Y + o_rows : nullptr, win), r_s = buf_rsrc_n(a.state + o_rows, win),
                                 r_a = buf_rsrc_n(a.agg + o_rows, win), r_c = buf_rsrc_n(a.xc ? a.xc + (size_t)m_beg * (32 * XT) : nullptr, (unsigned)rows * (unsigned)LINE_B);
    f32x16 acc[RT][NB];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int f = 0; f < NB; ++f)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[rt][f][v] = 0.0f;
    float mu_s[NB], mu_a[NB], mu_c[XT];
#pragma unroll
    for (int e = 0; e < NB; ++e) { mu_s[e] = a.mean ? a.mean[a.wrow_state + NB * i + e] : 0.0f; mu_a[e] = a.mean ? a.mean[a.wrow_agg + NB * i + e] : 0.0f; }
#pragma unroll
    for (int x = 0; x < XT; ++x) { const int wr = wgrad_wrow(a, S, 2 * S + 32 * x + i); mu_c[x] = (a.mean && wr >= 0) ? a.mean[wr] : 0.0f; }
    typedef __attribute__((address_space(3))) char lds_char;
    lds_char *ring = (lds_char *)tb_smem + wave * WAVE_B;
    const unsigned ring_addr = (unsigned)(size_t)ring;
    auto fill_main = [&](int slot, int s) {                 // the 16 rows of step s of this wave -> slot
        lds_char *dst = ring + slot * MAIN;
        const unsigned off = ((unsigned)(64 * s + 16 * wave) * (unsigned)S) * 4u + 16u * (unsigned)lane;
#pragma unroll
        for (int q = 0; q < ARR / 1024; ++q) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_g, (__attribute__((address_space(3))) void *)(dst + 1024 * q), 16, (int)(off + 1024u * q), 0, 0, TB_RING_AUX);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_s, (__attribute__((address_space(3))) void *)(dst + ARR + 1024 * q), 16, (int)(off + 1024u * q), 0, 0, TB_RING_AUX);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_a, (__attribute__((address_space(3))) void *)(dst + 2 * ARR + 1024 * q), 16, (int)(off + 1024u * q), 0, 0, TB_RING_AUX);
            if (HAS_Y) __builtin_amdgcn_raw_ptr_buffer_load_lds(r_y, (__attribute__((address_space(3))) void *)(dst + 3 * ARR + 1024 * q), 16, (int)(off + 1024u * q), 0, 0, TB_RING_AUX);
        }
    };
    auto fill_xc = [&](int xslot, int s) {                  // ... and of the constants line
        lds_char *dst = ring + D * MAIN + xslot * XCB;
        const unsigned off_c = (unsigned)(64 * s + 16 * wave) * (unsigned)LINE_B + 16u * (unsigned)lane;
#pragma unroll
        for (int q = 0; q < NGX; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_c, (__attribute__((address_space(3))) void *)(dst + 1024 * q), 16, (int)(off_c + 1024u * q), 0, 0, TB_RING_AUX);
    };
    struct Step { Piece<NB> gz[8], xs[8], xa[8], y[8]; Piece<1> xc[XT][8]; };
    const unsigned lane_addr = ring_addr + (unsigned)(8 * kg) * (unsigned)(S * 4) + (unsigned)(NB * i) * 4u, lane_addr_c = ring_addr + D * MAIN + (unsigned)(8 * kg) * (unsigned)LINE_B + (unsigned)i * 4u;
#define B8(v_) __builtin_bit_cast(bf16x8, v_)
    auto split8 = [&](const float (&x)[8], u32x4 &h, u32x4 &m, u32x4 &l) {
        unsigned hh[4], mm[4], ll[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (TB_ABL & 64) { hh[q] = __float_as_uint(x[2 * q]); mm[q] = __float_as_uint(x[2 * q + 1]); ll[q] = hh[q] ^ mm[q]; }      // (ablation: no split)
            else split3_pk((f32x2){x[2 * q], x[2 * q + 1]}, hh[q], mm[q], ll[q]);
        }
        h = (u32x4){hh[0], hh[1], hh[2], hh[3]}; m = (u32x4){mm[0], mm[1], mm[2], mm[3]}; l = (u32x4){ll[0], ll[1], ll[2], ll[3]};
    };
    // Issue order: [line of s + 2 | rows of s + D] at the top of step s, the prologue in the same order - so that at the top of a step the
    // (D - 1) NGM + (XR - 1) NGX youngest loads are exactly the ones the step does not need (loads retire in issue order).
    fill_xc(0, 0); fill_main(0, 0);
#pragma unroll
    for (int d = 1; d < D; ++d) { if (d < XR) fill_xc(d, d); fill_main(d, d); }
    int slot = 0;
#pragma unroll 1
    for (int s = 0; s < n_steps; ++s) {
        const int xslot = s & 1;
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((D - 1) * NGM + (XR - 1) * NGX) : "memory");       // the step's rows and line have landed
        Step b;
        {   // (inline assembly: see above; constant offsets in the instruction - the caller waits lgkmcnt(0))
            unsigned base = lane_addr + (unsigned)slot * MAIN, base_c = lane_addr_c + (unsigned)xslot * XCB;
            static_for<0, 8>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                lds_read_piece<NB, j * S * 4>(b.gz[j], base); lds_read_piece<NB, ARR + j * S * 4>(b.xs[j], base); lds_read_piece<NB, 2 * ARR + j * S * 4>(b.xa[j], base);
                if (HAS_Y) lds_read_piece<NB, 3 * ARR + j * S * 4>(b.y[j], base);
                lds_read_piece<1, j * LINE_B>(b.xc[0][j], base_c);
                if constexpr (XT > 1) lds_read_piece<1, j * LINE_B + 128>(b.xc[XT - 1][j], base_c);
            });
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                      // the slot is in registers: hand it to the loads of step s + 2
#pragma unroll
        for (int j = 0; j < 8; ++j) {             // (no instruction: every use of a value that was read comes after the wait - volatile statements keep their order)
#pragma unroll
            for (int e = 0; e < NB; ++e) { asm volatile("" : "+v"(b.gz[j].v[e])); asm volatile("" : "+v"(b.xs[j].v[e])); asm volatile("" : "+v"(b.xa[j].v[e])); if (HAS_Y) asm volatile("" : "+v"(b.y[j].v[e])); }
#pragma unroll
            for (int x = 0; x < XT; ++x) asm volatile("" : "+v"(b.xc[x][j].v[0]));
        }
        fill_xc(xslot, s + XR); fill_main(slot, s + D);
        slot = slot + 1 == D ? 0 : slot + 1;
        u32x4 zh[NB], zm[NB], zl[NB];
#pragma unroll
        for (int f = 0; f < NB; ++f) {
            float dz[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) dz[j] = HAS_Y ? b.gz[j].v[f] * activate_grad1<ACT>(b.y[j].v[f]) : b.gz[j].v[f];
            split8(dz, zh[f], zm[f], zl[f]);
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = rt < NB ? b.xs[j].v[rt < NB ? rt : 0] - mu_s[rt < NB ? rt : 0]
                                             : rt < 2 * NB ? b.xa[j].v[rt < 2 * NB && rt >= NB ? rt - NB : 0] - mu_a[rt < 2 * NB && rt >= NB ? rt - NB : 0]
                                             : b.xc[rt >= 2 * NB ? rt - 2 * NB : 0][j].v[0] - mu_c[rt >= 2 * NB ? rt - 2 * NB : 0];
            u32x4 xh, xm, xl;
            split8(x, xh, xm, xl);
            // small terms first; consecutive MFMAs go to different accumulators
#define MF(xp_, zp_) _Pragma("unroll") for (int f = 0; f < NB; ++f) acc[rt][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B8(xp_), B8(zp_[f]), acc[rt][f], 0, 0, 0)
            if (TB_ABL & 32) {           // (ablation of scripts/micro/rowgemm_bench.hip: no products)
#pragma unroll
                for (int f = 0; f < NB; ++f) acc[rt][f][0] += __uint_as_float(xl[0] ^ xm[1] ^ xh[2] ^ zh[f][3] ^ zm[f][0] ^ zl[f][1]);
            } else {
                MF(xl, zh); MF(xh, zl); MF(xm, zm); MF(xm, zh); MF(xh, zm); MF(xh, zh);
            }
#undef MF
        }
    }
#undef B8
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (the fills past the end)
    __syncthreads();                                        // every wave is done with its ring: the partial P takes its place
    wgrad32_store<NB, XT>(a, acc, tb_smem, bid);
}
template <int NB, int ACT, int XT = 1>
__global__ void __launch_bounds__(256, 1) k_train_wgrad_b6(TrainWgradArgs a) { train_wgrad_b6_body<NB, ACT, XT>(a, blockIdx.x); }
template <int NB, int ACT, int XT>           // every node type's rows in one launch (TypeLaunch; the partials of type t: its own `part`)
__global__ void __launch_bounds__(256, 1) k_train_wgrad_b6_types(TypeLaunch<TrainWgradArgs> m) {
    int bid, nblk;
    const TrainWgradArgs a = select_type(m, bid, nblk);
    train_wgrad_b6_body<NB, ACT, XT>(a, bid);
}
template <int NB, int ACT, int XT = 1>
inline size_t train_wgrad_b6_lds() {
    const size_t S = 32 * NB, na = ACT != GNN_ACT_LINEAR ? 4 : 3, ring = 4 * (2 * na * 16 * S * 4 + 2 * 2048 * XT), P = (2 * S + 32 * XT) * S * 4;
    return ring > P ? ring : P;
}

// ---- weight gradient AND input gradient of an iteration in one pass over its rows (round 5) -------------------------------------------------
// k_train_wgrad_b6 reads dZ | state | agg | constants (896 bytes a row at S = 64), k_train_bwd_dx_b6 reads dZ | agg again and writes dx (1 024
// bytes): 1 920 bytes a row and iteration, both kernels waiting on memory.  Here the rows a wave has in its LDS ring serve both products:
//   P  += [state | agg | constants]^T dZ      (v_mfma_f32_32x32x16_bf16, as k_train_wgrad_b6: lane (i, kg), operands paired along the rows)
//   dx  = BatchNorm'( dZ W^T )                (v_mfma_f32_16x16x32_bf16, as k_train_bwd_dx_b6: lane (c, g) = row c, W^T's bf16 planes in LDS)
// 1 412 bytes a row.  The dZ form only (TrainWgradArgs::Y == NULL; the input gradient does not read the state rows), and the launcher
// uses it for networks WITHOUT BatchNormalization: with it, m1 / m2 of the input gradient are column moments that come out of this very
// iteration's finished P and q (train_loop.hpp) - the kernel takes them as given (TrainBwdArgs::defer_state_bn form).  The ring's image of an [16, S] piece is XOR-swizzled by the row (position of 16-byte piece p of row r:
// r PPR + (p ^ r % PPR); LDS-DMA places a wave-load's lanes one after the other, so the SOURCE addresses carry the permutation): the
// row-per-lane reads of the second product would otherwise hit one bank group 16 times.  The constants line has one slot (filled one
// step ahead, issued BEFORE the rows of step s + 2 so that vmcnt(NGM) covers it), everything else two.  Every LDS read of the loop is
// inline assembly (hipcc puts vmcnt(0) in front of an LDS read it can see while LDS-DMA is in flight), the counts are in the code.
template <int NB>
__global__ void __launch_bounds__(256, 1) k_train_wgrad_dx_b6(TrainWgradArgs a, TrainBwdArgs ba) {
    constexpr int S = 32 * NB, RT = 2 * NB + 1, HQ = 2 * NB, NCT = 2 * HQ, HP = 16 * NCT, NKB = NB, PPR = S / 4;
    constexpr int PLANE_E = NKB * NCT * 64 * 8, PLANE_B = 2 * PLANE_E;         // elements / bytes of one bf16 plane of W^T
    constexpr int W_B = 3 * PLANE_B, COEF_B = 4 * HP * 4;
    constexpr int ARR = 16 * S * 4, NQ = ARR / 1024;                           // 16 rows of an [M, S] array; wave-loads that bring them
    constexpr int MAIN = 3 * ARR + 256, XCB = 2048, WAVE_B = 2 * MAIN + XCB;   // slot: dZ | state | agg | row scales;  a wave's ring: slot 0 | slot 1 | constants line
    constexpr int NGM = 3 * NQ + 1;                                            // LDS-DMA instructions of a slot
    extern __shared__ __attribute__((aligned(16))) float tb_smem[];
    typedef __attribute__((address_space(3))) char lds_char;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, kg = lane >> 5;            // first product
    const int c = lane & 15, g = lane >> 4;             // second product
    unsigned short *Wl = reinterpret_cast<unsigned short *>(tb_smem);
    float *coef = tb_smem + W_B / 4;
    {   // W^T's three bf16 planes, k_train_bwd_dx_b6's layout (eight independent loads a batch)
        const __amdgpu_buffer_rsrc_t r_w = buf_rsrc(ba.W);
        constexpr int TOT = 32 * NKB * HP, BATCH = 8;
        static_assert(TOT % (256 * BATCH) == 0, "weight staging: whole batches");
#pragma unroll 1
        for (int i0 = tid; i0 < TOT; i0 += 256 * BATCH) {
            float v[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int idx = i0 + u * 256, k = idx / HP, j = idx % HP;
                const int row = j < S ? ba.wrow_state + j : ba.wrow_agg + (j - S);
                v[u] = buf_ld_f32(r_w, k < ba.H ? ((unsigned)row * (unsigned)ba.ldw + (unsigned)k) * 4u : BUF_OFF);
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int idx = i0 + u * 256, k = idx / HP, j = idx % HP;
                const int q = k >> 4, gg = (k >> 2) & 3, e = k & 3;
                split3_store(Wl, PLANE_E, ((((q >> 1) * NCT + (j >> 4)) * 64 + 16 * gg + (j & 15)) * 8) + 4 * (q & 1) + e, v[u]);
            }
        }
    }
    for (int j = tid; j < HP; j += 256) {
        float Ac = 1.0f, Cc = 0.0f, M1 = 0.0f, Mu = 0.0f;
        if (ba.gamma) {
            const int k = j < S ? ba.wrow_state + j : ba.wrow_agg + (j - S);
            const float rstd = 1.0f / sqrtf(ba.var[k] + ba.eps);
            Ac = ba.gamma[k] * rstd; Cc = -Ac * rstd * ba.m2[k]; M1 = ba.m1[k]; Mu = ba.mean[k];
            if (j < S) { Cc = 0.0f; M1 = 0.0f; Mu = 0.0f; }                   // (TrainBwdArgs::defer_state_bn)
        }
        coef[j] = Ac; coef[HP + j] = Cc; coef[2 * HP + j] = M1; coef[3 * HP + j] = Mu;
    }
    const int m_beg = blockIdx.x * a.rows_per_wg, m_end = min(a.M, m_beg + a.rows_per_wg);
    const int rows = max(m_end - m_beg, 0);
    const int n_steps = (rows + 63) >> 6;
    const size_t o_rows = (size_t)m_beg * S;
    const unsigned win = (unsigned)rows * (unsigned)S * 4u;
    const __amdgpu_buffer_rsrc_t r_g = buf_rsrc_n(a.G + o_rows, win), r_s = buf_rsrc_n(a.state + o_rows, win), r_a = buf_rsrc_n(a.agg + o_rows, win),
                                 r_c = buf_rsrc_n(a.xc ? a.xc + (size_t)m_beg * 32 : nullptr, (unsigned)rows * 128u),
                                 r_rs = buf_rsrc_n(ba.agg_row_scale ? ba.agg_row_scale + m_beg : nullptr, (unsigned)rows * 4u),
                                 r_o = buf_rsrc_n(ba.dx + (size_t)m_beg * ba.ld_dx, (unsigned)rows * (unsigned)ba.ld_dx * 4u);
    const bool has_rs = ba.agg_row_scale != nullptr;
    f32x16 acc[RT][NB];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int f = 0; f < NB; ++f)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[rt][f][v] = 0.0f;
    float mu_s[NB], mu_a[NB], mu_c;
#pragma unroll
    for (int e = 0; e < NB; ++e) { mu_s[e] = a.mean ? a.mean[a.wrow_state + NB * i + e] : 0.0f; mu_a[e] = a.mean ? a.mean[a.wrow_agg + NB * i + e] : 0.0f; }
    { const int wr = wgrad_wrow(a, S, 2 * S + i); mu_c = (a.mean && wr >= 0) ? a.mean[wr] : 0.0f; }
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)(lds_char *)tb_smem;
    lds_char *ring = (lds_char *)tb_smem + W_B + COEF_B + wave * WAVE_B;
    const unsigned ring_addr = lds0 + W_B + COEF_B + (unsigned)wave * WAVE_B;
    // source offsets of this lane in wave-load q of a 16-row piece (the swizzle), relative to the piece's first byte
    unsigned src_q[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { const int pos = 64 * q + lane, r = pos / PPR, sw = pos % PPR; src_q[q] = (unsigned)(r * (S * 4) + ((sw ^ (r & (PPR - 1))) << 4)); }
    auto fill_main = [&](int slot, int s) {
        lds_char *dst = ring + slot * MAIN;
        const unsigned row0 = (unsigned)(64 * s + 16 * wave), base = row0 * (unsigned)(S * 4);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_g, (__attribute__((address_space(3))) void *)(dst + 1024 * q), 16, (int)(base + src_q[q]), 0, 0, TB_RING_AUX);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_s, (__attribute__((address_space(3))) void *)(dst + ARR + 1024 * q), 16, (int)(base + src_q[q]), 0, 0, TB_RING_AUX);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_a, (__attribute__((address_space(3))) void *)(dst + 2 * ARR + 1024 * q), 16, (int)(base + src_q[q]), 0, 0, TB_RING_AUX);
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_rs, (__attribute__((address_space(3))) void *)(dst + 3 * ARR), 4, (int)((row0 + (unsigned)lane) * 4u), 0, 0, 0);
    };
    auto fill_xc = [&](int s) {
        lds_char *dst = ring + 2 * MAIN;
        const unsigned off_c = (unsigned)(64 * s + 16 * wave) * 128u + 16u * (unsigned)lane;
#pragma unroll
        for (int q = 0; q < 2; ++q) __builtin_amdgcn_raw_ptr_buffer_load_lds(r_c, (__attribute__((address_space(3))) void *)(dst + 1024 * q), 16, (int)(off_c + 1024u * q), 0, 0, TB_RING_AUX);
    };
    // LDS offsets of this lane's reads inside a 16-row piece: the first product's eight rows 8 kg + j, the second product's pieces 4 q + g of row c
    unsigned rd1[8], rd2[HQ];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int r = 8 * kg + j, p = (NB * i) / 4; rd1[j] = (unsigned)((r * PPR + (p ^ (r & (PPR - 1)))) * 16 + ((NB * i) & 3) * 4); }
#pragma unroll
    for (int q = 0; q < HQ; ++q) rd2[q] = (unsigned)((c * PPR + ((4 * q + g) ^ (c & (PPR - 1)))) * 16);
    const unsigned xc_addr = ring_addr + 2 * MAIN + (unsigned)(8 * kg) * 128u + (unsigned)i * 4u;
    const unsigned w_addr = lds0 + (unsigned)lane * 16u, coef_addr = lds0 + W_B + (unsigned)g * 16u;
#define B8(v_) __builtin_bit_cast(bf16x8, v_)
    auto split8 = [&](const float (&x)[8], u32x4 &h, u32x4 &m, u32x4 &l) {
        unsigned hh[4], mm[4], ll[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) split3_pk((f32x2){x[2 * q], x[2 * q + 1]}, hh[q], mm[q], ll[q]);
        h = (u32x4){hh[0], hh[1], hh[2], hh[3]}; m = (u32x4){mm[0], mm[1], mm[2], mm[3]}; l = (u32x4){ll[0], ll[1], ll[2], ll[3]};
    };
    // The wait at the top of a step: loads retire in issue order AMONG LOADS, stores among stores, but a store may retire before an older
    // load (a first version counted "[rows of s + 1 | the NCT stores of step s - 1] may be out" - vmcnt(NGM + NCT) - and read a constants
    // line that had not landed whenever the stores overtook it: wrong constants rows of P in one run of three).  vmcnt(NGM) holds either
    // way: at most NGM operations out means at most NGM LOADS out, and those are the youngest ones - the rows of step s + 1.
    fill_main(0, 0); fill_xc(0); fill_main(1, 1);
#pragma unroll 1
    for (int s = 0; s < n_steps; ++s) {
        const int slot = s & 1;
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NGM) : "memory");            // slot and the constants line have landed
        Piece<NB> gz[8], xs[8], xa[8]; Piece<1> xc[8];
        f32x4 zq[HQ], xq[HQ]; float rs;
        {
            unsigned sb = ring_addr + (unsigned)slot * MAIN, xb = xc_addr;
            static_for<0, 8>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                lds_read_piece<NB, 0>(gz[j], sb + rd1[j]); lds_read_piece<NB, ARR>(xs[j], sb + rd1[j]); lds_read_piece<NB, 2 * ARR>(xa[j], sb + rd1[j]);
                lds_read_piece<1, j * 128>(xc[j], xb);
            });
            static_for<0, HQ>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                lds_read_b128<0>(zq[q], sb + rd2[q]); lds_read_b128<2 * ARR>(xq[q], sb + rd2[q]);
            });
            lds_read_f32<3 * ARR>(rs, sb + (unsigned)c * 4u);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                      // the slot is in registers: hand it to the loads of step s + 2
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int e = 0; e < NB; ++e) { asm volatile("" : "+v"(gz[j].v[e])); asm volatile("" : "+v"(xs[j].v[e])); asm volatile("" : "+v"(xa[j].v[e])); }
            asm volatile("" : "+v"(xc[j].v[0]));
        }
#pragma unroll
        for (int q = 0; q < HQ; ++q) { asm volatile("" : "+v"(zq[q])); asm volatile("" : "+v"(xq[q])); }
        asm volatile("" : "+v"(rs));
        fill_xc(s + 1); fill_main(slot, s + 2);
        // ---- P += X^T dZ ------------------------------------------------------------------------------------------------------------------
        {
            u32x4 zh[NB], zm[NB], zl[NB];
#pragma unroll
            for (int f = 0; f < NB; ++f) {
                float dz[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) dz[j] = gz[j].v[f];
                split8(dz, zh[f], zm[f], zl[f]);
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = rt < NB ? xs[j].v[rt < NB ? rt : 0] - mu_s[rt < NB ? rt : 0]
                                                 : rt < 2 * NB ? xa[j].v[rt < 2 * NB && rt >= NB ? rt - NB : 0] - mu_a[rt < 2 * NB && rt >= NB ? rt - NB : 0] : xc[j].v[0] - mu_c;
                u32x4 xh, xm, xl;
                split8(x, xh, xm, xl);
#define MF(xp_, zp_) _Pragma("unroll") for (int f = 0; f < NB; ++f) acc[rt][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B8(xp_), B8(zp_[f]), acc[rt][f], 0, 0, 0)
                MF(xl, zh); MF(xh, zl); MF(xm, zm); MF(xm, zh); MF(xh, zm); MF(xh, zh);
#undef MF
            }
        }
        // ---- dx = BatchNorm'(dZ W^T) -------------------------------------------------------------------------------------------------------
        {
            u32x4 dh[NKB], dm[NKB], dl[NKB];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) split3_x8pk(zq[2 * kb], zq[2 * kb + 1], dh[kb], dm[kb], dl[kb]);
            f32x4 dacc[NCT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) dacc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
            constexpr int CTG = 2, SPK = NCT / CTG, NS = NKB * SPK;
            u32x4 W[2][3][CTG];
            auto load_w = [&](auto stc, u32x4 (&w)[3][CTG]) {
                constexpr int st = decltype(stc)::value, kb = st / SPK, ct0 = (st % SPK) * CTG;
                unsigned wa_ = w_addr;
                static_for<0, 3>([&](auto plc) { constexpr int pl = decltype(plc)::value;
                    static_for<0, CTG>([&](auto uc) { constexpr int u = decltype(uc)::value; lds_read_b128<pl * PLANE_B + (kb * NCT + ct0 + u) * 1024>(w[pl][u], wa_); }); });
            };
            load_w(std::integral_constant<int, 0>{}, W[0]);
            static_for<0, NS>([&](auto stc) {
                constexpr int st = decltype(stc)::value, kb = st / SPK, ct0 = (st % SPK) * CTG;
                if constexpr (st + 1 < NS) { load_w(std::integral_constant<int, st + 1>{}, W[(st + 1) & 1]); asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(3 * CTG) : "memory"); }
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int u = 0; u < CTG; ++u) asm_tie(W[st & 1][pl][u]);
#define MF(pl_, x_) _Pragma("unroll") for (int u = 0; u < CTG; ++u) dacc[ct0 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B8(W[st & 1][pl_][u]), B8(x_), dacc[ct0 + u], 0, 0, 0)
                MF(2, dh[kb]); MF(0, dl[kb]); MF(1, dm[kb]); MF(1, dh[kb]); MF(0, dm[kb]); MF(0, dh[kb]);
#undef MF
            });
            const float rs_t = has_rs ? rs : 1.0f;
            const unsigned orow = ((unsigned)(64 * s + 16 * wave + c) * (unsigned)ba.ld_dx + 4u * (unsigned)g) * 4u;
            static_for<0, NCT>([&](auto ctc) {
                constexpr int ct = decltype(ctc)::value;
                f32x4 Ac, v; unsigned ca = coef_addr;
                lds_read_b128<ct * 64>(Ac, ca);
                if constexpr (ct < HQ) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); asm_tie(Ac);
                    v = Ac * dacc[ct];
                } else {
                    f32x4 Cc, M1, Mu;
                    lds_read_b128<HP * 4 + ct * 64>(Cc, ca); lds_read_b128<2 * HP * 4 + ct * 64>(M1, ca); lds_read_b128<3 * HP * 4 + ct * 64>(Mu, ca);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); asm_tie(Ac); asm_tie(Cc); asm_tie(M1); asm_tie(Mu);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(Ac[e], dacc[ct][e] - M1[e], Cc[e] * (xq[ct - HQ][e] - Mu[e])) * rs_t;
                }
                const u32x4 bits = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                __builtin_amdgcn_raw_buffer_store_b128(bits, r_o, (int)(orow + 64u * ct), 0, 0);
            });
        }
    }
#undef B8
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    wgrad32_store<NB>(a, acc, tb_smem + (W_B + COEF_B) / 4, blockIdx.x);
}
template <int NB>
inline size_t train_wgrad_dx_b6_lds() {
    const size_t S = 32 * NB, HQ = 2 * NB, NCT = 2 * HQ, fixed = 3 * (NB * NCT * 64 * 16) + 4 * 16 * NCT * 4, ring = 4 * (2 * (3 * 16 * S * 4 + 256) + 2048), P = (2 * S + 32) * S * 4;
    return fixed + (ring > P ? ring : P);
}

// ---- the transposed aggregate that leaves the PREVIOUS iteration's dZ (round 5) ---------------------------------------------------------------
// Back-propagation through iteration t ends with  G_{t-1} = dx_state + Adj . dx_agg  (k_aggregate_vec: arcs walked by source), and
// iteration t - 1 begins with  dZ_{t-1} = G_{t-1} (.) act'(Y_{t-1}),  Y_{t-1} = state_t - formed twice, by the weight-gradient kernel and by
// the input-gradient kernel, each reading G and Y (512 bytes per row each).  Here the aggregate's epilogue reads the state_t row once (it
// has the node's output row in registers anyway) and
//   * adds what k_train_bwd_dx left out of the state half's BatchNorm input gradient (TrainBwdArgs::defer_state_bn: that kernel then never
//     reads state_t):  G = dx_state' + Adj . dx_agg + Cc (state_t - mean) - Ac m1,   Ac = gamma rstd,  Cc = - Ac rstd m2  (iteration t's);
//   * writes dZ_{t-1} = G (.) act'(state_t).
// The two dense kernels of iteration t - 1 then read dZ alone (their LINEAR instances, Y = NULL): per row and iteration 256 bytes more
// here, 256 + 256 + 256 fewer there.
struct AggDzArgs {
    const float *Y; int ldy;                               // state_t [n_dst, S]
    const float *gamma, *var, *mean, *m1, *m2; float eps;  // iteration t's BatchNormalization (gamma NULL: none), by weight row
    int wrow_state;
};
template <int LPR, bool HAS_W, int ACT, bool BUF = false>
__global__ void __launch_bounds__(256, HAS_W ? 7 : 8)
k_aggregate_dz(int n_dst, const int *__restrict__ rowptr, const int *__restrict__ src, const float *__restrict__ w, const float *__restrict__ row_scale,
               const float *__restrict__ X, int ldx, float *__restrict__ out, int ldo, const float *__restrict__ addend, int ld_add, AggDzArgs z) {
    // 8 waves per SIMD or nothing: this access pattern is served by the number of waves with a gather outstanding (k_aggregate_stats lost 18 %
    // when one more register took it to 7).  So the epilogue's operands are not held across the walk: the per-column coefficients sit in
    // LDS, the node's own rows (dx_state', state_t) are requested when the walk is over - eight waves cover that trip.  (The first form kept
    // them in registers: 75 - 90 VGPRs, 5 - 6 waves per SIMD, 520 us per C4-size launch against 438 for k_aggregate_vec.)
    __shared__ __attribute__((aligned(16))) float coef[3][4 * LPR];      // Cc | - Ac m1 | mean of the state columns
    const int l4 = threadIdx.x % LPR;
    const int groups = blockDim.x / LPR;
    if (threadIdx.x < 4 * LPR) {
        float cC = 0.0f, cB = 0.0f, mu = 0.0f;
        if (z.gamma) {
            const int k = z.wrow_state + threadIdx.x;
            const float rstd = 1.0f / sqrtf(z.var[k] + z.eps), Ac = z.gamma[k] * rstd;
            cC = -Ac * rstd * z.m2[k]; cB = -Ac * z.m1[k]; mu = z.mean[k];
        }
        coef[0][threadIdx.x] = cC; coef[1][threadIdx.x] = cB; coef[2][threadIdx.x] = mu;
    }
    __syncthreads();
    for (int j = blockIdx.x * groups + threadIdx.x / LPR; j < n_dst; j += gridDim.x * groups) {
        const int beg = rowptr[j], end = rowptr[j + 1];
        f32x4 acc = gather_sum8<HAS_W, BUF>(beg, end, src, w, X, ldx, 4 * l4);                   // summed in arc order
        if (row_scale) acc *= row_scale[j];
        const f32x4 y = *reinterpret_cast<const f32x4 *>(z.Y + (size_t)j * z.ldy + 4 * l4);
        acc += *reinterpret_cast<const f32x4 *>(addend + (size_t)j * ld_add + 4 * l4);
        const f32x4 cC = *reinterpret_cast<const f32x4 *>(&coef[0][4 * l4]), cB = *reinterpret_cast<const f32x4 *>(&coef[1][4 * l4]),
                    mu = *reinterpret_cast<const f32x4 *>(&coef[2][4 * l4]);
        f32x4 dz;
#pragma unroll
        for (int e = 0; e < 4; ++e) dz[e] = (acc[e] + fmaf(cC[e], y[e] - mu[e], cB[e])) * activate_grad1<ACT>(y[e]);
        *reinterpret_cast<f32x4 *>(out + (size_t)j * ldo + 4 * l4) = dz;
    }
}

// ---- a thin output head at large M ---------------------------------------------------------------------------------------------------
// The reference's output network on a node-focused graph (GNN.py:273: net_output([state | labels][mask])) with EVERY node in the
// output set is one [BatchNormalization +] Dense((S + L) -> T) with T <= 4 classes over a million rows: its backward pass is pure
// row streaming, and on the general kernels it cost as much as an iteration of the loop (weight gradient 362 us + dZ . W^T 243 us +
// BatchNorm input gradient 239 us + scatter into the state gradient 216 us + a 256 MB memset at C4 size).  Two kernels instead:
//   k_head_wgrad<T> - P = [state | labels]^T dZ and q = colsum(dZ): a lane owns four consecutive input columns, 32 lanes a row, each row
//                     read once as 16-byte pieces; per-workgroup partials in the layout k_reduce_partials / k_first_layer_param_grads
//                     take ([K x T] then [T]), row groups and workgroups added in a fixed order;
//   k_head_dx<T>    - d loss / d state[m, j] = Ac_j (sum_h dZ[m, h] W[j, h] - m1_j) + Cc_j (state[m, j] - mean_j)  (the Dense input gradient with the
//                     BatchNorm input gradient as per-column coefficients, as k_train_bwd_dx), WRITTEN into the state gradient: every
//                     row is an output row, so nothing is scattered or zero-filled.  The label columns' input gradient is never needed.
struct HeadArgs {
    int M, S, L, T;
    const float *state; int ld_state;     // [M, S], S a multiple of 4
    const float *labels; int ld_labels;   // [M, L] or NULL (L = 0)
    const float *dZ; int ldz;             // [M, T]
    int rows_per_wg;
    float *part;                          // k_head_wgrad: [gridDim.x][(S + L) T + T]
    // k_head_dx
    const float *W;                       // first-layer kernel [(S + L) x T] (NOT folded)
    const float *gamma, *mean, *var, *m1, *m2; float eps;      // NULL gamma: no BatchNormalization
    float *dx; int ld_dx;                 // [M, S]
    int dz_act;                           // >= 0: the rows leave as dZ of the loop's LAST iteration, dx (.) act'(state) with this activation of the STATE
                                          // network (`state` is that iteration's output; k_aggregate_dz does the same for the iterations before): -1 = plain dx
};

template <int T>
__global__ void __launch_bounds__(256) k_head_wgrad(HeadArgs a) {
    __shared__ float red[8][32][4 * T + 1];
    __shared__ float redq[8][T];
    const int tid = threadIdx.x, lane = tid & 31, grp = tid >> 5;          // 8 row groups of 32 lanes
    const int ns4 = a.S >> 2, nl4 = (a.L + 3) >> 2;                          // 16-byte pieces of a state row / a label row
    const bool is_state = lane < ns4, is_label = !is_state && lane < ns4 + nl4;
    const int lc = 4 * (lane - ns4);                                          // first label column of a label lane
    const int m_beg = blockIdx.x * a.rows_per_wg, m_end = min(a.M, m_beg + a.rows_per_wg);
    float acc[4][T], q[T];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int h = 0; h < T; ++h) acc[e][h] = 0.0f;
#pragma unroll
    for (int h = 0; h < T; ++h) q[h] = 0.0f;
    for (int m = m_beg + grp; m < m_end; m += 8 * 2) {                        // two rows in flight per lane
        float x[2][4], dz[2][T];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = m + 8 * u;
            const bool in = r < m_end;
#pragma unroll
            for (int e = 0; e < 4; ++e) x[u][e] = 0.0f;
            if (in && is_state) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(a.state + (size_t)r * a.ld_state + 4 * lane);
                x[u][0] = v[0]; x[u][1] = v[1]; x[u][2] = v[2]; x[u][3] = v[3];
            } else if (in && is_label) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (lc + e < a.L) x[u][e] = a.labels[(size_t)r * a.ld_labels + lc + e];
            }
#pragma unroll
            for (int h = 0; h < T; ++h) dz[u][h] = in ? a.dZ[(size_t)r * a.ldz + h] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int h = 0; h < T; ++h) acc[e][h] = fmaf(x[u][e], dz[u][h], acc[e][h]);
#pragma unroll
            for (int h = 0; h < T; ++h) q[h] += dz[u][h];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int h = 0; h < T; ++h) red[grp][lane][e * T + h] = acc[e][h];
    if (lane == 0) {
#pragma unroll
        for (int h = 0; h < T; ++h) redq[grp][h] = q[h];
    }
    __syncthreads();
    const int K = a.S + a.L;
    float *out = a.part + (size_t)blockIdx.x * ((size_t)K * T + T);
    for (int i = tid; i < K * T + T; i += 256) {                              // row groups in order
        float s = 0.0f;
        if (i < K * T) {
            const int k = i / T, h = i % T;
            const int ln = k < a.S ? (k >> 2) : ns4 + ((k - a.S) >> 2), e = k < a.S ? (k & 3) : ((k - a.S) & 3);
#pragma unroll
            for (int g = 0; g < 8; ++g) s += red[g][ln][e * T + h];
        } else {
#pragma unroll
            for (int g = 0; g < 8; ++g) s += redq[g][i - K * T];
        }
        out[i] = s;
    }
}

template <int T>
__global__ void __launch_bounds__(256) k_head_dx(HeadArgs a) {
    const int ns4 = a.S >> 2;                                                  // lanes per row (<= 16: S <= 64)
    const int lpr = ns4 <= 4 ? 4 : ns4 <= 8 ? 8 : 16;
    const int l4 = threadIdx.x % lpr, groups = 256 / lpr;
    const bool act = l4 < ns4;
    float w[4][T], Ac[4], Cc[4], M1[4], Mu[4];          // (the centred form: see k_train_bwd_dx)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int j = 4 * l4 + e;
        Ac[e] = 1.0f; Cc[e] = 0.0f; M1[e] = 0.0f; Mu[e] = 0.0f;
#pragma unroll
        for (int h = 0; h < T; ++h) w[e][h] = act ? a.W[(size_t)j * T + h] : 0.0f;
        if (a.gamma && act) {
            const float rstd = 1.0f / sqrtf(a.var[j] + a.eps);
            Ac[e] = a.gamma[j] * rstd; Cc[e] = -Ac[e] * rstd * a.m2[j]; M1[e] = a.m1[j]; Mu[e] = a.mean[j];
        }
    }
    // four rows of a lane group in flight, their loads issued together and unconditionally (a row past the end reads row 0 and is not stored)
    constexpr int RW = 4;
    const bool need_x = a.gamma || a.dz_act >= 0;
    for (int m0 = (blockIdx.x * groups + threadIdx.x / lpr) * RW; m0 < a.M; m0 += gridDim.x * groups * RW) {
        float dz[RW][T];
        f32x4 x[RW];
#pragma unroll
        for (int u = 0; u < RW; ++u) {
            const size_t mm = m0 + u < a.M ? (size_t)(m0 + u) : 0;
#pragma unroll
            for (int h = 0; h < T; ++h) dz[u][h] = a.dZ[mm * a.ldz + h];
            x[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (need_x) x[u] = *reinterpret_cast<const f32x4 *>(a.state + mm * a.ld_state + 4 * (act ? l4 : 0));      // (uniform condition)
        }
        if (!act) continue;
#pragma unroll
        for (int u = 0; u < RW; ++u) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float dy = 0.0f;
#pragma unroll
                for (int h = 0; h < T; ++h) dy = fmaf(dz[u][h], w[e][h], dy);
                v[e] = fmaf(Ac[e], dy - M1[e], Cc[e] * (x[u][e] - Mu[e]));
                if (a.dz_act >= 0) v[e] *= activate_grad_from_output(a.dz_act, x[u][e]);
            }
            if (m0 + u < a.M) *reinterpret_cast<f32x4 *>(a.dx + (size_t)(m0 + u) * a.ld_dx + 4 * l4) = v;
        }
    }
}

// sum of n floats times scale, any n: grid-stride partials in a fixed order, then one block (the loss of a million-row batch)
__global__ void __launch_bounds__(256) k_sum_partials(const float *__restrict__ x, int n, float *__restrict__ part) {
    __shared__ float sh[256];
    float s = 0.0f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) s += x[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

}  // namespace gnn
