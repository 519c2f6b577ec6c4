// A plain Dense layer over the rows of ONE matrix at large M (hidden and last layers of deeper state / output networks: reference
// MLP.py:12-78, every Dense behind the first): Y = act(X . W + b), optionally with the convergence predicate of the loop in the epilogue
// (GNN.py:196-214) when Y is the new state.  k_segdense (kernels_general.hpp) stages 64 x 32 chunks of a VIRTUAL concatenation through
// LDS with 4-byte loads - the right tool for the first layer's narrow label segments, 350 us per million rows for a 64 -> 64 layer.
// Here the rows go straight into the matrix cores as in k_train_fwd (kernels_train_big.hpp): operands swapped (weights = A from LDS in
// fragment order, the lane's 16-byte row pieces = B), so the result comes out row-major: 16-byte loads and stores only, no staging.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_train_big.hpp"        // BFrag, TB_MFMA_DRAIN, TB_WAVES

namespace gnn {

struct RowDenseArgs {
    const int *gate;
    int M;
    const float *X; int ldx, K;           // [M, K], K % 4 == 0, K <= 16 KQ
    const float *W; int ldw;              // [K, H] row-major
    const float *bias;                    // [H] or NULL
    int H, act;                           // H % 4 == 0, H <= 16 NCT
    float *Y; int ldy;
    const float *pred_old; int ld_pred; float thr; int *pred_flag; float *pred_k; float pred_kval;     // optional predicate (width H)
};

template <int KQ, int NCT>
__global__ void __launch_bounds__(64 * TB_WAVES, 4) k_rowdense(RowDenseArgs a) {
    if (gate_closed(a.gate)) return;
    constexpr int HP = 16 * NCT;
    extern __shared__ __attribute__((aligned(16))) float rd_smem[];
    float *Wl = rd_smem;                                // [4 KQ k-steps][4 g][16 c][NCT]
    float *bias_l = rd_smem + 16 * KQ * HP;             // [HP]
    __shared__ int any_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    if (tid == 0) any_s = 0;
    for (int i = tid; i < 16 * KQ * HP; i += 64 * TB_WAVES) {
        const int k = i / HP, h = i % HP;
        const float v = (k < a.K && h < a.H) ? a.W[(size_t)k * a.ldw + h] : 0.0f;
        const int q = k >> 4, rem = k & 15, gg = rem >> 2, e = rem & 3;
        Wl[(((4 * q + e) * 4 + gg) * 16 + (h & 15)) * NCT + (h >> 4)] = v;
    }
    for (int h = tid; h < HP; h += 64 * TB_WAVES) bias_l[h] = (h < a.H && a.bias) ? a.bias[h] : 0.0f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r_x = buf_rsrc(a.X), r_y = buf_rsrc(a.Y), r_o = buf_rsrc(a.pred_old);
    const int n_tiles = (a.M + 15) >> 4;
    int any = 0;
#pragma unroll 1
    for (int t = blockIdx.x * TB_WAVES + wave; t < n_tiles; t += gridDim.x * TB_WAVES) {
        const int row = 16 * t + c;
        const bool in = row < a.M;
        f32x4 A[KQ], O[NCT];
#pragma unroll
        for (int q = 0; q < KQ; ++q)
            A[q] = buf_ld_f32x4(r_x, (in && 16 * q + 4 * g < a.K) ? ((unsigned)row * (unsigned)a.ldx + 16u * q + 4u * g) * 4u : BUF_OFF);
        if (a.pred_flag) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
                O[ct] = buf_ld_f32x4(r_o, (in && 16 * ct + 4 * g < a.H) ? ((unsigned)row * (unsigned)a.ld_pred + 16u * ct + 4u * g) * 4u : BUF_OFF);
        }
        f32x4 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = *reinterpret_cast<const f32x4 *>(bias_l + 16 * ct + 4 * g);
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                BFrag<NCT> w;
                w.load(Wl + (((4 * q + e) * 4 + g) * 16 + c) * NCT);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.v[ct], A[q][e], acc[ct], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        TB_MFMA_DRAIN();
        float d2 = 0.0f, n2 = 0.0f;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            f32x4 v = acc[ct];
            activate4(a.act, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (in && 16 * ct + 4 * g + e < a.H) ? v[e] : 0.0f;
            const u32x4 bits = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
            __builtin_amdgcn_raw_buffer_store_b128(bits, r_y, (in && 16 * ct + 4 * g < a.H) ? (int)(((unsigned)row * (unsigned)a.ldy + 16u * ct + 4u * g) * 4u) : (int)BUF_OFF, 0, 0);
            if (a.pred_flag) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float o = O[ct][e], d = v[e] - o; d2 = fmaf(d, d, d2); n2 = fmaf(o, o, n2); }
            }
        }
        if (a.pred_flag) {
            d2 += __shfl_xor(d2, 16, 64); d2 += __shfl_xor(d2, 32, 64);
            n2 += __shfl_xor(n2, 16, 64); n2 += __shfl_xor(n2, 32, 64);
            if (in && sqrtf(d2) > a.thr * sqrtf(n2)) any = 1;
        }
    }
    if (a.pred_flag && __any(any) && lane == 0) any_s = 1;
    __syncthreads();
    if (a.pred_flag && tid == 0) {
        if (any_s) atomicOr(a.pred_flag, 1);
        if (blockIdx.x == 0 && a.pred_k) *a.pred_k = a.pred_kval;
    }
}

template <int KQ, int NCT>
inline size_t rowdense_lds() { return (size_t)(16 * KQ * 16 * NCT + 16 * NCT) * sizeof(float); }

}  // namespace gnn

namespace gnn {

// The same for WIDE layers (state widths above 128 take the un-fused path: its first layer is [state | agg] . W + C with 400+ input
// and 200+ output columns, 72 % of an iteration in k_segdense, which re-stages its input for every 64 output columns through LDS
// with 4-byte accesses: 39 TFLOP/s).  Up to two contiguous input matrices and a per-row addend; the output columns are cut into
// passes of 64 (blockIdx.y): a workgroup keeps ITS pass of the weights in LDS (all K rows x 64 columns, fragment order), a lane
// loads every 16-byte piece of its row once per pass and the matrix cores run K / 4 x 4 MFMAs per 16-row tile.
struct RowDenseWideArgs {
    const int *gate; int M;
    const float *X[2]; int ldx[2], width[2], wrow[2]; int nseg;
    const float *W; int ldw;
    const float *bias; const float *addend; int ld_add;
    int H, act; float *Y; int ldy;
};

constexpr int RDW_WAVES = 16;             // one workgroup per CU (its pass of the weights fills most of the LDS): 16 waves = 4 per SIMD
template <int KQM>
__global__ void __launch_bounds__(64 * RDW_WAVES, 4) k_rowdense_wide(RowDenseWideArgs a) {
    if (gate_closed(a.gate)) return;
    constexpr int NCT = 4, HP = 64;
    extern __shared__ __attribute__((aligned(16))) float rd_smem[];
    const int Q0 = (a.width[0] + 15) >> 4, Q1 = a.nseg > 1 ? (a.width[1] + 15) >> 4 : 0, KQ = Q0 + Q1;
    float *Wl = rd_smem;                                // [4 KQ k-steps][4 g][16 c][NCT]
    float *bias_l = rd_smem + 16 * KQ * HP;             // [HP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int h0 = HP * blockIdx.y;
    for (int i = tid; i < 16 * KQ * HP; i += 64 * RDW_WAVES) {
        const int k = i / HP, hh = i % HP;
        const int q = k >> 4, sg = q < Q0 ? 0 : 1, col = k - 16 * (sg ? Q0 : 0);
        const float v = (col < a.width[sg] && h0 + hh < a.H) ? a.W[(size_t)(a.wrow[sg] + col) * a.ldw + h0 + hh] : 0.0f;
        const int rem = k & 15, gg = rem >> 2, e = rem & 3;
        Wl[(((4 * q + e) * 4 + gg) * 16 + (hh & 15)) * NCT + (hh >> 4)] = v;
    }
    for (int hh = tid; hh < HP; hh += 64 * RDW_WAVES) bias_l[hh] = (a.bias && h0 + hh < a.H) ? a.bias[h0 + hh] : 0.0f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r_x0 = buf_rsrc(a.X[0]), r_x1 = buf_rsrc(a.nseg > 1 ? a.X[1] : nullptr), r_c = buf_rsrc(a.addend), r_y = buf_rsrc(a.Y);
    // TWO 16-row tiles per wave and trip: every weight fragment read from LDS feeds two MFMAs.  (One tile per trip reads 16 bytes per
    // lane and MFMA quartet: 16 waves x 1 KB per 512 matrix-pipe cycles = the whole 128 B / clock of the LDS - the layer ran at 42 TFLOP/s.)
    const int n_pairs = (a.M + 31) >> 5;
#pragma unroll 1
    for (int t = blockIdx.x * RDW_WAVES + wave; t < n_pairs; t += gridDim.x * RDW_WAVES) {
        const int row0 = 32 * t + c, row1 = row0 + 16;
        const bool in0 = row0 < a.M, in1 = row1 < a.M;
        f32x4 acc0[NCT], acc1[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias_l + 16 * ct + 4 * g);
            acc0[ct] = bv; acc1[ct] = bv;
            if (a.addend) {
                const bool hv = h0 + 16 * ct + 4 * g < a.H;
                acc0[ct] += buf_ld_f32x4(r_c, (in0 && hv) ? ((unsigned)row0 * (unsigned)a.ld_add + (unsigned)(h0 + 16 * ct + 4 * g)) * 4u : BUF_OFF);
                acc1[ct] += buf_ld_f32x4(r_c, (in1 && hv) ? ((unsigned)row1 * (unsigned)a.ld_add + (unsigned)(h0 + 16 * ct + 4 * g)) * 4u : BUF_OFF);
            }
        }
        // the rows' 16-byte pieces in blocks of KQM chunks (all loads of a block in flight, then its MFMAs)
#pragma unroll 1
        for (int qb = 0; qb < KQ; qb += KQM) {
            f32x4 A0[KQM], A1[KQM];
#pragma unroll
            for (int qq = 0; qq < KQM; ++qq) {           // (chunks behind KQ: predicated off, never multiplied)
                const int q = qb + qq;
                const bool s1 = q >= Q0;
                const int col = 16 * (q - (s1 ? Q0 : 0)) + 4 * g;
                const bool cv = q < KQ && col < a.width[s1 ? 1 : 0];
                const unsigned ld = (unsigned)a.ldx[s1 ? 1 : 0];
                A0[qq] = buf_ld_f32x4(s1 ? r_x1 : r_x0, (in0 && cv) ? ((unsigned)row0 * ld + (unsigned)col) * 4u : BUF_OFF);   // (s1 is uniform: a scalar select)
                A1[qq] = buf_ld_f32x4(s1 ? r_x1 : r_x0, (in1 && cv) ? ((unsigned)row1 * ld + (unsigned)col) * 4u : BUF_OFF);
            }
#pragma unroll
            for (int qq = 0; qq < KQM; ++qq) {
                if (qb + qq < KQ) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        BFrag<NCT> w;
                        w.load(Wl + (((4 * (qb + qq) + e) * 4 + g) * 16 + c) * NCT);
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) {
                            acc0[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.v[ct], A0[qq][e], acc0[ct], 0, 0, 0);
                            acc1[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.v[ct], A1[qq][e], acc1[ct], 0, 0, 0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        TB_MFMA_DRAIN();
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int row = half ? row1 : row0;
            const bool in = half ? in1 : in0;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                f32x4 v = half ? acc1[ct] : acc0[ct];
                activate4(a.act, v);
                const int hcol = h0 + 16 * ct + 4 * g;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (in && hcol + e < a.H) ? v[e] : 0.0f;
                const u32x4 bits = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                __builtin_amdgcn_raw_buffer_store_b128(bits, r_y, (in && hcol < a.H) ? (int)(((unsigned)row * (unsigned)a.ldy + (unsigned)hcol) * 4u) : (int)BUF_OFF, 0, 0);
            }
        }
    }
}

}  // namespace gnn
