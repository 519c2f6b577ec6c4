// General (un-fused) gfx950 kernels of the message-passing loop: CSR aggregate, segmented dense layer on f32 MFMA,
// convergence predicate, softmax, BN folding, small copies.  Every kernel takes an optional `gate` word: when it
// points at 0 the launch returns immediately — that is how the host enqueues `max_iteration` iterations with no
// host synchronisation while the device stops doing work once the predicate (reference GNN.py:196-214) says stop.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gnnloop.h"

namespace gnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool gate_closed(const int *gate) { return gate != nullptr && *gate == 0; }

// Keras activations (elementwise ones; softmax is a row kernel).  selu follows TF's `scale*alpha*(exp(x)-1)`.
__device__ __forceinline__ float activate(int act, float x) {
    switch (act) {
        case GNN_ACT_RELU: return fmaxf(x, 0.0f);
        case GNN_ACT_SELU: return x > 0.0f ? 1.0507009873554805f * x
                                           : (1.0507009873554805f * 1.6732632423543772f) * (expf(x) - 1.0f);
        case GNN_ACT_TANH: return tanhf(x);
        case GNN_ACT_SIGMOID: return 1.0f / (1.0f + expf(-x));
        case GNN_ACT_ELU: return x > 0.0f ? x : expf(x) - 1.0f;
        case GNN_ACT_SOFTPLUS: return x > 20.0f ? x : log1pf(expf(x));
        default: return x;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// out[j, :F] = row_scale[j] * sum_e w_e X[src_e, :F]     (tf.sparse.sparse_dense_matmul(A, X, adjoint_a=True))
// G lanes walk one destination row together: lane f reads column f of every source row (coalesced per row), the
// per-destination sum runs in ascending-source order like the reference's CPU kernel.
// ---------------------------------------------------------------------------------------------------------------------
template <int G>
__global__ void __launch_bounds__(256)
k_aggregate(const int *gate, int n_dst, const int *__restrict__ rowptr, const int *__restrict__ src,
            const float *__restrict__ w, const float *__restrict__ row_scale, const float *__restrict__ X, int ldx,
            int F, float *__restrict__ out, int ldo) {
    if (gate_closed(gate)) return;
    const int lane = threadIdx.x % G;
    const int groups = blockDim.x / G;
    for (int j = blockIdx.x * groups + threadIdx.x / G; j < n_dst; j += gridDim.x * groups) {
        const int beg = rowptr[j], end = rowptr[j + 1];
        const float scale = row_scale ? row_scale[j] : 1.0f;
        for (int f = lane; f < F; f += G) {
            float acc = 0.0f;
            int e = beg;
            for (; e + 8 <= end; e += 8) {          // 8 independent source rows in flight, summed in arc order
                int sid[8]; float x[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) sid[i] = src[e + i];
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = X[(size_t)sid[i] * ldx + f];
#pragma unroll
                for (int i = 0; i < 8; ++i) acc = w ? fmaf(w[e + i], x[i], acc) : acc + x[i];
            }
            for (; e < end; ++e) acc = w ? fmaf(w[e], X[(size_t)src[e] * ldx + f], acc) : acc + X[(size_t)src[e] * ldx + f];
            out[(size_t)j * ldo + f] = acc * scale;
        }
    }
}

// Narrow rows (F <= G <= 16 floats: node labels, arc labels) with the whole CSR row in flight: the G lanes of a destination fetch its
// next 16 source ids in one coalesced trip (16 / G per lane), broadcast them, and issue all 16 row loads before the first add - raw
// buffer loads predicated off by an out-of-range offset, so nothing branches around a load.  k_aggregate above keeps 8 in flight and
// walks the remainder of a row one dependent chain per arc (C4: labels 257 -> ~110 us, arc labels 207 -> ~90 us, once per forward).
// Sums in arc order like k_aggregate: the same bits.  Arrays must fit 4 GiB buffer windows (the launcher checks).
template <int G, bool HAS_W>
__global__ void __launch_bounds__(256)
k_aggregate_narrow(const int *gate, int n_dst, const int *__restrict__ rowptr, const int *__restrict__ src, const float *__restrict__ w,
                   const float *__restrict__ row_scale, const float *__restrict__ X, int ldx, int F, float *__restrict__ out, int ldo) {
    if (gate_closed(gate)) return;
    constexpr int IPL = 16 / G;
    const int lane = threadIdx.x % G;
    const int groups = blockDim.x / G;
    const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X), 0, (int)0xFFFFFFF0u, 0x00020000),
                                 r_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(src), 0, (int)0xFFFFFFF0u, 0x00020000),
                                 r_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(HAS_W ? w : X), 0, (int)0xFFFFFFF0u, 0x00020000);
    constexpr unsigned OFF = 0xFFFFFFFFu;
    for (int j0 = blockIdx.x * groups; j0 < n_dst; j0 += gridDim.x * groups) {
        const int j = j0 + threadIdx.x / G;
        const bool in = j < n_dst;
        const int beg = in ? rowptr[j] : 0, end = in ? rowptr[j + 1] : 0;
        float acc = 0.0f;
        int eb = beg;
        while (__any(eb < end)) {
            int ids[IPL]; float ws[IPL];
#pragma unroll
            for (int u = 0; u < IPL; ++u) {
                const int e = eb + u * G + lane;
                ids[u] = __builtin_amdgcn_raw_buffer_load_b32(r_s, e < end ? 4u * (unsigned)e : OFF, 0, 0);
                ws[u] = HAS_W ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_w, e < end ? 4u * (unsigned)e : OFF, 0, 0)) : 0.0f;
            }
            float x[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const unsigned sid = (unsigned)__shfl(ids[i / G], i % G, G);
                x[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_x, (eb + i < end && lane < F) ? (sid * (unsigned)ldx + (unsigned)lane) * 4u : OFF, 0, 0));
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (HAS_W) acc = fmaf(__shfl(ws[i / G], i % G, G), x[i], acc);
                else acc += x[i];
            }
            eb += 16;
        }
        if (in && lane < F) out[(size_t)j * ldo + lane] = acc * (row_scale ? row_scale[j] : 1.0f);
    }
}

// The 16-byte row pieces of one destination's arcs, eight in flight, summed in arc order (k_aggregate_vec and the training aggregates of
// kernels_train_big.hpp).  BUF: raw buffer loads predicated off by an out-of-range offset - the eight source ids are requested together,
// then the eight rows; behind `ok ? load : 0` hipcc branches around every load and waits (vmcnt(0)) for each source id AND everything
// before it, i.e. one dependent pair of round trips per arc and wave, hidden only by occupancy (ISA of the round-4 kernels).  The arrays
// must fit 4 GiB buffer windows (n_src * ldx * 4 and nnz * 4 below 2^32: the launchers check and fall back to !BUF).  Same sums, same order.
template <bool HAS_W, bool BUF>
__device__ __forceinline__ f32x4 gather_sum8(int beg, int end, const int *__restrict__ src, const float *__restrict__ w, const float *__restrict__ X,
                                               int ldx, int col) {
    typedef unsigned int u32x4_g __attribute__((ext_vector_type(4)));
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (BUF) {
        const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X), 0, (int)0xFFFFFFF0u, 0x00020000),
                                     r_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(src), 0, (int)0xFFFFFFF0u, 0x00020000),
                                     r_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(HAS_W ? w : X), 0, (int)0xFFFFFFF0u, 0x00020000);
        constexpr unsigned OFF = 0xFFFFFFFFu;
        for (int e = beg; e < end; e += 8) {
            unsigned sid[8]; float wv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned o = e + i < end ? 4u * (unsigned)(e + i) : OFF;
                sid[i] = __builtin_amdgcn_raw_buffer_load_b32(r_s, (int)o, 0, 0);
                wv[i] = HAS_W ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_w, (int)o, 0, 0)) : 1.0f;
            }
            f32x4 x[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const u32x4_g v = __builtin_amdgcn_raw_buffer_load_b128(r_x, (int)(e + i < end ? (sid[i] * (unsigned)ldx + (unsigned)col) * 4u : OFF), 0, 0);
                x[i] = (f32x4){__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (HAS_W) acc += wv[i] * x[i];          // (past the end: w = 0, x = 0)
                else acc += x[i];
            }
        }
    } else {
        for (int e = beg; e < end; e += 8) {
            f32x4 x[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool ok = e + i < end;
                const int sid = ok ? src[e + i] : 0;
                x[i] = ok ? *reinterpret_cast<const f32x4 *>(X + (size_t)sid * ldx + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (HAS_W) acc += (e + i < end ? w[e + i] : 0.0f) * x[i];
                else acc += x[i];
            }
        }
    }
    return acc;
}

// The same product for rows that allow 16-B accesses (F, ldx, ldo multiples of 4, 16-B aligned bases: the padded state
// matrix of the un-fused and training paths): LPR = F/4 lanes own a destination row, each lane carries a float4 column
// chunk, 8 source rows in flight; a wave instruction moves 64/LPR whole rows instead of one.  `addend` (optional, the
// backward sweep's G_state = dx_state + Adj . dx_agg): out = addend + the product.
template <int LPR, bool HAS_W, bool BUF = false>
__global__ void __launch_bounds__(256)
k_aggregate_vec(const int *gate, int n_dst, const int *__restrict__ rowptr, const int *__restrict__ src,
                const float *__restrict__ w, const float *__restrict__ row_scale, const float *__restrict__ X, int ldx,
                float *__restrict__ out, int ldo, const float *__restrict__ addend = nullptr, int ld_add = 0) {
    if (gate_closed(gate)) return;
    const int l4 = threadIdx.x % LPR;
    const int groups = blockDim.x / LPR;
    for (int j = blockIdx.x * groups + threadIdx.x / LPR; j < n_dst; j += gridDim.x * groups) {
        const int beg = rowptr[j], end = rowptr[j + 1];
        f32x4 acc = gather_sum8<HAS_W, BUF>(beg, end, src, w, X, ldx, 4 * l4);                   // summed in arc order
        if (row_scale) acc *= row_scale[j];
        if (addend) acc += *reinterpret_cast<const f32x4 *>(addend + (size_t)j * ld_add + 4 * l4);
        *reinterpret_cast<f32x4 *>(out + (size_t)j * ldo + 4 * l4) = acc;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Segmented dense layer on the f32 matrix cores (v_mfma_f32_16x16x4_f32: exact f32, k-ordered fma chain):
//   Y[orow(m), :H] = act( sum_s X_s[row_s(m), :] . W[wrow_s : wrow_s + width_s, :H] + bias + addend[arow(m), :H] )
// The input row is the *virtual* concatenation of up to 6 column segments, each with its own base pointer, leading
// dimension and optional row-index list, so neither the reference's tf.concat (GNN.py:231, :241, :327) nor its
// boolean_mask / gather (GNN.py:242, :322) is ever materialised.
// ---------------------------------------------------------------------------------------------------------------------
#define GNN_MAX_SEGS 6
struct Seg {
    const float *ptr;
    const int *rowidx;   // nullable: row of `ptr` for logical row m is rowidx ? rowidx[m] : m
    int ld, width, wrow;
};
struct SegDenseArgs {
    const int *gate;
    int M, H, nseg;
    Seg seg[GNN_MAX_SEGS];
    const float *W; int ldw;
    const float *bias;
    const float *addend; int ld_add; const int *add_rowidx;
    int act;
    float *Y; int ldy; const int *out_rowidx;
    // optional fused convergence predicate (k_segdense only, H <= 64, out_rowidx == NULL): the rows just computed are the new
    // state, `pred_old` the previous one: *pred_flag |= any_row(||new - old|| > thr ||old||), *pred_k = pred_kval
    const float *pred_old; int ld_pred; float pred_thr; int *pred_flag; float *pred_k; float pred_kval;
    // optional training-mode BatchNormalization applied to the input columns as they are staged (k_segdense only):
    // x'[k] = (x[k] - mean[k]) gamma[k] / sqrt(var[k] + eps) + beta[k], k = the column's WEIGHT ROW (seg.wrow + offset inside the segment) -
    // computed in that order: the column mean leaves the value first, a folded beta - mean a would cancel it against a x afterwards
    const float *in_gamma, *in_beta, *in_mean, *in_var; float in_eps;
    // optional (k_segdense, k_thin_dense): in_center[weight row] is subtracted from every input value as it is staged - the consumer of a
    // CENTRED fold (FoldJob::centred: Wf = a W, bf = b + sum beta W), for layers whose kernel wants folded weights
    const float *in_center;
};

constexpr int SD_TM = 64, SD_TN = 64, SD_KC = 32, SD_LDX = 34, SD_LDW = 80, SD_LDY = 68;

// K chunks of 32 run over the *virtual* concatenation, so several narrow segments (labels 14 + aggregated labels 14 +
// aggregated arcs 3) share one chunk.  Staging is lane-contiguous: a wave reads 2 x 128 B of X rows / 256 B of a W row
// per instruction, and results leave through an LDS tile as 256-B row pieces.
// SC = chunks of 32 columns whose loads are issued together: 4 for small M (latency: a 95-column training layer costs one
// round trip instead of three), 1 for large M (throughput: ~64 VGPRs and 19 KB of LDS -> 8 waves / SIMD; this access
// pattern, like the gather, is served by the number of waves in flight, not by the loads in flight per wave: a software
// pipeline that issues the next chunk's loads before this chunk's MFMAs needs ~114 VGPRs = 4 waves / SIMD and measured
// SLOWER, 615 -> 870 us on a 1 M x 160 -> 64 layer, profiles/r02_notes.txt).
template <int SC>
__global__ void __launch_bounds__(256, SC == 1 ? 8 : 4) k_segdense(SegDenseArgs a) {
    if (gate_closed(a.gate)) return;
    constexpr int XW = SD_TM * SD_LDX + SD_KC * SD_LDW, YS = SD_TM * SD_LDY;
    __shared__ float smem_sd[XW > YS ? XW : YS];
    float *Xs = smem_sd, *Ws = smem_sd + SD_TM * SD_LDX;
    float *Ys = smem_sd;                                        // the output tile reuses the staging space (barriers between)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * SD_TM;
    int K = 0;
    for (int s = 0; s < a.nseg; ++s) K += a.seg[s].width;
    const int xc = tid & 31, xr0 = tid >> 5;                    // X staging: chunk column xc, rows xr0 + 8*pass

    for (int n0 = 0; n0 < a.H; n0 += SD_TN) {
        f32x4 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // K runs in super-chunks of SC x 32 virtual columns: the loads of a whole super-chunk (X: 8 values per thread
        // and chunk, W: 8) are issued before the first is used, so a 95-column training layer costs ONE global round
        // trip instead of three; the chunks then go through LDS one at a time.
        for (int k0 = 0; k0 < K; k0 += SC * SD_KC) {
            float xv[SC][8], wv[SC][8];
#pragma unroll
            for (int sc = 0; sc < SC; ++sc) {
                // ---- X chunk: this thread's virtual column is fixed, resolve its segment once ----
                const int kv = k0 + sc * SD_KC + xc;
                Seg sg = a.seg[0];                              // static indices only: a runtime-indexed kernel-argument
                int sbeg = 0, start = a.seg[0].width;           // array would be copied to scratch memory
#pragma unroll
                for (int s = 1; s < GNN_MAX_SEGS; ++s) {
                    if (s < a.nseg && kv >= start) { sg = a.seg[s]; sbeg = start; }
                    if (s < a.nseg) start += a.seg[s].width;
                }
                const int off = kv - sbeg;
                float bn_a = 1.0f, bn_c = 0.0f, bn_m = 0.0f;
                if (a.in_gamma && kv < K) {
                    const int kr = sg.wrow + off;
                    bn_a = a.in_gamma[kr] / sqrtf(a.in_var[kr] + a.in_eps);
                    bn_c = a.in_beta[kr]; bn_m = a.in_mean[kr];
                } else if (a.in_center && kv < K) bn_m = a.in_center[sg.wrow + off];
#pragma unroll
                for (int pass = 0; pass < 8; ++pass) {
                    const int m = m0 + xr0 + 8 * pass;
                    xv[sc][pass] = 0.0f;
                    if (m < a.M && kv < K) {
                        const float x = sg.ptr[(sg.rowidx ? (size_t)sg.rowidx[m] : (size_t)m) * sg.ld + off];
                        xv[sc][pass] = a.in_gamma ? fmaf(x - bn_m, bn_a, bn_c) : x - bn_m;
                    }
                }
                // ---- W chunk: weight rows of the virtual columns (row = seg.wrow + offset inside the segment) ----
#pragma unroll
                for (int pass = 0; pass < 8; ++pass) {
                    const int kw = k0 + sc * SD_KC + wave + 4 * pass;
                    int wrow = a.seg[0].wrow + kw, wstart = a.seg[0].width;
#pragma unroll
                    for (int s = 1; s < GNN_MAX_SEGS; ++s) {
                        if (s < a.nseg && kw >= wstart) wrow = a.seg[s].wrow + kw - wstart;
                        if (s < a.nseg) wstart += a.seg[s].width;
                    }
                    wv[sc][pass] = 0.0f;
                    if (kw < K && n0 + lane < a.H) wv[sc][pass] = a.W[(size_t)wrow * a.ldw + n0 + lane];
                }
            }
#pragma unroll
            for (int sc = 0; sc < SC; ++sc) {
                if (k0 + sc * SD_KC >= K) break;
#pragma unroll
                for (int pass = 0; pass < 8; ++pass) {
                    Xs[(xr0 + 8 * pass) * SD_LDX + xc] = xv[sc][pass];
                    Ws[(wave + 4 * pass) * SD_LDW + lane] = wv[sc][pass];
                }
                __syncthreads();
#pragma unroll
                for (int s4 = 0; s4 < SD_KC / 4; ++s4) {
                    const float av = Xs[(16 * wave + r) * SD_LDX + 4 * s4 + g];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float bv = Ws[(4 * s4 + g) * SD_LDW + 16 * c + r];
                        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[c], 0, 0, 0);
                    }
                }
                __syncthreads();
            }
        }
        // C/D layout of 16x16x4: col = lane & 15, row = 4 * (lane >> 4) + reg  -> LDS tile -> row-contiguous stores
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) Ys[(16 * wave + 4 * g + reg) * SD_LDY + 16 * c + r] = acc[c][reg];
        __syncthreads();
        const int col = n0 + lane;
        const float bcol = (a.bias && col < a.H) ? a.bias[col] : 0.0f;
        int moving = 0;
#pragma unroll 4
        for (int pass = 0; pass < 16; ++pass) {                 // a wave writes one 256-B row piece per instruction
            const int yr = wave + 4 * pass, m = m0 + yr;
            float d2 = 0.0f, o2 = 0.0f;
            if (m < a.M && col < a.H) {
                float v = Ys[yr * SD_LDY + lane] + bcol;
                if (a.addend) v += a.addend[(a.add_rowidx ? (size_t)a.add_rowidx[m] : (size_t)m) * a.ld_add + col];
                v = activate(a.act, v);
                a.Y[(a.out_rowidx ? (size_t)a.out_rowidx[m] : (size_t)m) * a.ldy + col] = v;
                if (a.pred_flag) {
                    const float o = a.pred_old[(size_t)m * a.ld_pred + col];
                    d2 = (v - o) * (v - o); o2 = o * o;
                }
            }
            if (a.pred_flag) {                                  // wave-uniform: the whole row lives in this wave (H <= 64)
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) { d2 += __shfl_xor(d2, off, 64); o2 += __shfl_xor(o2, off, 64); }
                if (m < a.M && sqrtf(d2) > a.pred_thr * sqrtf(o2)) moving = 1;
            }
        }
        if (a.pred_flag) {
            moving = __syncthreads_or(moving);
            if (tid == 0) {
                if (moving) atomicOr(a.pred_flag, 1);
                if (blockIdx.x == 0 && a.pred_k) *a.pred_k = a.pred_kval;
            }
        } else {
            __syncthreads();
        }
    }
}

// The same segmented dense layer for THIN outputs (H <= 4: class scores of the output network, GNN.py:273).  A 64-column
// MFMA tile would spend 62 of its columns on nothing and stage every input row through LDS; here 16 lanes share a row
// (16 B per lane where the segment allows it, one 256-B state row per load), keep H partial dot products each, and
// meet in a 4-step shuffle tree (fixed order: deterministic).  8 rows per lane group are in flight per trip, their loads issued
// together and unconditionally (behind `if (row valid)` hipcc branches around every load and waits for each alone: 220 us per 1 M rows).
constexpr int TD_ROWS = 8;
template <int H>
__global__ void __launch_bounds__(256) k_thin_dense(SegDenseArgs a, int K) {
    if (gate_closed(a.gate)) return;
    extern __shared__ __attribute__((aligned(16))) float tdW[];                       // [K][H], then [K (+ 3)] centres
    float *tdC = tdW + ((K * H + 3) & ~3);
    for (int i = threadIdx.x; i < K * H; i += blockDim.x) tdW[i] = a.W[(size_t)(i / H) * a.ldw + (i % H)];
    if (a.in_center) for (int i = threadIdx.x; i < K; i += blockDim.x) tdC[i] = a.in_center[i];
    __syncthreads();
    const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
    for (long base = ((long)blockIdx.x * 16 + grp) * TD_ROWS; base < a.M; base += (long)gridDim.x * 16 * TD_ROWS) {
        float acc[TD_ROWS][H];
#pragma unroll
        for (int r = 0; r < TD_ROWS; ++r)
#pragma unroll
            for (int h = 0; h < H; ++h) acc[r][h] = 0.0f;
#pragma unroll
        for (int s = 0; s < GNN_MAX_SEGS; ++s) {
            if (s >= a.nseg) break;
            const Seg sg = a.seg[s];
            long rowv[TD_ROWS];
            if (sg.rowidx) {                                    // (uniform; the TD_ROWS index loads issued together, unconditional: see the row loads below)
                int idx[TD_ROWS];
#pragma unroll
                for (int r = 0; r < TD_ROWS; ++r) idx[r] = sg.rowidx[base + r < a.M ? base + r : 0];
#pragma unroll
                for (int r = 0; r < TD_ROWS; ++r) rowv[r] = base + r < a.M ? (long)idx[r] : -1;
            } else {
#pragma unroll
                for (int r = 0; r < TD_ROWS; ++r) rowv[r] = base + r < a.M ? base + r : -1;
            }
            const bool vec = (sg.width % 4 == 0) && (sg.ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(sg.ptr) & 15) == 0);
            if (vec) {
                for (int c4 = l16; c4 * 4 < sg.width; c4 += 16) {
                    f32x4 x[TD_ROWS];
#pragma unroll
                    for (int r = 0; r < TD_ROWS; ++r)                   // unconditional loads (a row past the end reads row 0 and is zeroed after): behind `if (row >= 0)` hipcc
                        x[r] = *reinterpret_cast<const f32x4 *>(sg.ptr + (rowv[r] >= 0 ? rowv[r] : 0) * sg.ld + 4 * c4);      // branches around every load and waits for each alone
#pragma unroll
                    for (int r = 0; r < TD_ROWS; ++r) if (rowv[r] < 0) x[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    const float *w = tdW + (size_t)(sg.wrow + 4 * c4) * H;
                    if (a.in_center) {
                        const float *cp = tdC + sg.wrow + 4 * c4;            // (wrow need not be a multiple of 4: scalar reads)
                        const f32x4 cen = {cp[0], cp[1], cp[2], cp[3]};
#pragma unroll
                        for (int r = 0; r < TD_ROWS; ++r) if (rowv[r] >= 0) x[r] -= cen;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int h = 0; h < H; ++h) {
                            const float wv = w[e * H + h];
#pragma unroll
                            for (int r = 0; r < TD_ROWS; ++r) acc[r][h] = fmaf(x[r][e], wv, acc[r][h]);
                        }
                }
            } else {
                for (int c = l16; c < sg.width; c += 16) {
                    float x[TD_ROWS];
#pragma unroll
                    for (int r = 0; r < TD_ROWS; ++r) x[r] = sg.ptr[(rowv[r] >= 0 ? rowv[r] : 0) * sg.ld + c];          // (unconditional, as above)
                    const float cen = a.in_center ? tdC[sg.wrow + c] : 0.0f;
#pragma unroll
                    for (int r = 0; r < TD_ROWS; ++r) x[r] = rowv[r] >= 0 ? x[r] - cen : 0.0f;
                    const float *w = tdW + (size_t)(sg.wrow + c) * H;
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        const float wv = w[h];
#pragma unroll
                        for (int r = 0; r < TD_ROWS; ++r) acc[r][h] = fmaf(x[r], wv, acc[r][h]);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < TD_ROWS; ++r)
#pragma unroll
            for (int h = 0; h < H; ++h)
#pragma unroll
                for (int off = 8; off >= 1; off >>= 1) acc[r][h] += __shfl_xor(acc[r][h], off, 16);
        if (a.act == GNN_ACT_SOFTMAX) {                  // every lane holds all H sums: lane 0 finishes the whole row
            if (l16 == 0) {
#pragma unroll
                for (int r = 0; r < TD_ROWS; ++r) {
                    const long m = base + r;
                    if (m >= a.M) continue;
                    float v[H], mx = -3.4e38f, sum = 0.0f;
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        v[h] = acc[r][h];
                        if (a.bias) v[h] += a.bias[h];
                        if (a.addend) v[h] += a.addend[(size_t)(a.add_rowidx ? a.add_rowidx[m] : m) * a.ld_add + h];
                        mx = fmaxf(mx, v[h]);
                    }
#pragma unroll
                    for (int h = 0; h < H; ++h) { v[h] = expf(v[h] - mx); sum += v[h]; }
                    float *y = a.Y + (size_t)(a.out_rowidx ? a.out_rowidx[m] : m) * a.ldy;
#pragma unroll
                    for (int h = 0; h < H; ++h) y[h] = v[h] / sum;
                }
            }
        } else if (l16 < H) {                            // lane h of the group finishes output column h
#pragma unroll
            for (int r = 0; r < TD_ROWS; ++r) {
                const long m = base + r;
                if (m >= a.M) continue;
                float v = acc[r][0];
#pragma unroll
                for (int h = 1; h < H; ++h) v = l16 == h ? acc[r][h] : v;
                if (a.bias) v += a.bias[l16];
                if (a.addend) v += a.addend[(size_t)(a.add_rowidx ? a.add_rowidx[m] : m) * a.ld_add + l16];
                a.Y[(size_t)(a.out_rowidx ? a.out_rowidx[m] : m) * a.ldy + l16] = activate(a.act, v);
            }
        }
    }
}

// row softmax in place (Keras 'softmax' activation of the output network, starter.py:28): max-subtracted.
__global__ void __launch_bounds__(256)
k_softmax_rows(const int *gate, float *Y, int M, int H, int ldy, const int *rowidx) {
    if (gate_closed(gate)) return;
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float *row = Y + (rowidx ? (size_t)rowidx[m] : (size_t)m) * ldy;
    float mx = row[0];
    for (int h = 1; h < H; ++h) mx = fmaxf(mx, row[h]);
    float sum = 0.0f;
    for (int h = 0; h < H; ++h) { const float e = expf(row[h] - mx); row[h] = e; sum += e; }
    for (int h = 0; h < H; ++h) row[h] = row[h] / sum;
}

// ---------------------------------------------------------------------------------------------------------------------
// convergence predicate (GNN.py:196-212): flag |= any_j( sqrt(sum (s-so)^2) > thr * sqrt(sum so^2) ), strict.
// 16 lanes per node.  `so == nullptr` means state_old = ones (GNN.py:261).  Optionally records k (float, Q6).
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_converge(const int *gate, const float *__restrict__ s, const float *__restrict__ so, int N, int S, int ld_s,
           int ld_so, float thr, int *flag_out, float *k_out, float k_val, const int *skip_if_set) {
    if (gate_closed(gate)) return;
    if (skip_if_set != nullptr && *skip_if_set != 0) return;      // the answer ("some node still moves") is already known
    const int lane = threadIdx.x & 15;
    const int groups = blockDim.x / 16;
    int any = 0;
    for (int j0 = blockIdx.x * groups; j0 < N; j0 += gridDim.x * groups) {
        const int j = j0 + threadIdx.x / 16;
        float d2 = 0.0f, n2 = 0.0f;
        if (j < N) {
            for (int f = lane; f < S; f += 16) {
                const float o = so ? so[(size_t)j * ld_so + f] : 1.0f;
                const float d = s[(size_t)j * ld_s + f] - o;
                d2 = fmaf(d, d, d2);
                n2 = fmaf(o, o, n2);
            }
        }
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
            d2 += __shfl_xor(d2, off, 16);
            n2 += __shfl_xor(n2, off, 16);
        }
        if (j < N && sqrtf(d2) > thr * sqrtf(n2)) any = 1;
    }
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) {
        if (any) atomicOr(flag_out, 1);
        if (blockIdx.x == 0 && k_out) *k_out = k_val;
    }
}

// The XC variant of the wave-specialised kernel (kernel_state_fused4.hpp) multiplies a node's constant inputs on the matrix
// cores instead of reading the per-node constant C.  Xc[j] = [labels | aggregated labels | aggregated arc labels | 1 | 0 ..]
// (32 floats = one 128-byte line), Wc = the folded first-layer rows of those inputs, then the folded bias, then zeros.
// One node type at a time: its constant first-layer segments (TypePlan::cseg: own labels, aggregated labels, aggregated arc
// labels - each a node-indexed array) are concatenated into the rows of its nodes.
struct PackSegs { const float *ptr[3]; int ld[3], width[3], wrow[3]; int n; };
__global__ void __launch_bounds__(256)
k_pack_xc(int count, const int *__restrict__ rows, PackSegs ps, float *__restrict__ Xc) {
    const int Kc = ps.width[0] + ps.width[1] + ps.width[2];
    const size_t total = (size_t)count * 32;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i >> 5;
        const size_t j = rows ? (size_t)rows[m] : m;
        const int c = (int)(i & 31);
        float v = 0.0f;
        if (c < ps.width[0]) v = ps.ptr[0][j * ps.ld[0] + c];
        else if (c < ps.width[0] + ps.width[1]) v = ps.ptr[1][j * ps.ld[1] + (c - ps.width[0])];
        else if (c < Kc) v = ps.ptr[2][j * ps.ld[2] + (c - ps.width[0] - ps.width[1])];
        else if (c == Kc) v = 1.0f;
        Xc[j * 32 + c] = v;
    }
}
__global__ void __launch_bounds__(256)
k_pack_wc(const float *__restrict__ Wf, const float *__restrict__ bf, int H, PackSegs ps, float *__restrict__ Wc) {
    const int Kc = ps.width[0] + ps.width[1] + ps.width[2];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 32 * H; i += gridDim.x * blockDim.x) {
        const int c = i / H, h = i % H;
        float v = 0.0f;
        if (c < Kc) {
            const int k = c < ps.width[0] ? ps.wrow[0] + c
                        : (c < ps.width[0] + ps.width[1] ? ps.wrow[1] + (c - ps.width[0]) : ps.wrow[2] + (c - ps.width[0] - ps.width[1]));
            v = Wf[(size_t)k * H + h];
        } else if (c == Kc) v = bf[h];
        Wc[i] = v;
    }
}

// Independent convergence groups of a merged graph (gnn_loop_args_t::group_node_begin): nodes and 64-node tiles of group g.
// Passed by value in the kernel arguments and searched with a fully unrolled loop (constant indices: scalar loads from the
// kernel-argument segment, no scratch copy); tile numbers are wave-uniform.
constexpr int MAX_GROUPS = 32;
struct GroupTab {
    int n;                               // 0: no groups (one loop over everything)
    int node_begin[MAX_GROUPS + 1];
    int tile_begin[MAX_GROUPS + 1];
    // more than MAX_GROUPS groups (one workgroup per group, kernel_state_lds.hpp): the same two tables in device memory,
    // [n_dev + 1] each, searched by bisection; n is then unused
    const int *d_node_begin, *d_tile_begin;
    int n_dev;
};
struct GroupOfTile { int grp, node0, node_end, tile0, tile1; };
__device__ __forceinline__ GroupOfTile group_of_tile(const GroupTab &gt, int tile) {
    if (gt.d_tile_begin) {               // largest g with tile_begin[g] <= tile
        int lo = 0, hi = gt.n_dev - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (gt.d_tile_begin[mid] <= tile) lo = mid; else hi = mid - 1;
        }
        return GroupOfTile{lo, gt.d_node_begin[lo], gt.d_node_begin[lo + 1], gt.d_tile_begin[lo], gt.d_tile_begin[lo + 1]};
    }
    GroupOfTile r{0, gt.node_begin[0], gt.node_begin[1], gt.tile_begin[0], gt.tile_begin[1]};
#pragma unroll
    for (int g = 1; g < MAX_GROUPS; ++g)
        if (g < gt.n && tile >= gt.tile_begin[g])
            r = GroupOfTile{g, gt.node_begin[g], gt.node_begin[g + 1], gt.tile_begin[g], gt.tile_begin[g + 1]};
    return r;
}

// Fold an inference BatchNormalization into the Dense layer that follows it (Keras: y = x*inv + (beta - mean*inv),
// inv = gamma / sqrt(var + eps)):  Wf[k][h] = inv[k] * W[k][h],  bf[h] = b[h] + sum_k (beta[k] - mean[k]*inv[k]) W[k][h].
// One workgroup per output column h, threads stride over k; the shift sum meets in a fixed-order LDS tree.
// Up to GNN_MAX_TYPES + 1 networks (one state network per node type + the output network) fold in ONE launch, which also
// zeroes the loop's flag words and iteration counter (two small arrays): three launches and two memsets less per forward.
struct FoldJob {
    const float *W, *b, *gamma, *beta, *mean, *var;
    float *Wf, *bf;
    int K, H, blk_begin;
    float eps;
    int centred;                           // 1: the consumer subtracts `mean` from its input rows itself (large-graph training): bf = b + sum beta W
    int dyn0, dyn1, dyn_w;                 // dyn_w > 0: bf = the shift sum over the weight rows [dyn0, dyn0 + dyn_w) and [dyn1, dyn1 + dyn_w) ALONE, without b - the
                                           // rest of the first layer's constant part comes from somewhere else (train_composite_big.hpp: once per step)
};
struct FoldArgs {
    FoldJob job[GNN_MAX_TYPES + 1];
    int n_jobs;
    int *zero_a; int n_a;
    float *zero_b; int n_b;
};
__global__ void __launch_bounds__(128) k_fold_bn(FoldArgs fa) {
    __shared__ float part[128];
    for (int i = blockIdx.x * 128 + threadIdx.x; i < fa.n_a; i += gridDim.x * 128) fa.zero_a[i] = 0;
    for (int i = blockIdx.x * 128 + threadIdx.x; i < fa.n_b; i += gridDim.x * 128) fa.zero_b[i] = 0.0f;
    int j = 0;
#pragma unroll
    for (int t = 1; t < GNN_MAX_TYPES + 1; ++t)
        if (t < fa.n_jobs && (int)blockIdx.x >= fa.job[t].blk_begin) j = t;
    // static selection (a runtime-indexed kernel-argument array would be copied to scratch memory)
    FoldJob jb = fa.job[0];
#pragma unroll
    for (int t = 1; t < GNN_MAX_TYPES + 1; ++t) if (t == j) jb = fa.job[t];
    const int h = blockIdx.x - jb.blk_begin, K = jb.K, H = jb.H;
    float acc = 0.0f;
    for (int k = threadIdx.x; k < K; k += 128) {
        float inv = 1.0f, shift = 0.0f;
        if (jb.gamma) {
            inv = jb.gamma[k] / sqrtf(jb.var[k] + jb.eps);
            shift = jb.centred ? jb.beta[k] : jb.beta[k] - jb.mean[k] * inv;
        }
        const float wv = jb.W[(size_t)k * H + h];
        jb.Wf[(size_t)k * H + h] = wv * inv;
        if (jb.dyn_w > 0 && !((k >= jb.dyn0 && k < jb.dyn0 + jb.dyn_w) || (k >= jb.dyn1 && k < jb.dyn1 + jb.dyn_w))) shift = 0.0f;
        acc = fmaf(shift, wv, acc);
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 64; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) jb.bf[h] = ((jb.b && jb.dyn_w == 0) ? jb.b[h] : 0.0f) + part[0];
}

// dst[i, :width] = src[i, :width] with independent leading dimensions; pads dst columns [width, ld_dst_fill) with 0.
__global__ void __launch_bounds__(256)
k_copy2d(const int *gate, const float *__restrict__ src, int ld_src, float *__restrict__ dst, int ld_dst, int rows,
         int width, int fill_to) {
    if (gate_closed(gate)) return;
    const size_t total = (size_t)rows * fill_to;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t rr = i / fill_to;
        const int cc = (int)(i % fill_to);
        dst[rr * ld_dst + cc] = cc < width ? src[rr * ld_src + cc] : 0.0f;
    }
}

// Hub pre-pass: one workgroup sums the arcs [seg_beg[s], seg_end[s]) of the by-destination CSR (a slice of a hub row)
// into the virtual state row n_rows + s of the SAME buffer: 16 lanes x 16 B per source row (SP = 64; fewer lanes for
// narrower rows), lane groups stride over the arcs, partials meet in LDS and are added in group order (deterministic).
template <int SP>
__global__ void __launch_bounds__(256)
k_heavy_segments(const int *gate, const int *__restrict__ seg_beg, const int *__restrict__ seg_end, int n_seg,
                 const int *__restrict__ src, const float *__restrict__ w, float *__restrict__ state, int n_rows) {
    if (gate_closed(gate)) return;
    constexpr int LPR = SP / 4, NG = 256 / LPR;
    __shared__ f32x4 part[256];
    const int g = threadIdx.x / LPR, l4 = threadIdx.x % LPR;
    for (int s = blockIdx.x; s < n_seg; s += gridDim.x) {
        const int beg = seg_beg[s], end = seg_end[s];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int e = beg + g; e < end; e += NG) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(state + (size_t)src[e] * SP + 4 * l4);
            if (w) acc += w[e] * v; else acc += v;
        }
        part[threadIdx.x] = acc;
        __syncthreads();
        if (g == 0) {
            f32x4 tot = part[l4];
            for (int gg = 1; gg < NG; ++gg) tot += part[gg * LPR + l4];
            *reinterpret_cast<f32x4 *>(state + (size_t)(n_rows + s) * SP + 4 * l4) = tot;
        }
        __syncthreads();
    }
}

// dst[m, :width] = src[idx[m], :width]  — row gather (packs the halo rows a peer needs into a contiguous send buffer);
// whole rows, 16 B per lane when width % 4 == 0 and both leading dimensions are multiples of 4.
__global__ void __launch_bounds__(256)
k_gather_rows(const float *__restrict__ src, int ld_src, const int *__restrict__ idx, int M, int width,
              float *__restrict__ dst, int ld_dst) {
    if ((width & 3) == 0 && (ld_src & 3) == 0 && (ld_dst & 3) == 0) {
        const int w4 = width >> 2;
        const size_t total = (size_t)M * w4;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
            const size_t m = i / w4;
            const int c = (int)(i % w4);
            reinterpret_cast<f32x4 *>(dst + m * ld_dst)[c] = reinterpret_cast<const f32x4 *>(src + (size_t)idx[m] * ld_src)[c];
        }
        return;
    }
    const size_t total = (size_t)M * width;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / width;
        const int c = (int)(i % width);
        dst[m * ld_dst + c] = src[(size_t)idx[m] * ld_src + c];
    }
}

// state_out = (k odd ? buf1 : buf0): picks the buffer that holds the state after k iterations, k read on device.
__global__ void __launch_bounds__(256)
k_select_state(const float *k_ptr, const float *__restrict__ first, const float *__restrict__ buf0,
               const float *__restrict__ buf1, int ld_buf, float *__restrict__ dst, int ld_dst, int rows, int width) {
    const int k = (int)(*k_ptr);
    const float *src = k == 0 ? first : ((k & 1) ? buf1 : buf0);      // `first`: state_0 where it was read in place
    if (src == dst) return;                                            // the last iteration wrote the caller's buffer directly
    const size_t total = (size_t)rows * width;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t rr = i / width;
        const int cc = (int)(i % width);
        dst[rr * ld_dst + cc] = src[rr * ld_buf + cc];
    }
}

}  // namespace gnn
