// Fused state-transition iteration for gfx950 (the hot kernel).  One launch = one iteration of the reference's
// `convergence` + the `condition` of the next iteration (GNN/Models/GNN.py:217-236, :196-214), for one node type:
//
//   for every destination node j of a 64-node tile (persistent workgroups, XCD-contiguous tile ranges):
//     A. coalesced CSR walk: SP/4 lanes own node j, each lane accumulates 16 B of every neighbour state row
//        (one 256-B row per 16 lanes at d = 64) in ascending-source order; own state row + aggregate go to LDS;
//     B. [state | agg] (64 x 2SP, LDS) x W1[state rows ; agg rows] (2SP x S, LDS, loaded once per workgroup) on the
//        f32 matrix cores (v_mfma_f32_16x16x4_f32 — exact f32 fma chain), + per-node constant C (labels, label
//        aggregates, arc aggregates, bias, folded BatchNormalization), activation;
//     C. per-node predicate  ||new - old||_2 > thr ||old||_2  from the accumulator registers (16-lane shuffles),
//        OR-reduced to one flag word per launch; new state rows staged through LDS and written as whole rows.
//
// HBM traffic per iteration = the algorithmic bytes of SURVEY §8d:  E(4 + 4S [+4]) + N(4 + 4S + 4S + 4S).
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_general.hpp"

namespace gnn {

struct FusedType {
    const int *rows;   // node ids of this type (nullptr = identity)
    int count;
    const float *Wf;   // folded first layer [in_dim x H] row-major, H == S
    int wrow_state, wrow_agg;
    int H, act;
};

struct FusedArgs {
    const int *gate;
    const int *rowptr, *src;
    const float *w, *row_scale;
    const float *state_in;
    float *state_out;
    const float *C; int ldC;
    FusedType tp;
    int S;
    float thr;
    int *flag_next;
    float *k_out; float k_val;
};

constexpr int FUSED_TM = 64;

template <int SP>
struct FusedCfg {
    static constexpr int LPR = SP / 4;                       // lanes per node row (16 B each)
    static constexpr int NPP = 256 / LPR;                    // nodes per pass of the 256-thread workgroup
    static constexpr int LDX = 2 * SP + 4;                   // Xs row stride (floats), rows stay 16-B aligned
    static constexpr int LDW = (SP % 32 == 0) ? SP + 16 : SP + 32;   // == 16 (mod 32): conflict-free B-fragment reads
    static constexpr int NCT = SP / 16;                      // 16-column MFMA tiles
    static constexpr size_t LDS_BYTES = sizeof(float) * (FUSED_TM * LDX + 2 * SP * LDW) + sizeof(int) * FUSED_TM;
};

template <int SP, bool HAS_W>
__global__ void __launch_bounds__(256, 2) k_state_fused(FusedArgs a) {
    if (gate_closed(a.gate)) return;
    using Cfg = FusedCfg<SP>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *Xs = reinterpret_cast<float *>(smem);                         // [64][LDX]  : [state | agg]
    float *Ws = Xs + FUSED_TM * Cfg::LDX;                                // [2SP][LDW] : W1 rows (state ; agg)
    int *jid = reinterpret_cast<int *>(Ws + 2 * SP * Cfg::LDW);          // [64] node id of each tile row (-1 = pad)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int S = a.S;

    // W1 -> LDS once per workgroup, zero padded to [2SP][SP]
    for (int i = tid; i < 2 * SP * SP; i += 256) {
        const int k = i / SP, n = i % SP;
        const int kk = k < SP ? k : k - SP;
        float v = 0.0f;
        if (kk < S && n < S) v = a.tp.Wf[(size_t)((k < SP ? a.tp.wrow_state : a.tp.wrow_agg) + kk) * a.tp.H + n];
        Ws[k * Cfg::LDW + n] = v;
    }

    // XCD-contiguous tile ranges: workgroups b, b+8, b+16.. share an XCD (round-robin dispatch); give each XCD one
    // contiguous slice of the node range so CSR / state / C streams and any graph locality stay in its L2.
    const int ntiles = (a.tp.count + FUSED_TM - 1) / FUSED_TM;
    const int nblk = gridDim.x, xcd = blockIdx.x & 7, lb = blockIdx.x >> 3;
    const int blk_per_xcd = (nblk + 7 - xcd) >> 3;          // blocks with this residue
    const int tpx = (ntiles + 7) >> 3;
    const int t_end = min(ntiles, (xcd + 1) * tpx);

    int any = 0;
    const int node_in_pass = tid / Cfg::LPR;
    const int l4 = tid % Cfg::LPR;

    for (int tile = xcd * tpx + lb; tile < t_end; tile += blk_per_xcd) {
        __syncthreads();   // previous tile's Xs fully consumed (also orders the W fill before first use)
        // ---- A. gather + aggregate -------------------------------------------------------------------------------
#pragma unroll 1
        for (int pass = 0; pass < FUSED_TM / Cfg::NPP; ++pass) {
            const int nl = pass * Cfg::NPP + node_in_pass;
            const int m = tile * FUSED_TM + nl;
            f32x4 own = {0.f, 0.f, 0.f, 0.f}, acc = {0.f, 0.f, 0.f, 0.f};
            int j = -1;
            if (m < a.tp.count) {
                j = a.tp.rows ? a.tp.rows[m] : m;
                own = *reinterpret_cast<const f32x4 *>(a.state_in + (size_t)j * SP + 4 * l4);
                const int beg = a.rowptr[j], end = a.rowptr[j + 1];
                int e = beg;
                for (; e + 4 <= end; e += 4) {
                    const int s0 = a.src[e], s1 = a.src[e + 1], s2 = a.src[e + 2], s3 = a.src[e + 3];
                    const f32x4 v0 = *reinterpret_cast<const f32x4 *>(a.state_in + (size_t)s0 * SP + 4 * l4);
                    const f32x4 v1 = *reinterpret_cast<const f32x4 *>(a.state_in + (size_t)s1 * SP + 4 * l4);
                    const f32x4 v2 = *reinterpret_cast<const f32x4 *>(a.state_in + (size_t)s2 * SP + 4 * l4);
                    const f32x4 v3 = *reinterpret_cast<const f32x4 *>(a.state_in + (size_t)s3 * SP + 4 * l4);
                    if (HAS_W) {
                        acc += a.w[e] * v0; acc += a.w[e + 1] * v1; acc += a.w[e + 2] * v2; acc += a.w[e + 3] * v3;
                    } else {
                        acc += v0; acc += v1; acc += v2; acc += v3;
                    }
                }
                for (; e < end; ++e) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(a.state_in + (size_t)a.src[e] * SP + 4 * l4);
                    if (HAS_W) acc += a.w[e] * v; else acc += v;
                }
                if (a.row_scale) acc *= a.row_scale[j];
            }
            *reinterpret_cast<f32x4 *>(Xs + nl * Cfg::LDX + 4 * l4) = own;
            *reinterpret_cast<f32x4 *>(Xs + nl * Cfg::LDX + SP + 4 * l4) = acc;
            if (l4 == 0) jid[nl] = j;
        }
        __syncthreads();

        // ---- B. [state | agg] . W1 on the matrix cores --------------------------------------------------------------
        f32x4 c[Cfg::NCT];
#pragma unroll
        for (int ct = 0; ct < Cfg::NCT; ++ct) c[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float *xrow = Xs + (16 * wave + r) * Cfg::LDX + g;
        const float *wcol = Ws + g * Cfg::LDW + r;
#pragma unroll 8
        for (int s4 = 0; s4 < 2 * SP / 4; ++s4) {
            const float av = xrow[4 * s4];
#pragma unroll
            for (int ct = 0; ct < Cfg::NCT; ++ct) {
                const float bv = wcol[4 * s4 * Cfg::LDW + 16 * ct];
                c[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c[ct], 0, 0, 0);
            }
        }

        // ---- C. epilogue: + C, activation, predicate, stage new rows -----------------------------------------------
        // C/D layout: col = 16*ct + (lane & 15), row = 16*wave + 4*(lane >> 4) + reg
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = 16 * wave + 4 * g + reg;
            const int j = jid[row];
            float d2 = 0.0f, n2 = 0.0f;
#pragma unroll
            for (int ct = 0; ct < Cfg::NCT; ++ct) {
                const int col = 16 * ct + r;
                float nv = 0.0f;
                if (j >= 0 && col < S) nv = activate(a.tp.act, c[ct][reg] + a.C[(size_t)j * a.ldC + col]);
                const float ov = Xs[row * Cfg::LDX + col];
                const float d = nv - ov;
                d2 = fmaf(d, d, d2);
                n2 = fmaf(ov, ov, n2);
                Xs[row * Cfg::LDX + col] = nv;      // same lane read `ov` from this slot: no cross-lane hazard
            }
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) {
                d2 += __shfl_xor(d2, off, 16);
                n2 += __shfl_xor(n2, off, 16);
            }
            if (j >= 0 && sqrtf(d2) > a.thr * sqrtf(n2)) any = 1;
        }
        __syncthreads();
        // whole-row (4*SP bytes) coalesced stores of the new state
#pragma unroll 1
        for (int pass = 0; pass < FUSED_TM / Cfg::NPP; ++pass) {
            const int nl = pass * Cfg::NPP + node_in_pass;
            const int j = jid[nl];
            if (j >= 0)
                *reinterpret_cast<f32x4 *>(a.state_out + (size_t)j * SP + 4 * l4) =
                    *reinterpret_cast<const f32x4 *>(Xs + nl * Cfg::LDX + 4 * l4);
        }
    }

    any = __syncthreads_or(any);
    if (tid == 0) {
        if (any && a.flag_next) atomicOr(a.flag_next, 1);
        if (blockIdx.x == 0 && a.k_out) *a.k_out = a.k_val;
    }
}

template <int SP>
int launch_fused_sp(const FusedArgs &fa, int grid, hipStream_t st) {
    const size_t lds = FusedCfg<SP>::LDS_BYTES;
    hipError_t e;
    if (fa.w) {
        static bool attr_w = false;
        if (!attr_w) { e = hipFuncSetAttribute((const void *)k_state_fused<SP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e != hipSuccess) return 1; attr_w = true; }
        k_state_fused<SP, true><<<grid, 256, lds, st>>>(fa);
    } else {
        static bool attr_n = false;
        if (!attr_n) { e = hipFuncSetAttribute((const void *)k_state_fused<SP, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e != hipSuccess) return 1; attr_n = true; }
        k_state_fused<SP, false><<<grid, 256, lds, st>>>(fa);
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// Launch one fused iteration (one launch per node type).  `type_of(t)` yields the FusedType of type t.
template <typename TypeGetter>
int launch_state_fused(const gnn_loop_args_t &a, int T, int N, int S, int SP, const int *gate, const float *src,
                       float *dst, const float *C, int ldC, TypeGetter type_of, int *flag_next, float *k_out,
                       float k_val, hipStream_t st) {
    (void)N;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 1;
        n_cu = prop.multiProcessorCount;
    }
    bool k_written = false;
    for (int t = 0; t < T; ++t) {
        FusedArgs fa;
        fa.gate = gate;
        fa.rowptr = a.adjacency.rowptr; fa.src = a.adjacency.src; fa.w = a.adjacency.w; fa.row_scale = a.adjacency.row_scale;
        fa.state_in = src; fa.state_out = dst;
        fa.C = C; fa.ldC = ldC;
        fa.tp = type_of(t);
        fa.S = S; fa.thr = a.state_threshold;
        fa.flag_next = flag_next;
        fa.k_out = k_written ? nullptr : k_out; fa.k_val = k_val;
        if (fa.tp.count == 0) continue;
        k_written = true;
        const int ntiles = (fa.tp.count + FUSED_TM - 1) / FUSED_TM;
        int grid = std::min(ntiles, 2 * n_cu);
        grid = std::max(8, (grid + 7) / 8 * 8);
        int rc;
        switch (SP) {
            case 16: rc = launch_fused_sp<16>(fa, grid, st); break;
            case 32: rc = launch_fused_sp<32>(fa, grid, st); break;
            case 64: rc = launch_fused_sp<64>(fa, grid, st); break;
            default: return 1;
        }
        if (rc) return rc;
    }
    return 0;
}

}  // namespace gnn
