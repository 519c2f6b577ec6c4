// Many small batches (a MUTAG batch of 32 molecules: ~1 k nodes, ~2 k arcs), each an independent convergence group
// (gnn_loop_args_t::group_node_begin): ONE workgroup = one CU per batch, the batch's whole state matrix resident in LDS for
// every iteration of the reference's `while condition: convergence` (GNN/Models/GNN.py:265, :196-236).
//
// The small whole-loop kernel (kernel_state_small.hpp) spreads a batch over ~15 CUs and pays five cross-XCD round trips per
// iteration for it (rows out, drain, arrive, poll, rows in): 6.7 us per iteration with at most ~16 batches per launch.  A
// batch's state is 935 x 32 floats = 120 KB: it fits the 160 KB of ONE CU, where an iteration needs no global memory traffic
// and no grid barrier at all.  The iteration is then bound by the CU's f32 matrix pipe (935 x 64 x 32 multiply-adds = ~7 us)
// - the same latency - but 256 batches run side by side instead of 16.
//
//   * state [n, SP] floats in LDS (rows of 16-byte chunks); neighbour rows are gathered with ds_read_b128 straight into MFMA
//     A-fragment registers: lane (r, g) of a wave owns chunks g, g + 4, .. of tile row r, i.e. the k index is permuted
//     (A[q][e] = row[16 q + 4 g + e]) and W1's rows are held in the same order - in REGISTERS (2 SP / 4 k-steps x SP / 16 column
//     tiles values per lane), so the weights cost no LDS either; no transposition, no staging tile, no barrier inside a tile.
//   * a wave owns whole 16-node tiles (tile = wave, wave + 16, ..); the new rows go to a staging buffer in global memory (L2)
//     until every wave has finished reading the old state (one workgroup barrier), come back into LDS with L1-bypassing
//     (sc1) 16-byte loads, and a second barrier opens the next iteration: two barriers per iteration, both inside the CU.  (Holding
//     the new rows in registers instead needs ~80 of them per wave and a fully unrolled tile loop: hipcc rolled it back and
//     indexed the arrays in scratch.)
//   * each node's CSR row sits in LDS as a 16-byte record (in-degree, row scale, first 4 source ids as 16-bit local ids:
//     molecule graphs rarely exceed 4; longer rows read the rest from the CSR in global memory); the per-node constant C is
//     re-read from L2 each iteration, one tile ahead (4 H1 bytes per node; 120 KB per batch stays cached).
//   * predicate, activation and iteration count as everywhere else; k_out[g] is written once at the end.
// Used when every group fits (n_g * (4 SP + 16) <= LDS_BUDGET_BYTES), SP is 16 or 32 and the state network has one layer.
#pragma once
#include <hip/hip_runtime.h>
#include "kernel_state_fused2.hpp"
#include "kernel_state_fused4.hpp"      // activate4
#include "buffer_ops.hpp"

namespace gnn {

constexpr int LDS_NW = 16;                                   // waves per workgroup (4 per SIMD)
constexpr size_t LDS_BUDGET_BYTES = 158 * 1024;              // state rows + CSR records of one group; the rest of the 160 KB: flags, slack

struct LdsArgs {
    const int *node_begin;          // device [n_groups + 1]
    const int *tile64_begin;        // device [n_groups + 1]: 64-node tiles of the set-up kernel (pred0 is indexed by them)
    const int *pred0;               // state_0's predicate per 64-node tile (k_setup_small)
    const int *rowptr, *src; const float *w, *row_scale;
    const float *state0; int ld_s0; // [N, ld_s0]: caller's state_0 (or the label matrix when state_vect_dim == 0)
    const float *C; int ldC;
    const float *Wf; int wrow_state, wrow_agg, H, act;
    int S, max_iteration, no_exit;
    float thr;
    float *stage;                   // [N, SP] scratch rows in global memory (L2): the new state of an iteration on its way back to LDS
    float *state_out;               // [N, S] compact
    float *k_out;                   // [n_groups]
};

// Sum over the 16 lanes of a DPP row, complete in lane 15 of the row (row_shr 1, 2, 4, 8 with out-of-row reads as zero): four
// VALU adds instead of four LDS permutes + adds per value.
__device__ __forceinline__ float row16_sum_to_lane15(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xf, 0xf, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x112, 0xf, 0xf, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x114, 0xf, 0xf, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x118, 0xf, 0xf, true));
    return x;
}

// One node's CSR row as the loop needs it, 16 bytes in LDS: in-degree, row scale, the first 4 source ids (local, 16 bit each).
struct LdsRec { unsigned id01, id23; int deg; float scale; };

template <int SP, bool HAS_W>
__global__ void __launch_bounds__(64 * LDS_NW, 4) k_state_lds(LdsArgs a) {
    constexpr int NQ = SP / 16;                   // 16-byte chunks of a row owned by one lane
    constexpr int NCT = SP / 16;                  // 16-column output tiles
    constexpr int KS = 2 * SP / 4;                // MFMA k-steps over [state | agg]
    extern __shared__ __attribute__((aligned(16))) char smem_lds[];
    __shared__ int moving_s[2];                   // "some node of this group still moves", iteration it uses [it & 1]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int grp = blockIdx.x;
    const int nb = a.node_begin[grp], ne = a.node_begin[grp + 1], n = ne - nb;
    const int S = a.S;
    const int n_tiles = (n + 15) >> 4;
    float *St = reinterpret_cast<float *>(smem_lds);                        // [n][SP]
    LdsRec *Rec = reinterpret_cast<LdsRec *>(St + (size_t)n * SP);          // [n]

    // ---- W1 as B fragments in registers: k-step (half, q, e) supplies column kcol = 16 q + 4 g + e of that half ----------------
    float wreg[KS][NCT];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int half = ks / (SP / 4), qe = ks % (SP / 4);
        const int kcol = 16 * (qe / 4) + 4 * g + (qe & 3);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const int ncol = 16 * ct + r;
            wreg[ks][ct] = (kcol < S && ncol < a.H) ? a.Wf[(size_t)((half ? a.wrow_agg : a.wrow_state) + kcol) * a.H + ncol] : 0.0f;
        }
    }
    // ---- state_0 and the CSR records into LDS ------------------------------------------------------------------------------------------
    for (int i = tid; i < n * SP; i += 64 * LDS_NW) {
        const int j = i / SP, c = i % SP;
        St[i] = c < S ? a.state0[(size_t)(nb + j) * a.ld_s0 + c] : 0.0f;
    }
    for (int j = tid; j < n; j += 64 * LDS_NW) {
        const int beg = a.rowptr[nb + j], end = a.rowptr[nb + j + 1];
        unsigned id[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) id[u] = beg + u < end ? (unsigned)(a.src[beg + u] - nb) : 0u;
        Rec[j] = LdsRec{id[0] | (id[1] << 16), id[2] | (id[3] << 16), end - beg, a.row_scale ? a.row_scale[nb + j] : 1.0f};
    }
    if (tid == 0) { moving_s[0] = 0; moving_s[1] = 0; }
    int run = a.no_exit;
    if (!run) {
        const int t0 = a.tile64_begin[grp], t1 = a.tile64_begin[grp + 1];
        int v = 0;
        for (int i = t0 + lane; i < t1; i += 64) v |= a.pred0[i];
        run = __any(v != 0);
    }
    __syncthreads();

    // (the kernel is bound by its instruction count, not by the matrix pipe: 32-bit offsets off scalar bases, predicated-off
    // loads, one activation switch per 4 values)
    const __amdgpu_buffer_rsrc_t r_C = buf_rsrc(a.C + (size_t)nb * a.ldC), r_stage = buf_rsrc(a.stage + (size_t)nb * SP);
    auto load_c = [&](int t, f32x4 *c) {            // the per-node constant C of tile t in accumulator layout (row 4 g + reg, column 16 ct + r)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int rl = 16 * t + 4 * g + reg, col = 16 * ct + r;
                c[ct][reg] = buf_ld_f32(r_C, (rl < n && col < a.H) ? ((unsigned)rl * (unsigned)a.ldC + (unsigned)col) * 4u : BUF_OFF);
            }
    };

    int k_done = 0;
    for (int it = 0; run && it < a.max_iteration; ++it) {
        f32x4 cn[NCT];                                                         // C of the NEXT tile: in flight during this tile's gather and MFMAs
        int any = 0;
        if (wave < n_tiles) load_c(wave, cn);
#pragma unroll 1
        for (int t = wave; t < n_tiles; t += LDS_NW) {
            f32x4 c[NCT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) c[ct] = cn[ct];
            if (t + LDS_NW < n_tiles) load_c(t + LDS_NW, cn);
            const int jl = 16 * t + r;                                         // local node of this lane's A row
            const bool on = jl < n;
            const LdsRec rec = on ? Rec[jl] : LdsRec{0u, 0u, 0, 1.0f};
            // own row chunks and the neighbour sum, both in A-fragment order
            f32x4 own[NQ], agg[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                own[q] = on ? *reinterpret_cast<const f32x4 *>(St + jl * SP + 16 * q + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
                agg[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            const int beg = on ? 0 : 0;                                        // (records carry the first 4 ids; the row's extent is only needed beyond)
            (void)beg;
            const unsigned ids4[4] = {rec.id01 & 0xFFFFu, rec.id01 >> 16, rec.id23 & 0xFFFFu, rec.id23 >> 16};
            float w4[4] = {1.0f, 1.0f, 1.0f, 1.0f};
            if (HAS_W && on) {
                const int b0 = a.rowptr[nb + jl];
#pragma unroll
                for (int u = 0; u < 4; ++u) w4[u] = u < rec.deg ? a.w[b0 + u] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (u < rec.deg) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const f32x4 x = *reinterpret_cast<const f32x4 *>(St + ids4[u] * SP + 16 * q + 4 * g);
                        if (HAS_W) agg[q] += w4[u] * x; else agg[q] += x;
                    }
                }
            }
            if (__any(rec.deg > 4)) {                                          // rows with more than 4 arcs: the rest from the CSR in global memory
                const int b0 = on ? a.rowptr[nb + jl] : 0, e1 = b0 + rec.deg;
                for (int e = b0 + 4; __any(e < e1); e += 4) {
                    int id[4]; float wv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool ok = e + u < e1;
                        id[u] = ok ? a.src[e + u] - nb : 0;
                        wv[u] = ok ? (HAS_W ? a.w[e + u] : 1.0f) : 0.0f;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (e + u < e1) {
#pragma unroll
                            for (int q = 0; q < NQ; ++q) {
                                const f32x4 x = *reinterpret_cast<const f32x4 *>(St + id[u] * SP + 16 * q + 4 * g);
                                if (HAS_W) agg[q] += wv[u] * x; else agg[q] += x;
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) agg[q] *= rec.scale;
            // four independent MFMA chains instead of two: the state half accumulates onto C, the agg half onto zero, summed at the end
            f32x4 c2[NCT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) c2[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int qe = 0; qe < SP / 4; ++qe) {
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    c[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(own[qe / 4][qe & 3], wreg[qe][ct], c[ct], 0, 0, 0);
                    c2[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(agg[qe / 4][qe & 3], wreg[SP / 4 + qe][ct], c2[ct], 0, 0, 0);
                }
            }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) c[ct] += c2[ct];
            // activation + predicate against the old rows (still in LDS); the new rows leave for the staging buffer
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) activate4(a.act, c[ct]);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int rl = 16 * t + 4 * g + reg;
                const bool rin = rl < n;
                float d2 = 0.0f, n2 = 0.0f;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const int col = 16 * ct + r;
                    const float v = (rin && col < S) ? c[ct][reg] : 0.0f;
                    const float o = rin ? St[rl * SP + col] : 0.0f;
                    const float d = v - o;
                    d2 = fmaf(d, d, d2); n2 = fmaf(o, o, n2);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r_stage, rin ? (int)(((unsigned)rl * SP + col) * 4u) : (int)BUF_OFF, 0, 0);
                }
                d2 = row16_sum_to_lane15(d2); n2 = row16_sum_to_lane15(n2);
                if (rin && r == 15 && sqrtf(d2) > a.thr * sqrtf(n2)) any = 1;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // this wave's staged rows are in L2
        __syncthreads();                                                       // every wave is done reading the old state
        // two flag words: the one of the NEXT iteration is cleared here - every wave has read it (end of the previous iteration)
        // before it arrived at the barrier above, and the barrier below orders the clear before the next iteration's stores
        if (tid == 0) moving_s[(it + 1) & 1] = 0;
        if (any) moving_s[it & 1] = 1;                                         // benign race: every writer stores 1
        {   // staged rows back into LDS: sc1 loads, served by the L2 the stores went to (the CU's L1 may still hold last
            // iteration's lines of the staging buffer)
            const __amdgpu_buffer_rsrc_t rs = r_stage;
            for (int i = tid; i < n * (SP / 4); i += 64 * LDS_NW) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, i * 16, 0, 16);
                *reinterpret_cast<u32x4 *>(St + 4 * i) = v;
            }
        }
        __syncthreads();
        k_done = it + 1;
        if (!a.no_exit && moving_s[it & 1] == 0) break;                        // uniform: read after the barrier
    }
    // ---- result rows to the caller's compact buffer, k of this group ------------------------------------------------------------------
    for (int i = tid; i < n * S; i += 64 * LDS_NW) {
        const int j = i / S, c = i % S;
        a.state_out[(size_t)(nb + j) * S + c] = St[j * SP + c];
    }
    if (tid == 0) a.k_out[grp] = (float)k_done;
}

inline size_t lds_state_bytes(int n_nodes, int SP) { return (size_t)n_nodes * (SP * sizeof(float) + sizeof(LdsRec)); }
inline bool lds_group_fits(int n_nodes, int SP) { return lds_state_bytes(n_nodes, SP) <= LDS_BUDGET_BYTES && n_nodes < 65536; }

template <int SP, bool HAS_W>
int launch_lds_one(const LdsArgs &la, int n_groups, size_t lds_bytes, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void *)k_state_lds<SP, HAS_W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BUDGET_BYTES) != hipSuccess) return 1;
        attr = true;
    }
    GNN_SET_KERNEL_NAME("k_state_lds<%d,%s>", SP, HAS_W ? "true" : "false");
    k_state_lds<SP, HAS_W><<<n_groups, 64 * LDS_NW, lds_bytes, st>>>(la);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// lds_bytes: state bytes of the largest group (every workgroup requests the same amount)
inline int launch_lds(const LdsArgs &la, int SP, int n_groups, size_t lds_bytes, hipStream_t st) {
    switch (SP) {
        case 16: return la.w ? launch_lds_one<16, true>(la, n_groups, lds_bytes, st) : launch_lds_one<16, false>(la, n_groups, lds_bytes, st);
        case 32: return la.w ? launch_lds_one<32, true>(la, n_groups, lds_bytes, st) : launch_lds_one<32, false>(la, n_groups, lds_bytes, st);
        default: return 2;
    }
}

}  // namespace gnn
