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
//   * the matrix core is fed TRANSPOSED (round 3): W1 is the A operand, the node rows the B operand - both have the same register
//     layout for v_mfma_f32_16x16x4_f32 - so D[i][j] = new[row j][16 ct + i] leaves lane (r, g) holding FOUR CONSECUTIVE COLUMNS of
//     its own row: the constant C arrives as 16-byte loads, the predicate compares against the own-row chunks already in
//     registers, the new rows leave as 16-byte stores: a quarter of the epilogue's instructions (it was two thirds of an iteration);
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
    // group sets (gnn_loop_args_t::group_set_begin): the groups [set_first[g], set_first[g] + set_size[g]) share the loop's condition
    const int *set_first, *set_size;        // device [n_groups]
    unsigned long long *set_bar;            // [2 * n_groups], zero before the launch; NULL: every group on its own
    unsigned long long wait_ticks;          // bound of a set-barrier wait (buffer_ops.hpp: wait_until)
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

// DB (round 3): the group's state fits LDS TWICE (n (8 SP + 16) <= 158 KB: 594 nodes at 32-wide rows - what the planner's balanced
// cuts produce when there are CUs to spare): iteration it reads buffer it & 1 and writes the other one with 16-byte LDS stores.
// No staging buffer in L2, no store drain, no copy back, ONE workgroup barrier per iteration instead of two.
template <int SP, bool HAS_W, bool DB>
__global__ void __launch_bounds__(64 * LDS_NW, 4) k_state_lds(LdsArgs a) {
    constexpr int NQ = SP / 16;                   // 16-byte chunks of a row owned by one lane
    constexpr int NCT = SP / 16;                  // 16-column output tiles
    constexpr int KS = 2 * SP / 4;                // MFMA k-steps over [state | agg]
    constexpr int CPB = 5;                        // 16-byte chunks a thread copies back per trip (all in flight together)
    extern __shared__ __attribute__((aligned(16))) char smem_lds[];
    __shared__ int moving_s[3];                   // "some node of this group still moves": iteration it uses [it % 3] (DB) / [it & 1]
    __shared__ int set_go;                        // group sets: does any group of the set still move?

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int grp = blockIdx.x;
    const int nb = a.node_begin[grp], ne = a.node_begin[grp + 1], n = ne - nb;
    const int S = a.S;
    const int n_tiles = (n + 15) >> 4;
    float *St = reinterpret_cast<float *>(smem_lds);                        // [n][SP] (DB: two of them, iteration it reads the one at (it & 1) n SP)
    LdsRec *Rec = reinterpret_cast<LdsRec *>(St + (size_t)(DB ? 2 : 1) * n * SP);   // [n]
    float *const St_base = St;

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
    if (tid == 0) { moving_s[0] = 0; moving_s[1] = 0; moving_s[2] = 0; }
    // the groups of a set leave the loop together: state_0's predicate over the tiles of the whole set, one flag exchange per iteration
    const int set_lo = a.set_bar ? a.set_first[grp] : grp, set_n = a.set_bar ? a.set_size[grp] : 1;
    unsigned moved_seen[2] = {0u, 0u};
    int timed_out = 0;
    int run = a.no_exit;
    if (!run) {
        const int t0 = a.tile64_begin[set_lo], t1 = a.tile64_begin[set_lo + set_n];
        int v = 0;
        for (int i = t0 + lane; i < t1; i += 64) v |= a.pred0[i];
        run = __any(v != 0);
    }
    __syncthreads();

    // (the kernel is bound by its instruction count, not by the matrix pipe: 32-bit offsets off scalar bases, predicated-off
    // loads, 16-byte accesses everywhere, one activation switch per 4 values)
    const __amdgpu_buffer_rsrc_t r_C = buf_rsrc(a.C + (size_t)nb * a.ldC), r_stage = buf_rsrc(a.stage + (size_t)nb * SP);
    auto load_c = [&](int t, f32x4 *c) {            // the per-node constant C of tile t, row-major: row 16 t + r, columns 16 ct + 4 g ..
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const int rl = 16 * t + r, col = 16 * ct + 4 * g;
            c[ct] = buf_ld_f32x4(r_C, (rl < n && col < a.H) ? ((unsigned)rl * (unsigned)a.ldC + (unsigned)col) * 4u : BUF_OFF);
        }
    };

    int k_done = 0;
    for (int it = 0; run && it < a.max_iteration; ++it) {
        float *Snew = St_base;
        if (DB) { St = St_base + (size_t)(it & 1) * n * SP; Snew = St_base + (size_t)((it + 1) & 1) * n * SP; }
        f32x4 cn[NCT];                                                         // C of the NEXT tile: in flight during this tile's gather and MFMAs
        int any = 0;
        if (wave < n_tiles) load_c(wave, cn);
#pragma unroll 1
        for (int t = wave; t < n_tiles; t += LDS_NW) {
            f32x4 c[NCT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) c[ct] = cn[ct];
            if (t + LDS_NW < n_tiles) load_c(t + LDS_NW, cn);
            const int jl = 16 * t + r;                                         // local node of this lane's row
            const bool on = jl < n;
            const LdsRec rec = on ? Rec[jl] : LdsRec{0u, 0u, 0, 1.0f};
            // own row chunks and the neighbour sum, both in fragment order
            f32x4 own[NQ], agg[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                own[q] = on ? *reinterpret_cast<const f32x4 *>(St + jl * SP + 16 * q + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
                agg[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
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
            // two independent MFMA chains per column tile: the state half accumulates onto C, the agg half onto zero, summed at the end.
            // Operands swapped (weights = A, rows = B): the result is row-major - lane (r, g) gets columns 16 ct + 4 g .. + 3 of row r.
            f32x4 c2[NCT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) c2[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int qe = 0; qe < SP / 4; ++qe) {
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    c[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[qe][ct], own[qe / 4][qe & 3], c[ct], 0, 0, 0);
                    c2[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[SP / 4 + qe][ct], agg[qe / 4][qe & 3], c2[ct], 0, 0, 0);
                }
            }
            // activation + predicate against the old row chunks (still in registers); the new rows leave for the staging buffer
            float d2 = 0.0f, n2 = 0.0f;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                f32x4 v = c[ct] + c2[ct];
                activate4(a.act, v);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = (on && 16 * ct + 4 * g + e < S) ? v[e] : 0.0f;
                    const float o = own[ct][e], d = v[e] - o;
                    d2 = fmaf(d, d, d2); n2 = fmaf(o, o, n2);
                }
                if (DB) {
                    if (on) *reinterpret_cast<f32x4 *>(Snew + jl * SP + 16 * ct + 4 * g) = v;
                } else {
                    const u32x4 bits = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                    __builtin_amdgcn_raw_buffer_store_b128(bits, r_stage, on ? (int)(((unsigned)jl * SP + 16u * ct + 4u * g) * 4u) : (int)BUF_OFF, 0, 0);
                }
            }
            d2 += __shfl_xor(d2, 16, 64); d2 += __shfl_xor(d2, 32, 64);        // the four lane groups of a row
            n2 += __shfl_xor(n2, 16, 64); n2 += __shfl_xor(n2, 32, 64);
            if (on && sqrtf(d2) > a.thr * sqrtf(n2)) any = 1;
        }
        int mv_slot = it & 1;
        if (DB) {
            // three rotating flag words: iteration it sets [it % 3] before the barrier and reads it after; the word of iteration it + 2
            // is cleared after this barrier - its last readers (iteration it - 1) are behind every wave, its next writers two barriers away
            mv_slot = it % 3;
            if (any) moving_s[mv_slot] = 1;                                    // benign race: every writer stores 1
            __syncthreads();                                                   // new rows complete, old rows no longer read
            if (tid == 0) moving_s[(it + 2) % 3] = 0;
        } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // this wave's staged rows are in L2
        __syncthreads();                                                       // every wave is done reading the old state
        // two flag words: the one of the NEXT iteration is cleared here - every wave has read it (end of the previous iteration)
        // before it arrived at the barrier above, and the barrier below orders the clear before the next iteration's stores
        if (tid == 0) moving_s[(it + 1) & 1] = 0;
        if (any) moving_s[it & 1] = 1;                                         // benign race: every writer stores 1
        {   // staged rows back into LDS: sc1 loads, served by the L2 the stores went to (the CU's L1 may still hold last iteration's
            // lines of the staging buffer).  All of a thread's loads are issued before the first is stored (they are independent;
            // one at a time they cost a round trip to L2 each: up to nine per iteration).
            const __amdgpu_buffer_rsrc_t rs = r_stage;
            const int total = n * (SP / 4);
            for (int i0 = tid; i0 < total; i0 += 64 * LDS_NW * CPB) {
                u32x4 v[CPB];
#pragma unroll
                for (int u = 0; u < CPB; ++u) {
                    const int i = i0 + u * 64 * LDS_NW;
                    v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, i < total ? i * 16 : (int)BUF_OFF, 0, 16);
                }
#pragma unroll
                for (int u = 0; u < CPB; ++u) {
                    const int i = i0 + u * 64 * LDS_NW;
                    if (i < total) *reinterpret_cast<u32x4 *>(St + 4 * i) = v[u];
                }
            }
        }
        __syncthreads();
        }
        if (DB) St = Snew;                                                     // (what the result copy below reads if the loop ends here)
        k_done = it + 1;
        if (set_n > 1) {                                                       // (uniform per workgroup)
            // one 64-bit add carries this group's arrival and "some node of mine still moves"; the set's total tells every member
            // whether ANY of them moves (k_state_small's grid barrier, over the set's few workgroups and without any state rows)
            if (tid == 0) {
                unsigned long long *ctr = a.set_bar + 2 * set_lo + (it & 1);
                __hip_atomic_fetch_add(ctr, 1ull + ((unsigned long long)(moving_s[mv_slot] ? 1u : 0u) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned target = (unsigned)(it / 2 + 1) * (unsigned)set_n;
                unsigned long long v = 0;
                if (!wait_until(a.wait_ticks, [&]() { v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return (unsigned)v >= target; }))
                    timed_out = 1;                                             // a member never arrived (not resident?): reported through k
                const unsigned moved = (unsigned)(v >> 32);
                set_go = timed_out ? -1 : ((moved != moved_seen[it & 1]) ? 1 : 0);
                moved_seen[it & 1] = moved;
            }
            __syncthreads();
            // an expired wait ends this group's loop at once: its result is void (k < 0), and waiting out the bound again in each of the
            // remaining iterations would hold the CU - and every member queued behind it - for max_iteration x the bound
            if (set_go < 0) break;
            if (!a.no_exit && set_go == 0) break;
        } else if (!a.no_exit && moving_s[mv_slot] == 0) break;                // uniform: read after the barrier
    }
    // ---- result rows to the caller's compact buffer, k of this group ------------------------------------------------------------------
    for (int i = tid; i < n * S; i += 64 * LDS_NW) {
        const int j = i / S, c = i % S;
        a.state_out[(size_t)(nb + j) * S + c] = St[j * SP + c];
    }
    if (tid == 0) a.k_out[grp] = timed_out ? -1.0e9f : (float)k_done;
}

inline size_t lds_state_bytes(int n_nodes, int SP, bool db = false) { return (size_t)n_nodes * ((db ? 2 : 1) * SP * sizeof(float) + sizeof(LdsRec)); }
inline bool lds_group_fits(int n_nodes, int SP) { return lds_state_bytes(n_nodes, SP) <= LDS_BUDGET_BYTES && n_nodes < 65536; }
inline bool lds_group_fits_twice(int n_nodes, int SP) { return lds_state_bytes(n_nodes, SP, true) <= LDS_BUDGET_BYTES; }

template <int SP, bool HAS_W, bool DB>
int launch_lds_one(const LdsArgs &la_in, int n_groups, size_t lds_bytes, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void *)k_state_lds<SP, HAS_W, DB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BUDGET_BYTES) != hipSuccess) return 1;
        attr = true;
    }
    LdsArgs la = la_in;
    la.wait_ticks = wait_ticks();
    if (la.set_bar) {            // groups of a set wait for each other: every workgroup of the launch must be resident at once
        int dev = 0, n_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 1;
        if (!persistent_fits((const void *)k_state_lds<SP, HAS_W, DB>, 64 * LDS_NW, lds_bytes, n_groups, n_cu)) return 2;
    }
    GNN_SET_KERNEL_NAME("k_state_lds<%d,%s,%s>", SP, HAS_W ? "true" : "false", DB ? "true" : "false");
    k_state_lds<SP, HAS_W, DB><<<n_groups, 64 * LDS_NW, lds_bytes, st>>>(la);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// max_nodes: nodes of the largest group (every workgroup requests LDS for that one); the double-buffered form where it fits twice
inline int launch_lds(const LdsArgs &la, int SP, int n_groups, int max_nodes, hipStream_t st) {
    const bool db = lds_group_fits_twice(max_nodes, SP);
    const size_t bytes = std::max<size_t>(lds_state_bytes(max_nodes, SP, db), 90 * 1024);     // one workgroup per CU either way
#define LDS_CASE(SPV)                                                                                                            \
    case SPV:                                                                                                                    \
        if (db) return la.w ? launch_lds_one<SPV, true, true>(la, n_groups, bytes, st) : launch_lds_one<SPV, false, true>(la, n_groups, bytes, st); \
        return la.w ? launch_lds_one<SPV, true, false>(la, n_groups, bytes, st) : launch_lds_one<SPV, false, false>(la, n_groups, bytes, st);
    switch (SP) {
        LDS_CASE(16)
        LDS_CASE(32)
        default: return 2;
    }
#undef LDS_CASE
}

}  // namespace gnn
