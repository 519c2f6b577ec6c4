// Small graphs (a merged MUTAG batch: ~1 k nodes, ~2 k arcs): the WHOLE convergence loop in one launch.
// One workgroup owns one 64-node tile for every iteration of the reference's `while condition: convergence`
// (GNN/Models/GNN.py:265, :196-236), so everything that does not change between iterations stays on chip:
//   * W1 in LDS, the per-node constant C in the MFMA accumulator-init registers,
//   * the CSR of the tile's nodes (row pointers, first 16 source ids, weights, row scale) in registers.
// An iteration is then one round trip for the neighbour rows, the MFMA tile, the row stores, and a grid barrier; with
// one launch per iteration the same work also pays the kernel boundary, the gate, the W1 fill and the dependent chain
// node id -> row pointers -> source ids before the first row arrives.  Measured on MUTAG batches of 32 (d = 32): 9.0 us
// per iteration with one launch each, 6.6 us here; what is left is ~5 cross-XCD round trips of ~1 us (rows in, rows out
// + drain, counter add, counter poll): compiling out the neighbour loads or the barrier wait saves ~1 us each.
//
// Grid barrier = monotonic arrival counter in device memory, in the write-through form of the CDNA guide's hand-off
// (Guideline 16 / MI355X_MICROARCH.md "Valid forms"): every state row is stored `sc1` (16 B per lane, whole 128-B lines
// per instruction) and every load of a state row is an `sc1` buffer load, so neither an L2 write-back nor an L1
// invalidate is needed; each wave drains its stores (vmcnt(0)), the workgroup meets at its barrier, ONE lane adds to the
// counter (agent-scope atomic), polls it with agent-scope loads, and the other waves leave through a second workgroup
// barrier.  Correct for any placement of the workgroups over XCDs.  One workgroup per CU (the LDS request is padded
// past half a CU's LDS), every workgroup resident (grid <= CUs, checked by the launcher); polls sleep and are bounded.
// The convergence predicate travels with the arrival: each workgroup adds 1 + (some node still moves ? 2^32 : 0) to a
// 64-bit counter, so the total every workgroup reads after the barrier also says whether the next iteration runs.
#pragma once
#include <hip/hip_runtime.h>
#include "kernel_state_fused2.hpp"
#include "buffer_ops.hpp"

namespace gnn {

struct SmallArgs {
    Fused2Args f;            // f.state_in = source of iteration 0; f.gate / f.flag_next / f.k_val unused
    float *buf[2];           // iteration it writes buf[(it + 1) & 1]; iteration it > 0 reads buf[it & 1]
    int max_iteration;
    int no_exit;             // GNN_FLAG_NO_EARLY_EXIT
    int *flags;              // [max_iteration + 2], flags[0] = predicate of state_0 (set before the launch)
    const int *pred0;        // not null: the predicate of state_0 as n_pred0 words to OR (k_setup_small), flags[0] unused
    int n_pred0;
    float *state_final;      // not null: the loop's result rows go here as well ([n, ld_final], compact): no select pass
    int ld_final;
    GroupTab groups;         // n > 0 (homogeneous graphs): independent loops, one per group of tiles; each group has its own
                             // two arrival counters (bar + 2 * g), its own k (k_out[g]) and leaves the loop on its own
    unsigned long long *bar; // two arrival counters (even / odd iterations), zero before the launch:
                             // low word = arrivals, high word = workgroups that still saw a node move
    unsigned long long wait_ticks;   // bound of a barrier wait (buffer_ops.hpp: wait_until)
};

constexpr int small_waves(int SP) { return SP == 16 ? 4 : 8; }   // 16-wide rows: 4 lanes per row, 256 threads cover a 64-node tile in one pass

template <int SP, bool HAS_W, bool L2>
__global__ void __launch_bounds__(64 * small_waves(SP), 2) k_state_small(SmallArgs sa) {   // one workgroup per CU is all the grid needs: 256-VGPR budget
    using Cfg = Fused2Cfg<SP, 64, small_waves(SP)>;
    constexpr int TM = 64, NT = Cfg::NT, LPR = Cfg::LPR, IPL = Cfg::IPL, LDX = Cfg::LDX, LDW = Cfg::LDW, NPASS = Cfg::NPASS;
    const Fused2Args &a = sa.f;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *Xs = reinterpret_cast<float *>(smem);                         // [TM][LDX]  : [state | agg]
    float *Ws = Xs + TM * LDX;                                           // [2SP][LDW] : W1 rows (state ; agg)
    int *cont = reinterpret_cast<int *>(Ws + 2 * SP * LDW);              // one word: does the loop go on?
    float *W2s = Ws + 2 * SP * LDW + 64;                                 // L2: [SP][LDW] second Dense, then b2 [SP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int S = a.S;
    int ty = 0;
    while (ty + 1 < a.n_types && (int)blockIdx.x >= a.blk_begin[ty + 1]) ++ty;
    const FusedType tp = a.tp[ty];
    const int tile = blockIdx.x - a.blk_begin[ty];                       // one tile per workgroup, for the whole loop
    const int *__restrict__ rows = tp.rows;
    // rows [jbase, jend) of this type's node list; with groups: 64 nodes of one group, synchronising with that group only
    int jbase = tile * TM, count = tp.count, grp = 0;
    unsigned wg0 = 0, n_wg = gridDim.x;
    if (sa.groups.n > 0) {
        const GroupOfTile got = group_of_tile(sa.groups, blockIdx.x);
        jbase = got.node0 + ((int)blockIdx.x - got.tile0) * TM; count = got.node_end;
        grp = got.grp; wg0 = got.tile0; n_wg = got.tile1 - got.tile0;
    }

    for (int i = tid; i < 2 * SP * SP; i += NT) {
        const int k = i / SP, n = i % SP;
        const int kk = k < SP ? k : k - SP;
        float v = 0.0f;
        if (kk < S && n < tp.H) v = tp.Wf[(size_t)((k < SP ? tp.wrow_state : tp.wrow_agg) + kk) * tp.H + n];
        Ws[k * LDW + (Cfg::SWZ ? (n ^ ((k & 1) << 4)) : n)] = v;
    }
    if (L2) {                                   // second Dense of a two-layer state network: rows k < H, columns n < S
        for (int i = tid; i < SP * SP; i += NT) {
            const int k = i / SP, n = i % SP;
            W2s[k * LDW + (Cfg::SWZ ? (n ^ ((k & 1) << 4)) : n)] = (k < tp.H && n < S) ? tp.W2[(size_t)k * S + n] : 0.0f;
        }
        if (tid < SP) W2s[SP * LDW + tid] = tid < S ? tp.b2[tid] : 0.0f;
    }

    // ---- iteration-invariant per-lane state: the CSR rows of this lane group's nodes ---------------------------------
    const int q = tid / LPR, l4 = tid % LPR;
    int jn[NPASS], beg[NPASS], end[NPASS], ids[NPASS][IPL];
    float wts[NPASS][IPL], scl[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int m = jbase + p * Cfg::NPP + q;
        jn[p] = m < count ? (rows ? rows[m] : m) : -1;
        beg[p] = end[p] = 0; scl[p] = 1.0f;
        if (jn[p] >= 0) {
            beg[p] = a.rowptr[jn[p]]; end[p] = a.rowptr[jn[p] + 1];
            if (a.row_scale) scl[p] = a.row_scale[jn[p]];
        }
#pragma unroll
        for (int u = 0; u < IPL; ++u) {
            const int e = beg[p] + u * LPR + l4;
            ids[p][u] = e < end[p] ? a.src[e] : 0;
            wts[p][u] = (HAS_W && e < end[p]) ? a.w[e] : 0.0f;
        }
    }
    // ---- the per-node constant C in the accumulator layout (col = 16*ct + r, row = 16*rt + 4*g + reg) ----------------
    const int rt = wave % Cfg::RW, cw = wave / Cfg::RW;
    int jrow[4];
    f32x4 c0[Cfg::CT_PER_WAVE];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int m = jbase + 16 * rt + 4 * g + reg;
        jrow[reg] = m < count ? (rows ? rows[m] : m) : -1;
    }
#pragma unroll
    for (int ci = 0; ci < Cfg::CT_PER_WAVE; ++ci) {
        const int col = 16 * (cw * Cfg::CT_PER_WAVE + ci) + r;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            c0[ci][reg] = (jrow[reg] >= 0 && col < tp.H) ? a.C[(size_t)jrow[reg] * a.ldC + col] : 0.0f;
    }
    __syncthreads();

    int k_done = 0;
    unsigned moved_seen[2] = {0u, 0u};
    int timed_out = 0;
    // the predicate of state_0 (GNN.py:265 evaluates `condition` before the first iteration)
    bool run_first = sa.no_exit != 0;
    if (!run_first) {
        if (sa.pred0) {
            int v = 0;
            for (int i = lane; i < (int)n_wg; i += 64) v |= sa.pred0[wg0 + i];
            run_first = __any(v != 0);
        } else {
            run_first = __hip_atomic_load(&sa.flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        }
    }
    for (int it = 0; run_first && it < sa.max_iteration; ++it) {
        const __amdgpu_buffer_rsrc_t r_in = buf_rsrc(it == 0 ? a.state_in : sa.buf[it & 1]);
        const __amdgpu_buffer_rsrc_t r_out = buf_rsrc(sa.buf[(it + 1) & 1]);

        // ---- A. gather + aggregate: ids are already in registers, rows are the only round trip ----------------------
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            f32x4 own = {0.f, 0.f, 0.f, 0.f}, acc = {0.f, 0.f, 0.f, 0.f};
            own = buf_ld_sc1(r_in, jn[p] >= 0 ? (unsigned)(a.row_base + jn[p]) * (unsigned)(SP * 4) + 16u * l4 : BUF_OFF);
            int idc[IPL]; float wsc[IPL];
#pragma unroll
            for (int u = 0; u < IPL; ++u) { idc[u] = ids[p][u]; wsc[u] = wts[p][u]; }
            int rem = end[p] - beg[p], eb = beg[p];
#pragma unroll 1
            while (true) {
                f32x4 v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const unsigned off = (unsigned)__shfl(idc[i / LPR], i % LPR, LPR) * (unsigned)(SP * 4) + 16u * l4;
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
                for (int u = 0; u < IPL; ++u) {          // in-degree > 16: later ids are fetched in line, every iteration
                    const int e = eb + u * LPR + l4;
                    idc[u] = e < end[p] ? a.src[e] : 0;
                    wsc[u] = (HAS_W && e < end[p]) ? a.w[e] : 0.0f;
                }
            }
            acc *= scl[p];
            float *xr = Xs + (p * Cfg::NPP + q) * LDX + 4 * l4;
            *reinterpret_cast<float2 *>(xr) = make_float2(own[0], own[1]);
            *reinterpret_cast<float2 *>(xr + 2) = make_float2(own[2], own[3]);
            *reinterpret_cast<float2 *>(xr + SP) = make_float2(acc[0], acc[1]);
            *reinterpret_cast<float2 *>(xr + SP + 2) = make_float2(acc[2], acc[3]);
        }
        __syncthreads();

        // ---- B. [state | agg] . W1 + C on the f32 matrix cores ---------------------------------------------------------
        f32x4 c[Cfg::CT_PER_WAVE];
#pragma unroll
        for (int ci = 0; ci < Cfg::CT_PER_WAVE; ++ci) c[ci] = c0[ci];
        const float *xrow = Xs + (16 * rt + r) * LDX + g;
#pragma unroll 8
        for (int s4 = 0; s4 < 2 * SP / 4; ++s4) {
            const float av = xrow[4 * s4];
            const int k = 4 * s4 + g;
#pragma unroll
            for (int ci = 0; ci < Cfg::CT_PER_WAVE; ++ci) {
                const int n = 16 * (cw * Cfg::CT_PER_WAVE + ci) + r;
                const float bv = Ws[k * LDW + (Cfg::SWZ ? (n ^ ((k & 1) << 4)) : n)];
                c[ci] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c[ci], 0, 0, 0);
            }
        }

        if (L2) {
            // hidden layer: h = act1(c) into the agg half of the tile (dead after the loop above), then h . W2 + b2 with
            // the same fragment layout; a row's hidden units come from CW waves, hence the two workgroup barriers
            __syncthreads();
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = 16 * rt + 4 * g + reg;
#pragma unroll
                for (int ci = 0; ci < Cfg::CT_PER_WAVE; ++ci) {
                    const int col = 16 * (cw * Cfg::CT_PER_WAVE + ci) + r;
                    Xs[row * LDX + SP + col] = col < tp.H ? activate(tp.act, c[ci][reg]) : 0.0f;
                }
            }
            __syncthreads();
#pragma unroll
            for (int ci = 0; ci < Cfg::CT_PER_WAVE; ++ci) {
                const float b = W2s[SP * LDW + 16 * (cw * Cfg::CT_PER_WAVE + ci) + r];
                c[ci] = (f32x4){b, b, b, b};
            }
            const float *hrow = Xs + (16 * rt + r) * LDX + SP + g;
#pragma unroll 8
            for (int s4 = 0; s4 < SP / 4; ++s4) {
                const float av = hrow[4 * s4];
                const int k = 4 * s4 + g;
#pragma unroll
                for (int ci = 0; ci < Cfg::CT_PER_WAVE; ++ci) {
                    const int n = 16 * (cw * Cfg::CT_PER_WAVE + ci) + r;
                    const float bv = W2s[k * LDW + (Cfg::SWZ ? (n ^ ((k & 1) << 4)) : n)];
                    c[ci] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c[ci], 0, 0, 0);
                }
            }
        }
        const int act_out = L2 ? tp.act2 : tp.act;

        // ---- C. activation, predicate, new rows staged through LDS (same as k_state_fused2) ---------------------------
        int any = 0;
        float d2r[4], n2r[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = 16 * rt + 4 * g + reg;
            float d2 = 0.0f, n2 = 0.0f;
#pragma unroll
            for (int ci = 0; ci < Cfg::CT_PER_WAVE; ++ci) {
                const int col = 16 * (cw * Cfg::CT_PER_WAVE + ci) + r;
                const float nv = (jrow[reg] >= 0 && col < S) ? activate(act_out, c[ci][reg]) : 0.0f;
                const float ov = Xs[row * LDX + col];
                const float d = nv - ov;
                d2 = fmaf(d, d, d2);
                n2 = fmaf(ov, ov, n2);
                c[ci][reg] = nv;
            }
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) {
                d2 += __shfl_xor(d2, off, 16);
                n2 += __shfl_xor(n2, off, 16);
            }
            d2r[reg] = d2; n2r[reg] = n2;
        }
        if (Cfg::CW > 1) {
            __syncthreads();
            if (r == 0) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int row = 16 * rt + 4 * g + reg;
                    Xs[row * LDX + SP + 2 * cw] = d2r[reg];
                    Xs[row * LDX + SP + 2 * cw + 1] = n2r[reg];
                }
            }
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = 16 * rt + 4 * g + reg;
#pragma unroll
            for (int ci = 0; ci < Cfg::CT_PER_WAVE; ++ci)
                Xs[row * LDX + 16 * (cw * Cfg::CT_PER_WAVE + ci) + r] = c[ci][reg];
        }
        __syncthreads();
        if (Cfg::CW > 1) {
            if (cw == 0 && r == 0) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int row = 16 * rt + 4 * g + reg;
                    float d2 = 0.0f, n2 = 0.0f;
                    for (int w2 = 0; w2 < Cfg::CW; ++w2) { d2 += Xs[row * LDX + SP + 2 * w2]; n2 += Xs[row * LDX + SP + 2 * w2 + 1]; }
                    if (jrow[reg] >= 0 && sqrtf(d2) > a.thr * sqrtf(n2)) any = 1;
                }
            }
        } else {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                if (jrow[reg] >= 0 && sqrtf(d2r[reg]) > a.thr * sqrtf(n2r[reg])) any = 1;
        }
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            if (jn[p] >= 0) {
                const float *xr = Xs + (p * Cfg::NPP + q) * LDX + 4 * l4;
                const float2 lo = *reinterpret_cast<const float2 *>(xr), hi = *reinterpret_cast<const float2 *>(xr + 2);
                buf_st_sc1(r_out, (unsigned)(a.row_base + jn[p]) * (unsigned)(SP * 4) + 16u * l4, (f32x4){lo.x, lo.y, hi.x, hi.y});
            }
        }

        // ---- grid barrier: publish the rows and the predicate, wait for every workgroup --------------------------------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        any = __syncthreads_or(any);
        if (tid == 0) {
            // The arrival and the predicate travel in ONE 64-bit add.  Two counters alternate: a workgroup can reach the
            // barrier of iteration it+1 before a slow one has read iteration it's total, but never the one of it+2.
            unsigned long long *ctr = sa.bar + 2 * grp + (it & 1);
            __hip_atomic_fetch_add(ctr, 1ull + ((unsigned long long)(any ? 1u : 0u) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(it / 2 + 1) * n_wg;
            unsigned long long v = 0;
            if (!wait_until(sa.wait_ticks, [&]() { v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return (unsigned)v >= target; }))
                timed_out = 1;                         // some workgroup never arrived (not resident?): reported through k
            const unsigned moved = (unsigned)(v >> 32);
            *cont = timed_out ? -1 : ((moved != moved_seen[it & 1]) ? 1 : 0);
            moved_seen[it & 1] = moved;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    // no instruction: keeps the loads below the poll
        }
        __syncthreads();
        k_done = it + 1;
        if (*cont < 0) break;                          // an expired wait ends the loop at once (k < 0): not the bound again in every remaining iteration
        if (!sa.no_exit && *cont == 0) break;          // uniform: every workgroup read the same total
    }
    // The rows the loop ends on, straight into the caller's compact buffer: the tile still holds what the last iteration
    // stored (k_done > 0) or nothing ran and state_0 is the answer.  Every workgroup is past the last grid barrier, so no
    // one reads a state buffer any more, whichever of them the caller's buffer stands in for.
    if (sa.state_final) {
        const __amdgpu_buffer_rsrc_t r_first = buf_rsrc(a.state_in);
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            if (jn[p] < 0) continue;
            f32x4 v;
            if (k_done > 0) {
                const float *xr = Xs + (p * Cfg::NPP + q) * LDX + 4 * l4;
                const float2 lo = *reinterpret_cast<const float2 *>(xr), hi = *reinterpret_cast<const float2 *>(xr + 2);
                v = (f32x4){lo.x, lo.y, hi.x, hi.y};
            } else {
                v = buf_ld_sc1(r_first, (unsigned)(a.row_base + jn[p]) * (unsigned)(SP * 4) + 16u * l4);
            }
            float *dst = sa.state_final + (size_t)jn[p] * sa.ld_final + 4 * l4;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (4 * l4 + i < S) dst[i] = v[i];
        }
    }
    // k_out is zero before the launch: workgroup 0 adds k; a workgroup whose grid barrier timed out (the launch was not
    // fully resident: results are not valid) adds -1e9, so k < 0 reports it whatever the order of the two
    if (tid == 0 && a.k_out) {
        if (timed_out) atomicAdd(a.k_out + grp, -1.0e9f);
        if (blockIdx.x == wg0) atomicAdd(a.k_out + grp, (float)k_done);
    }
}

// one tile per workgroup, every workgroup resident: graphs of at most 64 * n_cu nodes
constexpr size_t SMALL_LDS = 96 * 1024;      // > half of a CU's 160 KB: at most one of these workgroups per CU
template <int SP, bool HAS_W, bool L2>
int launch_small_one(SmallArgs &sa, int n_cu, hipStream_t st, int group_tiles) {
    using Cfg = Fused2Cfg<SP, 64, small_waves(SP)>;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void *)k_state_small<SP, HAS_W, L2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)SMALL_LDS) != hipSuccess) return 1;
        attr = true;
    }
    Fused2Args &fa = sa.f;
    fa.blk_begin[0] = 0;
    for (int t = 0; t < fa.n_types; ++t) fa.blk_begin[t + 1] = fa.blk_begin[t] + (fa.tp[t].count + 63) / 64;
    const int grid = group_tiles > 0 ? group_tiles : fa.blk_begin[fa.n_types];   // with groups no tile straddles a group
    if (group_tiles > 0) fa.blk_begin[1] = group_tiles;
    if (grid == 0 || grid > n_cu) return 2;             // not applicable: the caller falls back to one launch per iteration
    if (!persistent_fits((const void *)k_state_small<SP, HAS_W, L2>, Cfg::NT, SMALL_LDS, grid, n_cu)) return 2;     // (its grid barrier needs every workgroup resident)
    sa.wait_ticks = wait_ticks();
    GNN_SET_KERNEL_NAME("k_state_small<%d,%s,%s>", SP, HAS_W ? "true" : "false", L2 ? "true" : "false");
    k_state_small<SP, HAS_W, L2><<<grid, Cfg::NT, SMALL_LDS, st>>>(sa);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

inline int small_tiles(const Fused2Args &fa) {
    int n = 0;
    for (int t = 0; t < fa.n_types; ++t) n += (fa.tp[t].count + 63) / 64;
    return n;
}

inline int launch_small(SmallArgs &sa, int SP, int n_cu, hipStream_t st, int group_tiles = 0) {
    const bool l2 = sa.f.n_types > 0 && sa.f.tp[0].W2 != nullptr;
#define SMALL_CASE(SPV)                                                                                              \
    case SPV:                                                                                                        \
        if (l2) return sa.f.w ? launch_small_one<SPV, true, true>(sa, n_cu, st, group_tiles) : launch_small_one<SPV, false, true>(sa, n_cu, st, group_tiles); \
        return sa.f.w ? launch_small_one<SPV, true, false>(sa, n_cu, st, group_tiles) : launch_small_one<SPV, false, false>(sa, n_cu, st, group_tiles);
    switch (SP) {
        SMALL_CASE(16)                                     // the starter configuration: state = the 14 label columns (state_vect_dim = 0)
        SMALL_CASE(32)
        SMALL_CASE(64)
        default: return 2;
    }
#undef SMALL_CASE
}

}  // namespace gnn
