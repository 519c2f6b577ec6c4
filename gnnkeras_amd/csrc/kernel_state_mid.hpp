// Mid-size graphs (more nodes than one 64-node tile per CU, few enough that a kernel boundary per iteration is a large
// share of the iteration): the WHOLE convergence loop (GNN/Models/GNN.py:265, :196-236) in one launch, several tiles per
// workgroup.  The iteration body is the phase-alternating kernel's (kernel_state_fused2.hpp: gather -> MFMA -> epilogue per
// 64-node tile, the next tiles' CSR rows prefetched while this tile's neighbour rows are in flight, two workgroups per CU
// so one gathers while the other multiplies); what changes is what surrounds it:
//   * W1 is filled into LDS once per LOOP, not once per iteration; no dispatch, gate read or ramp-up per iteration;
//   * state rows move with `sc1` loads / stores (through to memory, kernel_state_small.hpp has the argument), so the rows
//     another XCD's workgroup wrote in the last iteration are what this one reads; C and the CSR are ordinary cached loads;
//   * iterations are separated by a grid barrier.  With up to 2 workgroups per CU (512 of them) one shared arrival word
//     would be ~512 serialised memory-side atomics per iteration and 512 pollers of one line, so arrivals are two-level:
//     workgroup b adds to group counter b & 7, the last arrival of a group adds to the global counter, the last arrival of
//     all publishes `2 * (it + 1) + (some node still moves)` in eight release words, one per group, which is what the
//     waiting workgroups poll (64 pollers per line).  Counters are never reset (two sets alternate, targets grow with the
//     iteration; the predicate rides in the high halves, and the cumulative value each last arrival saw is parked in a
//     `seen` word for the last arrival two iterations later), so the chain is three dependent round trips.  Measured
//     alternatives (profiles/r02_mid_sweep.txt): resetting the counters puts two store round trips into the chain (16.7 us
//     per iteration on a 2 000-node graph); one level of eight counters polled by every workgroup is the shortest chain
//     on 32 workgroups (15.2 us) and the slowest on 512 (32 us at 30 000 nodes against 23.5: 4 096 line reads per poll).
//     Every workgroup must be resident: the launcher sizes the grid to 2 per CU (LDS and VGPR budgets allow exactly
//     that); polls are bounded and an expired one turns k negative.
#pragma once
#include <hip/hip_runtime.h>
#include "kernel_state_small.hpp"

namespace gnn {

constexpr int MID_LINE = 32;                       // words per 128-byte line: every barrier word sits on its own line
// lines: [2][8] group counters (+ their `seen` word) | [2] global counter (+ seen) | [8] release words
constexpr int MID_BAR_WORDS = (16 + 2 + 8) * MID_LINE;

template <int SP, bool HAS_W, int NW, int DEPTH>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 4 : 2) k_state_mid(SmallArgs sa) {
    constexpr int TM = 64;
    using Cfg = Fused2Cfg<SP, TM, NW>;
    constexpr int NT = Cfg::NT;
    constexpr int LPR = Cfg::LPR, IPL = Cfg::IPL, LDX = Cfg::LDX, LDW = Cfg::LDW;
    const Fused2Args &a = sa.f;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *Xs = reinterpret_cast<float *>(smem);                         // [TM][LDX]  : [state | agg]
    float *Ws = Xs + TM * LDX;                                           // [2SP][LDW] : W1 rows (state ; agg)
    int *jid = reinterpret_cast<int *>(Ws + 2 * SP * LDW);               // [TM] local node id per tile row, -1 = pad
    int *cont = jid + TM;                                                // one word: does the loop go on?

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int S = a.S;
    int ty = 0;
    while (ty + 1 < a.n_types && (int)blockIdx.x >= a.blk_begin[ty + 1]) ++ty;
    const FusedType tp = a.tp[ty];
    const int bid = blockIdx.x - a.blk_begin[ty], nblk = a.blk_begin[ty + 1] - a.blk_begin[ty];
    const int count = tp.count;
    const int *__restrict__ rows = tp.rows;

    for (int i = tid; i < 2 * SP * SP; i += NT) {
        const int k = i / SP, n = i % SP;
        const int kk = k < SP ? k : k - SP;
        float v = 0.0f;
        if (kk < S && n < S) v = tp.Wf[(size_t)((k < SP ? tp.wrow_state : tp.wrow_agg) + kk) * tp.H + n];
        Ws[k * LDW + (Cfg::SWZ ? (n ^ ((k & 1) << 4)) : n)] = v;
    }

    // XCD-contiguous tile ranges (workgroups b, b+8, .. share an XCD under round-robin dispatch; speed only)
    const int ntiles = (count + TM - 1) / TM;
    const int xcd = bid & 7, lb = bid >> 3;
    const int blk_per_xcd = (nblk + 7 - xcd) >> 3;
    const int tpx = (ntiles + 7) >> 3;
    const int t_end = min(ntiles, (xcd + 1) * tpx);
    const int t_first = xcd * tpx + lb;

    const char *__restrict__ cbase = reinterpret_cast<const char *>(a.C);
    const int q = tid / LPR;          // node slot inside a pass
    const int l4 = tid % LPR;         // 16-B column chunk of the row owned by this lane

    auto slot_m = [&](int n) -> int {               // global row number m of slot n for this lane group, or -1
        const int tile = t_first + (n / Cfg::NPASS) * blk_per_xcd;
        if (tile >= t_end) return -1;
        const int m = tile * TM + (n % Cfg::NPASS) * Cfg::NPP + q;
        return m < count ? m : -1;
    };
    auto node_of = [&](int m) -> int { return m < 0 ? -1 : (rows ? rows[m] : m); };

    // barrier words (zero before the launch)
    const unsigned grp = blockIdx.x & 7;
    const unsigned n_grp = (gridDim.x + 7 - grp) >> 3;                       // workgroups that share this group counter
    const unsigned n_groups = min(gridDim.x, 8u);
    unsigned *bar_w = reinterpret_cast<unsigned *>(sa.bar);
    unsigned *release = bar_w + (size_t)18 * MID_LINE;

    int k_done = 0, timed_out = 0;
    const bool run_first = sa.no_exit || __hip_atomic_load(&sa.flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    for (int it = 0; run_first && it < sa.max_iteration; ++it) {
        const __amdgpu_buffer_rsrc_t r_in = buf_rsrc(it == 0 ? a.state_in : sa.buf[it & 1]);
        const __amdgpu_buffer_rsrc_t r_out = buf_rsrc(sa.buf[(it + 1) & 1]);

        // ---- slot pipeline: slot index n counts (tile, pass) pairs of this workgroup ---------------------------------
        int n_slot = 0;
        int j0, beg0 = 0, end0 = 0, j1, beg1 = 0, end1 = 0, j2, j3;
        int ids0[IPL]; float ws0[IPL];
        j0 = node_of(slot_m(0)); j1 = node_of(slot_m(1)); j2 = node_of(slot_m(2)); j3 = node_of(slot_m(3));
        if (j0 >= 0) { beg0 = a.rowptr[j0]; end0 = a.rowptr[j0 + 1]; }
        if (j1 >= 0) { beg1 = a.rowptr[j1]; end1 = a.rowptr[j1 + 1]; }
#pragma unroll
        for (int u = 0; u < IPL; ++u) {
            const int e = beg0 + u * LPR + l4;
            ids0[u] = e < end0 ? a.src[e] : 0;
            ws0[u] = (HAS_W && e < end0) ? a.w[e] : 0.0f;
        }

        int any = 0;
        for (int tile = t_first; tile < t_end; tile += blk_per_xcd) {
            __syncthreads();   // previous tile's Xs fully consumed (and, first time, the W fill is visible)

            // accumulators start from the per-node constant C: D = [state|agg].W1 + C
            f32x4 c[Cfg::CT_PER_WAVE];
            int jrow[4];
            const int rt = wave % Cfg::RW, cw = wave / Cfg::RW;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int m = tile * TM + 16 * rt + 4 * g + reg;
                jrow[reg] = m < count ? (rows ? rows[m] : m) : -1;
            }
#pragma unroll
            for (int ci = 0; ci < Cfg::CT_PER_WAVE; ++ci) {
                const int col = 16 * (cw * Cfg::CT_PER_WAVE + ci) + r;
                const int colc = min(col, S - 1);
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {          // always-valid address, value masked afterwards: no branches
                    const unsigned coff = ((unsigned)max(jrow[reg], 0) * (unsigned)a.ldC + (unsigned)colc) * 4u;
                    const float cv = *reinterpret_cast<const float *>(cbase + coff);   // C spans < 4 GiB (launcher check)
                    c[ci][reg] = (jrow[reg] >= 0 && col < S) ? cv : 0.0f;
                }
            }

            // ---- A. gather + aggregate, one node slot per lane group and pass ----------------------------------------
#pragma unroll 1
            for (int pass = 0; pass < Cfg::NPASS; ++pass) {
                int ids1[IPL]; float ws1[IPL];
#pragma unroll
                for (int u = 0; u < IPL; ++u) {
                    const int e = beg1 + u * LPR + l4;
                    ids1[u] = e < end1 ? a.src[e] : 0;
                    ws1[u] = (HAS_W && e < end1) ? a.w[e] : 0.0f;
                }
                int beg2 = 0, end2 = 0;
                if (j2 >= 0) { beg2 = a.rowptr[j2]; end2 = a.rowptr[j2 + 1]; }
                const int j4 = node_of(slot_m(n_slot + 4));

                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const int deg = end0 - beg0;
                const f32x4 own = buf_ld_sc1(r_in, j0 >= 0 ? (unsigned)(a.row_base + j0) * (unsigned)(SP * 4) + 16u * l4 : BUF_OFF);
                {
                    int idc[IPL]; float wsc[IPL];
#pragma unroll
                    for (int u = 0; u < IPL; ++u) { idc[u] = ids0[u]; wsc[u] = ws0[u]; }
                    int rem = deg;
                    int eb = beg0;
#pragma unroll 1
                    while (true) {
#pragma unroll
                        for (int s0 = 0; s0 < 16; s0 += DEPTH) {          // DEPTH rows in flight per lane group
                            if (s0 > 0 && !__any(s0 < rem)) break;
                            f32x4 v[DEPTH];
#pragma unroll
                            for (int i = 0; i < DEPTH; ++i) {
                                const unsigned off = (unsigned)__shfl(idc[(s0 + i) / LPR], (s0 + i) % LPR, LPR) * (unsigned)(SP * 4) + 16u * l4;
                                v[i] = buf_ld_sc1(r_in, s0 + i < rem ? off : BUF_OFF);
                            }
#pragma unroll
                            for (int i = 0; i < DEPTH; ++i) {
                                if (HAS_W) acc += __shfl(wsc[(s0 + i) / LPR], (s0 + i) % LPR, LPR) * v[i];
                                else acc += v[i];
                            }
                        }
                        rem -= 16; eb += 16;
                        if (!__any(rem > 0)) break;
#pragma unroll
                        for (int u = 0; u < IPL; ++u) {
                            const int e = eb + u * LPR + l4;
                            idc[u] = e < end0 ? a.src[e] : 0;
                            wsc[u] = (HAS_W && e < end0) ? a.w[e] : 0.0f;
                        }
                    }
                }
                if (a.row_scale && j0 >= 0) acc *= a.row_scale[j0];

                const int nl = pass * Cfg::NPP + q;
                float *xr = Xs + nl * LDX + 4 * l4;                     // rows are 8-B aligned: two b64 stores each
                *reinterpret_cast<float2 *>(xr) = make_float2(own[0], own[1]);
                *reinterpret_cast<float2 *>(xr + 2) = make_float2(own[2], own[3]);
                *reinterpret_cast<float2 *>(xr + SP) = make_float2(acc[0], acc[1]);
                *reinterpret_cast<float2 *>(xr + SP + 2) = make_float2(acc[2], acc[3]);
                if (l4 == 0) jid[nl] = j0;

                j0 = j1; beg0 = beg1; end0 = end1;
#pragma unroll
                for (int u = 0; u < IPL; ++u) { ids0[u] = ids1[u]; ws0[u] = ws1[u]; }
                j1 = j2; beg1 = beg2; end1 = end2;
                j2 = j3; j3 = j4;
                ++n_slot;
            }
            __syncthreads();

            // ---- B. [state | agg] . W1 on the f32 matrix cores --------------------------------------------------------
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

            // ---- C. activation, predicate, stage new rows (as k_state_fused2) -----------------------------------------
            float d2r[4], n2r[4];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = 16 * rt + 4 * g + reg;
                float d2 = 0.0f, n2 = 0.0f;
#pragma unroll
                for (int ci = 0; ci < Cfg::CT_PER_WAVE; ++ci) {
                    const int col = 16 * (cw * Cfg::CT_PER_WAVE + ci) + r;
                    const float nv = (jrow[reg] >= 0 && col < S) ? activate(tp.act, c[ci][reg]) : 0.0f;
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
#pragma unroll 1
            for (int pass = 0; pass < Cfg::NPASS; ++pass) {
                const int nl = pass * Cfg::NPP + q;
                const int j = jid[nl];
                if (j >= 0) {
                    const float *xr = Xs + nl * LDX + 4 * l4;
                    const float2 lo = *reinterpret_cast<const float2 *>(xr), hi = *reinterpret_cast<const float2 *>(xr + 2);
                    buf_st_sc1(r_out, (unsigned)(a.row_base + j) * (unsigned)(SP * 4) + 16u * l4, (f32x4){lo.x, lo.y, hi.x, hi.y});
                }
            }
        }

        // ---- grid barrier: publish the rows and the predicate, wait for every workgroup ------------------------------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        any = __syncthreads_or(any);
        if (tid == 0) {
            const int par = it & 1;
            const unsigned turn = (unsigned)(it / 2 + 1), gen = (unsigned)(it + 1);
            unsigned *gline = bar_w + (size_t)(par * 8 + grp) * MID_LINE, *aline = bar_w + (size_t)(16 + par) * MID_LINE;
            // cumulative "moved" counts the last arrivals of this parity's previous turn saw (zero the first time)
            const unsigned g_seen = __hip_atomic_load(gline + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned a_seen = __hip_atomic_load(aline + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long mine = 1ull + ((unsigned long long)(any ? 1u : 0u) << 32);
            unsigned long long v = __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(gline), mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + mine;
            bool released = false;
            if ((unsigned)v == turn * n_grp) {                    // last arrival of this group
                const unsigned ghi = (unsigned)(v >> 32);
                __hip_atomic_store(gline + 2, ghi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // read two iterations from now
                const unsigned long long gm = 1ull + ((unsigned long long)(ghi != g_seen ? 1u : 0u) << 32);
                v = __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(aline), gm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + gm;
                if ((unsigned)v == turn * n_groups) {             // last arrival of all: release every group
                    const unsigned ahi = (unsigned)(v >> 32);
                    __hip_atomic_store(aline + 2, ahi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned word = 2u * gen + (ahi != a_seen ? 1u : 0u);
                    for (unsigned gg = 0; gg < n_groups; ++gg)
                        __hip_atomic_store(release + (size_t)gg * MID_LINE, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *cont = (int)(word & 1u);
                    released = true;
                }
            }
            if (!released) {
                const unsigned *rel = release + (size_t)grp * MID_LINE;
                unsigned w = 0;
                const bool arrived = wait_until(sa.wait_ticks, [&]() { w = __hip_atomic_load(rel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return (w >> 1) == gen; });
                *cont = (int)(w & 1u);
                if (!arrived) { timed_out = 1; *cont = -1; }            // some workgroup never arrived (not resident?): reported through k
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    // no instruction: keeps the loads below the poll
        }
        __syncthreads();
        k_done = it + 1;
        if (*cont < 0 || (!sa.no_exit && *cont == 0)) break;   // uniform: every workgroup read the same word (or gave up)
    }
    // k_out is zero before the launch: workgroup 0 adds k; a workgroup whose wait expired adds -1e9 (k < 0: not valid)
    if (tid == 0 && a.k_out) {
        if (timed_out) atomicAdd(a.k_out, -1.0e9f);
        if (blockIdx.x == 0) atomicAdd(a.k_out, (float)k_done);
    }
}

template <int SP, bool HAS_W, int NW, int DEPTH>
int launch_mid_one(SmallArgs &sa, int n_cu, hipStream_t st) {
    using Cfg = Fused2Cfg<SP, 64, NW>;
    constexpr size_t LDS = Cfg::LDS_BYTES + 64;
    static_assert(2 * LDS <= 160 * 1024, "two workgroups per CU must fit");
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void *)k_state_mid<SP, HAS_W, NW, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS) != hipSuccess) return 1;
        attr = true;
    }
    Fused2Args &fa = sa.f;
    const int budget = 2 * n_cu;                 // 2 workgroups per CU, every one resident for the whole loop
    long total_tiles = 0;
    for (int t = 0; t < fa.n_types; ++t) total_tiles += (fa.tp[t].count + 63) / 64;
    fa.blk_begin[0] = 0;
    for (int t = 0; t < fa.n_types; ++t) {
        const int ntiles = (fa.tp[t].count + 63) / 64;
        int nb = 0;
        if (ntiles > 0) {
            nb = (int)std::min<long>((ntiles + 7) / 8 * 8, std::max<long>(8, budget * (long)ntiles / std::max<long>(total_tiles, 1)));
            nb = std::max(8, nb / 8 * 8);
        }
        fa.blk_begin[t + 1] = fa.blk_begin[t] + nb;
    }
    const int grid = fa.blk_begin[fa.n_types];
    if (grid == 0 || grid > budget) return 2;
    if (!persistent_fits((const void *)k_state_mid<SP, HAS_W, NW, DEPTH>, Cfg::NT, LDS, grid, n_cu)) return 2;      // (its grid barrier needs every workgroup resident)
    sa.wait_ticks = wait_ticks();
    GNN_SET_KERNEL_NAME("k_state_mid<%d,%s,%d,%d>", SP, HAS_W ? "true" : "false", NW, DEPTH);
    k_state_mid<SP, HAS_W, NW, DEPTH><<<grid, Cfg::NT, LDS, st>>>(sa);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// one-layer state networks; weighted graphs run the 4-wave shape (register budget, as launch_fused2_w)
inline int launch_mid(SmallArgs &sa, int SP, int n_cu, hipStream_t st) {
    if (sa.f.n_types > 0 && sa.f.tp[0].W2 != nullptr) return 2;
    switch (SP) {
        case 32: return sa.f.w ? launch_mid_one<32, true, 4, 16>(sa, n_cu, st) : launch_mid_one<32, false, 8, 8>(sa, n_cu, st);
        case 64: return sa.f.w ? launch_mid_one<64, true, 4, 16>(sa, n_cu, st) : launch_mid_one<64, false, 8, 8>(sa, n_cu, st);
        default: return 2;
    }
}

}  // namespace gnn
