// Fused state-transition iteration for state widths 65 .. 128 (padded leading dimension SP = 128).
// Same contract and arguments as k_state_fused4 (one launch = one iteration of the reference's `convergence` + the
// `condition` of the next one, GNN/Models/GNN.py:217-236, :196-214; any `state_vect_dim` is legal there, :26-28).
//
// What changes at this width (numbers for C4-sized graphs, d = 128):
//   * W1 = [2 SP x SP] floats = 128 KB: it fits the CU's 160 KB of LDS exactly once, so ONE 1024-thread workgroup per CU
//     (16 waves, 128 VGPRs each) instead of two, and 32 KB are left for everything else;
//   * the dense part is no longer negligible: 2 * 256 * 128 FLOP per node = 65.5 GFLOP per iteration at 10^6 nodes = 417 us of
//     the f32 matrix pipes at their peak (v_mfma_f32_16x16x4_f32, 32 cycles per instruction per SIMD), against ~840 us of HBM
//     time for the 6.7 GB the iteration moves: the matrix waves must keep all four pipes at least half busy WHILE the gather
//     waves keep the memory system busy.
// Structure (wave-specialised like generation 4, no barrier after the W1 fill):
//   * 12 gather waves: 32 lanes own a destination node (16 B of its 512-B row each), two nodes per wave; the next 32 source
//     ids arrive in one coalesced load and are broadcast with ds_bpermute; 8 rows in flight per lane group; only the
//     NEIGHBOUR SUM goes into an LDS ring slot (16 rows x 528 B, 3 slots);
//   * 4 matrix waves (one per SIMD), each takes every 4th 16-node tile:
//       - own state rows straight from global memory in MFMA A-fragment order: the k index is permuted (lane (row r, group g)
//         holds columns 16q + 4g .. + 3 of its row, one 16-B load per q) and W1's rows are read in the same order, so no LDS
//         round trip and no transposition; the [state . W1_s] half (256 MFMAs, ~3.4 us) runs BEFORE the wave waits for its
//         ring slot, i.e. while the gather waves are still filling it;
//       - the slot's 16 rows are copied into the same 32 registers (8 ds_read_b128) and the slot is handed back at once: it is
//         busy for a fill plus one read, not for the 3.4 us of [agg . W1_a] that follow;
//       - output columns are permuted too: MFMA column r of tile ci is column 8r + ci, so a lane's 8 B-operands of one
//         k-step are 32 contiguous bytes of a W1 row (two ds_read_b128 per 8 MFMAs, XOR-swizzled by g: conflict-free), the
//         constant term C arrives as two 16-B loads per row, and the epilogue (activation, predicate, store) runs on whole
//         32-B pieces of rows straight from the accumulators.
// Exact float32 throughout (f32 MFMA = fma chain).  Bounded waits raise the sticky error word like generation 4.
#pragma once
#include <hip/hip_runtime.h>
#include "kernel_state_fused4.hpp"

namespace gnn {

struct WideCfg {
    static constexpr int SP = 128, NW = 16, NT = 64 * NW;
    static constexpr int NCONS = 4, NPROD = NW - NCONS;
    static constexpr int LPR = SP / 4;               // 32 lanes per node row
    static constexpr int RPWV = 64 / LPR;            // 2 rows per gather wave and job
    static constexpr int PPT = 16 / RPWV;            // 8 gather jobs per 16-row tile
    static constexpr int CH = 32;                    // source ids fetched per coalesced load (one per lane of the group)
    static constexpr int DEPTH = 8;                  // neighbour rows in flight per lane group
    static constexpr int LDA = SP + 4;               // slot row stride (floats): 528 B, staggers the banks of consecutive rows
    static constexpr int NS = 3;                     // ring slots
    static constexpr int SLOT = 16 * LDA;
    static constexpr size_t LDS_BYTES = sizeof(float) * ((size_t)2 * SP * SP + (size_t)NS * SLOT) + sizeof(int) * 2 * NS;
};

// W1 element (k', n) lives at k' * SP + (n ^ swz(k')): lanes of group g read B operands of row k' = .. + 4g + e, so an XOR of 4
// dwords on odd g keeps the two 16-B halves of neighbouring lane groups on different banks
__device__ __forceinline__ int wide_wcol(int kp, int n) { return n ^ (((kp >> 2) & 1) << 2); }

template <bool HAS_W, bool INIT = false, int DEPTH = WideCfg::DEPTH>
__global__ void __launch_bounds__(1024, 4) k_state_wide(Fused2Args a) {
    int open = a.gate == nullptr;
    for (int i = 0; i < a.n_gate; ++i) open |= a.gate[(size_t)i * a.gate_stride] != 0;
    using Cfg = WideCfg;
    constexpr int SP = Cfg::SP, NT = Cfg::NT, LPR = Cfg::LPR, LDA = Cfg::LDA, NS = Cfg::NS, CH = Cfg::CH;
    constexpr int SPIN_MAX = GNN_F4_SPIN_MAX;
    int bad = 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *Ws = reinterpret_cast<float *>(smem);                         // [2 SP][SP]   W1 rows (state ; agg), swizzled columns
    float *Xs = Ws + 2 * SP * SP;                                        // [NS][16][LDA] neighbour sums
    int *fill = reinterpret_cast<int *>(Xs + NS * Cfg::SLOT);            // [NS] gather-wave deposits so far
    int *freed = fill + NS;                                              // [NS] tiles consumed so far

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int S = a.S;
    // Widths below 128 keep the 128-float row stride (pad columns stay zero) but not the work: chunks behind column S are never
    // loaded or stored, k-steps behind it never multiplied (d = 96: three quarters of the bytes and of the MFMAs).
    const int S4 = (S + 3) >> 2;               // 16-byte chunks of a row that hold state columns
    const int SQ = (S + 15) >> 4;              // 16-column k blocks that hold state columns
    int ty = 0;
    while (ty + 1 < a.n_types && (int)blockIdx.x >= a.blk_begin[ty + 1]) ++ty;
    const FusedType tp = a.tp[ty];
    const int bid = blockIdx.x - a.blk_begin[ty], nblk = a.blk_begin[ty + 1] - a.blk_begin[ty];
    const int count = tp.count;
    const int *__restrict__ rows = tp.rows;

    // XCD-contiguous tile ranges (workgroups b, b+8, .. share an XCD under round-robin dispatch; speed only)
    const int ntiles = (count + 15) / 16;
    const int xcd = bid & 7, lb = bid >> 3;
    const int blk_per_xcd = (nblk + 7 - xcd) >> 3;
    const int tpx = (ntiles + 7) >> 3;
    const int t_end = min(ntiles, (xcd + 1) * tpx);
    const int t_first = xcd * tpx + lb;
    const int T = t_first < t_end ? (t_end - t_first + blk_per_xcd - 1) / blk_per_xcd : 0;   // tiles of this workgroup

    const __amdgpu_buffer_rsrc_t r_C = buf_rsrc(a.C), r_rows = buf_rsrc(tp.rows), r_state = buf_rsrc(a.state_in),
                                 r_rowptr = buf_rsrc(a.rowptr), r_src = buf_rsrc(a.src), r_w = buf_rsrc(HAS_W ? a.w : nullptr),
                                 r_scale = buf_rsrc(a.row_scale), r_init = buf_rsrc(INIT ? a.agg_init : nullptr);
    const bool has_scale = a.row_scale != nullptr;
    char *__restrict__ obase = reinterpret_cast<char *>(a.state_out);
    int any = 0;

    // ---- gather waves: the first job's CSR row (node id, row pointers, first CH source ids) before anything is waited for -----
    const int p = wave - Cfg::NCONS;
    const int qr = lane / LPR;                 // row of this lane group inside the wave's deposit (0 / 1)
    const int l4 = lane % LPR;                 // 16-B column chunk of the row owned by this lane
    const int njobs = T * Cfg::PPT;
    auto job_m = [&](int n) -> int {
        const int m = (t_first + (n / Cfg::PPT) * blk_per_xcd) * 16 + (n % Cfg::PPT) * Cfg::RPWV + qr;
        return (n < njobs && m < count) ? m : -1;
    };
    auto node_of = [&](int m) -> int {
        const int jr = buf_ld_i32(r_rows, m >= 0 ? 4u * (unsigned)m : BUF_OFF);
        return m >= 0 ? (rows ? jr : m) : -1;
    };
    int jA = -1, jB = -1, begA = 0, endA = 0, idA = 0;
    float wA = 0.0f;
    if (wave >= Cfg::NCONS) {
        jA = node_of(job_m(p)); jB = node_of(job_m(p + Cfg::NPROD));
        begA = buf_ld_i32(r_rowptr, jA >= 0 ? 4u * (unsigned)jA : BUF_OFF);
        endA = buf_ld_i32(r_rowptr, jA >= 0 ? 4u * (unsigned)jA + 4u : BUF_OFF);
        const int e = begA + l4;
        idA = buf_ld_i32(r_src, e < endA ? 4u * (unsigned)e : BUF_OFF);
        wA = HAS_W ? buf_ld_f32(r_w, e < endA ? 4u * (unsigned)e : BUF_OFF) : 0.0f;
    }

    // ---- W1 fill: rows k' < SP multiply the own state, rows SP + k' the neighbour sum; pad rows / columns are zero ---------
    if ((tp.H & 3) == 0 && (reinterpret_cast<uintptr_t>(tp.Wf) & 15) == 0) {
        // whole 16-byte pieces of weight rows, eight loads per thread all in flight (the scalar loop below makes 32 dependent trips:
        // ~16 us of every launch); the column swizzle flips bit 2: 4-column pieces stay whole
        constexpr int NV = 2 * SP * SP / 4 / NT;
        f32x4 v[NV];
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int i4 = tid + u * NT, k = i4 / (SP / 4), n = (i4 % (SP / 4)) * 4;
            const int kk = k < SP ? k : k - SP;
            const bool ok = kk < S && n < tp.H;
            const float *src = tp.Wf + (size_t)((k < SP ? tp.wrow_state : tp.wrow_agg) + (ok ? kk : 0)) * tp.H + (ok ? n : 0);
            const f32x4 t = *reinterpret_cast<const f32x4 *>(src);
            v[u] = ok ? t : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int i4 = tid + u * NT, k = i4 / (SP / 4), n = (i4 % (SP / 4)) * 4;
            *reinterpret_cast<f32x4 *>(Ws + k * SP + wide_wcol(k, n)) = v[u];
        }
    } else
    for (int i = tid; i < 2 * SP * SP; i += NT) {
        const int k = i / SP, n = i % SP;
        const int kk = k < SP ? k : k - SP;
        float v = 0.0f;
        if (kk < S && n < tp.H) v = tp.Wf[(size_t)((k < SP ? tp.wrow_state : tp.wrow_agg) + kk) * tp.H + n];
        Ws[k * SP + wide_wcol(k, n)] = v;
    }
    if (tid < 2 * NS) fill[tid] = 0;
    __syncthreads();
    if (!open) return;                         // uniform across the launch; nothing has left the CU yet

    if (wave >= Cfg::NCONS) {
        // ================================ gather waves ================================================================
        for (int n = p; n < njobs; n += Cfg::NPROD) {
            const int t = n / Cfg::PPT;
            const int row = (n % Cfg::PPT) * Cfg::RPWV + qr;
            const int j = jA;
            const int begB = buf_ld_i32(r_rowptr, jB >= 0 ? 4u * (unsigned)jB : BUF_OFF);
            const int endB = buf_ld_i32(r_rowptr, jB >= 0 ? 4u * (unsigned)jB + 4u : BUF_OFF);
            const int jC = node_of(job_m(n + 2 * Cfg::NPROD));
            const float scl = buf_ld_f32(r_scale, j >= 0 ? 4u * (unsigned)j : BUF_OFF);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (INIT) acc = buf_ld_f32x4(r_init, (j >= 0 && l4 < S4) ? (unsigned)j * (unsigned)(SP * 4) + 16u * l4 : BUF_OFF);   // sum of the arcs walked earlier (overlap)
            int idB = 0; float wB = 0.0f;
            int rem = endA - begA, eb = begA;
            int idc = idA; float wc = wA;
            bool first = true;
#pragma unroll 1
            while (true) {
#pragma unroll
                for (int s0 = 0; s0 < CH; s0 += DEPTH) {          // DEPTH rows in flight, summed in ascending-source order
                    if (s0 > 0 && !__any(s0 < rem)) break;
                    f32x4 v[DEPTH];
#pragma unroll
                    for (int i = 0; i < DEPTH; ++i) {
                        const unsigned sid = (unsigned)__shfl(idc, s0 + i, LPR);
                        v[i] = buf_ld_f32x4(r_state, (s0 + i < rem && l4 < S4) ? sid * (unsigned)(SP * 4) + 16u * l4 : BUF_OFF);
                    }
                    if (s0 == 0 && first) {        // the next job's row pointers have landed by now: fetch its first CH source ids
                        const int e = begB + l4;
                        idB = buf_ld_i32(r_src, e < endB ? 4u * (unsigned)e : BUF_OFF);
                        wB = HAS_W ? buf_ld_f32(r_w, e < endB ? 4u * (unsigned)e : BUF_OFF) : 0.0f;
                    }
#pragma unroll
                    for (int i = 0; i < DEPTH; ++i) {
                        if (HAS_W) acc += __shfl(wc, s0 + i, LPR) * v[i];
                        else acc += v[i];
                    }
                }
                first = false;
                rem -= CH; eb += CH;
                if (!__any(rem > 0)) break;
                const int e = eb + l4;                 // in-degree > CH: the next CH source ids, one coalesced load per lane group
                idc = buf_ld_i32(r_src, e < endA ? 4u * (unsigned)e : BUF_OFF);
                wc = HAS_W ? buf_ld_f32(r_w, e < endA ? 4u * (unsigned)e : BUF_OFF) : 0.0f;
            }
            if (has_scale) acc *= scl;
            jA = jB; begA = begB; endA = endB; jB = jC; idA = idB; wA = wB;      // rotate: A <- B <- C

            const int s = t % NS, round = t / NS;
            {
                int spin = 0;
                while (__builtin_amdgcn_readfirstlane(f4_ld_acquire(&freed[s])) < round) {
                    if (spin >= SPIN_MAX) { bad = 1; break; }
                    ++spin; __builtin_amdgcn_s_sleep(1);
                }
            }
            if (bad) break;
            *reinterpret_cast<f32x4 *>(Xs + s * Cfg::SLOT + row * LDA + 4 * l4) = acc;
            if (lane == 0) __hip_atomic_fetch_add(&fill[s], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else {
        // ================================ matrix waves ================================================================
        // MFMA fragments: A row / B-C column slot = lane & 15 (r), k sub-index / C row group = lane >> 4 (g).
        // k order: step (q, e) multiplies column 16q + 4g + e of [state | agg] with W1 row of the same index.
        // column order: accumulator tile ci, column slot r = output column 8r + ci; c[ci][reg] = row 4g + reg.
        const int r = lane & 15, g = lane >> 4;
        const int swz = (g & 1) << 2;
        // four LDS bases per lane, everything else is an immediate offset of the ds_read: B operands of k-step (q, e) sit at
        // base + (16q + e) * SP floats (two 16-B halves, own-state rows and neighbour-sum rows of W1)
        const float *ws0 = Ws + (4 * g) * SP + ((8 * r) ^ swz), *ws1 = Ws + (4 * g) * SP + ((8 * r + 4) ^ swz);
        const float *wa0 = ws0 + SP * SP, *wa1 = ws1 + SP * SP;
        for (int t = wave; t < T; t += Cfg::NCONS) {
            const int m0 = (t_first + t * blk_per_xcd) * 16;
            // node ids: of the fragment row r (A operand) and of the rows 4g + reg (C, predicate, store)
            const int mr = m0 + r;
            const int jfr_ = buf_ld_i32(r_rows, mr < count ? 4u * (unsigned)mr : BUF_OFF);
            const int jfr = mr < count ? (rows ? jfr_ : mr) : -1;
            int jrow[4];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int m = m0 + 4 * g + reg;
                const int jr = buf_ld_i32(r_rows, m < count ? 4u * (unsigned)m : BUF_OFF);
                jrow[reg] = m < count ? (rows ? jr : m) : -1;
            }
            // own state row r in k order, then the constant term straight into the accumulators (D = A.B + C)
            f32x4 A[8];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                A[q] = buf_ld_f32x4(r_state, (jfr >= 0 && 4 * q + g < S4) ? (unsigned)(a.row_base + jfr) * (unsigned)(SP * 4) + 64u * q + 16u * g : BUF_OFF);
            f32x4 c[8];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const unsigned off = (jrow[reg] >= 0 && 8 * r < tp.H) ? ((unsigned)jrow[reg] * (unsigned)a.ldC + 8u * r) * 4u : BUF_OFF;
                const f32x4 lo = buf_ld_f32x4(r_C, off), hi = buf_ld_f32x4(r_C, (jrow[reg] >= 0 && 8 * r + 4 < tp.H) ? off + 16u : BUF_OFF);
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) {
                    c[ci][reg] = 8 * r + ci < tp.H ? lo[ci] : 0.0f;
                    c[4 + ci][reg] = 8 * r + 4 + ci < tp.H ? hi[ci] : 0.0f;
                }
            }
            // ---- [own state] . W1_s: needs nothing from the gather waves, runs while they fill this tile's slot ----------------
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (q >= SQ) break;                    // (uniform: k blocks behind the state width multiply zeros)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x4 b0 = *reinterpret_cast<const f32x4 *>(ws0 + (16 * q + e) * SP);
                    const f32x4 b1 = *reinterpret_cast<const f32x4 *>(ws1 + (16 * q + e) * SP);
                    const float av = A[q][e];
#pragma unroll
                    for (int ci = 0; ci < 4; ++ci) {
                        c[ci] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0[ci], c[ci], 0, 0, 0);
                        c[4 + ci] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1[ci], c[4 + ci], 0, 0, 0);
                    }
                }
            }
            // ---- neighbour sums of the tile: slot -> the same registers, slot handed back at once ----------------------------------
            const int s = t % NS, round = t / NS;
            {
                int spin = 0;
                while (__builtin_amdgcn_readfirstlane(f4_ld_acquire(&fill[s])) < Cfg::PPT * (round + 1)) {
                    if (spin >= SPIN_MAX) { bad = 1; break; }
                    ++spin; __builtin_amdgcn_s_sleep(1);
                }
            }
            if (bad) break;
            {
                const float *xs = Xs + s * Cfg::SLOT + r * LDA + 4 * g;
#pragma unroll
                for (int q = 0; q < 8; ++q) A[q] = *reinterpret_cast<const f32x4 *>(xs + 16 * q);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the reads have landed in registers: the slot may be refilled
            if (lane == 0) __hip_atomic_store(&freed[s], round + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (q >= SQ) break;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x4 b0 = *reinterpret_cast<const f32x4 *>(wa0 + (16 * q + e) * SP);
                    const f32x4 b1 = *reinterpret_cast<const f32x4 *>(wa1 + (16 * q + e) * SP);
                    const float av = A[q][e];
#pragma unroll
                    for (int ci = 0; ci < 4; ++ci) {
                        c[ci] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0[ci], c[ci], 0, 0, 0);
                        c[4 + ci] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1[ci], c[4 + ci], 0, 0, 0);
                    }
                }
            }
            // ---- epilogue on rows 4g + reg, columns 8r .. 8r + 7: activation, predicate against the old row, store -------------
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int j = jrow[reg];
                const unsigned off = j >= 0 ? (unsigned)(a.row_base + j) * (unsigned)(SP * 4) + 32u * r : BUF_OFF;
                const f32x4 olo = buf_ld_f32x4(r_state, 8 * r < S ? off : BUF_OFF), ohi = buf_ld_f32x4(r_state, (j >= 0 && 8 * r + 4 < S) ? off + 16u : BUF_OFF);   // L1 / L2 hits: just read above
                f32x4 nlo = {c[0][reg], c[1][reg], c[2][reg], c[3][reg]}, nhi = {c[4][reg], c[5][reg], c[6][reg], c[7][reg]};
                activate4(tp.act, nlo); activate4(tp.act, nhi);
                float d2 = 0.0f, n2 = 0.0f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    nlo[e] = (j >= 0 && 8 * r + e < S) ? nlo[e] : 0.0f;
                    nhi[e] = (j >= 0 && 8 * r + 4 + e < S) ? nhi[e] : 0.0f;
                    const float dl = nlo[e] - olo[e], dh = nhi[e] - ohi[e];
                    d2 = fmaf(dl, dl, d2); d2 = fmaf(dh, dh, d2);
                    n2 = fmaf(olo[e], olo[e], n2); n2 = fmaf(ohi[e], ohi[e], n2);
                }
                if (j >= 0 && 8 * r < S) *reinterpret_cast<f32x4 *>(obase + off) = nlo;               // (pad chunks of the row stay zero: never written)
                if (j >= 0 && 8 * r + 4 < S) *reinterpret_cast<f32x4 *>(obase + off + 16u) = nhi;
#pragma unroll
                for (int o = 8; o >= 1; o >>= 1) {
                    d2 += __shfl_xor(d2, o, 16);
                    n2 += __shfl_xor(n2, o, 16);
                }
                if (j >= 0 && sqrtf(d2) > a.thr * sqrtf(n2)) any = 1;
            }
        }
    }

    any = __syncthreads_or(any);
    bad = __syncthreads_or(bad);
    if (tid == 0) {
        if (any && a.flag_next) atomicOr(a.flag_next, 1);
        if (bad && a.err) atomicOr(a.err, 1);
        if (blockIdx.x == 0 && a.k_out) *a.k_out = a.k_val;
    }
}

template <bool HAS_W, bool INIT, int DEPTH = WideCfg::DEPTH>
int launch_wide_one(Fused2Args &fa, int n_cu, hipStream_t st) {
    using Cfg = WideCfg;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void *)k_state_wide<HAS_W, INIT, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES) != hipSuccess)
            return 1;
        attr = true;
    }
    const int budget = n_cu;                     // one workgroup (16 waves, 157 KB of LDS) per CU, all co-resident
    long total_tiles = 0;
    for (int t = 0; t < fa.n_types; ++t) total_tiles += (fa.tp[t].count + 15) / 16;
    fa.blk_begin[0] = 0;
    for (int t = 0; t < fa.n_types; ++t) {
        const int ntiles = (fa.tp[t].count + 15) / 16;
        int nb = 0;
        if (ntiles > 0) {
            nb = (int)std::min<long>((ntiles + 3) / 4, std::max<long>(8, budget * (long)ntiles / std::max<long>(total_tiles, 1)));
            nb = std::max(8, nb / 8 * 8);          // multiples of 8 (one per XCD), rounded DOWN: the grid stays co-resident
        }
        fa.blk_begin[t + 1] = fa.blk_begin[t] + nb;
    }
    const int grid = fa.blk_begin[fa.n_types];
    if (grid == 0) return 0;
    GNN_SET_KERNEL_NAME("k_state_wide<%s,%s>", HAS_W ? "true" : "false", INIT ? "true" : "false");
    k_state_wide<HAS_W, INIT, DEPTH><<<grid, Cfg::NT, Cfg::LDS_BYTES, st>>>(fa);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

inline int launch_wide(Fused2Args &fa, int n_cu, hipStream_t st, int depth = 0) {
    (void)depth;
    if (fa.agg_init) return fa.w ? launch_wide_one<true, true>(fa, n_cu, st) : launch_wide_one<false, true>(fa, n_cu, st);
    return fa.w ? launch_wide_one<true, false>(fa, n_cu, st) : launch_wide_one<false, false>(fa, n_cu, st);
}

}  // namespace gnn
