// Fused state-transition iteration, third generation: the gather of tile t+1 is in flight while tile t is on the
// matrix cores.  Same contract and arguments as k_state_fused2 (one launch = one iteration of the reference's
// `convergence` + the `condition` of the next one, GNN/Models/GNN.py:217-236, :196-214, for every node type).
//
// Why: profiles/r01_v6 showed fused2 at 548 us / iteration on C4 while the bare access pattern (scripts/micro/
// gather_ceiling.hip: same rows, no arithmetic) takes 428 us, and fused2 with the MFMA loop compiled out 492 us: the
// matrix phase and the two barriers per tile were serialised with the gather instead of hidden under it.
//
// Shape: 512 threads; a lane group of SP/4 lanes owns ONE node per tile (tile = 2048/SP nodes: 32 at d = 64), so a
// tile's whole gather is one batch of <=16 neighbour rows + the own row per lane group, issued right after the
// tile's single barrier and consumed at the top of the next trip:
//
//     trip t:   consume rows(t) -> Xs[t&1]         (waits for the loads issued in trip t-1)
//               barrier
//               issue: C(t), node ids / row pointers / source ids of later tiles, rows(t+1)
//               MFMA(t) from Xs[t&1]  -> activation -> predicate partials(t) -> new rows stored straight
//               from the accumulator layout (4 rows x 64 B per store instruction)
//               predicate(t-1) from the partials written a trip ago
//
// Xs and the predicate partials are double-buffered, so one barrier per tile orders everything.  Every load whose
// issue depends on data (row pointers, masks, optional arrays) is a raw buffer load with an out-of-range offset when
// predicated off: the instruction stream is branch-free, so hipcc's vmcnt accounting stays exact and the MFMA phase
// never waits for the rows in flight.
#pragma once
#include <hip/hip_runtime.h>
#include "kernel_state_fused2.hpp"
#include "buffer_ops.hpp"

namespace gnn {

template <int SP, int NW>
struct Fused3Cfg {
    static constexpr int NT = 64 * NW;
    static constexpr int LPR = SP / 4;               // lanes per node row (16 B each)
    static constexpr int TM = NT / LPR;              // nodes per tile: one per lane group
    static constexpr int IPL = 16 / LPR;             // source ids held per lane (16 per node and chunk)
    static constexpr int LDX = 2 * SP + 2;
    static constexpr bool SWZ = SP >= 32;
    static constexpr int LDW = SWZ ? SP : SP + 32;
    static constexpr int RW = TM / 16;               // waves along rows
    static constexpr int CW = NW / RW;               // waves along columns == 16-column tiles: one MFMA tile per wave
    static_assert(CW * 16 == SP && RW * CW == NW, "one 16x16 output tile per wave");
    static constexpr size_t LDS_BYTES = sizeof(float) * ((size_t)2 * TM * LDX + 2 * SP * LDW + (size_t)2 * TM * CW * 2) + sizeof(int) * 2 * TM;
};

template <int SP, bool HAS_W, int NW>
__global__ void __launch_bounds__(64 * NW, NW == 12 ? 3 : 4) k_state_fused3(Fused2Args a) {
    {
        int open = a.gate == nullptr;
        for (int i = 0; i < a.n_gate && !open; ++i) open |= a.gate[(size_t)i * a.gate_stride] != 0;
        if (!open) return;
    }
    using Cfg = Fused3Cfg<SP, NW>;
    constexpr int NT = Cfg::NT, TM = Cfg::TM, LPR = Cfg::LPR, IPL = Cfg::IPL, LDX = Cfg::LDX, LDW = Cfg::LDW, CW = Cfg::CW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *Xs = reinterpret_cast<float *>(smem);                         // [2][TM][LDX] : [state | agg]
    float *Ws = Xs + 2 * TM * LDX;                                       // [2SP][LDW]   : W1 rows (state ; agg)
    float *Ps = Ws + 2 * SP * LDW;                                       // [2][TM][CW][2] : predicate partial sums
    int *jids = reinterpret_cast<int *>(Ps + 2 * TM * CW * 2);           // [2][TM] node id per tile row, -1 = pad

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int S = a.S;
    int ty = 0;
    while (ty + 1 < a.n_types && (int)blockIdx.x >= a.blk_begin[ty + 1]) ++ty;
    const FusedType tp = a.tp[ty];
    const int bid = blockIdx.x - a.blk_begin[ty], nblk = a.blk_begin[ty + 1] - a.blk_begin[ty];
    const int count = tp.count;

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

    const __amdgpu_buffer_rsrc_t r_state = buf_rsrc(a.state_in), r_C = buf_rsrc(a.C), r_rowptr = buf_rsrc(a.rowptr),
                                 r_src = buf_rsrc(a.src), r_w = buf_rsrc(HAS_W ? a.w : nullptr),
                                 r_scale = buf_rsrc(a.row_scale), r_rows = buf_rsrc(tp.rows);
    const bool has_rows = tp.rows != nullptr, has_scale = a.row_scale != nullptr;
    char *__restrict__ obase = reinterpret_cast<char *>(a.state_out);
    const int q = tid / LPR;          // node slot of this lane group inside the tile
    const int l4 = tid % LPR;         // 16-B column chunk of the row owned by this lane
    const int rt = wave % Cfg::RW, cw = wave / Cfg::RW;
    const int col = 16 * cw + r;      // output column of this lane in the accumulator layout

    auto tile_of = [&](int n) -> int { return t_first + n * blk_per_xcd; };
    // node id of gather slot n for this lane group (-1: none); a load only for composite graphs (type row lists)
    auto slot_node = [&](int n) -> int {
        const int m = tile_of(n) * TM + q;
        const bool ok = tile_of(n) < t_end && m < count;
        const int jr = buf_ld_i32(r_rows, ok ? 4u * (unsigned)m : BUF_OFF);
        return ok ? (has_rows ? jr : m) : -1;
    };
    auto load_C = [&](const int (&jr)[4], f32x4 &c) {
        const int colc = min(col, S - 1);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            c[reg] = buf_ld_f32(r_C, (jr[reg] >= 0 && col < S) ? ((unsigned)jr[reg] * (unsigned)a.ldC + (unsigned)colc) * 4u : BUF_OFF);
    };

    // ---- pipeline registers: A = the slot whose rows are (about to be) in flight, B/C/D the slots after it ----------
    int jA, begA, endA, jB, begB, endB, begC, endC, jC, jD, jE;
    int idsA[IPL], idsB[IPL];
    float wsA[IPL], wsB[IPL];
    float sclA = 1.0f;
    f32x4 own, v[16];

    jA = slot_node(0); jB = slot_node(1); jC = slot_node(2); jD = slot_node(3);
    begA = buf_ld_i32(r_rowptr, jA >= 0 ? 4u * (unsigned)jA : BUF_OFF); endA = buf_ld_i32(r_rowptr, jA >= 0 ? 4u * (unsigned)jA + 4u : BUF_OFF);
    begB = buf_ld_i32(r_rowptr, jB >= 0 ? 4u * (unsigned)jB : BUF_OFF); endB = buf_ld_i32(r_rowptr, jB >= 0 ? 4u * (unsigned)jB + 4u : BUF_OFF);
#pragma unroll
    for (int u = 0; u < IPL; ++u) {
        const int e = begA + u * LPR + l4;
        idsA[u] = buf_ld_i32(r_src, e < endA ? 4u * (unsigned)e : BUF_OFF);
        wsA[u] = HAS_W ? buf_ld_f32(r_w, e < endA ? 4u * (unsigned)e : BUF_OFF) : 0.0f;
    }

    // issue everything that can be known now: later slots' ids / row pointers / source ids, then slot A's rows
    auto issue = [&](int n_e) {
#pragma unroll
        for (int u = 0; u < IPL; ++u) {
            const int e = begB + u * LPR + l4;
            idsB[u] = buf_ld_i32(r_src, e < endB ? 4u * (unsigned)e : BUF_OFF);
            wsB[u] = HAS_W ? buf_ld_f32(r_w, e < endB ? 4u * (unsigned)e : BUF_OFF) : 0.0f;
        }
        begC = buf_ld_i32(r_rowptr, jC >= 0 ? 4u * (unsigned)jC : BUF_OFF);
        endC = buf_ld_i32(r_rowptr, jC >= 0 ? 4u * (unsigned)jC + 4u : BUF_OFF);
        jE = slot_node(n_e);
        sclA = buf_ld_f32(r_scale, jA >= 0 ? 4u * (unsigned)jA : BUF_OFF);
        own = buf_ld_f32x4(r_state, jA >= 0 ? (unsigned)(a.row_base + jA) * (unsigned)(SP * 4) + 16u * l4 : BUF_OFF);
        const int deg = endA - begA;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const unsigned sid = (unsigned)__shfl(idsA[i / LPR], i % LPR, LPR);
            v[i] = buf_ld_f32x4(r_state, i < deg ? sid * (unsigned)(SP * 4) + 16u * l4 : BUF_OFF);
        }
    };
    issue(4);

    int any = 0;
    int it = 0;
    for (; tile_of(it) < t_end; ++it) {
        float *X = Xs + (it & 1) * (TM * LDX);
        float *P = Ps + (it & 1) * (TM * CW * 2);

        // ---- 1. consume slot A (tile `it`): neighbour rows summed in ascending-source order ----------------------------
        {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (HAS_W) acc += __shfl(wsA[i / LPR], i % LPR, LPR) * v[i];
                else acc += v[i];
            }
            int rem = endA - begA - 16, eb = begA + 16;
#pragma unroll 1
            while (__any(rem > 0)) {              // in-degree > 16 (rare): fetch the next 16 ids and rows in line
                int idc[IPL]; float wsc[IPL];
#pragma unroll
                for (int u = 0; u < IPL; ++u) {
                    const int e = eb + u * LPR + l4;
                    idc[u] = buf_ld_i32(r_src, e < endA ? 4u * (unsigned)e : BUF_OFF);
                    wsc[u] = HAS_W ? buf_ld_f32(r_w, e < endA ? 4u * (unsigned)e : BUF_OFF) : 0.0f;
                }
                f32x4 x[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const unsigned sid = (unsigned)__shfl(idc[i / LPR], i % LPR, LPR);
                    x[i] = buf_ld_f32x4(r_state, i < rem ? sid * (unsigned)(SP * 4) + 16u * l4 : BUF_OFF);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (HAS_W) acc += __shfl(wsc[i / LPR], i % LPR, LPR) * x[i];
                    else acc += x[i];
                }
                rem -= 16; eb += 16;
            }
            if (has_scale) acc *= sclA;
            float *xr = X + q * LDX + 4 * l4;                        // rows are 8-B aligned: two b64 stores each
            *reinterpret_cast<float2 *>(xr) = make_float2(own[0], own[1]);
            *reinterpret_cast<float2 *>(xr + 2) = make_float2(own[2], own[3]);
            *reinterpret_cast<float2 *>(xr + SP) = make_float2(acc[0], acc[1]);
            *reinterpret_cast<float2 *>(xr + SP + 2) = make_float2(acc[2], acc[3]);
            if (l4 == 0) jids[(it & 1) * TM + q] = jA;
        }
        // rotate the slot pipeline: A <- B <- C <- D <- E
        jA = jB; begA = begB; endA = endB;
#pragma unroll
        for (int u = 0; u < IPL; ++u) { idsA[u] = idsB[u]; wsA[u] = wsB[u]; }
        jB = jC; begB = begC; endB = endC;
        jC = jD; jD = jE;

        __syncthreads();   // X complete; every wave is past trip it-1 (its reads of the other X / P buffers are done)

        // ---- 2. issue the loads of later tiles; nothing below waits for them -------------------------------------------
        int jrow[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) jrow[reg] = jids[(it & 1) * TM + 16 * rt + 4 * g + reg];
        f32x4 cC;                                  // per-node constant of this tile: oldest load, lands under the MFMA loop
        load_C(jrow, cC);
        issue(it + 5);

        // ---- 3. [state | agg] . W1 on the f32 matrix cores --------------------------------------------------------------
        const float *xrow = X + (16 * rt + r) * LDX + g;
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int s4 = 0; s4 < 2 * SP / 4; ++s4) {
            const float av = xrow[4 * s4];
            const int k = 4 * s4 + g;
            const float bv = Ws[k * LDW + (Cfg::SWZ ? (col ^ ((k & 1) << 4)) : col)];
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c, 0, 0, 0);
        }
        c += cC;

        // ---- 4. activation, predicate partial sums, new rows straight from the accumulator layout ----------------------
        // C/D layout: col = 16*cw + (lane & 15), row = 16*rt + 4*(lane >> 4) + reg.
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = 16 * rt + 4 * g + reg;
            const float nv = (jrow[reg] >= 0 && col < S) ? activate(tp.act, c[reg]) : 0.0f;
            const float ov = X[row * LDX + col];
            const float d = nv - ov;
            float d2 = d * d, n2 = ov * ov;
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) {
                d2 += __shfl_xor(d2, off, 16);
                n2 += __shfl_xor(n2, off, 16);
            }
            if (CW > 1) {
                if (r == 0) *reinterpret_cast<float2 *>(P + (row * CW + cw) * 2) = make_float2(d2, n2);
            } else if (jrow[reg] >= 0 && sqrtf(d2) > a.thr * sqrtf(n2)) any = 1;
            if (jrow[reg] >= 0)
                *reinterpret_cast<float *>(obase + ((unsigned)(a.row_base + jrow[reg]) * (unsigned)(SP * 4) + 4u * (unsigned)col)) = nv;
        }
        if (CW > 1 && it > 0 && cw == 0 && r == 0) {       // predicate of the previous tile: its partials are complete
            const float *Pp = Ps + ((it - 1) & 1) * (TM * CW * 2);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = 16 * rt + 4 * g + reg;
                float d2 = 0.0f, n2 = 0.0f;
#pragma unroll
                for (int w2 = 0; w2 < CW; ++w2) { d2 += Pp[(row * CW + w2) * 2]; n2 += Pp[(row * CW + w2) * 2 + 1]; }
                if (sqrtf(d2) > a.thr * sqrtf(n2)) any = 1;     // pad rows hold 0 > thr*0: false
            }
        }
    }

    __syncthreads();
    if (CW > 1 && it > 0 && cw == 0 && r == 0) {
        const float *Pp = Ps + ((it - 1) & 1) * (TM * CW * 2);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = 16 * rt + 4 * g + reg;
            float d2 = 0.0f, n2 = 0.0f;
#pragma unroll
            for (int w2 = 0; w2 < CW; ++w2) { d2 += Pp[(row * CW + w2) * 2]; n2 += Pp[(row * CW + w2) * 2 + 1]; }
            if (sqrtf(d2) > a.thr * sqrtf(n2)) any = 1;
        }
    }
    any = __syncthreads_or(any);
    if (tid == 0) {
        if (any && a.flag_next) atomicOr(a.flag_next, 1);
        if (blockIdx.x == 0 && a.k_out) *a.k_out = a.k_val;
    }
}

template <int SP, bool HAS_W, int NW>
int launch_fused3_one(Fused2Args &fa, int n_cu, hipStream_t st) {
    using Cfg = Fused3Cfg<SP, NW>;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void *)k_state_fused3<SP, HAS_W, NW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)Cfg::LDS_BYTES) != hipSuccess) return 1;
        attr = true;
    }
    const int budget = (NW == 12 ? 1 : 2) * n_cu;   // 12 or 16 waves per CU, all workgroups co-resident
    long total_tiles = 0;
    for (int t = 0; t < fa.n_types; ++t) total_tiles += (fa.tp[t].count + Cfg::TM - 1) / Cfg::TM;
    fa.blk_begin[0] = 0;
    for (int t = 0; t < fa.n_types; ++t) {
        const int ntiles = (fa.tp[t].count + Cfg::TM - 1) / Cfg::TM;
        int nb = 0;
        if (ntiles > 0) {
            nb = (int)std::min<long>((ntiles + 7) / 8 * 8, std::max<long>(8, budget * (long)ntiles / std::max<long>(total_tiles, 1)));
            nb = std::max(8, nb / 8 * 8);          // round DOWN: the whole grid must stay co-resident (no second wave)
        }
        fa.blk_begin[t + 1] = fa.blk_begin[t] + nb;
    }
    const int grid = fa.blk_begin[fa.n_types];
    if (grid == 0) return 0;
    k_state_fused3<SP, HAS_W, NW><<<grid, Cfg::NT, Cfg::LDS_BYTES, st>>>(fa);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// `waves` = 12 (default: one 768-thread workgroup per CU, 168-VGPR budget: 16 neighbour rows + the slot pipeline stay
// in registers across the MFMA phase) or 8 (two 512-thread workgroups per CU, 128 VGPRs)
template <int SP, bool HAS_W>
int launch_fused3_w(Fused2Args &fa, int waves, int n_cu, hipStream_t st) {
    return waves == 8 ? launch_fused3_one<SP, HAS_W, 8>(fa, n_cu, st) : launch_fused3_one<SP, HAS_W, 12>(fa, n_cu, st);
}

inline int launch_fused3(Fused2Args &fa, int SP, int waves, int n_cu, hipStream_t st) {
    switch (SP) {
        case 16: return fa.w ? launch_fused3_w<16, true>(fa, waves, n_cu, st) : launch_fused3_w<16, false>(fa, waves, n_cu, st);
        case 32: return fa.w ? launch_fused3_w<32, true>(fa, waves, n_cu, st) : launch_fused3_w<32, false>(fa, waves, n_cu, st);
        case 64: return fa.w ? launch_fused3_w<64, true>(fa, waves, n_cu, st) : launch_fused3_w<64, false>(fa, waves, n_cu, st);
        default: return 1;
    }
}

}  // namespace gnn
