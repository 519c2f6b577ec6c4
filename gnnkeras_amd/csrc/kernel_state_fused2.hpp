// Fused state-transition iteration: the hot kernel of the loop on gfx950.  One launch = one iteration of the
// reference's `convergence` + the `condition` of the next iteration (GNN/Models/GNN.py:217-236, :196-214) for every
// node type of the graph:
//   A. coalesced CSR walk: SP/4 lanes own a destination node and sum its neighbour state rows (16 B per lane);
//   B. [state | agg] (64 x 2SP tile in LDS) x W1[state rows ; agg rows] (LDS, loaded once per workgroup) on the f32
//      matrix cores (v_mfma_f32_16x16x4_f32: exact f32 fma chain) + per-node constant C, activation;
//   C. per-node predicate ||new - old||_2 > thr ||old||_2 from the accumulators, OR-reduced to one flag word per
//      launch; new state rows staged through LDS and written as whole rows.
// HBM traffic per iteration = the algorithmic bytes of SURVEY §8d: E(4 + 4S [+4]) + N(4 + 4S + 4S + 4S).
// The memory pipeline is built around what the first profile showed (profiles/r01_v1: the first-generation kernel was
// latency-bound on the dependent chain  node id -> rowptr -> source ids -> state rows):
//   * node slots are software-pipelined: while the state rows of slot s are in flight, the source ids of slot s+1,
//     the row pointers of slot s+2 and the node id of slot s+3 are already being fetched, across tile boundaries;
//   * the <=16 source ids of a node arrive with ONE coalesced load per lane group and are broadcast with
//     ds_bpermute; all of a node's neighbour rows (16 B per lane, one 256-B row per 16 lanes at d = 64) are issued
//     before the first is consumed, then summed in ascending-source order;
//   * the per-node constant C is loaded straight into the MFMA accumulators at tile start (D = A.B + C), so its
//     latency hides under the gather instead of stalling the epilogue;
//   * W1 sits un-padded in LDS behind an XOR swizzle (conflict-free B fragments), A rows use an odd-pair stride.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_general.hpp"

namespace gnn {

struct FusedType {
    const int *rows;   // node ids of this type (nullptr = identity)
    int count;
    const float *Wf;   // folded first layer [in_dim x H] row-major (H == S for one-layer state networks)
    int wrow_state, wrow_agg;
    int H, act;
    // second Dense of a two-layer state network (k_state_fused4<.., L2 = true> only; nullptr otherwise): H -> S
    const float *W2;   // [H x S] row-major
    const float *b2;   // [S]
    int act2;
    const float *Wc;   // XC form of k_state_fused4: [32 x H] folded weight rows of this type's constant inputs, bias row, zeros
    int out2;          // width of the second Dense's output when it is NOT the state (0 = S): W2 is [H x out2], the launch writes the
                       // hidden activations of a deeper network and the caller runs the remaining layers (gnnloop.hip: iteration_prefix)
};

struct Fused2Args {
    const int *gate; int n_gate, gate_stride;      // run iff OR of gate[i * gate_stride], i < n_gate, is non-zero
    const int *rowptr, *src;                       // CSR by (local) destination; src ids index rows of state_in
    const float *w, *row_scale;
    const float *state_in;                         // [n_src_rows, SP]
    float *state_out;                              // rows written at (row_base + j)
    int row_base;                                  // own rows of local node j live at row_base + j (0 on one GPU)
    const float *C; int ldC;
    int n_types;                                   // node types handled by this launch (1 = homogeneous)
    FusedType tp[GNN_MAX_TYPES];
    int blk_begin[GNN_MAX_TYPES + 1];              // workgroups [blk_begin[t], blk_begin[t+1]) serve type t; multiples of 8
    int S;
    float thr;
    int *flag_next;
    float *k_out; float k_val;
    int *err;                                      // sticky error word (workspace): a bounded in-launch wait expired
    // k_state_fused4<.., XC = true> (one node type): instead of the per-node constant C (4 H bytes per node and iteration) the
    // kernel reads the node's constant INPUTS Xc [n, 32] = [labels | agg labels | agg arcs | 1 | 0..] (128 bytes) and multiplies
    // them with the type's Wc [32, H] = folded first-layer rows of those inputs, then the folded bias, then zeros, on the matrix
    // cores (FusedType::Wc; Xc rows are indexed by node id, each laid out for its node's type)
    const float *Xc;
    const float *agg_init;                         // k_state_fused4<.., INIT = true>: [n_local, SP] partial neighbour sums (un-scaled)
                                                   // of the arcs this launch does NOT walk (own-range arcs, summed while the
                                                   // exchange was in flight: distributed.py overlap); nullptr otherwise
    const void *hdr; void *hdr_write;              // k_state_fused4<.., HDR = true> (experiment builds only): the first-job header / where to write it
    // k_state_fused4<.., PEERS = true> (node-range shards, SURVEY 8e "each rank writes its slice to all peers, one hop"): every new row is
    // stored to state_out AND, at the same offset, to the full state buffers of the other ranks (mapped with hipIpcOpenMemHandle:
    // gnn_ipc_open) - the exchange rides in the kernel's epilogue instead of following it as a collective
    float *peer_out[GNN_MAX_PEERS]; int n_peers;
};

// name of the state-transition kernel the calling thread launched last (gnn_last_kernel_name(): bench.py's roofline record)
inline char *last_kernel_name() { static thread_local char name[96] = ""; return name; }
#define GNN_SET_KERNEL_NAME(...) snprintf(::gnn::last_kernel_name(), 96, __VA_ARGS__)

template <int SP, int TM, int NW>
struct Fused2Cfg {
    static constexpr int NT = 64 * NW;               // threads per workgroup
    static constexpr int LPR = SP / 4;               // lanes per node row (16 B each)
    static constexpr int NPP = NT / LPR;             // node slots per pass of the workgroup
    static constexpr int NPASS = TM / NPP;
    static constexpr int IPL = 16 / LPR;             // source ids held per lane (16 per node and chunk)
    static constexpr int LDX = 2 * SP + 2;           // A rows: stride == 2 (mod 32) dwords -> conflict-free ds_read_b32
    static constexpr bool SWZ = SP >= 32;            // W1 un-padded, column ^= 16 on odd k
    static constexpr int LDW = SWZ ? SP : SP + 32;
    static constexpr int NCT = SP / 16;              // 16-column MFMA tiles
    static constexpr int RW = TM / 16;               // waves along rows
    static constexpr int CW = NW / RW;               // waves along columns
    static constexpr int CT_PER_WAVE = NCT / CW;
    static_assert(NPASS >= 1 && TM % NPP == 0, "tile must be a whole number of passes");
    static_assert(RW >= 1 && RW <= NW && NW % RW == 0 && NCT % CW == 0, "unsupported tile / width combination");
    static constexpr size_t LDS_BYTES = sizeof(float) * ((size_t)TM * LDX + 2 * SP * LDW) + sizeof(int) * TM;
};

template <int SP, bool HAS_W, int TM, int NW, int DEPTH>
__global__ void __launch_bounds__(64 * NW, NW >= 12 ? NW / 2 : (NW == 8 ? 4 : 2)) k_state_fused2(Fused2Args a) {
    {
        int open = a.gate == nullptr;
        for (int i = 0; i < a.n_gate && !open; ++i) open |= a.gate[(size_t)i * a.gate_stride] != 0;
        if (!open) return;
    }
    using Cfg = Fused2Cfg<SP, TM, NW>;
    constexpr int NT = Cfg::NT;
    constexpr int LPR = Cfg::LPR, IPL = Cfg::IPL, LDX = Cfg::LDX, LDW = Cfg::LDW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *Xs = reinterpret_cast<float *>(smem);                         // [TM][LDX]  : [state | agg]
    float *Ws = Xs + TM * LDX;                                           // [2SP][LDW] : W1 rows (state ; agg)
    int *jid = reinterpret_cast<int *>(Ws + 2 * SP * LDW);               // [TM] local node id per tile row, -1 = pad

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int S = a.S;
    // which node type this workgroup serves (one launch covers every type of a composite graph)
    int ty = 0;
    while (ty + 1 < a.n_types && (int)blockIdx.x >= a.blk_begin[ty + 1]) ++ty;
    const FusedType tp = a.tp[ty];
    const int bid = blockIdx.x - a.blk_begin[ty], nblk = a.blk_begin[ty + 1] - a.blk_begin[ty];
    const int count = tp.count;
    const int *__restrict__ rows = tp.rows;

    if (S == SP && tp.H == SP && (reinterpret_cast<uintptr_t>(tp.Wf) & 15) == 0) {
        // full-width state: 16-byte pieces of weight rows, all loads of the fill in flight before the first LDS store (see k_state_fused4)
        constexpr int N4 = 2 * SP * SP / 4, NV = (N4 + NT - 1) / NT;
        f32x4 v[NV];
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int i4 = min(tid + u * NT, N4 - 1), k = i4 / (SP / 4), n = (i4 % (SP / 4)) * 4;
            v[u] = *reinterpret_cast<const f32x4 *>(tp.Wf + (size_t)(k < SP ? tp.wrow_state + k : tp.wrow_agg + (k - SP)) * SP + n);
        }
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int i4 = tid + u * NT, k = i4 / (SP / 4), n = (i4 % (SP / 4)) * 4;
            if (i4 < N4) *reinterpret_cast<f32x4 *>(Ws + k * LDW + (Cfg::SWZ ? (n ^ ((k & 1) << 4)) : n)) = v[u];
        }
    } else
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

    const char *__restrict__ sbase = reinterpret_cast<const char *>(a.state_in);   // state_in spans < 4 GiB (checked by the launcher)
    const char *__restrict__ cbase = reinterpret_cast<const char *>(a.C);
    char *__restrict__ obase = reinterpret_cast<char *>(a.state_out);
    const int q = tid / LPR;          // node slot inside a pass
    const int l4 = tid % LPR;         // 16-B column chunk of the row owned by this lane

    // ---- slot pipeline: slot index n counts (tile, pass) pairs of this workgroup -------------------------------------
    auto slot_m = [&](int n) -> int {               // global row number m of slot n for this lane group, or -1
        const int tile = t_first + (n / Cfg::NPASS) * blk_per_xcd;
        if (tile >= t_end) return -1;
        const int m = tile * TM + (n % Cfg::NPASS) * Cfg::NPP + q;
        return m < count ? m : -1;
    };
    auto node_of = [&](int m) -> int { return m < 0 ? -1 : (rows ? rows[m] : m); };

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

        // ---- A. gather + aggregate, one node slot per lane group and pass --------------------------------------------
#pragma unroll 1
        for (int pass = 0; pass < Cfg::NPASS; ++pass) {
            // prefetch for later slots first (their addresses are already known)
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

            // current slot: every neighbour row in flight before the first add
            f32x4 own = {0.f, 0.f, 0.f, 0.f}, acc = {0.f, 0.f, 0.f, 0.f};
            const int deg = end0 - beg0;
            if (j0 >= 0) own = *reinterpret_cast<const f32x4 *>(sbase + ((unsigned)(a.row_base + j0) * (unsigned)(SP * 4) + 16u * l4));
            {
                // chunks of 16 neighbours; the first chunk's ids were prefetched a slot ago, later chunks (in-degree
                // > 16, rare) fetch theirs in line.  One copy of the 16-load block: keeps the kernel near 128 VGPRs.
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
                            // 32-bit byte offset off a wave-uniform base: SGPR base + VGPR offset addressing
                            const unsigned off = (unsigned)__shfl(idc[(s0 + i) / LPR], (s0 + i) % LPR, LPR) * (unsigned)(SP * 4) + 16u * l4;
                            v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                            if (s0 + i < rem) v[i] = *reinterpret_cast<const f32x4 *>(sbase + off);
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

            // rotate the pipeline
            j0 = j1; beg0 = beg1; end0 = end1;
#pragma unroll
            for (int u = 0; u < IPL; ++u) { ids0[u] = ids1[u]; ws0[u] = ws1[u]; }
            j1 = j2; beg1 = beg2; end1 = end2;
            j2 = j3; j3 = j4;
            ++n_slot;
        }
        __syncthreads();

        // ---- B. [state | agg] . W1 on the f32 matrix cores ------------------------------------------------------------
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

        // ---- C. activation, predicate, stage new rows ----------------------------------------------------------------
        // C/D layout: col = 16*ct + (lane & 15), row = 16*rt + 4*(lane >> 4) + reg.  With CW > 1 a row's columns are
        // split over CW waves: partial sums meet in LDS (the agg half of Xs is dead after the MFMA loop).
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
            __syncthreads();                                   // every wave is done reading Xs (A fragments, old state)
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
                Xs[row * LDX + 16 * (cw * Cfg::CT_PER_WAVE + ci) + r] = c[ci][reg];   // own (row, col) slot only
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
        // whole-row (4*SP bytes) coalesced stores of the new state
#pragma unroll 1
        for (int pass = 0; pass < Cfg::NPASS; ++pass) {
            const int nl = pass * Cfg::NPP + q;
            const int j = jid[nl];
            if (j >= 0) {
                const float *xr = Xs + nl * LDX + 4 * l4;
                const float2 lo = *reinterpret_cast<const float2 *>(xr), hi = *reinterpret_cast<const float2 *>(xr + 2);
                *reinterpret_cast<f32x4 *>(obase + ((unsigned)(a.row_base + j) * (unsigned)(SP * 4) + 16u * l4)) = (f32x4){lo.x, lo.y, hi.x, hi.y};
            }
        }
    }

    any = __syncthreads_or(any);
    if (tid == 0) {
        if (any && a.flag_next) atomicOr(a.flag_next, 1);
        if (blockIdx.x == 0 && a.k_out) *a.k_out = a.k_val;
    }
}

template <int SP, bool HAS_W, int TM, int NW, int DEPTH = 16>
int launch_fused2_one(Fused2Args &fa, int n_cu, hipStream_t st) {
    using Cfg = Fused2Cfg<SP, TM, NW>;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void *)k_state_fused2<SP, HAS_W, TM, NW, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)Cfg::LDS_BYTES) != hipSuccess) return 1;
        attr = true;
    }
    // workgroups per type: proportional to its tiles, whole multiples of 8 (one per XCD), never more than its tiles need
    const int blocks_per_cu = std::max(1, std::min(NW >= 8 ? 2 : (NW == 4 ? 4 : 8), (int)(160 * 1024 / Cfg::LDS_BYTES)));
    const int budget = blocks_per_cu * n_cu;
    long total_tiles = 0;
    for (int t = 0; t < fa.n_types; ++t) total_tiles += (fa.tp[t].count + TM - 1) / TM;
    fa.blk_begin[0] = 0;
    for (int t = 0; t < fa.n_types; ++t) {
        const int ntiles = (fa.tp[t].count + TM - 1) / TM;
        int nb = 0;
        if (ntiles > 0) {
            nb = (int)std::min<long>((ntiles + 7) / 8 * 8, std::max<long>(8, budget * (long)ntiles / std::max<long>(total_tiles, 1)));
            nb = std::max(8, nb / 8 * 8);          // round DOWN: the whole grid must stay co-resident (no second wave)
        }
        fa.blk_begin[t + 1] = fa.blk_begin[t] + nb;
    }
    const int grid = fa.blk_begin[fa.n_types];
    if (grid == 0) return 0;
    GNN_SET_KERNEL_NAME("k_state_fused2<%d,%s,%d,%d,%d>", SP, HAS_W ? "true" : "false", TM, NW, DEPTH);
    k_state_fused2<SP, HAS_W, TM, NW, DEPTH><<<grid, Cfg::NT, Cfg::LDS_BYTES, st>>>(fa);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// Per-arc weights cost ~50 more live VGPRs (weight broadcasts + products): under the 128-VGPR cap of the 8-wave
// workgroup that spills into the gather loop, so weighted graphs run the 4-wave shape (256-VGPR budget) instead.
template <int SP, int TM, int NW>
int launch_fused2_w(Fused2Args &fa, int n_cu, hipStream_t st) {
    if (fa.w) return launch_fused2_one<SP, true, TM, (NW == 8 ? 4 : NW)>(fa, n_cu, st);
    return launch_fused2_one<SP, false, TM, NW>(fa, n_cu, st);
}

// `waves` = 8 (default: 512-thread workgroups, 16 waves per CU) or 4 (256-thread workgroups, 256-VGPR budget)
inline int launch_fused2(Fused2Args &fa, int SP, int waves, int n_cu, hipStream_t st) {
    switch (SP) {
        case 16: return launch_fused2_w<16, 64, 4>(fa, n_cu, st);
        case 32: return waves == 8 ? launch_fused2_w<32, 64, 8>(fa, n_cu, st) : launch_fused2_w<32, 64, 4>(fa, n_cu, st);
        case 64:
            return waves == 8 ? launch_fused2_w<64, 64, 8>(fa, n_cu, st) : launch_fused2_w<64, 64, 4>(fa, n_cu, st);
        default: return 1;
    }
}

}  // namespace gnn
