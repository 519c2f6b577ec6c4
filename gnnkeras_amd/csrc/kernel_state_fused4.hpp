// Fused state-transition iteration, wave-specialised: gather waves and matrix waves run decoupled.
// Same contract and arguments as k_state_fused2 (one launch = one iteration of the reference's `convergence` + the
// `condition` of the next one, GNN/Models/GNN.py:217-236, :196-214, for every node type).
//
// Why (measurements in profiles/r01_gather_sweep.txt, scripts/micro/gather_sweep.hip): on this access pattern the
// memory system rewards the NUMBER OF WAVES with a gather outstanding, not the rows in flight per wave — the bare
// pattern runs at 7.7-7.9 TB/s (algorithmic) with 8 waves / SIMD at any depth from 2 to 12 rows, and at 5.2-6.2 TB/s
// with 4 waves / SIMD at any depth.  k_state_fused2 keeps 4 waves / SIMD (128 VGPRs: 16 rows in flight + MFMA) and
// parks all of them at two barriers per tile.  Here a 1024-thread workgroup (2 per CU = 8 waves / SIMD, 64 VGPRs) is
// split by role:
//   * 12 gather waves: each lane group of SP/4 lanes owns a node, walks its CSR row (DEPTH rows in flight), and
//     drops [own state | neighbour sum] into a 16-row slot of an LDS ring — no barrier anywhere, a wave only ever
//     waits for its own loads or for a free slot;
//   * 4 matrix waves (one per SIMD): each takes every 4th 16-row slot, starts from the per-node constant C (loaded
//     while it waits for the slot to fill), runs [state | agg] . W1 on v_mfma_f32_16x16x4_f32, applies the activation,
//     evaluates the convergence predicate for whole rows in registers and stores the new rows.
// Slots are handed over with two monotonic LDS counters per slot (rows filled / rounds consumed): release on the
// writer's side, acquire on the reader's, both workgroup scope.  Spin loops sleep and are bounded, and a wait that
// expires raises the sticky error word `a.err`, which the host entry folds into k (k < 0 = results invalid): a lost
// hand-off is an error the caller sees, never a silently wrong state and never a hung GPU.
// Every global load on either side is a raw buffer load whose offset is out of range when predicated off: no branch
// ever surrounds a memory operation, so hipcc's waits are exact counts (with `cond ? *p : 0` it wrapped each of the
// matrix waves' 16 C loads in its own exec-mask branch + vmcnt(0): 16 serial round trips per tile, 620 us/iteration
// instead of 480).  C4: 480 us/iteration = 7.0 TB/s algorithmic = 88 % of 8 TB/s; the bare pattern's best is 428 us.
#pragma once
#include <hip/hip_runtime.h>
#include "kernel_state_fused2.hpp"
#include "buffer_ops.hpp"

namespace gnn {

constexpr int XC_K = 32;                          // constant-input columns of the XC variant (padded; one of them is the bias' 1)

template <int SP, int NC = 4, bool L2 = false, bool XC = false, int VPL = 1>
struct Fused4Cfg {
    static constexpr int NW = 16, NT = 64 * NW;
    static constexpr int NCONS = NC;                 // matrix waves (wave ids 0..NC-1)
    static constexpr int NPROD = NW - NCONS;         // gather waves
    static constexpr int LPR = SP / (4 * VPL);       // lanes per node row (VPL 16-byte pieces each; VPL = 2: the C3 experiment of round 5,
                                                     // twice the nodes per gather wave and trip - profiles/r05_c3_experiments.txt)
    static constexpr int RPWV = 64 / LPR;            // rows a gather wave fills per tile
    static constexpr int PPT = 16 / RPWV;            // gather waves per 16-row tile
    static constexpr int IPL = 16 / LPR;             // source ids held per lane (16 per node and chunk)
    static constexpr int LDX = 2 * SP + 2;           // A rows: stride == 2 (mod 32) dwords -> conflict-free ds_read_b32
    static constexpr bool SWZ = SP >= 32;
    static constexpr int LDW = SWZ ? SP : SP + 32;
    static constexpr int NCT = SP / 16;              // 16-column MFMA tiles per row tile
    static constexpr int NS = SP == 64 ? (L2 ? 3 : (XC ? 4 : 5)) : 8;   // ring slots (two-layer networks trade slots for W2, XC one for Wc;
                                                             // 4 and 5 slots measure the same at C4: 480.4 / 478.7 vs 482.3 / 482.4 us)
    static constexpr int WROWS = 2 * SP + (XC ? XC_K : 0);   // rows of the weight matrix in LDS: state ; agg ; (XC) constant inputs
    static constexpr int SLOT = 16 * LDX;            // floats per slot
    static constexpr int W2F = L2 ? SP * LDW + SP : 0;        // floats of the second layer: W2 [SP][LDW] + b2 [SP]
    static constexpr size_t LDS_BYTES = sizeof(float) * ((size_t)NS * SLOT + WROWS * LDW + W2F) + sizeof(int) * 2 * NS;
};

// the activation of 4 values with ONE wave-uniform switch around them
__device__ __forceinline__ void activate4(int act, f32x4 &x) {
#define F4_ALL(expr)                                        \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {         \
        const float v = x[e];                               \
        x[e] = (expr);                                      \
    }
    switch (act) {
        case GNN_ACT_RELU: F4_ALL(fmaxf(v, 0.0f)) break;
        case GNN_ACT_SELU: F4_ALL(v > 0.0f ? 1.0507009873554805f * v : (1.0507009873554805f * 1.6732632423543772f) * (expf(v) - 1.0f)) break;
        case GNN_ACT_TANH: F4_ALL(tanhf(v)) break;
        case GNN_ACT_SIGMOID: F4_ALL(1.0f / (1.0f + expf(-v))) break;
        case GNN_ACT_ELU: F4_ALL(v > 0.0f ? v : expf(v) - 1.0f) break;
        case GNN_ACT_SOFTPLUS: F4_ALL(v > 20.0f ? v : log1pf(expf(v))) break;
        default: break;
    }
#undef F4_ALL
}

// -DGNN_F4_TIMELINE: four wall-clock stamps per workgroup (entry, W1 fill done, first deposit, exit: scripts/f4_timeline.py) - cheap
// enough to leave the launch as it is; experiment builds only
#ifdef GNN_F4_TIMELINE
__device__ unsigned long long g_f4_wg[1024][4];
__device__ __forceinline__ unsigned long long f4_stamp() { return wall_clock64(); }   // s_memrealtime: 100 MHz, the same base on every CU
                                                                                         // (s_memtime runs at the shader clock and differs from CU to CU)
#define F4_STAMP(i) do { if (blockIdx.x < 1024) g_f4_wg[blockIdx.x][i] = f4_stamp(); } while (0)
#else
#define F4_STAMP(i) do { } while (0)
#endif
// -DGNN_F4_PROFILE: phase timers (scripts/f4_prof.py); experiment only, the timers themselves drain the memory pipeline
#ifdef GNN_F4_PROFILE
__device__ __forceinline__ unsigned long long f4_now() {
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}
__device__ unsigned long long g_f4_prof[8];
#endif

__device__ __forceinline__ int f4_ld_acquire(const int *p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int SP, bool HAS_W, int DEPTH, int NC, bool L2, bool INIT = false, bool XC = false, int VPL = 1, bool HDR = false, bool PEERS = false>
__global__ void __launch_bounds__(1024, 8) k_state_fused4(Fused2Args a) {
    // The gate word(s), the W1 fill and the gather waves' first CSR row are all fetched before anything is waited for:
    // three dependent round trips at the head of every launch become one.  Nothing is written to global memory before
    // the gate has been checked (after the fill's barrier).
    int open = a.gate == nullptr;
    for (int i = 0; i < a.n_gate; ++i) open |= a.gate[(size_t)i * a.gate_stride] != 0;
    using Cfg = Fused4Cfg<SP, NC, L2, XC, VPL>;
    constexpr int NT = Cfg::NT, LPR = Cfg::LPR, IPL = Cfg::IPL, LDX = Cfg::LDX, LDW = Cfg::LDW, NS = Cfg::NS;
#ifndef GNN_F4_SPIN_MAX
#define GNN_F4_SPIN_MAX (1 << 22)
#endif
    constexpr int SPIN_MAX = GNN_F4_SPIN_MAX;      // -DGNN_F4_SPIN_MAX=0: debug build whose every wait expires at once
    int bad = 0;                                   // wave-uniform (scalar registers): some bounded wait of this wave expired
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *Xs = reinterpret_cast<float *>(smem);                         // [NS][16][LDX] : [state | agg]
    float *Ws = Xs + NS * Cfg::SLOT;                                     // [2SP][LDW]    : W1 rows (state ; agg)
    float *W2s = Ws + Cfg::WROWS * LDW;                                  // L2: [SP][LDW] second-layer kernel, then b2 [SP]
    int *fill = reinterpret_cast<int *>(W2s + Cfg::W2F);                 // [NS] gather-wave deposits so far
    int *freed = fill + NS;                                              // [NS] tiles consumed so far

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) F4_STAMP(0);
    const int S = a.S;
    int ty = 0;
    while (ty + 1 < a.n_types && (int)blockIdx.x >= a.blk_begin[ty + 1]) ++ty;
    const FusedType tp = a.tp[ty];
    const int O2 = tp.out2 > 0 ? tp.out2 : S;        // width of the second Dense's output (a deeper network's hidden layer, or the state)
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

    const __amdgpu_buffer_rsrc_t r_C = buf_rsrc(XC ? a.Xc : a.C), r_rows = buf_rsrc(tp.rows), r_state = buf_rsrc(a.state_in),
                                 r_rowptr = buf_rsrc(a.rowptr), r_src = buf_rsrc(a.src), r_w = buf_rsrc(HAS_W ? a.w : nullptr),
                                 r_scale = buf_rsrc(a.row_scale), r_init = buf_rsrc(INIT ? a.agg_init : nullptr);
    const bool has_scale = a.row_scale != nullptr;
    char *__restrict__ obase = reinterpret_cast<char *>(a.state_out);
    int any = 0;

    // ---- gather waves: the first job's CSR row (node id, row pointers, first 16 source ids) --------------------------------
    const int p = wave - Cfg::NCONS;
    const int qr = lane / LPR;                 // row of this lane group inside the wave's deposit
    const int l4 = lane % LPR;                 // 16-B column chunk of the row owned by this lane
    const int njobs = T * Cfg::PPT;
    auto job_m = [&](int n) -> int {            // global row number of this lane group's node in job n (-1: none)
        const int m = (t_first + (n / Cfg::PPT) * blk_per_xcd) * 16 + (n % Cfg::PPT) * Cfg::RPWV + qr;
        return (n < njobs && m < count) ? m : -1;
    };
    auto node_of = [&](int m) -> int {
        const int jr = buf_ld_i32(r_rows, m >= 0 ? 4u * (unsigned)m : BUF_OFF);
        return m >= 0 ? (rows ? jr : m) : -1;
    };
    int jA = -1, jB = -1, begA = 0, endA = 0;
    int idsA[IPL]; float wsA[IPL];
#pragma unroll
    for (int u = 0; u < IPL; ++u) { idsA[u] = 0; wsA[u] = 0.0f; }
    if (wave >= Cfg::NCONS) {
        jA = node_of(job_m(p)); jB = node_of(job_m(p + Cfg::NPROD));
        if (HDR && a.hdr) {
            // experiment: the first job's row pointers and first source ids of every lane from a per-launch-geometry
            // header (written once per graph by this kernel itself, below): ONE load instead of the dependent pair rowptr -> src
            const u32x4 h = __builtin_amdgcn_raw_buffer_load_b128(buf_rsrc(a.hdr), (int)((((unsigned)blockIdx.x * Cfg::NPROD + (unsigned)p) * 64u + (unsigned)lane) * 16u), 0, 0);
            begA = (int)h[0]; endA = (int)h[1]; idsA[0] = (int)h[2];
            if (IPL > 1) idsA[IPL > 1 ? 1 : 0] = (int)h[3];
        } else {
        begA = buf_ld_i32(r_rowptr, jA >= 0 ? 4u * (unsigned)jA : BUF_OFF);
        endA = buf_ld_i32(r_rowptr, jA >= 0 ? 4u * (unsigned)jA + 4u : BUF_OFF);
#pragma unroll
        for (int u = 0; u < IPL; ++u) {
            const int e = begA + u * LPR + l4;
            idsA[u] = buf_ld_i32(r_src, e < endA ? 4u * (unsigned)e : BUF_OFF);
            wsA[u] = HAS_W ? buf_ld_f32(r_w, e < endA ? 4u * (unsigned)e : BUF_OFF) : 0.0f;
        }
        }
        if (HDR && a.hdr_write) {       // the header-building launch: every gather lane leaves its first job's record and the launch ends
            u32x4 h = {(unsigned)begA, (unsigned)endA, (unsigned)idsA[0], (unsigned)idsA[IPL > 1 ? 1 : 0]};
            __builtin_amdgcn_raw_buffer_store_b128(h, buf_rsrc(a.hdr_write), (int)((((unsigned)blockIdx.x * Cfg::NPROD + (unsigned)p) * 64u + (unsigned)lane) * 16u), 0, 0);
        }
    }
    if (HDR && a.hdr_write) return;

    // rows 0 .. SP-1: state ; SP .. 2SP-1: agg ; (XC) 2SP .. 2SP+31: the constant inputs' folded weights, the bias row, zeros
    // (Round 3 tried leaving this fill to the matrix waves alone, the gather waves going straight to their first job after a bare
    // barrier: scripts/f4_timeline.py shows the fill's barrier 5 us - up to 10 - after entry on every workgroup of a 60 us C3 launch.
    // SLOWER: 256 threads fill 40 KB in ten dependent trips, the ring runs full before the first tile is consumed: C3 60.2 -> 63.4 us,
    // C4 460 -> 470 us.  All 1024 threads fill, then one barrier.)
    if (S == SP && tp.H == SP && ((reinterpret_cast<uintptr_t>(tp.Wf) | (XC ? reinterpret_cast<uintptr_t>(tp.Wc) : 0)) & 15) == 0) {
        // Full-width state and first layer (C3 / C4 / C5): whole 16-byte pieces of weight rows, every load of the fill issued before the
        // first LDS store.  The scalar loop below made ten dependent trips of this fill: 5 us (up to 10) of a 60 us C3 launch on every
        // workgroup (profiles/r03_c3_timeline.txt).  (The swizzle flips bit 4 of the column: 4-column pieces stay whole.)
        constexpr int N4 = Cfg::WROWS * SP / 4, NV = (N4 + NT - 1) / NT;
        f32x4 v[NV];
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int i4 = min(tid + u * NT, N4 - 1), k = i4 / (SP / 4), n = (i4 % (SP / 4)) * 4;
            const float *srow = (XC && k >= 2 * SP) ? tp.Wc + (size_t)(k - 2 * SP) * SP
                                                    : tp.Wf + (size_t)((k < SP ? tp.wrow_state + k : tp.wrow_agg + (k - SP))) * SP;
            v[u] = *reinterpret_cast<const f32x4 *>(srow + n);
        }
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int i4 = tid + u * NT, k = i4 / (SP / 4), n = (i4 % (SP / 4)) * 4;
            if (i4 < N4) *reinterpret_cast<f32x4 *>(Ws + k * LDW + (Cfg::SWZ ? (n ^ ((k & 1) << 4)) : n)) = v[u];
        }
    } else
    for (int i = tid; i < Cfg::WROWS * SP; i += NT) {
        const int k = i / SP, n = i % SP;
        float v = 0.0f;
        if (XC && k >= 2 * SP) {
            if (n < tp.H) v = tp.Wc[(size_t)(k - 2 * SP) * tp.H + n];
        } else {
            const int kk = k < SP ? k : k - SP;
            if (kk < S && n < tp.H) v = tp.Wf[(size_t)((k < SP ? tp.wrow_state : tp.wrow_agg) + kk) * tp.H + n];
        }
        Ws[k * LDW + (Cfg::SWZ ? (n ^ ((k & 1) << 4)) : n)] = v;
    }
    if (L2) {                                   // second Dense: rows k < H (hidden units), columns n < S, same swizzle
        for (int i = tid; i < SP * SP; i += NT) {
            const int k = i / SP, n = i % SP;
            W2s[k * LDW + (Cfg::SWZ ? (n ^ ((k & 1) << 4)) : n)] = (k < tp.H && n < O2) ? tp.W2[(size_t)k * O2 + n] : 0.0f;
        }
        if (tid < SP) W2s[SP * LDW + tid] = tid < O2 ? tp.b2[tid] : 0.0f;
    }
    if (tid < 2 * NS) fill[tid] = 0;
    __syncthreads();
    if (!open) return;                         // uniform across the launch; nothing has left the CU yet
    if (tid == 0) F4_STAMP(1);

    if (wave >= Cfg::NCONS) {
        // ================================ gather waves ================================================================
#ifdef GNN_F4_PROFILE
        const unsigned long long tr_ = f4_now();
#endif
        // jobs (tile, part) are dealt round-robin; tile t lives in ring slot t % NS, row = part * RPWV + qr.
        // Every load is a raw buffer load (predicated off = out of range = 0): no branches around memory operations, so
        // the waits hipcc inserts are exact counts.  The dependent chain  node id -> row pointers -> source ids -> rows
        // is cut by fetching the NEXT job's row pointers and first 16 source ids while this job's rows are in flight.
        for (int n = p; n < njobs; n += Cfg::NPROD) {
            const int t = n / Cfg::PPT;
            const int row = (n % Cfg::PPT) * Cfg::RPWV + qr;
            const int j = jA;
#ifdef GNN_F4_PROFILE
            unsigned long long tg_ = f4_now();
#endif
            // next job: row pointers now, node id of the job after it
            const int begB = buf_ld_i32(r_rowptr, jB >= 0 ? 4u * (unsigned)jB : BUF_OFF);
            const int endB = buf_ld_i32(r_rowptr, jB >= 0 ? 4u * (unsigned)jB + 4u : BUF_OFF);
            const int jC = node_of(job_m(n + 2 * Cfg::NPROD));
            const float scl = buf_ld_f32(r_scale, j >= 0 ? 4u * (unsigned)j : BUF_OFF);
            f32x4 own[VPL], acc[VPL];
#pragma unroll
            for (int x = 0; x < VPL; ++x) {
                own[x] = buf_ld_f32x4(r_state, j >= 0 ? (unsigned)(a.row_base + j) * (unsigned)(SP * 4) + 16u * (unsigned)(VPL * l4 + x) : BUF_OFF);
                acc[x] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (INIT) acc[x] = buf_ld_f32x4(r_init, j >= 0 ? (unsigned)j * (unsigned)(SP * 4) + 16u * (unsigned)(VPL * l4 + x) : BUF_OFF);   // sum of the arcs walked earlier
            }
            int idsB[IPL]; float wsB[IPL];
            int rem = endA - begA, eb = begA;
            int idc[IPL]; float wsc[IPL];
#pragma unroll
            for (int u = 0; u < IPL; ++u) { idc[u] = idsA[u]; wsc[u] = wsA[u]; }
            bool first = true;
#pragma unroll 1
            while (true) {
#pragma unroll
                for (int s0 = 0; s0 < 16; s0 += DEPTH) {          // DEPTH rows in flight, summed in ascending-source order
                    if (s0 > 0 && !__any(s0 < rem)) break;
                    f32x4 v[DEPTH][VPL];
#pragma unroll
                    for (int i = 0; i < DEPTH; ++i) {
                        const unsigned sid = (unsigned)__shfl(idc[(s0 + i) / LPR], (s0 + i) % LPR, LPR);
#pragma unroll
                        for (int x = 0; x < VPL; ++x)
                            v[i][x] = buf_ld_f32x4(r_state, s0 + i < rem ? sid * (unsigned)(SP * 4) + 16u * (unsigned)(VPL * l4 + x) : BUF_OFF);
                    }
                    if (s0 == 0 && first) {        // the next job's row pointers have landed by now: fetch its first 16 source ids
#pragma unroll
                        for (int u = 0; u < IPL; ++u) {
                            const int e = begB + u * LPR + l4;
                            idsB[u] = buf_ld_i32(r_src, e < endB ? 4u * (unsigned)e : BUF_OFF);
                            wsB[u] = HAS_W ? buf_ld_f32(r_w, e < endB ? 4u * (unsigned)e : BUF_OFF) : 0.0f;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < DEPTH; ++i) {
                        const float wv = HAS_W ? __shfl(wsc[(s0 + i) / LPR], (s0 + i) % LPR, LPR) : 1.0f;
#pragma unroll
                        for (int x = 0; x < VPL; ++x) {
                            if (HAS_W) acc[x] += wv * v[i][x];
                            else acc[x] += v[i][x];
                        }
                    }
                }
                first = false;
                rem -= 16; eb += 16;
                if (!__any(rem > 0)) break;
#pragma unroll
                for (int u = 0; u < IPL; ++u) {     // in-degree > 16: the next 16 source ids, one coalesced load per lane group
                    const int e = eb + u * LPR + l4;
                    idc[u] = buf_ld_i32(r_src, e < endA ? 4u * (unsigned)e : BUF_OFF);
                    wsc[u] = HAS_W ? buf_ld_f32(r_w, e < endA ? 4u * (unsigned)e : BUF_OFF) : 0.0f;
                }
            }
            if (has_scale) {
#pragma unroll
                for (int x = 0; x < VPL; ++x) acc[x] *= scl;
            }
            // rotate: A <- B <- C
            jA = jB; begA = begB; endA = endB; jB = jC;
#pragma unroll
            for (int u = 0; u < IPL; ++u) { idsA[u] = idsB[u]; wsA[u] = wsB[u]; }

            const int s = t % NS, round = t / NS;
#ifdef GNN_F4_PROFILE
            if (lane == 0) atomicAdd(&g_f4_prof[0], f4_now() - tg_);
            tg_ = f4_now();
#endif
            {
                int spin = 0;
                while (__builtin_amdgcn_readfirstlane(f4_ld_acquire(&freed[s])) < round) {
                    if (spin >= SPIN_MAX) { bad = 1; break; }
                    ++spin; __builtin_amdgcn_s_sleep(1);
                }
            }
            if (bad) break;        // the slot never came free: deposit nothing (the state is invalid anyway, k < 0 says so)
#ifdef GNN_F4_PROFILE
            if (lane == 0) atomicAdd(&g_f4_prof[1], f4_now() - tg_);
#endif
#pragma unroll
            for (int x = 0; x < VPL; ++x) {
                float *xr = Xs + s * Cfg::SLOT + row * LDX + 4 * (VPL * l4 + x);     // rows are 8-B aligned: two b64 stores each
                *reinterpret_cast<float2 *>(xr) = make_float2(own[x][0], own[x][1]);
                *reinterpret_cast<float2 *>(xr + 2) = make_float2(own[x][2], own[x][3]);
                *reinterpret_cast<float2 *>(xr + SP) = make_float2(acc[x][0], acc[x][1]);
                *reinterpret_cast<float2 *>(xr + SP + 2) = make_float2(acc[x][2], acc[x][3]);
            }
            if (l4 == 0) Xs[s * Cfg::SLOT + row * LDX + 2 * SP] = __int_as_float(j);      // the row's pad words carry its node id
            if (lane == 0) __hip_atomic_fetch_add(&fill[s], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (n == p && p == 0 && lane == 0) F4_STAMP(2);
        }
#ifdef GNN_F4_PROFILE
        if (lane == 0) atomicAdd(&g_f4_prof[6], f4_now() - tr_);
#endif
    } else {
        // ================================ matrix waves ================================================================
        // MFMA fragments: A row / B-C column = lane & 15, k / C row group = lane >> 4.  The epilogue runs row-major instead
        // (row 4*i + g, columns 4r .. 4r+3: whole 256-B rows per 16 lanes) on values passed through the slot's agg half.
        const int r = lane & 15, g = lane >> 4;
#ifdef GNN_F4_PROFILE
        const unsigned long long tr_ = f4_now();
#endif
        for (int t = wave; t < T; t += Cfg::NCONS) {
            // accumulators start from the per-node constant: D = [state|agg].W1 + C  (col = 16*ci + r, row = 4*g + reg).
            // Raw buffer loads (predicated off = out of range = 0): branch-free, all 4*NCT in flight at once, issued
            // BEFORE the slot is ready so they land while the gather waves fill it.
            f32x4 c[Cfg::NCT];
            float xa[XC ? XC_K / 4 : 1];
            if (XC) {
                // XC: the accumulators start from zero and the node's 32 constant inputs arrive as MFMA A fragments (row r of
                // the tile, columns 4q + g: one 128-byte line per node instead of C's two) - 8 loads per lane instead of 16
                const int m = (t_first + t * blk_per_xcd) * 16 + r;
                const int jr = buf_ld_i32(r_rows, m < count ? 4u * (unsigned)m : BUF_OFF);
                const int jx = m < count ? (rows ? jr : m) : -1;
#pragma unroll
                for (int q = 0; q < XC_K / 4; ++q)
                    xa[q] = buf_ld_f32(r_C, jx >= 0 ? ((unsigned)jx * (unsigned)XC_K + (unsigned)(4 * q + g)) * 4u : BUF_OFF);
#pragma unroll
                for (int ci = 0; ci < Cfg::NCT; ++ci) c[ci] = (f32x4){0.f, 0.f, 0.f, 0.f};
            } else {
                int jrow[4];
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int m = (t_first + t * blk_per_xcd) * 16 + 4 * g + reg;
                    const int jr = buf_ld_i32(r_rows, m < count ? 4u * (unsigned)m : BUF_OFF);
                    jrow[reg] = m < count ? (rows ? jr : m) : -1;
                }
#pragma unroll
                for (int ci = 0; ci < Cfg::NCT; ++ci) {
                    const int col = 16 * ci + r;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg)
                        c[ci][reg] = buf_ld_f32(r_C, (jrow[reg] >= 0 && col < tp.H) ? ((unsigned)jrow[reg] * (unsigned)a.ldC + (unsigned)col) * 4u : BUF_OFF);
                }
            }
            const int s = t % NS, round = t / NS;
#ifdef GNN_F4_PROFILE
            unsigned long long tc_ = f4_now();
#endif
            {
                int spin = 0;
                while (__builtin_amdgcn_readfirstlane(f4_ld_acquire(&fill[s])) < Cfg::PPT * (round + 1)) {
                    if (spin >= SPIN_MAX) { bad = 1; break; }
                    ++spin; __builtin_amdgcn_s_sleep(1);
                }
            }
            if (bad) break;        // the slot never filled: its node ids are not valid, store nothing
#ifdef GNN_F4_PROFILE
            if (lane == 0) { atomicAdd(&g_f4_prof[2], f4_now() - tc_); atomicAdd(&g_f4_prof[5], 1ull); }
            tc_ = f4_now();
#endif
            float *X = Xs + s * Cfg::SLOT;
            const float *xrow = X + r * LDX + g;
#pragma unroll 2
            for (int s4 = 0; s4 < 2 * SP / 4; ++s4) {
                const float av = xrow[4 * s4];
                const int k = 4 * s4 + g;
#pragma unroll
                for (int ci = 0; ci < Cfg::NCT; ++ci) {
                    const int n = 16 * ci + r;
                    const float bv = Ws[k * LDW + (Cfg::SWZ ? (n ^ ((k & 1) << 4)) : n)];
                    c[ci] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c[ci], 0, 0, 0);
                }
            }
            if (XC) {
                // the constant inputs' share of the product (one LDS base per column tile, constant offsets per k step: 2SP + 4q
                // is even, so the swizzle bit is g & 1); their loads were issued before the slot wait, as C's are
                const float *wb[Cfg::NCT];
#pragma unroll
                for (int ci = 0; ci < Cfg::NCT; ++ci)
                    wb[ci] = Ws + (2 * SP + g) * LDW + (Cfg::SWZ ? ((16 * ci + r) ^ ((g & 1) << 4)) : 16 * ci + r);
#pragma unroll
                for (int q = 0; q < XC_K / 4; ++q) {
#pragma unroll
                    for (int ci = 0; ci < Cfg::NCT; ++ci)
                        c[ci] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[q], wb[ci][4 * q * LDW], c[ci], 0, 0, 0);
                }
            }
#ifdef GNN_F4_PROFILE
            { const unsigned long long tm_ = f4_now(); if (lane == 0) atomicAdd(&g_f4_prof[3], tm_ - tc_); tc_ = tm_; }
#endif
            // accumulator layout -> row-major through the agg half of the slot (dead after the loop; same-wave LDS traffic
            // is ordered, and no other wave touches the slot before `freed` moves)
#pragma unroll
            for (int ci = 0; ci < Cfg::NCT; ++ci)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) X[(4 * g + reg) * LDX + SP + 16 * ci + r] = c[ci][reg];
            if (L2) {
                // hidden layer: activation in place (row-major, 4 trips), then h . W2 on the matrix cores with the A
                // fragments read from the same agg half, and the result stashed over it again
#pragma unroll 1
                for (int i = 0; i < 4; ++i) {
                    float *ph = X + (4 * i + g) * LDX + SP + 4 * r;
                    if (4 * r < SP) {
                        const float2 lo = *reinterpret_cast<const float2 *>(ph), hi = *reinterpret_cast<const float2 *>(ph + 2);
                        f32x4 hv = {lo.x, lo.y, hi.x, hi.y};
                        activate4(tp.act, hv);
#pragma unroll
                        for (int e = 0; e < 4; ++e) hv[e] = 4 * r + e < tp.H ? hv[e] : 0.0f;
                        *reinterpret_cast<float2 *>(ph) = make_float2(hv[0], hv[1]);
                        *reinterpret_cast<float2 *>(ph + 2) = make_float2(hv[2], hv[3]);
                    }
                }
#pragma unroll
                for (int ci = 0; ci < Cfg::NCT; ++ci) {
                    const float b = W2s[SP * LDW + 16 * ci + r];
                    c[ci] = (f32x4){b, b, b, b};
                }
                const float *hrow = X + r * LDX + SP + g;
#pragma unroll 2
                for (int s4 = 0; s4 < SP / 4; ++s4) {
                    const float av = hrow[4 * s4];
                    const int k = 4 * s4 + g;
#pragma unroll
                    for (int ci = 0; ci < Cfg::NCT; ++ci) {
                        const int n = 16 * ci + r;
                        const float bv = W2s[k * LDW + (Cfg::SWZ ? (n ^ ((k & 1) << 4)) : n)];
                        c[ci] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c[ci], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int ci = 0; ci < Cfg::NCT; ++ci)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) X[(4 * g + reg) * LDX + SP + 16 * ci + r] = c[ci][reg];
            }
            const int act_out = L2 ? tp.act2 : tp.act;
            const int W_out = L2 ? O2 : S;                          // real columns of the new state
#pragma unroll 1
            for (int i = 0; i < 4; ++i) {
                const float *px = X + (4 * i + g) * LDX;
                const int j = __float_as_int(px[2 * SP]);              // node id left by the gather wave (-1 = pad row)
                float d2 = 0.0f, n2 = 0.0f;
                f32x4 nv = {0.f, 0.f, 0.f, 0.f}, ov = {0.f, 0.f, 0.f, 0.f};
                if (4 * r < SP) {
                    const float2 plo = *reinterpret_cast<const float2 *>(px + SP + 4 * r), phi = *reinterpret_cast<const float2 *>(px + SP + 4 * r + 2);
                    const float2 olo = *reinterpret_cast<const float2 *>(px + 4 * r), ohi = *reinterpret_cast<const float2 *>(px + 4 * r + 2);
                    nv = (f32x4){plo.x, plo.y, phi.x, phi.y};
                    ov = (f32x4){olo.x, olo.y, ohi.x, ohi.y};
                }
                if (act_out == GNN_ACT_SOFTMAX) {
                    // a softmax state (any Keras activation is legal there, reference MLP.py:12-78): the row lives in the 16 lanes
                    // of this lane group - maximum, exponentials and their sum across them, pad columns left out
                    float m = -3.0e38f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) m = (4 * r + e < W_out && 4 * r < SP) ? fmaxf(m, nv[e]) : m;
#pragma unroll
                    for (int off = 8; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 16));
                    float ssum = 0.0f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        nv[e] = (4 * r + e < W_out && 4 * r < SP) ? expf(nv[e] - m) : 0.0f;
                        ssum += nv[e];
                    }
#pragma unroll
                    for (int off = 8; off >= 1; off >>= 1) ssum += __shfl_xor(ssum, off, 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) nv[e] = nv[e] / ssum;
                } else {
                    activate4(act_out, nv);
                }
                if (4 * r < SP) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        nv[e] = (j >= 0 && 4 * r + e < W_out) ? nv[e] : 0.0f;
                        const float d = nv[e] - ov[e];
                        d2 = fmaf(d, d, d2);
                        n2 = fmaf(ov[e], ov[e], n2);
                    }
                    if (j >= 0) {
                        const unsigned ooff = (unsigned)(a.row_base + j) * (unsigned)(SP * 4) + 16u * r;
                        *reinterpret_cast<f32x4 *>(obase + ooff) = nv;
                        if constexpr (PEERS) {              // ... and into every peer's full buffer, same offset (static indices: no scratch copy of the table)
#pragma unroll
                            for (int pi = 0; pi < GNN_MAX_PEERS; ++pi)
                                if (pi < a.n_peers) *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(a.peer_out[pi]) + ooff) = nv;
                        }
                    }
                }
#pragma unroll
                for (int off = 8; off >= 1; off >>= 1) {
                    d2 += __shfl_xor(d2, off, 16);
                    n2 += __shfl_xor(n2, off, 16);
                }
                if (j >= 0 && sqrtf(d2) > a.thr * sqrtf(n2)) any = 1;
            }
#ifdef GNN_F4_PROFILE
            if (lane == 0) atomicAdd(&g_f4_prof[4], f4_now() - tc_);
#endif
            if (lane == 0) __hip_atomic_store(&freed[s], round + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#ifdef GNN_F4_PROFILE
        if (lane == 0) atomicAdd(&g_f4_prof[7], f4_now() - tr_);
#endif
    }

    any = __syncthreads_or(any);
    bad = __syncthreads_or(bad);                 // (a predicate reduction, not a bitwise OR: one call per word)
    if (tid == 0) F4_STAMP(3);
    if (tid == 0) {
        if (any && a.flag_next) atomicOr(a.flag_next, 1);
        if (bad && a.err) atomicOr(a.err, 1);
        if (blockIdx.x == 0 && a.k_out) *a.k_out = a.k_val;
    }
}

template <int SP, bool HAS_W, int DEPTH, bool L2 = false, int NC = 4, bool INIT = false, bool XC = false, int VPL = 1, bool HDR = false, bool PEERS = false>
int launch_fused4_one(Fused2Args &fa, int n_cu, hipStream_t st) {
    using Cfg = Fused4Cfg<SP, NC, L2, XC, VPL>;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void *)k_state_fused4<SP, HAS_W, DEPTH, NC, L2, INIT, XC, VPL, HDR, PEERS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)Cfg::LDS_BYTES) != hipSuccess) return 1;
        attr = true;
    }
    // 2 workgroups (32 waves) per CU are resident at once.  (More, smaller workgroups - the dispatcher hands the next one to whichever slot
    // frees first - levelled the +-10 % spread of workgroup run times but paid one more ramp-up each: measured, not kept; docs/rounds/.)
    const int budget = 2 * n_cu;
    long total_tiles = 0;
    for (int t = 0; t < fa.n_types; ++t) total_tiles += (fa.tp[t].count + 15) / 16;
    fa.blk_begin[0] = 0;
    for (int t = 0; t < fa.n_types; ++t) {
        const int ntiles = (fa.tp[t].count + 15) / 16;
        int nb = 0;
        if (ntiles > 0) {
            // a workgroup wants >= 4 tiles (one per matrix wave, 16 gather jobs); never more workgroups than that allows
            nb = (int)std::min<long>((ntiles + 3) / 4, std::max<long>(8, budget * (long)ntiles / std::max<long>(total_tiles, 1)));
            nb = std::max(8, nb / 8 * 8);          // multiples of 8 (one per XCD), rounded DOWN: the grid stays co-resident
        }
        fa.blk_begin[t + 1] = fa.blk_begin[t] + nb;
    }
    const int grid = fa.blk_begin[fa.n_types];
    if (grid == 0) return 0;
    if (PEERS)
        GNN_SET_KERNEL_NAME("k_state_fused4<%d,%s,%d,%d,%s,%s,%s,%d,%s,true> (peer stores)", SP, HAS_W ? "true" : "false", DEPTH, NC, L2 ? "true" : "false", INIT ? "true" : "false", XC ? "true" : "false", VPL, HDR ? "true" : "false");
    else if (VPL == 1 && !HDR)
        GNN_SET_KERNEL_NAME("k_state_fused4<%d,%s,%d,%d,%s,%s,%s>", SP, HAS_W ? "true" : "false", DEPTH, NC, L2 ? "true" : "false", INIT ? "true" : "false", XC ? "true" : "false");
    else
        GNN_SET_KERNEL_NAME("k_state_fused4<%d,%s,%d,%d,%s,%s,%s,%d,%s>", SP, HAS_W ? "true" : "false", DEPTH, NC, L2 ? "true" : "false", INIT ? "true" : "false", XC ? "true" : "false", VPL, HDR ? "true" : "false");
    k_state_fused4<SP, HAS_W, DEPTH, NC, L2, INIT, XC, VPL, HDR, PEERS><<<grid, Cfg::NT, Cfg::LDS_BYTES, st>>>(fa);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

inline int launch_fused4(Fused2Args &fa, int SP, int n_cu, hipStream_t st) {
    if (fa.n_peers > 0) {          // the exchange in the epilogue: homogeneous one-layer shards without per-arc weights, widths 17 .. 64
        if (fa.w || fa.tp[0].W2 || (SP != 32 && SP != 64)) return 1;
#define F4_PEER(SPV)                                                                                                               \
        if (SP == SPV) {                                                                                                           \
            if (fa.agg_init) return fa.Xc ? launch_fused4_one<SPV, false, 4, false, 4, true, true, 1, false, true>(fa, n_cu, st)   \
                                          : launch_fused4_one<SPV, false, 4, false, 4, true, false, 1, false, true>(fa, n_cu, st); \
            return fa.Xc ? launch_fused4_one<SPV, false, 4, false, 4, false, true, 1, false, true>(fa, n_cu, st)                   \
                         : launch_fused4_one<SPV, false, 4, false, 4, false, false, 1, false, true>(fa, n_cu, st);                 \
        }
        F4_PEER(32)
        F4_PEER(64)
#undef F4_PEER
        return 1;
    }
#define F4_CASE(SPV)                                                                                                  \
    case SPV:                                                                                                         \
        if (fa.agg_init) {                                                                                            \
            if (fa.tp[0].W2) return 1;                                                                                \
            if (fa.w) return launch_fused4_one<SPV, true, 4, false, 4, true>(fa, n_cu, st);                            \
            return fa.Xc ? launch_fused4_one<SPV, false, 4, false, 4, true, true>(fa, n_cu, st) : launch_fused4_one<SPV, false, 4, false, 4, true>(fa, n_cu, st); \
        }                                                                                                             \
        if (fa.tp[0].W2) return fa.w ? launch_fused4_one<SPV, true, 4, true>(fa, n_cu, st) : launch_fused4_one<SPV, false, 4, true>(fa, n_cu, st); \
        if (fa.w) return launch_fused4_one<SPV, true, 4>(fa, n_cu, st);                                                \
        if (fa.Xc) return launch_fused4_one<SPV, false, 4, false, 4, false, true>(fa, n_cu, st);                       \
        return launch_fused4_one<SPV, false, 4>(fa, n_cu, st);
    switch (SP) {
        F4_CASE(16)
        F4_CASE(32)
        F4_CASE(64)
        default: return 1;
    }
#undef F4_CASE
}

}  // namespace gnn
