// Fused state-transition iteration for state widths 129 .. 256 (one Dense layer, homogeneous graphs): the reference's
// `convergence` + the `condition` of the next iteration (GNN/Models/GNN.py:217-236, :196-214) in one launch, as
// k_state_fused4 / k_state_wide do for narrower states.  Un-fused these widths run at 34 % of the HBM roofline (k_aggregate_vec writes the
// neighbour sums, k_rowdense_wide reads them back: 1.15 ms per iteration at d = 200 on 300 k nodes / 3 M arcs).
//
// W1 = [2 d x d] floats is 320 KB at d = 200 - it cannot live in the CU's 160 KB of LDS.  So the roles of LDS and L2 are swapped against
// the narrower kernels:
//   * the gathered rows [state | neighbour sum] of a 32-node tile live in LDS, filled by the 16 - NCB waves that are not matrix waves
//     (8 .. 11) - one wave per row, a lane per 16-byte chunk, rows drawn from a ticket counter in LDS, ONE rolling window of 16
//     neighbour rows + the own row in flight per wave, the row pointers fetched three rows and the source ids two rows ahead (this memory
//     system rewards the NUMBER of waves with a gather outstanding, profiles/r01_gather_sweep.txt: every wave that is not a matrix wave gathers);
//   * the weights stream from L2 as matrix operands: a set-up kernel lays them out once per call in fragment order (k_xwide_weights_q),
//     whole 1-KB lines per wave instruction; every tile re-reads the matrix (it stays resident in each XCD's 4 MB L2);
//   * NCB <= 8 matrix waves, one per 32-column block of the output.  Operands are swapped (weights = A, rows = B) so that a lane ends up
//     with columns 8 q + 4 (lane / 32) + 0..3 of ITS row: C, the old state and the new state all move as 16-byte pieces;
//   * the convergence predicate needs whole rows: every matrix wave leaves its block's share of |new - old|^2 and |old|^2 per row in LDS,
//     the wave that finishes the tile last adds the shares in block order (deterministic) and tests the rows.
// Rounds 3 - 5 ran the Dense layer on v_mfma_f32_32x32x2_f32 (two 32-row slots of f32 rows; 683 - 707 us at d = 200 = 0.55 - 0.57 of the
// roofline, a matrix wave's K loop at the pipe's whole time: git history, profiles/r05h_d200_*, profiles/r06_d200_f32_kernel_stats.csv).
// Round 6 (this file): the Dense layer on the bf16 matrix cores, every f32 operand split into three bf16 terms (bf16_split.hpp: the accuracy
// of an f32 product chain, not its bits).
//
// Why: the f32 form spends 25 600 matrix cycles per SIMD on every 32-row tile at 256 columns (v_mfma_f32_32x32x2_f32: 64 cycles for 2 k)
// next to a gather that takes ~ 33 000 cycles a tile by itself; six v_mfma_f32_32x32x16_bf16 (32 cycles each) contract 16 k: 9 600.
// What that changed against the f32 form:
//   * the gather waves SPLIT the rows they deposit (own state and neighbour sum: 36 VALU instructions a row and wave, once per row - not
//     once per matrix wave that reads it): a row lies in LDS as three bf16 planes [hi | mid | lo] of 2 KH values, 12 KH + 16 bytes;
//   * that is 1.5 x the f32 row, so two 32-row slots no longer fit (KH = 256: 197 KB): the slots become a RING of HS = 3 .. 4 half-slots
//     of 16 rows; tile i of the workgroup lies in half-slots (2 i) % HS and (2 i + 1) % HS, every half-slot is handed over by its own
//     pair of monotonic counters (rows deposited / times freed);
//   * the weights stay f32 in L2, in the fragment order of the 16-k instruction (k_xwide_weights_q, once per call: 32 bytes per lane
//     and k-step as before), and the matrix wave splits each piece in registers (36 VALU instructions a k-step; a piece is read by
//     one wave only, so nothing is split twice).  Three bf16 planes in L2 save that VALU work and cost 1.5 x the stream: measured equal at
//     d = 200 (638 / 636 us), slower at 160, faster at 256 by 2 % - not kept;
//   * the old state for the convergence predicate is rebuilt from the planes (hi + mid + lo is the f32 value exactly).
// Same hand-overs otherwise: ticket counter for the rows, monotonic LDS counters, bounded spins that raise the sticky error word.
//
// Measured (d = 200, 300 k nodes / 3 M arcs, same box, f32 form -> this): 685 - 697 -> 631 - 648 us per iteration (0.565 -> 0.61 of
// 8 TB/s on the algorithmic bytes); d = 160 on 1 M / 10 M 1 788 -> 1 687; d = 256 on 200 k / 2 M 619 -> 591.  What still bounds it
// (-DXB_EXPERIMENT ablations, scripts/dev/xwide_b3_ablate.py; -DXW_PROFILE phase clocks, scripts/dev/xb_prof.py):
//   * matrix waves that only hand the rows back: 438 - 459 us for 2.65 GB = 6.0 TB/s - the gather by itself;
//   * everything but the weight loads: 530 - 570; everything but the products: 541 - 605; neither: 525 - 551;
//   * a matrix wave's K loop takes ~ 1 000 clocks a k-step where its instructions need 364: a CU's loads return in issue order, so a
//     weight piece that hits in L2 comes back behind the gather's misses issued before it (~ 3 us), and XB_PD = 6 k-steps in flight
//     is what 128 registers hold next to the accumulators.  Chain per tile (rows ready -> K loop -> constant -> epilogue) ~ 34 000
//     clocks against ~ 33 000 the gather needs to fill one: both sides wait for each other a quarter of the time.
// Tried on the way, all measured on one box against 638 - 648: the per-node constant requested a block ahead or a round ahead (656,
// 675: its 128 scattered line requests then sit IN FRONT of the weight pieces in the CU's queue), two register sets for the rows (644),
// a raised wave priority for the matrix waves (no change), 3 / 4 / 6 matrix waves with two blocks each and more gather waves (680 - 840:
// the chain per wave doubles).  What would move it: 64-row tiles (half the weight stream and half the chain per row) - which needs the
// rows in LDS as f32 (4 bytes a value, split by every matrix wave that reads them) and fits widths up to ~ 208 only.
#pragma once
#include <hip/hip_runtime.h>
#include "kernel_state_fused4.hpp"
#include "buffer_ops.hpp"
#include "bf16_split.hpp"

namespace gnn {

struct XWideArgs {
    const int *gate; int n_gate, gate_stride;      // run iff OR of gate[i * gate_stride], i < n_gate, is non-zero
    const int *rowptr, *src;                       // CSR by destination
    const float *w, *row_scale;
    const float *state_in;                         // [n_src_rows, SP]
    float *state_out;                              // [N, SP]
    const float *C; int ldC;                       // per-node constant of the first layer (bias included)
    int N, S, SP;                                  // nodes, state width, leading dimension of the state buffers (multiple of 4)
    int KH, NCB;                                   // K per half (SP rounded up to 8), 32-column blocks
    const void *Wb;                                // first-layer weights, f32, in the fragment order of the 16-k matrix instruction (k_xwide_weights_q)
    int NKS, HS, NMW;                              // 16-k steps (= KH / 8), half-slots of the row ring, matrix waves
    int dbg;                                       // -DXB_EXPERIMENT builds: ablation bits (GNN_XB_DBG)
    int act;
    float thr;
    int *flag_next;
    float *k_out; float k_val;
    int *err;
};

constexpr int XW_NM = 8;          // matrix waves at most (wave ids 0 .. NCB - 1, block cb = wave id); the other waves gather
#ifndef GNN_F4_SPIN_MAX
#define GNN_F4_SPIN_MAX (1 << 22)
#endif
constexpr int XW_SPIN_MAX = GNN_F4_SPIN_MAX;   // (-DGNN_F4_SPIN_MAX=0: the debug build whose every wait expires at once, libgnnloop_spin0.so)

inline int xwide_kh(int SP) { return (SP + 7) & ~7; }
inline size_t xwide_weight_floats(int S, int SP) { return (size_t)((S + 31) / 32) * (xwide_kh(SP) / 4) * 256; }


// -DXW_PROFILE (experiment builds): shader-clock totals of the phases of matrix wave 0 and of the first gather wave of every workgroup
#ifdef XW_PROFILE
__device__ unsigned long long g_xw_prof[8];
__device__ __forceinline__ unsigned long long xw_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}
#define XW_T(var) const unsigned long long var = xw_now()
#define XW_ADD(i, expr) do { if (lane == 0) atomicAdd(&g_xw_prof[i], (unsigned long long)(expr)); } while (0)
#else
#define XW_T(var) do { } while (0)
#define XW_ADD(i, expr) do { } while (0)
#endif

__device__ __forceinline__ int xw_readlane_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float xw_readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }



#ifdef XB_EXPERIMENT
#define XB_DBG(a_) ((a_).dbg)
#else
#define XB_DBG(a_) 0
#endif
constexpr int XB_PD = 6;            // k-steps of weight pieces in flight per matrix wave (two 16-byte pieces per lane and k-step)
constexpr int XB_LDS_MAX = 160 * 1024 - 256;   // dynamic LDS a workgroup may ask for (the kernel has 256 bytes of static LDS: __syncthreads_or)
constexpr int XB_HS_MAX = 4;        // half-slots at most (more would let a matrix wave reach tile i + 2 while tile i is still being read)

inline int xb_row_bytes(int KH) { return 12 * KH + 16; }          // three planes of 2 KH bf16 + 16: an odd number of 16-byte pieces -> 16 rows, 64 banks
inline size_t xb_fixed_bytes() { return sizeof(float) * 2 * XW_NM * 64 + sizeof(int) * (2 * XB_HS_MAX + 2 + 1 + 1); }
inline int xb_half_slots(int KH) {
    const long budget = XB_LDS_MAX - (long)xb_fixed_bytes();
    return (int)std::min<long>(XB_HS_MAX, budget / (16L * xb_row_bytes(KH)));
}
inline size_t xb_lds_bytes(int KH, int HS) { return (size_t)HS * 16 * xb_row_bytes(KH) + xb_fixed_bytes(); }

// Wcat = [state rows (KH, zero padded) ; neighbour-sum rows (KH)] of the folded first layer Wf [in_dim x H], f32, in the order the matrix
// waves read it (two 16-byte pieces per lane and k-step):
// Wq[(((cb * NKS + ks) * 2 + half) * 64 + lane) * 4 + e] = Wcat[16 ks + 8 (lane / 32) + 4 half + e][32 cb + lane % 32]
__global__ void k_xwide_weights_q(const float *Wf, int H, int S, int wrow_state, int wrow_agg, int KH, int NKS, int NCB, float *Wq) {
    const long total = (long)NCB * NKS * 512;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 3), lane = (int)((i >> 2) & 63), half = (int)((i >> 8) & 1);
        const long rest = i >> 9;
        const int ks = (int)(rest % NKS), cb = (int)(rest / NKS);
        const int k = 16 * ks + 8 * (lane >> 5) + 4 * half + e, col = 32 * cb + (lane & 31);
        const int kk = k < KH ? k : k - KH;
        float v = 0.0f;
        if (kk < S && col < H) v = Wf[(size_t)((k < KH ? wrow_state : wrow_agg) + kk) * H + col];
        Wq[i] = v;
    }
}

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void xb_store_split(char *dst, int plane_bytes, const f32x4 &v) {
    unsigned h0, m0, l0, h1, m1, l1;
    split3_pk((f32x2){v[0], v[1]}, h0, m0, l0);
    split3_pk((f32x2){v[2], v[3]}, h1, m1, l1);
    *reinterpret_cast<u32x2 *>(dst) = (u32x2){h0, h1};
    *reinterpret_cast<u32x2 *>(dst + plane_bytes) = (u32x2){m0, m1};
    *reinterpret_cast<u32x2 *>(dst + 2 * plane_bytes) = (u32x2){l0, l1};
}
__device__ __forceinline__ f32x4 xb_load_joined(const char *src, int plane_bytes) {
    const u32x2 h = *reinterpret_cast<const u32x2 *>(src), m = *reinterpret_cast<const u32x2 *>(src + plane_bytes),
                l = *reinterpret_cast<const u32x2 *>(src + 2 * plane_bytes);
    f32x4 o;
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        o[2 * x] = (__uint_as_float(h[x] << 16) + __uint_as_float(m[x] << 16)) + __uint_as_float(l[x] << 16);
        o[2 * x + 1] = (__uint_as_float(h[x] & 0xFFFF0000u) + __uint_as_float(m[x] & 0xFFFF0000u)) + __uint_as_float(l[x] & 0xFFFF0000u);
    }
    return o;
}

template <bool HAS_W>
__global__ void __launch_bounds__(1024) k_state_xwide_b3(XWideArgs a) {
    int open = a.gate == nullptr;
    for (int i = 0; i < a.n_gate; ++i) open |= a.gate[(size_t)i * a.gate_stride] != 0;
    extern __shared__ __attribute__((aligned(16))) char xb_smem[];
    const int KH = a.KH, NKS = a.NKS, HS = a.HS, NMW = a.NMW, NCB = a.NCB;
    const int dbg = XB_DBG(a);
    const int RS = 12 * KH + 16, PL = 4 * KH;                           // bytes of a row; of a plane of a row
    char *ring = xb_smem;                                               // [HS][16] rows: [hi (2 KH bf16) | mid | lo | pad]
    float *part = reinterpret_cast<float *>(ring + (size_t)HS * 16 * RS);   // [2][NM][32][2] : per block and row |new - old|^2, |old|^2
    int *fill = reinterpret_cast<int *>(part + 2 * XW_NM * 64);         // [HS_MAX] rows deposited so far
    int *freed = fill + XB_HS_MAX;                                      // [HS_MAX] times handed back so far
    int *done = freed + XB_HS_MAX;                                      // [2] blocks finished so far (tiles of either parity)
    int *ticket = done + 2;                                             // next row (tile * 32 + row) to gather
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 2 * XB_HS_MAX + 3) fill[tid] = 0;
    __syncthreads();
    if (!open) return;                             // uniform across the launch; nothing has left the CU yet

    const int N = a.N, SP = a.SP, S = a.S;
    const int ntiles = (N + 31) >> 5;
    const int nT = (int)blockIdx.x < ntiles ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;   // tiles of this workgroup
    const __amdgpu_buffer_rsrc_t r_state = buf_rsrc(a.state_in), r_rowptr = buf_rsrc(a.rowptr), r_src = buf_rsrc(a.src),
                                 r_w = buf_rsrc(HAS_W ? a.w : nullptr), r_scale = buf_rsrc(a.row_scale), r_C = buf_rsrc(a.C),
                                 r_wb = buf_rsrc(a.Wb), r_out = buf_rsrc(a.state_out);
    int any = 0, bad = 0;
    XW_T(tk0_);

    if (wave >= NMW) {
        // ================================ gather waves ================================================================================
        // Rows are drawn from a ticket counter in LDS (ticket t = row t % 32 of this workgroup's tile t / 32; a wave's tickets grow, so no
        // row waits behind a later one).  A wave always holds four tickets: the row whose neighbour rows are being summed, the row whose
        // neighbour rows are being issued into the window slots the sum frees (ONE rolling window of 16 neighbour rows + the own row in
        // flight), the row whose source ids are in flight and the row whose row pointers are in flight.  The sum is split into three bf16
        // planes as it is deposited.
        const int CH = SP >> 2, CHZ = KH >> 2;                   // 16-byte chunks of a row; chunks of the padded half
        const bool act_l = lane < CH, zero_l = lane >= CH && lane < CHZ;
        const bool has_scale = a.row_scale != nullptr;
        const unsigned lane_off = 16u * (unsigned)lane;
        const int total = nT * 32;
        auto draw = [&]() -> int {
            int t = 0;
            if (lane == 0) t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return __builtin_amdgcn_readfirstlane(t);
        };
        auto node = [&](int t) -> int {                          // (wave-uniform)
            const long j = 32 * ((long)blockIdx.x + (long)(t >> 5) * gridDim.x) + (t & 31);
            return (t < total && j < N) ? (int)j : -1;
        };
        auto ld_rowptr = [&](int j) -> int {                     // even lanes: beg, odd lanes: end
            return buf_ld_i32(r_rowptr, j >= 0 ? 4u * (unsigned)j + 4u * (unsigned)(lane & 1) : BUF_OFF);
        };
        int t0 = draw(), t1 = draw(), t2 = draw(), t3 = draw();
        int rp0 = ld_rowptr(node(t0)), rp1 = ld_rowptr(node(t1)), rp2 = ld_rowptr(node(t2)), rp3 = ld_rowptr(node(t3));
        int beg0 = xw_readlane_i(rp0, 0), end0 = xw_readlane_i(rp0, 1);
        int beg1 = xw_readlane_i(rp1, 0), end1 = xw_readlane_i(rp1, 1);
        int id0 = buf_ld_i32(r_src, beg0 + lane < end0 ? 4u * (unsigned)(beg0 + lane) : BUF_OFF);       // up to 64 source ids of the row
        float w0 = HAS_W ? buf_ld_f32(r_w, beg0 + lane < end0 ? 4u * (unsigned)(beg0 + lane) : BUF_OFF) : 0.0f;
        int id1 = buf_ld_i32(r_src, beg1 + lane < end1 ? 4u * (unsigned)(beg1 + lane) : BUF_OFF);
        float w1 = HAS_W ? buf_ld_f32(r_w, beg1 + lane < end1 ? 4u * (unsigned)(beg1 + lane) : BUF_OFF) : 0.0f;
        f32x4 v[16], own;
        {
            const int j = node(t0), deg = j >= 0 ? end0 - beg0 : 0;
#pragma unroll
            for (int x = 0; x < 16; ++x) {
                const unsigned sid = (unsigned)xw_readlane_i(id0, x);
                v[x] = buf_ld_f32x4(r_state, (x < deg && act_l) ? sid * (unsigned)(SP * 4) + lane_off : BUF_OFF);
            }
            own = buf_ld_f32x4(r_state, (j >= 0 && act_l) ? (unsigned)j * (unsigned)(SP * 4) + lane_off : BUF_OFF);
        }
#pragma unroll 1
        while (t0 < total) {
            // the row two behind this one: its row pointers landed a row ago -> its source ids now; the row three behind: drawn now
            const int beg2 = xw_readlane_i(rp2, 0), end2 = xw_readlane_i(rp2, 1);
            const int id2 = buf_ld_i32(r_src, beg2 + lane < end2 ? 4u * (unsigned)(beg2 + lane) : BUF_OFF);
            const float w2 = HAS_W ? buf_ld_f32(r_w, beg2 + lane < end2 ? 4u * (unsigned)(beg2 + lane) : BUF_OFF) : 0.0f;
            const int t4 = draw();
            const int rp4 = ld_rowptr(node(t4));
            // this row: sum its window while the next row's neighbour rows take the freed registers
            const int j = node(t0), deg = j >= 0 ? end0 - beg0 : 0;
            const int jN = node(t1), degN = jN >= 0 ? end1 - beg1 : 0;
            const float scl = has_scale ? buf_ld_f32(r_scale, j >= 0 ? 4u * (unsigned)j : BUF_OFF) : 1.0f;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int x = 0; x < 16; ++x) {
                if (HAS_W) acc += xw_readlane_f(w0, x) * v[x];
                else acc += v[x];
                const unsigned sid = (unsigned)xw_readlane_i(id1, x);
                v[x] = buf_ld_f32x4(r_state, (x < degN && act_l) ? sid * (unsigned)(SP * 4) + lane_off : BUF_OFF);
            }
#pragma unroll 1
            for (int eb = 16; eb < deg; eb += 2) {                // in-degree > 16: two rows at a time (the window above stays in flight)
                if ((eb & 63) == 0) {                             // every 64 arcs: the next 64 source ids (uniform branch, rare)
                    const int e = beg0 + eb + lane;
                    id0 = buf_ld_i32(r_src, e < end0 ? 4u * (unsigned)e : BUF_OFF);
                    if (HAS_W) w0 = buf_ld_f32(r_w, e < end0 ? 4u * (unsigned)e : BUF_OFF);
                }
                f32x4 v2[2];
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    const unsigned sid = (unsigned)__builtin_amdgcn_readlane(id0, (eb + x) & 63);
                    v2[x] = buf_ld_f32x4(r_state, (eb + x < deg && act_l) ? sid * (unsigned)(SP * 4) + lane_off : BUF_OFF);
                }
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    if (HAS_W) acc += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w0), (eb + x) & 63)) * v2[x];
                    else acc += v2[x];
                }
            }
            if (has_scale) acc *= scl;
            const int i = t0 >> 5, r = t0 & 31, g = 2 * i + (r >> 4), h = g % HS, u = g / HS;
            {                                                     // the half-slot's previous rows must have been consumed
                XW_T(g0_);
                int spin = 0;
                while (__builtin_amdgcn_readfirstlane(f4_ld_acquire(&freed[h])) < u) {
                    if (spin >= XW_SPIN_MAX) { bad = 1; break; }
                    ++spin; __builtin_amdgcn_s_sleep(1);
                }
#ifdef XW_PROFILE
                if (wave == NMW) XW_ADD(4, xw_now() - g0_);
#endif
            }
            if (bad) break;                                       // the half-slot never came free: deposit nothing (k < 0 says so)
            char *xr = ring + (size_t)(h * 16 + (r & 15)) * RS + 8 * lane;        // k = 4 lane of the own-state half, plane 0
            if (act_l) {
                xb_store_split(xr, PL, own);
                xb_store_split(xr + 2 * KH, PL, acc);
            } else if (zero_l) {
                const u32x2 z = {0u, 0u};
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    *reinterpret_cast<u32x2 *>(xr + p * PL) = z;
                    *reinterpret_cast<u32x2 *>(xr + p * PL + 2 * KH) = z;
                }
            }
            if (lane == 0) __hip_atomic_fetch_add(&fill[h], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            own = buf_ld_f32x4(r_state, (jN >= 0 && act_l) ? (unsigned)jN * (unsigned)(SP * 4) + lane_off : BUF_OFF);   // (its registers are free now)
            // the stages move up
            t0 = t1; t1 = t2; t2 = t3; t3 = t4;
            beg0 = beg1; end0 = end1; id0 = id1; w0 = w1;
            beg1 = beg2; end1 = end2; id1 = id2; w1 = w2;
            rp2 = rp3; rp3 = rp4;
        }
    } else {
        // ================================ matrix waves: blocks wave, wave + NMW, .. of every tile ======================================
        // A wave's chain per tile (wait for the rows -> K loop -> epilogue) must stay under the time the gather waves need to fill a
        // tile, and every load of this CU returns in issue order BEHIND the gather's misses (~ 2 us each under load): nothing the chain
        // waits for may be requested inside it.  So the first XB_PD k-steps of the weights AND the per-node constant of the next block
        // are requested at the end of the previous one (they land while the wave waits for the rows), the weight stream runs XB_PD
        // k-steps ahead, and the rows of k-step ks + 1 are read from LDS behind the products of k-step ks, under the split of the next
        // weights (one register set).
        const int row = lane & 31, kg = lane >> 5;
        const int H = S;
        constexpr int PD = XB_PD;
        auto wbase_of = [&](int cb) -> unsigned { return ((unsigned)cb * (unsigned)NKS * 128u + (unsigned)lane) * 16u; };
        auto ld_w = [&](u32x4 (&w)[2], unsigned wb_, int ks, bool on) {  // the two pieces of k-step ks of the block at wb_
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                w[pl] = __builtin_amdgcn_raw_buffer_load_b128(r_wb, (int)((on && !(dbg & 2)) ? wb_ + (unsigned)ks * 2048u + (unsigned)pl * 1024u : BUF_OFF), 0, 0);
        };
        auto ld_c = [&](f32x4 (&c)[4], int cb, long jl_) {              // columns 32 cb + 8 q + 4 kg + e of row jl_ of the per-node constant
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col0 = 32 * cb + 8 * q + 4 * kg;
                c[q] = buf_ld_f32x4(r_C, (jl_ < N && col0 < a.ldC) ? ((unsigned)jl_ * (unsigned)a.ldC + (unsigned)col0) * 4u : BUF_OFF);
            }
        };
        u32x4 W[PD][2];
        {
            const unsigned wb = wbase_of(wave);
#pragma unroll
            for (int p = 0; p < PD; ++p) {
                ld_w(W[p], wb, p, p < NKS);
                __builtin_amdgcn_sched_barrier(0);                    // (issue order = the K loop's: its vmcnt waits are exact counts only then)
            }
        }
#pragma unroll 1
        for (int i = 0; i < nT; ++i) {
            const long T = (long)blockIdx.x + (long)i * gridDim.x;
            const long jl = 32 * T + row;
            const bool jv = jl < N;
            const unsigned j = (unsigned)jl;
            const int gA = 2 * i, hA = gA % HS, uA = gA / HS, gB = gA + 1, hB = gB % HS, uB = gB / HS;
            const int s = i & 1, round = i >> 1;
            XW_T(t0_);
            {
                int spin = 0;
                while (__builtin_amdgcn_readfirstlane(f4_ld_acquire(&fill[hA])) < 16 * (uA + 1) ||
                       __builtin_amdgcn_readfirstlane(f4_ld_acquire(&fill[hB])) < 16 * (uB + 1)) {
                    if (spin >= XW_SPIN_MAX) { bad = 1; break; }
                    ++spin; __builtin_amdgcn_s_sleep(1);
                }
            }
            if (bad) break;
            XW_T(t1_);
#ifdef XW_PROFILE
            if (wave == 0) { XW_ADD(0, t1_ - t0_); XW_ADD(3, 1); }
#endif
            const char *X = ring + (size_t)((row < 16 ? hA : hB) * 16 + (row & 15)) * RS;      // this lane's row of the tile
            const char *xk = X + 16 * kg;                                                          // k = 16 ks + 8 kg of plane 0
#pragma unroll 1
            for (int cb = wave; cb < NCB; cb += NMW) {
                if (dbg & 8) {
                    int last = 0;
                    if (lane == 0) last = __hip_atomic_fetch_add(&done[s], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == NCB * (round + 1) - 1;
                    if (__builtin_amdgcn_readfirstlane(last) && lane == 0) {
                        __hip_atomic_store(&freed[hA], uA + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(&freed[hB], uB + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    continue;
                }
                XW_T(tb_);
                f32x16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
                const unsigned wb = wbase_of(cb);
                // Whole rounds of PD k-steps; the loads of k-step ks + PD are issued BEHIND the matrix instructions of k-step ks, into the
                // registers they have just read (issued in front, hipcc rotates the pieces with copies behind a full drain).  Loads behind NKS are predicated off; the last NKS % PD k-steps are in W[] when the rounds end.
                const int n_main = NKS / PD * PD;
                u32x4 R[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) R[pl] = *reinterpret_cast<const u32x4 *>(xk + pl * PL);
                auto step = [&](u32x4 (&w)[2], int ks, bool refill) {
                    u32x4 wh, wm, wl;
                    split3_x8pk(__builtin_bit_cast(f32x4, w[0]), __builtin_bit_cast(f32x4, w[1]), wh, wm, wl);
                    if (!(dbg & 1)) acc = mfma_b6(wh, wm, wl, R[0], R[1], R[2], acc);
                    asm volatile("" : "+v"(acc));                     // (a load hoisted between the six products needs registers of its own: copies and
                    __builtin_amdgcn_sched_barrier(0);                //  a full drain at the loop's back edge)
                    const char *xn = xk + 32 * min(ks + 1, NKS - 1);  // the rows of the next k-step: they land under the split of its weights
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) R[pl] = *reinterpret_cast<const u32x4 *>(xn + pl * PL);
                    if (refill) ld_w(w, wb, ks + PD, ks + PD < NKS);
                    __builtin_amdgcn_sched_barrier(0);                // (the scheduler otherwise sinks all loads of a round to its end)
                };
#pragma unroll 1
                for (int ks0 = 0; ks0 < n_main; ks0 += PD) {
#pragma unroll
                    for (int p = 0; p < PD; ++p) step(W[p], ks0 + p, true);
                }
#pragma unroll
                for (int p = 0; p < PD - 1; ++p)
                    if (n_main + p < NKS) step(W[p], n_main + p, false);
                XW_T(t2_);
                f32x4 c4[4];
                ld_c(c4, cb, jl);
                asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");    // MFMA results are read behind a branch below (hipcc 7.2 hazard, kernels_train_big.hpp)
                float d2 = 0.0f, n2 = 0.0f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col0 = 32 * cb + 8 * q + 4 * kg;
                    f32x4 nv = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) nv[e] += col0 + e < H ? c4[q][e] : 0.0f;
                    activate4(a.act, nv);
                    const f32x4 ov = xb_load_joined(X + 2 * min(col0, KH - 4), PL);      // the old state of these columns: hi + mid + lo, exactly
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool ok = jv && col0 + e < S;
                        nv[e] = ok ? nv[e] : 0.0f;
                        const float o = ok ? ov[e] : 0.0f;
                        const float d = nv[e] - o;
                        d2 = fmaf(d, d, d2);
                        n2 = fmaf(o, o, n2);
                    }
                    const u32x4 bits = {__float_as_uint(nv[0]), __float_as_uint(nv[1]), __float_as_uint(nv[2]), __float_as_uint(nv[3])};
                    __builtin_amdgcn_raw_buffer_store_b128(bits, r_out, (jv && col0 < SP) ? (int)((j * (unsigned)SP + (unsigned)col0) * 4u) : (int)BUF_OFF, 0, 0);
                }
                d2 += __shfl_xor(d2, 32);
                n2 += __shfl_xor(n2, 32);
                float *pp = part + ((s * XW_NM + cb) * 32 + row) * 2;
                if (lane < 32) *reinterpret_cast<float2 *>(pp) = make_float2(d2, n2);
                int last = 0;
                if (lane == 0) last = __hip_atomic_fetch_add(&done[s], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == NCB * (round + 1) - 1;
                last = __builtin_amdgcn_readfirstlane(last);
                if (last) {
                    // every block of the tile is done (their shares and their reads of the rows are behind the counter): rows' predicate
                    // from the shares in block order, then both half-slots go back to the gather waves
                    float D2 = 0.0f, N2 = 0.0f;
                    for (int b = 0; b < NCB; ++b) {
                        const float2 sh = *reinterpret_cast<const float2 *>(part + ((s * XW_NM + b) * 32 + row) * 2);
                        D2 += sh.x; N2 += sh.y;
                    }
                    if (lane < 32 && jv && sqrtf(D2) > a.thr * sqrtf(N2)) any = 1;
                    if (lane == 0) {
                        __hip_atomic_store(&freed[hA], uA + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(&freed[hB], uB + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
#ifdef XW_PROFILE
                if (wave == 0) { const unsigned long long t3_ = xw_now(); XW_ADD(1, t2_ - tb_); XW_ADD(2, t3_ - t2_); }
#endif
                {   // the next block's first k-steps and constant (this tile's next block, or the first block of the next tile)
                    const bool same = cb + NMW < NCB;
                    const int cbn = same ? cb + NMW : wave;
                    const unsigned wbn = wbase_of(cbn);
#pragma unroll
                    for (int p = 0; p < PD; ++p) {
                        ld_w(W[p], wbn, p, p < NKS);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    }

#ifdef XW_PROFILE
    if (wave == NMW) XW_ADD(5, xw_now() - tk0_);         // the first gather wave: its whole loop
    if (wave == 0) XW_ADD(6, xw_now() - tk0_);           // matrix wave 0: its whole loop
#endif
    any = __syncthreads_or(any);
    bad = __syncthreads_or(bad);
    if (tid == 0) {
        if (any && a.flag_next) atomicOr(a.flag_next, 1);
        if (bad && a.err) atomicOr(a.err, 1);
        if (blockIdx.x == 0 && a.k_out) *a.k_out = a.k_val;
    }
}

inline int launch_xwide_weights_b3(const float *Wf, int H, int S, int wrow_state, int wrow_agg, int SP, void *Wb, hipStream_t st) {
    const int KH = xwide_kh(SP), NKS = KH / 8, NCB = (S + 31) / 32;
    const long total = (long)NCB * NKS * 512;
    k_xwide_weights_q<<<(int)std::min<long>((total + 255) / 256, 1024), 256, 0, st>>>(Wf, H, S, wrow_state, wrow_agg, KH, NKS, NCB, (float *)Wb);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// matrix_waves: 0 = one per 32-column block (the f32 form's division), else that many (each takes every matrix_waves-th block)
template <bool HAS_W>
int launch_xwide_b3_one(XWideArgs &xa, int grid, size_t lds, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void *)k_state_xwide_b3<HAS_W>, hipFuncAttributeMaxDynamicSharedMemorySize, XB_LDS_MAX) != hipSuccess) return 1;
        attr = true;
    }
    GNN_SET_KERNEL_NAME("k_state_xwide_b3<%s>", HAS_W ? "true" : "false");
    k_state_xwide_b3<HAS_W><<<grid, 1024, lds, st>>>(xa);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
inline int launch_xwide_b3(XWideArgs &xa, int n_cu, int matrix_waves, hipStream_t st) {
    xa.KH = xwide_kh(xa.SP); xa.NKS = xa.KH / 8; xa.NCB = (xa.S + 31) / 32;
    xa.HS = xb_half_slots(xa.KH);
    xa.NMW = matrix_waves > 0 ? std::min(matrix_waves, xa.NCB) : xa.NCB;
    if (xa.HS < 3) return 1;
    const size_t lds = xb_lds_bytes(xa.KH, xa.HS);
    const int ntiles = (xa.N + 31) / 32;
    const int grid = std::max(1, std::min(n_cu, ntiles));
    return xa.w ? launch_xwide_b3_one<true>(xa, grid, lds, st) : launch_xwide_b3_one<false>(xa, grid, lds, st);
}

}  // namespace gnn
