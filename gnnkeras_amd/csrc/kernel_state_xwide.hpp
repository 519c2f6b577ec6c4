// Fused state-transition iteration for state widths 129 .. 256 (one Dense layer, homogeneous graphs): the reference's
// `convergence` + the `condition` of the next iteration (GNN/Models/GNN.py:217-236, :196-214) in one launch, as
// k_state_fused4 / k_state_wide do for narrower states.  Before this kernel these widths ran un-fused (k_aggregate_vec writes
// the neighbour sums, k_rowdense_wide reads them back: 1.15 ms per iteration at d = 200 on 300 k nodes / 3 M arcs = 34 % of the
// HBM roofline, the Dense layer alone at 47 TFLOP/s).
//
// What is different at these widths: W1 = [2 d x d] floats is 320 KB at d = 200 - it cannot live in the CU's 160 KB of LDS.
// So the roles of LDS and L2 are swapped against the narrower kernels:
//   * the gathered rows [state | neighbour sum] of a 32-node tile live in LDS (two slots of 32 x (2 KH + 4) floats: 132 KB at
//     KH = 256), filled by the 16 - NCB waves that are not matrix waves (8 .. 11) - one wave per row, a lane per 16-byte chunk, rows
//     drawn from a ticket counter in LDS, ONE rolling window of 16 neighbour rows + the own row in flight per wave, the row
//     pointers fetched three rows and the source ids two rows ahead;
//   * the weights stream from L2 as MFMA operands: a set-up kernel lays them out once per call in fragment order
//     (k_xwide_weights), so that a matrix wave's lane reads ONE 16-byte piece per four MFMAs, whole 1-KB lines per wave
//     instruction, XW_PD pieces in flight; every tile re-reads the matrix (L2 traffic ~ (2 KH x 32 NCB x 4 B) per 32 rows, about
//     the size of the gather traffic; the matrix itself stays resident in each XCD's 4 MB L2);
//   * NCB <= 8 matrix waves, one per 32-column block of the output, on v_mfma_f32_32x32x2_f32: this chip sustains 156 TFLOP/s on that
//     instruction against 104-126 on the 16x16x4 one the narrower kernels use (scripts/micro/mfma_peak.hip), and a 32 x 32 block
//     needs one operand value per lane and 4 096 FLOP.  Operands are swapped (weights = A, rows = B) so that a lane ends up with
//     columns 8 q + 4 (lane / 32) + 0..3 of ITS row: C, the old state and the new state all move as 16-byte pieces.
// Measured (d = 200, 300 k nodes / 3 M arcs): 683 us per iteration against 1 150 un-fused (34 -> 57 % of the HBM roofline of its
// algorithmic bytes); d = 160 on 1 M / 10 M 3 197 -> 1 754 (60 %).  With 8 statically assigned gather waves (the first layout, 742 us;
// scripts/dev/xw_prof.py on a -DXW_PROFILE build) a matrix wave spends per tile 24 100 cycles in the K loop (two waves
// share a SIMD's matrix pipe: 25 600 would be the pipe's whole time), 8 300 in the epilogue and 16 200 waiting for the gather waves;
// rocprofv3: SQ_VALU_MFMA_BUSY_CYCLES 8.4e8 = 46 % of the SIMD-cycles of the launch at the 2.25 GHz GRBM_GUI_ACTIVE shows.  The launch
// is bound by the gather: 8 gather waves per CU move 4.7 TB/s of lines next to the weight stream (6.2 TB/s with the MFMAs compiled
// out, 526 us) - the working set (two 240 MB state buffers) is past the Infinity Cache, and this memory system rewards the NUMBER of
// waves with a gather outstanding (profiles/r01_gather_sweep.txt): hence every wave that is not a matrix wave gathers now (9 at
// d = 200: 742 -> 683 us; 11 at d = 160: 1 983 -> 1 754).  Tried and
// measured slower: 4 / 8 / 10 / 12 weight pieces in flight (722 - 772 us); two blocks per matrix wave on alternating tiles so that
// one wave's epilogue lies under the other's K loop (779 us: the gather still sets the pace); 12 gather waves drawing rows from a
// ticket counter + 4 matrix waves (one per SIMD, two blocks each): the gather keeps up (the matrix waves wait 2 000 cycles per tile
// instead of 16 000) but one wave per SIMD runs its K loop at 59 % of the pipe (43 600 cycles per tile against 25 600): 751 us.
// Slots are handed over with monotonic LDS counters (rows deposited / waves done / rounds freed), workgroup scope, bounded
// spins that raise the sticky error word (k < 0) exactly as in k_state_fused4.  The convergence predicate needs whole rows:
// every matrix wave leaves its block's share of |new - old|^2 and |old|^2 per row in LDS, the wave that finishes the tile last
// adds the shares in block order (deterministic) and tests the rows.
#pragma once
#include <hip/hip_runtime.h>
#include "kernel_state_fused4.hpp"
#include "buffer_ops.hpp"

namespace gnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct XWideArgs {
    const int *gate; int n_gate, gate_stride;      // run iff OR of gate[i * gate_stride], i < n_gate, is non-zero
    const int *rowptr, *src;                       // CSR by destination
    const float *w, *row_scale;
    const float *state_in;                         // [n_src_rows, SP]
    float *state_out;                              // [N, SP]
    const float *C; int ldC;                       // per-node constant of the first layer (bias included)
    const float *Wx;                               // first-layer weights in fragment order (k_xwide_weights)
    int N, S, SP;                                  // nodes, state width, leading dimension of the state buffers (multiple of 4)
    int KH, NG, NCB;                               // K per half (SP rounded up to 8), 8-k groups (= KH / 4), 32-column blocks
    const void *Wb;                                // kernel_state_xwide_b3.hpp: the weights in the fragment order of the 16-k instruction
    int dbg;                                       //   -DXB_EXPERIMENT builds: ablation bits (GNN_XB_DBG)
    int NKS, HS, NMW;                              //   16-k steps (= KH / 8), half-slots of the row ring, matrix waves
    int act;
    float thr;
    int *flag_next;
    float *k_out; float k_val;
    int *err;
};

constexpr int XW_NM = 8;          // matrix waves at most (wave ids 0 .. NCB - 1, block cb = wave id); the other waves gather
constexpr int XW_NS = 2;          // LDS slots
#ifndef XW_PD_VALUE
#define XW_PD_VALUE 6
#endif
constexpr int XW_PD = XW_PD_VALUE;   // weight pieces in flight per matrix wave
#ifndef GNN_F4_SPIN_MAX
#define GNN_F4_SPIN_MAX (1 << 22)
#endif
constexpr int XW_SPIN_MAX = GNN_F4_SPIN_MAX;   // (-DGNN_F4_SPIN_MAX=0: the debug build whose every wait expires at once, libgnnloop_spin0.so)

inline int xwide_kh(int SP) { return (SP + 7) & ~7; }
inline size_t xwide_weight_floats(int S, int SP) { return (size_t)((S + 31) / 32) * (xwide_kh(SP) / 4) * 256; }      // (either fragment order: this file's or kernel_state_xwide_b3.hpp's)
inline size_t xwide_lds_bytes(int KH) {
    return sizeof(float) * ((size_t)XW_NS * 32 * (2 * KH + 4) + (size_t)XW_NS * XW_NM * 32 * 2) + sizeof(int) * (3 * XW_NS + 1);
}

// Wx[((cb * NG + jg) * 64 + lane) * 4 + e] = Wcat[8 jg + 4 (lane / 32) + e][32 cb + lane % 32], Wcat = [state rows (KH, zero padded) ;
// neighbour-sum rows (KH)] of the folded first layer Wf [in_dim x H]
__global__ void k_xwide_weights(const float *Wf, int H, int S, int wrow_state, int wrow_agg, int KH, int NG, int NCB, float *Wx) {
    const long total = (long)NCB * NG * 256;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 3), lane = (int)((i >> 2) & 63);
        const long rest = i >> 8;
        const int jg = (int)(rest % NG), cb = (int)(rest / NG);
        const int k = 8 * jg + 4 * (lane >> 5) + e, col = 32 * cb + (lane & 31);
        const int kk = k < KH ? k : k - KH;
        float v = 0.0f;
        if (kk < S && col < H) v = Wf[(size_t)((k < KH ? wrow_state : wrow_agg) + kk) * H + col];
        Wx[i] = v;
    }
}

// -DXW_PROFILE (experiment builds): shader-clock totals of the phases of matrix wave 0 and gather wave 8 of every workgroup
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

template <bool HAS_W>
__global__ void __launch_bounds__(1024) k_state_xwide(XWideArgs a) {
    int open = a.gate == nullptr;
    for (int i = 0; i < a.n_gate; ++i) open |= a.gate[(size_t)i * a.gate_stride] != 0;
    extern __shared__ __attribute__((aligned(16))) char xw_smem[];
    const int KH = a.KH, NG = a.NG, LDX = 2 * KH + 4, SLOT = 32 * LDX;
    float *Xs = reinterpret_cast<float *>(xw_smem);                     // [NS][32][LDX] : [state (KH) | neighbour sum (KH) | pad]
    float *part = Xs + XW_NS * SLOT;                                    // [NS][NM][32][2] : per block and row |new - old|^2, |old|^2
    int *fill = reinterpret_cast<int *>(part + XW_NS * XW_NM * 64);     // [NS] rows deposited so far
    int *freed = fill + XW_NS;                                          // [NS] rounds consumed so far
    int *done = freed + XW_NS;                                          // [NS] matrix waves finished so far
    int *ticket = done + XW_NS;                                         // next row (tile * 32 + row) to gather
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 3 * XW_NS + 1) fill[tid] = 0;
    __syncthreads();
    if (!open) return;                             // uniform across the launch; nothing has left the CU yet

    const int N = a.N, SP = a.SP, S = a.S;
    const int ntiles = (N + 31) >> 5;
    const int nT = (int)blockIdx.x < ntiles ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;   // tiles of this workgroup
    const __amdgpu_buffer_rsrc_t r_state = buf_rsrc(a.state_in), r_rowptr = buf_rsrc(a.rowptr), r_src = buf_rsrc(a.src),
                                 r_w = buf_rsrc(HAS_W ? a.w : nullptr), r_scale = buf_rsrc(a.row_scale), r_C = buf_rsrc(a.C),
                                 r_wx = buf_rsrc(a.Wx), r_out = buf_rsrc(a.state_out);
    int any = 0, bad = 0;

    XW_T(tk0_);
    if (wave >= a.NCB) {
        // ================================ gather waves ================================================================
        // Every wave that is not a matrix wave gathers (16 - NCB of them: 8 at 256 columns, 11 at 160).  Rows are drawn from a ticket
        // counter in LDS (ticket t = row t % 32 of this workgroup's tile t / 32; a wave's tickets grow, so no row waits behind a
        // later one).  A wave always holds four tickets: the row whose neighbour rows are being summed, the row whose neighbour
        // rows are being issued into the window slots the sum frees (ONE rolling window of 16 neighbour rows + the own row in
        // flight), the row whose source ids are in flight and the row whose row pointers are in flight.
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
            const int i = t0 >> 5, r = t0 & 31, s = i % XW_NS, round = i / XW_NS;
            {                                                     // the slot's previous tile must have been consumed
                XW_T(g0_);
                int spin = 0;
                while (__builtin_amdgcn_readfirstlane(f4_ld_acquire(&freed[s])) < round) {
                    if (spin >= XW_SPIN_MAX) { bad = 1; break; }
                    ++spin; __builtin_amdgcn_s_sleep(1);
                }
#ifdef XW_PROFILE
                if (wave == a.NCB) XW_ADD(4, xw_now() - g0_);
#endif
            }
            if (bad) break;                                       // the slot never came free: deposit nothing (k < 0 says so)
            float *xr = Xs + s * SLOT + r * LDX + 4 * lane;
            if (act_l) {
                *reinterpret_cast<f32x4 *>(xr) = own;
                *reinterpret_cast<f32x4 *>(xr + KH) = acc;
            } else if (zero_l) {
                *reinterpret_cast<f32x4 *>(xr) = (f32x4){0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4 *>(xr + KH) = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            if (lane == 0) __hip_atomic_fetch_add(&fill[s], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            own = buf_ld_f32x4(r_state, (jN >= 0 && act_l) ? (unsigned)jN * (unsigned)(SP * 4) + lane_off : BUF_OFF);   // (its registers are free now)
            // the stages move up
            t0 = t1; t1 = t2; t2 = t3; t3 = t4;
            beg0 = beg1; end0 = end1; id0 = id1; w0 = w1;
            beg1 = beg2; end1 = end2; id1 = id2; w1 = w2;
            rp2 = rp3; rp3 = rp4;
        }
    } else if (wave < a.NCB) {
        // ================================ matrix waves ================================================================
        const int cb = wave;
        const int row = lane & 31, kh = lane >> 5;
        const unsigned wbase = ((unsigned)cb * (unsigned)NG * 64u + (unsigned)lane) * 16u;
        const int H = S;
        // the first XW_PD weight pieces of a tile are fetched before the previous tile's epilogue (they are the same pieces every tile)
        f32x4 Bf[XW_PD];
#pragma unroll
        for (int p = 0; p < XW_PD; ++p) Bf[p] = buf_ld_f32x4(r_wx, p < NG ? wbase + (unsigned)p * 1024u : BUF_OFF);
#pragma unroll 1
        for (int i = 0; i < nT; ++i) {
            const long T = (long)blockIdx.x + (long)i * gridDim.x;
            const long jl = 32 * T + row;
            const bool jv = jl < N;
            const unsigned j = (unsigned)jl;
            // the per-node constant (columns 32 cb + 8 q + 4 kh + e of this lane's row): fetched now, added behind the K loop
            f32x4 c4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col0 = 32 * cb + 8 * q + 4 * kh;
                c4[q] = buf_ld_f32x4(r_C, (jv && col0 < a.ldC) ? (j * (unsigned)a.ldC + (unsigned)col0) * 4u : BUF_OFF);
            }
            f32x16 acc, acc1;
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[e] = 0.0f; acc1[e] = 0.0f; }
            const int s = i % XW_NS, round = i / XW_NS;
            XW_T(t0_);
            {
                int spin = 0;
                while (__builtin_amdgcn_readfirstlane(f4_ld_acquire(&fill[s])) < 32 * (round + 1)) {
                    if (spin >= XW_SPIN_MAX) { bad = 1; break; }
                    ++spin; __builtin_amdgcn_s_sleep(1);
                }
            }
            if (bad) break;
            XW_T(t1_);
            const float *X = Xs + s * SLOT + row * LDX;
            const float *xrow = X + 4 * kh;
            // Whole rounds of XW_PD pieces; the load of piece jg + XW_PD is issued BEHIND the MFMAs of piece jg, into the registers they
            // have just read (issued in front of them it needs other registers, and hipcc then rotates the pieces with copies at the
            // loop's back edge - behind an s_waitcnt vmcnt(0): the stream drained once per round, 60 % of the matrix rate).  Loads
            // behind NG are predicated off (no branch around a memory operation: the vmcnt waits stay exact counts); the last
            // NG % XW_PD pieces are already in Bf[] when the rounds end and run behind a uniform branch that holds no load.
            // Two accumulators (even / odd k pairs) keep one wave's MFMAs independent of each other.
            const int n_main = NG / XW_PD * XW_PD;
            static_assert(XW_PD % 2 == 0, "the rows' pieces alternate between two register sets");
            f32x4 A[2];                                           // the rows' piece for jg + 1 is read from LDS in front of the MFMAs of piece jg
            A[0] = *reinterpret_cast<const f32x4 *>(xrow);
#pragma unroll 1
            for (int jg0 = 0; jg0 < n_main; jg0 += XW_PD) {
#pragma unroll
                for (int p = 0; p < XW_PD; ++p) {
                    const int jg = jg0 + p;
                    A[(p + 1) & 1] = *reinterpret_cast<const f32x4 *>(xrow + 8 * min(jg + 1, NG - 1));
                    const f32x4 a4 = A[p & 1];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Bf[p][0], a4[0], acc, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(Bf[p][1], a4[1], acc1, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Bf[p][2], a4[2], acc, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(Bf[p][3], a4[3], acc1, 0, 0, 0);
                    Bf[p] = buf_ld_f32x4(r_wx, jg + XW_PD < NG ? wbase + (unsigned)(jg + XW_PD) * 1024u : BUF_OFF);
                    __builtin_amdgcn_sched_barrier(0);            // (the scheduler otherwise sinks all loads of a round to its end)
                }
            }
#pragma unroll
            for (int p = 0; p < XW_PD - 1; ++p) {
                if (n_main + p < NG) {
                    A[(p + 1) & 1] = *reinterpret_cast<const f32x4 *>(xrow + 8 * min(n_main + p + 1, NG - 1));
                    const f32x4 a4 = A[p & 1];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Bf[p][0], a4[0], acc, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(Bf[p][1], a4[1], acc1, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Bf[p][2], a4[2], acc, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(Bf[p][3], a4[3], acc1, 0, 0, 0);
                }
            }
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] += acc1[e];
            XW_T(t2_);
#pragma unroll
            for (int p = 0; p < XW_PD; ++p) Bf[p] = buf_ld_f32x4(r_wx, p < NG ? wbase + (unsigned)p * 1024u : BUF_OFF);
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");    // MFMA results are read behind a branch below (hipcc 7.2 hazard, kernels_train_big.hpp)
            float d2 = 0.0f, n2 = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col0 = 32 * cb + 8 * q + 4 * kh;
                f32x4 nv = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) nv[e] += col0 + e < H ? c4[q][e] : 0.0f;
                activate4(a.act, nv);
                const f32x4 ov = *reinterpret_cast<const f32x4 *>(X + min(col0, KH - 4));
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
            if (lane == 0) last = __hip_atomic_fetch_add(&done[s], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == a.NCB * (round + 1) - 1;
            last = __builtin_amdgcn_readfirstlane(last);
            if (last) {
                // every block of the tile is done (their shares and their reads of the slot are behind the counter): rows' predicate
                // from the shares in block order, then the slot goes back to the gather waves
                float D2 = 0.0f, N2 = 0.0f;
                for (int b = 0; b < a.NCB; ++b) {
                    const float2 sh = *reinterpret_cast<const float2 *>(part + ((s * XW_NM + b) * 32 + row) * 2);
                    D2 += sh.x; N2 += sh.y;
                }
                if (lane < 32 && jv && sqrtf(D2) > a.thr * sqrtf(N2)) any = 1;
                if (lane == 0) __hip_atomic_store(&freed[s], round + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
#ifdef XW_PROFILE
            if (wave == 0) { const unsigned long long t3_ = xw_now(); XW_ADD(0, t1_ - t0_); XW_ADD(1, t2_ - t1_); XW_ADD(2, t3_ - t2_); XW_ADD(3, 1); }
#endif
        }
    }

#ifdef XW_PROFILE
    if (wave == a.NCB) XW_ADD(5, xw_now() - tk0_);       // the first gather wave: its whole loop
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

inline int launch_xwide_weights(const float *Wf, int H, int S, int wrow_state, int wrow_agg, int SP, float *Wx, hipStream_t st) {
    const int KH = xwide_kh(SP), NG = KH / 4, NCB = (S + 31) / 32;
    const long total = (long)NCB * NG * 256;
    k_xwide_weights<<<(int)std::min<long>((total + 255) / 256, 1024), 256, 0, st>>>(Wf, H, S, wrow_state, wrow_agg, KH, NG, NCB, Wx);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

inline int launch_xwide(XWideArgs &xa, int n_cu, hipStream_t st) {
    xa.KH = xwide_kh(xa.SP); xa.NG = xa.KH / 4; xa.NCB = (xa.S + 31) / 32;
    const size_t lds = xwide_lds_bytes(xa.KH);
    static bool attr[2] = {false, false};
    const int hw = xa.w ? 1 : 0;
    if (!attr[hw]) {
        const void *f = hw ? (const void *)k_state_xwide<true> : (const void *)k_state_xwide<false>;
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)xwide_lds_bytes(256)) != hipSuccess) return 1;
        attr[hw] = true;
    }
    const int ntiles = (xa.N + 31) / 32;
    const int grid = std::max(1, std::min(n_cu, ntiles));
    GNN_SET_KERNEL_NAME("k_state_xwide<%s>", hw ? "true" : "false");
    if (hw) k_state_xwide<true><<<grid, 1024, lds, st>>>(xa);
    else    k_state_xwide<false><<<grid, 1024, lds, st>>>(xa);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

}  // namespace gnn
