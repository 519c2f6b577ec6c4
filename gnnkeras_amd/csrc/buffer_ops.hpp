// Raw buffer loads / stores (gfx950 `buffer_load_* ... offen`) used by the fused iteration kernels.
// Two properties matter here:
//   * a load whose offset is out of range returns 0 and touches no memory: predicating a load off by giving it
//     BUF_OFF keeps the instruction stream branch-free, so hipcc's s_waitcnt vmcnt(N) counts stay exact (with
//     `cond ? *p : 0` it branches around each load and waits for it alone: CDNA guide, trap (c) of the split-K section);
//   * the `aux` cache bits: 16 = sc1, the access goes through to memory instead of stopping in this XCD's L2 / this
//     CU's L1 (inter-workgroup hand-off inside a launch, kernel_state_small.hpp).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <mutex>
#include <vector>
#include "kernels_general.hpp"

namespace gnn {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void *p) {
    // 4 GiB window (the launcher checks every array fits); a null array becomes a zero-record buffer: loads return 0
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, p ? (int)0xFFFFFFF0u : 0, 0x00020000);
}
// ... and a window of exactly `bytes` bytes: offsets past the array's end are out of range by themselves (row-streaming kernels whose
// last tile is ragged need no select per load)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc_n(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, p ? (int)bytes : 0, 0x00020000);
}
// a kernel argument the compiler has come to hold in vector registers (it reloads arguments inside divergent control flow) makes every
// buffer instruction on its descriptor a waterfall loop: pass the pointer through here first
__device__ __forceinline__ const void *uniform_ptr(const void *p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const void *)(((unsigned long long)hi << 32) | lo);
}
constexpr unsigned BUF_OFF = 0xFFFFFFFFu;     // out of range for every descriptor: the load returns 0, no memory access

__device__ __forceinline__ int buf_ld_i32(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return (int)__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0);
}
__device__ __forceinline__ float buf_ld_f32(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}
__device__ __forceinline__ f32x4 buf_ld_f32x4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    return (f32x4){__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
}
// ... with cache-policy bits (2 = nt: rows that are streamed once per launch)
template <int AUX> __device__ __forceinline__ f32x4 buf_ld_f32x4_aux(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, AUX);
    return (f32x4){__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
}

// sc1 = system-coherent level 1: the access goes through to memory instead of stopping in this XCD's L2 / this CU's L1
__device__ __forceinline__ f32x4 buf_ld_sc1(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 16);
    return (f32x4){__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
}
__device__ __forceinline__ void buf_st_sc1(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 x) {
    const u32x4 v = {__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3])};
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)off, 0, 16);
}

// Bounded wait of ONE lane on a word another workgroup of the same launch will write (grid barriers of the whole-loop kernels, set
// barriers of the one-CU-per-group kernel).  The bound is WALL-CLOCK time (wall_clock64: s_memrealtime, 100 MHz, the same base on
// every CU), not a poll count: a persistent launch whose grid fits the GPU (checked at launch, persistent_fits() in gnnloop.hip)
// becomes fully resident as soon as whatever else runs on the GPU lets go of the CUs it needs, so the wait only has to outlast a
// co-tenant - `budget_ticks` comes from GNN_WAIT_MS (default 2 000 ms; the old bound of 2^22 polls was ~0.4 s) - while a protocol
// bug or a launch that can never be resident still ends, loudly (k < 0), instead of hanging the GPU.  budget_ticks == 0 (GNN_WAIT_MS=0)
// makes every wait expire at once, satisfied or not: the test hook that drives the callers' recovery paths.
// Set (never cleared by a kernel) when a wait below runs out of its budget: a device word a TEST can hand to gnn_debug_occupy_until as the
// co-tenant's release flag - the co-tenant then leaves exactly when the first wait of the launch under test has expired (include/gnnloop.h:
// gnn_debug_expiry_beacon).  One relaxed store on the failure path; nothing on the path that is waited through.
__device__ int g_wait_expired_beacon;

template <typename Done>
__device__ __forceinline__ bool wait_until(unsigned long long budget_ticks, Done done) {
    if (budget_ticks == 0) return false;
    // The clock is read every 64th poll only, the first time to start the budget (a few microseconds late: nothing against
    // milliseconds) - a wait that is satisfied at once, or within its first polls, costs what the bare poll loop cost
    // (s_memrealtime is a scalar memory operation with a long latency: read at the start of every wait it showed up in the
    // 15 us iterations of the one-CU-per-group kernel).
    unsigned long long t0 = 0;
    for (int spin = 0;; ++spin) {
        if (done()) return true;
        if ((spin & 63) == 63) {
            const unsigned long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > budget_ticks) { __hip_atomic_store(&g_wait_expired_beacon, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// ---- host side of the same contract -----------------------------------------------------------------------------------------------
// GNN_WAIT_MS -> ticks of the 100 MHz wall clock (default 2 000 ms; 0 = every wait expires at once, the recovery paths' test hook)
inline unsigned long long wait_ticks() {         // (read at every launch: the co-tenant tests shorten the bound inside one process)
    const char *e = getenv("GNN_WAIT_MS");
    const long long ms = e ? atoll(e) : 2000;
    return (unsigned long long)((ms < 0 ? 0 : ms) * 100000ll);
}

// Can `grid` workgroups of this kernel be resident AT ONCE on the device (hipOccupancyMaxActiveBlocksPerMultiprocessor x CUs)?  A
// launch whose workgroups wait for each other must not be larger than that, or its first barrier can never complete; asked once per
// (kernel, block size, dynamic LDS) and remembered.
inline bool persistent_fits(const void *fn, int threads, size_t lds, int grid, int n_cu) {
    struct Key { const void *fn; int threads; size_t lds; int per_cu; };
    static std::vector<Key> seen;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    for (const Key &k : seen)
        if (k.fn == fn && k.threads == threads && k.lds == lds) return (long)k.per_cu * n_cu >= grid;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
    seen.push_back(Key{fn, threads, lds, per_cu});
    return (long)per_cu * n_cu >= grid;
}

}  // namespace gnn
