// Raw buffer loads / stores (gfx950 `buffer_load_* ... offen`) used by the fused iteration kernels.
// Two properties matter here:
//   * a load whose offset is out of range returns 0 and touches no memory: predicating a load off by giving it
//     BUF_OFF keeps the instruction stream branch-free, so hipcc's s_waitcnt vmcnt(N) counts stay exact (with
//     `cond ? *p : 0` it branches around each load and waits for it alone: CDNA guide, trap (c) of the split-K section);
//   * the `aux` cache bits: 16 = sc1, the access goes through to memory instead of stopping in this XCD's L2 / this
//     CU's L1 (inter-workgroup hand-off inside a launch, kernel_state_small.hpp).
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_general.hpp"

namespace gnn {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void *p) {
    // 4 GiB window (the launcher checks every array fits); a null array becomes a zero-record buffer: loads return 0
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, p ? (int)0xFFFFFFF0u : 0, 0x00020000);
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

// sc1 = system-coherent level 1: the access goes through to memory instead of stopping in this XCD's L2 / this CU's L1
__device__ __forceinline__ f32x4 buf_ld_sc1(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 16);
    return (f32x4){__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
}
__device__ __forceinline__ void buf_st_sc1(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 x) {
    const u32x4 v = {__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3])};
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)off, 0, 16);
}

}  // namespace gnn
