// The three-term bf16 split of an f32 operand and the six matrix instructions of a split product (shared by the training kernels,
// kernels_train_big.hpp - where the scheme, its accuracy and its measurements are described - and the wide forward kernel,
// kernel_state_xwide.hpp).  x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid), round to nearest at every level:
// exact (8 + 8 + 8 significand bits, signed), and  x w = hi wh + (hi wm + mid wh) + (mid wm + hi wl + lo wh) + O(2^-24 |x w|).
#pragma once
#include <hip/hip_runtime.h>
#include "buffer_ops.hpp"

namespace gnn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ... the same split on packed f32 pairs: 9 VALU instructions a pair (v_cvt_pk_bf16_f32 x 3, the two unpacks x 2, v_pk_add_f32 x 2).  The
// conversion is inline assembly so that hipcc keeps the PAIR conversion (it otherwise converts the low element a second time, alone, to
// shift it: 100 conversions a tile where 60 do); the bits are split3_pair's.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(f32x2 x) {
    const bf16x2 hv = {(__bf16)x[0], (__bf16)x[1]};
    unsigned r = __builtin_bit_cast(unsigned, hv);
    asm("" : "+v"(r));                  // (opaque from here on: no instruction, and hipcc cannot look through the shifts below to the conversion)
    return r;
}
__device__ __forceinline__ void split3_pk(f32x2 x, unsigned &h, unsigned &m, unsigned &l) {
    h = cvt_pk_bf16(x);
    const f32x2 r1 = x - (f32x2){__uint_as_float(h << 16), __uint_as_float(h & 0xFFFF0000u)};
    m = cvt_pk_bf16(r1);
    const f32x2 r2 = r1 - (f32x2){__uint_as_float(m << 16), __uint_as_float(m & 0xFFFF0000u)};
    l = cvt_pk_bf16(r2);
}

__device__ __forceinline__ void split3_x8pk(const f32x4 &x0, const f32x4 &x1, u32x4 &h, u32x4 &m, u32x4 &l) {
    unsigned hh[4], mm[4], ll[4];
    split3_pk((f32x2){x0[0], x0[1]}, hh[0], mm[0], ll[0]); split3_pk((f32x2){x0[2], x0[3]}, hh[1], mm[1], ll[1]);
    split3_pk((f32x2){x1[0], x1[1]}, hh[2], mm[2], ll[2]); split3_pk((f32x2){x1[2], x1[3]}, hh[3], mm[3], ll[3]);
    h = (u32x4){hh[0], hh[1], hh[2], hh[3]}; m = (u32x4){mm[0], mm[1], mm[2], mm[3]}; l = (u32x4){ll[0], ll[1], ll[2], ll[3]};
}

__device__ __forceinline__ f32x16 mfma_b6(const u32x4 &wh, const u32x4 &wm, const u32x4 &wl, const u32x4 &xh, const u32x4 &xm, const u32x4 &xl, f32x16 acc) {
#define B8(v_) __builtin_bit_cast(bf16x8, v_)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B8(wl), B8(xh), acc, 0, 0, 0);      // small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B8(wh), B8(xl), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B8(wm), B8(xm), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B8(wm), B8(xh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B8(wh), B8(xm), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B8(wh), B8(xh), acc, 0, 0, 0);
#undef B8
    return acc;
}

}  // namespace gnn
