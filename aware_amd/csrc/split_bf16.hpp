// Shared by gemm_x3.hip and gemm_h2.hip: vector typedefs, the exact three-way bf16 split of an f32 pair, epilogue kinds.
#pragma once
#include <hip/hip_runtime.h>

namespace aware {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// (x, y) -> packed bf16 pairs of the three planes; residuals are exact f32 subtractions
__device__ __forceinline__ void split_pair(float x, float y, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = cvt_pk_bf16(x, y);
    const float rx = x - __uint_as_float(p0 << 16), ry = y - __uint_as_float(p0 & 0xFFFF0000u);
    p1 = cvt_pk_bf16(rx, ry);
    const float sx = rx - __uint_as_float(p1 << 16), sy = ry - __uint_as_float(p1 & 0xFFFF0000u);
    p2 = cvt_pk_bf16(sx, sy);
}

enum { X3_PLAIN = 0, X3_FWD = 1, X3_BWD = 2, X3_FWD_LAST = 3 };   // 3: FWD + split-K partials of the next (last, skinny) conv

}  // namespace aware
