// bf16 storage helpers of the reduced-precision twins (BASELINE configs[3]/[4]): activations and activation gradients
// live in HBM as bf16 NHWC, every sum is taken in fp32 (MFMA accumulators, BatchNorm statistics, weight gradients).
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace unetpp {

typedef unsigned short bf16_t;  // raw storage

__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float bf_to_f(bf16_t b) { return __uint_as_float(static_cast<unsigned>(b) << 16); }
// round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  const bf16x2 v = {static_cast<__bf16>(lo), static_cast<__bf16>(hi)};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ bf16_t f_to_bf(float f) { return static_cast<bf16_t>(pack_bf2(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float bf_round(float f) { return bf_lo(pack_bf2(f, 0.f)); }

// 8 consecutive bf16 (16 bytes) <-> 8 floats
__device__ __forceinline__ void unpack8(const u32x4& u, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = bf_lo(u[i]);
    f[2 * i + 1] = bf_hi(u[i]);
  }
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
  return u32x4{pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7])};
}

inline bool bf16_view_aligned(const unetpp_view& v) {
  return ((v.C | v.c_off | v.c_len) & 7) == 0 && (reinterpret_cast<uintptr_t>(v.ptr) & 15) == 0 &&
         (v.gate == nullptr || (reinterpret_cast<uintptr_t>(v.gate) & 15) == 0);
}

}  // namespace unetpp
