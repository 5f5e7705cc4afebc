// Counter-based dropout keep mask shared by the fp32 and bf16 head kernels.
#pragma once
#include "common.h"

namespace unetpp {

// Counter-based: one splitmix64 hash per group of 4 channels of one pixel gives four 16-bit
// uniforms; forward and backward regenerate the same mask from (seed, pixel, group).
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t keep_bits(uint64_t seed, long pixel, int cgroups4, int g4) {
  return mix64(seed + 0x9E3779B97F4A7C15ULL * (static_cast<uint64_t>(pixel) * cgroups4 + g4 + 1));
}
// keep flag of slice channel c of pixel p
__device__ __forceinline__ bool keep_one(uint64_t bits, int c_in_group, uint32_t thr16) {
  return ((bits >> (16 * c_in_group)) & 0xFFFFu) < thr16;
}


constexpr int kHeadMaxC = 128;
constexpr int kHeadMaxCls = 8;

inline uint32_t keep_threshold(float p_drop) {
  const double keep = 1.0 - static_cast<double>(p_drop);
  uint32_t t = static_cast<uint32_t>(keep * 65536.0 + 0.5);
  return t > 65536u ? 65536u : t;
}

}  // namespace unetpp
