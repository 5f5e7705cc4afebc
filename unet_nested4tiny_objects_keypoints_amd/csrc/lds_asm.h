// Hand-placed LDS reads for the MFMA phases of the Winograd kernels (gemm_wino.hip, wgrad_wino.hip).
//
// hipcc sinks every ds_read to just in front of its first use, which leaves a lone wave's matrix pipe idle for an LDS
// round trip a dozen times per 32 MFMAs.  These helpers issue the reads as inline asm (compile-time byte offsets), a
// whole batch of MFMAs ahead, and collect them with one s_waitcnt per batch.  The wait "modifies" the registers it
// guards, so no consumer can be scheduled above it; __builtin_amdgcn_sched_barrier(0) around a batch keeps its MFMAs
// between the reads and the wait.  Waits are always lgkmcnt(0): counted waits are unsafe next to scalar loads, which
// return out of order on the same counter.
#pragma once
#include <utility>

#include "common.h"

namespace unetpp {

template <int I>
struct IC {
  static constexpr int v = I;
};
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(IC<I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {  // f(IC<0>{}) ... f(IC<N-1>{}): indices usable as asm immediates
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}
__device__ __forceinline__ unsigned lds_offset(const float* p) {  // low half of a flat LDS address = LDS byte offset
  return static_cast<unsigned>(reinterpret_cast<uintptr_t>(p));
}
template <int OFF>
__device__ __forceinline__ void lds_read_b32(float& v, unsigned addr) {
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}
template <int O0, int O1>  // two 8-byte reads 512 * O0 and 512 * O1 bytes above addr
__device__ __forceinline__ void lds_read2st64_b64(f32x4& v, unsigned addr) {
  asm volatile("ds_read2st64_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(O0), "n"(O1));
}
__device__ __forceinline__ void lds_wait(f32x4& a, f32x4& b) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void lds_wait4(float (&d)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
}
__device__ __forceinline__ void lds_wait16(float (&d)[16]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]));
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(d[8]), "+v"(d[9]), "+v"(d[10]), "+v"(d[11]), "+v"(d[12]), "+v"(d[13]), "+v"(d[14]), "+v"(d[15]));
}

}  // namespace unetpp
