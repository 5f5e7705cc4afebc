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
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <int O0, int O1>  // two 4-byte reads 4 * O0 and 4 * O1 bytes above addr (O0, O1 <= 255) into a register pair
__device__ __forceinline__ void lds_read2_b32(f32x2_t& v, unsigned addr) {
  asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(O0), "n"(O1));
}
__device__ __forceinline__ void lds_wait8x2(f32x2_t (&d)[8]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]));
}
__device__ __forceinline__ void lds_wait4x2(f32x2_t (&d)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
}
// Packed fp32 adds on register pairs (v_pk_add_f32, one issue slot for two adds; tools/probes/pk_add_probe.hip checks
// the operand-select / negate modifiers on the device).  Results are bit-identical to the scalar expressions.
__device__ __forceinline__ f32x2_t pk_add(f32x2_t a, f32x2_t b) {  // (a.x + b.x, a.y + b.y)
  f32x2_t r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x2_t pk_sub(f32x2_t a, f32x2_t b) {  // (a.x - b.x, a.y - b.y)
  f32x2_t r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x2_t pk_sub_add_x(f32x2_t a, f32x2_t b) {  // (a.x - b.x, a.y + b.x)
  f32x2_t r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x2_t pk_cross_sub(f32x2_t a, f32x2_t b) {  // (b.x - a.y, a.y - b.y)
  f32x2_t r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(b), "v"(a));
  return r;
}
__device__ __forceinline__ f32x2_t pk_sum_diff(f32x2_t a) {  // (a.x + a.y, a.x - a.y)
  f32x2_t r;
  asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(a));
  return r;
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
