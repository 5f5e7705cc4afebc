#include <hip/hip_runtime.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// (a.x - b.x, a.y + b.x)
__device__ __forceinline__ f32x2 pk_va(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// (b.x - a.y, a.y - b.y)  with operands (b, a)
__device__ __forceinline__ f32x2 pk_vb(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(b), "v"(a));
  return r;
}
__global__ void k(const f32x2* in, f32x2* out) {
  f32x2 a = in[threadIdx.x * 2], b = in[threadIdx.x * 2 + 1];
  out[threadIdx.x * 4 + 0] = pk_sub(a, b);
  out[threadIdx.x * 4 + 1] = pk_add(a, b);
  out[threadIdx.x * 4 + 2] = pk_va(a, b);
  out[threadIdx.x * 4 + 3] = pk_vb(a, b);
}
int main() {
  f32x2 h[128], *d, *o, r[256];
  for (int i = 0; i < 128; ++i) h[i] = f32x2{float(i) + 0.25f, float(3 * i) - 0.5f};
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, 1, 64, 0, 0, d, o);
  hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 64; ++t) {
    f32x2 a = h[2 * t], b = h[2 * t + 1];
    f32x2 e0 = {a.x - b.x, a.y - b.y}, e1 = {a.x + b.x, a.y + b.y}, e2 = {a.x - b.x, a.y + b.x}, e3 = {b.x - a.y, a.y - b.y};
    f32x2 ex[4] = {e0, e1, e2, e3};
    for (int q = 0; q < 4; ++q) if (r[4 * t + q].x != ex[q].x || r[4 * t + q].y != ex[q].y) { if (bad < 5) printf("t %d q %d got %f %f want %f %f\n", t, q, r[4*t+q].x, r[4*t+q].y, ex[q].x, ex[q].y); ++bad; }
  }
  printf("bad %d\n", bad);
  return bad != 0;
}
