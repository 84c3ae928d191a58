// fp32 instantiation of the stencil with two sites per thread (16-byte accesses; option fp32_pairs).  See hopping.hip / hopping_impl.inc.
#include "hopping_common.h"

#define HOP_CTX_GAUGE(ctx) ((ctx)->gauge32)
#define HOP_CTX_GAUGE_READY(ctx) ((ctx)->gauge32_set)
#define HOP_CTX_OCC(ctx) ((ctx)->opt_occ32)
// ---- fp32, two sites per thread: x = (re of site i, re of site i+1), y = (im, im) ----------------------------------
namespace hop32p {
typedef v2f ET;
typedef float vf2 __attribute__((ext_vector_type(2)));
typedef float vf4 __attribute__((ext_vector_type(4)));
struct V2T {
  vf2 x, y;
  __device__ __forceinline__ V2T &operator+=(const V2T &o) { x += o.x; y += o.y; return *this; }
  __device__ __forceinline__ V2T &operator-=(const V2T &o) { x -= o.x; y -= o.y; return *this; }
};
__device__ __forceinline__ V2T operator+(V2T a, V2T b) { return V2T{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ V2T operator-(V2T a, V2T b) { return V2T{a.x - b.x, a.y - b.y}; }
typedef vf2 RT;
template <bool NT> __device__ __forceinline__ V2T ldc(const ET *p) {  // sites i, i+1: one 16-byte access
  const vf4 v = NT ? __builtin_nontemporal_load((const vf4 *)p) : *(const vf4 *)p;
  return V2T{vf2{v.x, v.z}, vf2{v.y, v.w}};
}
template <bool NT> __device__ __forceinline__ void stc(ET *p, V2T v) {
  const vf4 w = vf4{v.x.x, v.y.x, v.x.y, v.y.y};
  if (NT) __builtin_nontemporal_store(w, (vf4 *)p);
  else *(vf4 *)p = w;
}
__device__ __forceinline__ V2T ldc2(const ET *p0, const ET *p1) {
  const ET a = *p0, b = *p1;
  return V2T{vf2{a.x, b.x}, vf2{a.y, b.y}};
}
__device__ __forceinline__ V2T czero() { return V2T{vf2{0.f, 0.f}, vf2{0.f, 0.f}}; }
__device__ __forceinline__ V2T cbcast(double re, double im) { return V2T{vf2{(float)re, (float)re}, vf2{(float)im, (float)im}}; }
__device__ __forceinline__ double cdotd(V2T w, V2T r) {
  return (double)w.x.x * r.x.x + (double)w.y.x * r.y.x + (double)w.x.y * r.x.y + (double)w.y.y * r.y.y;
}
#define HOP_SITES 2
#define HOP_COMPLEX_COMPONENTWISE 1
#include "hopping_impl.inc"
#undef HOP_SITES
#undef HOP_COMPLEX_COMPONENTWISE
}  // namespace hop32p
#undef HOP_CTX_OCC
#undef HOP_CTX_GAUGE
#undef HOP_CTX_GAUGE_READY

