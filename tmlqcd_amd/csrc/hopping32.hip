// fp32 instantiation of the stencil (one site per thread): spinor32 fields, float2 gauge copy.  See hopping.hip / hopping_impl.inc.
#include "hopping_common.h"

#define HOP_CTX_GAUGE(ctx) ((ctx)->gauge32)
#define HOP_CTX_GAUGE_READY(ctx) ((ctx)->gauge32_set)
#define HOP_CTX_OCC(ctx) ((ctx)->opt_occ32)
#define HOP_CTX_STG(ctx) ((ctx)->opt_stg32)
// The fp32 gauge copy pairs the link elements: per (parity, direction) four planes of float4 = elements (0,1) (2,3) (4,5) (6,7)
// and one plane of float2 = element 8, so a link is five loads per lane (four of them 16 bytes wide) instead of nine 8-byte ones.
// Private to this translation unit and tmhip_prepare_fp32's conversion kernel (mixed.hip).
// The fp32 spinor fields pair their components the same way: six planes of float4 = components (0,1) (2,3) ... (10,11)
// ([6][ns] float4; half-spinor faces [3][face] float4), so a spinor is six 16-byte accesses per lane instead of twelve 8-byte
// ones -- at 8 bytes per lane the stencil was bound by the number of memory instructions, not by bytes (profiles/r02_fp32.md).
// The same layout is read and written by the fp32 linalg / conversion kernels (mixed.hip, cg.hip).
#define HOP_GAUGE_PACKED 1
#define HOP_SPINOR_PACKED 1
namespace hop32 {
TMHIP_SCALAR_COMPLEX_OPS(v2f, float)
typedef float vf4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ void ld6(V2T *s, const ET *f, size_t stride, int j, int blk) {
  const vf4 *p = reinterpret_cast<const vf4 *>(f);
#pragma unroll
  for (int m = 0; m < 3; m++) {
    const vf4 *q = p + (size_t)(3 * blk + m) * stride + j;
    const vf4 v = NT ? __builtin_nontemporal_load(q) : *q;
    s[2 * m] = V2T{v.x, v.y}; s[2 * m + 1] = V2T{v.z, v.w};
  }
}
template <bool NT> __device__ __forceinline__ void st6(ET *f, size_t stride, int j, int blk, const V2T *s) {
  vf4 *p = reinterpret_cast<vf4 *>(f);
#pragma unroll
  for (int m = 0; m < 3; m++) {
    const vf4 v = vf4{s[2 * m].x, s[2 * m].y, s[2 * m + 1].x, s[2 * m + 1].y};
    vf4 *q = p + (size_t)(3 * blk + m) * stride + j;
    if (NT) __builtin_nontemporal_store(v, q);
    else *q = v;
  }
}
// system-scope (sc0 sc1) loads of the six components 6 blk .. 6 blk + 5: see hopping_common.h
__device__ __forceinline__ void ld6_fresh(V2T *s, const ET *f, size_t stride, int j, int blk) {
  const vf4 *p = reinterpret_cast<const vf4 *>(f);
#pragma unroll
  for (int m = 0; m < 3; m++) {
    const unsigned long long *q = reinterpret_cast<const unsigned long long *>(p + (size_t)(3 * blk + m) * stride + j);
    const unsigned long long w0 = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long w1 = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_memcpy(&s[2 * m], &w0, 8); __builtin_memcpy(&s[2 * m + 1], &w1, 8);
  }
}
// write-through (sc0 sc1) stores of the same six components, 16 bytes each: see hopping_common.h
__device__ __forceinline__ void st6_through(ET *f, size_t stride, int j, int blk, const V2T *s) {
  vf4 *p = reinterpret_cast<vf4 *>(f);
#pragma unroll
  for (int m = 0; m < 3; m++) {
    const vf4 v = vf4{s[2 * m].x, s[2 * m].y, s[2 * m + 1].x, s[2 * m + 1].y};
    vf4 *q = p + (size_t)(3 * blk + m) * stride + j;
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(q), "v"(v) : "memory");   // (s_nop: the store-data hazard the compiler cannot see inside the asm)
  }
}
constexpr int HOP_STAGE_BYTES = 6 * 64 * (int)sizeof(vf4);   // per wave: [6][64] float4
__device__ __forceinline__ void stage_put(unsigned char *region, int lid, const ET *f, size_t ns, int i) {
  vf4 *st = reinterpret_cast<vf4 *>(region);
  const vf4 *p = reinterpret_cast<const vf4 *>(f);
#pragma unroll
  for (int m = 0; m < 6; m++) st[m * 64 + lid] = p[(size_t)m * ns + i];
}
__device__ __forceinline__ void stage_get6(V2T *s, const unsigned char *region, int jl, int blk) {
  const vf4 *st = reinterpret_cast<const vf4 *>(region);
#pragma unroll
  for (int m = 0; m < 3; m++) {
    const vf4 v = st[(3 * blk + m) * 64 + jl];
    s[2 * m] = V2T{v.x, v.y}; s[2 * m + 1] = V2T{v.z, v.w};
  }
}
constexpr int HOP_SW = 6;     // word-wise access (one word = a float4 = two components), see hopping_common.h
typedef vf4 SWT;
__device__ __forceinline__ SWT sw_stage(const unsigned char *region, int jl, int w) { return reinterpret_cast<const vf4 *>(region)[w * 64 + jl]; }
__device__ __forceinline__ SWT sw_ld(const ET *f, size_t stride, int j, int w) { return reinterpret_cast<const vf4 *>(f)[(size_t)w * stride + j]; }
__device__ __forceinline__ void sw_unpack(V2T *s, int w, SWT v) { s[2 * w] = V2T{v.x, v.y}; s[2 * w + 1] = V2T{v.z, v.w}; }
#define HOP_SITES 1
#include "hopping_impl.inc"
#undef HOP_SITES
}  // namespace hop32
#undef HOP_GAUGE_PACKED
#undef HOP_SPINOR_PACKED

#undef HOP_CTX_OCC
#undef HOP_CTX_STG
#undef HOP_CTX_GAUGE
#undef HOP_CTX_GAUGE_READY
