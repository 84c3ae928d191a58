// Shared prelude of the three stencil translation units (hopping.hip: fp64, hopping32.hip: fp32, hopping32p.hip: fp32 with two
// sites per thread).  Each includes hopping_impl.inc once inside its own namespace; splitting them lets make compile them in parallel.
#pragma once
#include "tmhip_internal.h"

typedef int v4i __attribute__((ext_vector_type(4)));

// ---- one site per thread: the value type is the memory element itself -------------------------------------------
#define TMHIP_SCALAR_COMPLEX_OPS(ETYPE, RTYPE)                                                                   \
  typedef ETYPE ET;                                                                                              \
  typedef ETYPE V2T;                                                                                             \
  typedef RTYPE RT;                                                                                              \
  template <bool NT> __device__ __forceinline__ V2T ldc(const ET *p) { if (NT) return __builtin_nontemporal_load(p); return *p; } \
  template <bool NT> __device__ __forceinline__ void stc(ET *p, V2T v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; } \
  __device__ __forceinline__ V2T ldc2(const ET *p0, const ET *) { return *p0; }                                  \
  __device__ __forceinline__ V2T czero() { return V2T{0, 0}; }                                                   \
  __device__ __forceinline__ V2T cbcast(double re, double im) { return V2T{(RT)re, (RT)im}; }                    \
  __device__ __forceinline__ double cdotd(V2T w, V2T r) { return (double)w.x * (double)r.x + (double)w.y * (double)r.y; }


// entry points of the fp32 instantiations (defined in hopping32.hip / hopping32p.hip by hopping_impl.inc)
#define TMHIP_DECLARE_HOP32(NS)                                                                                                \
  namespace NS {                                                                                                               \
  int launch_hopping(tmhip_ctx *ctx, int ieo, v2f *out, const v2f *in, const v2f *p, int epi, double cre, double cim, int comm, const v2f *cw); \
  int launch_hopping_dot(tmhip_ctx *ctx, int ieo, v2f *out, const v2f *in, const v2f *p, const v2f *dotv, double cre, double cim, \
                         int *npartials, int mode, v2f *resid, const double *scal, const v2f *cw, int chained_in);                              \
  }
