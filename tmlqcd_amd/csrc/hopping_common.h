// Shared prelude of the three stencil translation units (hopping.hip: fp64, hopping32.hip: fp32, hopping32p.hip: fp32 with two
// sites per thread).  Each includes hopping_impl.inc once inside its own namespace; splitting them lets make compile them in parallel.
#pragma once
#ifndef HOP_GAUGE_NT
#define HOP_GAUGE_NT true   /* gauge links loaded non-temporally (A/B builds: -DHOP_GAUGE_NT=false) */
#endif
#include "tmhip_internal.h"

typedef int v4i __attribute__((ext_vector_type(4)));

// ---- one site per thread: the value type is the memory element itself -------------------------------------------
#define TMHIP_SCALAR_COMPLEX_OPS(ETYPE, RTYPE)                                                                   \
  typedef ETYPE ET;                                                                                              \
  typedef ETYPE V2T;                                                                                             \
  typedef RTYPE RT;                                                                                              \
  template <bool NT> __device__ __forceinline__ V2T ldc(const ET *p) { if (NT) return __builtin_nontemporal_load(p); return *p; } \
  template <bool NT> __device__ __forceinline__ void stc(ET *p, V2T v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; } \
  __device__ __forceinline__ V2T czero() { return V2T{0, 0}; }                                                   \
  __device__ __forceinline__ V2T cbcast(double re, double im) { return V2T{(RT)re, (RT)im}; }                    \
  __device__ __forceinline__ double cdotd(V2T w, V2T r) { return (double)w.x * (double)r.x + (double)w.y * (double)r.y; }


// Spinor-field access of the stencil kernels, one plane per component: field f = [12][stride] complex values (faces: [6][stride]).
// ld6 / st6 move the six components 6*blk .. 6*blk+5 of site j; the per-wave LDS staging region is [12][64].
#define TMHIP_SPINOR_IO_PLANES                                                                                                  \
  template <bool NT> __device__ __forceinline__ void ld6(V2T *s, const ET *f, size_t stride, int j, int blk) {                    \
    _Pragma("unroll") for (int c = 0; c < 6; c++) s[c] = ldc<NT>(f + (size_t)(6 * blk + c) * stride + j);                         \
  }                                                                                                                               \
  template <bool NT> __device__ __forceinline__ void st6(ET *f, size_t stride, int j, int blk, const V2T *s) {                    \
    _Pragma("unroll") for (int c = 0; c < 6; c++) stc<NT>(f + (size_t)(6 * blk + c) * stride + j, s[c]);                          \
  }                                                                                                                               \
  /* the same six components with system-scope loads (sc0 sc1: never served by this CU's L1): data another kernel -- of this GPU,  \
     or a neighbour's through an IPC mapping -- has just written while this one was already running (the exchanged faces), without  \
     paying an L1 invalidate for the whole CU */                                                                                   \
  __device__ __forceinline__ void ld6_fresh(V2T *s, const ET *f, size_t stride, int j, int blk) {                                 \
    _Pragma("unroll") for (int c = 0; c < 6; c++) {                                                                               \
      const ET *p = f + (size_t)(6 * blk + c) * stride + j;                                                                       \
      if constexpr (sizeof(ET) == 16) {                                                                                           \
        const double *q = reinterpret_cast<const double *>(p);                                                                    \
        s[c].x = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);                                                \
        s[c].y = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);                                            \
      } else {                                                                                                                    \
        const unsigned long long w = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); \
        __builtin_memcpy(&s[c], &w, 8);                                                                                           \
      }                                                                                                                           \
    }                                                                                                                             \
  }                                                                                                                               \
  /* ... and write-through stores (sc0 sc1: the bytes leave for memory, the line is dropped from this XCD's L2): data handed to      \
     another kernel that runs beside this one; the storing wave drains them (s_waitcnt vmcnt(0)) before it signals.  One 16-byte    \
     store per component: 8-byte sc1 stores are separate fabric writes (the stencil's boundary waves took 94 us instead of 43) */   \
  __device__ __forceinline__ void st6_through(ET *f, size_t stride, int j, int blk, const V2T *s) {                               \
    static_assert(sizeof(ET) == 16, "one 16-byte element per component");                                                         \
    _Pragma("unroll") for (int c = 0; c < 6; c++) {                                                                               \
      ET *p = f + (size_t)(6 * blk + c) * stride + j;                                                                             \
      v4i w;                                                                                                                      \
      __builtin_memcpy(&w, &s[c], 16);                                                                                            \
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");                                     \
    }                                                                                                                             \
  }                                                                                                                               \
  constexpr int HOP_STAGE_BYTES = 12 * 64 * (int)sizeof(V2T); /* per wave */                                                      \
  __device__ __forceinline__ void stage_put(unsigned char *region, int lid, const ET *f, size_t ns, int i) {                      \
    V2T *st = reinterpret_cast<V2T *>(region);                                                                                    \
    _Pragma("unroll") for (int c = 0; c < 12; c++) st[c * 64 + lid] = ldc<false>(f + (size_t)c * ns + i);                         \
  }                                                                                                                               \
  __device__ __forceinline__ void stage_get6(V2T *s, const unsigned char *region, int jl, int blk) {                              \
    const V2T *st = reinterpret_cast<const V2T *>(region);                                                                        \
    _Pragma("unroll") for (int c = 0; c < 6; c++) s[c] = st[(6 * blk + c) * 64 + jl];                                             \
  }                                                                                                                             \
  /* word-wise access for the per-lane choice between the staging region and memory (one word = one component here) */          \
  constexpr int HOP_SW = 12;                                                                                                      \
  typedef V2T SWT;                                                                                                                \
  __device__ __forceinline__ SWT sw_stage(const unsigned char *region, int jl, int w) { return reinterpret_cast<const V2T *>(region)[w * 64 + jl]; } \
  __device__ __forceinline__ SWT sw_ld(const ET *f, size_t stride, int j, int w) { return ldc<false>(f + (size_t)w * stride + j); } \
  __device__ __forceinline__ void sw_unpack(V2T *s, int w, SWT v) { s[w] = v; }

// entry points of the fp32 instantiations (defined in hopping32.hip / hopping32p.hip by hopping_impl.inc)
#define TMHIP_DECLARE_HOP32(NS)                                                                                                \
  namespace NS {                                                                                                               \
  int launch_hopping(tmhip_ctx *ctx, int ieo, v2f *out, const v2f *in, const v2f *p, int epi, double cre, double cim, int comm, const v2f *cw); \
  int launch_hopping_dot(tmhip_ctx *ctx, int ieo, v2f *out, const v2f *in, const v2f *p, const v2f *dotv, double cre, double cim, \
                         int *npartials, int mode, v2f *resid, const double *scal, const v2f *cw, int chained, const HopSelfAlpha *self = nullptr);                              \
  }
