// fp32 twins of the hot path for the mixed-precision CG:
//   Hopping_Matrix_32 / Qtm_pm_psi_32        operator/Hopping_Matrix_32.c:97-127, operator/tm_operators_32.c
//   assign_to_32 / assign_to_64              linalg/assign_to_32.c, assign_to_64.c
//   square_norm_32, scalar_prod_r_32, assign_add_mul_r_32, assign_mul_add_r_32   linalg/*_32.c
// SoA like the fp64 fields, with the twelve components paired into six float4 planes (96 B/site, 16-byte accesses); the
// stencil is the fp64 kernel instantiated for float2 arithmetic (hopping_impl.inc, hopping32.hip).  Reductions accumulate in double.
#include "tmhip_internal.h"

// fp64 field [12][ns] v2d  <->  fp32 field [6][ns] v4f (components 2m, 2m+1 paired); blockIdx.y = m
template <bool TO32>
__global__ __launch_bounds__(LA_BS) void convert_kernel(v4f *__restrict__ F, v2d *__restrict__ D, int ns, int N) {
  v4f *f = F + (size_t)blockIdx.y * ns;
  v2d *d0 = D + (size_t)(2 * blockIdx.y) * ns, *d1 = d0 + ns;
  const int base = blockIdx.x * LA_BS * LA_UNROLL + threadIdx.x;
#pragma unroll
  for (int u = 0; u < LA_UNROLL; u++) {
    const int i = base + u * LA_BS;
    if (i < N) {
      if (TO32) { const v2d a = d0[i], b = d1[i]; f[i] = v4f{(float)a.x, (float)a.y, (float)b.x, (float)b.y}; }
      else { const v4f a = f[i]; d0[i] = v2d{(double)a.x, (double)a.y}; d1[i] = v2d{(double)a.z, (double)a.w}; }
    }
  }
}

// P += (double) x      (assign_to_64 + add of mixed_cg_her.c:158-159 in one pass)
__global__ __launch_bounds__(LA_BS) void add_from32_kernel(v2d *__restrict__ P, const v4f *__restrict__ X, int ns, int N) {
  v2d *p0 = P + (size_t)(2 * blockIdx.y) * ns, *p1 = p0 + ns;
  const v4f *x = X + (size_t)blockIdx.y * ns;
  const int base = blockIdx.x * LA_BS * LA_UNROLL + threadIdx.x;
#pragma unroll
  for (int u = 0; u < LA_UNROLL; u++) {
    const int i = base + u * LA_BS;
    if (i < N) {
      const v4f a = x[i];
      const v2d b = p0[i], c = p1[i];
      p0[i] = v2d{b.x + (double)a.x, b.y + (double)a.y};
      p1[i] = v2d{c.x + (double)a.z, c.y + (double)a.w};
    }
  }
}

// fp64 gauge copy [2 * 8][9][gs] -> fp32 twin in the packed layout of hopping32.hip: per (parity, direction) the elements
// (0,1) (2,3) (4,5) (6,7) interleaved site by site (four float4 planes) and element 8 as a float2 plane
__global__ void gauge_to32_kernel(v2f *__restrict__ d, const v2d *__restrict__ s, size_t n, size_t gs) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const size_t pd = idx / (9 * gs), r = idx - pd * 9 * gs, e = r / gs, i = r - e * gs;
  const v2d a = s[idx];
  const size_t o = pd * 9 * gs + (e < 8 ? (e >> 1) * 2 * gs + 2 * i + (e & 1) : 8 * gs + i);
  d[o] = v2f{(float)a.x, (float)a.y};
}

__device__ __forceinline__ void block_reduce_store32(double v, double *partials) {
  __shared__ double wsum[LA_BS / 64];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < LA_BS / 64; k++) s += wsum[k];
    partials[blockIdx.y * gridDim.x + blockIdx.x] = s;
  }
}

// MODE 0: |S|^2   1: Re<S,R>
template <int MODE>
__global__ __launch_bounds__(LA_BS) void reduce32_kernel(const v4f *__restrict__ S, const v4f *__restrict__ R, int ns, int N,
                                                         double *partials) {
  const v4f *s = S + (size_t)blockIdx.y * ns;
  const v4f *r = MODE ? R + (size_t)blockIdx.y * ns : nullptr;
  double acc = 0.0;
  const int base = blockIdx.x * LA_BS * LA_UNROLL + threadIdx.x;
#pragma unroll
  for (int u = 0; u < LA_UNROLL; u++) {
    const int i = base + u * LA_BS;
    if (i < N) {
      const v4f a = s[i];
      const v4f b = MODE ? r[i] : a;
      acc += ((double)a.x * b.x + (double)a.y * b.y) + ((double)a.z * b.z + (double)a.w * b.w);
    }
  }
  block_reduce_store32(acc, partials);
}

// MODE 0: P += c Q   1: R = c R + S
template <int MODE>
__global__ __launch_bounds__(LA_BS) void stream32_kernel(v4f *__restrict__ X, const v4f *__restrict__ Y, float c, int ns, int N) {
  v4f *x = X + (size_t)blockIdx.y * ns;
  const v4f *y = Y + (size_t)blockIdx.y * ns;
  const int base = blockIdx.x * LA_BS * LA_UNROLL + threadIdx.x;
#pragma unroll
  for (int u = 0; u < LA_UNROLL; u++) {
    const int i = base + u * LA_BS;
    if (i < N) {
      const v4f a = x[i], b = y[i];
      x[i] = MODE == 0 ? a + c * b : c * a + b;
    }
  }
}

// host spinor32[n] (float AoS: [n][12] float2 = [n][6] float4) <-> device [6][ns] float4
__global__ __launch_bounds__(256) void aos_to_soa32_kernel(const v4f *__restrict__ aos, v4f *__restrict__ soa, int ns, int n) {
  const long tid = (long)blockIdx.x * 256 + threadIdx.x;
  if (tid >= 6L * n) return;
  soa[(size_t)(tid % 6) * ns + tid / 6] = aos[tid];
}
__global__ __launch_bounds__(256) void soa_to_aos32_kernel(const v4f *__restrict__ soa, v4f *__restrict__ aos, int ns, int n) {
  const long tid = (long)blockIdx.x * 256 + threadIdx.x;
  if (tid >= 6L * n) return;
  aos[tid] = soa[(size_t)(tid % 6) * ns + tid / 6];
}

static int need32(const tmhip_field *f, const char *who) {
  if (!f || f->kind != TMHIP_FIELD_EO || f->prec != 1) { fprintf(stderr, "[tmlqcd_hip] %s: needs a one-parity fp32 field\n", who); return 1; }
  return 0;
}
static int need64(const tmhip_field *f, const char *who) {
  if (!f || f->kind != TMHIP_FIELD_EO || f->prec != 0) { fprintf(stderr, "[tmlqcd_hip] %s: needs a one-parity fp64 field\n", who); return 1; }
  return 0;
}

int tmhip_prepare_fp32(tmhip_ctx *ctx) {
  if (!ctx->gauge_set) TMHIP_FAIL("fp32 operators called before tmhip_set_gauge");
  TMHIP_CHECK(hipSetDevice(ctx->device));
  const size_t n = (size_t)2 * 72 * ctx->gs;
  if (!ctx->gauge32) TMHIP_CHECK(hipMalloc((void **)&ctx->gauge32, n * sizeof(v2f)));
  if (!ctx->gauge32_set) {  // g_gauge_field_32 / copy_32: converted from the fp64 links (update_backward_gauge.c:244-312)
    hipLaunchKernelGGL(gauge_to32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->gauge32, ctx->gauge, n, (size_t)ctx->gs);
    TMHIP_CHECK(hipGetLastError());
    ctx->gauge32_set = true;
  }
  for (int i = 0; i < 2; i++) if (!ctx->scratch32[i] && tmhip_field_alloc_prec(ctx, TMHIP_FIELD_EO, 1, &ctx->scratch32[i])) return 1;
  for (int i = 0; i < 4; i++) if (!ctx->sf32[i] && tmhip_field_alloc_prec(ctx, TMHIP_FIELD_EO, 1, &ctx->sf32[i])) return 1;
  return 0;
}

extern "C" {

int tmhip_field_alloc32(tmhip_ctx *ctx, tmhip_field **out) {
  TMHIP_CHECK(hipSetDevice(ctx->device));
  return tmhip_field_alloc_prec(ctx, TMHIP_FIELD_EO, 1, out);
}

int tmhip_field_upload32(tmhip_ctx *ctx, tmhip_field *f, const void *host, int nsites) {
  if (need32(f, "tmhip_field_upload32") || !host) return 1;
  if (nsites <= 0 || nsites > ctx->Vh) TMHIP_FAIL("tmhip_field_upload32: nsites out of range");
  const size_t bytes = (size_t)nsites * 12 * sizeof(v2f);
  if (tmhip_stage_reserve(ctx, bytes)) return 1;
  TMHIP_CHECK(hipMemcpyAsync(ctx->stage, host, bytes, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(aos_to_soa32_kernel, dim3((unsigned)((6L * nsites + 255) / 256)), dim3(256), 0, ctx->stream,
                     (const v4f *)ctx->stage, (v4f *)f->d32, f->ns, nsites);
  TMHIP_CHECK(hipGetLastError());
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  return 0;
}

int tmhip_field_download32(tmhip_ctx *ctx, tmhip_field *f, void *host, int nsites) {
  if (need32(f, "tmhip_field_download32") || !host) return 1;
  if (nsites <= 0 || nsites > ctx->Vh) TMHIP_FAIL("tmhip_field_download32: nsites out of range");
  const size_t bytes = (size_t)nsites * 12 * sizeof(v2f);
  if (tmhip_stage_reserve(ctx, bytes)) return 1;
  hipLaunchKernelGGL(soa_to_aos32_kernel, dim3((unsigned)((6L * nsites + 255) / 256)), dim3(256), 0, ctx->stream,
                     (const v4f *)f->d32, (v4f *)ctx->stage, f->ns, nsites);
  TMHIP_CHECK(hipGetLastError());
  TMHIP_CHECK(hipMemcpyAsync(host, ctx->stage, bytes, hipMemcpyDeviceToHost, ctx->stream));
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  return 0;
}

/* linalg/assign_to_32.c */
int tmhip_assign_to_32(tmhip_ctx *ctx, tmhip_field *R32, tmhip_field *S64, int N) {
  if (need32(R32, "assign_to_32") || need64(S64, "assign_to_32")) return 1;
  LA_CHECK_N("assign_to_32", (void)0);
  hipLaunchKernelGGL(convert_kernel<true>, la_grid32(N), dim3(LA_BS), 0, ctx->stream, (v4f *)R32->d32, S64->d, R32->ns, N);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}
/* linalg/assign_to_64.c */
int tmhip_assign_to_64(tmhip_ctx *ctx, tmhip_field *R64, tmhip_field *S32, int N) {
  if (need64(R64, "assign_to_64") || need32(S32, "assign_to_64")) return 1;
  LA_CHECK_N("assign_to_64", (void)0);
  hipLaunchKernelGGL(convert_kernel<false>, la_grid32(N), dim3(LA_BS), 0, ctx->stream, (v4f *)S32->d32, R64->d, R64->ns, N);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}
int tmhip_add_from_32(tmhip_ctx *ctx, tmhip_field *P64, tmhip_field *X32, int N) {
  if (need64(P64, "add_from_32") || need32(X32, "add_from_32")) return 1;
  LA_CHECK_N("add_from_32", (void)0);
  hipLaunchKernelGGL(add_from32_kernel, la_grid32(N), dim3(LA_BS), 0, ctx->stream, P64->d, (const v4f *)X32->d32, P64->ns, N);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}

/* linalg/square_norm_32.c, scalar_prod_r_32.c (double accumulation) */
int tmhip_square_norm_32(tmhip_ctx *ctx, tmhip_field *P, int N, int parallel, double *out) {
  if (need32(P, "square_norm_32")) return 1;
  LA_CHECK_N("square_norm_32", *out = 0.0);
  const dim3 g = la_grid32(N);
  hipLaunchKernelGGL(reduce32_kernel<0>, g, dim3(LA_BS), 0, ctx->stream, (const v4f *)P->d32, (const v4f *)nullptr, P->ns, N, ctx->partials);
  return tmhip_reduce_finish(ctx, g.x * g.y, parallel, out);
}
int tmhip_scalar_prod_r_32(tmhip_ctx *ctx, tmhip_field *S, tmhip_field *R, int N, int parallel, double *out) {
  if (need32(S, "scalar_prod_r_32") || need32(R, "scalar_prod_r_32")) return 1;
  LA_CHECK_N("scalar_prod_r_32", *out = 0.0);
  const dim3 g = la_grid32(N);
  hipLaunchKernelGGL(reduce32_kernel<1>, g, dim3(LA_BS), 0, ctx->stream, (const v4f *)S->d32, (const v4f *)R->d32, S->ns, N, ctx->partials);
  return tmhip_reduce_finish(ctx, g.x * g.y, parallel, out);
}
/* linalg/assign_add_mul_r_32.c, assign_mul_add_r_32.c */
int tmhip_assign_add_mul_r_32(tmhip_ctx *ctx, tmhip_field *P, tmhip_field *Q, float c, int N) {
  if (need32(P, "assign_add_mul_r_32") || need32(Q, "assign_add_mul_r_32")) return 1;
  LA_CHECK_N("assign_add_mul_r_32", (void)0);
  hipLaunchKernelGGL(stream32_kernel<0>, la_grid32(N), dim3(LA_BS), 0, ctx->stream, (v4f *)P->d32, (const v4f *)Q->d32, c, P->ns, N);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}
int tmhip_assign_mul_add_r_32(tmhip_ctx *ctx, tmhip_field *R, float c, tmhip_field *S, int N) {
  if (need32(R, "assign_mul_add_r_32") || need32(S, "assign_mul_add_r_32")) return 1;
  LA_CHECK_N("assign_mul_add_r_32", (void)0);
  hipLaunchKernelGGL(stream32_kernel<1>, la_grid32(N), dim3(LA_BS), 0, ctx->stream, (v4f *)R->d32, (const v4f *)S->d32, c, R->ns, N);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}

/* operator/Hopping_Matrix_32.c:97-127 */
int tmhip_hopping_matrix_32(tmhip_ctx *ctx, int ieo, tmhip_field *l, tmhip_field *k) {
  if (need32(l, "Hopping_Matrix_32") || need32(k, "Hopping_Matrix_32")) return 1;
  if (tmhip_prepare_fp32(ctx)) return 1;
  return tmhip_launch_hopping32(ctx, ieo, l->d32, k->d32, nullptr, EPI_STORE, 0, 0, true);
}

/* operator/tm_operators_32.c Qtm_pm_psi_32: same algebra as Qtm_pm_psi, twists fused into the stencil epilogues */
int tmhip_Qtm_pm_psi_32(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k) {
  if (need32(l, "Qtm_pm_psi_32") || need32(k, "Qtm_pm_psi_32")) return 1;
  if (tmhip_prepare_fp32(ctx)) return 1;
  const double mu = ctx->mu, nrm = 1. / (1. + mu * mu);
  v2f *s0 = ctx->scratch32[0]->d32, *s1 = ctx->scratch32[1]->d32;
  return tmhip_launch_hopping32(ctx, TMHIP_EO, s1, k->d32, nullptr, EPI_TM_TIMES, nrm, nrm * mu, true) ||
         tmhip_launch_hopping32(ctx, TMHIP_OE, s0, s1, k->d32, EPI_TM_SUB_G5, 1., -mu, true) ||
         tmhip_launch_hopping32(ctx, TMHIP_EO, s1, s0, nullptr, EPI_TM_TIMES, nrm, -nrm * mu, true) ||
         tmhip_launch_hopping32(ctx, TMHIP_OE, l->d32, s1, s0, EPI_TM_SUB_G5, 1., mu, true);
}

}  // extern "C"

// ---- fp32 site-diagonal twists on HOST arrays of any length (the _32 instances of operator/mul_one_pm_imu_inv_body.c and
// operator/mul_one_pm_imu_sub_mul_body.c that tm_operators.c:8-20,35-47 generates; solver/Msap.c calls them on domain blocks) ----
// The reference's AoS spinor32 is processed as it is: element e of a site is (spin e/3, colour e%3), spin 2,3 take conj(z).
__global__ __launch_bounds__(256) void diag32_aos_kernel(v2f *__restrict__ l, const v2f *__restrict__ k, const v2f *__restrict__ j, float zre, float zim, size_t n) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const float zi = (idx % 12) >= 6 ? -zim : zim;
  const v2f a = k[idx];
  v2f r = v2f{zre * a.x - zi * a.y, zre * a.y + zi * a.x};
  if (j) r -= j[idx];
  l[idx] = r;
}

extern "C" int tmhip_diag32_host(tmhip_ctx *ctx, void *l, const void *k, const void *j, double zre, double zim, int N) {
  if (!l || !k) TMHIP_FAIL("tmhip_diag32_host: null argument");
  if (N < 0) TMHIP_FAIL("tmhip_diag32_host: N = %d", N);
  if (N == 0) return 0;
  TMHIP_CHECK(hipSetDevice(ctx->device));
  const size_t n = (size_t)N * 12, bytes = n * sizeof(v2f);
  if (tmhip_stage_reserve(ctx, (j ? 3 : 2) * bytes)) return 1;
  v2f *dl = (v2f *)ctx->stage, *dk = dl + n, *dj = j ? dk + n : nullptr;
  TMHIP_CHECK(hipMemcpyAsync(dk, k, bytes, hipMemcpyHostToDevice, ctx->stream));
  if (j) TMHIP_CHECK(hipMemcpyAsync(dj, j, bytes, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(diag32_aos_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, dl, (const v2f *)dk, (const v2f *)dj, (float)zre, (float)zim, n);
  TMHIP_CHECK(hipGetLastError());
  TMHIP_CHECK(hipMemcpyAsync(l, dl, bytes, hipMemcpyDeviceToHost, ctx->stream));
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  return 0;
}
