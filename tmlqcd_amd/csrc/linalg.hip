// Spinor linear algebra + site-diagonal twisted-mass operators on SoA device fields (gfx950).
//
// Reference semantics: linalg/{square_norm,scalar_prod_r,assign_add_mul_r,assign_mul_add_r,
// assign_mul_add_r_and_square,diff,assign}.c, gamma.c:77-98 and the site-diagonal ops of
// operator/tm_operators.c:587-858.  All are HBM-bound streams: 16 B/lane loads of one SoA
// plane, grid.y = the 12 (spin,colour) planes so no index arithmetic is needed.
//
// Reductions: per-thread partial -> 64-lane wavefront butterfly (__shfl_xor, lowered to
// DPP/ds_bpermute) -> LDS across the waves of the block -> one partial per block ->
// single-block fixed-order final pass (bitwise reproducible run to run; no atomics).
// The reference compensates a *sequential* sum with Kahan (square_norm.c:275-296); the
// pairwise tree used here has an error bound of O(log N) ulp, tighter than the
// sequential one, so no compensation is needed to stay within 1e-13 relative.
#include "tmhip_internal.h"


__device__ __forceinline__ double wave_reduce(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ void block_reduce_store(double v, double *partials) {
  __shared__ double wsum[LA_BS / 64];
  v = wave_reduce(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) wsum[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < LA_BS / 64; k++) s += wsum[k];
    partials[blockIdx.y * gridDim.x + blockIdx.x] = s;
  }
}

__global__ __launch_bounds__(LA_BS) void sqnorm_kernel(const v2d *__restrict__ P, int ns, int N, double *partials) {
  const v2d *p = P + (size_t)blockIdx.y * ns;
  double acc = 0.0;
  const int base = blockIdx.x * LA_BS * LA_UNROLL + threadIdx.x;
#pragma unroll
  for (int u = 0; u < LA_UNROLL; u++) {
    const int i = base + u * LA_BS;
    if (i < N) { const v2d a = p[i]; acc += a.x * a.x + a.y * a.y; }
  }
  block_reduce_store(acc, partials);
}

// Re <S,R> = sum Re(r * conj(s))   (scalar_prod_r.c:161-165)
__global__ __launch_bounds__(LA_BS) void dotr_kernel(const v2d *__restrict__ S, const v2d *__restrict__ R, int ns, int N,
                                                     double *partials) {
  const v2d *s = S + (size_t)blockIdx.y * ns, *r = R + (size_t)blockIdx.y * ns;
  double acc = 0.0;
  const int base = blockIdx.x * LA_BS * LA_UNROLL + threadIdx.x;
#pragma unroll
  for (int u = 0; u < LA_UNROLL; u++) {
    const int i = base + u * LA_BS;
    if (i < N) { const v2d a = s[i], b = r[i]; acc += a.x * b.x + a.y * b.y; }
  }
  block_reduce_store(acc, partials);
}

// R = c R + S, returns |R|^2   (assign_mul_add_r_and_square.c:165-196)
__global__ __launch_bounds__(LA_BS) void xpay_sq_kernel(v2d *__restrict__ R, double c, const v2d *__restrict__ S, int ns, int N,
                                                        double *partials) {
  v2d *r = R + (size_t)blockIdx.y * ns;
  const v2d *s = S + (size_t)blockIdx.y * ns;
  double acc = 0.0;
  const int base = blockIdx.x * LA_BS * LA_UNROLL + threadIdx.x;
#pragma unroll
  for (int u = 0; u < LA_UNROLL; u++) {
    const int i = base + u * LA_BS;
    if (i < N) {
      v2d a = r[i];
      const v2d b = s[i];
      a = v2d{c * a.x + b.x, c * a.y + b.y};
      r[i] = a;
      acc += a.x * a.x + a.y * a.y;
    }
  }
  block_reduce_store(acc, partials);
}

__global__ __launch_bounds__(256) void reduce_final_kernel(const double *__restrict__ partials, int n, double *out) {
  __shared__ double sm[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += partials[i];
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = sm[0];
}

// MODE 0: P += c Q   1: R = c R + S   2: Q = R - S   3: R = S   4: Q = R + S   5: R = c S
template <int MODE>
__global__ __launch_bounds__(LA_BS) void stream_kernel(v2d *__restrict__ X, const v2d *__restrict__ Y, const v2d *__restrict__ Z,
                                                       double c, int ns, int N) {
  v2d *x = X + (size_t)blockIdx.y * ns;
  const v2d *y = Y + (size_t)blockIdx.y * ns;
  const v2d *z = Z ? Z + (size_t)blockIdx.y * ns : nullptr;
  const int base = blockIdx.x * LA_BS * LA_UNROLL + threadIdx.x;
#pragma unroll
  for (int u = 0; u < LA_UNROLL; u++) {
    const int i = base + u * LA_BS;
    if (i < N) {
      if (MODE == 0) { const v2d a = x[i], b = y[i]; x[i] = v2d{a.x + c * b.x, a.y + c * b.y}; }
      if (MODE == 1) { const v2d a = x[i], b = y[i]; x[i] = v2d{c * a.x + b.x, c * a.y + b.y}; }
      if (MODE == 2) { x[i] = y[i] - z[i]; }
      if (MODE == 3) { x[i] = y[i]; }
      if (MODE == 4) { x[i] = y[i] + z[i]; }
      if (MODE == 5) { const v2d b = y[i]; x[i] = v2d{c * b.x, c * b.y}; }
    }
  }
}

// l = sigma_c * ( zc (.) k - beta * j ),  zc = z for spin 0,1 and conj(z) for spin 2,3,
// sigma_c = -1 on spin 2,3 when g5 is set.  Covers mul_one_pm_imu_inv, assign_mul_one_pm_imu[_inv],
// mul_one_pm_imu_sub_mul[_gamma5], gamma5.  Element-wise => alias-safe for l==k, l==j.
__global__ __launch_bounds__(LA_BS) void diag_kernel(v2d *L, const v2d *K, const v2d *J, double zre, double zim, int beta, int g5,
                                                     int ns, int N) {
  const int comp = blockIdx.y;
  const bool lower = comp >= 6;
  const double zi = lower ? -zim : zim;
  const double sg = (lower && g5) ? -1.0 : 1.0;
  v2d *l = L + (size_t)comp * ns;
  const v2d *k = K + (size_t)comp * ns;
  const v2d *j = J ? J + (size_t)comp * ns : nullptr;
  const int base = blockIdx.x * LA_BS * LA_UNROLL + threadIdx.x;
#pragma unroll
  for (int u = 0; u < LA_UNROLL; u++) {
    const int i = base + u * LA_BS;
    if (i < N) {
      const v2d a = k[i];
      v2d r = v2d{zre * a.x - zi * a.y, zre * a.y + zi * a.x};
      if (beta) r -= j[i];
      l[i] = v2d{sg * r.x, sg * r.y};
    }
  }
}


static int check_eo(const tmhip_field *f, const char *who) {
  if (!f || f->kind != TMHIP_FIELD_EO || f->prec != 0) { fprintf(stderr, "[tmlqcd_hip] %s: needs a one-parity (EO) field\n", who); return 1; }
  return 0;
}

int tmhip_reduce_finish(tmhip_ctx *ctx, int nblocks, int parallel, double *out) {
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, ctx->stream, ctx->partials, nblocks, ctx->result_dev);
  if (parallel && tmhip_reduce_over_ranks(ctx)) {
    // MPI_Allreduce(..., MPI_SUM) of the reference (square_norm.c:314): one double over RCCL
    if (ctx->direct.on && ctx->direct.sums_on) { if (tmhip_direct_allreduce(ctx, ctx->result_dev)) return 1; }
    else {
      if (ctx->shm) { if (tmhip_shm_allreduce(ctx, ctx->stream, ctx->result_dev, 1)) return 1; }
      else { TMHIP_NCCL_CHECK(ncclAllReduce(ctx->result_dev, ctx->result_dev, 1, ncclDouble, ncclSum, ctx->comm_red, ctx->stream)); }
    }
  }
  TMHIP_CHECK(hipMemcpyAsync(ctx->result_host, ctx->result_dev, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  *out = *ctx->result_host;
  return tmhip_check_async_error(ctx);
}

extern "C" {

int tmhip_square_norm(tmhip_ctx *ctx, tmhip_field *P, int N, int parallel, double *out) {
  if (check_eo(P, "square_norm")) return 1;
  LA_CHECK_N("square_norm", *out = 0.0);
  const dim3 g = la_grid(N);
  hipLaunchKernelGGL(sqnorm_kernel, g, dim3(LA_BS), 0, ctx->stream, P->d, P->ns, N, ctx->partials);
  return tmhip_reduce_finish(ctx, g.x * g.y, parallel, out);
}

int tmhip_scalar_prod_r(tmhip_ctx *ctx, tmhip_field *S, tmhip_field *R, int N, int parallel, double *out) {
  if (check_eo(S, "scalar_prod_r") || check_eo(R, "scalar_prod_r")) return 1;
  LA_CHECK_N("scalar_prod_r", *out = 0.0);
  const dim3 g = la_grid(N);
  hipLaunchKernelGGL(dotr_kernel, g, dim3(LA_BS), 0, ctx->stream, S->d, R->d, S->ns, N, ctx->partials);
  return tmhip_reduce_finish(ctx, g.x * g.y, parallel, out);
}

int tmhip_assign_mul_add_r_and_square(tmhip_ctx *ctx, tmhip_field *R, double c, tmhip_field *S, int N, int parallel,
                                      double *out) {
  if (check_eo(R, "assign_mul_add_r_and_square") || check_eo(S, "assign_mul_add_r_and_square")) return 1;
  LA_CHECK_N("assign_mul_add_r_and_square", *out = 0.0);
  const dim3 g = la_grid(N);
  hipLaunchKernelGGL(xpay_sq_kernel, g, dim3(LA_BS), 0, ctx->stream, R->d, c, S->d, R->ns, N, ctx->partials);
  return tmhip_reduce_finish(ctx, g.x * g.y, parallel, out);
}

int tmhip_assign_add_mul_r(tmhip_ctx *ctx, tmhip_field *P, tmhip_field *Q, double c, int N) {
  if (check_eo(P, "assign_add_mul_r") || check_eo(Q, "assign_add_mul_r")) return 1;
  LA_CHECK_N("assign_add_mul_r", (void)0);
  hipLaunchKernelGGL(stream_kernel<0>, la_grid(N), dim3(LA_BS), 0, ctx->stream, P->d, Q->d, (const v2d *)nullptr, c, P->ns, N);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}

int tmhip_assign_mul_add_r(tmhip_ctx *ctx, tmhip_field *R, double c, tmhip_field *S, int N) {
  if (check_eo(R, "assign_mul_add_r") || check_eo(S, "assign_mul_add_r")) return 1;
  LA_CHECK_N("assign_mul_add_r", (void)0);
  hipLaunchKernelGGL(stream_kernel<1>, la_grid(N), dim3(LA_BS), 0, ctx->stream, R->d, S->d, (const v2d *)nullptr, c, R->ns, N);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}

int tmhip_diff(tmhip_ctx *ctx, tmhip_field *Q, tmhip_field *R, tmhip_field *S, int N) {
  if (check_eo(Q, "diff") || check_eo(R, "diff") || check_eo(S, "diff")) return 1;
  LA_CHECK_N("diff", (void)0);
  hipLaunchKernelGGL(stream_kernel<2>, la_grid(N), dim3(LA_BS), 0, ctx->stream, Q->d, R->d, S->d, 0.0, Q->ns, N);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}

int tmhip_add(tmhip_ctx *ctx, tmhip_field *Q, tmhip_field *R, tmhip_field *S, int N) {
  if (check_eo(Q, "add") || check_eo(R, "add") || check_eo(S, "add")) return 1;
  LA_CHECK_N("add", (void)0);
  hipLaunchKernelGGL(stream_kernel<4>, la_grid(N), dim3(LA_BS), 0, ctx->stream, Q->d, R->d, S->d, 0.0, Q->ns, N);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}

int tmhip_mul_r(tmhip_ctx *ctx, tmhip_field *R, double c, tmhip_field *S, int N) {
  if (check_eo(R, "mul_r") || check_eo(S, "mul_r")) return 1;
  LA_CHECK_N("mul_r", (void)0);
  hipLaunchKernelGGL(stream_kernel<5>, la_grid(N), dim3(LA_BS), 0, ctx->stream, R->d, S->d, (const v2d *)nullptr, c, R->ns, N);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}

int tmhip_assign(tmhip_ctx *ctx, tmhip_field *R, tmhip_field *S, int N) {
  if (check_eo(R, "assign") || check_eo(S, "assign")) return 1;
  LA_CHECK_N("assign", (void)0);
  if (R->d == S->d) return 0;
  hipLaunchKernelGGL(stream_kernel<3>, la_grid(N), dim3(LA_BS), 0, ctx->stream, R->d, S->d, (const v2d *)nullptr, 0.0, R->ns, N);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}

static int launch_diag(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k, tmhip_field *j, double zre, double zim, int beta,
                       int g5, int N) {
  if (check_eo(l, "diag") || check_eo(k, "diag") || (j && check_eo(j, "diag"))) return 1;
  LA_CHECK_N("site-diagonal operator", (void)0);
  hipLaunchKernelGGL(diag_kernel, la_grid(N), dim3(LA_BS), 0, ctx->stream, l->d, k->d, j ? j->d : (const v2d *)nullptr, zre,
                     zim, beta, g5, l->ns, N);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}

/* mul_one_pm_imu_inv_body.c:1-41 : z = nrm (1 -+ i mu) for sign = +-1 */
int tmhip_mul_one_pm_imu_inv(tmhip_ctx *ctx, tmhip_field *l, double _sign, int N) {
  const double nrm = 1. / (1. + ctx->mu * ctx->mu), sign = _sign < 0. ? 1. : -1.;
  return launch_diag(ctx, l, l, nullptr, nrm, sign * nrm * ctx->mu, 0, 0, N);
}
/* mul_one_pm_imu_inv_body.c:43-80 */
int tmhip_assign_mul_one_pm_imu_inv(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k, double _sign, int N) {
  const double nrm = 1. / (1. + ctx->mu * ctx->mu), sign = _sign < 0. ? 1. : -1.;
  return launch_diag(ctx, l, k, nullptr, nrm, sign * nrm * ctx->mu, 0, 0, N);
}
/* tm_operators.c:669-720 : z = 1 +- i mu */
int tmhip_assign_mul_one_pm_imu(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k, double _sign, int N) {
  const double sign = _sign < 0. ? -1. : 1.;
  return launch_diag(ctx, l, k, nullptr, 1., sign * ctx->mu, 0, 0, N);
}
/* tm_operators.c:627-667 */
int tmhip_mul_one_pm_imu(tmhip_ctx *ctx, tmhip_field *l, double _sign) {
  const double sign = _sign < 0. ? -1. : 1.;
  return launch_diag(ctx, l, l, nullptr, 1., sign * ctx->mu, 0, 0, ctx->Vh);
}
/* mul_one_pm_imu_sub_mul_body.c:1-48 */
int tmhip_mul_one_pm_imu_sub_mul(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k, tmhip_field *j, double _sign, int N) {
  const double sign = _sign < 0. ? -1. : 1.;
  return launch_diag(ctx, l, k, j, 1., sign * ctx->mu, 1, 0, N);
}
/* tm_operators.c:813-858 */
int tmhip_mul_one_pm_imu_sub_mul_gamma5(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k, tmhip_field *j, double _sign) {
  const double sign = _sign < 0. ? -1. : 1.;
  return launch_diag(ctx, l, k, j, 1., sign * ctx->mu, 1, 1, ctx->Vh);
}
/* tm_operators.c:781-810 : l = g5 (k - j) */
int tmhip_mul_one_sub_mul_gamma5(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k, tmhip_field *j) {
  return launch_diag(ctx, l, k, j, 1., 0., 1, 1, ctx->Vh);
}
/* gamma.c:77-98 */
int tmhip_gamma5(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k, int N) {
  return launch_diag(ctx, l, k, nullptr, 1., 0., 0, 1, N);
}

}  // extern "C"
