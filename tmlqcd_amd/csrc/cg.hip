// cg_her (solver/cg_her.c:62-141) kept entirely in HBM.
//
// The reference's loop needs three scalars per iteration on the host (pro, err, and the stopping
// test).  Here alpha, beta, normsq and the stopping test live in a small device-side state block that
// the reduction epilogues update, and every field update reads its coefficient from there, so the
// host only enqueues kernels.  Exactness is preserved by a device-side `done` flag: once the test of
// cg_her.c:108 fires, all later updates of P / sf0 / sf2 are skipped, so P and the iteration count
// are those of the sequential algorithm even though the host polls `done` only every few iterations.
#include "tmhip_internal.h"

static inline v4f *F4(tmhip_field *f) { return reinterpret_cast<v4f *>(f->d32); }   // fp32 field as six float4 planes

struct CgState {
  double normsq, pro, err, alpha, beta, squarenorm, eps_sq;
  int rel_prec, done, iters, it;
  // inner loop of mixed_cg_her (mixed_cg_her.c:141): stop when err <= innereps * sqnrm_outer, after max_inner
  // iterations, or when 1.3 err already meets the outer target
  int inner; int max_inner; double innereps, sqnrm_outer;
  // inner loops of rg_mixed_cg_her (rg_mixed_cg_her.c:74-176), inner = 2 (float scalars) / 3 (double scalars):
  // run while rho > delta * rhomax and iter_base + j <= max_total; leave early when 1.3 rho < eps_sq
  double delta, rhomax; int iter_base, max_total;
  // fused Qtm_pm_psi iteration: alpha is known (and P += alpha p still owed) between the two scalar updates
  int x_pending;
};

// two copies of the state (the self-summing small-lattice iteration reads one and writes the other) and {pro, alpha} of the running iteration
static int cg_state_alloc(tmhip_ctx *ctx) {
  if (ctx->cg_state) return 0;
  TMHIP_CHECK(hipMalloc(&ctx->cg_state, 2 * sizeof(CgState) + 2 * sizeof(double)));
  TMHIP_CHECK(hipMemsetAsync(ctx->cg_state, 0, 2 * sizeof(CgState) + 2 * sizeof(double), ctx->stream));
  return 0;
}


__device__ __forceinline__ double cg_wave_reduce(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ void cg_block_reduce_store(double v, double *partials) {
  __shared__ double wsum[LA_BS / 64];
  v = cg_wave_reduce(v);
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

// The field kernels are written once for both precisions: V = v2d (fp64 field, twelve planes of complex doubles) or v4f (fp32
// field, six planes of component pairs); scalars have the element type, sums are accumulated in double.
__device__ __forceinline__ double la_dot(v2d a, v2d b) { return a.x * b.x + a.y * b.y; }
__device__ __forceinline__ double la_dot(v4f a, v4f b) { return ((double)a.x * b.x + (double)a.y * b.y) + ((double)a.z * b.z + (double)a.w * b.w); }

// pro = <sf2, sf0>   (cg_her.c:93)
template <class V>
__global__ __launch_bounds__(LA_BS) void cg_dot_kernel(const V *__restrict__ S, const V *__restrict__ R, int ns, int N,
                                                       double *partials, const CgState *st) {
  if (st->done) return;
  const V *s = S + (size_t)blockIdx.y * ns, *r = R + (size_t)blockIdx.y * ns;
  double acc = 0.0;
  const int base = blockIdx.x * LA_BS * LA_UNROLL + threadIdx.x;
#pragma unroll
  for (int u = 0; u < LA_UNROLL; u++) {
    const int i = base + u * LA_BS;
    if (i < N) acc += la_dot(s[i], r[i]);
  }
  cg_block_reduce_store(acc, partials);
}

// P += alpha sf2 ; sf0 = -alpha sf0 + sf1 ; partial |sf0|^2   (cg_her.c:95,101 fused: same bytes, one launch)
template <class V>
__global__ __launch_bounds__(LA_BS) void cg_update_kernel(V *__restrict__ P, const V *__restrict__ SF2, V *__restrict__ SF0,
                                                          const V *__restrict__ SF1, int ns, int N, double *partials,
                                                          const CgState *st) {
  if (st->done) return;
  typedef decltype(V{}.x) R;
  const R alpha = (R)st->alpha;
  const size_t off = (size_t)blockIdx.y * ns;
  V *p = P + off, *r0 = SF0 + off;
  const V *s2 = SF2 + off, *r1 = SF1 + off;
  double acc = 0.0;
  const int base = blockIdx.x * LA_BS * LA_UNROLL + threadIdx.x;
#pragma unroll
  for (int u = 0; u < LA_UNROLL; u++) {
    const int i = base + u * LA_BS;
    if (i < N) {
      p[i] = p[i] + alpha * s2[i];
      const V c = -alpha * r0[i] + r1[i];
      r0[i] = c;
      acc += la_dot(c, c);
    }
  }
  cg_block_reduce_store(acc, partials);
}

// sf2 = beta sf2 + sf0   (cg_her.c:122)
template <class V>
__global__ __launch_bounds__(LA_BS) void cg_xpay_kernel(V *__restrict__ SF2, const V *__restrict__ SF0, int ns, int N,
                                                        const CgState *st) {
  if (st->done) return;
  typedef decltype(V{}.x) R;
  const R beta = (R)st->beta;
  V *x = SF2 + (size_t)blockIdx.y * ns;
  const V *y = SF0 + (size_t)blockIdx.y * ns;
  const int base = blockIdx.x * LA_BS * LA_UNROLL + threadIdx.x;
#pragma unroll
  for (int u = 0; u < LA_UNROLL; u++) {
    const int i = base + u * LA_BS;
    if (i < N) x[i] = beta * x[i] + y[i];
  }
}

// Fused-iteration tail: P += alpha p (cg_her.c:95, owed since alpha became known) and p = beta p + r (cg_her.c:122) in
// one pass over p.  After convergence the direction update is skipped but the owed P update still happens -- exactly
// once: the next alpha-kernel clears x_pending when it finds `done` set.
template <class V>
__global__ __launch_bounds__(LA_BS) void cg_xp_kernel(V *__restrict__ X, V *__restrict__ Pd, const V *__restrict__ Rr, int ns, int N,
                                                      const CgState *st) {
  const bool upd_x = st->x_pending != 0, upd_p = !st->done;
  if (!upd_x && !upd_p) return;
  typedef decltype(V{}.x) R;
  const R alpha = (R)st->alpha, beta = (R)st->beta;
  const size_t off = (size_t)blockIdx.y * ns;
  V *x = X + off, *p = Pd + off;
  const V *r = Rr + off;
  const int base = blockIdx.x * LA_BS * LA_UNROLL + threadIdx.x;
#pragma unroll
  for (int u = 0; u < LA_UNROLL; u++) {
    const int i = base + u * LA_BS;
    if (i < N) {
      const V pv = p[i];
      if (upd_x) x[i] = x[i] + alpha * pv;
      if (upd_p) p[i] = beta * pv + r[i];
    }
  }
}

// The tail of a fused iteration on SMALL lattices without the sum + scalar kernel in front: every block adds up the partial sums of
// |r|^2 the residual stencil left (tmhip_block_sum256: the same value in every block), applies the stopping test of cg_her.c:108 and
// the beta of :121 itself, and does its share of  P += alpha p ; p = beta p + r  (cg_her.c:95,122).  The state is read from `cur` by
// everybody and written -- complete -- to `nxt` by block 0 only (a block that starts late must not see a half-updated state); the host
// alternates the two copies.  alpha was left in pa[1] by the residual stencil (HopSelfAlpha).  After convergence nothing changes any more.
__global__ __launch_bounds__(LA_BS) void cg_xp_self_kernel(v2d *__restrict__ X, v2d *__restrict__ Pd, const v2d *__restrict__ Rr, int ns, int N,
                                                           const double *__restrict__ partials, int n, const double *__restrict__ pa,
                                                           const CgState *__restrict__ cur, CgState *__restrict__ nxt, double *hist, int hist_len) {
  __shared__ double wsum[4];
  // the fields first (they do not depend on the sum), the partial sums behind them: one memory round trip instead of two
  const size_t off = (size_t)blockIdx.y * ns;
  v2d *x = X + off, *p = Pd + off;
  const v2d *r = Rr + off;
  const int base = blockIdx.x * LA_BS * LA_UNROLL + threadIdx.x;
  v2d pv[LA_UNROLL], xv[LA_UNROLL], rv[LA_UNROLL];
#pragma unroll
  for (int u = 0; u < LA_UNROLL; u++) {
    const int i = base + u * LA_BS;
    if (i < N) { pv[u] = p[i]; xv[u] = x[i]; rv[u] = r[i]; }
  }
  const double err = tmhip_block_sum256(partials, n, wsum);
  const int done = cur->done;
  const double normsq = cur->normsq, alpha = pa[1];
  const bool conv = ((err <= cur->eps_sq) && (cur->rel_prec == 0)) || ((err <= cur->eps_sq * cur->squarenorm) && (cur->rel_prec == 1));
  const double beta = err / normsq;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    CgState s = *cur;
    if (!done) {
      s.pro = pa[0]; s.alpha = alpha; s.err = err;
      s.it += 1;
      if (hist && s.it - 1 < hist_len) hist[s.it - 1] = err;
      if (conv) { s.done = 1; s.iters = s.it; }
      else { s.beta = beta; s.normsq = err; }
    }
    *nxt = s;
  }
  if (done) return;
#pragma unroll
  for (int u = 0; u < LA_UNROLL; u++) {
    const int i = base + u * LA_BS;
    if (i < N) {
      x[i] = xv[u] + alpha * pv[u];
      if (!conv) p[i] = beta * pv[u] + rv[u];
    }
  }
}

// fixed-order sum of the per-block partials into *out (same scheme as linalg.hip)
#define CG_SUM_BS 1024   // one block; the fused stencils deliver one partial per wave (8192 at 32^4), so use the widest block
// (no look at st->done: after convergence the sum is computed and ignored, which is cheaper than a dependent load in front of it)
__global__ __launch_bounds__(CG_SUM_BS) void cg_sum_kernel(const double *__restrict__ partials, int n, double *out, const CgState *st) {
  __shared__ double sm[CG_SUM_BS / 64];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += CG_SUM_BS) acc += partials[i];
  acc = cg_wave_reduce(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < CG_SUM_BS / 64; w++) tot += sm[w];
    *out = tot;
  }
}

// WHICH 0: after the dot  -> alpha = normsq / pro            (cg_her.c:93-94)
// WHICH 1: after |sf0|^2  -> stopping test, beta, normsq      (cg_her.c:101-126)
template <int WHICH>
__device__ __forceinline__ void cg_scalar_update(CgState *st, const double *sum, double *hist, int hist_len) {
  if (WHICH == 0) {
    st->pro = *sum;
    if (st->inner == 2 || st->inner == 1) st->alpha = (double)((float)st->normsq / (float)st->pro);   // float pro, alpha_cg: rg_mixed_cg_her.c:126, mixed_cg_her.c:67-69,127-128
    else st->alpha = st->normsq / st->pro;
    st->x_pending = 1;
  } else if (st->inner >= 2) {
    st->it += 1;
    bool stop;
    if (st->inner == 2) {          // rg_mixed_cg_her.c:122-145, scalars in float like the reference
      const float rho = (float)*sum, eps = (float)st->eps_sq, delta = (float)st->delta;
      float rhomax = (float)st->rhomax;
      st->err = rho;
      st->beta = (double)(rho / (float)st->normsq);
      st->normsq = rho;
      stop = 1.3 * rho < eps;
      if (!stop) {
        if (rho > rhomax) rhomax = rho;
        st->rhomax = rhomax;
        stop = !(rho > delta * rhomax && st->it + st->iter_base <= st->max_total);
      }
    } else {                       // inner_loop_high, rg_mixed_cg_her.c:74-108
      const double rho = *sum;
      st->err = rho;
      st->beta = rho / st->normsq;
      st->normsq = rho;
      stop = 1.3 * rho < st->eps_sq;
      if (!stop) {
        if (rho > st->rhomax) st->rhomax = rho;
        stop = !(rho > st->delta * st->rhomax && st->it + st->iter_base <= st->max_total);
      }
    }
    if (stop) { st->done = 1; st->iters = st->it; }
  } else {
    const double err = st->inner ? (double)(float)*sum : *sum;   // mixed_cg_her.c:67-69: err, beta_cg, sqnrm2 are floats in the inner loop
    st->err = err;
    st->it += 1;
    if (hist && st->it - 1 < hist_len) hist[st->it - 1] = err;
    bool conv;
    if (st->inner)
      conv = (err <= st->innereps * st->sqnrm_outer) || (st->it - 1 == st->max_inner) ||
             ((1.3 * err <= st->eps_sq) && (st->rel_prec == 0)) || ((1.3 * err <= st->eps_sq * st->squarenorm) && (st->rel_prec == 1));
    else
      conv = ((err <= st->eps_sq) && (st->rel_prec == 0)) || ((err <= st->eps_sq * st->squarenorm) && (st->rel_prec == 1));
    if (conv) { st->done = 1; st->iters = st->it; }
    else { st->beta = st->inner ? (double)((float)err / (float)st->normsq) : err / st->normsq; st->normsq = err; }
  }
}

// one wave: the state travels through LDS (one round trip to memory instead of a chain of dependent loads)
template <int WHICH>
__global__ __launch_bounds__(64) void cg_scalar_kernel(CgState *st, const double *sum, double *hist, int hist_len) {
  constexpr int NW = sizeof(CgState) / sizeof(double);
  static_assert(sizeof(CgState) % sizeof(double) == 0 && NW <= 64, "CgState is copied as doubles by one wave");
  __shared__ CgState ls;
  __shared__ double s;
  const int l = threadIdx.x;
  if (l < NW) reinterpret_cast<double *>(&ls)[l] = reinterpret_cast<const double *>(st)[l];
  if (l == 63) s = *sum;
  __syncthreads();
  if (ls.done) { if (WHICH == 0 && l == 0) st->x_pending = 0; return; }
  if (l == 0) cg_scalar_update<WHICH>(&ls, &s, hist, hist_len);
  __syncthreads();
  if (l < NW) reinterpret_cast<double *>(st)[l] = reinterpret_cast<const double *>(&ls)[l];
}

// single rank: the fixed-order sum of the partials and the scalar update in one launch (no all-reduce in between)
template <int WHICH>
__global__ __launch_bounds__(CG_SUM_BS) void cg_sum_scalar_kernel(const double *__restrict__ partials, int n, double *out, CgState *st, double *hist,
                                                                  int hist_len) {
  // The state is fetched into LDS by the last wave while the others are already loading partials: the scalar update then works
  // on LDS instead of walking a chain of dependent global loads (done, inner, normsq, ...) after the reduction.
  static_assert(sizeof(CgState) % sizeof(double) == 0 && sizeof(CgState) / sizeof(double) <= 64, "CgState is copied as doubles by one wave");
  constexpr int NW = sizeof(CgState) / sizeof(double);
  __shared__ CgState ls;
  __shared__ double sm[CG_SUM_BS / 64];
  const int lw = (int)threadIdx.x - (CG_SUM_BS - 64);
  if (lw >= 0 && lw < NW) reinterpret_cast<double *>(&ls)[lw] = reinterpret_cast<const double *>(st)[lw];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += CG_SUM_BS) acc += partials[i];
  acc = cg_wave_reduce(acc);                       // fixed order: strided per-lane sums, butterfly per wave, then the 16 wave sums
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (ls.done) { if (WHICH == 0 && threadIdx.x == 0) st->x_pending = 0; return; }   // block-uniform
  if (threadIdx.x == 0) {
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < CG_SUM_BS / 64; w++) tot += sm[w];
    *out = tot;
    cg_scalar_update<WHICH>(&ls, &tot, hist, hist_len);
  }
  __syncthreads();
  if (lw >= 0 && lw < NW) reinterpret_cast<double *>(st)[lw] = reinterpret_cast<const double *>(&ls)[lw];
}

// T-split ranks over the direct carrier: the same, with the sum over the ranks (tmhip_direct_sum_wave: one wave storing into every
// rank's block) between the local sum and the scalar update -- ONE launch per reduction instead of sum kernel + all-reduce + scalar kernel
template <int WHICH>
__global__ __launch_bounds__(CG_SUM_BS) void cg_sum_xsum_scalar_kernel(const double *__restrict__ partials, int n, double *out, CgState *st, double *hist,
                                                                       int hist_len, const TmhipSumArgs xs) {
  constexpr int NW = sizeof(CgState) / sizeof(double);
  __shared__ CgState ls;
  __shared__ double sm[CG_SUM_BS / 64];
  const int lw = (int)threadIdx.x - (CG_SUM_BS - 64);
  if (lw >= 0 && lw < NW) reinterpret_cast<double *>(&ls)[lw] = reinterpret_cast<const double *>(st)[lw];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += CG_SUM_BS) acc += partials[i];
  acc = cg_wave_reduce(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  // `done` is a function of global sums: the same on every rank, so a reduction that is skipped is skipped by everybody (its number with it)
  if (ls.done) { if (WHICH == 0 && threadIdx.x == 0) st->x_pending = 0; return; }   // block-uniform
  if (threadIdx.x < 64) {
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < CG_SUM_BS / 64; w++) tot += sm[w];
    const double gtot = tmhip_direct_sum_wave(tot, xs);
    if (threadIdx.x == 0) {
      double g = gtot;
      *out = g;
      cg_scalar_update<WHICH>(&ls, &g, hist, hist_len);
    }
  }
  __syncthreads();
  if (lw >= 0 && lw < NW) reinterpret_cast<double *>(st)[lw] = reinterpret_cast<const double *>(&ls)[lw];
}

int tmhip_apply_op(tmhip_ctx *ctx, int op, tmhip_field *l, tmhip_field *k) {
  switch (op) {
    case TMHIP_OP_QTM_PM: return tmhip_Qtm_pm_psi(ctx, l, k);
    case TMHIP_OP_QTM_PLUS: return tmhip_Qtm_plus_psi(ctx, l, k);
    case TMHIP_OP_QTM_MINUS: return tmhip_Qtm_minus_psi(ctx, l, k);
    case TMHIP_OP_MTM_PLUS: return tmhip_Mtm_plus_psi(ctx, l, k);
    case TMHIP_OP_MTM_MINUS: return tmhip_Mtm_minus_psi(ctx, l, k);
    case TMHIP_OP_QSW_PM: return tmhip_Qsw_pm_psi(ctx, l, k);
  }
  fprintf(stderr, "[tmlqcd_hip] cg_her: unknown operator id %d\n", op);
  return 1;
}

// straightforward port of the reference loop: three host round trips per iteration (kept for A/B: option cg_sync=1)
static int cg_her_sync(tmhip_ctx *ctx, tmhip_field *P, tmhip_field *Q, int max_iter, double eps_sq, int rel_prec, int N, int op,
                       int *iters, double *res_hist, int hist_len) {
  tmhip_field *sf0 = ctx->sf[0], *sf1 = ctx->sf[1], *sf2 = ctx->sf[2], *stmp;
  double normsq, pro, err = 0, alpha_cg, beta_cg, squarenorm;
  int iteration;
  if (tmhip_square_norm(ctx, Q, N, 1, &squarenorm)) return 1;
  if (tmhip_apply_op(ctx, op, sf0, P)) return 1;
  if (tmhip_diff(ctx, sf1, Q, sf0, N)) return 1;
  if (tmhip_assign(ctx, sf2, sf1, N)) return 1;
  if (tmhip_square_norm(ctx, sf1, N, 1, &normsq)) return 1;
  for (iteration = 1; iteration <= max_iter; iteration++) {
    if (tmhip_apply_op(ctx, op, sf0, sf2)) return 1;
    if (tmhip_scalar_prod_r(ctx, sf2, sf0, N, 1, &pro)) return 1;
    alpha_cg = normsq / pro;
    if (tmhip_assign_add_mul_r(ctx, P, sf2, alpha_cg, N)) return 1;
    if (tmhip_assign_mul_add_r_and_square(ctx, sf0, -alpha_cg, sf1, N, 1, &err)) return 1;
    if (res_hist && iteration - 1 < hist_len) res_hist[iteration - 1] = err;
    if (((err <= eps_sq) && (rel_prec == 0)) || ((err <= eps_sq * squarenorm) && (rel_prec == 1))) break;
    beta_cg = err / normsq;
    if (tmhip_assign_mul_add_r(ctx, sf2, beta_cg, sf0, N)) return 1;
    stmp = sf0; sf0 = sf1; sf1 = stmp;
    normsq = err;
  }
  *iters = iteration > max_iter ? -1 : iteration;
  return 0;
}

static int cg_allreduce(tmhip_ctx *ctx, double *x) {
  if (tmhip_reduce_over_ranks(ctx)) {
    if (ctx->direct.on && ctx->direct.sums_on) return tmhip_direct_allreduce(ctx, x);   // one wave storing into every rank's block: no communicator involved
    if (ctx->shm) return tmhip_shm_allreduce(ctx, ctx->stream, x, 1);
    TMHIP_NCCL_CHECK(ncclAllReduce(x, x, 1, ncclDouble, ncclSum, ctx->comm_red, ctx->stream));
  }
  return 0;
}

// partials -> (all-reduce over ranks) -> alpha (WHICH 0) or stopping test / beta (WHICH 1) in the device state
template <int WHICH>
static int cg_reduce_update(tmhip_ctx *ctx, int n, CgState *st, double *hist, int hist_len) {
  double *sum = ctx->result_dev + 1;
  if (tmhip_reduce_over_ranks(ctx) && ctx->direct.on && ctx->direct.sums_on) {
    TmhipSumArgs xs;
    if (tmhip_direct_sum_args(ctx, &xs)) return 1;
    hipLaunchKernelGGL(cg_sum_xsum_scalar_kernel<WHICH>, dim3(1), dim3(CG_SUM_BS), 0, ctx->stream, ctx->partials, n, sum, st, hist, hist_len, xs);
  } else if (tmhip_reduce_over_ranks(ctx)) {
    hipLaunchKernelGGL(cg_sum_kernel, dim3(1), dim3(CG_SUM_BS), 0, ctx->stream, ctx->partials, n, sum, st);
    if (cg_allreduce(ctx, sum)) return 1;
    hipLaunchKernelGGL(cg_scalar_kernel<WHICH>, dim3(1), dim3(64), 0, ctx->stream, st, sum, hist, hist_len);
  } else {
    hipLaunchKernelGGL(cg_sum_scalar_kernel<WHICH>, dim3(1), dim3(CG_SUM_BS), 0, ctx->stream, ctx->partials, n, sum, st, hist, hist_len);
  }
  return 0;
}

// One CG iteration on Qtm_pm_psi = Q_+ Q_- (tm_operators.c:338-345) with every vector operation except the final
// (P, p) update riding in stencil epilogues (single rank, whole blocks):
//   s1 = A_-^-1 H_eo p                         stencil 1
//   s0 = Q_- p ,  pro = |s0|^2                 stencil 2 + norm      (<p, Q_+ Q_- p> = |Q_- p|^2 as Q_+ = Q_-^dagger)
//   alpha = normsq / pro                       one small kernel
//   s1 = A_+^-1 H_eo s0                        stencil 3
//   r -= alpha Q_+ s0 ,  err = |r|^2           stencil 4 + residual update; A p itself is never written
//   stopping test, beta, normsq                one small kernel
//   P += alpha p ; p = beta p + r              one pass
// i.e. cg_her.c:91-126 with 960 B/site of vector traffic next to the four stencils instead of 1728.
static int cg_enqueue_fused_qtm(tmhip_ctx *ctx, bool fp32, tmhip_field *x, tmhip_field *p, tmhip_field *r, CgState *st, double *hist,
                                int hist_len, int N, bool clover = false, int self_parity = -1) {
  const double mu = ctx->mu, nrm = 1. / (1. + mu * mu);
  const dim3 g = fp32 ? la_grid32(N) : la_grid(N);
  const size_t gs = ctx->gs;
  int n1 = 0, n2 = 0;
  if (clover && fabs(mu) > 0 && ctx->sw_inv_sets < 2) TMHIP_FAIL("Qsw_pm_psi with mu != 0 needs both sets of sw_inv (sw_invert with the current mu)");
  if (fp32) {
    v2f *s0 = ctx->scratch32[0]->d32, *s1 = ctx->scratch32[1]->d32;
    if (clover) {   // Qsw_pm_psi_32 (clovertm_operators_32.c): clover_inv / clover_gamma5 epilogues instead of the twists
      const v2f *wim = ctx->sw_inv32 + (size_t)(fabs(mu) > 0 ? 1 : 0) * 72 * gs, *wip = ctx->sw_inv32, *wo = ctx->sw32 + (size_t)54 * gs;
      if (tmhip_launch_hopping32(ctx, TMHIP_EO, s1, p->d32, nullptr, EPI_CLOVER_INV, 0, 0, HOP_COMM | HOP_FEED, wim)) return 1;
      if (tmhip_launch_hopping_dot32(ctx, TMHIP_OE, s0, s1, p->d32, nullptr, 0, -(mu + ctx->mu3), &n1, 1, nullptr, nullptr, wo, 3)) return 1;
      if (cg_reduce_update<0>(ctx, n1, st, hist, hist_len)) return 1;
      if (tmhip_launch_hopping32(ctx, TMHIP_EO, s1, s0, nullptr, EPI_CLOVER_INV, 0, 0, HOP_COMM | HOP_CHAINED | HOP_FEED, wip)) return 1;
      if (tmhip_launch_hopping_dot32(ctx, TMHIP_OE, nullptr, s1, s0, nullptr, 0, +(mu + ctx->mu3), &n2, 2, r->d32, &st->alpha, wo, 1)) return 1;
    } else {
      if (tmhip_launch_hopping32(ctx, TMHIP_EO, s1, p->d32, nullptr, EPI_TM_TIMES, nrm, nrm * mu, HOP_COMM | HOP_FEED)) return 1;
      if (tmhip_launch_hopping_dot32(ctx, TMHIP_OE, s0, s1, p->d32, nullptr, 1., -mu, &n1, 1, nullptr, nullptr, nullptr, 3)) return 1;
      if (cg_reduce_update<0>(ctx, n1, st, hist, hist_len)) return 1;
      if (tmhip_launch_hopping32(ctx, TMHIP_EO, s1, s0, nullptr, EPI_TM_TIMES, nrm, -nrm * mu, HOP_COMM | HOP_CHAINED | HOP_FEED)) return 1;
      if (tmhip_launch_hopping_dot32(ctx, TMHIP_OE, nullptr, s1, s0, nullptr, 1., mu, &n2, 2, r->d32, &st->alpha, nullptr, 1)) return 1;
    }
    if (cg_reduce_update<1>(ctx, n2, st, hist, hist_len)) return 1;
    hipLaunchKernelGGL(cg_xp_kernel<v4f>, g, dim3(LA_BS), 0, ctx->stream, F4(x), F4(p), (const v4f *)F4(r), p->ns, N, st);
  } else {
    v2d *s0 = ctx->scratch[0]->d, *s1 = ctx->scratch[1]->d;
    if (clover) {   // Qsw_pm_psi (clovertm_operators.c:233-245)
      const v2d *wim = ctx->sw_inv + (size_t)(fabs(mu) > 0 ? 1 : 0) * 72 * gs, *wip = ctx->sw_inv, *wo = ctx->sw + (size_t)54 * gs;
      if (tmhip_launch_hopping(ctx, TMHIP_EO, s1, p->d, nullptr, EPI_CLOVER_INV, 0, 0, HOP_COMM | HOP_FEED, wim)) return 1;
      if (tmhip_launch_hopping_dot(ctx, TMHIP_OE, s0, s1, p->d, nullptr, 0, -(mu + ctx->mu3), &n1, 1, nullptr, nullptr, wo, 3)) return 1;
      if (cg_reduce_update<0>(ctx, n1, st, hist, hist_len)) return 1;
      if (tmhip_launch_hopping(ctx, TMHIP_EO, s1, s0, nullptr, EPI_CLOVER_INV, 0, 0, HOP_COMM | HOP_CHAINED | HOP_FEED, wip)) return 1;
      if (tmhip_launch_hopping_dot(ctx, TMHIP_OE, nullptr, s1, s0, nullptr, 0, +(mu + ctx->mu3), &n2, 2, r->d, &st->alpha, wo, 1)) return 1;
    } else if (self_parity >= 0) {
      // small unsplit lattices (tmhip_hopping_self_alpha_ok): no sum + scalar kernels -- the residual stencil adds up the partials of
      // |Q_- p|^2 itself (alpha), the (P, p) kernel those of |r|^2 (stopping test, beta); `st` is the pair of state copies
      double *pa = reinterpret_cast<double *>(st + 2);
      const CgState *cur = st + self_parity;
      CgState *nxt = st + (1 - self_parity);
      if (tmhip_launch_hopping(ctx, TMHIP_EO, s1, p->d, nullptr, EPI_TM_TIMES, nrm, nrm * mu, HOP_COMM | HOP_FEED)) return 1;
      if (tmhip_launch_hopping_dot(ctx, TMHIP_OE, s0, s1, p->d, nullptr, 1., -mu, &n1, 1, nullptr, nullptr, nullptr, 3)) return 1;
      if (tmhip_launch_hopping(ctx, TMHIP_EO, s1, s0, nullptr, EPI_TM_TIMES, nrm, -nrm * mu, HOP_COMM | HOP_CHAINED | HOP_FEED)) return 1;
      const HopSelfAlpha self = {ctx->partials, n1, &cur->normsq, pa};
      if (tmhip_launch_hopping_dot(ctx, TMHIP_OE, nullptr, s1, s0, nullptr, 1., mu, &n2, 2, r->d, nullptr, nullptr, 1, &self)) return 1;
      hipLaunchKernelGGL(cg_xp_self_kernel, g, dim3(LA_BS), 0, ctx->stream, x->d, p->d, (const v2d *)r->d, p->ns, N,
                         (const double *)(ctx->partials + ctx->max_partials / 2), n2, (const double *)pa, cur, nxt, hist, hist_len);
      return 0;
    } else {
      if (tmhip_launch_hopping(ctx, TMHIP_EO, s1, p->d, nullptr, EPI_TM_TIMES, nrm, nrm * mu, HOP_COMM | HOP_FEED)) return 1;
      if (tmhip_launch_hopping_dot(ctx, TMHIP_OE, s0, s1, p->d, nullptr, 1., -mu, &n1, 1, nullptr, nullptr, nullptr, 3)) return 1;
      if (cg_reduce_update<0>(ctx, n1, st, hist, hist_len)) return 1;
      if (tmhip_launch_hopping(ctx, TMHIP_EO, s1, s0, nullptr, EPI_TM_TIMES, nrm, -nrm * mu, HOP_COMM | HOP_CHAINED | HOP_FEED)) return 1;
      if (tmhip_launch_hopping_dot(ctx, TMHIP_OE, nullptr, s1, s0, nullptr, 1., mu, &n2, 2, r->d, &st->alpha, nullptr, 1)) return 1;
    }
    if (cg_reduce_update<1>(ctx, n2, st, hist, hist_len)) return 1;
    hipLaunchKernelGGL(cg_xp_kernel<v2d>, g, dim3(LA_BS), 0, ctx->stream, x->d, p->d, (const v2d *)r->d, p->ns, N, st);
  }
  return 0;
}

extern "C" int tmhip_cg_her(tmhip_ctx *ctx, tmhip_field *P, tmhip_field *Q, int max_iter, double eps_sq, int rel_prec, int N,
                            int op, int *iters, double *res_hist, int hist_len) {
  if (!P || !Q || P->kind != TMHIP_FIELD_EO || Q->kind != TMHIP_FIELD_EO) TMHIP_FAIL("cg_her needs one-parity (EO) fields");
  if (N != ctx->Vh) TMHIP_FAIL("cg_her: N must be VOLUME/2");
  if (ctx->opt_cg_sync) return cg_her_sync(ctx, P, Q, max_iter, eps_sq, rel_prec, N, op, iters, res_hist, hist_len);
  TMHIP_CHECK(hipSetDevice(ctx->device));
  if (cg_state_alloc(ctx)) return 1;
  if (max_iter > ctx->cg_hist_len) {
    if (ctx->cg_hist) TMHIP_CHECK(hipFree(ctx->cg_hist));
    TMHIP_CHECK(hipMalloc((void **)&ctx->cg_hist, sizeof(double) * (size_t)max_iter));
    ctx->cg_hist_len = max_iter;
  }
  CgState *st = (CgState *)ctx->cg_state;
  tmhip_field *sf0 = ctx->sf[0], *sf1 = ctx->sf[1], *sf2 = ctx->sf[2], *stmp;
  // initial residual (cg_her.c:82-88): once per solve, host-visible scalars are fine here
  CgState h;
  memset(&h, 0, sizeof(h));
  if (tmhip_square_norm(ctx, Q, N, 1, &h.squarenorm)) return 1;
  if (tmhip_apply_op(ctx, op, sf0, P)) return 1;
  if (tmhip_diff(ctx, sf1, Q, sf0, N)) return 1;
  if (tmhip_assign(ctx, sf2, sf1, N)) return 1;
  if (tmhip_square_norm(ctx, sf1, N, 1, &h.normsq)) return 1;
  h.eps_sq = eps_sq; h.rel_prec = rel_prec;
  TMHIP_CHECK(hipMemcpyAsync(st, &h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
  const dim3 g = la_grid(N);
  const int nblk = g.x * g.y;
  const int batch = ctx->opt_cg_batch > 0 ? ctx->opt_cg_batch : 4;
  const bool split = ctx->g.nproc_t > 1 || ctx->loopback;   // T-split rank (or its single-rank rehearsal): reductions fused into the interior + boundary kernels
  const bool fusable = ctx->opt_cg_fused_dot && ctx->Vh % (split ? 256 : tmhip_hop_block(ctx)) == 0 &&
                       (!split || (ctx->face % 256 == 0 && ctx->g.T >= 3));
  const bool fused = fusable && !split && op == TMHIP_OP_QTM_PM;                            // scalar product in the last stencil (cg_fused_dot = 1)
  const bool fused_full = fusable && ctx->opt_cg_fused_dot >= 2 && (op == TMHIP_OP_QTM_PM || op == TMHIP_OP_QSW_PM);
  // small unsplit lattices: the iteration that adds up its partial sums inside the stencil and the (P, p) kernel (two state copies,
  // read / written alternately: iteration j reads copy j & 1); "cg_self" 0 keeps the sum + scalar kernels
  const bool self = fused_full && !split && op == TMHIP_OP_QTM_PM && ctx->opt_cg_self && tmhip_hopping_self_alpha_ok(ctx);
  if (self) TMHIP_CHECK(hipMemcpyAsync(st + 1, &h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
  int enq = 0, done = 0;
  int *flag = (int *)(ctx->result_host + 2);
  double *err_host = ctx->result_host + 3;
  const double target = rel_prec == 1 ? eps_sq * h.squarenorm : eps_sq;
  bool near = false;   // within 10^3 of the target: poll every iteration so that no stencil is enqueued past convergence
  while (enq < max_iter && !done) {
    const int want = near ? 1 : batch;
    const int nb = (max_iter - enq) < want ? (max_iter - enq) : want;
    for (int b = 0; b < nb; b++) {
      int ndot = nblk;
      if (fused_full) {   // default: everything but the (P, p) update rides in stencil epilogues
        if (cg_enqueue_fused_qtm(ctx, false, P, sf2, sf1, st, ctx->cg_hist, max_iter, N, op == TMHIP_OP_QSW_PM, self ? ((enq + b) & 1) : -1)) return 1;
        continue;
      }
      if (fused) {
        // Qtm_pm_psi (tm_operators.c:338-345) with pro = <sf2, Q sf2> accumulated by the last stencil's epilogue
        const double mu = ctx->mu, nrm = 1. / (1. + mu * mu);
        tmhip_field *s0 = ctx->scratch[0], *s1 = ctx->scratch[1];
        if (tmhip_launch_hopping(ctx, TMHIP_EO, s1->d, sf2->d, nullptr, EPI_TM_TIMES, nrm, nrm * mu, true)) return 1;
        if (tmhip_launch_hopping(ctx, TMHIP_OE, s0->d, s1->d, sf2->d, EPI_TM_SUB_G5, 1., -mu, true)) return 1;
        if (tmhip_launch_hopping(ctx, TMHIP_EO, s1->d, s0->d, nullptr, EPI_TM_TIMES, nrm, -nrm * mu, true)) return 1;
        if (tmhip_launch_hopping_dot(ctx, TMHIP_OE, sf0->d, s1->d, s0->d, sf2->d, 1., mu, &ndot)) return 1;
      } else {
        if (tmhip_apply_op(ctx, op, sf0, sf2)) return 1;
        hipLaunchKernelGGL(cg_dot_kernel<v2d>, g, dim3(LA_BS), 0, ctx->stream, sf2->d, sf0->d, sf2->ns, N, ctx->partials, st);
      }
      if (cg_reduce_update<0>(ctx, ndot, st, ctx->cg_hist, max_iter)) return 1;
      hipLaunchKernelGGL(cg_update_kernel<v2d>, g, dim3(LA_BS), 0, ctx->stream, P->d, sf2->d, sf0->d, sf1->d, P->ns, N, ctx->partials, st);
      if (cg_reduce_update<1>(ctx, nblk, st, ctx->cg_hist, max_iter)) return 1;
      hipLaunchKernelGGL(cg_xpay_kernel<v2d>, g, dim3(LA_BS), 0, ctx->stream, sf2->d, sf0->d, sf2->ns, N, st);
      stmp = sf0; sf0 = sf1; sf1 = stmp;
    }
    enq += nb;
    if (self) st = (CgState *)ctx->cg_state + (enq & 1);   // the copy the last enqueued iteration wrote
    TMHIP_CHECK(hipGetLastError());
    TMHIP_CHECK(hipMemcpyAsync(flag, &st->done, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    TMHIP_CHECK(hipMemcpyAsync(err_host, &st->err, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
    done = *flag;
    if (tmhip_check_async_error(ctx)) return 1;   // T-split rank: a halo exchange that never completed must not yield a result
    near = *err_host <= 1.0e3 * target;
    if (self) st = (CgState *)ctx->cg_state;       // (cg_enqueue_fused_qtm takes the base of the pair)
  }
  if (self) st = (CgState *)ctx->cg_state + (enq & 1);
  TMHIP_CHECK(hipMemcpyAsync(&h, st, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  *iters = h.done ? h.iters : -1;
  if (res_hist && hist_len > 0) {
    const int n = h.it < hist_len ? h.it : hist_len;
    if (n > 0) TMHIP_CHECK(hipMemcpy(res_hist, ctx->cg_hist, sizeof(double) * n, hipMemcpyDeviceToHost));
  }
  return 0;
}

// ------------------------------------------------------------------ mixed precision
/* solver/mixed_cg_her.c:65-202: fp32 inner CG on the defect equation until err <= innereps * |delta|^2, then the
 * solution is accumulated and the defect recomputed in fp64 (x += x32; delta = Q - A x).  The inner loop is the
 * device-resident loop of tmhip_cg_her instantiated for float2 fields with the stopping rule of :141. */
extern "C" int tmhip_mixed_cg_her(tmhip_ctx *ctx, tmhip_field *P, tmhip_field *Q, int max_iter, double eps_sq, int rel_prec, int N,
                                  int op, double innereps, int max_inner_it, int *iters, int *outer_iters) {
  if (!P || !Q || P->kind != TMHIP_FIELD_EO || Q->kind != TMHIP_FIELD_EO || P->prec || Q->prec) TMHIP_FAIL("mixed_cg_her needs fp64 one-parity fields");
  if (N != ctx->Vh) TMHIP_FAIL("mixed_cg_her: N must be VOLUME/2");
  if (op != TMHIP_OP_QTM_PM && op != TMHIP_OP_QSW_PM) TMHIP_FAIL("mixed_cg_her: fp32 operators exist for Qtm_pm_psi and Qsw_pm_psi only");
  TMHIP_CHECK(hipSetDevice(ctx->device));
  if (tmhip_prepare_fp32(ctx)) return 1;
  const bool clover = op == TMHIP_OP_QSW_PM;
  if (clover && tmhip_prepare_clover32(ctx)) return 1;
  if (cg_state_alloc(ctx)) return 1;
  CgState *st = (CgState *)ctx->cg_state;
  int N_outer = max_iter / (max_inner_it > 0 ? max_inner_it : 1);
  if (N_outer < 10) N_outer = 10;                                   /* mixed_cg_her.c:83-85 */
  tmhip_field *delta = ctx->sf[0], *y = ctx->sf[1];
  tmhip_field *x = ctx->sf32[3];
  double sourcesquarenorm, sqnrm_d;
  if (tmhip_square_norm(ctx, Q, N, 1, &sourcesquarenorm)) return 1;
  sqnrm_d = sourcesquarenorm;
  if (tmhip_assign(ctx, delta, Q, N)) return 1;
  if (tmhip_field_zero(ctx, P)) return 1;
  const dim3 g = la_grid32(N);   // every field kernel of the inner loop works on fp32 fields
  const int nblk = g.x * g.y;
  int *flag = (int *)(ctx->result_host + 2);
  const int batch = ctx->opt_cg_batch > 0 ? ctx->opt_cg_batch : 4;
  const bool split = ctx->g.nproc_t > 1 || ctx->loopback;
  const bool fused = ctx->opt_cg_fused_dot && tmhip_fused_dot32_ok(ctx) && (!split || ctx->opt_cg_fused_dot >= 2);   // the older mode-0 fusion is unsplit only
  const double mu = ctx->mu, nrm = 1. / (1. + mu * mu);
  int iter = 0;
  ctx->mixed_trace_n = 0;
  for (int i = 0; i < N_outer; i++) {
    tmhip_field *sf0 = ctx->sf32[0], *sf1 = ctx->sf32[1], *sf2 = ctx->sf32[2], *stmp;
    if (tmhip_field_zero(ctx, x)) return 1;
    if (tmhip_assign_to_32(ctx, sf1, delta, N)) return 1;
    if (tmhip_assign_to_32(ctx, sf2, delta, N)) return 1;
    CgState h;
    memset(&h, 0, sizeof(h));
    h.normsq = (double)(float)sqnrm_d; h.sqnrm_outer = h.normsq; h.squarenorm = sourcesquarenorm;
    h.eps_sq = eps_sq; h.rel_prec = rel_prec; h.inner = 1; h.max_inner = max_inner_it; h.innereps = innereps;
    TMHIP_CHECK(hipMemcpyAsync(st, &h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
    int done = 0, enq = 0;
    // the inner loops are short (a handful of iterations each): poll every iteration once the residual is within 10^3 of
    // whichever stopping rule of mixed_cg_her.c:141 is closer, so that no stencil is enqueued past the restart
    double *err_host = ctx->result_host + 3;
    const double t_outer = (rel_prec == 1 ? eps_sq * sourcesquarenorm : eps_sq) / 1.3, t_inner = innereps * h.sqnrm_outer;
    const double tgt = t_inner > t_outer ? t_inner : t_outer;
    bool near = false;
    while (!done && enq <= max_inner_it) {
      const int want = near ? 1 : batch;
      for (int b = 0; b < want; b++) {
        int ndot = nblk;
        v2f *s0 = ctx->scratch32[0]->d32, *s1 = ctx->scratch32[1]->d32;
        if (fused && ctx->opt_cg_fused_dot >= 2) {
          if (cg_enqueue_fused_qtm(ctx, true, x, sf2, sf1, st, (double *)nullptr, 0, N, clover)) return 1;
          continue;
        }
        if (clover) {
          if (tmhip_Qsw_pm_psi_32(ctx, sf0, sf2)) return 1;
          hipLaunchKernelGGL(cg_dot_kernel<v4f>, g, dim3(LA_BS), 0, ctx->stream, (const v4f *)F4(sf2), (const v4f *)F4(sf0), sf2->ns, N, ctx->partials, st);
        } else {
        if (tmhip_launch_hopping32(ctx, TMHIP_EO, s1, sf2->d32, nullptr, EPI_TM_TIMES, nrm, nrm * mu, true)) return 1;
        if (tmhip_launch_hopping32(ctx, TMHIP_OE, s0, s1, sf2->d32, EPI_TM_SUB_G5, 1., -mu, true)) return 1;
        if (tmhip_launch_hopping32(ctx, TMHIP_EO, s1, s0, nullptr, EPI_TM_TIMES, nrm, -nrm * mu, true)) return 1;
        if (fused) {
          if (tmhip_launch_hopping_dot32(ctx, TMHIP_OE, sf0->d32, s1, s0, sf2->d32, 1., mu, &ndot)) return 1;
        } else {
          if (tmhip_launch_hopping32(ctx, TMHIP_OE, sf0->d32, s1, s0, EPI_TM_SUB_G5, 1., mu, true)) return 1;
          hipLaunchKernelGGL(cg_dot_kernel<v4f>, g, dim3(LA_BS), 0, ctx->stream, (const v4f *)F4(sf2), (const v4f *)F4(sf0), sf2->ns, N, ctx->partials, st);
        }
        }
        if (cg_reduce_update<0>(ctx, ndot, st, (double *)nullptr, 0)) return 1;
        hipLaunchKernelGGL(cg_update_kernel<v4f>, g, dim3(LA_BS), 0, ctx->stream, F4(x), (const v4f *)F4(sf2), F4(sf0), (const v4f *)F4(sf1),
                           x->ns, N, ctx->partials, st);
        if (cg_reduce_update<1>(ctx, nblk, st, (double *)nullptr, 0)) return 1;
        hipLaunchKernelGGL(cg_xpay_kernel<v4f>, g, dim3(LA_BS), 0, ctx->stream, F4(sf2), (const v4f *)F4(sf0), sf2->ns, N, st);
        stmp = sf0; sf0 = sf1; sf1 = stmp;
      }
      enq += want;
      TMHIP_CHECK(hipGetLastError());
      TMHIP_CHECK(hipMemcpyAsync(flag, &st->done, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
      TMHIP_CHECK(hipMemcpyAsync(err_host, &st->err, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
      done = *flag;
    if (tmhip_check_async_error(ctx)) return 1;   // T-split rank: a halo exchange that never completed must not yield a result
      near = *err_host <= 1.0e3 * tgt;
    }
    TMHIP_CHECK(hipMemcpyAsync(&h, st, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (!h.done) TMHIP_FAIL("mixed_cg_her: inner loop did not terminate");
    iter += h.iters - 1;                                             /* "iter += j" with j = completed iterations - 1 */
    if (ctx->mixed_trace_n < (int)(sizeof(ctx->mixed_trace) / sizeof(int))) ctx->mixed_trace[ctx->mixed_trace_n++] = h.iters - 1;
    /* defect in double precision (mixed_cg_her.c:157-162) */
    if (tmhip_add_from_32(ctx, P, x, N)) return 1;
    if (tmhip_apply_op(ctx, op, y, P)) return 1;
    if (tmhip_diff(ctx, delta, Q, y, N)) return 1;
    if (tmhip_square_norm(ctx, delta, N, 1, &sqnrm_d)) return 1;
    if (((sqnrm_d <= eps_sq) && (rel_prec == 0)) || ((sqnrm_d <= eps_sq * sourcesquarenorm) && (rel_prec == 1))) {
      *iters = iter + i;
      if (outer_iters) *outer_iters = i + 1;
      return 0;
    }
    iter++;
  }
  *iters = -1;
  if (outer_iters) *outer_iters = N_outer;
  return 0;
}

/* restart points of the last tmhip_mixed_cg_her: the reference's j (mixed_cg_her.c:152 "iter += j") of every outer iteration */
extern "C" int tmhip_mixed_cg_restarts(tmhip_ctx *ctx, int *inner_iters, int cap, int *n_outer) {
  const int n = ctx->mixed_trace_n < cap ? ctx->mixed_trace_n : cap;
  for (int i = 0; i < n; i++) inner_iters[i] = ctx->mixed_trace[i];
  *n_outer = ctx->mixed_trace_n;
  return 0;
}

// ------------------------------------------------------------------ reliable-update mixed CG
// One CG iteration enqueued on ctx->stream, fp32 or fp64 fields: q = A p ; alpha ; x += alpha p ; r -= alpha q ;
// rho ; beta ; p = beta p + r.   f.q and f.r swap roles each iteration exactly as sf[0]/sf[1] do in tmhip_cg_her.
struct RgFields { tmhip_field *x, *p, *q, *r; };

static int rg_enqueue_iteration(tmhip_ctx *ctx, int op, bool fp32, bool fused, RgFields &f, CgState *st, int N) {
  const dim3 g = fp32 ? la_grid32(N) : la_grid(N);
  const int nblk = g.x * g.y;
  const double mu = ctx->mu, nrm = 1. / (1. + mu * mu);
  int ndot = nblk;
  if (fp32 && fused && ctx->opt_cg_fused_dot >= 2) return cg_enqueue_fused_qtm(ctx, true, f.x, f.p, f.r, st, (double *)nullptr, 0, N, op == TMHIP_OP_QSW_PM);
  if (fp32) {
    if (op == TMHIP_OP_QSW_PM) {
      if (tmhip_Qsw_pm_psi_32(ctx, f.q, f.p)) return 1;
      hipLaunchKernelGGL(cg_dot_kernel<v4f>, g, dim3(LA_BS), 0, ctx->stream, (const v4f *)F4(f.p), (const v4f *)F4(f.q), f.p->ns, N, ctx->partials, st);
    } else {
      v2f *s0 = ctx->scratch32[0]->d32, *s1 = ctx->scratch32[1]->d32;
      if (tmhip_launch_hopping32(ctx, TMHIP_EO, s1, f.p->d32, nullptr, EPI_TM_TIMES, nrm, nrm * mu, true)) return 1;
      if (tmhip_launch_hopping32(ctx, TMHIP_OE, s0, s1, f.p->d32, EPI_TM_SUB_G5, 1., -mu, true)) return 1;
      if (tmhip_launch_hopping32(ctx, TMHIP_EO, s1, s0, nullptr, EPI_TM_TIMES, nrm, -nrm * mu, true)) return 1;
      if (fused) {
        if (tmhip_launch_hopping_dot32(ctx, TMHIP_OE, f.q->d32, s1, s0, f.p->d32, 1., mu, &ndot)) return 1;
      } else {
        if (tmhip_launch_hopping32(ctx, TMHIP_OE, f.q->d32, s1, s0, EPI_TM_SUB_G5, 1., mu, true)) return 1;
        hipLaunchKernelGGL(cg_dot_kernel<v4f>, g, dim3(LA_BS), 0, ctx->stream, (const v4f *)F4(f.p), (const v4f *)F4(f.q), f.p->ns, N, ctx->partials, st);
      }
    }
  } else {
    if (tmhip_apply_op(ctx, op, f.q, f.p)) return 1;
    hipLaunchKernelGGL(cg_dot_kernel<v2d>, g, dim3(LA_BS), 0, ctx->stream, f.p->d, f.q->d, f.p->ns, N, ctx->partials, st);
  }
  if (cg_reduce_update<0>(ctx, ndot, st, (double *)nullptr, 0)) return 1;
  if (fp32)
    hipLaunchKernelGGL(cg_update_kernel<v4f>, g, dim3(LA_BS), 0, ctx->stream, F4(f.x), (const v4f *)F4(f.p), F4(f.q), (const v4f *)F4(f.r),
                       f.x->ns, N, ctx->partials, st);
  else
    hipLaunchKernelGGL(cg_update_kernel<v2d>, g, dim3(LA_BS), 0, ctx->stream, f.x->d, (const v2d *)f.p->d, f.q->d, (const v2d *)f.r->d, f.x->ns, N,
                       ctx->partials, st);
  if (cg_reduce_update<1>(ctx, nblk, st, (double *)nullptr, 0)) return 1;
  if (fp32)
    hipLaunchKernelGGL(cg_xpay_kernel<v4f>, g, dim3(LA_BS), 0, ctx->stream, F4(f.p), (const v4f *)F4(f.q), f.p->ns, N, st);
  else
    hipLaunchKernelGGL(cg_xpay_kernel<v2d>, g, dim3(LA_BS), 0, ctx->stream, f.p->d, (const v2d *)f.q->d, f.p->ns, N, st);
  tmhip_field *t = f.q; f.q = f.r; f.r = t;   // the new residual now sits in the former q
  return 0;
}

// inner_loop / inner_loop_high of rg_mixed_cg_her.c:74-176 (non-pipelined, Fletcher-Reeves beta): returns j, updates *rho1
static int rg_inner_loop(tmhip_ctx *ctx, int op, bool fp32, bool fused, RgFields f, double *rho1, double delta, double eps_sq, int N,
                         int iter_base, int max_iter, int *j_out) {
  CgState *st = (CgState *)ctx->cg_state;
  CgState h;
  memset(&h, 0, sizeof(h));
  h.normsq = *rho1; h.rhomax = *rho1; h.delta = delta; h.eps_sq = eps_sq;
  h.inner = fp32 ? 2 : 3; h.iter_base = iter_base; h.max_total = max_iter;
  *j_out = 0;
  // the reference tests the loop condition before the first iteration as well (rho = rhomax => rho > delta rhomax iff delta < 1)
  const bool enter = fp32 ? ((float)*rho1 > (float)delta * (float)*rho1) : (*rho1 > delta * *rho1);
  if (!enter || iter_base > max_iter) return 0;
  TMHIP_CHECK(hipMemcpyAsync(st, &h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
  int *flag = (int *)(ctx->result_host + 2);
  const int batch = ctx->opt_cg_batch > 0 ? ctx->opt_cg_batch : 4;
  int done = 0, enq = 0;
  double *err_host = ctx->result_host + 3;
  const double t_rel = delta * *rho1, t_abs = eps_sq / 1.3;      // the two exits of rg_mixed_cg_her.c:122-145 (rhomax >= rho1)
  const double tgt = t_rel > t_abs ? t_rel : t_abs;
  bool near = false;                                             // as in tmhip_mixed_cg_her: poll every iteration close to the restart
  while (!done) {
    const int want = near ? 1 : batch;
    for (int b = 0; b < want; b++)
      if (rg_enqueue_iteration(ctx, op, fp32, fused, f, st, N)) return 1;
    enq += want;
    TMHIP_CHECK(hipGetLastError());
    TMHIP_CHECK(hipMemcpyAsync(flag, &st->done, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    TMHIP_CHECK(hipMemcpyAsync(err_host, &st->err, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
    done = *flag;
    if (tmhip_check_async_error(ctx)) return 1;   // T-split rank: a halo exchange that never completed must not yield a result
    near = *err_host <= 1.0e3 * tgt;
    if (!done && enq > max_iter + batch) TMHIP_FAIL("rg_mixed_cg_her: inner loop did not terminate");
  }
  TMHIP_CHECK(hipMemcpyAsync(&h, st, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  *j_out = h.iters;
  *rho1 = h.normsq;
  return 0;
}

/* solver/rg_mixed_cg_her.c:180-347.  fp32 CG with true reliable updates: the inner loop runs until the iterated
 * residual has dropped by `delta` relative to its maximum since the last update, then the solution is accumulated
 * and the residual recomputed in fp64; when the outer-iteration estimate N_outer is nearly used up the solver
 * falls back to fp64 inner loops.  Returns the reference's count iter_out + iter_in_sp + iter_in_dp, or -1. */
extern "C" int tmhip_rg_mixed_cg_her(tmhip_ctx *ctx, tmhip_field *P, tmhip_field *Q, int max_iter, double eps_sq, int rel_prec, int N,
                                     int op, double delta_in, int *iters, int *iter_out_p, int *iter_sp_p, int *iter_dp_p) {
  if (!P || !Q || P->kind != TMHIP_FIELD_EO || Q->kind != TMHIP_FIELD_EO || P->prec || Q->prec) TMHIP_FAIL("rg_mixed_cg_her needs fp64 one-parity fields");
  if (N != ctx->Vh) TMHIP_FAIL("rg_mixed_cg_her: N must be VOLUME/2");
  if (op != TMHIP_OP_QTM_PM && op != TMHIP_OP_QSW_PM) TMHIP_FAIL("rg_mixed_cg_her: fp32 operators exist for Qtm_pm_psi and Qsw_pm_psi only");
  TMHIP_CHECK(hipSetDevice(ctx->device));
  if (tmhip_prepare_fp32(ctx)) return 1;
  const bool clover = op == TMHIP_OP_QSW_PM;
  if (clover && tmhip_prepare_clover32(ctx)) return 1;
  if (cg_state_alloc(ctx)) return 1;
  if (!ctx->sf_extra && tmhip_field_alloc(ctx, TMHIP_FIELD_EO, &ctx->sf_extra)) return 1;
  const bool split = ctx->g.nproc_t > 1 || ctx->loopback;
  const bool fused = ctx->opt_cg_fused_dot && tmhip_fused_dot32_ok(ctx) && (!split || ctx->opt_cg_fused_dot >= 2);   // the older mode-0 fusion is unsplit only
  const float delta = (float)delta_in;                                /* :185 */
  int iter_in_sp = 0, iter_in_dp = 0, iter_out = 0, high_control = 0, j;
  double rho_dp, sourcesquarenorm, target_eps_sq;
  float rho_sp;
  tmhip_field *qhigh = ctx->sf[0], *rhigh = ctx->sf[1], *xhigh = ctx->sf[2], *phigh = ctx->sf_extra;   /* :211-214 */
  tmhip_field *q = ctx->sf32[0], *p = ctx->sf32[1], *r = ctx->sf32[2], *x = ctx->sf32[3];             /* :216-219 */
  if (tmhip_square_norm(ctx, Q, N, 1, &sourcesquarenorm)) return 1;
  target_eps_sq = rel_prec == 1 ? eps_sq * sourcesquarenorm : eps_sq;  /* :225-232 */
  const int N_outer = (int)ceil(log10(sourcesquarenorm * delta / target_eps_sq));   /* :236 */
  if (tmhip_field_zero(ctx, x) || tmhip_field_zero(ctx, P)) return 1;
  if (tmhip_assign(ctx, phigh, Q, N) || tmhip_assign(ctx, rhigh, Q, N)) return 1;
  if (tmhip_square_norm(ctx, rhigh, N, 1, &rho_dp)) return 1;
  if (tmhip_assign_to_32(ctx, r, rhigh, N)) return 1;
  rho_sp = (float)rho_dp;
  if (tmhip_assign_to_32(ctx, p, rhigh, N)) return 1;                   /* assign_32(p, r): same rounding of the same source */
#define RG_SP_LOOP()                                                                                                   \
  do {                                                                                                                 \
    double rho1 = rho_sp;                                                                                              \
    RgFields f = {x, p, q, r};                                                                                         \
    if (rg_inner_loop(ctx, op, true, fused, f, &rho1, delta, (double)(float)target_eps_sq, N,                          \
                      iter_out + iter_in_sp + iter_in_dp, max_iter, &j)) return 1;                                     \
    rho_sp = (float)rho1; iter_in_sp += j;                                                                             \
  } while (0)
#define RG_RETURN(val)                                                                                                 \
  do {                                                                                                                 \
    *iters = (val);                                                                                                    \
    if (iter_out_p) *iter_out_p = iter_out;                                                                            \
    if (iter_sp_p) *iter_sp_p = iter_in_sp;                                                                            \
    if (iter_dp_p) *iter_dp_p = iter_in_dp;                                                                            \
    return 0;                                                                                                          \
  } while (0)
  RG_SP_LOOP();
  for (iter_out = 1; iter_out < N_outer; ++iter_out) {
    if (high_control == 0) {                                           /* :260-269 */
      if (tmhip_add_from_32(ctx, P, x, N)) return 1;
      if (tmhip_apply_op(ctx, op, qhigh, P)) return 1;
      if (tmhip_diff(ctx, rhigh, Q, qhigh, N)) return 1;
      if (tmhip_square_norm(ctx, rhigh, N, 1, &rho_dp)) return 1;
    }
    if (high_control == 1) {                                           /* :272-286 double precision fail-safe */
      if (tmhip_assign(ctx, phigh, rhigh, N) || tmhip_field_zero(ctx, xhigh)) return 1;
      RgFields f = {xhigh, phigh, qhigh, rhigh};
      if (rg_inner_loop(ctx, op, false, false, f, &rho_dp, delta, target_eps_sq, N, iter_out + iter_in_sp + iter_in_dp, max_iter, &j)) return 1;
      iter_in_dp += j;
      rho_sp = (float)rho_dp;
      if (tmhip_assign_add_mul_r(ctx, P, xhigh, 1.0, N)) return 1;     /* add(P, P, xhigh) */
      if (tmhip_apply_op(ctx, op, qhigh, P)) return 1;
      if (tmhip_diff(ctx, rhigh, Q, qhigh, N)) return 1;
      if (tmhip_square_norm(ctx, rhigh, N, 1, &rho_dp)) return 1;
    }
    const int total = iter_in_sp + iter_in_dp + iter_out;
    if (rho_dp <= target_eps_sq || total >= max_iter) RG_RETURN(total >= max_iter ? -1 : total);   /* :294-307 */
    if (iter_out >= N_outer - 2) {                                     /* :310-313 */
      high_control = 1;
      continue;
    }
    if (tmhip_assign_to_32(ctx, r, rhigh, N) || tmhip_assign_to_32(ctx, p, rhigh, N)) return 1;   /* :315-318 */
    rho_sp = (float)rho_dp;
    if (tmhip_field_zero(ctx, x)) return 1;
    RG_SP_LOOP();
  }
  RG_RETURN(-1);                                                       /* :326-330 convergence failure */
#undef RG_SP_LOOP
#undef RG_RETURN
}
