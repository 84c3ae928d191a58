// Clover twisted mass on top of the stencil (invert_clover_eo.c:63-165).
//
// The site-local 6x6 blocks are INPUTS, exactly like the gauge field: the host computes
//   sw     = 1 + T            su3 sw[VOLUME][3][2]      (sw_term,   operator/clover_term.c:88-200)
//   sw_inv = (1+T+-i mu g5)^-1 su3 sw_inv[VOLUME][4][2]  (sw_invert, operator/clover_invert.c:170-257; even sites,
//                                                        +mu in [0,V/2), -mu in [V/2,V))
// and tmhip_set_clover() re-sorts them into SoA device arrays.  clover_inv / clover_gamma5 / clover are fused into the
// stencil epilogues (EPI_CLOVER_*), so Qsw_pm_psi = 4 launches; stand-alone site kernels serve the drop-in symbols.
#include "tmhip_internal.h"

// sw[ix][a][b] (lexicographic) -> swd[par][2a+b][e][i]
__global__ __launch_bounds__(256) void sw_sort_kernel(const v2d *__restrict__ raw, v2d *__restrict__ d, int gs, int Vh, int LX, int LY,
                                                      int LZ, int toff) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Vh) return;
  const int par = blockIdx.y;
  const int LZh = LZ / 2;
  int r = i / LZh;
  const int y = r % LY;
  r /= LY;
  const int x = r % LX, t = r / LX;
  const int o = (t + x + y + toff + par) & 1;
  const size_t ix = 2 * (size_t)i + o;
  const v2d *src = raw + ix * 54;
  v2d *dst = d + (size_t)par * 54 * gs + i;
#pragma unroll 6
  for (int e = 0; e < 54; e++) dst[(size_t)e * gs] = src[e];
}
// sw_inv[icy][a][b], icy = e/o index of the even site (+ V/2 for the -mu set) -> swinv[sign][2a+b][e][i]
__global__ __launch_bounds__(256) void swinv_sort_kernel(const v2d *__restrict__ raw, v2d *__restrict__ d, int gs, int Vh) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Vh) return;
  const int sign = blockIdx.y;
  const v2d *src = raw + ((size_t)sign * Vh + i) * 72;
  v2d *dst = d + (size_t)sign * 72 * gs + i;
#pragma unroll 6
  for (int e = 0; e < 72; e++) dst[(size_t)e * gs] = src[e];
}
__global__ void clover_to32_kernel(v2f *__restrict__ d, const v2d *__restrict__ s, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const v2d a = s[i]; d[i] = v2f{(float)a.x, (float)a.y}; }
}

__device__ __forceinline__ v2d cl_cfma(v2d a, v2d b, v2d c) { return v2d{c.x + a.x * b.x - a.y * b.y, c.y + a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ v2d cl_cfmac(v2d a, v2d b, v2d c) { return v2d{c.x + a.x * b.x + a.y * b.y, c.y + a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ void cl_block(v2d (&r)[3], const v2d *__restrict__ w, size_t gs, int i, int blk, const v2d *s, bool dagger, bool acc) {
  v2d u[9];
#pragma unroll
  for (int e = 0; e < 9; e++) u[e] = w[((size_t)blk * 9 + e) * gs + i];
#pragma unroll
  for (int row = 0; row < 3; row++) {
    v2d t = acc ? r[row] : v2d{0.0, 0.0};
    if (dagger) t = cl_cfmac(u[6 + row], s[2], cl_cfmac(u[3 + row], s[1], cl_cfmac(u[row], s[0], t)));
    else t = cl_cfma(u[3 * row + 2], s[2], cl_cfma(u[3 * row + 1], s[1], cl_cfma(u[3 * row], s[0], t)));
    r[row] = t;
  }
}

// MODE 0: l = W_inv l (clover_inv, in place)   1: l = g5((1+T+i mu g5) k - j)   2: same without g5
template <int MODE>
__global__ __launch_bounds__(256) void clover_site_kernel(v2d *L, const v2d *K, const v2d *J, const v2d *__restrict__ w, int ns, int gs,
                                                          int N, double mu) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
#pragma unroll
  for (int b = 0; b < 2; b++) {
    v2d sa[3], sb[3], r1[3], r2[3];
#pragma unroll
    for (int c = 0; c < 3; c++) { sa[c] = K[(size_t)(6 * b + c) * ns + i]; sb[c] = K[(size_t)(6 * b + 3 + c) * ns + i]; }
    if (MODE == 0) {
      cl_block(r1, w, gs, i, 0 * 2 + b, sa, false, false); cl_block(r1, w, gs, i, 1 * 2 + b, sb, false, true);
      cl_block(r2, w, gs, i, 3 * 2 + b, sa, false, false); cl_block(r2, w, gs, i, 2 * 2 + b, sb, false, true);
#pragma unroll
      for (int c = 0; c < 3; c++) { L[(size_t)(6 * b + c) * ns + i] = r1[c]; L[(size_t)(6 * b + 3 + c) * ns + i] = r2[c]; }
    } else {
      cl_block(r1, w, gs, i, 0 * 2 + b, sa, false, false); cl_block(r1, w, gs, i, 1 * 2 + b, sb, false, true);
      cl_block(r2, w, gs, i, 1 * 2 + b, sa, true, false);  cl_block(r2, w, gs, i, 2 * 2 + b, sb, false, true);
      const double m = b == 0 ? mu : -mu;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        r1[c] = v2d{r1[c].x - m * sa[c].y, r1[c].y + m * sa[c].x};
        r2[c] = v2d{r2[c].x - m * sb[c].y, r2[c].y + m * sb[c].x};
        const v2d j1 = J[(size_t)(6 * b + c) * ns + i], j2 = J[(size_t)(6 * b + 3 + c) * ns + i];
        const bool flip = MODE == 1 && b == 1;
        L[(size_t)(6 * b + c) * ns + i] = flip ? j1 - r1[c] : r1[c] - j1;
        L[(size_t)(6 * b + 3 + c) * ns + i] = flip ? j2 - r2[c] : r2[c] - j2;
      }
    }
  }
}

static int need64(const tmhip_field *f, const char *who) {
  if (!f || f->kind != TMHIP_FIELD_EO || f->prec != 0) { fprintf(stderr, "[tmlqcd_hip] %s: needs a one-parity fp64 field\n", who); return 1; }
  return 0;
}
static inline const v2d *swinv(tmhip_ctx *ctx, int tau3sign, double mu) {   /* clovertm_operators.c:298-300 */
  return ctx->sw_inv + (size_t)((tau3sign < 0 && fabs(mu) > 0) ? 1 : 0) * 72 * ctx->gs;
}
static inline const v2d *swpar(tmhip_ctx *ctx, int ieo) { return ctx->sw + (size_t)(ieo ? 1 : 0) * 54 * ctx->gs; }

int tmhip_prepare_clover32(tmhip_ctx *ctx) {
  if (!ctx->clover_set) TMHIP_FAIL("clover operator called before tmhip_set_clover");
  if (tmhip_prepare_fp32(ctx)) return 1;
  const size_t n1 = (size_t)2 * 54 * ctx->gs, n2 = (size_t)2 * 72 * ctx->gs;
  if (!ctx->sw32) TMHIP_CHECK(hipMalloc((void **)&ctx->sw32, n1 * sizeof(v2f)));
  if (!ctx->sw_inv32) TMHIP_CHECK(hipMalloc((void **)&ctx->sw_inv32, n2 * sizeof(v2f)));
  if (!ctx->clover32_set) {   /* copy_32_sw_fields (operator.c:367) */
    hipLaunchKernelGGL(clover_to32_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, ctx->stream, ctx->sw32, ctx->sw, n1);
    hipLaunchKernelGGL(clover_to32_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, ctx->stream, ctx->sw_inv32, ctx->sw_inv, n2);
    TMHIP_CHECK(hipGetLastError());
    ctx->clover32_set = true;
  }
  return 0;
}

extern "C" {

int tmhip_set_clover(tmhip_ctx *ctx, const void *sw_host, const void *sw_inv_host) {
  if (!sw_host || !sw_inv_host) TMHIP_FAIL("tmhip_set_clover: null argument");
  if (ctx->g.nproc_t > 1) TMHIP_FAIL("tmhip_set_clover: single-rank lattices only in this round");
  TMHIP_CHECK(hipSetDevice(ctx->device));
  const size_t n1 = (size_t)2 * 54 * ctx->gs, n2 = (size_t)2 * 72 * ctx->gs;
  if (!ctx->sw) TMHIP_CHECK(hipMalloc((void **)&ctx->sw, n1 * sizeof(v2d)));
  if (!ctx->sw_inv) TMHIP_CHECK(hipMalloc((void **)&ctx->sw_inv, n2 * sizeof(v2d)));
  const size_t b1 = (size_t)ctx->V * 54 * sizeof(v2d), b2 = (size_t)ctx->V * 72 * sizeof(v2d);
  void *raw = nullptr;
  TMHIP_CHECK(hipMalloc(&raw, b2));
  TMHIP_CHECK(hipMemcpyAsync(raw, sw_host, b1, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(sw_sort_kernel, dim3((ctx->Vh + 255) / 256, 2), dim3(256), 0, ctx->stream, (const v2d *)raw, ctx->sw, ctx->gs, ctx->Vh,
                     ctx->g.LX, ctx->g.LY, ctx->g.LZ, ctx->g.proc_t * ctx->g.T);
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  TMHIP_CHECK(hipMemcpyAsync(raw, sw_inv_host, b2, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(swinv_sort_kernel, dim3((ctx->Vh + 255) / 256, 2), dim3(256), 0, ctx->stream, (const v2d *)raw, ctx->sw_inv, ctx->gs, ctx->Vh);
  TMHIP_CHECK(hipGetLastError());
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  TMHIP_CHECK(hipFree(raw));
  ctx->clover_set = true;
  ctx->clover32_set = false;
  return 0;
}

/* clovertm_operators.c:287-350 */
int tmhip_clover_inv(tmhip_ctx *ctx, tmhip_field *l, int tau3sign, double mu) {
  if (need64(l, "clover_inv")) return 1;
  if (!ctx->clover_set) TMHIP_FAIL("clover_inv called before tmhip_set_clover");
  hipLaunchKernelGGL(clover_site_kernel<0>, dim3((ctx->Vh + 255) / 256), dim3(256), 0, ctx->stream, l->d, (const v2d *)l->d, (const v2d *)nullptr,
                     swinv(ctx, tau3sign, mu), l->ns, ctx->gs, ctx->Vh, 0.0);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}
/* clovertm_operators.c:448-520 */
int tmhip_clover_gamma5(tmhip_ctx *ctx, int ieo, tmhip_field *l, tmhip_field *k, tmhip_field *j, double mu) {
  if (need64(l, "clover_gamma5") || need64(k, "clover_gamma5") || need64(j, "clover_gamma5")) return 1;
  if (!ctx->clover_set) TMHIP_FAIL("clover_gamma5 called before tmhip_set_clover");
  hipLaunchKernelGGL(clover_site_kernel<1>, dim3((ctx->Vh + 255) / 256), dim3(256), 0, ctx->stream, l->d, (const v2d *)k->d, (const v2d *)j->d,
                     swpar(ctx, ieo), l->ns, ctx->gs, ctx->Vh, mu);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}
/* clovertm_operators.c:535-600 */
int tmhip_clover(tmhip_ctx *ctx, int ieo, tmhip_field *l, tmhip_field *k, tmhip_field *j, double mu) {
  if (need64(l, "clover") || need64(k, "clover") || need64(j, "clover")) return 1;
  if (!ctx->clover_set) TMHIP_FAIL("clover called before tmhip_set_clover");
  hipLaunchKernelGGL(clover_site_kernel<2>, dim3((ctx->Vh + 255) / 256), dim3(256), 0, ctx->stream, l->d, (const v2d *)k->d, (const v2d *)j->d,
                     swpar(ctx, ieo), l->ns, ctx->gs, ctx->Vh, mu);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}
/* clovertm_operators.c:268-272 */
int tmhip_H_eo_sw_inv_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k, int ieo, int tau3sign, double mu) {
  if (need64(l, "H_eo_sw_inv_psi") || need64(k, "H_eo_sw_inv_psi")) return 1;
  if (!ctx->clover_set) TMHIP_FAIL("H_eo_sw_inv_psi called before tmhip_set_clover");
  return tmhip_launch_hopping(ctx, ieo, l->d, k->d, nullptr, EPI_CLOVER_INV, 0, 0, true, swinv(ctx, tau3sign, mu));
}
/* clovertm_operators.c:233-245 (g_mu3 = 0): 4 stencil launches with the clover blocks applied in the epilogues */
int tmhip_Qsw_pm_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k) {
  if (need64(l, "Qsw_pm_psi") || need64(k, "Qsw_pm_psi")) return 1;
  if (!ctx->clover_set) TMHIP_FAIL("Qsw_pm_psi called before tmhip_set_clover");
  const double mu = ctx->mu;
  v2d *s0 = ctx->scratch[0]->d, *s1 = ctx->scratch[1]->d;
  return tmhip_launch_hopping(ctx, TMHIP_EO, s1, k->d, nullptr, EPI_CLOVER_INV, 0, 0, true, swinv(ctx, -1, mu)) ||
         tmhip_launch_hopping(ctx, TMHIP_OE, s0, s1, k->d, EPI_CLOVER_G5, 0, -mu, true, swpar(ctx, TMHIP_OE)) ||
         tmhip_launch_hopping(ctx, TMHIP_EO, s1, s0, nullptr, EPI_CLOVER_INV, 0, 0, true, swinv(ctx, +1, mu)) ||
         tmhip_launch_hopping(ctx, TMHIP_OE, l->d, s1, s0, EPI_CLOVER_G5, 0, +mu, true, swpar(ctx, TMHIP_OE));
}
/* clovertm_operators.c:256-261 */
int tmhip_Msw_plus_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k) {
  if (need64(l, "Msw_plus_psi") || need64(k, "Msw_plus_psi")) return 1;
  if (!ctx->clover_set) TMHIP_FAIL("Msw_plus_psi called before tmhip_set_clover");
  const double mu = ctx->mu;
  v2d *s1 = ctx->scratch[1]->d;
  return tmhip_launch_hopping(ctx, TMHIP_EO, s1, k->d, nullptr, EPI_CLOVER_INV, 0, 0, true, swinv(ctx, +1, mu)) ||
         tmhip_launch_hopping(ctx, TMHIP_OE, l->d, s1, k->d, EPI_CLOVER, 0, +mu, true, swpar(ctx, TMHIP_OE));
}
/* clovertm_operators_32.c Qsw_pm_psi_32 */
int tmhip_Qsw_pm_psi_32(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k) {
  if (!l || !k || l->prec != 1 || k->prec != 1) TMHIP_FAIL("Qsw_pm_psi_32 needs fp32 fields");
  if (tmhip_prepare_clover32(ctx)) return 1;
  const double mu = ctx->mu;
  const size_t gs = ctx->gs;
  const v2f *wim = ctx->sw_inv32 + (size_t)(fabs(mu) > 0 ? 1 : 0) * 72 * gs, *wip = ctx->sw_inv32, *wo = ctx->sw32 + (size_t)54 * gs;
  v2f *s0 = ctx->scratch32[0]->d32, *s1 = ctx->scratch32[1]->d32;
  return tmhip_launch_hopping32(ctx, TMHIP_EO, s1, k->d32, nullptr, EPI_CLOVER_INV, 0, 0, true, wim) ||
         tmhip_launch_hopping32(ctx, TMHIP_OE, s0, s1, k->d32, EPI_CLOVER_G5, 0, -mu, true, wo) ||
         tmhip_launch_hopping32(ctx, TMHIP_EO, s1, s0, nullptr, EPI_CLOVER_INV, 0, 0, true, wip) ||
         tmhip_launch_hopping32(ctx, TMHIP_OE, l->d32, s1, s0, EPI_CLOVER_G5, 0, +mu, true, wo);
}

}  // extern "C"
