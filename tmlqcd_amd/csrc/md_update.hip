// Molecular-dynamics updates with the links resident in HBM ("gauge-copy refresh per MD step"):
//   update_gauge   (update_gauge.c:51-110)   U_mu(x) <- restoresu3(exposu3(step * P_mu(x))) U_mu(x) for every link
//   update_momenta (update_momenta.c:67-72)  P -= step * derivative
// The lexicographic gauge field ([VPR][4][9] complex, what tmhip_set_gauge received) stays on the device, the momenta
// ([V][4][8] doubles, hamiltonian_field_t::momenta) are uploaded once per trajectory or kept resident, and the stencil's
// gauge copy is re-sorted from the updated links on the device: no host <-> device copy of the gauge field per MD step.
#include "tmhip_internal.h"

namespace {
typedef v2d cd;   // .x = re, .y = im
__device__ __forceinline__ cd cmul(cd a, cd b) { return cd{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cd cconj(cd a) { return cd{a.x, -a.y}; }
__device__ __forceinline__ cd rmul(double r, cd a) { return cd{r * a.x, r * a.y}; }

// u = v w   (su3.h:583-592: each element the sum of three products, left to right)
__device__ __forceinline__ void m3mul(cd (&u)[9], const cd (&v)[9], const cd (&w)[9]) {
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) u[3 * i + j] = cmul(v[3 * i], w[j]) + cmul(v[3 * i + 1], w[3 + j]) + cmul(v[3 * i + 2], w[6 + j]);
}

// expo.c:56-97 (Cayley-Hamilton form of exp(v), v anti-hermitian traceless from the su3adj vector p = d1..d8)
__device__ __forceinline__ void exposu3(cd (&vr)[9], const double (&p)[8]) {
  cd v[9], v2[9];
  const double d1 = p[0], d2 = p[1], d3 = p[2], d4 = p[3], d5 = p[4], d6 = p[5], d7 = p[6], d8 = p[7];
  v[0] = cd{0.0, 0.5773502691896258 * d8 + d3};     // _make_su3, su3adj.h:45-54
  v[1] = cd{d2, d1};
  v[2] = cd{d5, d4};
  v[3] = cd{-d2, d1};
  v[4] = cd{0.0, 0.5773502691896258 * d8 - d3};
  v[5] = cd{d7, d6};
  v[6] = cd{-d5, d4};
  v[7] = cd{-d7, d6};
  v[8] = cd{0.0, -(1.154700538379252 * d8)};
  m3mul(v2, v, v);
  const double a = 0.5 * (v2[0].x + v2[4].x + v2[8].x);
  // 1/3 Im tr(v v2): the nine products in the order of expo.c:70-72
  const cd tr = cmul(v[0], v2[0]) + cmul(v[1], v2[3]) + cmul(v[2], v2[6]) + cmul(v[3], v2[1]) + cmul(v[4], v2[4]) + cmul(v[5], v2[7]) +
                cmul(v[6], v2[2]) + cmul(v[7], v2[5]) + cmul(v[8], v2[8]);
  const double b = 0.33333333333333333 * tr.y;
  cd a0 = cd{0.16059043836821615e-9, 0.0}, a1 = cd{0.11470745597729725e-10, 0.0}, a2 = cd{0.76471637318198165e-12, 0.0};
  double fac = 0.20876756987868099e-8, r = 12.0;
#pragma unroll
  for (int i = 3; i <= 15; ++i) {
    const cd a1p = a0 + rmul(a, a2);
    a0 = cd{fac - b * a2.y, b * a2.x};              // fac + (b i) a2
    a2 = a1;
    a1 = a1p;
    fac *= r;
    r -= 1.0;
  }
#pragma unroll
  for (int e = 0; e < 9; e++) vr[e] = cmul(a1, v[e]) + cmul(a2, v2[e]);
  vr[0] += a0; vr[4] += a0; vr[8] += a0;
}

// expo.c:118-137: rows 0 and 1 normalised, row 2 = conj(row0 x row1)
__device__ __forceinline__ void restoresu3(cd (&vr)[9], const cd (&u)[9]) {
  const double n0 = 1.0 / sqrt((u[0].x * u[0].x + u[0].y * u[0].y) + (u[1].x * u[1].x + u[1].y * u[1].y) + (u[2].x * u[2].x + u[2].y * u[2].y));
  const double n1 = 1.0 / sqrt((u[3].x * u[3].x + u[3].y * u[3].y) + (u[4].x * u[4].x + u[4].y * u[4].y) + (u[5].x * u[5].x + u[5].y * u[5].y));
#pragma unroll
  for (int e = 0; e < 3; e++) { vr[e] = rmul(n0, u[e]); vr[3 + e] = rmul(n1, u[3 + e]); }
  vr[6] = cconj(cmul(vr[1], vr[5]) - cmul(vr[2], vr[4]));
  vr[7] = cconj(cmul(vr[2], vr[3]) - cmul(vr[0], vr[5]));
  vr[8] = cconj(cmul(vr[0], vr[4]) - cmul(vr[1], vr[3]));
}

// one thread per link l = 4 ix + mu of the local volume
__global__ __launch_bounds__(256) void update_gauge_kernel(v2d *__restrict__ raw, const double *__restrict__ mom, size_t nlinks, double step) {
  const size_t l = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (l >= nlinks) return;
  double d[8];
#pragma unroll
  for (int k = 0; k < 8; k++) d[k] = step * mom[l * 8 + k];     // _su3adj_assign_const_times_su3adj, update_gauge.c:85
  cd w[9], v[9], z[9], out[9];
  exposu3(w, d);
  restoresu3(v, w);
#pragma unroll
  for (int e = 0; e < 9; e++) z[e] = raw[l * 9 + e];
  m3mul(out, v, z);
#pragma unroll
  for (int e = 0; e < 9; e++) raw[l * 9 + e] = out[e];
}

// momenta[ix][mu][8] -= step * deriv[par][mu][8][Vh]   (update_momenta.c:67-72); one thread per site of one parity
__global__ __launch_bounds__(256) void update_momenta_kernel(double *__restrict__ mom, const double *__restrict__ d, int Vh, int LX, int LY, int LZ, int toff, double step) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Vh) return;
  const int par = blockIdx.y;
  const int LZh = LZ / 2;
  int r = i / LZh;
  const int y = r % LY;
  r /= LY;
  const int x = r % LX, t = r / LX;
  const int o = (t + x + y + par + toff) & 1;
  double *dst = mom + (2 * (size_t)i + o) * 32;
  const double *src = d + (size_t)par * 32 * Vh + i;
#pragma unroll 8
  for (int e = 0; e < 32; e++) dst[e] -= step * src[(size_t)e * Vh];
}

// t = 0 and t = T-1 slices of the lexicographic field -> the neighbours' halo slabs (xchange_gauge, geometry_eo.c:292-299)
}  // namespace
int tmhip_exchange_gauge_halo(tmhip_ctx *ctx) {
  if (ctx->g.nproc_t < 2) return 0;
  if (!ctx->comm_ready) TMHIP_FAIL("nproc_t > 1 but tmhip_comm_init was not called");
  const size_t XYZ = (size_t)ctx->g.LX * ctx->g.LY * ctx->g.LZ, n = XYZ * 36 * 2;   // doubles per slice
  v2d *raw = ctx->gauge_raw;
  v2d *first = raw, *last = raw + (size_t)(ctx->g.T - 1) * XYZ * 36;
  v2d *slab_up = raw + (size_t)ctx->V * 36, *slab_dn = slab_up + XYZ * 36;        // t = T, t = -1
  const int np = ctx->g.nproc_t, up = (ctx->g.proc_t + 1) % np, dn = (ctx->g.proc_t + np - 1) % np;
  TMHIP_NCCL_CHECK(ncclGroupStart());
  TMHIP_NCCL_CHECK(ncclSend(first, n, ncclDouble, dn, ctx->comm_red, ctx->stream));    // our t = 0 is the down neighbour's t = T
  TMHIP_NCCL_CHECK(ncclRecv(slab_up, n, ncclDouble, up, ctx->comm_red, ctx->stream));
  TMHIP_NCCL_CHECK(ncclSend(last, n, ncclDouble, up, ctx->comm_red, ctx->stream));     // our t = T-1 is the up neighbour's t = -1
  TMHIP_NCCL_CHECK(ncclRecv(slab_dn, n, ncclDouble, dn, ctx->comm_red, ctx->stream));
  TMHIP_NCCL_CHECK(ncclGroupEnd());
  return 0;
}

extern "C" {

int tmhip_momenta_upload(tmhip_ctx *ctx, const void *host) {
  if (!host) TMHIP_FAIL("tmhip_momenta_upload: null argument");
  TMHIP_CHECK(hipSetDevice(ctx->device));
  const size_t bytes = (size_t)ctx->V * 32 * sizeof(double);
  if (!ctx->momenta) TMHIP_CHECK(hipMalloc((void **)&ctx->momenta, bytes));
  TMHIP_CHECK(hipMemcpyAsync(ctx->momenta, host, bytes, hipMemcpyHostToDevice, ctx->stream));
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  return 0;
}

int tmhip_momenta_download(tmhip_ctx *ctx, void *host) {
  if (!host) TMHIP_FAIL("tmhip_momenta_download: null argument");
  if (!ctx->momenta) TMHIP_FAIL("tmhip_momenta_download: no momenta on the device");
  TMHIP_CHECK(hipSetDevice(ctx->device));
  TMHIP_CHECK(hipMemcpyAsync(host, ctx->momenta, (size_t)ctx->V * 32 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  return 0;
}

/* update_momenta.c:67-72 with the derivative accumulated on the device by tmhip_deriv_Sb / tmhip_sw_all */
int tmhip_update_momenta(tmhip_ctx *ctx, double step) {
  if (!ctx->momenta) TMHIP_FAIL("tmhip_update_momenta: no momenta on the device (tmhip_momenta_upload)");
  if (!ctx->deriv) TMHIP_FAIL("tmhip_update_momenta: no derivative field on the device");
  TMHIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(update_momenta_kernel, dim3((ctx->Vh + 255) / 256, 2), dim3(256), 0, ctx->stream, ctx->momenta, (const double *)ctx->deriv,
                     ctx->Vh, ctx->g.LX, ctx->g.LY, ctx->g.LZ, ctx->g.proc_t * ctx->g.T, step);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}

/* update_gauge.c:51-110 on the device-resident links, then the halo slabs of a T-split rank (xchange_gauge) and the re-sort
 * of the stencil's gauge copy (update_backward_gauge.c:185-242) -- all in HBM */
int tmhip_update_gauge(tmhip_ctx *ctx, double step) {
  if (!ctx->gauge_raw || !ctx->gauge_raw_valid) TMHIP_FAIL("tmhip_update_gauge: the links are not resident (tmhip_set_gauge first)");
  if (!ctx->momenta) TMHIP_FAIL("tmhip_update_gauge: no momenta on the device (tmhip_momenta_upload)");
  TMHIP_CHECK(hipSetDevice(ctx->device));
  const size_t nlinks = (size_t)ctx->V * 4;
  hipLaunchKernelGGL(update_gauge_kernel, dim3((unsigned)((nlinks + 255) / 256)), dim3(256), 0, ctx->stream, ctx->gauge_raw, (const double *)ctx->momenta, nlinks, step);
  TMHIP_CHECK(hipGetLastError());
  if (tmhip_exchange_gauge_halo(ctx)) return 1;
  // clover blocks belong to the old links: tmhip_sw_term (gauge = NULL: from the resident links) / tmhip_sw_invert again
  ctx->sw_set = false; ctx->clover_set = false; ctx->clover32_set = false;
  return tmhip_resort_gauge(ctx);
}

/* The same on a T-split lattice held by n contexts of one process (peer copies instead of RCCL, as tmhip_multi_sw_all): every
 * rank updates its links, the t = 0 / t = T-1 slices travel to the ring neighbours' halo slabs, the stencil copies are re-sorted. */
int tmhip_multi_update_gauge(int n, tmhip_ctx **ctxs, double step) {
  if (n < 2) TMHIP_FAIL("tmhip_multi_update_gauge needs >= 2 contexts");
  for (int r = 0; r < n; r++) {
    tmhip_ctx *c = ctxs[r];
    if (c->g.nproc_t != n || c->g.proc_t != r) TMHIP_FAIL("context %d is not rank %d of a %d-way T split", r, r, n);
    if (!c->gauge_raw || !c->gauge_raw_valid || !c->momenta) TMHIP_FAIL("tmhip_multi_update_gauge: rank %d has no resident links / momenta", r);
    TMHIP_CHECK(hipSetDevice(c->device));
    const size_t nlinks = (size_t)c->V * 4;
    hipLaunchKernelGGL(update_gauge_kernel, dim3((unsigned)((nlinks + 255) / 256)), dim3(256), 0, c->stream, c->gauge_raw, (const double *)c->momenta, nlinks, step);
    TMHIP_CHECK(hipGetLastError());
  }
  for (int r = 0; r < n; r++) { TMHIP_CHECK(hipSetDevice(ctxs[r]->device)); TMHIP_CHECK(hipStreamSynchronize(ctxs[r]->stream)); }
  const size_t XYZ = (size_t)ctxs[0]->g.LX * ctxs[0]->g.LY * ctxs[0]->g.LZ, sb = XYZ * 36 * sizeof(v2d);
  for (int r = 0; r < n; r++) {
    tmhip_ctx *c = ctxs[r], *up = ctxs[(r + 1) % n], *dn = ctxs[(r + n - 1) % n];
    TMHIP_CHECK(hipSetDevice(c->device));
    v2d *slab_up = c->gauge_raw + (size_t)c->V * 36, *slab_dn = slab_up + XYZ * 36;                                    // t = T, t = -1
    TMHIP_CHECK(hipMemcpyPeerAsync(slab_up, c->device, up->gauge_raw, up->device, sb, c->stream));                                            // the up neighbour's t = 0
    TMHIP_CHECK(hipMemcpyPeerAsync(slab_dn, c->device, dn->gauge_raw + (size_t)(dn->g.T - 1) * XYZ * 36, dn->device, sb, c->stream));         // the down neighbour's t = T-1
    c->sw_set = false; c->clover_set = false; c->clover32_set = false;
    if (tmhip_resort_gauge(c)) return 1;
  }
  for (int r = 0; r < n; r++) { TMHIP_CHECK(hipSetDevice(ctxs[r]->device)); TMHIP_CHECK(hipStreamSynchronize(ctxs[r]->stream)); }
  return 0;
}

/* the device-resident links in the host layout of g_gauge_field ([VOLUMEPLUSRAND][4] su3) */
int tmhip_gauge_download(tmhip_ctx *ctx, void *host) {
  if (!host) TMHIP_FAIL("tmhip_gauge_download: null argument");
  if (!ctx->gauge_raw || !ctx->gauge_raw_valid) TMHIP_FAIL("tmhip_gauge_download: the links are not resident");
  TMHIP_CHECK(hipSetDevice(ctx->device));
  TMHIP_CHECK(hipMemcpyAsync(host, ctx->gauge_raw, (size_t)ctx->VPR * 36 * sizeof(v2d), hipMemcpyDeviceToHost, ctx->stream));
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  return 0;
}

}  // extern "C"
