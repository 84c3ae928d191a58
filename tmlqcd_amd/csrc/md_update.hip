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

// One kernel for "links -> stencil copy" with or without the molecular-dynamics update in front:
//   UPDATE   U_mu(x) <- restoresu3(exposu3(step P_mu(x))) U_mu(x)  (update_gauge.c:51-110), written back to the lexicographic field
//   always   the link goes straight to its TWO slots of the stencil's gauge copy g[par][dir][e][site]: dir 2 mu of its own site and
//            dir 2 mu + 1 of the site at x + mu (update_backward_gauge.c:185-242 turned from a gather into a scatter)
// so an MD step reads every link once and never re-reads the updated field for a separate re-sort (832 + 1152 -> 640 B per link
// site-direction).  A block owns 64 consecutive lexicographic sites = 256 links: the host-layout arrays (144-byte links, 64-byte
// momenta) are moved between HBM and LDS as whole contiguous rows, 16 bytes per lane (a lane-per-link access has a 144-byte lane
// stride: 363 us for 1.48 GB at 32^4, 0.51 of the peak); each thread then takes ONE link out of LDS (lane stride 36 words:
// conflict-free 16-byte reads; the momenta rows are padded to 80 bytes for the same reason).
// T-split ranks: the link U_t at t = T-1 is the backward link of a site of the up neighbour (not stored here), and the backward
// t-links of the t = 0 sites come from the t = -1 halo slab once the neighbours' slices have arrived (halo_backward_kernel).
template <bool UPDATE>
__global__ __launch_bounds__(256) void links_kernel(v2d *__restrict__ raw, const double *__restrict__ mom, v2d *__restrict__ g, int gs, int V, int Vh,
                                                    int T, int LX, int LY, int LZ, int toff, int split, double step) {
  __shared__ v2d sl[256 * 9];
  __shared__ v2d sm[UPDATE ? 256 * 5 : 1];
  const int tid = threadIdx.x;
  const size_t l0 = (size_t)blockIdx.x * 256;            // first link of the block
  const int nl = (int)(((size_t)V * 4 - l0) < 256 ? ((size_t)V * 4 - l0) : 256);
#pragma unroll
  for (int j = 0; j < 9; j++) {
    const int idx = j * 256 + tid;
    if (idx < nl * 9) sl[idx] = raw[l0 * 9 + idx];
  }
  if (UPDATE) {
    const v2d *m2 = reinterpret_cast<const v2d *>(mom) + l0 * 4;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int idx = j * 256 + tid;
      if (idx < nl * 4) sm[(idx >> 2) * 5 + (idx & 3)] = m2[idx];
    }
  }
  __syncthreads();
  if (UPDATE) {
    if (tid < nl) {
      cd z[9], out[9];
#pragma unroll
      for (int e = 0; e < 9; e++) z[e] = sl[tid * 9 + e];
      double d[8];
#pragma unroll
      for (int k = 0; k < 4; k++) { const v2d m = sm[tid * 5 + k]; d[2 * k] = step * m.x; d[2 * k + 1] = step * m.y; }   // _su3adj_assign_const_times_su3adj, update_gauge.c:85
      cd w[9], v[9];
      exposu3(w, d);
      restoresu3(v, w);
      m3mul(out, v, z);
#pragma unroll
      for (int e = 0; e < 9; e++) sl[tid * 9 + e] = out[e];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 9; j++) {
      const int idx = j * 256 + tid;
      if (idx < nl * 9) raw[l0 * 9 + idx] = sl[idx];
    }
  }
  // scatter into the stencil copy, re-mapped: thread = (parity p, direction mu, pair s) so that the 32 lanes of one (p, mu) write 32
  // consecutive sites of ONE plane -- 512 contiguous bytes per plane and store instruction instead of 128
  const int p = tid >> 7, mu = (tid >> 5) & 3, sp = tid & 31;
  const int ix0 = (int)(l0 >> 2);
  const int ie = ix0 + 2 * sp;                              // even member of the pair (z even)
  if (ie >= V) return;
  const int z0 = ie % LZ;
  int r = ie / LZ;
  const int y = r % LY;
  r /= LY;
  const int x = r % LX, t = r / LX;
  const int rp = (t + x + y + z0 + toff) & 1;               // parity of the even member
  const int ix = ie + (p ^ rp), z = z0 + (p ^ rp);          // the pair's site of parity p
  const v2d *src = sl + ((ix - ix0) * 4 + mu) * 9;
  cd out[9];
#pragma unroll
  for (int e = 0; e < 9; e++) out[e] = src[e];
  v2d *gf = g + ((size_t)p * 72 + (size_t)(2 * mu) * 9) * gs + (ix >> 1);
#pragma unroll
  for (int e = 0; e < 9; e++) __builtin_nontemporal_store(out[e], gf + (size_t)e * gs);   // (written once, read by the next stencil at the earliest)
  int jx;                                                  // x + mu, periodic
  if (mu == 0) {
    if (t + 1 < T) jx = ix + LX * LY * LZ;
    else { if (split) return; jx = ix - (T - 1) * LX * LY * LZ; }
  } else if (mu == 1) jx = (x + 1 < LX) ? ix + LY * LZ : ix - (LX - 1) * LY * LZ;
  else if (mu == 2) jx = (y + 1 < LY) ? ix + LZ : ix - (LY - 1) * LZ;
  else jx = (z + 1 < LZ) ? ix + 1 : ix - (LZ - 1);
  v2d *gb = g + ((size_t)(1 - p) * 72 + (size_t)(2 * mu + 1) * 9) * gs + (jx >> 1);
#pragma unroll
  for (int e = 0; e < 9; e++) __builtin_nontemporal_store(out[e], gb + (size_t)e * gs);
  (void)Vh;
}

// T-split rank: dir 1 (U_t(x - t)) of the t = 0 sites <- the t = -1 halo slab of the lexicographic field (geometry_eo.c:296-298)
__global__ __launch_bounds__(256) void halo_backward_kernel(const v2d *__restrict__ raw, v2d *__restrict__ g, int gs, int V, int XYZ, int LX, int LY, int LZ, int toff) {
  const int j = blockIdx.x * 256 + threadIdx.x;          // site of the t = 0 slice, lexicographic
  if (j >= XYZ) return;
  const int z = j % LZ;
  int r = j / LZ;
  const int y = r % LY, x = r / LY;
  const int par = (x + y + z + toff) & 1;
  const v2d *u = raw + ((size_t)(V + XYZ + j) * 4 + 0) * 9;
  v2d *gb = g + ((size_t)par * 72 + 9) * gs + (j >> 1);
#pragma unroll
  for (int e = 0; e < 9; e++) gb[(size_t)e * gs] = u[e];
  (void)LX;
}

// momenta[ix][mu][8] -= step * deriv[par][mu][8][Vh]   (update_momenta.c:67-72).  A block owns 64 consecutive lexicographic sites =
// the e/o indices [32 b, 32 b + 32) of both parities: the derivative's 2 x 32 planes are read as 256-byte runs and transposed
// through LDS (rows padded to 33 doubles: conflict-free), the momenta -- 16 KB of the host layout -- are updated as one contiguous run.
__global__ __launch_bounds__(256) void update_momenta_kernel(double *__restrict__ mom, const double *__restrict__ d, int V, int Vh, int LX, int LY, int LZ, int toff, double step) {
  __shared__ double ld[2][32][33];
  const int tid = threadIdx.x;
  const int i0 = blockIdx.x * 32;
#pragma unroll
  for (int par = 0; par < 2; par++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int idx = j * 256 + tid, e = idx >> 5, sidx = idx & 31;
      if (i0 + sidx < Vh) ld[par][sidx][e] = d[((size_t)par * 32 + e) * Vh + i0 + sidx];
    }
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * 64 * 32;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int idx = j * 256 + tid;                         // double of the block's momenta run
    const int sl = idx >> 5, e = idx & 31;
    const int ix = blockIdx.x * 64 + sl;
    if (ix >= V) continue;
    const int z = ix % LZ;
    int r = ix / LZ;
    const int y = r % LY;
    r /= LY;
    const int x = r % LX, t = r / LX;
    const int par = (t + x + y + z + toff) & 1;
    mom[base + idx] -= step * ld[par][sl >> 1][e];
  }
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
  if (ctx->shm) return tmhip_shm_ring(ctx, ctx->stream, first, last, slab_up, slab_dn, n * sizeof(double));
  TMHIP_NCCL_CHECK(ncclGroupStart());
  TMHIP_NCCL_CHECK(ncclSend(first, n, ncclDouble, dn, ctx->comm_red, ctx->stream));    // our t = 0 is the down neighbour's t = T
  TMHIP_NCCL_CHECK(ncclRecv(slab_up, n, ncclDouble, up, ctx->comm_red, ctx->stream));
  TMHIP_NCCL_CHECK(ncclSend(last, n, ncclDouble, up, ctx->comm_red, ctx->stream));     // our t = T-1 is the up neighbour's t = -1
  TMHIP_NCCL_CHECK(ncclRecv(slab_dn, n, ncclDouble, dn, ctx->comm_red, ctx->stream));
  TMHIP_NCCL_CHECK(ncclGroupEnd());
  return 0;
}

// links (-> updated links) -> stencil copy on ctx->stream; `halo`: the halo slabs of a T-split rank are current, finish the t = 0 sites
static int launch_links(tmhip_ctx *ctx, bool update, double step) {
  const int toff = ctx->g.proc_t * ctx->g.T, split = ctx->g.nproc_t > 1 ? 1 : 0;
  const unsigned nb = (unsigned)(((size_t)ctx->V * 4 + 255) / 256);
  if (update)
    hipLaunchKernelGGL(links_kernel<true>, dim3(nb), dim3(256), 0, ctx->stream, ctx->gauge_raw, (const double *)ctx->momenta, ctx->gauge, ctx->gs, ctx->V, ctx->Vh,
                       ctx->g.T, ctx->g.LX, ctx->g.LY, ctx->g.LZ, toff, split, step);
  else
    hipLaunchKernelGGL(links_kernel<false>, dim3(nb), dim3(256), 0, ctx->stream, ctx->gauge_raw, (const double *)nullptr, ctx->gauge, ctx->gs, ctx->V, ctx->Vh,
                       ctx->g.T, ctx->g.LX, ctx->g.LY, ctx->g.LZ, toff, split, 0.0);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}
static int launch_halo_backward(tmhip_ctx *ctx) {
  if (ctx->g.nproc_t < 2) return 0;
  const int XYZ = ctx->g.LX * ctx->g.LY * ctx->g.LZ;
  hipLaunchKernelGGL(halo_backward_kernel, dim3((XYZ + 255) / 256), dim3(256), 0, ctx->stream, (const v2d *)ctx->gauge_raw, ctx->gauge, ctx->gs, ctx->V, XYZ,
                     ctx->g.LX, ctx->g.LY, ctx->g.LZ, ctx->g.proc_t * ctx->g.T);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}
static void links_changed(tmhip_ctx *ctx) {
  ctx->gauge_set = true;
  ctx->gauge_copy_current = true;
  ctx->gauge32_set = false;       // the fp32 twin is rebuilt lazily from the new links
  ctx->gauge_recon_dev = -1.0;
}

// the stencil's gauge copy (and everything derived from the links) from the device-resident lexicographic field, halo slabs included
int tmhip_resort_gauge(tmhip_ctx *ctx) {
  if (!ctx->gauge_raw || !ctx->gauge_raw_valid) TMHIP_FAIL("no lexicographic gauge field on the device");
  if (launch_links(ctx, false, 0.0) || launch_halo_backward(ctx)) return 1;
  links_changed(ctx);
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
  hipLaunchKernelGGL(update_momenta_kernel, dim3((ctx->V + 63) / 64), dim3(256), 0, ctx->stream, ctx->momenta, (const double *)ctx->deriv,
                     ctx->V, ctx->Vh, ctx->g.LX, ctx->g.LY, ctx->g.LZ, ctx->g.proc_t * ctx->g.T, step);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}

/* update_gauge.c:51-110 on the device-resident links, then the halo slabs of a T-split rank (xchange_gauge) and the re-sort
 * of the stencil's gauge copy (update_backward_gauge.c:185-242) -- all in HBM */
int tmhip_update_gauge(tmhip_ctx *ctx, double step) {
  if (!ctx->gauge_raw || !ctx->gauge_raw_valid) TMHIP_FAIL("tmhip_update_gauge: the links are not resident (tmhip_set_gauge first)");
  if (!ctx->momenta) TMHIP_FAIL("tmhip_update_gauge: no momenta on the device (tmhip_momenta_upload)");
  TMHIP_CHECK(hipSetDevice(ctx->device));
  if (launch_links(ctx, true, step)) return 1;                 // exp(step P) U, back to the lexicographic field and into the stencil copy
  if (tmhip_exchange_gauge_halo(ctx)) return 1;
  if (launch_halo_backward(ctx)) return 1;                     // T-split: the backward t-links of the t = 0 sites from the neighbour's updated slice
  // clover blocks belong to the old links: tmhip_sw_term (gauge = NULL: from the resident links) / tmhip_sw_invert again
  ctx->sw_set = false; ctx->clover_set = false; ctx->clover32_set = false;
  links_changed(ctx);
  return 0;
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
    if (launch_links(c, true, step)) return 1;
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
    if (launch_halo_backward(c)) return 1;
    links_changed(c);
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
