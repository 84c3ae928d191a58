// Context, device fields, host<->device layout conversion, operator compositions, CG,
// halo exchange and the benchmark loop of libtmlqcd_hip.so (gfx950).
#include "tmhip_internal.h"
#include <cmath>
#include <cstring>
#include <new>

// ------------------------------------------------------------------ layout kernels
// host AoS spinor[n] = v2d[n][12]  <->  device SoA [12][ns]
__global__ __launch_bounds__(256) void aos_to_soa_kernel(const v2d *__restrict__ aos, v2d *__restrict__ soa, int ns, int n) {
  const long tid = (long)blockIdx.x * 256 + threadIdx.x;
  if (tid >= 12L * n) return;
  const int site = (int)(tid / 12), c = (int)(tid % 12);
  soa[(size_t)c * ns + site] = aos[tid];
}
__global__ __launch_bounds__(256) void soa_to_aos_kernel(const v2d *__restrict__ soa, v2d *__restrict__ aos, int ns, int n) {
  const long tid = (long)blockIdx.x * 256 + threadIdx.x;
  if (tid >= 12L * n) return;
  const int site = (int)(tid / 12), c = (int)(tid % 12);
  aos[tid] = soa[(size_t)c * ns + site];
}
// lexicographic spinor[V] <-> two e/o halves (even at soa, odd at soa + 12*ns).
// parity from global coordinates (geometry_eo.c:807-811); e/o sub-index = ix/2 (LZ even).
template <bool TO_DEVICE>
__global__ __launch_bounds__(256) void lexic_eo_kernel(v2d *aos, v2d *soa, int ns, int V, int LX, int LY, int LZ, int toff) {
  const long tid = (long)blockIdx.x * 256 + threadIdx.x;
  if (tid >= 12L * V) return;
  const int ix = (int)(tid / 12), c = (int)(tid % 12);
  const int z = ix % LZ;
  int r = ix / LZ;
  const int y = r % LY;
  r /= LY;
  const int x = r % LX, t = r / LX;
  const int par = (t + x + y + z + toff) & 1;
  v2d *dst = soa + (size_t)par * 12 * ns + (size_t)c * ns + (ix >> 1);
  if (TO_DEVICE) *dst = aos[tid];
  else aos[tid] = *dst;
}

// (raw g_gauge_field -> the stencil's gauge copy g[par][dir][e][i]: links_kernel, md_update.hip)

// max over all links of |row2 - conj(row0 x row1)|: how far the resident links are from exact SU(3).  Non-negative
// doubles order like their bit patterns, so the maximum is an integer atomicMax.
__global__ __launch_bounds__(256) void gauge_recon_dev_kernel(const v2d *__restrict__ g, int gs, int Vh, unsigned long long *out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  double dev = 0.0;
  if (i < Vh) {
    const v2d *gp = g + (size_t)blockIdx.y * 9 * gs + i;    // blockIdx.y = parity * 8 + direction
    v2d u[9];
#pragma unroll
    for (int e = 0; e < 9; e++) u[e] = gp[(size_t)e * gs];
    auto cm = [](v2d a, v2d b) { return v2d{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; };
    const v2d a0 = cm(u[1], u[5]) - cm(u[2], u[4]), a1 = cm(u[2], u[3]) - cm(u[0], u[5]), a2 = cm(u[0], u[4]) - cm(u[1], u[3]);
    dev = fmax(fmax(fabs(u[6].x - a0.x), fabs(u[6].y + a0.y)), fmax(fmax(fabs(u[7].x - a1.x), fabs(u[7].y + a1.y)),
                                                                    fmax(fabs(u[8].x - a2.x), fabs(u[8].y + a2.y))));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) dev = fmax(dev, __shfl_xor(dev, off, 64));
  if ((threadIdx.x & 63) == 0 && dev > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(dev));
}

// ------------------------------------------------------------------ helpers
int tmhip_check_async_error(tmhip_ctx *ctx) {
  if (ctx->hop_seq) {   // a split-phase stencil has run: did a bounded wait for the faces give up (wave_wait_flag, hopping_impl.inc)?
    unsigned int err = 0;
    TMHIP_CHECK(hipMemcpyAsync(&err, ctx->sync_flags + 2, sizeof(err), hipMemcpyDeviceToHost, ctx->stream));
    TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (err) {
      // reported once: the word is cleared (behind everything enqueued so far on both streams), so a neighbour that was late once
      // does not poison the context
      TMHIP_CHECK(hipStreamSynchronize(ctx->comm_stream));
      TMHIP_CHECK(hipMemsetAsync(ctx->sync_flags + 2, 0, sizeof(err), ctx->stream));
      TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
      TMHIP_FAIL("the wait for the neighbours' faces gave up after %.3f s (option flag_timeout_ms): results since the last check are invalid",
                 (double)ctx->flag_timeout_ticks * 1.0e-8);
    }
  }
  if (ctx->shm) return tmhip_shm_failed(ctx);
  if (!ctx->comm_ready) return 0;
  ncclResult_t st = ncclSuccess;
  TMHIP_NCCL_CHECK(ncclCommGetAsyncError(ctx->comm, &st));
  if (st == ncclSuccess && ctx->comm_red != ctx->comm) TMHIP_NCCL_CHECK(ncclCommGetAsyncError(ctx->comm_red, &st));
  if (st != ncclSuccess && st != ncclInProgress) TMHIP_FAIL("the halo communicator reports an asynchronous error (%s): results since the last check are invalid", ncclGetErrorString(st));
  return 0;
}

int tmhip_stage_reserve(tmhip_ctx *ctx, size_t bytes) {
  if (ctx->stage_bytes >= bytes) return 0;
  if (ctx->stage) { TMHIP_CHECK(hipFree(ctx->stage)); ctx->stage = nullptr; ctx->stage_bytes = 0; }
  TMHIP_CHECK(hipMalloc(&ctx->stage, bytes));
  ctx->stage_bytes = bytes;
  return 0;
}

int tmhip_field_alloc_prec(tmhip_ctx *ctx, int kind, int prec, tmhip_field **out) {
  tmhip_field *f = new (std::nothrow) tmhip_field();
  if (!f) TMHIP_FAIL("out of host memory");
  f->kind = kind; f->prec = prec; f->ns = ctx->ns; f->view = false; f->half[0] = f->half[1] = nullptr; f->d = nullptr; f->d32 = nullptr;
  const size_t elems = (size_t)12 * ctx->ns * (kind == TMHIP_FIELD_FULL ? 2 : 1);
  if (prec) {
    if (kind != TMHIP_FIELD_EO) TMHIP_FAIL("fp32 fields are one-parity fields");
    TMHIP_CHECK(hipMalloc((void **)&f->d32, elems * sizeof(v2f)));
    TMHIP_CHECK(hipMemsetAsync(f->d32, 0, elems * sizeof(v2f), ctx->stream));
    *out = f;
    return 0;
  }
  TMHIP_CHECK(hipMalloc((void **)&f->d, elems * sizeof(v2d)));
  TMHIP_CHECK(hipMemsetAsync(f->d, 0, elems * sizeof(v2d), ctx->stream));
  if (kind == TMHIP_FIELD_FULL) {
    for (int p = 0; p < 2; p++) {
      tmhip_field *h = new tmhip_field();
      h->kind = TMHIP_FIELD_EO; h->prec = 0; h->d32 = nullptr; h->ns = ctx->ns; h->view = true; h->half[0] = h->half[1] = nullptr;
      h->d = f->d + (size_t)p * 12 * ctx->ns;
      f->half[p] = h;
    }
  }
  *out = f;
  return 0;
}
static int field_alloc_impl(tmhip_ctx *ctx, int kind, tmhip_field **out) { return tmhip_field_alloc_prec(ctx, kind, 0, out); }

extern "C" {

const char *tmhip_version(void) { return "tmlqcd_hip 0.1 (gfx950)"; }

int tmhip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int tmhip_create(const tmhip_geom *geom, int device, tmhip_ctx **out) {
  if (!geom || !out) TMHIP_FAIL("tmhip_create: null argument");
  const tmhip_geom g = *geom;
  // same constraints as the reference's e/o build (mpi_init.c:784-799: LZ even; e/o needs even extents)
  if (g.T < 2 || g.LX < 2 || g.LY < 2 || g.LZ < 2 || (g.T & 1) || (g.LX & 1) || (g.LY & 1) || (g.LZ & 1))
    TMHIP_FAIL("tmhip_create: local extents must be even and >= 2 (got %d %d %d %d)", g.T, g.LX, g.LY, g.LZ);
  if (g.nproc_t < 1 || g.proc_t < 0 || g.proc_t >= g.nproc_t) TMHIP_FAIL("tmhip_create: bad T decomposition");
  if ((double)g.T * g.LX * g.LY * g.LZ > 1.0e9) TMHIP_FAIL("tmhip_create: local volume too large for 32-bit site indices");
  TMHIP_CHECK(hipSetDevice(device));
  tmhip_ctx *ctx = new (std::nothrow) tmhip_ctx();
  if (!ctx) TMHIP_FAIL("out of host memory");
  memset(ctx, 0, sizeof(*ctx));
  ctx->g = g; ctx->device = device;
  ctx->V = g.T * g.LX * g.LY * g.LZ; ctx->Vh = ctx->V / 2; ctx->face = g.LX * g.LY * g.LZ / 2;
  ctx->ns = (ctx->Vh + 63) / 64 * 64; ctx->gs = ctx->ns;
  ctx->VPR = ctx->V + (g.nproc_t > 1 ? 2 * g.LX * g.LY * g.LZ : 0);
  ctx->opt_block = 0; ctx->opt_xcd = 2; ctx->opt_minw = 0; ctx->opt_occ = 3; ctx->opt_cg_sync = 0;
  ctx->opt_cg_batch = 4; ctx->opt_cg_fused_dot = 2; ctx->opt_cg_self = 1; ctx->opt_comm_split = 1; ctx->opt_split_sync = 0; ctx->opt_prepack = 1; ctx->opt_direct_form = -1; ctx->opt_direct_order = 3; ctx->opt_direct_sums = 1;
  {
    // bound of the device-side waits for the neighbours' faces: TMLQCD_HIP_FLAG_TIMEOUT_S in the environment (0 = none), default 120 s
    double sec = 120.0;
    const char *e = getenv("TMLQCD_HIP_FLAG_TIMEOUT_S");
    if (e && *e) { char *end = nullptr; const double v = strtod(e, &end); if (end != e && v >= 0.0) sec = v; }
    ctx->flag_timeout_ticks = (unsigned long long)(sec * 1.0e8);
  }
  ctx->opt_stg = 1; ctx->opt_stg32 = 0; ctx->opt_hopsplit = -1; ctx->opt_occ32 = 0; ctx->opt_recon = 0; ctx->opt_swall_order = 2; ctx->opt_swterm_order = 1; ctx->opt_gauge_cache = -1;
  ctx->gauge_recon_dev = -1.0;
  TMHIP_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  {  // boundary pipeline (pack, exchange, boundary kernels) must not queue behind the interior kernel's blocks
    int lo = 0, hi = 0;
    TMHIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    TMHIP_CHECK(hipStreamCreateWithPriority(&ctx->comm_stream, hipStreamNonBlocking, hi));
  }
  TMHIP_CHECK(hipEventCreateWithFlags(&ctx->ev_pack, hipEventDisableTiming));
  TMHIP_CHECK(hipEventCreateWithFlags(&ctx->ev_comm, hipEventDisableTiming));
  for (int i = 0; i < 16; i++) TMHIP_CHECK(hipEventCreate(&ctx->ev_slots[i]));
  TMHIP_CHECK(hipMalloc((void **)&ctx->gauge, (size_t)2 * 72 * ctx->gs * sizeof(v2d)));
  ctx->max_partials = 12 * ((ctx->ns + 1023) / 1024 + 1);
  {
    // fused stencil+reduction: one per wave of the (padded) grid, plus -- on the split path -- one per wave of the exterior kernel
    int need = 4 * ((ctx->ns + 255) / 256 + 8 + 7 * ctx->g.T) + 4 * 128;
    if (ctx->ns <= 262144 && need < 2 * (ctx->ns / 16 + 64)) need = 2 * (ctx->ns / 16 + 64);   // hop-split kernel on small lattices: four partials per 64 sites, twice (the self-summing CG iteration keeps two sets)
    if (ctx->max_partials < need) ctx->max_partials = need;
  }
  TMHIP_CHECK(hipMalloc((void **)&ctx->partials, ctx->max_partials * sizeof(double)));
  TMHIP_CHECK(hipMalloc((void **)&ctx->result_dev, 4 * sizeof(double)));
  TMHIP_CHECK(hipHostMalloc((void **)&ctx->result_host, 4 * sizeof(double)));
  const size_t fb = (size_t)6 * ctx->face * sizeof(v2d);
  // [send_dn | send_up] and [recv_up | recv_dn]: what goes down arrives at the lower neighbour as its "up" face and vice versa, so
  // the periodic wrap onto ourselves (loopback) is ONE copy of both faces
  TMHIP_CHECK(hipMalloc((void **)&ctx->send_dn, 2 * fb));
  TMHIP_CHECK(hipMalloc((void **)&ctx->recv_up, 2 * fb));
  ctx->send_up = ctx->send_dn + (size_t)6 * ctx->face;
  ctx->recv_dn = ctx->recv_up + (size_t)6 * ctx->face;
  TMHIP_CHECK(hipMalloc((void **)&ctx->sync_flags, 64));
  TMHIP_CHECK(hipMemsetAsync(ctx->sync_flags, 0, 64, ctx->stream));
  TMHIP_CHECK(hipMemsetAsync(ctx->recv_up, 0, 2 * fb, ctx->stream));
  for (int i = 0; i < 3; i++) {
    if (field_alloc_impl(ctx, TMHIP_FIELD_EO, &ctx->scratch[i])) return 1;
    if (field_alloc_impl(ctx, TMHIP_FIELD_EO, &ctx->sf[i])) return 1;
  }
  const double th[4] = {0, 0, 0, 0};
  tmhip_set_boundary(ctx, 0.125, th);
  ctx->mu = 0.0;
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  // TMLQCD_HIP_OPTIONS="gauge_recon=12,prepack=0": tmhip_set_option for executables that are linked against the drop-in unmodified
  if (const char *e = getenv("TMLQCD_HIP_OPTIONS")) {
    char buf[512];
    snprintf(buf, sizeof(buf), "%s", e);
    for (char *save = nullptr, *tok = strtok_r(buf, ",; ", &save); tok; tok = strtok_r(nullptr, ",; ", &save)) {
      char *eq = strchr(tok, '=');
      char *end = nullptr;
      const long v = eq ? strtol(eq + 1, &end, 10) : 0;
      if (!eq || end == eq + 1 || *end) { tmhip_destroy(ctx); TMHIP_FAIL("TMLQCD_HIP_OPTIONS: '%s' is not of the form name=integer", tok); }
      *eq = 0;
      if (tmhip_set_option(ctx, tok, (int)v)) { tmhip_destroy(ctx); return 1; }
    }
  }
  *out = ctx;
  return 0;
}

void tmhip_destroy(tmhip_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  for (int i = 0; i < 3; i++) { tmhip_field_free(ctx, ctx->scratch[i]); tmhip_field_free(ctx, ctx->sf[i]); }
  tmhip_field_free(ctx, ctx->sf_extra);
  for (int i = 0; i < 2; i++) tmhip_field_free(ctx, ctx->scratch32[i]);
  for (int i = 0; i < 4; i++) tmhip_field_free(ctx, ctx->sf32[i]);
  if (ctx->gauge32) (void)hipFree(ctx->gauge32);
  if (ctx->sw) (void)hipFree(ctx->sw);
  if (ctx->sw_inv) (void)hipFree(ctx->sw_inv);
  if (ctx->sw32) (void)hipFree(ctx->sw32);
  if (ctx->sw_inv32) (void)hipFree(ctx->sw_inv32);
  if (ctx->sw_fail) (void)hipFree(ctx->sw_fail);
  if (ctx->swpm) (void)hipFree(ctx->swpm);
  if (ctx->gauge_raw) (void)hipFree(ctx->gauge_raw);
  if (ctx->deriv) (void)hipFree(ctx->deriv);
  if (ctx->momenta) (void)hipFree(ctx->momenta);
  if (ctx->force_send) (void)hipFree(ctx->force_send);
  if (ctx->force_recv) (void)hipFree(ctx->force_recv);
  if (ctx->sw_ins) (void)hipFree(ctx->sw_ins);
  if (ctx->io_sums) (void)hipFree(ctx->io_sums);
  if (ctx->swpm_halo_send) (void)hipFree(ctx->swpm_halo_send);
  if (ctx->swpm_halo_recv) (void)hipFree(ctx->swpm_halo_recv);
  tmhip_direct_destroy(ctx);
  if (ctx->shm) tmhip_shm_destroy(ctx);
  else if (ctx->comm_ready) { if (ctx->comm_red != ctx->comm) ncclCommDestroy(ctx->comm_red); ncclCommDestroy(ctx->comm); }
  (void)hipFree(ctx->gauge); (void)hipFree(ctx->partials); (void)hipFree(ctx->result_dev);
  (void)hipHostFree(ctx->result_host);
  (void)hipFree(ctx->sync_flags);
  (void)hipFree(ctx->send_dn); (void)hipFree(ctx->recv_up);
  if (ctx->stage) (void)hipFree(ctx->stage);
  if (ctx->cg_state) (void)hipFree(ctx->cg_state);
  if (ctx->cg_hist) (void)hipFree(ctx->cg_hist);
  (void)hipEventDestroy(ctx->ev_pack); (void)hipEventDestroy(ctx->ev_comm);
  for (int i = 0; i < 16; i++) (void)hipEventDestroy(ctx->ev_slots[i]);
  (void)hipStreamDestroy(ctx->stream); (void)hipStreamDestroy(ctx->comm_stream);
  delete ctx;
}

int tmhip_sync(tmhip_ctx *ctx) {
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  TMHIP_CHECK(hipStreamSynchronize(ctx->comm_stream));
  return tmhip_check_async_error(ctx);
}

/* boundary.c:40-55 */
int tmhip_set_boundary(tmhip_ctx *ctx, double kappa, const double theta[4]) {
  const double PI_ = 3.14159265358979;  // boundary.c:36 (the reference's own truncated pi)
  const int ext[4] = {ctx->g.T * ctx->g.nproc_t, ctx->g.LX, ctx->g.LY, ctx->g.LZ};
  ctx->kappa = kappa;
  for (int m = 0; m < 4; m++) {
    ctx->theta[m] = theta[m];
    const double x = theta[m] * PI_ / ext[m];
    ctx->ka[m][0] = kappa * cos(x);
    ctx->ka[m][1] = kappa * sin(x);
  }
  return 0;
}

int tmhip_set_ka(tmhip_ctx *ctx, const double ka[8]) {
  for (int m = 0; m < 4; m++) { ctx->ka[m][0] = ka[2 * m]; ctx->ka[m][1] = ka[2 * m + 1]; }
  return 0;
}

int tmhip_set_mu(tmhip_ctx *ctx, double mu) { ctx->mu = mu; return 0; }
int tmhip_set_mu3(tmhip_ctx *ctx, double mu3) { ctx->mu3 = mu3; return 0; }

int tmhip_gauge_su3_deviation(tmhip_ctx *ctx, double *maxdev) {
  if (!ctx->gauge_set) TMHIP_FAIL("tmhip_gauge_su3_deviation called before tmhip_set_gauge");
  const int saved = ctx->opt_recon;
  ctx->opt_recon = 0;                       // measure only, do not toggle the option
  const int rc = tmhip_check_gauge_recon(ctx);
  ctx->opt_recon = saved;
  if (rc) return 1;
  *maxdev = ctx->gauge_recon_dev;
  return 0;
}

int tmhip_set_option(tmhip_ctx *ctx, const char *name, int value) {
  if (!strcmp(name, "block")) { if (value != 0 && value != 64 && value != 256) TMHIP_FAIL("block must be 0 (automatic), 64 or 256"); ctx->opt_block = value; }
  else if (!strcmp(name, "minw")) { if (value < 0 || value > 8) TMHIP_FAIL("minw must be in [0, 8] waves per SIMD"); ctx->opt_minw = value; }
  else if (!strcmp(name, "occ")) { if (value < 0 || value > 8) TMHIP_FAIL("occ must be in [0, 8] waves per SIMD (0 = no cap)"); ctx->opt_occ = value; }
  else if (!strcmp(name, "xcd")) { if (value < 0 || value > 4) TMHIP_FAIL("xcd must be 0 (none), 1 (chunk), 2 (automatic), 3 (slab) or 4 (tile)"); ctx->opt_xcd = value; }
  else if (!strcmp(name, "tgrp")) { if (value < 0 || value > ctx->g.T) TMHIP_FAIL("tgrp must be in [0, T]"); ctx->opt_tgrp = value; }
  else if (!strcmp(name, "split_sync")) { if (value < 0 || value > 1) TMHIP_FAIL("split_sync must be 0 (the exterior kernel and the pack kernel wait for a word of the other stream) or 1 (the streams are ordered by HIP events: no wait on the device)"); ctx->opt_split_sync = value; }
  else if (!strcmp(name, "flag_timeout_ms")) { if (value < 0) TMHIP_FAIL("flag_timeout_ms must be >= 0 (0 = wait without bound)"); ctx->flag_timeout_ticks = (unsigned long long)value * 100000ull; }
  else if (!strcmp(name, "direct_form")) { if (value < -1 || value > 1) TMHIP_FAIL("direct_form must be -1 (automatic), 0 (stencil + exterior kernel) or 1 (one kernel per stencil whenever the shape allows)"); ctx->opt_direct_form = value; }
  else if (!strcmp(name, "direct_sums")) { if (ctx->direct.on) TMHIP_FAIL("direct_sums must be set before tmhip_comm_init_ipc"); ctx->opt_direct_sums = value != 0; }
  else if (!strcmp(name, "direct_order")) { if (value < 0 || value > 3) TMHIP_FAIL("direct_order: bit 0 / bit 1 = boundary time-slices first for a stencil whose faces are packed now / were pushed ahead"); ctx->opt_direct_order = value; }
  else if (!strcmp(name, "prepack")) { ctx->opt_prepack = value != 0; ctx->prepacked = nullptr; }
  else if (!strcmp(name, "comm_split")) { if (ctx->comm_ready) TMHIP_FAIL("comm_split must be set before the communicator is created"); ctx->opt_comm_split = value != 0; }
  else if (!strcmp(name, "cg_fused_dot")) ctx->opt_cg_fused_dot = value;
  else if (!strcmp(name, "gauge_cache")) { if (value < -1 || value > 1) TMHIP_FAIL("gauge_cache must be -1 (automatic), 0 or 1"); ctx->opt_gauge_cache = value; }
  else if (!strcmp(name, "swall_order")) { if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8) TMHIP_FAIL("swall_order must be 0 (chunk per XCD), 1 (slab order) or 2 (tile order; 4 / 8: tiles of that many x-planes)"); ctx->opt_swall_order = value; }
  else if (!strcmp(name, "swterm_order")) { if (value < 0 || value > 1) TMHIP_FAIL("swterm_order must be 0 (chunk per XCD) or 1 (tiles through all time-slices, default)"); ctx->opt_swterm_order = value; }
  else if (!strcmp(name, "occ32")) { if (value < 0 || value > 8) TMHIP_FAIL("occ32 must be in [0, 8]"); ctx->opt_occ32 = value; }
  else if (!strcmp(name, "gauge_recon")) {
    if (value != 12 && value != 18 && value != 0) TMHIP_FAIL("gauge_recon must be 12 or 18");
    ctx->opt_recon = value == 12 ? 12 : 0;
    if (ctx->opt_recon == 12) return tmhip_check_gauge_recon(ctx);
  }
  else if (!strcmp(name, "hopsplit")) { if (value < -1 || value > 1) TMHIP_FAIL("hopsplit must be -1 (automatic), 0 or 1"); ctx->opt_hopsplit = value; }
  else if (!strcmp(name, "lds32")) { if (value < 0 || value > 1) TMHIP_FAIL("lds32 must be 0 or 1"); ctx->opt_stg32 = value; }
  else if (!strcmp(name, "lds")) { if (value < 0 || value > 1) TMHIP_FAIL("lds must be 0 (gather kernel) or 1 (per-wave LDS staging of the own-site spinors)"); ctx->opt_stg = value; }
  else if (!strcmp(name, "cg_self")) ctx->opt_cg_self = value != 0;
  else if (!strcmp(name, "cg_sync")) ctx->opt_cg_sync = value;
  else if (!strcmp(name, "cg_batch")) ctx->opt_cg_batch = value > 0 ? value : 1;
  else TMHIP_FAIL("unknown option %s", name);
  return 0;
}

int tmhip_set_gauge(tmhip_ctx *ctx, const void *host) {
  if (!host) TMHIP_FAIL("tmhip_set_gauge: null gauge field");
  TMHIP_CHECK(hipSetDevice(ctx->device));
  const size_t bytes = (size_t)ctx->VPR * 4 * 9 * sizeof(v2d);
  // The lexicographic links stay on the device (604 MB at 32^4 of 288 GB): the clover term and force walk them (clover.hip) and
  // the molecular-dynamics update works on them in place (md_update.hip), so a trajectory needs this upload once.
  if (!ctx->gauge_raw) TMHIP_CHECK(hipMalloc((void **)&ctx->gauge_raw, bytes));
  TMHIP_CHECK(hipMemcpyAsync(ctx->gauge_raw, host, bytes, hipMemcpyHostToDevice, ctx->stream));
  ctx->gauge_raw_valid = true;
  if (tmhip_resort_gauge(ctx)) return 1;
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (ctx->opt_recon == 12) return tmhip_check_gauge_recon(ctx);
  return 0;
}

/* "gauge_recon" = 12 is only exact for SU(3) links: measure how far the resident links are from it and drop back to the
 * full 18-real read (with a message) when they are not unitary to rounding, e.g. smeared or deliberately non-SU(3) input. */
int tmhip_check_gauge_recon(tmhip_ctx *ctx) {
  if (!ctx->gauge_set) return 0;     // checked again by tmhip_set_gauge
  if (ctx->gauge_recon_dev < 0.0) {
    unsigned long long *d = (unsigned long long *)ctx->result_dev;
    TMHIP_CHECK(hipMemsetAsync(d, 0, sizeof(*d), ctx->stream));
    hipLaunchKernelGGL(gauge_recon_dev_kernel, dim3((ctx->Vh + 255) / 256, 16), dim3(256), 0, ctx->stream, (const v2d *)ctx->gauge, ctx->gs, ctx->Vh, d);
    TMHIP_CHECK(hipGetLastError());
    unsigned long long h = 0;
    TMHIP_CHECK(hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
    memcpy(&ctx->gauge_recon_dev, &h, sizeof(double));
  }
  if (ctx->opt_recon == 12 && ctx->gauge_recon_dev > 1.0e-13) {
    fprintf(stderr, "[tmlqcd_hip] gauge_recon=12 refused: links deviate from SU(3) by %.3e (> 1e-13); using the full 18-real read\n",
            ctx->gauge_recon_dev);
    ctx->opt_recon = 0;
  }
  return 0;
}

// ------------------------------------------------------------------ fields
int tmhip_field_alloc(tmhip_ctx *ctx, int kind, tmhip_field **out) {
  if (kind != TMHIP_FIELD_EO && kind != TMHIP_FIELD_FULL) TMHIP_FAIL("tmhip_field_alloc: bad kind %d", kind);
  TMHIP_CHECK(hipSetDevice(ctx->device));
  return field_alloc_impl(ctx, kind, out);
}

void tmhip_field_free(tmhip_ctx *ctx, tmhip_field *f) {
  if (!f || f->view) return;
  (void)ctx;
  if (f->d) (void)hipFree(f->d);
  if (f->d32) (void)hipFree(f->d32);
  if (f->half[0]) delete f->half[0];
  if (f->half[1]) delete f->half[1];
  delete f;
}

tmhip_field *tmhip_field_even(tmhip_field *full) { return (full && full->kind == TMHIP_FIELD_FULL) ? full->half[0] : nullptr; }
tmhip_field *tmhip_field_odd(tmhip_field *full) { return (full && full->kind == TMHIP_FIELD_FULL) ? full->half[1] : nullptr; }

int tmhip_field_zero(tmhip_ctx *ctx, tmhip_field *f) {
  const size_t elems = (size_t)12 * f->ns * (f->kind == TMHIP_FIELD_FULL ? 2 : 1);
  if (f->prec) TMHIP_CHECK(hipMemsetAsync(f->d32, 0, elems * sizeof(v2f), ctx->stream));
  else TMHIP_CHECK(hipMemsetAsync(f->d, 0, elems * sizeof(v2d), ctx->stream));
  return 0;
}

int tmhip_field_upload(tmhip_ctx *ctx, tmhip_field *f, const void *host, int nsites) {
  if (!f || !host) TMHIP_FAIL("tmhip_field_upload: null argument");
  const int maxn = f->kind == TMHIP_FIELD_FULL ? ctx->V : ctx->Vh;
  if (nsites <= 0 || nsites > maxn) TMHIP_FAIL("tmhip_field_upload: nsites %d out of range (max %d)", nsites, maxn);
  if (f->kind == TMHIP_FIELD_FULL && nsites != ctx->V) TMHIP_FAIL("tmhip_field_upload: FULL fields take exactly V sites");
  const size_t bytes = (size_t)nsites * 12 * sizeof(v2d);
  if (tmhip_stage_reserve(ctx, bytes)) return 1;
  TMHIP_CHECK(hipMemcpyAsync(ctx->stage, host, bytes, hipMemcpyHostToDevice, ctx->stream));
  const int nb = (int)((12L * nsites + 255) / 256);
  if (f->kind == TMHIP_FIELD_EO)
    hipLaunchKernelGGL(aos_to_soa_kernel, dim3(nb), dim3(256), 0, ctx->stream, (const v2d *)ctx->stage, f->d, f->ns, nsites);
  else
    hipLaunchKernelGGL(lexic_eo_kernel<true>, dim3(nb), dim3(256), 0, ctx->stream, (v2d *)ctx->stage, f->d, f->ns, ctx->V,
                       ctx->g.LX, ctx->g.LY, ctx->g.LZ, ctx->g.proc_t * ctx->g.T);
  TMHIP_CHECK(hipGetLastError());
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));  // staging buffer is reused by the next call
  return 0;
}

int tmhip_field_download(tmhip_ctx *ctx, tmhip_field *f, void *host, int nsites) {
  if (!f || !host) TMHIP_FAIL("tmhip_field_download: null argument");
  const int maxn = f->kind == TMHIP_FIELD_FULL ? ctx->V : ctx->Vh;
  if (nsites <= 0 || nsites > maxn) TMHIP_FAIL("tmhip_field_download: nsites %d out of range (max %d)", nsites, maxn);
  if (f->kind == TMHIP_FIELD_FULL && nsites != ctx->V) TMHIP_FAIL("tmhip_field_download: FULL fields take exactly V sites");
  const size_t bytes = (size_t)nsites * 12 * sizeof(v2d);
  if (tmhip_stage_reserve(ctx, bytes)) return 1;
  const int nb = (int)((12L * nsites + 255) / 256);
  if (f->kind == TMHIP_FIELD_EO)
    hipLaunchKernelGGL(soa_to_aos_kernel, dim3(nb), dim3(256), 0, ctx->stream, (const v2d *)f->d, (v2d *)ctx->stage, f->ns, nsites);
  else
    hipLaunchKernelGGL(lexic_eo_kernel<false>, dim3(nb), dim3(256), 0, ctx->stream, (v2d *)ctx->stage, f->d, f->ns, ctx->V,
                       ctx->g.LX, ctx->g.LY, ctx->g.LZ, ctx->g.proc_t * ctx->g.T);
  TMHIP_CHECK(hipGetLastError());
  TMHIP_CHECK(hipMemcpyAsync(host, ctx->stage, bytes, hipMemcpyDeviceToHost, ctx->stream));
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  return tmhip_check_async_error(ctx);
}

/* page-locked host memory for callers that must keep the runtime away from their own pages (the drop-in's lazy mode: user arrays
 * whose protection changes must never be registered with the driver) */
int tmhip_pinned_alloc(unsigned long bytes, void **out) {
  if (!out) TMHIP_FAIL("tmhip_pinned_alloc: null argument");
  TMHIP_CHECK(hipHostMalloc(out, bytes, hipHostMallocDefault));
  return 0;
}
int tmhip_pinned_free(void *p) {
  if (p) TMHIP_CHECK(hipHostFree(p));
  return 0;
}
// sites [first, first + count) of a one-parity field in the host's AoS layout (the drop-in's page-wise lazy synchronisation)
__global__ __launch_bounds__(256) void soa_to_aos_range_kernel(const v2d *__restrict__ soa, v2d *__restrict__ aos, int ns, int first, int n) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  if (tid >= 12 * n) return;
  const int site = tid / 12, c = tid % 12;
  aos[tid] = soa[(size_t)c * ns + first + site];
}
/* Writes straight into PAGE-LOCKED host memory (tmhip_pinned_alloc): no staging buffer of the context is involved, so this is the
 * one transfer that may run on a thread other than the one driving the context -- the drop-in's fault handler serves host threads
 * while the master thread is inside another call.  EO fields: sites [first, first + count); FULL fields: all of it (first = 0). */
int tmhip_field_download_range(tmhip_ctx *ctx, tmhip_field *f, void *host, int first, int count) {
  if (!f || !host || f->prec != 0) TMHIP_FAIL("tmhip_field_download_range: needs an fp64 field and a page-locked host buffer");
  const int maxn = f->kind == TMHIP_FIELD_FULL ? ctx->V : ctx->Vh;
  if (first < 0 || count <= 0 || first + count > maxn) TMHIP_FAIL("tmhip_field_download_range: sites [%d, %d) outside [0, %d)", first, first + count, maxn);
  if (f->kind == TMHIP_FIELD_FULL && (first != 0 || count != ctx->V)) TMHIP_FAIL("tmhip_field_download_range: FULL fields are taken whole");
  TMHIP_CHECK(hipSetDevice(ctx->device));
  void *dst = nullptr;
  if (hipHostGetDevicePointer(&dst, host, 0) != hipSuccess || !dst) { (void)hipGetLastError(); TMHIP_FAIL("tmhip_field_download_range: the host buffer is not page-locked memory of this process"); }
  const int nb = (int)((12L * count + 255) / 256);
  if (f->kind == TMHIP_FIELD_EO)
    hipLaunchKernelGGL(soa_to_aos_range_kernel, dim3(nb), dim3(256), 0, ctx->stream, (const v2d *)f->d, (v2d *)dst, f->ns, first, count);
  else
    hipLaunchKernelGGL(lexic_eo_kernel<false>, dim3(nb), dim3(256), 0, ctx->stream, (v2d *)dst, f->d, f->ns, ctx->V,
                       ctx->g.LX, ctx->g.LY, ctx->g.LZ, ctx->g.proc_t * ctx->g.T);
  TMHIP_CHECK(hipGetLastError());
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  return 0;
}

// ------------------------------------------------------------------ stencil entry points
static int need_eo(const tmhip_field *f, const char *who) {
  if (!f || f->kind != TMHIP_FIELD_EO || f->prec != 0) { fprintf(stderr, "[tmlqcd_hip] %s: needs a one-parity (EO) field\n", who); return 1; }
  return 0;
}

int tmhip_hopping_matrix(tmhip_ctx *ctx, int ieo, tmhip_field *l, tmhip_field *k) {
  if (need_eo(l, "Hopping_Matrix") || need_eo(k, "Hopping_Matrix")) return 1;
  return tmhip_launch_hopping(ctx, ieo, l->d, k->d, nullptr, EPI_STORE, 0, 0, true);
}
int tmhip_hopping_matrix_nocom(tmhip_ctx *ctx, int ieo, tmhip_field *l, tmhip_field *k) {
  if (need_eo(l, "Hopping_Matrix_nocom") || need_eo(k, "Hopping_Matrix_nocom")) return 1;
  return tmhip_launch_hopping(ctx, ieo, l->d, k->d, nullptr, EPI_STORE, 0, 0, false);
}
int tmhip_tm_times_hopping_matrix(tmhip_ctx *ctx, int ieo, tmhip_field *l, tmhip_field *k, double cre, double cim) {
  if (need_eo(l, "tm_times_Hopping_Matrix") || need_eo(k, "tm_times_Hopping_Matrix")) return 1;
  return tmhip_launch_hopping(ctx, ieo, l->d, k->d, nullptr, EPI_TM_TIMES, cre, cim, true);
}
int tmhip_tm_sub_hopping_matrix(tmhip_ctx *ctx, int ieo, tmhip_field *l, tmhip_field *p, tmhip_field *k, double cre, double cim) {
  if (need_eo(l, "tm_sub_Hopping_Matrix") || need_eo(p, "tm_sub_Hopping_Matrix") || need_eo(k, "tm_sub_Hopping_Matrix")) return 1;
  return tmhip_launch_hopping(ctx, ieo, l->d, k->d, p->d, EPI_TM_SUB_G5, cre, cim, true);
}

/* D_psi_body.c:266-375: P = (1 + i mu g5) Q + sum phase_mu hop_mu(Q), phase_mu = -ka_mu (boundary.c:51-54)
 * => per parity:  P_p = (1 + i mu g5) Q_p - H_{p,1-p} Q_{1-p}  = the tm_sub epilogue without g5. */
int tmhip_D_psi(tmhip_ctx *ctx, tmhip_field *P, tmhip_field *Q) {
  if (!P || !Q || P->kind != TMHIP_FIELD_FULL || Q->kind != TMHIP_FIELD_FULL) TMHIP_FAIL("D_psi needs FULL fields");
  if (P == Q || P->d == Q->d) {  // D_psi_body.c:267-272
    fprintf(stderr, "Error in D_psi (operator.c):\nArguments must be different spinor fields\nProgram aborted\n");
    return 1;
  }
  for (int par = 0; par < 2; par++)
    if (tmhip_launch_hopping(ctx, par, P->half[par]->d, Q->half[1 - par]->d, Q->half[par]->d, EPI_TM_SUB, 1.0, ctx->mu, true))
      return 1;
  return 0;
}

// ------------------------------------------------------------------ e/o compositions (tm_operators.c)
/* tm_operators.c:508-526 */
int tmhip_H_eo_tm_inv_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k, int ieo, double _sign) {
  if (need_eo(l, "H_eo_tm_inv_psi") || need_eo(k, "H_eo_tm_inv_psi")) return 1;
  const double nrm = 1. / (1. + ctx->mu * ctx->mu), sign = _sign < 0. ? 1. : -1.;
  return tmhip_tm_times_hopping_matrix(ctx, ieo, l, k, nrm, sign * nrm * ctx->mu);
}
// raw-pointer forms
static int hop_tm_inv(tmhip_ctx *ctx, v2d *l, const v2d *k, int ieo, double _sign, int flags) {   /* tm_operators.c:508-526 */
  const double nrm = 1. / (1. + ctx->mu * ctx->mu), sign = _sign < 0. ? 1. : -1.;
  return tmhip_launch_hopping(ctx, ieo, l, k, nullptr, EPI_TM_TIMES, nrm, sign * nrm * ctx->mu, flags);
}
static int hop_tm_sub_g5(tmhip_ctx *ctx, v2d *l, const v2d *p, const v2d *k, int ieo, double _sign, int flags) {   /* :528-546 */
  return tmhip_launch_hopping(ctx, ieo, l, k, p, EPI_TM_SUB_G5, 1., (_sign < 0. ? -1. : 1.) * ctx->mu, flags);
}
#define HOP_FIRST HOP_COMM
#define HOP_NEXT (HOP_COMM | HOP_CHAINED)
/* l may alias k for these (invert_eo.c:270 calls Qtm_minus_psi in place): the last stencil reads
 * k only through the element-wise epilogue `p`, never as a gathered neighbour field. */
/* tm_operators.c:172-177 */
int tmhip_Qtm_plus_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k) {
  if (need_eo(l, "Qtm_plus_psi") || need_eo(k, "Qtm_plus_psi")) return 1;
  return hop_tm_inv(ctx, ctx->scratch[1]->d, k->d, TMHIP_EO, +1., HOP_FIRST | HOP_FEED) ||
         hop_tm_sub_g5(ctx, l->d, k->d, ctx->scratch[1]->d, TMHIP_OE, +1., HOP_NEXT);
}
/* tm_operators.c:216-221 */
int tmhip_Qtm_minus_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k) {
  if (need_eo(l, "Qtm_minus_psi") || need_eo(k, "Qtm_minus_psi")) return 1;
  return hop_tm_inv(ctx, ctx->scratch[1]->d, k->d, TMHIP_EO, -1., HOP_FIRST | HOP_FEED) ||
         hop_tm_sub_g5(ctx, l->d, k->d, ctx->scratch[1]->d, TMHIP_OE, -1., HOP_NEXT);
}
/* tm_operators.c:245-250 */
int tmhip_Mtm_plus_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k) {
  if (need_eo(l, "Mtm_plus_psi") || need_eo(k, "Mtm_plus_psi")) return 1;
  if (tmhip_H_eo_tm_inv_psi(ctx, ctx->scratch[1], k, TMHIP_EO, +1.)) return 1;
  return tmhip_launch_hopping(ctx, TMHIP_OE, l->d, ctx->scratch[1]->d, k->d, EPI_TM_SUB, 1., ctx->mu, true);
}
/* tm_operators.c:289-294 */
int tmhip_Mtm_minus_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k) {
  if (need_eo(l, "Mtm_minus_psi") || need_eo(k, "Mtm_minus_psi")) return 1;
  if (tmhip_H_eo_tm_inv_psi(ctx, ctx->scratch[1], k, TMHIP_EO, -1.)) return 1;
  return tmhip_launch_hopping(ctx, TMHIP_OE, l->d, ctx->scratch[1]->d, k->d, EPI_TM_SUB, 1., -ctx->mu, true);
}
/* tm_operators.c:338-345 : 4 stencil launches, twists fused into the epilogues */
int tmhip_Qtm_pm_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k) {
  if (need_eo(l, "Qtm_pm_psi") || need_eo(k, "Qtm_pm_psi")) return 1;
  v2d *s0 = ctx->scratch[0]->d, *s1 = ctx->scratch[1]->d;
  return hop_tm_inv(ctx, s1, k->d, TMHIP_EO, -1., HOP_FIRST | HOP_FEED) || hop_tm_sub_g5(ctx, s0, k->d, s1, TMHIP_OE, -1., HOP_NEXT | HOP_FEED) ||
         hop_tm_inv(ctx, s1, s0, TMHIP_EO, +1., HOP_NEXT | HOP_FEED) || hop_tm_sub_g5(ctx, l->d, s0, s1, TMHIP_OE, +1., HOP_NEXT);
}
/* The "symmetric" e/o preconditioning family (tm_operators.c:186-192,223-229,259-265,296-302):
 *   X_sym = k - (1 +- i mu g5)^-1 H_oe (1 +- i mu g5)^-1 H_eo k.
 * Both inverse twists ride in the stencil epilogues (EPI_TM_TIMES); the subtraction is one streaming pass.
 * Used by the non-hermitian solvers of invert_eo.c:177-280 (bicgstab, gmres, gcr, cgs ...). */
static int sym_core(tmhip_ctx *ctx, tmhip_field *k, double sign) {
  return tmhip_H_eo_tm_inv_psi(ctx, ctx->scratch[1], k, TMHIP_EO, sign) ||
         tmhip_H_eo_tm_inv_psi(ctx, ctx->scratch[0], ctx->scratch[1], TMHIP_OE, sign);
}
/* tm_operators.c:186-192 */
int tmhip_Qtm_plus_sym_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k) {
  if (need_eo(l, "Qtm_plus_sym_psi") || need_eo(k, "Qtm_plus_sym_psi")) return 1;
  return sym_core(ctx, k, +1.) || tmhip_mul_one_sub_mul_gamma5(ctx, l, k, ctx->scratch[0]);
}
/* tm_operators.c:223-229 */
int tmhip_Qtm_minus_sym_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k) {
  if (need_eo(l, "Qtm_minus_sym_psi") || need_eo(k, "Qtm_minus_sym_psi")) return 1;
  return sym_core(ctx, k, -1.) || tmhip_mul_one_sub_mul_gamma5(ctx, l, k, ctx->scratch[0]);
}
/* tm_operators.c:259-265 */
int tmhip_Mtm_plus_sym_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k) {
  if (need_eo(l, "Mtm_plus_sym_psi") || need_eo(k, "Mtm_plus_sym_psi")) return 1;
  return sym_core(ctx, k, +1.) || tmhip_diff(ctx, l, k, ctx->scratch[0], ctx->Vh);
}
/* tm_operators.c:296-302 */
int tmhip_Mtm_minus_sym_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k) {
  if (need_eo(l, "Mtm_minus_sym_psi") || need_eo(k, "Mtm_minus_sym_psi")) return 1;
  return sym_core(ctx, k, -1.) || tmhip_diff(ctx, l, k, ctx->scratch[0], ctx->Vh);
}
/* tm_operators.c:312-322 : (Mtm_plus_sym)^dagger = 1 - g5 H_oe A_-^-1 H_eo A_-^-1 g5 ; l is used as work space first,
 * exactly like the reference, so l must not alias k. */
int tmhip_Mtm_plus_sym_dagg_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k) {
  if (need_eo(l, "Mtm_plus_sym_dagg_psi") || need_eo(k, "Mtm_plus_sym_dagg_psi")) return 1;
  if (l == k || l->d == k->d) TMHIP_FAIL("Mtm_plus_sym_dagg_psi: l must not alias k");
  return tmhip_gamma5(ctx, l, k, ctx->Vh) || tmhip_mul_one_pm_imu_inv(ctx, l, -1., ctx->Vh) ||
         tmhip_H_eo_tm_inv_psi(ctx, ctx->scratch[1], l, TMHIP_EO, -1.) ||
         tmhip_hopping_matrix(ctx, TMHIP_OE, ctx->scratch[0], ctx->scratch[1]) ||
         tmhip_gamma5(ctx, ctx->scratch[1], ctx->scratch[0], ctx->Vh) || tmhip_diff(ctx, l, k, ctx->scratch[1], ctx->Vh);
}
/* tm_operators.c:347-364.  The reference body overwrites its Q_- result: the second pair of stencils writes
 * l and DUM_MATRIX+1 only, then l is rebuilt from k and DUM_MATRIX, so what it returns is
 *   l = g5 ( k - A_+^-1 A_-^-1 H_oe A_-^-1 H_eo k ).
 * A drop-in has to return the same field, so that is what is computed here (2 stencils instead of 4). */
int tmhip_Qtm_pm_sym_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k) {
  if (need_eo(l, "Qtm_pm_sym_psi") || need_eo(k, "Qtm_pm_sym_psi")) return 1;
  return sym_core(ctx, k, -1.) || tmhip_mul_one_pm_imu_inv(ctx, ctx->scratch[0], +1., ctx->Vh) ||
         tmhip_mul_one_sub_mul_gamma5(ctx, l, k, ctx->scratch[0]);
}
/* tm_operators.c:117-128 :  X_new = (1 + i mu g5) X - H Y */
int tmhip_M_full(tmhip_ctx *ctx, tmhip_field *En, tmhip_field *On, tmhip_field *E, tmhip_field *O) {
  if (need_eo(En, "M_full") || need_eo(On, "M_full") || need_eo(E, "M_full") || need_eo(O, "M_full")) return 1;
  return tmhip_launch_hopping(ctx, TMHIP_EO, En->d, O->d, E->d, EPI_TM_SUB, 1., ctx->mu, true) ||
         tmhip_launch_hopping(ctx, TMHIP_OE, On->d, E->d, O->d, EPI_TM_SUB, 1., ctx->mu, true);
}

// ------------------------------------------------------------------ halo exchange
int tmhip_comm_get_unique_id(char id[TMHIP_UNIQUE_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) <= TMHIP_UNIQUE_ID_BYTES, "unique id does not fit");
  ncclUniqueId u;
  TMHIP_NCCL_CHECK(ncclGetUniqueId(&u));
  memset(id, 0, TMHIP_UNIQUE_ID_BYTES);
  memcpy(id, &u, sizeof(u));
  return 0;
}

// second communicator over the same ranks for everything issued on the main stream (collective: every rank calls it at the same point)
static int comm_make_red(tmhip_ctx *ctx, int rank) {
  ctx->comm_red = ctx->comm; ctx->comm_split = false;
  if (!ctx->opt_comm_split) return 0;
  const ncclResult_t rs = ncclCommSplit(ctx->comm, 0, rank, &ctx->comm_red, nullptr);
  if (rs != ncclSuccess) {
    // (an RCCL without ncclCommSplit: every rank fails here alike.)  One communicator is enough for correctness: a reduction is never
    // in flight together with a face exchange (hopping_split.inc launch_split), so this is a fallback, not an error.
    fprintf(stderr, "[tmlqcd_hip] ncclCommSplit failed (%s): reductions share the face communicator\n", ncclGetErrorString(rs));
    ctx->comm_red = ctx->comm;
    return 0;
  }
  ctx->comm_split = true;
  return 0;
}

// one thread that sleeps until the 100 MHz clock has advanced by `ticks` (bounded: at most 20 s, tmhip_comm_stream_delay_ms)
__global__ void delay_kernel(unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

int tmhip_comm_init(tmhip_ctx *ctx, const char id[TMHIP_UNIQUE_ID_BYTES]) {
  if (ctx->g.nproc_t < 2) return 0;
  TMHIP_CHECK(hipSetDevice(ctx->device));
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  TMHIP_NCCL_CHECK(ncclCommInitRank(&ctx->comm, ctx->g.nproc_t, u, ctx->g.proc_t));
  if (comm_make_red(ctx, ctx->g.proc_t)) return 1;
  ctx->comm_ready = true;
  return 0;
}

int tmhip_comm_count(tmhip_ctx *ctx, int *nranks_faces, int *nranks_reduce) {
  if (!ctx->comm_ready) { *nranks_faces = *nranks_reduce = 0; return 0; }
  if (ctx->shm) { *nranks_faces = *nranks_reduce = ctx->g.nproc_t; return 0; }
  TMHIP_NCCL_CHECK(ncclCommCount(ctx->comm, nranks_faces));
  TMHIP_NCCL_CHECK(ncclCommCount(ctx->comm_red, nranks_reduce));
  return 0;
}

/* 1: reductions run on their own communicator (ncclCommSplit), 0: they share the face communicator, -1: no communicator */
int tmhip_comm_is_split(tmhip_ctx *ctx) { return ctx->comm_ready ? (ctx->comm_split ? 1 : 0) : -1; }

/* Test hook: hold the comm stream back for `ms` milliseconds in front of whatever is enqueued on it next (the next halo exchange) --
 * a neighbour that arrives late, as seen from this rank.  Results do not change; only their time of arrival does. */
int tmhip_comm_stream_delay_ms(tmhip_ctx *ctx, int ms) {
  if (ms < 0 || ms > 20000) TMHIP_FAIL("tmhip_comm_stream_delay_ms: 0 .. 20000 ms");
  TMHIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(1), 0, ctx->comm_stream, (unsigned long long)ms * 100000ull);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}

int tmhip_comm_set_loopback(tmhip_ctx *ctx, int on) {
  if (ctx->g.nproc_t > 1) TMHIP_FAIL("loopback is a single-rank self-test");
  if (on < 0 || on > 3) TMHIP_FAIL("loopback: 0 off, 1 device-to-device copies, 2 one-rank RCCL communicator, 3 direct carrier onto oneself");
  ctx->loopback = on != 0;
  ctx->loopback_rccl = on == 2;
  ctx->prepacked = nullptr; ctx->direct.ahead_field = nullptr;
  if (on == 3) { if (tmhip_direct_init_self(ctx)) return 1; }
  else if (ctx->direct.on) {   // back to a carrier on the comm stream: everything pushed so far has been consumed or is abandoned
    TMHIP_CHECK(hipStreamSynchronize(ctx->stream)); TMHIP_CHECK(hipStreamSynchronize(ctx->comm_stream));
    ctx->direct.on = false;
  }
  if (on == 2 && !ctx->comm_ready) {  // one-rank RCCL communicator: faces travel through ncclSend/ncclRecv to self
    TMHIP_CHECK(hipSetDevice(ctx->device));
    ncclUniqueId u;
    TMHIP_NCCL_CHECK(ncclGetUniqueId(&u));
    TMHIP_NCCL_CHECK(ncclCommInitRank(&ctx->comm, 1, u, 0));
    if (comm_make_red(ctx, 0)) return 1;
    ctx->comm_ready = true;
  }
  return 0;
}

}  // extern "C"

// Runs on ctx->comm_stream.  Ring along T (mpi_init.c:391-394 g_nb_t_up/dn):
//   send_dn -> rank-1 (lands in its recv_up),  send_up -> rank+1 (lands in its recv_dn).
int tmhip_halo_exchange(tmhip_ctx *ctx) {
  const size_t n = (size_t)6 * ctx->face * 2;  // doubles per face
  if (ctx->g.nproc_t == 1 && !ctx->loopback_rccl) {  // periodic wrap onto ourselves (loopback self-test)
    TMHIP_CHECK(hipMemcpyAsync(ctx->recv_up, ctx->send_dn, 2 * n * sizeof(double), hipMemcpyDeviceToDevice, ctx->comm_stream));   // (both faces: buffers are contiguous)
    return 0;
  }
  if (!ctx->comm_ready) TMHIP_FAIL("nproc_t > 1 but tmhip_comm_init was not called");
  if (ctx->shm) return tmhip_shm_ring(ctx, ctx->comm_stream, ctx->send_dn, ctx->send_up, ctx->recv_up, ctx->recv_dn, n * sizeof(double));   // host-staged: copy out, one host function, copy in -- on the comm stream like the RCCL calls below
  const int np = ctx->g.nproc_t, up = (ctx->g.proc_t + 1) % np, dn = (ctx->g.proc_t + np - 1) % np;
  TMHIP_NCCL_CHECK(ncclGroupStart());
  TMHIP_NCCL_CHECK(ncclSend(ctx->send_dn, n, ncclDouble, dn, ctx->comm, ctx->comm_stream));
  TMHIP_NCCL_CHECK(ncclRecv(ctx->recv_up, n, ncclDouble, up, ctx->comm, ctx->comm_stream));
  TMHIP_NCCL_CHECK(ncclSend(ctx->send_up, n, ncclDouble, up, ctx->comm, ctx->comm_stream));
  TMHIP_NCCL_CHECK(ncclRecv(ctx->recv_dn, n, ncclDouble, dn, ctx->comm, ctx->comm_stream));
  TMHIP_NCCL_CHECK(ncclGroupEnd());
  return 0;
}

// ------------------------------------------------------------------ measurement
extern "C" {

int tmhip_event_record(tmhip_ctx *ctx, int slot) {
  if (slot < 0 || slot >= 16) TMHIP_FAIL("event slot out of range");
  TMHIP_CHECK(hipEventRecord(ctx->ev_slots[slot], ctx->stream));
  return 0;
}
int tmhip_event_elapsed_ms(tmhip_ctx *ctx, int a, int b, double *ms) {
  if (a < 0 || a >= 16 || b < 0 || b >= 16) TMHIP_FAIL("event slot out of range");
  TMHIP_CHECK(hipEventSynchronize(ctx->ev_slots[b]));
  float f = 0;
  TMHIP_CHECK(hipEventElapsedTime(&f, ctx->ev_slots[a], ctx->ev_slots[b]));
  *ms = f;
  return 0;
}

/* benchmark.c:291-300 */
int tmhip_bench_hopping(tmhip_ctx *ctx, tmhip_field *f0, tmhip_field *f1, tmhip_field *f2, int iters, double *ms_total) {
  if (need_eo(f0, "bench") || need_eo(f1, "bench") || need_eo(f2, "bench")) return 1;
  if (tmhip_event_record(ctx, 14)) return 1;
  for (int j = 0; j < iters; j++) {
    if (tmhip_launch_hopping(ctx, 0, f1->d, f0->d, nullptr, EPI_STORE, 0, 0, HOP_COMM | HOP_FEED)) return 1;   // benchmark.c:295-296 (f1 is gathered by the next call)
    if (tmhip_launch_hopping(ctx, 1, f2->d, f1->d, nullptr, EPI_STORE, 0, 0, HOP_COMM | HOP_CHAINED)) return 1;   // f1 is the previous stencil's output
  }
  if (tmhip_event_record(ctx, 15)) return 1;
  return tmhip_event_elapsed_ms(ctx, 14, 15, ms_total);
}

}  // extern "C"
