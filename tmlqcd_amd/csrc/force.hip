// Fermion force, hopping part: deriv_Sb (deriv_Sb.c:401-700), SURVEY §8f rank 3.
//
// The reference loops over the sites x of parity ieo and scatters two contributions per direction, to the links
// (x, mu) and (x - mu, mu) -- hence its "_nonlocal" (atomic) accumulate under OpenMP.  Every link of the lattice
// receives exactly ONE contribution per call, so here the loop is turned inside out: one thread per site y of
// EITHER parity owns its four forward links (y, mu) and gathers
//   parity(y) == ieo :  "+mu" term   phi = P+_mu g5 l(y),   psi = P+_mu k(y+mu),   t = phi (x) psi^dagger
//   parity(y) != ieo :  "-mu" term of x = y+mu:  psi = P-_mu k(y),  phi = P-_mu g5 l(y+mu),  t = psi (x) phi^dagger
// then  df(y,mu) += 2 factor trlambda( ka_mu U_mu(y) t^dagger ).  No atomics, coalesced SoA reads, and U t^dagger is
// formed as (U v) (x) u^dagger from two 3x3 mat-vecs instead of a 3x3 x 3x3 product.
//
// Device derivative field: double deriv[2 parity][4 mu][8][Vh]; tmhip_derivative_download converts to the host's
// su3adj df[VOLUME][4] (lexicographic).
#include "tmhip_internal.h"

struct ForceArgs {
  const v2d *own[2];      // field living on parity p: own[ieo] = l, own[1-ieo] = k
  const v2d *g;           // gauge [2][8][9][gs]
  double *deriv;          // [2][4][8][Vh]
  int ns, gs, Vh, T, LX, LY, LZ, ieo;
  int toff;               // proc_t * T: global parity offset of a T-split rank
  const v2d *halo;        // T-split: [24][face] = t=0 slices of the up-neighbour's l (components 0..11) and k (12..23)
  int face;
  double ka[4][2];
  double fac;             // 2 * factor
};

__device__ __forceinline__ v2d f_cmul(v2d a, v2d b) { return v2d{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ v2d f_cmulc(v2d a, v2d b) { return v2d{a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y}; }   // a conj(b)
__device__ __forceinline__ v2d f_itimes(v2d a) { return v2d{-a.y, a.x}; }

// two-component projection (1 +- gamma_mu) of a 4-spinor held as s[spin][colour]; hopping.h:578-672 conventions
__device__ __forceinline__ void f_project(const v2d (&s)[4][3], int mu, bool plus, v2d (&a)[3], v2d (&b)[3]) {
#pragma unroll
  for (int c = 0; c < 3; c++) {
    switch (mu) {
      case 0: a[c] = plus ? s[0][c] + s[2][c] : s[0][c] - s[2][c]; b[c] = plus ? s[1][c] + s[3][c] : s[1][c] - s[3][c]; break;
      case 1: a[c] = plus ? s[0][c] + f_itimes(s[3][c]) : s[0][c] - f_itimes(s[3][c]);
              b[c] = plus ? s[1][c] + f_itimes(s[2][c]) : s[1][c] - f_itimes(s[2][c]); break;
      case 2: a[c] = plus ? s[0][c] + s[3][c] : s[0][c] - s[3][c]; b[c] = plus ? s[1][c] - s[2][c] : s[1][c] + s[2][c]; break;
      default: a[c] = plus ? s[0][c] + f_itimes(s[2][c]) : s[0][c] - f_itimes(s[2][c]);
               b[c] = plus ? s[1][c] - f_itimes(s[3][c]) : s[1][c] + f_itimes(s[3][c]); break;
    }
  }
}

__device__ __forceinline__ void f_load(const v2d *__restrict__ f, int ns, int idx, bool g5, v2d (&s)[4][3]) {
#pragma unroll
  for (int sp = 0; sp < 4; sp++)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      v2d v = f[(size_t)(3 * sp + c) * ns + idx];
      if (g5 && sp >= 2) v = v2d{-v.x, -v.y};
      s[sp][c] = v;
    }
}

__global__ __launch_bounds__(128) void deriv_Sb_kernel(ForceArgs a) {
  const int i = blockIdx.x * 128 + threadIdx.x;
  if (i >= a.Vh) return;
  const int par = blockIdx.y;
  const bool plus = par == a.ieo;         // this site carries the left vector l
  const int LZh = a.LZ / 2;
  const int kz = i % LZh;
  int r = i / LZh;
  const int y = r % a.LY;
  r /= a.LY;
  const int x = r % a.LX, t = r / a.LX;
  const int o = (t + x + y + par + a.toff) & 1;
  const int z = 2 * kz + o;
  const int row = (t * a.LX + x) * a.LY + y;   // lexic = row * LZ + z
  int up[4];
  up[0] = ((((t + 1) % a.T) * a.LX + x) * a.LY + y) * LZh + kz;      // same (x,y,z): z parity flips with the site parity
  up[1] = ((t * a.LX + (x + 1) % a.LX) * a.LY + y) * LZh + kz;
  up[2] = ((t * a.LX + x) * a.LY + (y + 1) % a.LY) * LZh + kz;
  up[3] = (row * a.LZ + (z + 1) % a.LZ) >> 1;
  v2d own[4][3];
  f_load(a.own[par], a.ns, i, plus, own);
  const v2d *gp = a.g + (size_t)par * 72 * a.gs + i;
  double *dp = a.deriv + (size_t)par * 32 * a.Vh + i;
#pragma unroll
  for (int mu = 0; mu < 4; mu++) {
    v2d nb[4][3];
    if (mu == 0 && a.halo && t == a.T - 1)   // +t neighbour lives on the next rank: its t=0 slice was exchanged (xchange_2fields, deriv_Sb.c:102)
      f_load(a.halo + (size_t)(plus ? 12 : 0) * a.face, a.face, i - (a.T - 1) * a.face, !plus, nb);
    else
      f_load(a.own[1 - par], a.ns, up[mu], !plus, nb);
    v2d phia[3], phib[3], psia[3], psib[3];
    // phi* <- projections of g5 l, psi* <- projections of k
    if (plus) { f_project(own, mu, true, phia, phib); f_project(nb, mu, true, psia, psib); }
    else      { f_project(own, mu, false, psia, psib); f_project(nb, mu, false, phia, phib); }
    // t = u (x) v^dagger + w (x) z^dagger with (u,v,w,z) = (phia,psia,phib,psib) for "+", (psia,phia,psib,phib) for "-"
    // U t^dagger = (U v) (x) u^dagger + (U z) (x) w^dagger
    v2d U[9];
#pragma unroll
    for (int e = 0; e < 9; e++) U[e] = gp[(size_t)((2 * mu) * 9 + e) * a.gs];
    v2d Uv[3], Uz[3];
#pragma unroll
    for (int rr = 0; rr < 3; rr++) {
      v2d s1 = v2d{0.0, 0.0}, s2 = v2d{0.0, 0.0};
#pragma unroll
      for (int c = 0; c < 3; c++) {
        s1 += f_cmul(U[3 * rr + c], plus ? psia[c] : phia[c]);
        s2 += f_cmul(U[3 * rr + c], plus ? psib[c] : phib[c]);
      }
      Uv[rr] = s1; Uz[rr] = s2;
    }
    const v2d ka = v2d{a.ka[mu][0], a.ka[mu][1]};
    v2d m[3][3];
#pragma unroll
    for (int rr = 0; rr < 3; rr++)
#pragma unroll
      for (int c = 0; c < 3; c++)
        m[rr][c] = f_cmul(ka, f_cmulc(Uv[rr], plus ? phia[c] : psia[c]) + f_cmulc(Uz[rr], plus ? phib[c] : psib[c]));
    // su3adj.h:164-172
    double *d = dp + (size_t)mu * 8 * a.Vh;
    const size_t st = a.Vh;
    d[0 * st] += a.fac * (-m[1][0].y - m[0][1].y);
    d[1 * st] += a.fac * (+m[1][0].x - m[0][1].x);
    d[2 * st] += a.fac * (-m[0][0].y + m[1][1].y);
    d[3 * st] += a.fac * (-m[2][0].y - m[0][2].y);
    d[4 * st] += a.fac * (+m[2][0].x - m[0][2].x);
    d[5 * st] += a.fac * (-m[2][1].y - m[1][2].y);
    d[6 * st] += a.fac * (+m[2][1].x - m[1][2].x);
    d[7 * st] += a.fac * ((-m[0][0].y - m[1][1].y + 2.0 * m[2][2].y) * 0.577350269189625);
  }
}

// deriv[par][mu][8][Vh] -> su3adj df[V][4] (lexicographic)
// t=0 slices of l and k -> [24][face] send buffer (full spinors: the force needs all four spin components)
__global__ __launch_bounds__(256) void force_pack_kernel(const v2d *__restrict__ l, const v2d *__restrict__ k, int ns, int face, v2d *__restrict__ send) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= face) return;
  const int c = blockIdx.y;                       // 0..23
  send[(size_t)c * face + j] = c < 12 ? l[(size_t)c * ns + j] : k[(size_t)(c - 12) * ns + j];
}

__global__ __launch_bounds__(256) void deriv_to_lexic_kernel(const double *__restrict__ d, double *__restrict__ out, int Vh, int LX, int LY, int LZ, int toff) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Vh) return;
  const int par = blockIdx.y;
  const int LZh = LZ / 2;
  int r = i / LZh;
  const int y = r % LY;
  r /= LY;
  const int x = r % LX, t = r / LX;
  const int o = (t + x + y + par + toff) & 1;
  double *dst = out + (2 * (size_t)i + o) * 32;
  const double *src = d + (size_t)par * 32 * Vh + i;
#pragma unroll 8
  for (int e = 0; e < 32; e++) dst[e] = src[(size_t)e * Vh];
}

extern "C" {

int tmhip_derivative_zero(tmhip_ctx *ctx) {
  TMHIP_CHECK(hipSetDevice(ctx->device));
  const size_t bytes = (size_t)2 * 32 * ctx->Vh * sizeof(double);
  if (!ctx->deriv) TMHIP_CHECK(hipMalloc((void **)&ctx->deriv, bytes));
  TMHIP_CHECK(hipMemsetAsync(ctx->deriv, 0, bytes, ctx->stream));
  return 0;
}

static int force_check(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k) {
  if (!l || !k || l->kind != TMHIP_FIELD_EO || k->kind != TMHIP_FIELD_EO || l->prec || k->prec) TMHIP_FAIL("deriv_Sb needs fp64 one-parity fields");
  if (!ctx->gauge_set) TMHIP_FAIL("deriv_Sb called before tmhip_set_gauge");
  if (l->ns != k->ns) TMHIP_FAIL("deriv_Sb: fields with different strides");
  if (!ctx->deriv && tmhip_derivative_zero(ctx)) return 1;
  return 0;
}
static int force_halo_alloc(tmhip_ctx *ctx) {
  const size_t bytes = (size_t)24 * ctx->face * sizeof(v2d);
  if (!ctx->force_send) TMHIP_CHECK(hipMalloc((void **)&ctx->force_send, bytes));
  if (!ctx->force_recv) TMHIP_CHECK(hipMalloc((void **)&ctx->force_recv, bytes));
  return 0;
}
static int force_pack(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k) {
  hipLaunchKernelGGL(force_pack_kernel, dim3((ctx->face + 255) / 256, 24), dim3(256), 0, ctx->stream, (const v2d *)l->d, (const v2d *)k->d, l->ns,
                     ctx->face, ctx->force_send);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}
static int force_launch(tmhip_ctx *ctx, int ieo, tmhip_field *l, tmhip_field *k, double factor, const v2d *halo) {
  ForceArgs a;
  memset(&a, 0, sizeof(a));
  a.own[ieo ? 1 : 0] = l->d; a.own[ieo ? 0 : 1] = k->d;
  a.g = ctx->gauge; a.deriv = ctx->deriv;
  a.ns = l->ns; a.gs = ctx->gs; a.Vh = ctx->Vh; a.T = ctx->g.T; a.LX = ctx->g.LX; a.LY = ctx->g.LY; a.LZ = ctx->g.LZ; a.ieo = ieo ? 1 : 0;
  a.toff = ctx->g.proc_t * ctx->g.T; a.halo = halo; a.face = ctx->face;
  for (int mu = 0; mu < 4; mu++) { a.ka[mu][0] = ctx->ka[mu][0]; a.ka[mu][1] = ctx->ka[mu][1]; }
  a.fac = 2. * factor;
  hipLaunchKernelGGL(deriv_Sb_kernel, dim3((ctx->Vh + 127) / 128, 2), dim3(128), 0, ctx->stream, a);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}

/* deriv_Sb(ieo, l, k, hf, factor)  deriv_Sb.c:401 -- accumulates into the device-resident derivative field.
 * T-split ranks first exchange the t=0 slices of both fields with the ring neighbours (the reference's xchange_2fields,
 * deriv_Sb.c:102; only the +t halo is needed here because every thread looks forward): RCCL send to rank-1 / receive
 * from rank+1 on the compute stream.  The force is not latency-critical, so no overlap is attempted. */
int tmhip_deriv_Sb(tmhip_ctx *ctx, int ieo, tmhip_field *l, tmhip_field *k, double factor) {
  if (force_check(ctx, l, k)) return 1;
  TMHIP_CHECK(hipSetDevice(ctx->device));
  if (ctx->g.nproc_t == 1 && !ctx->loopback) return force_launch(ctx, ieo, l, k, factor, nullptr);
  if (force_halo_alloc(ctx) || force_pack(ctx, l, k)) return 1;
  const size_t n = (size_t)24 * ctx->face * 2;   // doubles
  if (ctx->g.nproc_t == 1) {                      // loopback self-test: our own t=0 slice is the periodic +t neighbour
    TMHIP_CHECK(hipMemcpyAsync(ctx->force_recv, ctx->force_send, n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  } else {
    if (!ctx->comm_ready) TMHIP_FAIL("nproc_t > 1 but tmhip_comm_init was not called");
    const int np = ctx->g.nproc_t, up = (ctx->g.proc_t + 1) % np, dn = (ctx->g.proc_t + np - 1) % np;
    if (ctx->shm) { if (tmhip_shm_ring(ctx, ctx->stream, ctx->force_send, nullptr, ctx->force_recv, nullptr, n * sizeof(double))) return 1; }
    else {
    TMHIP_NCCL_CHECK(ncclGroupStart());
    TMHIP_NCCL_CHECK(ncclSend(ctx->force_send, n, ncclDouble, dn, ctx->comm_red, ctx->stream));
    TMHIP_NCCL_CHECK(ncclRecv(ctx->force_recv, n, ncclDouble, up, ctx->comm_red, ctx->stream));
    TMHIP_NCCL_CHECK(ncclGroupEnd());
    }
  }
  return force_launch(ctx, ieo, l, k, factor, ctx->force_recv);
}

/* Single-process ring (n contexts holding a T-split lattice, as tmhip_multi_hopping_matrix): slices move by peer copies. */
int tmhip_multi_deriv_Sb(int n, tmhip_ctx **ctxs, int ieo, tmhip_field **l, tmhip_field **k, double factor) {
  if (n < 2) TMHIP_FAIL("tmhip_multi_deriv_Sb needs >= 2 contexts");
  for (int r = 0; r < n; r++) {
    tmhip_ctx *c = ctxs[r];
    if (c->g.nproc_t != n || c->g.proc_t != r) TMHIP_FAIL("context %d is not rank %d of a %d-way T split", r, r, n);
    TMHIP_CHECK(hipSetDevice(c->device));
    if (force_check(c, l[r], k[r]) || force_halo_alloc(c) || force_pack(c, l[r], k[r])) return 1;
  }
  for (int r = 0; r < n; r++) { TMHIP_CHECK(hipSetDevice(ctxs[r]->device)); TMHIP_CHECK(hipStreamSynchronize(ctxs[r]->stream)); }
  const size_t bytes = (size_t)24 * ctxs[0]->face * sizeof(v2d);
  for (int r = 0; r < n; r++) {
    tmhip_ctx *c = ctxs[r], *up = ctxs[(r + 1) % n];
    TMHIP_CHECK(hipSetDevice(c->device));
    TMHIP_CHECK(hipMemcpyPeerAsync(c->force_recv, c->device, up->force_send, up->device, bytes, c->stream));
    if (force_launch(c, ieo, l[r], k[r], factor, c->force_recv)) return 1;
  }
  for (int r = 0; r < n; r++) { TMHIP_CHECK(hipSetDevice(ctxs[r]->device)); TMHIP_CHECK(hipStreamSynchronize(ctxs[r]->stream)); }
  return 0;
}

/* Copies the device derivative field into the host's su3adj df[VOLUME][4] (hamiltonian_field_t::derivative);
 * accumulate != 0 adds to what the host array holds (other monomials' contributions), else overwrites. */
int tmhip_derivative_download(tmhip_ctx *ctx, void *host_df, int accumulate) {
  if (!host_df) TMHIP_FAIL("tmhip_derivative_download: null argument");
  if (!ctx->deriv) TMHIP_FAIL("tmhip_derivative_download: no derivative field on the device");
  TMHIP_CHECK(hipSetDevice(ctx->device));
  const size_t n = (size_t)ctx->V * 32, bytes = n * sizeof(double);
  if (tmhip_stage_reserve(ctx, bytes)) return 1;
  hipLaunchKernelGGL(deriv_to_lexic_kernel, dim3((ctx->Vh + 255) / 256, 2), dim3(256), 0, ctx->stream, (const double *)ctx->deriv,
                     (double *)ctx->stage, ctx->Vh, ctx->g.LX, ctx->g.LY, ctx->g.LZ, ctx->g.proc_t * ctx->g.T);
  TMHIP_CHECK(hipGetLastError());
  if (!accumulate) {
    TMHIP_CHECK(hipMemcpyAsync(host_df, ctx->stage, bytes, hipMemcpyDeviceToHost, ctx->stream));
    TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
  }
  double *tmp = (double *)malloc(bytes);
  if (!tmp) TMHIP_FAIL("out of host memory");
  TMHIP_CHECK(hipMemcpyAsync(tmp, ctx->stage, bytes, hipMemcpyDeviceToHost, ctx->stream));
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  double *h = (double *)host_df;
  for (size_t i = 0; i < n; i++) h[i] += tmp[i];
  free(tmp);
  return 0;
}

}  // extern "C"
