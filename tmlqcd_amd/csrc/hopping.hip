// Even/odd Wilson twisted-mass hopping stencil for gfx950 (MI355X), fp64.
//
// Computes, for every site x of parity ieo,
//   l(x) = sum_mu [ ka_mu U_mu(x) (1+g_mu) k(x+mu) + conj(ka_mu) U_mu(x-mu)^dag (1-g_mu) k(x-mu) ]
// exactly as the reference's generic body (operator/hopping_body_dbl.c:27-181 with the
// macros of operator/hopping.h:574-694), optionally followed by the fused epilogues of
// tm_times_Hopping_Matrix (hopping.h:674-678) and tm_sub_Hopping_Matrix (hopping.h:680-688).
//
// MI355X mapping:
//  * one thread per output site, lanes consecutive in the e/o sub-index -> every global
//    load is a 16 B/lane, 1 KiB/wave coalesced read of one SoA plane (12 spinor planes,
//    72 gauge planes); the z/y/x/t neighbours of a wave are contiguous runs of the same
//    planes, so the 8-fold spinor re-use is served by L1/L2/Infinity Cache and HBM sees
//    each input spinor ~once.
//  * neighbour indices are computed arithmetically from (t,x,y,k): the reference's
//    g_hi gather table (64 B/site, geometry_eo.c:1470-1535) is never read.
//  * gauge links are used exactly once per call -> loaded non-temporally so they do
//    not evict the re-used spinor lines from L2.
//  * MFMA is not used: 3x3 complex mat-vec at 1 flop/B, HBM-bound by ~10x.
#include "hopping_common.h"

namespace hop64 {
TMHIP_SCALAR_COMPLEX_OPS(v2d, double)
TMHIP_SPINOR_IO_PLANES
#define HOP_SITES 1
#define HOP_CTX_GAUGE(ctx) ((ctx)->gauge)
#define HOP_CTX_GAUGE_READY(ctx) ((ctx)->gauge_set)
#define HOP_CTX_OCC(ctx) ((ctx)->opt_occ)
#define HOP_CTX_STG(ctx) ((ctx)->opt_stg)
#include "hopping_impl.inc"
#undef HOP_CTX_OCC
#undef HOP_CTX_STG
#undef HOP_CTX_GAUGE
#undef HOP_CTX_GAUGE_READY
#undef HOP_SITES
}  // namespace hop64

TMHIP_DECLARE_HOP32(hop32)

int tmhip_launch_hopping(tmhip_ctx *ctx, int ieo, v2d *out, const v2d *in, const v2d *p, int epi,
                         double cre, double cim, int comm, const v2d *cw) {
  return hop64::launch_hopping(ctx, ieo, out, in, p, epi, cre, cim, comm, cw);
}
int tmhip_launch_hopping_dot(tmhip_ctx *ctx, int ieo, v2d *out, const v2d *in, const v2d *p, const v2d *dotv,
                             double cre, double cim, int *npartials, int mode, v2d *resid, const double *scal, const v2d *cw, int chained, const HopSelfAlpha *self) {
  return hop64::launch_hopping_dot(ctx, ieo, out, in, p, dotv, cre, cim, npartials, mode, resid, scal, cw, chained, self);
}
int tmhip_launch_hopping32(tmhip_ctx *ctx, int ieo, v2f *out, const v2f *in, const v2f *p, int epi,
                           double cre, double cim, int comm, const v2f *cw) {
  return hop32::launch_hopping(ctx, ieo, out, in, p, epi, cre, cim, comm, cw);
}
int tmhip_launch_hopping_dot32(tmhip_ctx *ctx, int ieo, v2f *out, const v2f *in, const v2f *p, const v2f *dotv,
                               double cre, double cim, int *npartials, int mode, v2f *resid, const double *scal, const v2f *cw, int chained) {
  return hop32::launch_hopping_dot(ctx, ieo, out, in, p, dotv, cre, cim, npartials, mode, resid, scal, cw, chained);
}
bool tmhip_hopping_self_alpha_ok(const tmhip_ctx *ctx) {
  return hop64::use_split4(ctx) && ctx->opt_recon != 12 && ctx->Vh % 64 == 0 && ctx->Vh / 64 <= ctx->max_partials / 2;
}
bool tmhip_fused_dot32_ok(const tmhip_ctx *ctx) {
  const bool split = ctx->g.nproc_t > 1 || ctx->loopback;
  const int sites = 1;
  const int spb = (split ? 256 : tmhip_hop_block(ctx)) * sites;
  return ctx->Vh % spb == 0;
}

// Single-process ring: n contexts (one per GPU, or several on one GPU for the self-test) that
// together hold a T-split lattice; faces move by peer copies instead of RCCL.  Collective over
// all contexts because the host enqueues for every rank:
//   1. every rank packs its two faces            (after its neighbours finished reading the previous ones)
//   2. every rank pulls the neighbours' faces on its comm stream
//   3. every rank runs the stencil over all its sites but for the hops across the cuts, then -- once its pulls have landed -- the
//      exterior kernel that adds those
extern "C" int tmhip_multi_hopping_matrix(int n, tmhip_ctx **ctxs, int ieo, tmhip_field **l, tmhip_field **k) {
  if (n < 2) TMHIP_FAIL("tmhip_multi_hopping_matrix needs >= 2 contexts");
  for (int r = 0; r < n; r++) {
    tmhip_ctx *c = ctxs[r];
    if (c->g.nproc_t != n || c->g.proc_t != r) TMHIP_FAIL("context %d is not rank %d of a %d-way T split", r, r, n);
    if (!c->gauge_set) TMHIP_FAIL("Hopping_Matrix called before tmhip_set_gauge");
    if (l[r]->kind != TMHIP_FIELD_EO || k[r]->kind != TMHIP_FIELD_EO || l[r]->d == k[r]->d) TMHIP_FAIL("bad fields for rank %d", r);
  }
  const size_t fb = (size_t)6 * ctxs[0]->face * sizeof(v2d);
  for (int r = 0; r < n; r++) {
    tmhip_ctx *c = ctxs[r], *up = ctxs[(r + 1) % n], *dn = ctxs[(r + n - 1) % n];
    TMHIP_CHECK(hipSetDevice(c->device));
    TMHIP_CHECK(hipStreamWaitEvent(c->stream, up->ev_comm, 0));  // neighbours still pulling the previous faces
    TMHIP_CHECK(hipStreamWaitEvent(c->stream, dn->ev_comm, 0));
    hop64::launch_pack(c, k[r]->d, c->stream);
    TMHIP_CHECK(hipEventRecord(c->ev_pack, c->stream));
  }
  for (int r = 0; r < n; r++) {
    tmhip_ctx *c = ctxs[r], *up = ctxs[(r + 1) % n], *dn = ctxs[(r + n - 1) % n];
    TMHIP_CHECK(hipSetDevice(c->device));
    TMHIP_CHECK(hipStreamWaitEvent(c->comm_stream, up->ev_pack, 0));
    TMHIP_CHECK(hipStreamWaitEvent(c->comm_stream, dn->ev_pack, 0));
    // ... and after this rank's own pack: it sits behind the previous call's exterior kernel in c->stream order, and that
    // kernel still reads recv_up / recv_dn -- the new faces must not land under it (back-to-back calls, slow ranks)
    TMHIP_CHECK(hipStreamWaitEvent(c->comm_stream, c->ev_pack, 0));
    TMHIP_CHECK(hipMemcpyPeerAsync(c->recv_up, c->device, up->send_dn, up->device, fb, c->comm_stream));
    TMHIP_CHECK(hipMemcpyPeerAsync(c->recv_dn, c->device, dn->send_up, dn->device, fb, c->comm_stream));
    TMHIP_CHECK(hipEventRecord(c->ev_comm, c->comm_stream));
  }
  for (int r = 0; r < n; r++) {
    tmhip_ctx *c = ctxs[r];
    TMHIP_CHECK(hipSetDevice(c->device));
    hop64::HopArgs a;
    hop64::fill_args(a, c, ieo, l[r]->d, k[r]->d, nullptr, 0, 0);
    hop64::launch_epi<1>(a, EPI_STORE, c->stream, hop64::launch_opts(c, tmhip_hop_block(c)));
    TMHIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_comm, 0));
    hop64::launch_exterior(c, a, EPI_STORE, c->stream);
    TMHIP_CHECK(hipGetLastError());
  }
  return 0;
}
