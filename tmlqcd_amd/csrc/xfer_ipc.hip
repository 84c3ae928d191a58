// The direct face carrier of a T-split job (round 4): every rank maps its two ring neighbours' receive buffers and arrival words
// (hipIpcGetMemHandle / hipIpcOpenMemHandle) and the kernels that PRODUCE the projected half-spinor faces store them straight into the
// neighbour's memory -- over xGMI between the GPUs of a node, through the same mapping between processes that share one GPU (the
// one-GPU rehearsals).  It replaces, for the faces only, the MPI_Isend / MPI_Irecv pair of xchange/xchange_halffield.c:176-263 and the
// wait of operator/halfspinor_body.c:281-317; everything else (scalar sums, the force / gauge / clover halos, the ILDG checksum gather)
// stays on the communicator the context already has (RCCL, or the host-staged transport), which also carries the 96 bytes per rank this
// file needs once: the IPC handle and the PCI bus id of the rank's GPU.
//
// Why stores and not the copy engine (tools/micro/ipc_probe.hip, profiles/r04_ipc_probe.log): hipMemcpyDeviceToDeviceNoCU needs no
// compute unit and is indifferent to a busy chip, but moves the two faces of a 32^3 time-slice (3 MB) in 64 us (14 us + 50 GB/s) --
// longer than the whole stencil of an 8 x 32^3 rank (40 us); the default device-to-device copy is a kernel that queues behind the stencil's
// blocks.  The producing waves' own 16-byte write-through stores cost nothing but the bytes.
//
// The receive buffers are UNCACHED device memory (hipDeviceMallocUncached, what RCCL uses for its own buffers on this architecture):
// lines written by another agent must never be served from a stale copy in this GPU's L2.
#include "tmhip_internal.h"

#include <string>
#include <vector>

namespace {

struct Card {                    // what every rank tells the others (<= 120 bytes: one slot of the host-staged transport's gather)
  hipIpcMemHandle_t handle;      // 64 bytes
  char busid[32];                // PCI bus id of the rank's GPU: ranks with the same one share a GPU
};
static_assert(sizeof(Card) <= 120, "one gather slot");

size_t face_bytes(const tmhip_ctx *ctx) { return (size_t)6 * ctx->face * sizeof(v2d); }

int alloc_mine(tmhip_ctx *ctx) {
  TmhipDirect &d = ctx->direct;
  const size_t fb = face_bytes(ctx), total = 4 * fb + 4096;
  // TMLQCD_HIP_DIRECT_ALLOC = plain | finegrained | uncached (A/B): what the receive buffers are made of
  int kind = 3;
  if (const char *e = getenv("TMLQCD_HIP_DIRECT_ALLOC")) kind = !strcmp(e, "plain") ? 0 : (!strcmp(e, "finegrained") ? 1 : 3);
  if (kind == 0) TMHIP_CHECK(hipMalloc(&d.mine, total));
  else TMHIP_CHECK(hipExtMallocWithFlags(&d.mine, total, kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached));
  TMHIP_CHECK(hipMemsetAsync(d.mine, 0, total, ctx->stream));
  TMHIP_CHECK(hipMalloc((void **)&d.count, 33 * 128));
  TMHIP_CHECK(hipMemsetAsync(d.count, 0, 33 * 128, ctx->stream));
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  char *b = (char *)d.mine;
  for (int p = 0; p < 2; p++) for (int w = 0; w < 2; w++) d.rbuf[p][w] = (v2d *)(b + (size_t)(2 * p + w) * fb);
  d.arr[0] = (unsigned int *)(b + 4 * fb); d.arr[1] = (unsigned int *)(b + 4 * fb + 128);
  return 0;
}

// pointers into a neighbour's allocation (same layout as alloc_mine)
void point_at(tmhip_ctx *ctx, int which /* 0 up, 1 down */, char *base) {
  TmhipDirect &d = ctx->direct;
  const size_t fb = face_bytes(ctx);
  // the up neighbour receives our t = T-1 projections in ITS "from down" buffer (index 1) and looks at ITS arr[1]; the down neighbour
  // our t = 0 projections in its "from up" buffer (index 0) and arr[0]
  const int theirs = which == 0 ? 1 : 0;
  for (int p = 0; p < 2; p++) d.peer_buf[p][which] = (v2d *)(base + (size_t)(2 * p + theirs) * fb);
  d.peer_arr[which] = (unsigned int *)(base + 4 * fb + (size_t)theirs * 128);
}

void reset_state(tmhip_ctx *ctx) {
  ctx->direct.push_seq = 0; ctx->direct.ahead_field = nullptr; ctx->direct.ahead_push = 0;
  ctx->prepacked = nullptr;
}

}  // namespace

// loopback 3: the periodic wrap onto ourselves through the direct carrier's code path (no mapping: our own buffers are the neighbours')
int tmhip_direct_init_self(tmhip_ctx *ctx) {
  TmhipDirect &d = ctx->direct;
  if (d.mine) { d.on = true; return 0; }   // (switched off and on again: buffers, words and push numbers carry on)
  TMHIP_CHECK(hipSetDevice(ctx->device));
  if (alloc_mine(ctx)) return 1;
  point_at(ctx, 0, (char *)d.mine);
  point_at(ctx, 1, (char *)d.mine);
  d.peer_map[0] = d.peer_map[1] = nullptr;
  d.sharers = 1;
  reset_state(ctx);
  d.on = true;
  return 0;
}

void tmhip_direct_destroy(tmhip_ctx *ctx) {
  TmhipDirect &d = ctx->direct;
  if (!d.mine) return;
  for (int r = 0; r < TMHIP_DIRECT_MAX_RANKS; r++) if (d.peer_all[r] && d.peer_all[r] != d.mine) (void)hipIpcCloseMemHandle(d.peer_all[r]);
  (void)hipFree(d.mine);
  if (d.count) (void)hipFree(d.count);
  memset(&d, 0, sizeof(d));
}

// ---- direct sums: MPI_Allreduce(..., MPI_SUM) of one double (linalg/square_norm.c:314, the alpha and the stopping test of cg_her) --------
// tmhip_direct_sum_wave (tmhip_internal.h) is one wave's exchange; here it is a kernel of its own, and cg.hip puts it between the sum of a
// stencil's partial sums and the scalar update of the CG state, in one launch.  Two slot rows in turn are enough: a rank cannot be two
// reductions ahead of one whose contribution it still needs.
__global__ __launch_bounds__(64) void direct_allreduce_kernel(double *x, const TmhipSumArgs a) {
  const double total = tmhip_direct_sum_wave(*x, a);
  if (threadIdx.x == 0) *x = total;
}

int tmhip_direct_sum_args(tmhip_ctx *ctx, TmhipSumArgs *out) {
  TmhipDirect &d = ctx->direct;
  if (!d.on || !d.sums_on) TMHIP_FAIL("a direct sum without the direct sums (tmhip_comm_init_ipc, option direct_sums)");
  const size_t off = 4 * face_bytes(ctx) + 1024;               // the slot rows lie behind the arrival words (alloc_mine)
  memset(out, 0, sizeof(*out));
  out->np = ctx->g.nproc_t; out->me = ctx->g.proc_t;
  for (int r = 0; r < out->np; r++) out->peer[r] = (TmhipSumSlot *)((char *)d.peer_all[r] + off);
  out->mine = (TmhipSumSlot *)((char *)d.mine + off);
  out->seq = ++d.sum_seq;
  out->err = ctx->sync_flags + 2; out->ticks = ctx->flag_timeout_ticks;
  return 0;
}

int tmhip_direct_allreduce(tmhip_ctx *ctx, double *x) {
  TmhipSumArgs a;
  if (tmhip_direct_sum_args(ctx, &a)) return 1;
  hipLaunchKernelGGL(direct_allreduce_kernel, dim3(1), dim3(64), 0, ctx->stream, x, a);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}

extern "C" {

/* Collective over the ranks of the T split, after tmhip_comm_init or tmhip_comm_init_shm: from here on the half-spinor faces of every
 * stencil travel as direct stores into the ring neighbours' memory.  Returns non-zero (and leaves the context on its communicator's own
 * face exchange) when a neighbour's memory cannot be mapped. */
int tmhip_comm_init_ipc(tmhip_ctx *ctx) {
  TmhipDirect &d = ctx->direct;
  if (d.on) TMHIP_FAIL("tmhip_comm_init_ipc: the direct carrier is already on");
  // A single rank has nothing to map -- except behind the one-rank RCCL communicator of loopback 2, where the whole collective set-up
  // (gather of the cards, the all-or-none sum, both over RCCL) runs with np = 1 and the rank becomes its own neighbour: the single-GPU
  // rehearsal of this function's RCCL calls and of the direct sums
  if (ctx->g.nproc_t < 2 && !(ctx->loopback_rccl && ctx->comm_ready && !ctx->shm)) return 0;
  if (!ctx->comm_ready) TMHIP_FAIL("tmhip_comm_init_ipc: call tmhip_comm_init or tmhip_comm_init_shm first (the handles travel over that communicator)");
  TMHIP_CHECK(hipSetDevice(ctx->device));
  const int np = ctx->g.nproc_t, me = ctx->g.proc_t, up = (me + 1) % np, dn = (me + np - 1) % np;
  if (d.mine) tmhip_direct_destroy(ctx);     // (a block of an earlier loopback 3)
  if (alloc_mine(ctx)) return 1;
  Card mine;
  memset(&mine, 0, sizeof(mine));
  TMHIP_CHECK(hipIpcGetMemHandle(&mine.handle, d.mine));
  if (hipDeviceGetPCIBusId(mine.busid, (int)sizeof(mine.busid), ctx->device) != hipSuccess) snprintf(mine.busid, sizeof(mine.busid), "device%d", ctx->device);
  // gather the cards of all ranks over the communicator the context has
  if (tmhip_stage_reserve(ctx, (size_t)(np + 1) * sizeof(Card))) return 1;
  char *dev_mine = (char *)ctx->stage, *dev_all = dev_mine + sizeof(Card);
  TMHIP_CHECK(hipMemcpyAsync(dev_mine, &mine, sizeof(Card), hipMemcpyHostToDevice, ctx->stream));
  if (ctx->shm) {
    if (tmhip_shm_allgather(ctx, ctx->stream, dev_mine, dev_all, sizeof(Card))) return 1;
  } else {
    TMHIP_NCCL_CHECK(ncclAllGather(dev_mine, dev_all, sizeof(Card), ncclChar, ctx->comm_red, ctx->stream));
  }
  std::vector<Card> all((size_t)np);
  TMHIP_CHECK(hipMemcpyAsync(all.data(), dev_all, (size_t)np * sizeof(Card), hipMemcpyDeviceToHost, ctx->stream));
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (tmhip_check_async_error(ctx)) return 1;
  if (memcmp(&all[(size_t)me], &mine, sizeof(Card))) TMHIP_FAIL("tmhip_comm_init_ipc: the gather did not return this rank's own card in slot %d", me);
  d.sharers = 0;
  for (int r = 0; r < np; r++) if (!strncmp(all[(size_t)r].busid, mine.busid, sizeof(mine.busid))) d.sharers++;
  if (d.count == nullptr) TMHIP_FAIL("tmhip_comm_init_ipc: no count-in words");
  // map the ring neighbours (faces) and -- for the sums -- everybody else too; [me] is this rank's own block
  d.peer_all[me] = d.mine;
  hipError_t e = hipSuccess, e_all = np <= TMHIP_DIRECT_MAX_RANKS && ctx->opt_direct_sums ? hipSuccess : hipErrorNotSupported;
  for (int k = 1; k < np && k < TMHIP_DIRECT_MAX_RANKS; k++) {
    const int r = (me + k) % np;                         // (neighbours first)
    const bool ring = r == up || r == dn;
    if (!ring && e_all != hipSuccess) continue;
    const hipError_t er = hipIpcOpenMemHandle(&d.peer_all[r], all[(size_t)r].handle, hipIpcMemLazyEnablePeerAccess);
    if (er != hipSuccess) {
      d.peer_all[r] = nullptr;
      if (ring) e = er; else e_all = er;
      fprintf(stderr, "[tmlqcd_hip] tmhip_comm_init_ipc: rank %d cannot map rank %d's block (%s)\n", me, r, hipGetErrorString(er));
    }
  }
  // all ranks or none: a rank that pushes to a neighbour that still posts receives would hang both (and the same for the sums)
  double bad[2] = {e != hipSuccess ? 1.0 : 0.0, e_all != hipSuccess ? 1.0 : 0.0};
  TMHIP_CHECK(hipMemcpyAsync(ctx->result_dev, bad, sizeof(bad), hipMemcpyHostToDevice, ctx->stream));
  if (ctx->shm) { if (tmhip_shm_allreduce(ctx, ctx->stream, ctx->result_dev, 2)) return 1; }
  else TMHIP_NCCL_CHECK(ncclAllReduce(ctx->result_dev, ctx->result_dev, 2, ncclDouble, ncclSum, ctx->comm_red, ctx->stream));
  TMHIP_CHECK(hipMemcpyAsync(bad, ctx->result_dev, sizeof(bad), hipMemcpyDeviceToHost, ctx->stream));
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (tmhip_check_async_error(ctx)) return 1;
  if (bad[0] != 0.0) {
    tmhip_direct_destroy(ctx);
    TMHIP_FAIL("tmhip_comm_init_ipc: %d rank(s) could not map a neighbour: the faces stay on the communicator", (int)bad[0]);
  }
  d.peer_map[0] = d.peer_map[1] = nullptr;               // (the mappings are kept in peer_all)
  point_at(ctx, 0, (char *)d.peer_all[up]);
  point_at(ctx, 1, (char *)d.peer_all[dn]);
  reset_state(ctx);
  d.sum_seq = 0;
  d.sums_on = bad[1] == 0.0;                              // every rank has every block: the scalar sums travel the same way as the faces
  d.on = true;
  return 0;
}

/* 0: faces travel over the context's communicator; 1: direct stores into the neighbours' memory.  *sharers (may be null): ranks of the
 * job on this rank's GPU. */
int tmhip_comm_faces_direct(tmhip_ctx *ctx, int *sharers) {
  if (sharers) *sharers = ctx->direct.on ? ctx->direct.sharers : 0;
  return ctx->direct.on ? 1 : 0;
}

/* 1 when the scalar sums over the ranks travel as direct stores too (tmhip_direct_allreduce), 0: over the communicator */
int tmhip_comm_sums_direct(tmhip_ctx *ctx) { return ctx->direct.on && ctx->direct.sums_on ? 1 : 0; }

}  // extern "C"
