// Internal definitions shared by the HIP translation units of libtmlqcd_hip.so.
// gfx950 only.  Device layouts:
//   spinor, one parity : v2d d[12][ns]        comp c = 3*spin + colour, site = e/o sub-index
//   gauge copy         : v2d g[2][8][9][gs]   [parity of the output site][dir +t,-t,+x,-x,+y,-y,+z,-z][row-major su3 element][site]
//   half-spinor faces  : v2d h[6][face]       comp c = 3*half + colour
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include "../../include/tmlqcd_hip.h"

typedef double v2d __attribute__((ext_vector_type(2)));  // one complex double: .x = re, .y = im
typedef float v2f __attribute__((ext_vector_type(2)));   // one complex float (mixed-precision CG)
// fp32 spinor fields pair their components: [6][ns] of v4f = components (2m, 2m+1) of a site (hopping32.hip); a tmhip_field's
// d32 pointer keeps the element type v2f (12 * ns of them), the kernels view it as 6 * ns v4f
typedef float v4f __attribute__((ext_vector_type(4)));

#define TMHIP_CHECK(expr)                                                                           \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess) {                                                                         \
      fprintf(stderr, "[tmlqcd_hip] %s:%d: %s failed: %s\n", __FILE__, __LINE__, #expr,             \
              hipGetErrorString(_e));                                                               \
      return 1;                                                                                     \
    }                                                                                               \
  } while (0)

#define TMHIP_NCCL_CHECK(expr)                                                                      \
  do {                                                                                              \
    ncclResult_t _r = (expr);                                                                       \
    if (_r != ncclSuccess) {                                                                        \
      fprintf(stderr, "[tmlqcd_hip] %s:%d: %s failed: %s\n", __FILE__, __LINE__, #expr,             \
              ncclGetErrorString(_r));                                                              \
      return 1;                                                                                     \
    }                                                                                               \
  } while (0)

#define TMHIP_FAIL(...)                                                                             \
  do {                                                                                              \
    fprintf(stderr, "[tmlqcd_hip] " __VA_ARGS__);                                                   \
    fprintf(stderr, "\n");                                                                          \
    return 1;                                                                                       \
  } while (0)

struct tmhip_field {
  int kind;            // TMHIP_FIELD_EO | TMHIP_FIELD_FULL
  int prec;            // 0: fp64 (d), 1: fp32 (d32; `spinor32` of su3.h:65-68), EO only
  v2f *d32;
  v2d *d;              // EO: [12][ns]; FULL: even half then odd half, each [12][ns]
  int ns;              // site stride of one parity
  bool view;           // true for the even/odd views of a FULL field
  tmhip_field *half[2];  // FULL only: views
};

// The direct face carrier (tmhip_comm_init_ipc; loopback 3 = onto oneself): the ring neighbours' receive buffers and arrival words are
// mapped into this process (hipIpcOpenMemHandle) and whoever produces the projected faces -- the pack kernel, the exterior kernel, the
// boundary waves of a stencil kernel -- stores them there itself (write-through, system scope); the last wave to finish writes the push
// number into the neighbours' words.  No copy, no kernel of a communication library, nothing on the receiving GPU's compute units.
// Double-buffered by push parity; protocol and flow control: launch_direct (hopping_split.inc), DESIGN.md section 7a.
#define TMHIP_DIRECT_MAX_RANKS 64
struct TmhipDirect {
  bool on;
  void *mine;                   // ONE uncached device allocation: [2 push parities][from up | from dn] faces, then the arrival words
  v2d *rbuf[2][2];              // [parity][0: written by the up neighbour (halo_up), 1: by the down neighbour (halo_dn)]
  unsigned int *arr[2];         // arrival words, 128 bytes apart: [0] written by the up neighbour, [1] by the down neighbour
  void *peer_map[2];            // mappings to close at the end ([1] null when both neighbours are one rank, both null for the self-push)
  v2d *peer_buf[2][2];          // [parity][0: the up neighbour's "from dn" buffer (our t = T-1 projections go there), 1: the down neighbour's "from up" buffer (our t = 0)]
  unsigned int *peer_arr[2];    // [0] the up neighbour's arr[1], [1] the down neighbour's arr[0]
  unsigned int *count;          // sharded count-in words of the kernels that push ahead: [1 + 32 shards][32 words] (faces_count_in, hopping_impl.inc)
  unsigned int push_seq;        // pushes issued so far (the same number on every rank: all ranks run the same sequence of stencils)
  unsigned int last_push;       // the push the last communicating stencil consumed (Hopping_Matrix_nocom reads those faces again)
  const void *ahead_field; unsigned int ahead_push;   // the field whose faces were pushed AHEAD by the stencil that wrote it, and under which number
  int sharers;                  // ranks of this job that sit on this physical GPU (1 in production; the one-GPU rehearsals have more)
  // direct sums (tmhip_direct_allreduce): every rank's block is mapped (not only the ring neighbours'), a rank stores its partial sum into
  // slot [parity of the reduction's number][its rank] of EVERY rank and adds up its own row in rank order -- the same bits everywhere
  bool sums_on;
  void *peer_all[TMHIP_DIRECT_MAX_RANKS];   // mapping of rank r's block (nullptr: not mapped; [me] = mine)
  unsigned long long sum_seq;   // reductions done so far (the same number on every rank)
};

struct tmhip_ctx {
  tmhip_geom g;
  int device;
  int V, Vh, face;     // VOLUME, VOLUME/2, LX*LY*LZ/2
  int ns;              // padded spinor stride: Vh (+ 2*face reserved) rounded up to 64
  int gs;              // padded gauge stride
  int VPR;             // host VOLUMEPLUSRAND
  double kappa, mu, theta[4];
  double mu3;          // g_mu3 (global.h:197): the odd-odd clover term of the e/o clover operators twists with mu + mu3
  double ka[4][2];     // ka0..ka3 (re, im)
  hipStream_t stream, comm_stream;
  hipEvent_t ev_pack, ev_comm, ev_slots[16];
  v2d *gauge;          // [2][8][9][gs]
  bool gauge_set;
  // clover twisted mass: site-local blocks uploaded from the host's sw / sw_inv
  v2d *sw;             // [2 parity][6][9][gs]   sw[ix][a][b] -> block 2a+b
  v2d *sw_inv;         // [2 sign: +mu, -mu][8][9][gs]  sw_inv[icy][a][b] -> block 2a+b (even sites)
  bool clover_set;     // sw and sw_inv both valid
  bool sw_set;         // sw valid (after tmhip_sw_term / tmhip_set_clover)
  int sw_inv_sets;     // 2 when the -mu set of sw_inv is valid as well (mu != 0), else 1
  v2d *swpm;           // clover-force accumulators swm / swp (clover_leaf.c:141-172): [2][4][9][V], site = parity * Vh + e/o index
  v2d *gauge_raw;      // lexicographic gauge field [VPR][4][9] kept from the last tmhip_sw_term for tmhip_sw_all; gauge_raw_valid
  bool gauge_raw_valid;
  bool gauge_copy_current;   // the stencil's gauge copy was sorted from the links now in gauge_raw
  unsigned *io_sums;   // SciDAC checksum words A, B accumulated by the ILDG pack / unpack kernels (ildg.hip)
  int *sw_fail;        // device counter of near-singular pivots met by tmhip_sw_invert
  v2f *sw32, *sw_inv32; bool clover32_set;
  v2f *gauge32;        // fp32 twin of the gauge copy (g_gauge_field_copy_32), built on first use
  bool gauge32_set;
  // staging for host<->device layout conversion
  void *stage; size_t stage_bytes;
  // reductions
  double *partials; int max_partials; double *result_dev; double *result_host;
  // private scratch = DUM_MATRIX..DUM_MATRIX+2 of tm_operators.c:173-176
  tmhip_field *scratch[3];
  // solver work fields
  tmhip_field *sf[3]; tmhip_field *sf_extra;   // sf_extra: 4th fp64 field of rg_mixed_cg_her (allocated on first use)
  // fp32 twins for the mixed-precision CG (allocated on first use): scratch32 = g_spinor_field32[0..1]
  // (tm_operators_32.c Qtm_pm_psi_32), sf32 = solver_field32[0..3] (mixed_cg_her.c:72-102)
  tmhip_field *scratch32[2]; tmhip_field *sf32[4];
  // halo exchange
  // Two communicators over the same ranks: `comm` carries the half-spinor faces on comm_stream, `comm_red` (ncclCommSplit of
  // `comm`) everything issued on the main stream (scalar all-reduces, force halos) -- no communicator is driven from two streams.
  // comm_split false: ncclCommSplit is unavailable (or switched off, "comm_split" 0) and comm_red == comm; still correct, because
  // a face exchange is never in flight together with a main-stream collective (launch_split, hopping_split.inc).
  ncclComm_t comm, comm_red; bool comm_ready; bool comm_split; bool loopback; bool loopback_rccl;
  struct TmhipShm *shm;   // != nullptr: the ranks talk through the host-staged shared-memory transport (xfer_shm.hip) instead of RCCL; everything on `stream`
  v2d *send_up, *send_dn, *recv_up, *recv_dn;   // [6][face] each
  unsigned int *sync_flags; unsigned int hop_seq;  // [0] main stream reached stencil n, [1] faces of stencil n received, [2] a bounded wait gave up (table of all words: DESIGN.md section 7)
  unsigned long long flag_timeout_ticks;           // bound of the device-side flag waits in ticks of the 100 MHz clock (0 = none)
  TmhipDirect direct;                              // the direct face carrier (off unless tmhip_comm_init_ipc / loopback 3)
  const void *prepacked;                           // the field whose boundary-slice projections sit in the send buffers (written by the last exterior kernel), or nullptr
  unsigned int bcount_total;                       // direct carrier: blocks of its pack kernels counted in sync_flags[4] so far (the count is cumulative, never reset)
  int last_ext_partials;                           // partial sums the last split-phase stencil's exterior kernel wrote in front of the stencil kernel's (0: it had none)
  // fermion-force accumulator (force.hip): double [2 parity][4 mu][8][Vh]
  double *deriv;
  double *momenta;     // hamiltonian_field_t::momenta, su3adj [V][4] = double [V][4][8], resident for tmhip_update_gauge (md_update.hip)
  v2d *sw_ins;         // sw_all: the six anti-hermitian insertion matrices of every site, compact [6][5][V] (clover.hip, sw_insertion_kernel)
  v2d *swpm_halo_send, *swpm_halo_recv;   // T-split sw_all (owner-computes): [2 slices][30][LX LY LZ] our t = 0 / T-1 slices of sw_ins / the neighbours' t = T, -1
  v2d *force_send, *force_recv;   // T-split deriv_Sb: [24][face] t=0 slices of (l, k), ours / the up-neighbour's
  // device-resident CG state (cg.hip)
  void *cg_state; double *cg_hist; int cg_hist_len;
  int mixed_trace[256]; int mixed_trace_n;   // inner iteration count of every outer iteration of the last tmhip_mixed_cg_her
  // options
  int opt_block, opt_xcd, opt_minw, opt_occ, opt_occ32;        // stencil launch shape (tmhip_set_option, include/tmlqcd_hip.h)
  int opt_tgrp;
  int opt_gauge_cache;                                                  // -1 automatic; 0 / 1: small-lattice launches load the links with / without the streaming hint
  int opt_hopsplit;                                                     // -1 automatic, 0 / 1: the eight hops of a site spread over four waves (small unsplit lattices)
  int opt_stg32;                                                        // the same for the fp32 stencil (default 0: measured slower there)
  int opt_stg;                                                          // 1 = LDS-staged stencil (own-block input spinors staged once, y/z neighbours read from LDS)
  int opt_recon;                                                        // 12 = rebuild the third row of every link in registers (opt-in)
  int opt_split_sync;                                                   // 0: the exterior kernel / the pack kernel wait for a flag of the other stream (default); 1: HIP events, no device-side wait
  int opt_direct_form;                                                  // direct carrier: -1 automatic (one kernel per stencil while the boundary waves fit the wait budget), 0 stencil + exterior kernel, 1 one kernel whenever the shape allows
  int opt_direct_sums;                                                  // 1 (default): with the direct carrier the scalar sums over the ranks travel the same way (tmhip_direct_allreduce) instead of ncclAllReduce
  int opt_direct_order;                                                 // direct carrier, one-kernel form: bit 0 / bit 1 = boundary time-slices FIRST for a stencil whose faces are packed now / were pushed ahead (else last)
  int opt_prepack;                                                      // 1 (default): the exterior kernel projects the faces of its output for the next stencil of a chain
  int opt_comm_split;                                                   // 0: do not split off a second communicator (exercises the one-communicator fallback)
  int opt_cg_sync, opt_cg_batch, opt_cg_fused_dot, opt_cg_self;         // cg_her
  int opt_swall_order;                                                  // block order of the owner-computes sw_all: 0 one contiguous chunk per XCD, 1 slab order, 2 tile order (default; 4 / 8: x-planes per tile)
  int opt_swterm_order;                                                 // block order of sw_term: 0 one contiguous chunk per XCD, 1 (default) tiles walked through all time-slices
  double gauge_recon_dev;   // max |U_row2 - conj(row0 x row1)| over all links of the resident gauge field (-1: not measured)
};

// N is a site count of a one-parity field: 0 is a legal empty loop in the reference (linalg/*.c), anything outside [0, V/2] would
// run past the device arrays
#define LA_CHECK_N(who, zero_out)                                                                                       \
  do {                                                                                                                  \
    if (N < 0 || N > ctx->Vh) TMHIP_FAIL("%s: N = %d is outside [0, VOLUME/2 = %d]", who, N, ctx->Vh);                  \
    if (N == 0) { zero_out; return 0; }                                                                                 \
  } while (0)

// Threads per stencil block: "block" 0 (default) picks 64 for small local lattices -- below ~2 blocks of 256 per CU a launch is
// latency-bound and four times as many independent blocks finish sooner (8^4: 5.7 vs 7.5 us, 12^4: 9.9 vs 14.1, 20^4: 25.3 vs
// 33.0; level at 24^4, 256 ahead from there: profiles/r01_diagnostics.md) -- and 256 otherwise.
static inline int tmhip_hop_block(const tmhip_ctx *ctx) { return ctx->opt_block ? ctx->opt_block : (ctx->Vh < 131072 ? 64 : 256); }   // (8 x 32^3 = 131072 sites per parity: 256 threads + the LDS-staged kernel, 80.8 -> 75.2 us per {H_eo, H_oe} with the slab order; 4 x 32^3: 64 threads, 34.5 against 41.6)

// Reductions go through RCCL on T-split ranks -- and in the one-rank RCCL loopback (tmhip_comm_set_loopback(ctx, 2)), so that the
// multi-rank code path (partial sums, ncclAllReduce, scalar update as separate steps) runs in the single-GPU tests too.
static inline bool tmhip_reduce_over_ranks(const tmhip_ctx *ctx) { return ctx->comm_ready && (ctx->g.nproc_t > 1 || ctx->loopback_rccl); }

// ---- host-staged shared-memory transport (xfer_shm.hip): stream-ordered ring exchange / sum / gather over the ranks of the node ----
void tmhip_shm_destroy(tmhip_ctx *ctx);
int tmhip_shm_failed(tmhip_ctx *ctx);
int tmhip_shm_ring(tmhip_ctx *ctx, hipStream_t st, const void *to_dn, const void *to_up, void *from_up, void *from_dn, size_t bytes);
int tmhip_shm_allreduce(tmhip_ctx *ctx, hipStream_t st, double *x, int n);
int tmhip_shm_allgather(tmhip_ctx *ctx, hipStream_t st, const void *mine, void *all, size_t bytes);

// ---- direct face carrier (xfer_ipc.hip) ----
int tmhip_direct_init_self(tmhip_ctx *ctx);   // loopback 3
void tmhip_direct_destroy(tmhip_ctx *ctx);
// One wave's part of a direct sum (all 64 lanes call it; every lane returns the total): lane r stores this rank's value into slot
// [row of the reduction's number][me] of rank r's block -- value, drained, then the number: two stores in order, nothing relies on a
// 16-byte store arriving whole --, waits (bounded) for rank r's contribution to THIS reduction in the rank's own row, and the np values
// are added in rank order: the same bits on every rank.
struct TmhipSumSlot { double v; unsigned long long seq; };
struct TmhipSumArgs { TmhipSumSlot *peer[TMHIP_DIRECT_MAX_RANKS]; TmhipSumSlot *mine; int np, me; unsigned long long seq; unsigned int *err; unsigned long long ticks; };
__device__ __forceinline__ double tmhip_direct_sum_wave(double mine_v, const TmhipSumArgs &a) {
  const int r = (int)(threadIdx.x & 63);
  const int row = (int)(a.seq & 1ull) * TMHIP_DIRECT_MAX_RANKS;
  double v = 0.0;
  if (r < a.np) {
    TmhipSumSlot *dst = a.peer[r] + row + a.me;
    __hip_atomic_store(&dst->v, mine_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(&dst->seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const TmhipSumSlot *src = a.mine + row + r;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)(__hip_atomic_load(&src->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - a.seq) < 0) {
      __builtin_amdgcn_s_sleep(4);
      if (a.ticks && __builtin_amdgcn_s_memrealtime() - t0 > a.ticks) { __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    v = __hip_atomic_load(&src->v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  double total = 0.0;
  for (int k = 0; k < a.np; k++) total += __shfl(v, k, 64);
  return total;
}
int tmhip_direct_sum_args(tmhip_ctx *ctx, TmhipSumArgs *out);   // the arguments of the NEXT direct sum of this context (advances its number)
int tmhip_direct_allreduce(tmhip_ctx *ctx, double *x);   // sum of *x (device) over the ranks, in place, added in rank order; enqueued on ctx->stream (needs direct.sums_on)

// ---- launch helpers implemented across the .hip files ----
enum { EPI_STORE = 0, EPI_TM_TIMES = 1, EPI_TM_SUB_G5 = 2, EPI_TM_SUB = 3, EPI_TM_SUB_G5_DOT = 4, EPI_CLOVER_INV = 5, EPI_CLOVER_G5 = 6, EPI_CLOVER = 7,
       EPI_TM_SUB_G5_NRM = 8 /* + partials of |out|^2 */, EPI_TM_SUB_G5_RES = 9 /* resid -= alpha out, partials of |resid|^2; out not stored */,
       EPI_CLOVER_G5_NRM = 10, EPI_CLOVER_G5_RES = 11 /* the same two on top of the clover_gamma5 epilogue */ };
// `comm`: 0 no halo exchange (Hopping_Matrix_nocom), HOP_COMM exchange the faces of `in` first, HOP_COMM | HOP_CHAINED additionally
// promises that `in` is the output of this context's previous split-phase stencil and has not been written since (a composition
// like Qtm_pm_psi, the stencils of a fused CG iteration): its faces were projected by that stencil's exterior kernel already
// HOP_FEED: the caller expects the NEXT stencil of this context to gather this one's output (direct carrier: its faces are then
// pushed into the neighbours' buffers by the waves that complete the boundary slices); a wrong guess costs one unused push, never a result
enum { HOP_COMM = 1, HOP_CHAINED = 2, HOP_FEED = 4 };
int tmhip_launch_hopping(tmhip_ctx *ctx, int ieo, v2d *out, const v2d *in, const v2d *p, int epi,
                         double cre, double cim, int comm, const v2d *cw = nullptr);
// Small unsplit lattices (the hop-split kernel): mode 2 can compute its coefficient itself -- EVERY block adds up the `n` per-wave partial
// sums the previous reducing stencil left in `partials` (fixed order: the same value in every block, bitwise reproducible) and uses
// alpha = *normsq / sum; block 0 leaves sum and alpha in out2[0..1].  No sum + scalar kernel between the two stencils
// (a ~4.7 us floor each at any size: the difference between 14.8k and 17k CG iterations per second at 16^4).
struct HopSelfAlpha { const double *partials; int n; const double *normsq; double *out2; };
bool tmhip_hopping_self_alpha_ok(const tmhip_ctx *ctx);   // the launch below would take the hop-split kernel
// fixed-order sum of n doubles by a 256-thread block (every thread returns the total); wsum: 4 doubles of shared memory.  Shared by
// the stencil above and cg_xp_self_kernel (cg.hip): both must add in exactly the same order.
__device__ __forceinline__ double tmhip_block_sum256(const double *__restrict__ v, int n, double *wsum) {
  double acc = 0.0;
  for (int j = threadIdx.x; j < n; j += 256) acc += v[j];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  return (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}
// mode 0: partials of <dotv, out>; 1: of |out|^2; 2: resid -= (*scal) * out without storing out, partials of |resid|^2
int tmhip_launch_hopping_dot(tmhip_ctx *ctx, int ieo, v2d *out, const v2d *in, const v2d *p, const v2d *dotv,
                             double cre, double cim, int *npartials, int mode = 0, v2d *resid = nullptr, const double *scal = nullptr,
                             const v2d *cw = nullptr, int chained = 0, const HopSelfAlpha *self = nullptr);   // cw: clover blocks => clover_gamma5 epilogue (modes 1, 2 only); chained: HOP_CHAINED
int tmhip_launch_hopping32(tmhip_ctx *ctx, int ieo, v2f *out, const v2f *in, const v2f *p, int epi,
                           double cre, double cim, int comm, const v2f *cw = nullptr);
int tmhip_launch_hopping_dot32(tmhip_ctx *ctx, int ieo, v2f *out, const v2f *in, const v2f *p, const v2f *dotv,
                               double cre, double cim, int *npartials, int mode = 0, v2f *resid = nullptr, const double *scal = nullptr,
                               const v2f *cw = nullptr, int chained = 0);
bool tmhip_fused_dot32_ok(const tmhip_ctx *ctx);
int tmhip_reduce_finish(tmhip_ctx *ctx, int nblocks, int parallel, double *out);
// After a host-visible synchronisation of a T-split rank: non-zero (with a message) when a bounded device-side wait for the
// neighbours' faces gave up ("flag_timeout_ms") or the communicator reports an asynchronous error, i.e. the result just
// synchronised cannot be trusted.  Reported once per occurrence -- the error word is cleared, the next call starts clean.
int tmhip_check_async_error(tmhip_ctx *ctx);
extern "C" int tmhip_check_gauge_recon(tmhip_ctx *ctx);   // context.hip: unitarity guard of the gauge_recon=12 option
int tmhip_stage_reserve(tmhip_ctx *ctx, size_t bytes);
int tmhip_field_alloc_prec(tmhip_ctx *ctx, int kind, int prec, tmhip_field **out);
int tmhip_halo_exchange(tmhip_ctx *ctx);
int tmhip_apply_op(tmhip_ctx *ctx, int op, tmhip_field *l, tmhip_field *k);
int tmhip_prepare_fp32(tmhip_ctx *ctx);
int tmhip_exchange_gauge_halo(tmhip_ctx *ctx);   // md_update.hip: t = 0 / T-1 slices of the resident links -> the ring neighbours' halo slabs
int tmhip_resort_gauge(tmhip_ctx *ctx);   // md_update.hip: stencil gauge copy from the device-resident lexicographic links
int tmhip_prepare_clover32(tmhip_ctx *ctx);  // fp32 gauge copy + fp32 scratch / solver fields
// launch geometry shared by linalg.hip and cg.hip
#define LA_BS 256
#define LA_UNROLL 4
static inline dim3 la_grid(int N) { return dim3((N + LA_BS * LA_UNROLL - 1) / (LA_BS * LA_UNROLL), 12); }
static inline dim3 la_grid32(int N) { return dim3((N + LA_BS * LA_UNROLL - 1) / (LA_BS * LA_UNROLL), 6); }   // fp32 fields: six float4 planes
