// Host-staged transport of a T-split job whose ranks share a node but not an RCCL communicator: faces, halo slices and scalars
// travel device -> page-locked host buffer -> a POSIX shared-memory segment -> the neighbour's page-locked buffer -> device.  It is what
// the reference does with MPI on host memory (xchange/xchange_field.c:98-250: MPI_Isend / MPI_Irecv / MPI_Waitall, linalg/square_norm.c:
// 299-316: MPI_Allreduce), stream-ordered: every operation is copy-out, ONE host function on the stream, copy-in -- a drop-in for the
// RCCL call at the same place (the faces on the comm stream under the stencil kernel, with the flags of launch_split; everything
// else on the compute stream).  PCIe both ways: this is not the fast path (RCCL over xGMI is) -- it is the path that needs nothing
// but the node's memory, and the one that lets several ranks of a job share ONE GPU, i.e. run the multi-rank code, split-phase
// protocol included, as real processes on a one-GPU box (tests/test_gpu_multiprocess.py, bench.py with TMLQCD_BENCH_TRANSPORT=shm).
//
// Every rank issues the same sequence of operations, so a host function may simply wait for its neighbours: they are in the same call
// or on their way to it.  The mailboxes carry ONE kind of message at a time: a face exchange runs between the start of its stencil kernel
// and the exterior kernel, and no ring operation of the compute stream lies in between; the sums use slots of their own.
// Bounded like every wait of this library (flag_timeout_ms): a dead neighbour becomes an error of the next synchronising call.
#include "tmhip_internal.h"

#include <atomic>
#include <chrono>
#include <cerrno>
#include <fcntl.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

namespace {

struct Mailbox {                       // one message at a time, one writer (the neighbour), one reader (the owner)
  std::atomic<unsigned long long> written, consumed;
  unsigned long long bytes;
  char pad[40];
};
struct Slot {                          // one rank's contribution to a reduction / gather (two of them in turn)
  std::atomic<unsigned long long> seq;
  char payload[120];
};
struct Header {
  std::atomic<unsigned long long> magic;
  std::atomic<int> arrived, left;
  int nranks;
  unsigned long long mailbox_bytes;
  int creator_pid;                     // rank 0 of the run that made this segment: a joiner that finds it dead has opened a leftover
  int pad0;
  unsigned long long nonce;            // its start time: told apart from an earlier run that recycled the pid
  char pad[16];
};
static_assert(sizeof(Mailbox) == 64 && sizeof(Slot) == 128 && sizeof(Header) == 64, "layout shared between processes");
const unsigned long long MAGIC = 0x746d6869705f7368ull;

}  // namespace

struct TmhipShm {
  char name[96];
  void *base; size_t total;
  Header *hdr;
  int nranks, rank;
  size_t mailbox_bytes;
  std::atomic<unsigned long long> sent[2], received[2];   // [0]: to / from the down neighbour, [1]: up (atomics: the host functions of the two streams may run on different threads)
  std::atomic<unsigned long long> red_seq;
  char *stage_send[2], *stage_recv[2];       // page-locked, mailbox_bytes each
  char *stage_small;                          // page-locked, reductions
  double timeout_s;
  std::atomic<int> failed;
};

namespace {

inline char *rank_base(const TmhipShm *s, int r) {
  const size_t per_rank = 2 * (sizeof(Mailbox) + s->mailbox_bytes) + 2 * sizeof(Slot);
  return (char *)s->base + sizeof(Header) + (size_t)r * per_rank;
}
// mailbox `which` of rank r: 0 = messages from its DOWN neighbour, 1 = from its UP neighbour
inline Mailbox *mailbox(const TmhipShm *s, int r, int which) { return (Mailbox *)(rank_base(s, r) + (size_t)which * (sizeof(Mailbox) + s->mailbox_bytes)); }
inline Slot *slot(const TmhipShm *s, int r, int parity) { return (Slot *)(rank_base(s, r) + 2 * (sizeof(Mailbox) + s->mailbox_bytes)) + parity; }

template <class F> bool wait_for(TmhipShm *s, F ready) {
  if (ready()) return true;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; spins++) {
    if (ready()) return true;
    if (s->failed.load(std::memory_order_relaxed)) return false;
    if (spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(20));
    if ((spins & 1023) == 1023 && s->timeout_s > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > s->timeout_s) {
      s->failed.store(1);
      return false;
    }
  }
}

struct RingArgs { TmhipShm *s; size_t bytes; bool to_dn, to_up, from_up, from_dn; };
void ring_cb(void *p) {
  RingArgs *a = (RingArgs *)p;
  TmhipShm *s = a->s;
  const int np = s->nranks, up = (s->rank + 1) % np, dn = (s->rank + np - 1) % np;
  // what goes DOWN arrives at the down neighbour as a message from its UP neighbour (mailbox 1), and the other way round
  for (int dir = 0; dir < 2; dir++) {
    if (!(dir == 0 ? a->to_dn : a->to_up)) continue;
    Mailbox *m = mailbox(s, dir == 0 ? dn : up, dir == 0 ? 1 : 0);
    const unsigned long long n = s->sent[dir].load();
    if (!wait_for(s, [&] { return m->consumed.load(std::memory_order_acquire) == n; })) break;   // the previous message has been taken out
    memcpy((char *)(m + 1), s->stage_send[dir], a->bytes);
    m->bytes = a->bytes;
    m->written.store(n + 1, std::memory_order_release);
    s->sent[dir].store(n + 1);
  }
  for (int dir = 0; dir < 2; dir++) {                  // dir 0: from the up neighbour (our mailbox 1) -> stage_recv[0]; dir 1: from down
    if (!(dir == 0 ? a->from_up : a->from_dn)) continue;
    Mailbox *m = mailbox(s, s->rank, dir == 0 ? 1 : 0);
    const unsigned long long n = s->received[dir].load();
    if (!wait_for(s, [&] { return m->written.load(std::memory_order_acquire) == n + 1; })) break;
    if (m->bytes != a->bytes) s->failed.store(2);      // the ranks are not in the same operation
    memcpy(s->stage_recv[dir], (const char *)(m + 1), a->bytes);
    m->consumed.store(n + 1, std::memory_order_release);
    s->received[dir].store(n + 1);
  }
  delete a;
}

struct GatherArgs { TmhipShm *s; size_t bytes; int reduce_doubles; };   // reduce_doubles > 0: sum that many doubles over the ranks, in rank order
void gather_cb(void *p) {
  GatherArgs *a = (GatherArgs *)p;
  TmhipShm *s = a->s;
  const unsigned long long seq = s->red_seq.fetch_add(1) + 1;
  const int par = (int)(seq & 1);
  Slot *mine = slot(s, s->rank, par);
  memcpy(mine->payload, s->stage_small, a->bytes);
  mine->seq.store(seq, std::memory_order_release);
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int r = 0; r < s->nranks; r++) {
    Slot *sl = slot(s, r, par);
    if (!wait_for(s, [&] { return sl->seq.load(std::memory_order_acquire) == seq; })) break;
    if (a->reduce_doubles) { for (int k = 0; k < a->reduce_doubles; k++) acc[k] += ((const double *)sl->payload)[k]; }   // rank order: the same bits on every rank
    else memcpy(s->stage_small + 128 + (size_t)r * a->bytes, sl->payload, a->bytes);
  }
  if (a->reduce_doubles) memcpy(s->stage_small, acc, a->reduce_doubles * sizeof(double));
  delete a;
}

}  // namespace

extern "C" int tmhip_comm_init_shm(tmhip_ctx *ctx, const char *job) {
  if (ctx->g.nproc_t < 2) return 0;
  if (ctx->comm_ready) TMHIP_FAIL("tmhip_comm_init_shm: this context already has a communicator");
  if (!job || !*job || strlen(job) > 64) TMHIP_FAIL("tmhip_comm_init_shm: job name of 1 .. 64 characters");
  TMHIP_CHECK(hipSetDevice(ctx->device));
  TmhipShm *s = new TmhipShm();
  memset((void *)s, 0, sizeof(*s));
  snprintf(s->name, sizeof(s->name), "/tmhip_%s", job);
  s->nranks = ctx->g.nproc_t; s->rank = ctx->g.proc_t;
  // the largest message is a time-slice of links (update_gauge's halo, 576 B per site) or of insertion matrices (sw_all, 480 B per site)
  s->mailbox_bytes = ((size_t)ctx->g.LX * ctx->g.LY * ctx->g.LZ * 576 + 4095) / 4096 * 4096;
  s->timeout_s = ctx->flag_timeout_ticks ? (double)ctx->flag_timeout_ticks * 1.0e-8 : 0.0;
  const size_t per_rank = 2 * (sizeof(Mailbox) + s->mailbox_bytes) + 2 * sizeof(Slot);
  s->total = sizeof(Header) + (size_t)s->nranks * per_rank;
  // Meeting in the segment is bounded by INIT_WAIT_S whatever the flag timeout says (0 = none must not mean "wait for a rank that died
  // before it got here" for ever), and a segment left behind by a job that died during ITS meeting is recognised, not joined:
  // rank 0 removes the name and creates the segment anew (O_EXCL), stamping it with its pid and start time; a joiner takes a segment
  // only while its creator is alive and not everybody has arrived in it yet, and otherwise lets go of it and opens the name again.
  const double INIT_WAIT_S = 120.0;
  const auto t_init = std::chrono::steady_clock::now();
  auto init_left = [&]() { return INIT_WAIT_S - std::chrono::duration<double>(std::chrono::steady_clock::now() - t_init).count(); };
  auto fail = [&](int fd_) { if (fd_ >= 0) close(fd_); if (s->base && s->base != MAP_FAILED) munmap(s->base, s->total); delete s; };
  int fd = -1;
  if (s->rank == 0) {
    shm_unlink(s->name);
    fd = shm_open(s->name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)s->total)) {
      fprintf(stderr, "[tmlqcd_hip] tmhip_comm_init_shm: cannot create %s (%zu bytes): %s\n", s->name, s->total, strerror(errno));
      fail(fd);
      return 1;
    }
    s->base = mmap(nullptr, s->total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd); fd = -1;
    if (s->base == MAP_FAILED) { fprintf(stderr, "[tmlqcd_hip] tmhip_comm_init_shm: mmap of %s failed: %s\n", s->name, strerror(errno)); fail(-1); return 1; }
    s->hdr = (Header *)s->base;       // (a fresh segment is zero-filled: counters and sequence numbers start at 0)
    s->hdr->nranks = s->nranks; s->hdr->mailbox_bytes = s->mailbox_bytes;
    s->hdr->creator_pid = (int)getpid();
    s->hdr->nonce = (unsigned long long)std::chrono::system_clock::now().time_since_epoch().count();
    s->hdr->magic.store(MAGIC, std::memory_order_release);
    s->hdr->arrived.fetch_add(1);
  } else {
    for (;;) {   // until this rank sits in a segment that fills up
      if (init_left() <= 0) { fprintf(stderr, "[tmlqcd_hip] tmhip_comm_init_shm: rank 0 did not create a live %s within %.0f s\n", s->name, INIT_WAIT_S); fail(fd); return 1; }
      fd = shm_open(s->name, O_RDWR, 0600);
      struct stat st;
      if (fd < 0 || fstat(fd, &st) || (size_t)st.st_size != s->total) {
        if (fd >= 0) { close(fd); fd = -1; }
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
        continue;
      }
      const ino_t ino = st.st_ino;
      void *b = mmap(nullptr, s->total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      close(fd); fd = -1;
      if (b == MAP_FAILED) { fprintf(stderr, "[tmlqcd_hip] tmhip_comm_init_shm: mmap of %s failed: %s\n", s->name, strerror(errno)); fail(-1); return 1; }
      Header *h = (Header *)b;
      // wait (briefly) for the creator's stamp, then judge the segment
      bool seated = false;
      for (int spin = 0; spin < 400 && init_left() > 0; spin++) {
        if (h->magic.load(std::memory_order_acquire) == MAGIC) {
          const bool alive = h->creator_pid > 0 && (kill((pid_t)h->creator_pid, 0) == 0 || errno == EPERM);
          if (h->nranks != s->nranks || h->mailbox_bytes != s->mailbox_bytes) {
            fprintf(stderr, "[tmlqcd_hip] tmhip_comm_init_shm: %s belongs to another job (ranks / lattice differ)\n", s->name);
            munmap(b, s->total); fail(-1);
            return 1;
          }
          // claim a seat; a segment in which everybody had already arrived (or whose creator is gone) is a leftover of a dead run
          if (alive) {
            if (h->arrived.fetch_add(1) < s->nranks) seated = true;
            else h->arrived.fetch_sub(1);
          }
          break;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
      }
      // seated: wait for the others -- but keep an eye on the NAME: if it now leads to another segment (or to none, while seats are still
      // empty here), this one is the leftover of a run that died during its meeting and the live run's rank 0 has started over
      bool full = false, stale = !seated;
      for (unsigned it = 0; seated && !full && !stale && init_left() > 0; it++) {
        full = h->arrived.load() >= s->nranks;
        if (full) break;
        if (it % 64 == 63) {
          const int fd2 = shm_open(s->name, O_RDWR, 0600);
          struct stat st2;
          if (fd2 < 0) stale = h->arrived.load() < s->nranks;
          else { stale = !fstat(fd2, &st2) && st2.st_ino != ino; close(fd2); }
        }
        std::this_thread::sleep_for(std::chrono::microseconds(it < 2000 ? 50 : 2000));
      }
      if (full) { s->base = b; s->hdr = h; break; }
      if (seated && !stale) { fprintf(stderr, "[tmlqcd_hip] tmhip_comm_init_shm: only %d of %d ranks arrived in %s within %.0f s\n", h->arrived.load(), s->nranks, s->name, INIT_WAIT_S); munmap(b, s->total); fail(-1); return 1; }
      munmap(b, s->total);
      std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
  }
  if (s->rank == 0) {
    const double saved = s->timeout_s;
    s->timeout_s = init_left() > 1.0 ? init_left() : 1.0;           // bounded, also with flag_timeout 0
    const bool all = wait_for(s, [&] { return s->hdr->arrived.load() >= s->nranks; });
    s->timeout_s = saved;
    if (!all) {
      fprintf(stderr, "[tmlqcd_hip] tmhip_comm_init_shm: only %d of %d ranks arrived in %s within %.0f s\n", s->hdr->arrived.load(), s->nranks, s->name, INIT_WAIT_S);
      shm_unlink(s->name);
      fail(-1);
      return 1;
    }
    s->failed.store(0);
  }
  if (s->rank == 0) shm_unlink(s->name);   // everybody has it mapped: the name can go, the memory goes with the last rank
  for (int k = 0; k < 2; k++) {
    TMHIP_CHECK(hipHostMalloc((void **)&s->stage_send[k], s->mailbox_bytes, hipHostMallocDefault));
    TMHIP_CHECK(hipHostMalloc((void **)&s->stage_recv[k], s->mailbox_bytes, hipHostMallocDefault));
  }
  TMHIP_CHECK(hipHostMalloc((void **)&s->stage_small, 128 + 128 * (size_t)s->nranks, hipHostMallocDefault));
  ctx->shm = s;
  ctx->comm_ready = true; ctx->comm_split = false;
  return 0;
}

void tmhip_shm_destroy(tmhip_ctx *ctx) {
  TmhipShm *s = ctx->shm;
  if (!s) return;
  for (int k = 0; k < 2; k++) { (void)hipHostFree(s->stage_send[k]); (void)hipHostFree(s->stage_recv[k]); }
  (void)hipHostFree(s->stage_small);
  munmap(s->base, s->total);
  delete s;
  ctx->shm = nullptr;
}

int tmhip_shm_failed(tmhip_ctx *ctx) {
  if (!ctx->shm) return 0;
  const int f = ctx->shm->failed.exchange(0);
  if (f == 1) TMHIP_FAIL("the shared-memory transport gave up waiting for a neighbour after %.0f s: results since the last check are invalid", ctx->shm->timeout_s);
  if (f) TMHIP_FAIL("the shared-memory transport met a message of another size: the ranks are not executing the same sequence of operations");
  return 0;
}

// device pointers; a null pointer = nothing sent / expected in that direction.  to_dn arrives in the down neighbour's from_up.
int tmhip_shm_ring(tmhip_ctx *ctx, hipStream_t st, const void *to_dn, const void *to_up, void *from_up, void *from_dn, size_t bytes) {
  TmhipShm *s = ctx->shm;
  if (!s) TMHIP_FAIL("tmhip_shm_ring without the shared-memory transport");
  if (bytes > s->mailbox_bytes) TMHIP_FAIL("tmhip_shm_ring: message of %zu bytes, mailboxes hold %zu", bytes, s->mailbox_bytes);
  if (to_dn) TMHIP_CHECK(hipMemcpyAsync(s->stage_send[0], to_dn, bytes, hipMemcpyDeviceToHost, st));
  if (to_up) TMHIP_CHECK(hipMemcpyAsync(s->stage_send[1], to_up, bytes, hipMemcpyDeviceToHost, st));
  RingArgs *a = new RingArgs{s, bytes, to_dn != nullptr, to_up != nullptr, from_up != nullptr, from_dn != nullptr};
  TMHIP_CHECK(hipLaunchHostFunc(st, ring_cb, a));
  if (from_up) TMHIP_CHECK(hipMemcpyAsync(from_up, s->stage_recv[0], bytes, hipMemcpyHostToDevice, st));
  if (from_dn) TMHIP_CHECK(hipMemcpyAsync(from_dn, s->stage_recv[1], bytes, hipMemcpyHostToDevice, st));
  return 0;
}

// sum of n <= 8 doubles over the ranks, in place on the device, added in rank order on every rank (bitwise the same everywhere)
int tmhip_shm_allreduce(tmhip_ctx *ctx, hipStream_t st, double *x, int n) {
  TmhipShm *s = ctx->shm;
  if (!s || n < 1 || n > 8) TMHIP_FAIL("tmhip_shm_allreduce: 1 .. 8 doubles over the shared-memory transport");
  TMHIP_CHECK(hipMemcpyAsync(s->stage_small, x, n * sizeof(double), hipMemcpyDeviceToHost, st));
  TMHIP_CHECK(hipLaunchHostFunc(st, gather_cb, new GatherArgs{s, n * sizeof(double), n}));
  TMHIP_CHECK(hipMemcpyAsync(x, s->stage_small, n * sizeof(double), hipMemcpyHostToDevice, st));
  return 0;
}

// `bytes` <= 120 of every rank, in rank order, into all[nranks * bytes] on the device
int tmhip_shm_allgather(tmhip_ctx *ctx, hipStream_t st, const void *mine, void *all, size_t bytes) {
  TmhipShm *s = ctx->shm;
  if (!s || bytes < 1 || bytes > 120) TMHIP_FAIL("tmhip_shm_allgather: 1 .. 120 bytes per rank over the shared-memory transport");
  TMHIP_CHECK(hipMemcpyAsync(s->stage_small, mine, bytes, hipMemcpyDeviceToHost, st));
  TMHIP_CHECK(hipLaunchHostFunc(st, gather_cb, new GatherArgs{s, bytes, 0}));
  TMHIP_CHECK(hipMemcpyAsync(all, s->stage_small + 128, (size_t)s->nranks * bytes, hipMemcpyHostToDevice, st));
  return 0;
}
