// libtmlqcd_dropin.so -- tmLQCD's own hot-path symbols on top of the HIP core library.
//
// Host side only (no device code here): reads the reference's globals at call time, keeps
// a registry host-pointer -> device mirror, and forwards to include/tmlqcd_hip.h.
// Each entry point cites the reference function it replaces (paths under /root/reference).
#include "../../include/tmlqcd_dropin.h"
#include "../../include/tmlqcd_hip.h"

#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <vector>

#include <atomic>
#include <cerrno>
#include <dlfcn.h>
#include <fcntl.h>
#include <pthread.h>
#include <signal.h>
#include <stdint.h>
#include <sys/mman.h>
#include <ucontext.h>
#include <unistd.h>

extern "C" {
// ---- globals owned by the host program (global.h, boundary.h) ----
extern int T, LX, LY, LZ, VOLUME, RAND, VOLUMEPLUSRAND;         /* global.h:82-84 */
extern int g_nproc_t, g_nproc_x, g_nproc_y, g_nproc_z;           /* global.h:206 */
extern int g_proc_coords[4];                                      /* global.h:207 */
extern su3 **g_gauge_field;                                       /* global.h:176 */
extern int g_update_gauge_copy;                                   /* global.h:73  */
extern int g_update_gauge_copy_32 __attribute__((weak));          /* global.h:74: the host's fp32 gauge copy is refreshed by host code when it needs it */
extern double g_mu;                                               /* global.h:198 */
extern double g_mu3 __attribute__((weak));                        /* global.h:197: odd-odd twist of the e/o clover operators is g_mu + g_mu3 */
extern TM_COMPLEX ka0, ka1, ka2, ka3;                             /* boundary.h:25 */
extern su3 ***sw __attribute__((weak));                           /* clovertm_operators.c:58 */
extern su3 ***sw_inv __attribute__((weak));                       /* clovertm_operators.c:59 */
extern double mixcg_innereps __attribute__((weak));               /* read_input.h:112 (only needed by mixed_cg_her) */
extern int mixcg_maxinnersolverit __attribute__((weak));          /* read_input.h:113 */
// Present in a full tmLQCD link (update_backward_gauge.c, libhmc.a); refreshes the HOST gauge
// copy that deriv_Sb.c:405-408,472 still reads, and clears g_update_gauge_copy.
void update_backward_gauge(su3 **const gf) __attribute__((weak));
// ILDG I/O (io/gauge_read.c, io/gauge_write.c): the reader's precision switch and the IO-check switch are owned by the input parser
extern int gauge_precision_read_flag __attribute__((weak));       /* read_input.l; default 64 */
extern int g_disable_IO_checks __attribute__((weak));             /* global.h:74 */
extern int T_global __attribute__((weak));                        /* global.h:82 */
extern int L __attribute__((weak));
}

namespace {

// Third mirror shape next to the two field kinds of the core library: the first `n` spinors of a host array taken as a plain
// sequence (the reference's linalg and site-diagonal routines loop over ANY 0 <= N; tests/test_linalg_spinor.c uses N = 2 and
// 1000, block solvers use block volumes).  Stored in a FULL-sized device field without the lexicographic <-> e/o permutation:
// sites [0, VOLUME/2) in its first half, [VOLUME/2, n) in its second.
#define KIND_LIN 2

struct Mirror {
  tmhip_field *f = nullptr;
  int kind = TMHIP_FIELD_EO;
  int n = 0;                // KIND_LIN: number of sites mirrored
  bool dev_valid = false;   // device copy holds the current data
  bool host_valid = true;   // host copy holds the current data
  unsigned long long last_use = 0;
  // lazy mode (TMLQCD_HIP_LAZY): the host array's pages are protected so that the host's own loads and stores say when a copy is needed
  size_t bytes = 0;         // extent of the host array this mirror stands for
  int prot = 0;             // P_RW: untouched; P_RO: both copies current, a host store must be noticed; P_NONE: the host copy is stale
  std::vector<unsigned char> page_ok;   // P_NONE: pages of the span already brought up to date one by one.  Sized ONCE, when the mirror is made
                                        // (lazy mode): the SIGSEGV handler and everything it calls only ever overwrite it
  int faults = 0;           // page-wise read synchronisations since the device last wrote the field
  bool nowatch = false;     // lazy mode: this array cannot be watched (malloc heap / arena, shared or file-backed mapping) -- never protected, copied per call like the coherent mode
  bool classified = false;  // lazy mode: nowatch / unsafe were decided
  bool unsafe = false;      // test hook TMLQCD_HIP_LAZY_FORCE_WATCH: watched although unwatchable -- a fault on it ends the program with a message (never a hang)
};
enum { P_RW = 0, P_RO = 1, P_NONE = 2 };
unsigned long long g_tick = 0;
size_t g_mirror_cap = 64;   // TMLQCD_HIP_MAX_MIRRORS: host programs that allocate work fields per solve (solver_field.c) would otherwise
                            // grow the registry without bound; mirrors whose host copy is current can be dropped at any time

tmhip_ctx *g_ctx = nullptr;
int g_device = -1;
int g_mode = TMLQCD_HIP_COHERENT;
int g_dims[6] = {0, 0, 0, 0, 0, 0};
std::unordered_map<const void *, Mirror> g_reg;
// What the SIGSEGV handler of the lazy mode walks instead of the map: a fixed array of (host array, its mirror) kept in step with g_reg
// under the lock (mirrors live in map nodes: their addresses are stable).  Reading it allocates nothing and follows no bucket chain.
struct Watch { const void *host; Mirror *m; };
constexpr int WATCH_CAP = 4096;
Watch g_watch[WATCH_CAP];
int g_nwatch = 0;
bool g_gauge_uploaded = false;   // the current context holds a gauge copy
bool g_dev_links_newer = false;   // resident mode: the device links are ahead of g_gauge_field until tmlqcd_hip_sync_gauge_to_host
bool g_momenta_resident = false;  // the momenta live on the device (tmlqcd_hip_update_momenta), not re-uploaded by tmlqcd_hip_update_gauge
bool g_clover_uploaded = false;
tmhip_field *g_full_tmp = nullptr; // FULL-lattice scratch of Q_pm_psi / D_dagg_psi (tm_operators.c:380-397)
tmhip_field *g_f32[3] = {nullptr, nullptr, nullptr};   // device fields of the fp32 host-pointer symbols (Hopping_Matrix_32 ...)

// ONE lock for the registry: taken around every change of g_reg or of a mirror's state by the entry points and for the whole body of
// the SIGSEGV handler of the lazy mode, which walks the map -- a host thread faulting on a stale field while the master thread is
// inside a drop-in call must never see a rehash in progress.  Recursive per thread (mirror() -> evict; a fault of the thread that
// holds it, e.g. in the memcpy of an upload, is served in place: no structural change is in progress then).  A spin lock: pthread
// mutexes are not async-signal-safe.  The owner's thread id IS the lock word (0: free): "do I hold it already" is then one atomic
// load that only the asking thread itself can have made true -- an owner id kept next to a separate flag can be read stale by a
// thread that held the lock before, which then walks in beside the new owner.
std::atomic<uintptr_t> g_reg_owner(0);
int g_reg_depth = 0;                              // touched by the owner only
struct RegLock {
  RegLock() {
    const uintptr_t me = (uintptr_t)pthread_self();
    if (g_reg_owner.load(std::memory_order_relaxed) == me) { g_reg_depth++; return; }
    uintptr_t expected = 0;
    while (!g_reg_owner.compare_exchange_weak(expected, me, std::memory_order_acquire)) { expected = 0; __builtin_ia32_pause(); }
    g_reg_depth = 1;
  }
  ~RegLock() {
    if (--g_reg_depth == 0) g_reg_owner.store(0, std::memory_order_release);
  }
};

[[noreturn]] void die(const char *what) {
  fprintf(stderr, "[tmlqcd_dropin] fatal: %s\n", what);
  exit(1);  // the reference's error convention (fatal_error.c)
}
#define CK(call) do { if ((call) != 0) die(#call); } while (0)

void install_lazy_handler();
tmhip_ctx *ctx() {
  if (!g_ctx) {
    if (g_nproc_x != 1 || g_nproc_y != 1 || g_nproc_z != 1)
      die("only T-direction decomposition is supported (g_nproc_x/y/z must be 1)");
    if (g_device < 0) {
      const char *e = getenv("TMLQCD_HIP_DEVICE");
      g_device = e ? atoi(e) : 0;
    }
    const char *r = getenv("TMLQCD_HIP_RESIDENCY");       // unmodified executables: TMLQCD_HIP_RESIDENCY=lazy ./benchmark
    if (r && !strcmp(r, "lazy") && g_mode == TMLQCD_HIP_COHERENT) { install_lazy_handler(); g_mode = TMLQCD_HIP_LAZY; }
    else if (r && !strcmp(r, "resident") && g_mode == TMLQCD_HIP_COHERENT) g_mode = TMLQCD_HIP_RESIDENT;
    tmhip_geom g = {T, LX, LY, LZ, g_nproc_t < 1 ? 1 : g_nproc_t, g_proc_coords[0]};
    CK(tmhip_create(&g, g_device, &g_ctx));
    g_dims[0] = T; g_dims[1] = LX; g_dims[2] = LY; g_dims[3] = LZ; g_dims[4] = g.nproc_t; g_dims[5] = g.proc_t;
  } else if (g_dims[0] != T || g_dims[1] != LX || g_dims[2] != LY || g_dims[3] != LZ) {
    die("lattice extents changed after the first call");
  }
  return g_ctx;
}

// Re-read everything the reference reads through globals (SURVEY §8b "Data it reads through globals").
unsigned long g_calls = 0;   // entry-point calls served (tmlqcd_hip_calls): lets an integration test see that a symbol resolved to this library
tmhip_ctx *refresh(bool need_gauge) {
  tmhip_ctx *c = ctx();
  g_calls++;
  const double ka[8] = {__real__ ka0, __imag__ ka0, __real__ ka1, __imag__ ka1,
                        __real__ ka2, __imag__ ka2, __real__ ka3, __imag__ ka3};
  CK(tmhip_set_ka(c, ka));
  CK(tmhip_set_mu(c, g_mu));
  CK(tmhip_set_mu3(c, &g_mu3 ? g_mu3 : 0.));
  if (need_gauge && (g_update_gauge_copy || !g_gauge_uploaded)) {   /* Hopping_Matrix.c:135-139 */
    // A raised flag always means "the host's links changed since the device last saw them": the device paths that bring both sides
    // to the same state (tmlqcd_hip_update_gauge, read_gauge_field, tmlqcd_hip_sync_gauge_to_host) clear it themselves
    // (links_in_step), so a raise that follows -- the reject step restoring the old links (update_tm.c), a host-side
    // reunitarisation -- is never mistaken for our own.
    if (update_backward_gauge) update_backward_gauge(g_gauge_field);  // host copy + flag, as the reference
    else g_update_gauge_copy = 0;
    CK(tmhip_set_gauge(c, &g_gauge_field[0][0]));
    g_dev_links_newer = false;                                        // the host's links are the truth again
    g_gauge_uploaded = true;
  }
  return c;
}

// ------------------------------------------------------------------ lazy coherence (TMLQCD_HIP_LAZY)
// An UNMODIFIED host program keeps its fields in HBM: after a device operation wrote a field, the pages of the host array are made
// inaccessible; the host's first load from one of them faults, the handler brings that page up to date from the device mirror (a few
// microseconds: 21 spinors) and lets the load go on -- or the whole field once the host keeps reading (more than LAZY_PAGE_FAULTS pages)
// or stores to it.  After an upload the pages are read-only, so a host store invalidates the mirror.  benchmark.c's loop (it reads one
// number of the output per iteration, :291-300) then runs at the resident rate with no source change.  Limits, hence opt-in: the
// kernel does not raise SIGSEGV for its own accesses -- a field handed to write(2) / MPI while its host copy is stale fails with
// EFAULT instead of being synchronised (tmlqcd_hip_sync_to_host first); pages shared with neighbouring data are handled, at the price
// of a synchronisation when that data is touched.
#define LAZY_PAGE_FAULTS 8
uintptr_t g_page = 4096;
struct sigaction g_old_segv;
bool g_handler_installed = false;
std::atomic<uintptr_t> g_handler_thread(0);       // the thread the SIGSEGV handler is running on (0: none) -- one word, see RegLock
inline bool in_handler_here() { return g_handler_thread.load(std::memory_order_relaxed) == (uintptr_t)pthread_self(); }
unsigned long g_lazy_stats[4] = {0, 0, 0, 0};   // faults served, pages fetched one by one, whole-field fetches, stores noticed (tmlqcd_hip_lazy_stats)

inline uintptr_t span_lo(const void *h) { return (uintptr_t)h & ~(g_page - 1); }
inline uintptr_t span_hi(const void *h, size_t bytes) { return ((uintptr_t)h + bytes + g_page - 1) & ~(g_page - 1); }
inline int prot_flags(int p) { return p == P_RW ? (PROT_READ | PROT_WRITE) : (p == P_RO ? PROT_READ : PROT_NONE); }
// what mirror m asks for page `page` of its span
inline int page_want(const void *host, const Mirror &m, uintptr_t page) {
  if (m.prot != P_NONE) return m.prot;
  const size_t idx = (page - span_lo(host)) / g_page;
  return idx < m.page_ok.size() && m.page_ok[idx] ? P_RO : P_NONE;
}
// the strictest protection any mirror asks for this page (pages at the edge of a field are shared with its neighbours)
int page_need(uintptr_t page, const std::unordered_map<const void *, Mirror> & /* the watch table mirrors it */) {
  int need = P_RW;
  for (int k = 0; k < g_nwatch; k++) {
    const Mirror &m = *g_watch[k].m;
    const void *host = g_watch[k].host;
    if (m.prot == P_RW || !m.bytes) continue;
    if (page < span_lo(host) || page >= span_hi(host, m.bytes)) continue;
    const int w = page_want(host, m, page);
    if (w > need) need = w;
  }
  return need;
}
// g_reg changed (insert / erase): bring the handler's table in step.  Entry-point context, under the lock.
void rebuild_watch() {
  g_nwatch = 0;
  for (auto &kv : g_reg) {
    if (g_nwatch == WATCH_CAP) { fprintf(stderr, "[tmlqcd_dropin] fatal: more than %d mirrored host arrays\n", WATCH_CAP); exit(1); }
    g_watch[g_nwatch++] = Watch{kv.first, &kv.second};
  }
}
// (re)apply the protection of one mirror's span; interior pages belong to it alone, the two edge pages are negotiated
void apply_prot(const void *host, Mirror &m, const std::unordered_map<const void *, Mirror> &reg) {
  if (!m.bytes) return;
  const uintptr_t lo = span_lo(host), hi = span_hi(host, m.bytes);
  for (uintptr_t pg = lo; pg < hi; pg += g_page) {
    const bool edge = pg < (uintptr_t)host || pg + g_page > (uintptr_t)host + m.bytes;
    if (edge) { mprotect((void *)pg, g_page, prot_flags(page_need(pg, reg))); continue; }
    // run of interior pages with the same wish
    const int w = page_want(host, m, pg);
    uintptr_t end = pg + g_page;
    while (end < hi && end + g_page <= (uintptr_t)host + m.bytes && page_want(host, m, end) == w) end += g_page;
    mprotect((void *)pg, end - pg, prot_flags(w));
    pg = end - g_page;
  }
}
void set_prot(const void *host, Mirror &m, int prot, const std::unordered_map<const void *, Mirror> &reg) {
  if (m.prot == prot && prot != P_NONE) return;
  if (m.prot == P_NONE && prot == P_NONE) {
    // the device wrote the field again while the host copy was already closed: only the pages the host had fetched in between
    // need closing (none at all in a loop of device calls -- an mprotect over the whole 100 MB span costs milliseconds)
    if (m.faults == 0) return;
    const uintptr_t lo = span_lo(host);
    for (size_t i = 0; i < m.page_ok.size(); i++)
      if (m.page_ok[i]) { m.page_ok[i] = 0; mprotect((void *)(lo + i * g_page), g_page, prot_flags(page_need(lo + i * g_page, reg))); }
    m.faults = 0;
    return;
  }
  m.prot = prot;
  const size_t npages = (span_hi(host, m.bytes) - span_lo(host)) / g_page;
  if (prot == P_NONE && m.page_ok.size() != npages) m.page_ok.assign(npages, 0);   // (entry points only: the handler never closes a span)
  else std::fill(m.page_ok.begin(), m.page_ok.end(), 0);
  if (prot == P_NONE) m.faults = 0;
  apply_prot(host, m, reg);
}

int kind_of_N(int N) {
  if (N == VOLUME / 2) return TMHIP_FIELD_EO;
  if (N == VOLUME) return TMHIP_FIELD_FULL;
  if (N > 0 && N < VOLUME) return KIND_LIN;
  die("linalg/site-diagonal call with N outside [0, VOLUME]");
}

// The element-wise routines work part by part: one part for a one-parity field or a short prefix, two for a FULL field (its two
// halves) or a prefix longer than VOLUME/2.
struct Parts { int n; int cnt[2]; };
Parts parts_of(int kind, int N) {
  const int Vh = VOLUME / 2;
  if (kind == TMHIP_FIELD_EO) return {1, {Vh, 0}};
  if (kind == TMHIP_FIELD_FULL) return {2, {Vh, Vh}};
  if (N <= Vh) return {1, {N, 0}};
  return {2, {Vh, N - Vh}};
}

int nsites(int kind) { return kind == TMHIP_FIELD_FULL ? VOLUME : VOLUME / 2; }

// Lazy mode never lets the runtime touch the program's own pages: a copy from / to pageable memory registers those pages with the
// driver, and every later mprotect on them goes through its MMU notifier (measured: 28 ms per call instead of microseconds).  Data
// moves through a page-locked bounce buffer instead; uploads and whole-field downloads are the rare events in this mode.
// Two of them: the entry points' and the fault handler's.  An upload copies host -> bounce with memcpy, and that copy can itself
// fault (an edge page shared with a neighbouring field whose host copy is stale, a stale mirror overlapping the span); the handler's
// whole-field download of that neighbour must not land in -- or re-allocate -- the buffer the interrupted copy is filling.
void *g_bounce[2] = {nullptr, nullptr};
size_t g_bounce_bytes[2] = {0, 0};
void *g_page_tmp = nullptr;     // page-locked: the handler's page-wise fetches (64 spinors)
[[noreturn]] void handler_die(const char *msg) {   // async-signal-safe exit with a message
  (void)!write(2, msg, strlen(msg));
  _exit(1);
}
void *bounce(size_t bytes) {
  const int k = in_handler_here() ? 1 : 0;
  if (k == 1 && g_bounce_bytes[1] < bytes) handler_die("[tmlqcd_dropin] fatal: lazy mode: the fault handler's staging buffer is smaller than the field it has to fetch\n");
  if (g_bounce_bytes[k] < bytes) {
    if (g_bounce[k]) tmhip_pinned_free(g_bounce[k]);
    g_bounce[k] = nullptr; g_bounce_bytes[k] = 0;
    CK(tmhip_pinned_alloc(bytes, &g_bounce[k]));
    g_bounce_bytes[k] = bytes;
  }
  return g_bounce[k];
}

// Called by mirror() in lazy mode (entry-point context): whatever the fault handler will need for an array of this size exists before
// the array is ever watched -- its page-locked staging buffer, the page buffer.  The handler allocates nothing.
void prepare_handler_buffers(size_t bytes) {
  if (g_bounce_bytes[1] < bytes) {
    if (g_bounce[1]) tmhip_pinned_free(g_bounce[1]);
    g_bounce[1] = nullptr; g_bounce_bytes[1] = 0;
    CK(tmhip_pinned_alloc(bytes, &g_bounce[1]));
    g_bounce_bytes[1] = bytes;
  }
  if (!g_page_tmp) CK(tmhip_pinned_alloc(64 * sizeof(spinor), &g_page_tmp));
}

// host <-> device for a mirror of any shape (KIND_LIN: the two halves are plain prefixes, no site permutation)
void upload(tmhip_ctx *c, const void *host_user, Mirror &m) {
  const void *host = host_user;
  if (g_mode == TMLQCD_HIP_LAZY) { void *b = bounce(m.bytes); memcpy(b, host_user, m.bytes); host = b; }
  if (m.kind != KIND_LIN) { CK(tmhip_field_upload(c, m.f, host, nsites(m.kind))); return; }
  const Parts pt = parts_of(KIND_LIN, m.n);
  CK(tmhip_field_upload(c, tmhip_field_even(m.f), host, pt.cnt[0]));
  if (pt.n > 1) CK(tmhip_field_upload(c, tmhip_field_odd(m.f), (const spinor *)host + VOLUME / 2, pt.cnt[1]));
}
// Other host threads may be reading the very field that is being brought up to date (an OpenMP loop over it: one thread's fault
// triggers the fetch, the others read on).  A page must therefore never be readable before its new contents are in place: opening
// the span, then copying, lets those threads read the old data for as long as the copy takes.  The new contents are assembled in a
// private mapping nobody else knows and moved over the program's pages with mremap(MREMAP_FIXED), which swaps the pages in one step:
// a reader sees a closed page (faults, waits for the lock, runs again) or the new one.  A page the field shares with other data is
// first taken out with MREMAP_DONTUNMAP (its address stays mapped, closed and empty), completed in private and moved back.
// [host, host + bytes) lies in the pages [lo, hi); src holds its new contents; the caller has already recorded the mirror's new
// state, so page_need() gives the protection every page ends up with.  false: this memory cannot be moved (not private anonymous
// memory, or a kernel before 5.7) and nothing was changed -- the caller falls back to open-then-copy.
bool g_install_ok = true;
bool take_page(uintptr_t page, char *to) {
  if (mprotect((void *)page, g_page, PROT_NONE)) return false;
  if (mremap((void *)page, g_page, g_page, MREMAP_MAYMOVE | MREMAP_FIXED | MREMAP_DONTUNMAP, to) != (void *)to) return false;
  return mprotect(to, g_page, PROT_READ | PROT_WRITE) == 0;
}
bool move_over(char *from, uintptr_t to, size_t len) {
  if (mprotect(from, len, prot_flags(page_need(to, g_reg)))) return false;
  return mremap(from, len, len, MREMAP_MAYMOVE | MREMAP_FIXED, (void *)to) == (void *)to;
}
bool install_pages(uintptr_t host, size_t bytes, uintptr_t lo, uintptr_t hi, const char *src) {
  if (!g_install_ok) return false;
  const size_t len = hi - lo;
  char *sc = (char *)mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (sc == MAP_FAILED) return false;
  const uintptr_t last = hi - g_page;
  const bool head = host > lo, tail = host + bytes < hi && (last != lo || !head);
  bool took_head = false;
  if (head) {
    if (!take_page(lo, sc)) { g_install_ok = false; munmap(sc, len); return false; }
    took_head = true;
  }
  if (tail && !take_page(last, sc + (last - lo))) {
    g_install_ok = false;
    if (took_head) { mprotect(sc, g_page, PROT_NONE); mremap(sc, g_page, g_page, MREMAP_MAYMOVE | MREMAP_FIXED, (void *)lo); munmap(sc + g_page, len - g_page); }
    else munmap(sc, len);
    return false;
  }
  memcpy(sc + (host - lo), src, bytes);
  // up to three pieces (the two shared pages are mappings of their own by now); each move is one step for every other thread
  bool ok = true;
  uintptr_t a = lo, b = hi;
  if (head) { ok = move_over(sc, lo, g_page) && ok; a = lo + g_page; }
  if (tail) { ok = move_over(sc + (last - lo), last, g_page) && ok; b = last; }
  if (a < b) ok = move_over(sc + (a - lo), a, b - a) && ok;
  if (!ok) die("lazy mode: mremap failed half-way while bringing a host array up to date");
  return true;
}

void download(tmhip_ctx *c, const void *host_user, Mirror &m) {
  const bool watched = m.prot != P_RW;                                  // lazy mode: the span is (partly) closed
  const bool staged = g_mode == TMLQCD_HIP_LAZY || watched;
  const void *host = staged ? bounce(m.bytes) : host_user;
  const Parts pt = parts_of(m.kind == KIND_LIN ? KIND_LIN : TMHIP_FIELD_EO, m.n);
  if (staged) {
    // straight into the page-locked bounce buffer: nothing of the context's own staging is touched, so the fault handler can do this
    // on a host thread while the master thread is inside another call
    if (m.kind != KIND_LIN) {
      CK(tmhip_field_download_range(c, m.f, const_cast<void *>(host), 0, nsites(m.kind)));
    } else {
      CK(tmhip_field_download_range(c, tmhip_field_even(m.f), const_cast<void *>(host), 0, pt.cnt[0]));
      if (pt.n > 1) CK(tmhip_field_download_range(c, tmhip_field_odd(m.f), (spinor *)const_cast<void *>(host) + VOLUME / 2, 0, pt.cnt[1]));
    }
  } else if (m.kind != KIND_LIN) {
    CK(tmhip_field_download(c, m.f, const_cast<void *>(host), nsites(m.kind)));
  } else {
    CK(tmhip_field_download(c, tmhip_field_even(m.f), const_cast<void *>(host), pt.cnt[0]));
    if (pt.n > 1) CK(tmhip_field_download(c, tmhip_field_odd(m.f), (spinor *)const_cast<void *>(host) + VOLUME / 2, pt.cnt[1]));
  }
  m.host_valid = true;
  if (!watched) { if (staged) memcpy(const_cast<void *>(host_user), host, m.bytes); return; }
  m.prot = g_mode == TMLQCD_HIP_LAZY ? P_RO : P_RW;                     // both copies current: watch for host stores
  std::fill(m.page_ok.begin(), m.page_ok.end(), 0); m.faults = 0;
  const uintptr_t lo = span_lo(host_user), hi = span_hi(host_user, m.bytes);
  if (install_pages((uintptr_t)host_user, m.bytes, lo, hi, (const char *)host)) return;
  mprotect((void *)lo, hi - lo, PROT_READ | PROT_WRITE);                // (memory that cannot be moved: open, copy, close)
  memcpy(const_cast<void *>(host_user), host, m.bytes);
  apply_prot(host_user, m, g_reg);
}
// the host array of a mirror is gone (freed and not handed out again: mincore says ENOMEM for an unmapped page): nothing to bring up to date
bool host_unmapped(const void *host, const Mirror &m) {
  if (!m.bytes) return false;
  unsigned char vec;
  const uintptr_t pg = span_lo((const char *)host + m.bytes / 2);
  return mincore((void *)pg, g_page, &vec) != 0 && errno == ENOMEM;
}
bool mapping_replaced(const void *host, const Mirror &m);
// a mirror is about to go away (or to stop being watched): bring the host up to date and give it its pages back
void release_host(tmhip_ctx *c, const void *host, Mirror &m) {
  // freed by the program (and possibly mapped again for something else, which a download would overwrite): there is no host copy
  // to bring up to date
  if (g_mode == TMLQCD_HIP_LAZY && m.prot != P_RW && (host_unmapped(host, m) || mapping_replaced(host, m))) {
    m.prot = P_RW; std::fill(m.page_ok.begin(), m.page_ok.end(), 0); m.dev_valid = false; m.host_valid = true;
    return;
  }
  if (m.f && m.dev_valid && !m.host_valid) download(c, host, m);
  if (m.prot != P_RW) set_prot(host, m, P_RW, g_reg);
}

// drop the least recently used mirrors that hold nothing the host does not have
void evict_if_crowded(tmhip_ctx *c, const void *keep) {
  static bool read_env = false;
  if (!read_env) { const char *e = getenv("TMLQCD_HIP_MAX_MIRRORS"); if (e && atoi(e) > 8) g_mirror_cap = (size_t)atoi(e); read_env = true; }
  RegLock lk;
  while (g_reg.size() > g_mirror_cap) {
    const void *victim = nullptr;
    unsigned long long oldest = ~0ull;
    for (auto &kv : g_reg)
      if (kv.first != keep && kv.second.host_valid && kv.second.last_use < oldest) { oldest = kv.second.last_use; victim = kv.first; }
    if (!victim) return;   // everything else is device-only data (resident mode): keep it
    release_host(c, victim, g_reg[victim]);
    if (g_reg[victim].f) tmhip_field_free(c, g_reg[victim].f);
    g_reg.erase(victim);
    rebuild_watch();
  }
}

// Lazy mode trusts a mirror across calls because it expects to SEE every host store (write-protected pages) and every host load of
// stale data (inaccessible pages).  That breaks when the program frees the array and gets the same address back: a large calloc is
// munmap'ed and mmap'ed again (solver/solver_field.c does this per solve in the solvers this library does not replace), the new
// pages are readable and writable, and nothing faults.  So before a watched mirror is trusted, one page of its span that this
// mirror alone protects is probed with system calls that fail with EFAULT instead of raising SIGSEGV:
//   P_NONE: write(2) FROM the page must fail;   P_RO: read(2) INTO the page (of the byte it already holds) must fail.
// If the probe succeeds the mapping is not the one this library protected: the host copy is the truth, the mirror starts over.
int g_probe_pipe[2] = {-1, -1};
bool mapping_replaced(const void *host, const Mirror &m) {
  if (m.prot == P_RW || !m.bytes) return false;
  const uintptr_t base = (uintptr_t)host, first = (base + g_page - 1) & ~(g_page - 1), last = (base + m.bytes) & ~(g_page - 1);   // interior pages [first, last)
  if (first >= last) return false;                                    // the field owns no whole page: its edge pages cannot have been unmapped alone
  uintptr_t pg = first + ((last - first) / g_page / 2) * g_page;      // a page in the middle
  if (m.prot == P_NONE) {                                             // ... that the host has not fetched meanwhile (those are read-only)
    const size_t i0 = (first - span_lo(host)) / g_page, i1 = (last - span_lo(host)) / g_page;
    size_t i = (pg - span_lo(host)) / g_page;
    if (i < m.page_ok.size() && m.page_ok[i]) {
      for (i = i0; i < i1 && i < m.page_ok.size() && m.page_ok[i]; i++) {}
      if (i >= i1 || i >= m.page_ok.size()) return false;             // every interior page already fetched: nothing left to tell by
      pg = span_lo(host) + i * g_page;
    }
  }
  if (g_probe_pipe[0] < 0 && pipe2(g_probe_pipe, O_NONBLOCK | O_CLOEXEC)) die("pipe() for the lazy mode's mapping probe failed");   // (non-blocking: a probe never waits)
  if (m.prot == P_NONE) {
    if (write(g_probe_pipe[1], (const void *)pg, 1) == 1) { char b; (void)!read(g_probe_pipe[0], &b, 1); return true; }
    return false;                                                     // EFAULT: still inaccessible, still ours
  }
  const char b = *(const volatile char *)pg;                          // P_RO: readable by construction
  if (write(g_probe_pipe[1], &b, 1) != 1) return false;               // (cannot probe: trust the mirror as before)
  if (read(g_probe_pipe[0], (void *)pg, 1) == 1) return true;         // the kernel could store into the page (the same byte): not write-protected any more
  char d; (void)!read(g_probe_pipe[0], &d, 1);                        // EFAULT: take the byte back out (if the kernel left it there)
  return false;
}

// Lazy mode watches an array by taking its pages away.  That is only sound for memory the program addresses and nobody else does:
//  * NOT inside a malloc arena -- the main one ("[heap]") or a thread's (a 64 MB-aligned mapping of at most 64 MB, read-write at the
//    bottom, PROT_NONE above: glibc's HEAP_MAX_SIZE): such pages also hold the allocator's chunk headers, free() / malloc() touch them
//    while they hold the arena's lock, and a fault taken there cannot be served (the handler's own callees allocate).  glibc serves a
//    request from an arena whenever a free chunk fits, whatever M_MMAP_THRESHOLD says (a 200 KB numpy array in a process that has
//    freed a few MB) -- which is why this library does NOT touch the program's malloc settings any more (it pinned the mmap threshold
//    until round 3; blocks above glibc's 32 MB ceiling of that threshold -- tmLQCD's fields at production sizes -- are mappings of
//    their own in any case);
//  * private and anonymous ("rw-p", no file): the handler swaps pages in with mremap(MREMAP_FIXED), which would silently turn a
//    MAP_SHARED / file-backed / hugetlb / SysV segment into private memory.
// Anything else is simply not watched: it is copied on every call, as in the coherent mode.  /proc/self/maps is read when a mirror is
// made (or its array was re-mapped), in entry-point context.
uintptr_t g_heap_lo = 0;   // start of the "[heap]" mapping (the initial program break: it never moves), 1 = there is none
bool below_program_break(const void *host) { return g_heap_lo > 1 && (uintptr_t)host >= g_heap_lo && (uintptr_t)host < (uintptr_t)sbrk(0); }
const char *unwatchable(const void *host, size_t bytes) {
  const uintptr_t a = (uintptr_t)host, b = a + bytes;
  FILE *fp = fopen("/proc/self/maps", "r");
  if (!fp) return "cannot read /proc/self/maps";
  // The array may lie in SEVERAL lines: this library's own mprotect calls (a neighbouring field's read-only or closed pages) split the
  // block's mapping by protection.  So the contiguous run of private anonymous lines around it is taken as a whole, whatever their
  // permissions are at the moment; it must cover [a, b).
  const char *why = nullptr;
  char line[512];
  bool in_run = false, last_none = false;
  unsigned long first_lo = 0, covered = 0;
  const unsigned long ARENA = (unsigned long)64 << 20;     // glibc's HEAP_MAX_SIZE
  while (fgets(line, sizeof(line), fp)) {
    unsigned long lo = 0, hi = 0, off = 0, ino = 0; char perm[8] = "", dev[16] = ""; int consumed = 0;
    if (sscanf(line, "%lx-%lx %7s %lx %15s %lu %n", &lo, &hi, perm, &off, dev, &ino, &consumed) < 6) continue;
    const char *name = line + consumed;
    const bool heap = strstr(name, "[heap]") != nullptr;
    if (!g_heap_lo && heap) g_heap_lo = lo;
    const bool private_anon = perm[3] == 'p' && ino == 0 && (!name[0] || name[0] == '\n');
    if (!in_run) {
      if (a < lo || a >= hi) continue;
      in_run = true; first_lo = lo; covered = hi;
      if (heap) { why = "inside the malloc heap"; break; }
      if (perm[3] != 'p') { why = "a shared mapping"; break; }
      if (!private_anon) { why = "a file-backed or named mapping"; break; }
      last_none = !strncmp(perm, "---", 3);
      continue;
    }
    if (lo != covered || !private_anon) {       // the run ends here
      if (covered < b) why = lo != covered ? "not mapped contiguously" : "spans mappings of different kinds";
      break;
    }
    covered = hi; last_none = !strncmp(perm, "---", 3);
    if (covered - first_lo > ARENA) break;        // (longer than any arena: enough is known)
  }
  fclose(fp);
  if (!g_heap_lo) g_heap_lo = 1;   // (the heap line comes before any mmap region: if it was not seen up to the array's line, there is none)
  if (!in_run) return "not mapped";
  if (why) return why;
  if (covered < b) return "not mapped contiguously";
  // a thread's arena: 64 MB-aligned, read-write at the bottom, its PROT_NONE reserve up to the 64 MB boundary
  if (first_lo % ARENA == 0 && covered == first_lo + ARENA && last_none) return "inside a thread's malloc arena";
  return nullptr;
}

Mirror &mirror(tmhip_ctx *c, const void *host, int kind, int n = 0) {
  RegLock lk;
  const size_t bytes = (size_t)(kind == KIND_LIN ? n : nsites(kind)) * sizeof(spinor);
  const bool known = g_reg.find(host) != g_reg.end();
  bool remapped = false;
  if (!known) evict_if_crowded(c, host);
  if (g_mode == TMLQCD_HIP_LAZY) {
    // One device mirror per host byte, checked on EVERY call: an array the program now addresses from another base (the halves of
    // a full field, a block inside a field) or with another extent at the SAME base (the even half at X becomes the full field at
    // X, a prefix grows) must not leave a second, independently valid copy of some of its bytes in HBM -- e.g. Hopping_Matrix into
    // g_spinor_field[k] and [k+1], then D_psi or square_norm(., VOLUME) on the pair.
    std::vector<const void *> overlap;
    for (auto &kv : g_reg)
      if (kv.first != host && (uintptr_t)kv.first < (uintptr_t)host + bytes && (uintptr_t)host < (uintptr_t)kv.first + kv.second.bytes) overlap.push_back(kv.first);
    for (const void *o : overlap) {
      release_host(c, o, g_reg[o]);
      if (g_reg[o].f) tmhip_field_free(c, g_reg[o].f);
      g_reg.erase(o);
    }
    if (!overlap.empty()) rebuild_watch();
    if (known && mapping_replaced(host, g_reg[host])) {   // freed and re-allocated at the same address: the host copy is the truth
      remapped = true;
      Mirror &old = g_reg[host];
      old.dev_valid = false; old.host_valid = true; old.prot = P_RW; std::fill(old.page_ok.begin(), old.page_ok.end(), 0); old.faults = 0;
    }
  }
  // (the main heap may have grown over a recycled address since the array was classified: one comparison with the program break, no file)
  const bool reclassify = g_mode == TMLQCD_HIP_LAZY && known && g_reg[host].prot == P_RW && !g_reg[host].nowatch && below_program_break(host);
  Mirror &m = g_reg[host];
  if (m.f && (m.kind != kind || (kind == KIND_LIN && m.n != n))) {   // same host buffer re-used with another shape (or another prefix length)
    release_host(c, host, m);
    tmhip_field_free(c, m.f);
    m = Mirror();
  }
  bool fresh = false;
  if (!m.f) {
    CK(tmhip_field_alloc(c, kind == KIND_LIN ? TMHIP_FIELD_FULL : kind, &m.f));
    m.kind = kind; m.n = n; m.dev_valid = false; m.host_valid = true; m.bytes = bytes; m.prot = P_RW;
    fresh = true;
  }
  if (g_mode == TMLQCD_HIP_LAZY && m.prot == P_RW && (fresh || reclassify || remapped || !m.classified)) {
    // decided while the array is unwatched, and everything the fault handler will need for it is made NOW
    const char *why = unwatchable(host, bytes);
    static const bool force = getenv("TMLQCD_HIP_LAZY_FORCE_WATCH") != nullptr;     // test hook: watch it anyway, a fault on it must end loudly
    m.nowatch = why != nullptr && !force;
    m.unsafe = why != nullptr && force;
    m.classified = true;
    static const bool dbg = getenv("TMLQCD_HIP_LAZY_DEBUG") != nullptr;
    if (why && dbg) fprintf(stderr, "[tmlqcd_dropin] lazy mode: the array at %p (%zu bytes) is %s: %s\n", host, bytes, force ? "WATCHED ALTHOUGH IT SHOULD NOT BE (test hook)" : "not watched, copied per call", why);
    if (!m.nowatch) {
      prepare_handler_buffers(bytes);
      m.page_ok.assign((span_hi(host, bytes) - span_lo(host)) / g_page, 0);
    }
  }
  if (!known || fresh) rebuild_watch();
  m.last_use = ++g_tick;   // after the reset above: a mirror in use by the current call must never be the eviction victim of its sibling
  return m;
}

tmhip_field *in(tmhip_ctx *c, const void *host, int kind, int n = 0) {
  RegLock lk;   // a mirror's state changes under the lock too: the fault handler reads it on other threads
  Mirror &m = mirror(c, host, kind, n);
  const bool copy_always = g_mode == TMLQCD_HIP_COHERENT || (g_mode == TMLQCD_HIP_LAZY && m.nowatch);
  if (copy_always || !m.dev_valid) {
    if (!(m.dev_valid && !m.host_valid))   // never overwrite newer device data with a stale host copy
      upload(c, host, m);
    m.dev_valid = true;
  }
  if (g_mode == TMLQCD_HIP_LAZY && !m.nowatch && m.host_valid && m.prot == P_RW) set_prot(host, m, P_RO, g_reg);   // the mirror stays good until the host stores to the array
  return m.f;
}

tmhip_field *out(tmhip_ctx *c, const void *host, int kind, int n = 0) { return mirror(c, host, kind, n).f; }

void done(tmhip_ctx *c, const void *host) {
  RegLock lk;
  Mirror &m = g_reg[host];
  m.dev_valid = true; m.host_valid = false;
  if (g_mode == TMLQCD_HIP_COHERENT || (g_mode == TMLQCD_HIP_LAZY && m.nowatch)) {
    download(c, host, m);
    m.dev_valid = false;   // coherent mode: the host copy is the truth (it may be rewritten or its address recycled)
  } else if (g_mode == TMLQCD_HIP_LAZY) {
    set_prot(host, m, P_NONE, g_reg);   // the host's next load from the array faults and fetches what it needs
  }
}

// SIGSEGV on a protected page of a mirrored host array (lazy mode); anything else goes to the handler that was there before.
//
// What this handler may do, and why it cannot hang (round-3 review, item 5; the hang of gpurun_out/r03_mp_*.log was a fault taken
// inside malloc, on a watched page of the malloc heap, with the handler's callees then waiting for the allocator's lock):
//  * It allocates nothing itself: the table it walks is a fixed array (g_watch), every mirror's page map was sized when the mirror
//    was made, its page-locked buffers (staging buffer of the largest watched array, the 64-spinor page buffer) exist before an array is
//    first watched (prepare_handler_buffers).  A request beyond them ends the program with a message (handler_die), never a retry.
//  * It DOES enter the HIP runtime: tmhip_field_download_range = one kernel launch that writes into page-locked memory + a stream
//    synchronisation.  The runtime takes its own locks there and may allocate.  That is safe because the INTERRUPTED thread can hold
//    neither a runtime lock nor an allocator lock at the moment of the fault:
//      - the only code that ever touches a watched page is the program's own loads and stores and this library's host -> bounce
//        memcpy of an upload (which holds only the registry lock, recursive for its owner).  The HIP runtime never sees a pointer
//        into the program's arrays in this mode -- every transfer goes through the page-locked bounce buffers -- so no fault can be
//        raised from inside the runtime (the "bounce-buffer argument");
//      - no watched page holds allocator state: arrays inside a malloc arena, main or per-thread, are not watched (unwatchable());
//        an mmap'ed block's own header lies in front of the user pointer, and free() of such a block takes no arena lock.
//    ANOTHER thread may be inside the runtime or the allocator (the master thread in an entry point while an OpenMP worker faults):
//    then this handler waits for an ordinary lock whose holder is running -- a delay, not a cycle; the registry lock is the only one
//    held across, and its holder never waits for a faulting thread.
//  * TMLQCD_HIP_LAZY_FORCE_WATCH (test hook) watches an unwatchable array anyway; a fault on one of its pages is answered with a
//    message and _exit(1) before anything else is called (tests/test_gpu_lazy.py).
void lazy_fault(int sig, siginfo_t *si, void *uctx) {
  const uintptr_t addr = (uintptr_t)si->si_addr, page = addr & ~(g_page - 1);
  bool ours = false;
  // Host threads (an OpenMP loop over a stale field) may fault at the same time, and the master thread may be inside an entry point
  // that changes the registry: one at a time in here, under the registry's lock.  A fault of the thread that already is in the
  // handler would be a bug of this handler: let it crash instead of recursing.
  const bool nested = in_handler_here();
  static const bool trace = getenv("TMLQCD_HIP_LAZY_DEBUG") != nullptr && atoi(getenv("TMLQCD_HIP_LAZY_DEBUG")) > 1;
  if (trace) {   // (debugging aid, TMLQCD_HIP_LAZY_DEBUG=2: names the object the faulting instruction lives in -- dladdr is not async-signal-safe)
    Dl_info di; memset(&di, 0, sizeof(di));
    void *ip = (void *)((ucontext_t *)uctx)->uc_mcontext.gregs[REG_RIP];
    dladdr(ip, &di);
    char m[384]; const int n = snprintf(m, sizeof(m), "[lazy] fault %p %s enter, instruction %p in %s (%s)\n", si->si_addr, (((ucontext_t *)uctx)->uc_mcontext.gregs[REG_ERR] & 2) ? "store" : "load", ip, di.dli_fname ? di.dli_fname : "?", di.dli_sname ? di.dli_sname : "?"); (void)!write(2, m, (size_t)n);
  }
  if (g_ctx && si->si_code == SEGV_ACCERR && !nested) {
    RegLock lk;
    g_handler_thread.store((uintptr_t)pthread_self(), std::memory_order_relaxed);
    const bool store = (((ucontext_t *)uctx)->uc_mcontext.gregs[REG_ERR] & 2) != 0;
    for (int wk = 0; wk < g_nwatch; wk++) {
      Mirror &m = *g_watch[wk].m;
      const void *host = g_watch[wk].host;
      if (!m.bytes || page < span_lo(host) || page >= span_hi(host, m.bytes)) continue;
      ours = true;                                   // (also when another thread has opened the page in the meantime: just run again)
      if (m.prot == P_RW) continue;
      if (m.unsafe) handler_die("[tmlqcd_dropin] fatal: lazy mode: fault on a watched page of an array that must not be watched (malloc arena / shared mapping; TMLQCD_HIP_LAZY_FORCE_WATCH): ending instead of risking a deadlock\n");
      if (store) {
        g_lazy_stats[3]++;                                   // the host is about to change the array: its copy becomes the only good one
        if (!m.host_valid) download(g_ctx, host, m);
        m.dev_valid = false;
        set_prot(host, m, P_RW, g_reg);
      } else if (m.prot == P_NONE) {
        const size_t idx = (page - span_lo(host)) / g_page;
        if (m.page_ok[idx]) continue;                // (the page was closed by a neighbour's wish only)
        if (m.kind != TMHIP_FIELD_EO || ++m.faults > LAZY_PAGE_FAULTS) {
          g_lazy_stats[2]++;
          download(g_ctx, host, m);                  // the host reads on: fetch the rest in one go (both copies stay current, P_RO)
        } else {
          const uintptr_t base = (uintptr_t)host, lo = page > base ? page : base, hi = page + g_page < base + m.bytes ? page + g_page : base + m.bytes;
          const int s0 = (int)((lo - base) / sizeof(spinor)), s1 = (int)((hi - base + sizeof(spinor) - 1) / sizeof(spinor));
          void *tmp = g_page_tmp;                     // page-locked, made when the first array was watched (prepare_handler_buffers)
          if (!tmp || s1 - s0 > 64) handler_die("[tmlqcd_dropin] fatal: lazy mode: no page buffer for a page-wise fetch\n");
          if (tmhip_field_download_range(g_ctx, m.f, tmp, s0, s1 - s0)) handler_die("[tmlqcd_dropin] fatal: lazy synchronisation of a page failed\n");
          m.page_ok[idx] = 1;
          const char *from = (const char *)tmp + (lo - (base + (size_t)s0 * sizeof(spinor)));
          if (!install_pages(lo, hi - lo, page, page + g_page, from)) {
            mprotect((void *)page, g_page, PROT_READ | PROT_WRITE);
            memcpy((void *)lo, from, hi - lo);
          }
          g_lazy_stats[1]++;
        }
      }
    }
    if (ours) { g_lazy_stats[0]++; mprotect((void *)page, g_page, prot_flags(page_need(page, g_reg))); }
    g_handler_thread.store(0, std::memory_order_relaxed);
  }
  if (trace) { char m[64]; const int n = snprintf(m, sizeof(m), "[lazy] fault %p leave ours=%d\n", si->si_addr, (int)ours); (void)!write(2, m, (size_t)n); }
  if (ours) return;                                  // the faulting instruction runs again
  {
    // TMLQCD_HIP_LAZY_DEBUG=1: say what is being passed on (the program's own crash, or a bug of this handler) before the next handler sees it
    static const bool dbg = getenv("TMLQCD_HIP_LAZY_DEBUG") != nullptr;
    if (dbg) {
      char msg[256];
      const int n = snprintf(msg, sizeof(msg), "[tmlqcd_dropin] SIGSEGV at %p (si_code %d, %s) is not on a watched page of %zu mirrors%s: passed on\n", si->si_addr,
                             si->si_code, (((ucontext_t *)uctx)->uc_mcontext.gregs[REG_ERR] & 2) ? "store" : "load", g_reg.size(), nested ? ", raised inside this handler" : "");
      if (n > 0) (void)!write(2, msg, (size_t)n);
    }
  }
  if (g_old_segv.sa_flags & SA_SIGINFO) { if (g_old_segv.sa_sigaction) { g_old_segv.sa_sigaction(sig, si, uctx); return; } }
  else if (g_old_segv.sa_handler != SIG_DFL && g_old_segv.sa_handler != SIG_IGN) { g_old_segv.sa_handler(sig); return; }
  signal(SIGSEGV, SIG_DFL);                          // not ours, nobody else's: die the ordinary way when the instruction faults again
}
void install_lazy_handler() {
  if (g_handler_installed) return;
  g_page = (uintptr_t)sysconf(_SC_PAGESIZE);
  // (the program's malloc settings are left alone: arrays that cannot be watched are recognised one by one, unwatchable())
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = lazy_fault;
  sa.sa_flags = SA_SIGINFO | SA_NODEFER;
  sigemptyset(&sa.sa_mask);
  if (sigaction(SIGSEGV, &sa, &g_old_segv)) die("cannot install the SIGSEGV handler of the lazy residency mode");
  g_handler_installed = true;
}

tmhip_field *half(tmhip_field *f, int kind, int par) {
  if (kind == TMHIP_FIELD_EO) return f;
  return par ? tmhip_field_odd(f) : tmhip_field_even(f);
}

}  // namespace

extern "C" {

// ------------------------------------------------------------------ residency control
void tmlqcd_hip_set_device(int device) { g_device = device; }
void tmlqcd_hip_lazy_stats(unsigned long out[4]) { for (int k = 0; k < 4; k++) out[k] = g_lazy_stats[k]; }
void tmlqcd_hip_set_residency(int mode) {
  if (mode != TMLQCD_HIP_COHERENT && mode != TMLQCD_HIP_RESIDENT && mode != TMLQCD_HIP_LAZY) die("tmlqcd_hip_set_residency: unknown mode");
  RegLock lk;
  if (mode == TMLQCD_HIP_COHERENT && g_mode == TMLQCD_HIP_RESIDENT) tmlqcd_hip_sync_all_to_host();
  if (g_mode == TMLQCD_HIP_LAZY && mode != TMLQCD_HIP_LAZY)        // leaving lazy mode: every host array current and unwatched again
    for (auto &kv : g_reg) release_host(ctx(), kv.first, kv.second);
  if (mode == TMLQCD_HIP_LAZY) install_lazy_handler();
  // whenever the host copy is current it is authoritative: a mirror left over from an earlier call may belong to a
  // host array that has since been rewritten, or to a freed one whose address was recycled
  for (auto &kv : g_reg) if (kv.second.host_valid) kv.second.dev_valid = false;
  g_mode = mode;
}
void tmlqcd_hip_sync_to_host(spinor *field) {
  RegLock lk;
  auto it = g_reg.find(field);
  if (it == g_reg.end() || !it->second.f) return;
  if (it->second.dev_valid && !it->second.host_valid) download(ctx(), field, it->second);
}
void tmlqcd_hip_sync_all_to_host(void) {
  RegLock lk;
  for (auto &kv : g_reg)
    if (kv.second.f && kv.second.dev_valid && !kv.second.host_valid) download(ctx(), kv.first, kv.second);
}
void tmlqcd_hip_host_modified(spinor *field) {
  RegLock lk;
  auto it = g_reg.find(field);
  if (it != g_reg.end()) {
    it->second.dev_valid = false; it->second.host_valid = true;
    if (it->second.prot != P_RW) set_prot(field, it->second, P_RW, g_reg);
  }
}
void tmlqcd_hip_forget(spinor *field) {
  RegLock lk;
  auto it = g_reg.find(field);
  if (it == g_reg.end()) return;
  if (it->second.prot != P_RW) { it->second.host_valid = true; set_prot(field, it->second, P_RW, g_reg); }   // (the array is being freed: nothing to fetch)
  if (it->second.f) tmhip_field_free(g_ctx, it->second.f);
  g_reg.erase(it);
  rebuild_watch();
}
void tmlqcd_hip_comm_init(const char unique_id[128]) { CK(tmhip_comm_init(ctx(), unique_id)); }
void tmlqcd_hip_comm_init_shm(const char *job) { CK(tmhip_comm_init_shm(ctx(), job)); }
int tmlqcd_hip_comm_init_ipc(void) { return tmhip_comm_init_ipc(ctx()); }   // (non-zero: the faces stay on the communicator, on every rank -- not fatal)
void tmlqcd_hip_finalize(void) {
  if (!g_ctx) return;
  RegLock lk;
  tmlqcd_hip_sync_all_to_host();
  for (auto &kv : g_reg) { if (kv.second.prot != P_RW) set_prot(kv.first, kv.second, P_RW, g_reg); if (kv.second.f) tmhip_field_free(g_ctx, kv.second.f); }
  g_reg.clear();
  g_nwatch = 0;
  if (g_full_tmp) { tmhip_field_free(g_ctx, g_full_tmp); g_full_tmp = nullptr; }
  for (int k = 0; k < 3; k++) if (g_f32[k]) { tmhip_field_free(g_ctx, g_f32[k]); g_f32[k] = nullptr; }
  for (int k = 0; k < 2; k++) if (g_bounce[k]) { tmhip_pinned_free(g_bounce[k]); g_bounce[k] = nullptr; g_bounce_bytes[k] = 0; }
  if (g_page_tmp) { tmhip_pinned_free(g_page_tmp); g_page_tmp = nullptr; }
  tmhip_destroy(g_ctx);
  g_ctx = nullptr;
  g_gauge_uploaded = false;
  g_clover_uploaded = false;
  g_dev_links_newer = g_momenta_resident = false;
}

// ------------------------------------------------------------------ stencil
/* operator/Hopping_Matrix.c:131-156 */
void Hopping_Matrix(const int ieo, spinor *const l, spinor *const k) {
  tmhip_ctx *c = refresh(true);
  tmhip_field *fk = in(c, k, TMHIP_FIELD_EO), *fl = out(c, l, TMHIP_FIELD_EO);
  CK(tmhip_hopping_matrix(c, ieo, fl, fk));
  done(c, l);
}
/* operator/Hopping_Matrix_nocom.c:48-56 */
void Hopping_Matrix_nocom(const int ieo, spinor *const l, spinor *const k) {
  tmhip_ctx *c = refresh(true);
  tmhip_field *fk = in(c, k, TMHIP_FIELD_EO), *fl = out(c, l, TMHIP_FIELD_EO);
  CK(tmhip_hopping_matrix_nocom(c, ieo, fl, fk));
  done(c, l);
}
/* operator/tm_times_Hopping_Matrix.c:72-153 */
void tm_times_Hopping_Matrix(const int ieo, spinor *const l, spinor *const k, TM_COMPLEX const cfactor) {
  tmhip_ctx *c = refresh(true);
  tmhip_field *fk = in(c, k, TMHIP_FIELD_EO), *fl = out(c, l, TMHIP_FIELD_EO);
  CK(tmhip_tm_times_hopping_matrix(c, ieo, fl, fk, __real__ cfactor, __imag__ cfactor));
  done(c, l);
}
/* operator/tm_sub_Hopping_Matrix.c:73-157 */
void tm_sub_Hopping_Matrix(const int ieo, spinor *const l, spinor *p, spinor *const k, TM_COMPLEX const cfactor) {
  tmhip_ctx *c = refresh(true);
  tmhip_field *fk = in(c, k, TMHIP_FIELD_EO), *fp = in(c, p, TMHIP_FIELD_EO), *fl = out(c, l, TMHIP_FIELD_EO);
  CK(tmhip_tm_sub_hopping_matrix(c, ieo, fl, fp, fk, __real__ cfactor, __imag__ cfactor));
  done(c, l);
}
/* D_psi_body.c:314-316: with g_c_sw > 0 the site term of D_psi is the clover one, (1 + T(x) + i mu g5) from the host's sw array.
 * On the two parities of a full field that is Msw_full (clovertm_operators.c:96-110): new = (1 + T + i mu g5) own - H other. */
extern double g_c_sw __attribute__((weak));   /* global.h:198; a host program without it has no clover term */
static void ensure_clover(tmhip_ctx *c);
static void d_psi_core(tmhip_ctx *c, tmhip_field *fp, tmhip_field *fq) {
  if (&g_c_sw && g_c_sw > 0.) {
    ensure_clover(c);
    CK(tmhip_Msw_full(c, tmhip_field_even(fp), tmhip_field_odd(fp), tmhip_field_even(fq), tmhip_field_odd(fq)));
  } else {
    CK(tmhip_D_psi(c, fp, fq));
  }
}
/* operator/D_psi.c:1133-1140 -> D_psi_body.c:266-375 */
void D_psi(spinor *const P, spinor *const Q) {
  if (P == Q) {   /* D_psi_body.c:267-272 */
    printf("Error in D_psi (operator.c):\nArguments must be different spinor fields\nProgram aborted\n");
    exit(1);
  }
  tmhip_ctx *c = refresh(true);
  tmhip_field *fq = in(c, Q, TMHIP_FIELD_FULL), *fp = out(c, P, TMHIP_FIELD_FULL);
  d_psi_core(c, fp, fq);
  done(c, P);
}

// ------------------------------------------------------------------ e/o operators (tm_operators.c)
#define EO_OP(NAME, CORE)                                                                  \
  void NAME(spinor *const l, spinor *const k) {                                            \
    tmhip_ctx *c = refresh(true);                                                          \
    tmhip_field *fk = in(c, k, TMHIP_FIELD_EO), *fl = out(c, l, TMHIP_FIELD_EO);           \
    CK(CORE(c, fl, fk));                                                                   \
    done(c, l);                                                                            \
  }
#define EO_OP_CLOVER(NAME, CORE)                                                           \
  void NAME(spinor *const l, spinor *const k) {                                            \
    tmhip_ctx *c = refresh_clover();                                                       \
    tmhip_field *fk = in(c, k, TMHIP_FIELD_EO), *fl = out(c, l, TMHIP_FIELD_EO);           \
    CK(CORE(c, fl, fk));                                                                   \
    done(c, l);                                                                            \
  }
EO_OP(Qtm_plus_psi, tmhip_Qtm_plus_psi)        /* tm_operators.c:172-177 */
EO_OP(Qtm_minus_psi, tmhip_Qtm_minus_psi)      /* tm_operators.c:216-221 */
EO_OP(Mtm_plus_psi, tmhip_Mtm_plus_psi)        /* tm_operators.c:245-250 */
EO_OP(Mtm_minus_psi, tmhip_Mtm_minus_psi)      /* tm_operators.c:289-294 */
EO_OP(Qtm_pm_psi, tmhip_Qtm_pm_psi)            /* tm_operators.c:338-345 */
EO_OP(Qtm_plus_sym_psi, tmhip_Qtm_plus_sym_psi)            /* tm_operators.c:186-192 */
EO_OP(Qtm_minus_sym_psi, tmhip_Qtm_minus_sym_psi)          /* tm_operators.c:223-229 */
EO_OP(Mtm_plus_sym_psi, tmhip_Mtm_plus_sym_psi)            /* tm_operators.c:259-265 */
EO_OP(Mtm_minus_sym_psi, tmhip_Mtm_minus_sym_psi)          /* tm_operators.c:296-302 */
EO_OP(Mtm_plus_sym_dagg_psi, tmhip_Mtm_plus_sym_dagg_psi)  /* tm_operators.c:312-322 */
EO_OP(Qtm_pm_sym_psi, tmhip_Qtm_pm_sym_psi)                /* tm_operators.c:347-364 */
void Qtm_plus_sym_psi_nocom(spinor *const l, spinor *const k) { Qtm_plus_sym_psi(l, k); }   /* :194-200 */
void Mtm_plus_sym_psi_nocom(spinor *const l, spinor *const k) { Mtm_plus_sym_psi(l, k); }   /* :267-273 */
void Mtm_minus_sym_psi_nocom(spinor *const l, spinor *const k) { Mtm_minus_sym_psi(l, k); } /* :304-310 */
/* The _nocom variants differ from the above only by skipping the halo exchange
 * (tm_operators.c:179-184,252-257,369-379); on one GPU they are the same function. */
void Qtm_plus_psi_nocom(spinor *const l, spinor *const k) { Qtm_plus_psi(l, k); }
void Mtm_plus_psi_nocom(spinor *const l, spinor *const k) { Mtm_plus_psi(l, k); }
void Qtm_pm_psi_nocom(spinor *const l, spinor *const k) { Qtm_pm_psi(l, k); }

/* tm_operators.c:508-526 */
void H_eo_tm_inv_psi(spinor *const l, spinor *const k, const int ieo, const double sign) {
  tmhip_ctx *c = refresh(true);
  tmhip_field *fk = in(c, k, TMHIP_FIELD_EO), *fl = out(c, l, TMHIP_FIELD_EO);
  CK(tmhip_H_eo_tm_inv_psi(c, fl, fk, ieo, sign));
  done(c, l);
}
/* tm_operators.c:117-128 */
void M_full(spinor *const En, spinor *const On, spinor *const E, spinor *const O) {
  tmhip_ctx *c = refresh(true);
  tmhip_field *fe = in(c, E, TMHIP_FIELD_EO), *fo = in(c, O, TMHIP_FIELD_EO);
  tmhip_field *fen = out(c, En, TMHIP_FIELD_EO), *fon = out(c, On, TMHIP_FIELD_EO);
  CK(tmhip_M_full(c, fen, fon, fe, fo));
  done(c, En); done(c, On);
}
/* tm_operators.c:130-143 */
void Q_full(spinor *const En, spinor *const On, spinor *const E, spinor *const O) {
  tmhip_ctx *c = refresh(true);
  tmhip_field *fe = in(c, E, TMHIP_FIELD_EO), *fo = in(c, O, TMHIP_FIELD_EO);
  tmhip_field *fen = out(c, En, TMHIP_FIELD_EO), *fon = out(c, On, TMHIP_FIELD_EO);
  CK(tmhip_M_full(c, fen, fon, fe, fo));
  CK(tmhip_gamma5(c, fen, fen, VOLUME / 2));
  CK(tmhip_gamma5(c, fon, fon, VOLUME / 2));
  done(c, En); done(c, On);
}
/* tm_operators.c:145-155 */
void M_minus_1_timesC(spinor *const En, spinor *const On, spinor *const E, spinor *const O) {
  tmhip_ctx *c = refresh(true);
  tmhip_field *fe = in(c, E, TMHIP_FIELD_EO), *fo = in(c, O, TMHIP_FIELD_EO);
  tmhip_field *fen = out(c, En, TMHIP_FIELD_EO), *fon = out(c, On, TMHIP_FIELD_EO);
  CK(tmhip_H_eo_tm_inv_psi(c, fen, fo, TMHIP_EO, +1.));
  CK(tmhip_H_eo_tm_inv_psi(c, fon, fe, TMHIP_OE, +1.));
  done(c, En); done(c, On);
}

// ------------------------------------------------------------------ clover twisted mass
static void ensure_clover(tmhip_ctx *c) {
  if (!g_clover_uploaded) {
    if (!&sw || !&sw_inv || !sw || !sw_inv) die("clover operator called but the host program has no sw / sw_inv (init_sw_fields)");
    CK(tmhip_set_clover(c, &sw[0][0][0], &sw_inv[0][0][0]));
    g_clover_uploaded = true;
  }
}
static tmhip_ctx *refresh_clover() {
  tmhip_ctx *c = refresh(true);
  ensure_clover(c);
  return c;
}
void tmlqcd_hip_update_clover(void) { g_clover_uploaded = false; }
void tmlqcd_hip_set_max_mirrors(int n) { if (n >= 8) g_mirror_cap = (size_t)n; }
unsigned long tmlqcd_hip_calls(void) { return g_calls; }
/* sw_term(g_gauge_field, kappa, c_sw) (operator/clover_term.c:88) computed in HBM; the host's sw array, if the program
 * has one (init_sw_fields), receives a copy so that host-side consumers (sw_trace, sw_deriv ...) keep working. */
void tmlqcd_hip_sw_term(const double kappa, const double c_sw) {
  tmhip_ctx *c = refresh(false);
  CK(tmhip_sw_term(c, &g_gauge_field[0][0], kappa, c_sw));
  if (&sw && sw) CK(tmhip_get_clover(c, &sw[0][0][0], nullptr));
  g_clover_uploaded = false;
}
/* sw_invert(ieo, mu) (operator/clover_invert.c:170) from the device-resident clover term */
void tmlqcd_hip_sw_invert(const int ieo, const double mu) {
  tmhip_ctx *c = refresh(false);
  CK(tmhip_sw_invert(c, ieo, mu));
  if (&sw_inv && sw_inv) CK(tmhip_get_clover(c, nullptr, &sw_inv[0][0][0]));
  g_clover_uploaded = true;   // the device copy is the fresh one
}
EO_OP_CLOVER(Qsw_pm_psi, tmhip_Qsw_pm_psi)      /* clovertm_operators.c:233-245 */
EO_OP_CLOVER(Msw_plus_psi, tmhip_Msw_plus_psi)  /* clovertm_operators.c:256-261 */
EO_OP_CLOVER(Qsw_psi, tmhip_Qsw_psi)              /* :201-206 */
EO_OP_CLOVER(Qsw_minus_psi, tmhip_Qsw_minus_psi)  /* :209-214 */
EO_OP_CLOVER(Qsw_plus_psi, tmhip_Qsw_plus_psi)    /* :217-222 */
EO_OP_CLOVER(Qsw_sq_psi, tmhip_Qsw_sq_psi)        /* :225-237 */
EO_OP_CLOVER(Msw_psi, tmhip_Msw_psi)              /* :247-252 */
EO_OP_CLOVER(Msw_minus_psi, tmhip_Msw_minus_psi)  /* :261-266 */
/* clovertm_operators.c:96-110 */
void Msw_full(spinor *const En, spinor *const On, spinor *const E, spinor *const O) {
  tmhip_ctx *c = refresh_clover();
  tmhip_field *fe = in(c, E, TMHIP_FIELD_EO), *fo = in(c, O, TMHIP_FIELD_EO);
  tmhip_field *fen = out(c, En, TMHIP_FIELD_EO), *fon = out(c, On, TMHIP_FIELD_EO);
  CK(tmhip_Msw_full(c, fen, fon, fe, fo));
  done(c, En); done(c, On);
}
/* operator/assign_mul_one_sw_pm_imu_inv_block_body.c:1-72 */
void assign_mul_one_sw_pm_imu(const int ieo, spinor *const k, spinor *const l, const double mu) {
  tmhip_ctx *c = refresh_clover();
  tmhip_field *fl = in(c, l, TMHIP_FIELD_EO), *fk = out(c, k, TMHIP_FIELD_EO);
  CK(tmhip_assign_mul_one_sw_pm_imu(c, ieo, fk, fl, mu));
  done(c, k);
}
/* operator/assign_mul_one_sw_pm_imu_inv_block_body.c:143-196 (ieo and mu are not looked at, as in the reference) */
void assign_mul_one_sw_pm_imu_inv(const int ieo, spinor *const k, spinor *const l, const double mu) {
  tmhip_ctx *c = refresh_clover();
  tmhip_field *fl = in(c, l, TMHIP_FIELD_EO), *fk = out(c, k, TMHIP_FIELD_EO);
  CK(tmhip_assign_mul_one_sw_pm_imu_inv(c, ieo, fk, fl, mu));
  done(c, k);
}
/* clovertm_operators.c:873-940, 1098-1140: the even-site forms of the two above */
void Mee_sw_psi(spinor *const k, spinor *const l, const double mu) { assign_mul_one_sw_pm_imu(0, k, l, mu); }
void Mee_sw_inv_psi(spinor *const k, spinor *const l, const double mu) { assign_mul_one_sw_pm_imu_inv(0, k, l, mu); }
/* clovertm_operators.c:268-272 */
void H_eo_sw_inv_psi(spinor *const l, spinor *const k, const int ieo, const int tau3sign, const double mu) {
  tmhip_ctx *c = refresh_clover();
  tmhip_field *fk = in(c, k, TMHIP_FIELD_EO), *fl = out(c, l, TMHIP_FIELD_EO);
  CK(tmhip_H_eo_sw_inv_psi(c, fl, fk, ieo, tau3sign, mu));
  done(c, l);
}
/* clovertm_operators.c:287-350 (in place) */
void clover_inv(spinor *const l, const int tau3sign, const double mu) {
  tmhip_ctx *c = refresh_clover();
  tmhip_field *fl = in(c, l, TMHIP_FIELD_EO);
  CK(tmhip_clover_inv(c, fl, tau3sign, mu));
  done(c, l);
}
/* clovertm_operators.c:448-520 */
void clover_gamma5(const int ieo, spinor *const l, const spinor *const k, const spinor *const j, const double mu) {
  tmhip_ctx *c = refresh_clover();
  tmhip_field *fk = in(c, k, TMHIP_FIELD_EO), *fj = in(c, j, TMHIP_FIELD_EO), *fl = out(c, l, TMHIP_FIELD_EO);
  CK(tmhip_clover_gamma5(c, ieo, fl, fk, fj, mu));
  done(c, l);
}
/* clovertm_operators.c:535-600 */
void clover(const int ieo, spinor *const l, const spinor *const k, const spinor *const j, const double mu) {
  tmhip_ctx *c = refresh_clover();
  tmhip_field *fk = in(c, k, TMHIP_FIELD_EO), *fj = in(c, j, TMHIP_FIELD_EO), *fl = out(c, l, TMHIP_FIELD_EO);
  CK(tmhip_clover(c, ieo, fl, fk, fj, mu));
  done(c, l);
}

// ------------------------------------------------------------------ site-diagonal ops
/* mul_one_pm_imu_inv_body.c:1-41 */
void mul_one_pm_imu_inv(spinor *const l, const double _sign, const int N) {
  tmhip_ctx *c = refresh(false);
  if (N == 0) return;   /* an empty loop in the reference */
  const int kind = kind_of_N(N);
  const Parts pt = parts_of(kind, N);
  tmhip_field *fl = in(c, l, kind, N);
  for (int p = 0; p < pt.n; p++) CK(tmhip_mul_one_pm_imu_inv(c, half(fl, kind, p), _sign, pt.cnt[p]));
  done(c, l);
}
/* mul_one_pm_imu_inv_body.c:43-80 */
void assign_mul_one_pm_imu_inv(spinor *const l, spinor *const k, const double _sign, const int N) {
  tmhip_ctx *c = refresh(false);
  if (N == 0) return;   /* an empty loop in the reference */
  const int kind = kind_of_N(N);
  const Parts pt = parts_of(kind, N);
  tmhip_field *fk = in(c, k, kind, N), *fl = out(c, l, kind, N);
  for (int p = 0; p < pt.n; p++) CK(tmhip_assign_mul_one_pm_imu_inv(c, half(fl, kind, p), half(fk, kind, p), _sign, pt.cnt[p]));
  done(c, l);
}
/* tm_operators.c:669-720 */
void assign_mul_one_pm_imu(spinor *const l, spinor *const k, const double _sign, const int N) {
  tmhip_ctx *c = refresh(false);
  if (N == 0) return;   /* an empty loop in the reference */
  const int kind = kind_of_N(N);
  const Parts pt = parts_of(kind, N);
  tmhip_field *fk = in(c, k, kind, N), *fl = out(c, l, kind, N);
  for (int p = 0; p < pt.n; p++) CK(tmhip_assign_mul_one_pm_imu(c, half(fl, kind, p), half(fk, kind, p), _sign, pt.cnt[p]));
  done(c, l);
}
/* tm_operators.c:627-667 */
void mul_one_pm_imu(spinor *const l, const double _sign) {
  tmhip_ctx *c = refresh(false);
  tmhip_field *fl = in(c, l, TMHIP_FIELD_EO);
  CK(tmhip_mul_one_pm_imu(c, fl, _sign));
  done(c, l);
}
/* mul_one_pm_imu_sub_mul_body.c:1-48 */
void mul_one_pm_imu_sub_mul(spinor *const l, spinor *const k, spinor *const j, const double _sign, const int N) {
  tmhip_ctx *c = refresh(false);
  if (N == 0) return;   /* an empty loop in the reference */
  const int kind = kind_of_N(N);
  const Parts pt = parts_of(kind, N);
  tmhip_field *fk = in(c, k, kind, N), *fj = in(c, j, kind, N), *fl = out(c, l, kind, N);
  for (int p = 0; p < pt.n; p++)
    CK(tmhip_mul_one_pm_imu_sub_mul(c, half(fl, kind, p), half(fk, kind, p), half(fj, kind, p), _sign, pt.cnt[p]));
  done(c, l);
}
/* tm_operators.c:813-858 */
void mul_one_pm_imu_sub_mul_gamma5(spinor *const l, spinor *const k, spinor *const j, const double _sign) {
  tmhip_ctx *c = refresh(false);
  tmhip_field *fk = in(c, k, TMHIP_FIELD_EO), *fj = in(c, j, TMHIP_FIELD_EO), *fl = out(c, l, TMHIP_FIELD_EO);
  CK(tmhip_mul_one_pm_imu_sub_mul_gamma5(c, fl, fk, fj, _sign));
  done(c, l);
}
/* tm_operators.c:781-810 (external linkage in the reference although no header declares it) */
void mul_one_sub_mul_gamma5(spinor *const l, spinor *const k, spinor *const j) {
  tmhip_ctx *c = refresh(false);
  tmhip_field *fk = in(c, k, TMHIP_FIELD_EO), *fj = in(c, j, TMHIP_FIELD_EO), *fl = out(c, l, TMHIP_FIELD_EO);
  CK(tmhip_mul_one_sub_mul_gamma5(c, fl, fk, fj));
  done(c, l);
}
/* tm_operators.c:723-775: l = (1 + i mu g5) k with an explicit mu */
void Mee_psi(spinor *const l, spinor *const k, const double mu) {
  tmhip_ctx *c = refresh(false);
  tmhip_field *fk = in(c, k, TMHIP_FIELD_EO), *fl = out(c, l, TMHIP_FIELD_EO);
  CK(tmhip_set_mu(c, mu));
  CK(tmhip_assign_mul_one_pm_imu(c, fl, fk, +1., VOLUME / 2));
  CK(tmhip_set_mu(c, g_mu));
  done(c, l);
}
/* tm_operators.c:587-625: l = (1 - i mu g5)/(1+mu^2) k with an explicit mu */
void Mee_inv_psi(spinor *const l, spinor *const k, const double mu) {
  tmhip_ctx *c = refresh(false);
  tmhip_field *fk = in(c, k, TMHIP_FIELD_EO), *fl = out(c, l, TMHIP_FIELD_EO);
  CK(tmhip_set_mu(c, mu));
  CK(tmhip_assign_mul_one_pm_imu_inv(c, fl, fk, +1., VOLUME / 2));
  CK(tmhip_set_mu(c, g_mu));
  done(c, l);
}
/* gamma.c:77-98 */
void gamma5(spinor *const l, spinor *const k, const int V) {
  tmhip_ctx *c = refresh(false);
  if (V == 0) return;   /* an empty loop in the reference */
  const int kind = kind_of_N(V);
  const Parts pt = parts_of(kind, V);
  tmhip_field *fk = in(c, k, kind, V), *fl = out(c, l, kind, V);
  for (int p = 0; p < pt.n; p++) CK(tmhip_gamma5(c, half(fl, kind, p), half(fk, kind, p), pt.cnt[p]));
  done(c, l);
}

// ------------------------------------------------------------------ full-lattice operators
/* The reference toggles g_mu's sign around D_psi (tm_operators.c:380-492); refresh() re-reads it. */
static tmhip_field *full_tmp(tmhip_ctx *c) {
  if (!g_full_tmp) CK(tmhip_field_alloc(c, TMHIP_FIELD_FULL, &g_full_tmp));
  return g_full_tmp;
}
static void g5_full(tmhip_ctx *c, tmhip_field *l, tmhip_field *k) {
  CK(tmhip_gamma5(c, tmhip_field_even(l), tmhip_field_even(k), VOLUME / 2));
  CK(tmhip_gamma5(c, tmhip_field_odd(l), tmhip_field_odd(k), VOLUME / 2));
}
/* tm_operators.c:111-114 */
void Q_psi(spinor *const P, spinor *const Q) {
  tmhip_ctx *c = refresh(true);
  tmhip_field *fq = in(c, Q, TMHIP_FIELD_FULL), *fp = out(c, P, TMHIP_FIELD_FULL);
  d_psi_core(c, fp, fq); g5_full(c, fp, fp);
  done(c, P);
}
/* tm_operators.c:486-490 */
void Q_plus_psi(spinor *const l, spinor *const k) { Q_psi(l, k); }
/* tm_operators.c:460-466 */
void Q_minus_psi(spinor *const l, spinor *const k) {
  tmhip_ctx *c = refresh(true);
  tmhip_field *fk = in(c, k, TMHIP_FIELD_FULL), *fl = out(c, l, TMHIP_FIELD_FULL);
  CK(tmhip_set_mu(c, -g_mu)); d_psi_core(c, fl, fk); CK(tmhip_set_mu(c, g_mu));
  g5_full(c, fl, fl);
  done(c, l);
}
/* tm_operators.c:468-473 */
void M_minus_psi(spinor *const l, spinor *const k) {
  tmhip_ctx *c = refresh(true);
  tmhip_field *fk = in(c, k, TMHIP_FIELD_FULL), *fl = out(c, l, TMHIP_FIELD_FULL);
  CK(tmhip_set_mu(c, -g_mu)); d_psi_core(c, fl, fk); CK(tmhip_set_mu(c, g_mu));
  done(c, l);
}
/* tm_operators.c:380-388 : Q_+ Q_- on the full lattice */
void Q_pm_psi(spinor *const l, spinor *const k) {
  tmhip_ctx *c = refresh(true);
  tmhip_field *fk = in(c, k, TMHIP_FIELD_FULL), *fl = out(c, l, TMHIP_FIELD_FULL), *tmp = full_tmp(c);
  CK(tmhip_set_mu(c, -g_mu)); d_psi_core(c, fl, fk);
  g5_full(c, tmp, fl);
  CK(tmhip_set_mu(c, g_mu)); d_psi_core(c, fl, tmp);
  g5_full(c, fl, fl);
  done(c, l);
}
/* tm_operators.c:453-461 : Q_pm_psi with the first twist scaled by 10 (mu -> -10 mu, then +mu) */
void Q_pm_psi2(spinor *const l, spinor *const k) {
  tmhip_ctx *c = refresh(true);
  tmhip_field *fk = in(c, k, TMHIP_FIELD_FULL), *fl = out(c, l, TMHIP_FIELD_FULL), *tmp = full_tmp(c);
  CK(tmhip_set_mu(c, -10. * g_mu)); d_psi_core(c, fl, fk);
  g5_full(c, tmp, fl);
  CK(tmhip_set_mu(c, g_mu)); d_psi_core(c, fl, tmp);
  g5_full(c, fl, fl);
  done(c, l);
}
/* tm_operators.c:402-436 : Q_pm_psi with spinorPrecondition() (solver/dirac_operator_eigenvectors.c, FFTW-based, stays in
 * tmLQCD) applied before, between and after the two D_psi where g_prec_sequence_d_dagger_d[] is non-zero.  Host-level
 * composition over this library's own assign / D_psi / gamma5; the preconditioner and its globals are weak references, so a host
 * program that does not link them still resolves this symbol and gets the unpreconditioned sequence. */
extern void *g_precWS __attribute__((weak));                       /* global.h:267 */
extern double g_prec_sequence_d_dagger_d[3] __attribute__((weak)); /* solver/dirac_operator_eigenvectors.h:67 */
extern int L __attribute__((weak));                                /* global.h:82 */
void spinorPrecondition(spinor *spinor_out, const spinor *spinor_in, void *ws, int tt, int ll, const TM_COMPLEX alpha,
                        unsigned int dagger, unsigned int autofft) __attribute__((weak));
void Q_pm_psi_prec(spinor *const l, spinor *const k) {
  double seq[3] = {0., 0., 0.};
  if (g_prec_sequence_d_dagger_d) for (int i = 0; i < 3; i++) seq[i] = g_prec_sequence_d_dagger_d[i];
  const bool any = seq[0] != 0. || seq[1] != 0. || seq[2] != 0.;
  if (any && (!spinorPrecondition || !&g_precWS || !&L))
    die("Q_pm_psi_prec: g_prec_sequence_d_dagger_d is set but spinorPrecondition (solver/dirac_operator_eigenvectors.c) is not linked");
  static spinor *tmp = nullptr;
  static int tmp_sites = 0;
  if (tmp_sites < VOLUMEPLUSRAND) {
    free(tmp);
    tmp = (spinor *)malloc((size_t)VOLUMEPLUSRAND * sizeof(spinor));
    if (!tmp) die("Q_pm_psi_prec: out of host memory");
    tmp_sites = VOLUMEPLUSRAND;
  }
  auto prec = [&](spinor *out_, const spinor *in_, double a) {
    TM_COMPLEX alpha = a;
    spinorPrecondition(out_, in_, g_precWS, T, L, alpha, 0, 1);
    tmlqcd_hip_host_modified(out_);
  };
  if (seq[0] != 0.) { tmlqcd_hip_sync_to_host(k); prec(l, k, seq[0]); } else assign(l, k, VOLUME);
  g_mu = -g_mu;
  D_psi(tmp, l);
  gamma5(l, tmp, VOLUME);
  g_mu = -g_mu;
  if (seq[1] != 0.) { tmlqcd_hip_sync_to_host(l); prec(l, l, seq[1]); }
  D_psi(tmp, l);
  gamma5(l, tmp, VOLUME);
  if (seq[2] != 0.) { tmlqcd_hip_sync_to_host(l); prec(l, l, seq[2]); }
}
/* tm_operators.c:440-449 : "version for the gpu", gamma5 applied to the INPUT in place first, none at the end */
void Q_pm_psi_gpu(spinor *const l, spinor *const k) {
  tmhip_ctx *c = refresh(true);
  tmhip_field *fk = in(c, k, TMHIP_FIELD_FULL), *fl = out(c, l, TMHIP_FIELD_FULL), *tmp = full_tmp(c);
  g5_full(c, fk, fk);
  CK(tmhip_set_mu(c, -g_mu)); d_psi_core(c, fl, fk);
  g5_full(c, tmp, fl);
  CK(tmhip_set_mu(c, g_mu)); d_psi_core(c, fl, tmp);
  done(c, k); done(c, l);
}
/* tm_operators.c:476-483 */
void Q_minus_psi_gpu(spinor *const l, spinor *const k) {
  tmhip_ctx *c = refresh(true);
  tmhip_field *fk = in(c, k, TMHIP_FIELD_FULL), *fl = out(c, l, TMHIP_FIELD_FULL);
  g5_full(c, fk, fk);
  CK(tmhip_set_mu(c, -g_mu)); d_psi_core(c, fl, fk); CK(tmhip_set_mu(c, g_mu));
  g5_full(c, fl, fl);
  done(c, k); done(c, l);
}
/* tm_operators.c:390-397 */
void D_dagg_psi(spinor *const l, spinor *const k) {
  tmhip_ctx *c = refresh(true);
  tmhip_field *fk = in(c, k, TMHIP_FIELD_FULL), *fl = out(c, l, TMHIP_FIELD_FULL), *tmp = full_tmp(c);
  g5_full(c, fl, fk);
  CK(tmhip_set_mu(c, -g_mu)); d_psi_core(c, tmp, fl); CK(tmhip_set_mu(c, g_mu));
  g5_full(c, fl, tmp);
  done(c, l);
}

// ------------------------------------------------------------------ linalg
/* linalg/square_norm.c:253-320 */
double square_norm(const spinor *const P, const int N, const int parallel) {
  tmhip_ctx *c = refresh(false);
  if (N == 0) return 0.;   /* an empty loop in the reference */
  const int kind = kind_of_N(N);
  const Parts pt = parts_of(kind, N);
  tmhip_field *fp = in(c, P, kind, N);
  double res = 0, r;
  for (int p = 0; p < pt.n; p++) { CK(tmhip_square_norm(c, half(fp, kind, p), pt.cnt[p], parallel, &r)); res += r; }
  return res;
}
/* linalg/scalar_prod_r.c:135-197 */
double scalar_prod_r(const spinor *const S, const spinor *const R, const int N, const int parallel) {
  tmhip_ctx *c = refresh(false);
  if (N == 0) return 0.;   /* an empty loop in the reference */
  const int kind = kind_of_N(N);
  const Parts pt = parts_of(kind, N);
  tmhip_field *fs = in(c, S, kind, N), *fr = in(c, R, kind, N);
  double res = 0, r;
  for (int p = 0; p < pt.n; p++) { CK(tmhip_scalar_prod_r(c, half(fs, kind, p), half(fr, kind, p), pt.cnt[p], parallel, &r)); res += r; }
  return res;
}
/* linalg/assign_add_mul_r.c:346-381 */
void assign_add_mul_r(spinor *const P, spinor *const Q, const double cc, const int N) {
  tmhip_ctx *c = refresh(false);
  if (N == 0) return;   /* an empty loop in the reference */
  const int kind = kind_of_N(N);
  const Parts pt = parts_of(kind, N);
  tmhip_field *fp = in(c, P, kind, N), *fq = in(c, Q, kind, N);
  for (int p = 0; p < pt.n; p++) CK(tmhip_assign_add_mul_r(c, half(fp, kind, p), half(fq, kind, p), cc, pt.cnt[p]));
  done(c, P);
}
/* linalg/assign_mul_add_r.c:340-377 */
void assign_mul_add_r(spinor *const R, const double cc, const spinor *const S, const int N) {
  tmhip_ctx *c = refresh(false);
  if (N == 0) return;   /* an empty loop in the reference */
  const int kind = kind_of_N(N);
  const Parts pt = parts_of(kind, N);
  tmhip_field *fr = in(c, R, kind, N), *fs = in(c, S, kind, N);
  for (int p = 0; p < pt.n; p++) CK(tmhip_assign_mul_add_r(c, half(fr, kind, p), cc, half(fs, kind, p), pt.cnt[p]));
  done(c, R);
}
/* linalg/assign_mul_add_r_and_square.c:145-213 */
double assign_mul_add_r_and_square(spinor *const R, const double cc, const spinor *const S, const int N, const int parallel) {
  tmhip_ctx *c = refresh(false);
  if (N == 0) return 0.;   /* an empty loop in the reference */
  const int kind = kind_of_N(N);
  const Parts pt = parts_of(kind, N);
  tmhip_field *fr = in(c, R, kind, N), *fs = in(c, S, kind, N);
  double res = 0, r;
  for (int p = 0; p < pt.n; p++) {
    CK(tmhip_assign_mul_add_r_and_square(c, half(fr, kind, p), cc, half(fs, kind, p), pt.cnt[p], parallel, &r));
    res += r;
  }
  done(c, R);
  return res;
}
/* linalg/diff.c:270-309 */
void diff(spinor *const Q, const spinor *const R, const spinor *const S, const int N) {
  tmhip_ctx *c = refresh(false);
  if (N == 0) return;   /* an empty loop in the reference */
  const int kind = kind_of_N(N);
  const Parts pt = parts_of(kind, N);
  tmhip_field *fr = in(c, R, kind, N), *fs = in(c, S, kind, N), *fq = out(c, Q, kind, N);
  for (int p = 0; p < pt.n; p++) CK(tmhip_diff(c, half(fq, kind, p), half(fr, kind, p), half(fs, kind, p), pt.cnt[p]));
  done(c, Q);
}
/* linalg/add.c:45-80 */
void add(spinor *const Q, const spinor *const R, const spinor *const S, const int N) {
  tmhip_ctx *c = refresh(false);
  if (N == 0) return;   /* an empty loop in the reference */
  const int kind = kind_of_N(N);
  const Parts pt = parts_of(kind, N);
  tmhip_field *fr = in(c, R, kind, N), *fs = in(c, S, kind, N), *fq = out(c, Q, kind, N);
  for (int p = 0; p < pt.n; p++) CK(tmhip_add(c, half(fq, kind, p), half(fr, kind, p), half(fs, kind, p), pt.cnt[p]));
  done(c, Q);
}
/* linalg/mul_r.c:40-75 */
void mul_r(spinor *const R, const double cc, spinor *const S, const int N) {
  tmhip_ctx *c = refresh(false);
  if (N == 0) return;   /* an empty loop in the reference */
  const int kind = kind_of_N(N);
  const Parts pt = parts_of(kind, N);
  tmhip_field *fs = in(c, S, kind, N), *fr = out(c, R, kind, N);
  for (int p = 0; p < pt.n; p++) CK(tmhip_mul_r(c, half(fr, kind, p), cc, half(fs, kind, p), pt.cnt[p]));
  done(c, R);
}
/* linalg/assign.c:42-46 */
void assign(spinor *const R, spinor *const S, const int N) {
  tmhip_ctx *c = refresh(false);
  if (N == 0) return;   /* an empty loop in the reference */
  const int kind = kind_of_N(N);
  const Parts pt = parts_of(kind, N);
  tmhip_field *fs = in(c, S, kind, N), *fr = out(c, R, kind, N);
  for (int p = 0; p < pt.n; p++) CK(tmhip_assign(c, half(fr, kind, p), half(fs, kind, p), pt.cnt[p]));
  done(c, R);
}

/* fp32 instances of the site-diagonal twists (tm_operators.c:8-47 -> mul_one_pm_imu_inv_body.c, mul_one_pm_imu_sub_mul_body.c);
 * callers hand over domain blocks of any length (solver/Msap.c:409-418), so these go through a staging buffer, not the registry */
void mul_one_pm_imu_inv_32(spinor32 *const l, const double _sign, const int N) {
  tmhip_ctx *c = refresh(false);
  const float nrm = (float)(1. / (1. + g_mu * g_mu));
  CK(tmhip_diag32_host(c, l, l, nullptr, nrm, (_sign < 0. ? 1. : -1.) * nrm * g_mu, N));
}
void assign_mul_one_pm_imu_inv_32(spinor32 *const l, spinor32 *const k, const double _sign, const int N) {
  tmhip_ctx *c = refresh(false);
  const float nrm = (float)(1. / (1. + g_mu * g_mu));
  CK(tmhip_diag32_host(c, l, k, nullptr, nrm, (_sign < 0. ? 1. : -1.) * nrm * g_mu, N));
}
void mul_one_pm_imu_sub_mul_32(spinor32 *const l, spinor32 *const k, spinor32 *const j, const double _sign, const int N) {
  tmhip_ctx *c = refresh(false);
  CK(tmhip_diag32_host(c, l, k, j, 1., (_sign < 0. ? -1. : 1.) * g_mu, N));
}

// ------------------------------------------------------------------ fp32 twins on host spinor32 arrays
// Hopping_Matrix_32 (operator/Hopping_Matrix_32.c:97-127), Qtm_pm_psi_32 (operator/tm_operators_32.c:94-112) and the fp32 linalg
// (linalg/*_32.c) by their reference names.  Host spinor32 arrays are not kept in the registry (the solvers that iterate in fp32 --
// mixed_cg_her, rg_mixed_cg_her -- run device-resident through their own entry points below): every call copies its operands in and its
// result out through a small pool of device fields, i.e. the coherent semantics of the fp64 symbols, PCIe-bound and exact.
// The `_orphaned` convention (SURVEY 8b "Threading"): the reference's fp32 operators are called INSIDE an enclosing OpenMP parallel
// region by all its threads (Qtm_pm_psi_32 opens the region, operator/tm_operators_32.c:96-110).  Here ONE thread issues the device
// call and all threads of the team meet before and after it -- orphaned `omp barrier` / `omp master`, which bind to whatever
// region encloses the call and are no-ops outside of one (this file is compiled with -fopenmp).
extern "C++" {
namespace {
tmhip_field *f32(tmhip_ctx *c, int k) {
  if (!g_f32[k]) CK(tmhip_field_alloc32(c, &g_f32[k]));
  return g_f32[k];
}
tmhip_field *in32(tmhip_ctx *c, int k, const spinor32 *host, int N) {
  tmhip_field *f = f32(c, k);
  CK(tmhip_field_upload32(c, f, host, N));
  return f;
}
void need_N32(int N, const char *who) {
  if (N < 0 || N > VOLUME / 2) { fprintf(stderr, "[tmlqcd_dropin] %s: N = %d outside [0, VOLUME/2] (fp32 fields are one-parity fields)\n", who, N); exit(1); }
}
template <class F> inline void team_once(F body) {
#pragma omp barrier
#pragma omp master
  body();
#pragma omp barrier
}
}  // namespace
}  // extern "C++"

void Hopping_Matrix_32(const int ieo, spinor32 *const l, spinor32 *const k) {   /* called from the master thread outside any parallel region */
  tmhip_ctx *c = refresh(true);
  if ((void *)l == (void *)k) die("Hopping_Matrix_32: l and k must differ");
  tmhip_field *fk = in32(c, 0, k, VOLUME / 2), *fl = f32(c, 1);
  CK(tmhip_hopping_matrix_32(c, ieo, fl, fk));
  CK(tmhip_field_download32(c, fl, l, VOLUME / 2));
}
void Hopping_Matrix_32_orphaned(const int ieo, spinor32 *const l, spinor32 *const k) {   /* by every thread of the enclosing team */
  team_once([&] { Hopping_Matrix_32(ieo, l, k); });
}
void Qtm_pm_psi_32(spinor32 *const l, spinor32 *const k) {
  tmhip_ctx *c = refresh(true);
  tmhip_field *fk = in32(c, 0, k, VOLUME / 2), *fl = f32(c, 1);
  CK(tmhip_Qtm_pm_psi_32(c, fl, fk));
  CK(tmhip_field_download32(c, fl, l, VOLUME / 2));
}
float square_norm_32(const spinor32 *const P, const int N, const int parallel) {   /* linalg/square_norm_32.c:95 */
  tmhip_ctx *c = refresh(false);
  need_N32(N, "square_norm_32");
  if (N == 0) return 0.f;
  double r = 0;
  CK(tmhip_square_norm_32(c, in32(c, 0, P, N), N, parallel, &r));
  return (float)r;
}
float scalar_prod_r_32(const spinor32 *const S, const spinor32 *const R, const int N, const int parallel) {   /* linalg/scalar_prod_r_32.c:109 */
  tmhip_ctx *c = refresh(false);
  need_N32(N, "scalar_prod_r_32");
  if (N == 0) return 0.f;
  double r = 0;
  tmhip_field *fs = in32(c, 0, S, N), *fr = (const void *)S == (const void *)R ? fs : in32(c, 1, R, N);
  CK(tmhip_scalar_prod_r_32(c, fs, fr, N, parallel, &r));
  return (float)r;
}
void assign_add_mul_r_32(spinor32 *const R, spinor32 *const S, const float cc, const int N) {   /* linalg/assign_add_mul_r_32.c:104: R += c S */
  tmhip_ctx *c = refresh(false);
  need_N32(N, "assign_add_mul_r_32");
  if (N == 0) return;
  tmhip_field *fr = in32(c, 0, R, N), *fs = (void *)S == (void *)R ? fr : in32(c, 1, S, N);
  CK(tmhip_assign_add_mul_r_32(c, fr, fs, cc, N));
  CK(tmhip_field_download32(c, fr, R, N));
}
void assign_mul_add_r_32(spinor32 *const R, const float cc, const spinor32 *const S, const int N) {   /* linalg/assign_mul_add_r_32.c:81: R = c R + S */
  tmhip_ctx *c = refresh(false);
  need_N32(N, "assign_mul_add_r_32");
  if (N == 0) return;
  tmhip_field *fr = in32(c, 0, R, N), *fs = (const void *)S == (const void *)R ? fr : in32(c, 1, S, N);
  CK(tmhip_assign_mul_add_r_32(c, fr, cc, fs, N));
  CK(tmhip_field_download32(c, fr, R, N));
}
void diff_32(spinor32 *const Q, const spinor32 *const R, const spinor32 *const S, const int N) {   /* linalg/diff_32.c:39: Q = R - S */
  tmhip_ctx *c = refresh(false);
  need_N32(N, "diff_32");
  if (N == 0) return;
  tmhip_field *fq = in32(c, 0, S, N), *fr = in32(c, 1, R, N);     // Q = -1 * S + R
  CK(tmhip_assign_mul_add_r_32(c, fq, -1.f, fr, N));
  CK(tmhip_field_download32(c, fq, Q, N));
}
void assign_to_32(spinor32 *const R, spinor *const S, const int N) {   /* linalg/assign_to_32.c:37: the fp64 operand goes through the registry like any other input */
  tmhip_ctx *c = refresh(false);
  need_N32(N, "assign_to_32");
  if (N == 0) return;
  tmhip_field *fs = in(c, S, TMHIP_FIELD_EO), *fr = f32(c, 0);
  CK(tmhip_assign_to_32(c, fr, fs, N));
  CK(tmhip_field_download32(c, fr, R, N));
}
void assign_to_64(spinor *const R, spinor32 *const S, const int N) {   /* linalg/assign_to_32.c:84 */
  tmhip_ctx *c = refresh(false);
  need_N32(N, "assign_to_64");
  if (N == 0) return;
  if (N != VOLUME / 2) die("assign_to_64: N must be VOLUME/2 (the fp64 result is a registered one-parity field)");
  tmhip_field *fs = in32(c, 0, S, N), *fr = out(c, R, TMHIP_FIELD_EO);
  CK(tmhip_assign_to_64(c, fr, fs, N));
  done(c, R);
}

// ------------------------------------------------------------------ solver
/* solver/cg_her.c:62-141.  For the e/o operators of this library the whole solve runs
 * device-resident (tmhip_cg_her); for any other `f` the reference loop is executed with the
 * drop-in linalg in COHERENT mode, which is correct for an arbitrary host-side f. */
int cg_her(spinor *const P, spinor *const Q, const int max_iter, double eps_sq, const int rel_prec, const int N,
           matrix_mult f) {
  int op = -1;
  if (f == &Qtm_pm_psi) op = TMHIP_OP_QTM_PM;
  else if (f == &Qtm_plus_psi) op = TMHIP_OP_QTM_PLUS;
  else if (f == &Qtm_minus_psi) op = TMHIP_OP_QTM_MINUS;
  else if (f == &Mtm_plus_psi) op = TMHIP_OP_MTM_PLUS;
  else if (f == &Mtm_minus_psi) op = TMHIP_OP_MTM_MINUS;
  else if (f == &Qsw_pm_psi) op = TMHIP_OP_QSW_PM;
  if (op >= 0 && N == VOLUME / 2) {
    tmhip_ctx *c = op == TMHIP_OP_QSW_PM ? refresh_clover() : refresh(true);
    tmhip_field *fq = in(c, Q, TMHIP_FIELD_EO), *fp = in(c, P, TMHIP_FIELD_EO);
    int iters = -1;
    CK(tmhip_cg_her(c, fp, fq, max_iter, eps_sq, rel_prec, N, op, &iters, nullptr, 0));
    done(c, P);          // coherent mode: the solution is on the host when we return; resident mode: after tmlqcd_hip_sync_to_host
    return iters;
  }
  // generic path: reference algorithm verbatim on host-visible fields
  const int saved = g_mode;
  tmlqcd_hip_set_residency(TMLQCD_HIP_COHERENT);
  const size_t Vf = (size_t)(N == VOLUME ? VOLUMEPLUSRAND : VOLUMEPLUSRAND / 2);
  spinor *blk = (spinor *)calloc(3 * Vf + 1, sizeof(spinor));   /* solver_field.c:31-71 */
  if (!blk) die("cg_her: out of memory");
  spinor *sf[3] = {blk, blk + Vf, blk + 2 * Vf}, *stmp;
  double normsq, pro, err, alpha_cg, beta_cg, squarenorm;
  int iteration;
  squarenorm = square_norm(Q, N, 1);
  f(sf[0], P);
  diff(sf[1], Q, sf[0], N);
  assign(sf[2], sf[1], N);
  normsq = square_norm(sf[1], N, 1);
  for (iteration = 1; iteration <= max_iter; iteration++) {
    f(sf[0], sf[2]);
    pro = scalar_prod_r(sf[2], sf[0], N, 1);
    alpha_cg = normsq / pro;
    assign_add_mul_r(P, sf[2], alpha_cg, N);
    err = assign_mul_add_r_and_square(sf[0], -alpha_cg, sf[1], N, 1);
    if (((err <= eps_sq) && (rel_prec == 0)) || ((err <= eps_sq * squarenorm) && (rel_prec == 1))) break;
    beta_cg = err / normsq;
    assign_mul_add_r(sf[2], beta_cg, sf[0], N);
    stmp = sf[0]; sf[0] = sf[1]; sf[1] = stmp;
    normsq = err;
  }
  for (int i = 0; i < 3; i++) tmlqcd_hip_forget(blk + i * Vf);  // addresses are about to be recycled
  free(blk);
  g_mode = saved;
  if (iteration > max_iter) return -1;
  return iteration;
}

/* solver/mixed_cg_her.c:65-202 with f = Qtm_pm_psi: fp32 inner CG + fp64 defect correction, all in HBM */
static_assert(sizeof(tmlqcd_solver_params) == 144 && offsetof(tmlqcd_solver_params, mcg_delta) == 52,
              "solver_params_t layout (solver/solver_params.h:46-109)");
int mixed_cg_her(spinor *const P, spinor *const Q, tmlqcd_solver_params, const int max_iter, double eps_sq,
                 const int rel_prec, const int N, matrix_mult f, matrix_mult32) {
  if ((f != &Qtm_pm_psi && f != &Qsw_pm_psi) || N != VOLUME / 2) die("mixed_cg_her: only f = Qtm_pm_psi / Qsw_pm_psi on VOLUME/2 sites runs on the device");
  const int op = f == &Qsw_pm_psi ? TMHIP_OP_QSW_PM : TMHIP_OP_QTM_PM;
  const double innereps = &mixcg_innereps ? mixcg_innereps : 5.0e-5;           /* default_input_values.h:193 */
  const int max_inner = &mixcg_maxinnersolverit ? mixcg_maxinnersolverit : 5000; /* default_input_values.h:194 */
  tmhip_ctx *c = op == TMHIP_OP_QSW_PM ? refresh_clover() : refresh(true);
  tmhip_field *fq = in(c, Q, TMHIP_FIELD_EO), *fp = out(c, P, TMHIP_FIELD_EO);
  int iters = -1, outer = 0;
  CK(tmhip_mixed_cg_her(c, fp, fq, max_iter, eps_sq, rel_prec, N, op, innereps, max_inner, &iters, &outer));
  done(c, P);
  return iters;
}

/* solver/rg_mixed_cg_her.c:180-347 with (f, f32) = (Qtm_pm_psi, Qtm_pm_psi_32) or (Qsw_pm_psi, Qsw_pm_psi_32) */
int rg_mixed_cg_her(spinor *const P, spinor *const Q, tmlqcd_solver_params solver_params, const int max_iter,
                    const double eps_sq, const int rel_prec, const int N, matrix_mult f, matrix_mult32) {
  if ((f != &Qtm_pm_psi && f != &Qsw_pm_psi) || N != VOLUME / 2) die("rg_mixed_cg_her: only f = Qtm_pm_psi / Qsw_pm_psi on VOLUME/2 sites runs on the device");
  const int op = f == &Qsw_pm_psi ? TMHIP_OP_QSW_PM : TMHIP_OP_QTM_PM;
  tmhip_ctx *c = op == TMHIP_OP_QSW_PM ? refresh_clover() : refresh(true);
  tmhip_field *fq = in(c, Q, TMHIP_FIELD_EO), *fp = out(c, P, TMHIP_FIELD_EO);
  int iters = -1;
  CK(tmhip_rg_mixed_cg_her(c, fp, fq, max_iter, eps_sq, rel_prec, N, op, solver_params.mcg_delta, &iters, nullptr, nullptr, nullptr));
  done(c, P);
  return iters;
}

// ------------------------------------------------------------------ fermion force
static bool g_deriv_pending = false;
/* deriv_Sb.c:401-700 */
void deriv_Sb(const int ieo, spinor *const l, spinor *const k, hamiltonian_field_t *const hf, const double factor) {
  tmhip_ctx *c = refresh(true);
  tmhip_field *fl = in(c, l, TMHIP_FIELD_EO), *fk = in(c, k, TMHIP_FIELD_EO);
  if (!g_deriv_pending) CK(tmhip_derivative_zero(c));
  CK(tmhip_deriv_Sb(c, ieo, fl, fk, factor));
  g_deriv_pending = true;
  if (g_mode != TMLQCD_HIP_RESIDENT) tmlqcd_hip_flush_derivative(hf);   // (lazy mode watches spinor arrays only)
}
/* Clover part of the force under helper names (the reference keeps sw_deriv_nd / sw_spinor in the same objects, which therefore
 * stay on the link line): the statements of cloverdet_derivative, monomial/cloverdet_monomial.c:67-72,125-147 */
void tmlqcd_hip_swpm_zero(void) { CK(tmhip_swpm_zero(refresh(false))); }
void tmlqcd_hip_sw_spinor_eo(const int ieo, const spinor *const kk, const spinor *const ll, const double fac) {   /* clover_deriv.c:252 */
  tmhip_ctx *c = refresh(false);
  tmhip_field *fk = in(c, kk, TMHIP_FIELD_EO), *fl = in(c, ll, TMHIP_FIELD_EO);
  CK(tmhip_sw_spinor_eo(c, ieo, fk, fl, fac));
}
void tmlqcd_hip_sw_deriv(const int ieo, const double mu) {   /* clover_deriv.c:72 */
  tmhip_ctx *c = refresh_clover();
  CK(tmhip_sw_deriv(c, ieo, mu));
}
void tmlqcd_hip_sw_all(hamiltonian_field_t *const hf, const double kappa, const double c_sw) {   /* clover_accumulate_deriv.c:58 */
  tmhip_ctx *c = refresh(true);
  if (!g_deriv_pending) CK(tmhip_derivative_zero(c));
  CK(tmhip_sw_all(c, &hf->gaugefield[0][0], kappa, c_sw));
  g_deriv_pending = true;
  if (g_mode != TMLQCD_HIP_RESIDENT) tmlqcd_hip_flush_derivative(hf);   // (lazy mode watches spinor arrays only)
}
void tmlqcd_hip_flush_derivative(hamiltonian_field_t *const hf) {
  if (!g_deriv_pending) return;
  CK(tmhip_derivative_download(ctx(), &hf->derivative[0][0], 1));
  g_deriv_pending = false;
}

// ------------------------------------------------------------------ molecular dynamics with the links in HBM
/* update_gauge(step, hf) (update_gauge.c:51-110): U <- restoresu3(exposu3(step P)) U for every link, on the device-resident
 * links; the stencil's gauge copy is re-sorted there too (update_backward_gauge.c:185-242), so an MD step moves no gauge
 * field over PCIe.  Coherent mode: hf->gaugefield receives the new links before the call returns and the reference's flags
 * are raised (update_gauge.c:104-106) -- the next stencil call refreshes the HOST's backward copy if the program has one, but
 * does not upload again.  Resident mode: g_gauge_field stays behind until tmlqcd_hip_sync_gauge_to_host.
 * The clover blocks become stale exactly as in the reference (the monomials call sw_term / sw_invert again). */
// Host links and device links have just been brought to the same state by a download.  update_gauge.c:104-106 raises the flags
// here and leaves the refresh of the host's backward copy to the next consumer; this library IS the next consumer of
// g_update_gauge_copy, and it must be able to tell its own raise from a later one by the host program (which means "my links
// changed: upload them").  So the host's backward copy is refreshed right away (update_backward_gauge clears the flag) and the
// flag stays down: whoever raises it afterwards forces an upload.  The host's fp32 copy is host business: its flag is raised.
static void links_in_step(su3 **gf, hamiltonian_field_t *hf) {
  if (update_backward_gauge) update_backward_gauge(gf);
  g_update_gauge_copy = 0;
  if (hf) hf->update_gauge_copy = 0;
  if (&g_update_gauge_copy_32) g_update_gauge_copy_32 = 1;
  g_gauge_uploaded = true;
}
void tmlqcd_hip_update_gauge(const double step, hamiltonian_field_t *const hf) {
  tmhip_ctx *c = refresh(true);                                   // first call of a trajectory: the host's links go up once
  if (!g_momenta_resident) CK(tmhip_momenta_upload(c, &hf->momenta[0][0]));
  CK(tmhip_update_gauge(c, step));
  g_clover_uploaded = false;
  if (g_mode != TMLQCD_HIP_RESIDENT) {
    CK(tmhip_gauge_download(c, &hf->gaugefield[0][0]));
    links_in_step(hf->gaugefield, hf);
  } else {
    g_dev_links_newer = true;
  }
}
void tmlqcd_hip_sync_gauge_to_host(hamiltonian_field_t *const hf) {
  if (!g_dev_links_newer) return;
  CK(tmhip_gauge_download(ctx(), &hf->gaugefield[0][0]));
  links_in_step(hf->gaugefield, hf);                              // host-side consumers find the backward copy current
  g_dev_links_newer = false;
}
/* update_momenta.c:67-72 for a force that was accumulated on the device only (deriv_Sb / tmlqcd_hip_sw_all in resident mode,
 * not flushed): P -= step * derivative with both resident; the momenta then stay on the device until
 * tmlqcd_hip_sync_momenta_to_host.  Contributions other monomials left in hf->derivative are NOT included. */
void tmlqcd_hip_update_momenta(const double step, hamiltonian_field_t *const hf) {
  tmhip_ctx *c = refresh(false);
  if (!g_momenta_resident) { CK(tmhip_momenta_upload(c, &hf->momenta[0][0])); g_momenta_resident = true; }
  CK(tmhip_update_momenta(c, step));
  g_deriv_pending = false;                                        // consumed
}
void tmlqcd_hip_sync_momenta_to_host(hamiltonian_field_t *const hf) {
  if (!g_momenta_resident) return;
  CK(tmhip_momenta_download(ctx(), &hf->momenta[0][0]));
  g_momenta_resident = false;
}

// ------------------------------------------------------------------ ILDG gauge configurations
/* io/gauge_read.c:26-27: the record read_gauge_field fills (defined by the object this library replaces) */
paramsGaugeInfo GaugeInfo = {0., 0, {0, 0}, NULL, NULL};

/* io/gauge_read.c:28-198 read_gauge_field(filename, gf): LIME records walked on the host, the binary record unpacked and
 * check-summed in HBM; gf (the host's g_gauge_field) is filled, GaugeInfo set, g_update_gauge_copy raised; returns 0 or -1 with the
 * reference's messages. */
int read_gauge_field(char *filename, su3 **const gf) {
  tmhip_ctx *c = ctx();      // (T-split ranks: every rank reads its part of the record, tmlqcd_hip_comm_init must have been called)
  g_calls++;
  static tmhip_gauge_info info;
  const int prec = &gauge_precision_read_flag && gauge_precision_read_flag == 32 ? 32 : 64;
  const int checks = !(&g_disable_IO_checks && g_disable_IO_checks);
  GaugeInfo.gaugeRead = 0;
  const int rc = tmhip_read_gauge_field(c, filename, prec, checks, &gf[0][0], &info);
  if (rc == -1) return -1;
  if (rc != 0) die("tmhip_read_gauge_field");
  GaugeInfo.gaugeRead = info.gauge_read;
  GaugeInfo.checksum.suma = info.suma; GaugeInfo.checksum.sumb = info.sumb;
  if (info.xlf_info[0]) { free(GaugeInfo.xlfInfo); GaugeInfo.xlfInfo = strdup(info.xlf_info); }
  if (info.ildg_data_lfn[0]) { free(GaugeInfo.ildg_data_lfn); GaugeInfo.ildg_data_lfn = strdup(info.ildg_data_lfn); }
  g_update_gauge_copy = 1;                                          /* gauge_read.c:190 */
  g_clover_uploaded = false;
  // (the flag stays raised exactly as the reference leaves it: the host program still has its xchange_gauge to do, and the next
  // operator call uploads g_gauge_field once more -- 11 ms per configuration read at 32^4 -- rather than guess that nothing changed)
  if (gf == g_gauge_field && info.gauge_read) g_dev_links_newer = false;
  return 0;
}

/* io/gauge_write.c:22-59 write_gauge_field(filename, prec, xlfInfo): the records of the reference in its order; the binary record and
 * its checksum come from the links in HBM (uploaded from g_gauge_field first unless the device copy is the current one) */
int write_gauge_field(char *filename, const int prec, paramsXlfInfo const *xlfInfo) {
  if (g_nproc_t > 1) die("write_gauge_field: single-rank writer (T-split ranks: tmhip_gauge_pack_ildg for their part of the record)");
  tmhip_ctx *c = ctx();
  g_calls++;
  if (!g_dev_links_newer) { CK(tmhip_set_gauge(c, &g_gauge_field[0][0])); g_gauge_uploaded = true; g_clover_uploaded = false; }
  char msg[1024];
  msg[0] = 0;
  if (xlfInfo) {                                                    /* io/utils_write_xlf.c:35-55: plain text, what write_gauge_field (io/gauge_write.c:35) writes */
    if (xlfInfo->kappa != 0.0)
      snprintf(msg, sizeof(msg), "plaquette = %14.12f\n trajectory nr = %d\n beta = %.12f, kappa = %.12f, mu = %.12f, c2_rec = %f\n time = %ld\n"
               " hmcversion = %s\n mubar = %.12f\n epsilonbar = %.12f\n date = %s",
               xlfInfo->plaq, xlfInfo->counter, xlfInfo->beta, xlfInfo->kappa, xlfInfo->mu, xlfInfo->c2_rec, xlfInfo->time, xlfInfo->package_version,
               xlfInfo->mubar, xlfInfo->epsilonbar, xlfInfo->date);
    else
      snprintf(msg, sizeof(msg), "plaquette = %e\n trajectory nr = %d\n beta = %.12f\n kappa = %.12f\n 2*kappa*mu = %.12f\n c2_rec = %f\n date = %s",
               xlfInfo->plaq, xlfInfo->counter, xlfInfo->beta, xlfInfo->kappa, xlfInfo->mu, xlfInfo->c2_rec, xlfInfo->date);
  }
  unsigned cs[2];
  return tmhip_write_gauge_field(c, filename, prec, msg[0] ? msg : nullptr, cs) ? -1 : 0;
}

// ------------------------------------------------------------------ benchmark helper
/* benchmark.c:291-300 with the three fields resident in HBM */
double tmlqcd_hip_benchmark_loop(spinor *f0, spinor *f1, spinor *f2, int iters) {
  RegLock lk;
  tmhip_ctx *c = refresh(true);
  const int saved = g_mode;
  g_mode = TMLQCD_HIP_RESIDENT;
  tmhip_field *d0 = in(c, f0, TMHIP_FIELD_EO), *d1 = out(c, f1, TMHIP_FIELD_EO), *d2 = out(c, f2, TMHIP_FIELD_EO);
  double ms = 0;
  CK(tmhip_bench_hopping(c, d0, d1, d2, iters, &ms));
  g_reg[f1].dev_valid = true; g_reg[f1].host_valid = false;
  g_reg[f2].dev_valid = true; g_reg[f2].host_valid = false;
  g_mode = saved;
  if (g_mode == TMLQCD_HIP_COHERENT) { tmlqcd_hip_sync_to_host(f1); tmlqcd_hip_sync_to_host(f2); }
  if (g_mode == TMLQCD_HIP_LAZY)
    for (spinor *f : {f1, f2}) {
      Mirror &m = g_reg[f];
      if (m.nowatch) { download(c, f, m); m.dev_valid = false; }   // (an array inside the malloc heap: copied, never watched)
      else set_prot(f, m, P_NONE, g_reg);
    }
  return ms * 1e-3;
}

}  // extern "C"
