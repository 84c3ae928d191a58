/* Test stand-in for the HOST PROGRAM (benchmark / invert / hmc_tm): defines the tmLQCD globals
 * that libtmlqcd_dropin.so reads at call time (global.h:73-207, boundary.h:25) and lets a test
 * fill them the way tmlqcd_mpi_init (mpi_init.c:748-778), init_gauge_field
 * (init/init_gauge_field.c:51-68) and boundary() (boundary.c:40-55) would.  Not part of the product. */
#include <complex.h>
#include <math.h>
#include <stdlib.h>

typedef struct { double _Complex c00, c01, c02, c10, c11, c12, c20, c21, c22; } su3;

int T, LX, LY, LZ, VOLUME, RAND, VOLUMEPLUSRAND;
int g_nproc_t = 1, g_nproc_x = 1, g_nproc_y = 1, g_nproc_z = 1;
int g_proc_coords[4] = {0, 0, 0, 0};
su3 **g_gauge_field = NULL;
int g_update_gauge_copy = 1;
double g_mu = 0.0, g_kappa = 0.125;
double _Complex ka0, ka1, ka2, ka3;
double mixcg_innereps = 5.0e-5;   /* read_input.l:2912, default_input_values.h:193 */
int mixcg_maxinnersolverit = 5000;
static su3 *gauge_block = NULL;

su3 *stub_init(int T_, int LX_, int LY_, int LZ_) {
  T = T_; LX = LX_; LY = LY_; LZ = LZ_;
  VOLUME = T * LX * LY * LZ; RAND = 0; VOLUMEPLUSRAND = VOLUME;
  free(gauge_block); free(g_gauge_field);
  gauge_block = (su3 *)calloc(4 * (size_t)VOLUMEPLUSRAND + 1, sizeof(su3));
  g_gauge_field = (su3 **)calloc(VOLUMEPLUSRAND, sizeof(su3 *));
  for (int i = 0; i < VOLUMEPLUSRAND; i++) g_gauge_field[i] = gauge_block + 4 * (size_t)i;
  g_update_gauge_copy = 1;
  return gauge_block;
}
/* one rank of a T-split job (PARALLELT: mpi_init.c:240-242,330-332): RAND = the two halo time-slices, t = T first, then t = -1
 * (geometry_eo.c:292-299); the caller fills them like xchange_gauge would */
su3 *stub_init_rank(int T_, int LX_, int LY_, int LZ_, int nproc_t, int proc_t) {
  su3 *p;
  g_nproc_t = nproc_t; g_proc_coords[0] = proc_t;
  T = T_; LX = LX_; LY = LY_; LZ = LZ_;
  VOLUME = T * LX * LY * LZ; RAND = nproc_t > 1 ? 2 * LX * LY * LZ : 0; VOLUMEPLUSRAND = VOLUME + RAND;
  free(gauge_block); free(g_gauge_field);
  gauge_block = (su3 *)calloc(4 * (size_t)VOLUMEPLUSRAND + 1, sizeof(su3));
  g_gauge_field = (su3 **)calloc(VOLUMEPLUSRAND, sizeof(su3 *));
  for (int i = 0; i < VOLUMEPLUSRAND; i++) g_gauge_field[i] = gauge_block + 4 * (size_t)i;
  g_update_gauge_copy = 1;
  p = gauge_block;
  return p;
}
void stub_boundary(double kappa, double x0, double x1, double x2, double x3) {
  const double PI_ = 3.14159265358979;
  g_kappa = kappa;
  ka0 = kappa * cexp(x0 * PI_ / (T * g_nproc_t) * I); ka1 = kappa * cexp(x1 * PI_ / LX * I);
  ka2 = kappa * cexp(x2 * PI_ / LY * I); ka3 = kappa * cexp(x3 * PI_ / LZ * I);
}
void stub_set_mu(double mu) { g_mu = mu; }
double stub_get_mu(void) { return g_mu; }
double g_c_sw = 0.0;                         /* global.h:198 */
double g_mu3 = 0.0;                          /* global.h:197 */
void stub_set_mu3(double mu3) { g_mu3 = mu3; }
void stub_set_csw(double c_sw) { g_c_sw = c_sw; }
void stub_mark_gauge_dirty(void) { g_update_gauge_copy = 1; }
int stub_gauge_flag(void) { return g_update_gauge_copy; }

/* clover fields, allocated the way init_sw_fields does (operator/clovertm_operators.c:1162-1210):
 * sw[ix][a] -> &block[(ix*3 + a)*2], sw_inv[icy][a] -> &block[(icy*4 + a)*2] */
su3 ***sw = NULL, ***sw_inv = NULL;
static su3 *sw_block = NULL, *swinv_block = NULL;
static su3 **sw1 = NULL, **swinv1 = NULL;
static int sw_volume = 0;   /* the lattice the clover arrays were allocated for: tests re-initialise the stub with other sizes */
su3 *stub_init_clover(int which) {
  const int V = VOLUME;
  if (sw && sw_volume != V) {
    free(sw); free(sw_inv); free(sw1); free(swinv1); free(sw_block); free(swinv_block);
    sw = NULL; sw_inv = NULL;
  }
  if (!sw) {
    sw_volume = V;
    sw = (su3 ***)calloc(V, sizeof(su3 **)); sw_inv = (su3 ***)calloc(V, sizeof(su3 **));
    sw1 = (su3 **)calloc(3 * (size_t)V, sizeof(su3 *)); swinv1 = (su3 **)calloc(4 * (size_t)V, sizeof(su3 *));
    sw_block = (su3 *)calloc(6 * (size_t)V + 1, sizeof(su3)); swinv_block = (su3 *)calloc(8 * (size_t)V + 1, sizeof(su3));
    for (int i = 0; i < V; i++) {
      sw[i] = sw1 + 3 * (size_t)i; sw_inv[i] = swinv1 + 4 * (size_t)i;
      for (int a = 0; a < 3; a++) sw[i][a] = sw_block + ((size_t)i * 3 + a) * 2;
      for (int a = 0; a < 4; a++) sw_inv[i][a] = swinv_block + ((size_t)i * 4 + a) * 2;
    }
  }
  return which ? swinv_block : sw_block;
}

/* input-parser globals the ILDG reader looks at (read_input.l: GaugeConfigReadPrecision; global.h:74) */
int gauge_precision_read_flag = 64;
int g_disable_IO_checks = 0;
void stub_set_io(int prec, int disable_checks) { gauge_precision_read_flag = prec; g_disable_IO_checks = disable_checks; }

/* benchmark.c:291-300 as an UNMODIFIED host program runs it: two stencil calls per iteration through the reference's symbol, and
 * the host reads one number of the last output after each iteration (the reference's guard against the loop being optimised away) */
typedef struct { double _Complex c[12]; } stub_spinor;
double stub_benchmark_loop(void (*hop)(int, void *, void *), stub_spinor *f0, stub_spinor *f1, stub_spinor *f2, int iters) {
  double antioptaway = 0.0;
  for (int j = 0; j < iters; j++) {
    hop(0, f1, f0);
    hop(1, f2, f1);
    antioptaway += creal(f2[0].c[0]);
  }
  return antioptaway;
}
/* host code between device calls: scale one spinor in place (a store to a mirrored array) and sum a strided sample of another */
void stub_host_scale(stub_spinor *f, int site, double a) { for (int k = 0; k < 12; k++) f[site].c[k] *= a; }
double stub_host_sample(const stub_spinor *f, int n, int stride) {
  double s = 0.0;
  for (int i = 0; i < n; i += stride) s += creal(f[i].c[3]) + cimag(f[i].c[7]);
  return s;
}

/* host threads reading one array at the same time (an OpenMP loop of the host program over a field): nthreads contiguous chunks */
#include <pthread.h>
typedef struct { const stub_spinor *f; int lo, hi; double s; } stub_job;
static void *stub_sum_job(void *p) {
  stub_job *j = (stub_job *)p;
  double s = 0.0;
  for (int i = j->lo; i < j->hi; i++) for (int k = 0; k < 12; k++) s += creal(j->f[i].c[k]) - cimag(j->f[i].c[k]);
  j->s = s;
  return NULL;
}
double stub_host_sum_threads(const stub_spinor *f, int n, int nthreads) {
  pthread_t th[16];
  stub_job job[16];
  if (nthreads > 16) nthreads = 16;
  for (int t = 0; t < nthreads; t++) {
    job[t].f = f; job[t].lo = (int)((long)n * t / nthreads); job[t].hi = (int)((long)n * (t + 1) / nthreads); job[t].s = 0.0;
    pthread_create(&th[t], NULL, stub_sum_job, &job[t]);
  }
  double s = 0.0;
  for (int t = 0; t < nthreads; t++) { pthread_join(th[t], NULL); s += job[t].s; }
  return s;
}

/* host threads read a stale array WHILE the master thread keeps calling the library on other arrays (registry inserts and erases):
 * the threads start first, the master then issues `ncalls` stencil calls into `nout` different output arrays, forgetting one of
 * them after every call (an insert and an erase per call) */
double stub_threads_read_while_master_calls(void (*hop)(int, void *, void *), void (*forget)(void *), const stub_spinor *stale, int n, int nthreads,
                                            stub_spinor *in, stub_spinor **outs, int nout, int ncalls) {
  pthread_t th[16];
  stub_job job[16];
  if (nthreads > 16) nthreads = 16;
  for (int t = 0; t < nthreads; t++) {
    job[t].f = stale; job[t].lo = (int)((long)n * t / nthreads); job[t].hi = (int)((long)n * (t + 1) / nthreads); job[t].s = 0.0;
    pthread_create(&th[t], NULL, stub_sum_job, &job[t]);
  }
  for (int c = 0; c < ncalls; c++) {
    hop(c & 1, outs[c % nout], in);
    forget(outs[(c + nout / 2) % nout]);
  }
  double s = 0.0;
  for (int t = 0; t < nthreads; t++) { pthread_join(th[t], NULL); s += job[t].s; }
  return s;
}
/* a work field the way solver/solver_field.c gets one -- calloc, used, freed -- in the form glibc gives a block of that size when
 * the heap has no free chunk for it: an anonymous mapping of its own, unmapped again by free (the user pointer 16 bytes in) */
#include <sys/mman.h>
void *stub_calloc(size_t bytes) {
  char *m = mmap(NULL, bytes + 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  return m == MAP_FAILED ? NULL : m + 16;
}
void stub_free(void *p, size_t bytes) { munmap((char *)p - 16, bytes + 4096); }

/* A host program that has its OWN SIGSEGV handler before the library installs the lazy mode's (a debugger hook, a guard-page
 * allocator): faults that are not the library's must still reach it.  The probe handler owns one inaccessible guard page, counts the
 * faults on it and leaves them with siglongjmp; anything else it declines (default action). */
#include <setjmp.h>
#include <signal.h>
#include <string.h>
static sigjmp_buf stub_probe_jb;
static volatile int stub_probe_faults = 0;
static char *stub_probe_guard = NULL;
static void stub_probe_segv(int sig, siginfo_t *si, void *uctx) {
  (void)uctx;
  if (stub_probe_guard && (char *)si->si_addr >= stub_probe_guard && (char *)si->si_addr < stub_probe_guard + 4096) {
    stub_probe_faults++;
    siglongjmp(stub_probe_jb, 1);
  }
  signal(sig, SIG_DFL);   /* not ours either: die the ordinary way when the instruction runs again */
}
int stub_install_segv_probe(void) {
  stub_probe_guard = mmap(NULL, 4096, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (stub_probe_guard == MAP_FAILED) return -1;
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = stub_probe_segv;
  sa.sa_flags = SA_SIGINFO | SA_NODEFER;
  sigemptyset(&sa.sa_mask);
  return sigaction(SIGSEGV, &sa, NULL);
}
/* load from the guard page: returns the number of faults the probe handler has seen so far (-1: the load did not fault) */
int stub_touch_guard(void) {
  if (!sigsetjmp(stub_probe_jb, 1)) {
    volatile char c = *(volatile char *)stub_probe_guard;
    (void)c;
    return -1;
  }
  return stub_probe_faults;
}
