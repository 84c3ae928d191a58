/* TEST INFRASTRUCTURE ONLY -- see oracle/README.md.
 *
 * Driver TU for building the *reference's own* hot-path objects
 * (/root/reference, compiled in place, nothing copied) into oracle/_ref/.
 *
 * It instantiates tmLQCD's globals through the reference's own mechanism
 * (global.h:54-58: "#if defined INIT_GLOBALS -> EXTERN is empty"), which in a
 * normal tmLQCD build happens inside the flex-generated parser (read_input.l:55).
 * The init order below follows benchmark.c:127-259 and the scalar branch of
 * tmlqcd_mpi_init (mpi_init.c:748-778).
 *
 * No reference algorithm is restated here: everything numerical that this
 * library does is executed by object code compiled from /root/reference.
 */
#define INIT_GLOBALS
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#ifdef TM_USE_OMP
#include <omp.h>
#endif
#include "global.h"
#include "su3.h"
#include "geometry_eo.h"
#include "boundary.h"
#include "start.h"
extern double X0, X1, X2, X3; /* boundary.c:37 */
#include "init/init_gauge_field.h"
#include "init/init_geometry_indices.h"
#include "operator/clovertm_operators.h"
#include "operator/clover_leaf.h"
#ifdef TM_USE_OMP
#include "init/init_omp_accumulators.h"
#endif
#ifdef _USE_HALFSPINOR
#include "init/init_dirac_halfspinor.h"
#endif

static spinor *tmref_spinor_base = NULL;
static int tmref_nfields = 0;

/* Set lattice, allocate fields, build index tables and boundary phases. */
int tmref_init(int T_, int LX_, int LY_, int LZ_, double kappa, double mu,
               int nfields, int nthreads) {
  g_nproc = 1; g_proc_id = 0; g_nproc_x = g_nproc_y = g_nproc_z = g_nproc_t = 1;
  g_cart_id = 0; g_stdio_proc = 0;
  g_proc_coords[0] = g_proc_coords[1] = g_proc_coords[2] = g_proc_coords[3] = 0;
  T_global = T_; T = T_; L = LX_; LX = LX_; LY = LY_; LZ = LZ_;
  VOLUME = T * LX * LY * LZ; SPACEVOLUME = VOLUME / T;
  RAND = 0; EDGES = 0; VOLUMEPLUSRAND = VOLUME; SPACERAND = 0;
  N_PROC_T = N_PROC_X = N_PROC_Y = N_PROC_Z = 1;
  g_dbw2rand = 0; lowmem_flag = 0; g_debug_level = 0;
  g_kappa = kappa; g_mu = mu; g_c_sw = 0.0; g_rgi_C1 = 1.;
  g_sloppy_precision = 0; g_sloppy_precision_flag = 0;
  X0 = X1 = X2 = X3 = 0.0;
  DUM_DERI = nfields - 4; DUM_MATRIX = nfields - 3; NO_OF_SPINORFIELDS = nfields;
#ifdef TM_USE_OMP
  omp_num_threads = nthreads > 0 ? nthreads : 1;
  omp_set_num_threads(omp_num_threads);
  init_omp_accumulators(omp_num_threads);
#else
  (void)nthreads;
#endif
  if (init_gauge_field(VOLUMEPLUSRAND + g_dbw2rand, 1) != 0) return 1;
  if (init_geometry_indices(VOLUMEPLUSRAND + g_dbw2rand) != 0) return 2;
  /* spinor fields: one block, field i = base + i*VOLUMEPLUSRAND (full-lattice
     sized so the same fields serve D_psi and the e/o operators) */
  tmref_nfields = nfields;
  tmref_spinor_base = (spinor *)calloc((size_t)nfields * VOLUMEPLUSRAND + 1, sizeof(spinor));
  g_spinor_field = (spinor **)calloc(nfields, sizeof(spinor *));
  if (!tmref_spinor_base || !g_spinor_field) return 3;
  for (int i = 0; i < nfields; i++) g_spinor_field[i] = tmref_spinor_base + (size_t)i * VOLUMEPLUSRAND;
#ifdef TM_USE_OMP
  /* first touch inside an OpenMP region with the static partition the stencil loops use (Hopping_Matrix.c / hopping_body_dbl.c:
   * "#pragma omp for" over [0, VOLUME/2)): every thread's share of each field lands on its own NUMA node.  calloc'ed pages
   * are not resident until written, so this is where they are placed; the gauge copy is placed the same way by the
   * reference's own parallel update_backward_gauge. */
  for (int i = 0; i < nfields; i++) {
    spinor *f = g_spinor_field[i];
    for (int half = 0; half < 2; half++) {
      spinor *h = f + (size_t)half * (VOLUME / 2);
#pragma omp parallel for schedule(static)
      for (int ix = 0; ix < VOLUME / 2; ix++) memset(h + ix, 0, sizeof(spinor));
    }
  }
#endif
  geometry();
  boundary(g_kappa);
#ifdef _USE_HALFSPINOR
  /* benchmark.c:219-236 (default build) + invert.c:176-299 for the fp32 twins */
  if (init_dirac_halfspinor() != 0) return 4;
  if (init_gauge_field_32(VOLUMEPLUSRAND + g_dbw2rand, 1) != 0) return 5;
  if (init_dirac_halfspinor32() != 0) return 6;
  NO_OF_SPINORFIELDS_32 = 6;
  g_spinor_field32 = (spinor32 **)calloc(NO_OF_SPINORFIELDS_32, sizeof(spinor32 *));
  for (int i = 0; i < NO_OF_SPINORFIELDS_32; i++) g_spinor_field32[i] = (spinor32 *)calloc(VOLUMEPLUSRAND / 2 + 1, sizeof(spinor32));
#endif
  return 0;
}

#ifdef _USE_HALFSPINOR
/* update_gauge(step, hf) (update_gauge.c:51) on g_gauge_field with the momenta handed in as su3adj [VOLUME][4]; only in this
 * build because the function ends by converting the links into g_gauge_field_32 */
#include "hamiltonian_field.h"
#include "update_gauge.h"
void tmref_update_gauge(double step, su3adj *mom) {
  hamiltonian_field_t hf;
  su3adj **mp = malloc((size_t)VOLUMEPLUSRAND * sizeof(su3adj *));
  for (int i = 0; i < VOLUMEPLUSRAND; i++) mp[i] = mom + 4 * (size_t)i;
  hf.gaugefield = g_gauge_field; hf.momenta = mp; hf.derivative = NULL; hf.update_gauge_copy = 0; hf.traj_counter = 0;
  update_gauge(step, &hf);
  free(mp);
}
/* invert.c:299 */
void tmref_convert_gauge_32(void) { convert_32_gauge_field(g_gauge_field_32, g_gauge_field, VOLUMEPLUSRAND); g_update_gauge_copy_32 = 1; }

#endif

#if defined(_USE_HALFSPINOR) || defined(TMREF_HOSTPROG)
/* Calls rg_mixed_cg_her (solver/rg_mixed_cg_her.c:180) the way solver/monomial_solve.c does: solver_params_t by
 * value with mcg_delta set, f = Qtm_pm_psi.  In the half-spinor build this is the reference's own solver with
 * f32 = Qtm_pm_psi_32; in the host-program build (TMREF_HOSTPROG) the symbol resolves to the drop-in library,
 * which proves the by-value struct + stack-passed f32 calling sequence against compiler-generated reference code. */
#include "operator/tm_operators.h"
#include "operator/tm_operators_32.h"
#include "solver/solver_params.h"
#include "solver/rg_mixed_cg_her.h"
int tmref_rg_mixed_cg_her(spinor *P, spinor *Q, double delta, int max_iter, double eps_sq, int rel_prec, int N, int debug) {
  solver_params_t sp;
  memset(&sp, 0x5a, sizeof(sp));          /* everything but mcg_delta is junk on purpose: nothing else may be read */
  sp.mcg_delta = (float)delta;
  const int saved = g_debug_level;
  g_debug_level = debug;
#ifdef _USE_HALFSPINOR
  const int it = rg_mixed_cg_her(P, Q, sp, max_iter, eps_sq, rel_prec, N, &Qtm_pm_psi, &Qtm_pm_psi_32);
#else
  const int it = rg_mixed_cg_her(P, Q, sp, max_iter, eps_sq, rel_prec, N, &Qtm_pm_psi, (matrix_mult32)0);
#endif
  g_debug_level = saved;
  return it;
}
#endif

#ifndef TMREF_HOSTPROG
/* Fermion force, hopping part: deriv_Sb(ieo, l, k, hf, factor) (deriv_Sb.c:401) accumulating into a derivative field
 * owned by the harness, laid out like df0 (su3adj [VOLUMEPLUSRAND][4], init/init_moment_field.c). */
#include "hamiltonian_field.h"
#include "deriv_Sb.h"
static su3adj *tmref_df = NULL, **tmref_dfp = NULL;
double *tmref_derivative(void) {
  if (!tmref_df) {
    tmref_df = calloc((size_t)4 * VOLUMEPLUSRAND, sizeof(su3adj));
    tmref_dfp = malloc((size_t)VOLUMEPLUSRAND * sizeof(su3adj *));
    for (int i = 0; i < VOLUMEPLUSRAND; i++) tmref_dfp[i] = tmref_df + 4 * (size_t)i;
  }
  return (double *)tmref_df;
}
/* Clover part of the force (cloverdet_monomial.c:110-147): swm / swp accumulators (init_swpm, clover_leaf.c:141) and sw_all */
#ifndef TMREF_NO_CLOVER
su3 *tmref_swpm(int which) {       /* 0: swm base, 1: swp base; both [VOLUME][4] su3; allocated (zeroed) on first use */
  init_swpm(VOLUME);
  return which ? &swp[0][0] : &swm[0][0];
}
void tmref_swpm_zero(void) {
  init_swpm(VOLUME);
  memset(&swp[0][0], 0, (size_t)4 * VOLUME * sizeof(su3));
  memset(&swm[0][0], 0, (size_t)4 * VOLUME * sizeof(su3));
}
void tmref_sw_all(double kappa, double c_sw) {
  hamiltonian_field_t hf;
  (void)tmref_derivative();
  hf.gaugefield = g_gauge_field; hf.momenta = NULL; hf.derivative = tmref_dfp; hf.update_gauge_copy = 0; hf.traj_counter = 0;
  sw_all(&hf, kappa, c_sw);
}
#endif
void tmref_deriv_Sb(int ieo, spinor *l, spinor *k, double factor) {
  hamiltonian_field_t hf;
  (void)tmref_derivative();
  hf.gaugefield = g_gauge_field; hf.momenta = NULL; hf.derivative = tmref_dfp; hf.update_gauge_copy = 0; hf.traj_counter = 0;
  deriv_Sb(ieo, l, k, &hf, factor);
}
#endif

void tmref_set_theta(double x0, double x1, double x2, double x3) {
  X0 = x0; X1 = x1; X2 = x2; X3 = x3;
  boundary(g_kappa);
}

void tmref_set_kappa_mu(double kappa, double mu) {
  g_kappa = kappa; g_mu = mu;
  boundary(g_kappa);
}

/* benchmark.c:247-259 */
void tmref_random_fields(int seed) {
  start_ranlux(1, seed);
  random_gauge_field(1, g_gauge_field);
  random_spinor_field_eo(g_spinor_field[0], 1, RN_GAUSS);
}

void tmref_random_spinor_eo(int i) { random_spinor_field_eo(g_spinor_field[i], 1, RN_GAUSS); }

#ifndef TMREF_NO_CLOVER
/* clover term and its inverse on the even sites, as operator.c:329-330,364 prepare them for invert_clover_eo */
void tmref_clover(double c_sw, double mu) {
  g_c_sw = c_sw; g_mu = mu; g_mu3 = 0.;
  init_sw_fields();
  sw_term((const su3 **)g_gauge_field, g_kappa, g_c_sw);
  sw_invert(0 /* EE, operator/Hopping_Matrix.h:26 */, g_mu);
  g_c_sw = 0.;   /* keep D_psi on its non-clover branch (D_psi_body.c:314-316) */
}
su3 *tmref_sw(void) { return &sw[0][0][0]; }
su3 *tmref_sw_inv(void) { return &sw_inv[0][0][0]; }
#endif

su3 *tmref_gauge(void) { return &g_gauge_field[0][0]; }
spinor *tmref_spinor(int i) { return g_spinor_field[i]; }
int *tmref_hi(void) { return g_hi; }
int *tmref_eo2lexic(void) { return g_eo2lexic; }
int *tmref_lexic2eosub(void) { return g_lexic2eosub; }
void tmref_mark_gauge_dirty(void) { g_update_gauge_copy = 1; }
int tmref_threads(void) {
#ifdef TM_USE_OMP
  return omp_num_threads;
#else
  return 1;
#endif
}
