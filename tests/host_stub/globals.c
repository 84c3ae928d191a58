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
void stub_boundary(double kappa, double x0, double x1, double x2, double x3) {
  const double PI_ = 3.14159265358979;
  g_kappa = kappa;
  ka0 = kappa * cexp(x0 * PI_ / T * I); ka1 = kappa * cexp(x1 * PI_ / LX * I);
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
