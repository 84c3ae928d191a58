/* A C host program linked AT LINK TIME against the drop-in library, in the shape of the reference's benchmark.c
 * (benchmark.c:262-331: j_max x { Hopping_Matrix(0, f1, f0); Hopping_Matrix(1, f2, f1); }, Mflops = 1608 / (us per site)).
 * It owns the tmLQCD globals (../host_stub/globals.c is compiled into it), allocates host AoS fields, calls the
 * reference-named symbols and checks the result against the CPU oracle (test infrastructure: this program is a test).
 *
 *   gcc -O2 -std=gnu99 mini_benchmark.c ../host_stub/globals.c -I../../include -I../../oracle \
 *       -L../../tmlqcd_amd/lib -ltmlqcd_dropin -ltmlqcd_hip -L../../oracle -ltmoracle -lm -o mini_benchmark
 *
 * Usage: mini_benchmark T L iterations      (prints one JSON line)
 */
#include <complex.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "tmlqcd_dropin.h"
#include "tm_oracle.h"

/* the host program's side of the boundary (tests/host_stub/globals.c) */
extern int T, LX, LY, LZ, VOLUME;
su3 *stub_init(int T_, int LX_, int LY_, int LZ_);
void stub_boundary(double kappa, double x0, double x1, double x2, double x3);
void stub_set_mu(double mu);

static unsigned long long rng_state = 88172645463325252ULL;
static double rnd(void) {   /* xorshift64*, uniform in (-1, 1) */
  rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
  return (double)((rng_state * 2685821657736338717ULL) >> 11) / 4503599627370496.0 - 1.0;
}
/* two random vectors -> Gram-Schmidt -> third row = conj(cross product): the construction of start.c:387-425 */
static void random_su3(su3 *u) {
  double _Complex a[3], b[3], c[3], dot = 0;
  double na = 0, nb = 0;
  for (int i = 0; i < 3; i++) { a[i] = rnd() + I * rnd(); b[i] = rnd() + I * rnd(); na += creal(a[i] * conj(a[i])); }
  for (int i = 0; i < 3; i++) a[i] /= sqrt(na);
  for (int i = 0; i < 3; i++) dot += conj(a[i]) * b[i];
  for (int i = 0; i < 3; i++) { b[i] -= dot * a[i]; nb += creal(b[i] * conj(b[i])); }
  for (int i = 0; i < 3; i++) b[i] /= sqrt(nb);
  c[0] = conj(a[1] * b[2] - a[2] * b[1]); c[1] = conj(a[2] * b[0] - a[0] * b[2]); c[2] = conj(a[0] * b[1] - a[1] * b[0]);
  u->c00 = a[0]; u->c01 = a[1]; u->c02 = a[2]; u->c10 = b[0]; u->c11 = b[1]; u->c12 = b[2]; u->c20 = c[0]; u->c21 = c[1]; u->c22 = c[2];
}
static double now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

int main(int argc, char **argv) {
  const int Tt = argc > 1 ? atoi(argv[1]) : 8, L = argc > 2 ? atoi(argv[2]) : 8, j_max = argc > 3 ? atoi(argv[3]) : 10;
  const double kappa = 0.125;
  su3 *gauge = stub_init(Tt, L, L, L);
  for (int i = 0; i < 4 * VOLUME; i++) random_su3(gauge + i);
  stub_boundary(kappa, 0., 0., 0., 0.);
  stub_set_mu(0.01);
  const int N = VOLUME / 2;
  spinor *f0 = calloc(N, sizeof(spinor)), *f1 = calloc(N, sizeof(spinor)), *f2 = calloc(N, sizeof(spinor));
  for (int i = 0; i < N * 24; i++) ((double *)f0)[i] = rnd();

  /* (1) plain drop-in use, coherent mode: every call returns with the host array filled (PCIe both ways) */
  double t0 = now();
  for (int j = 0; j < j_max; j++) { Hopping_Matrix(0, f1, f0); Hopping_Matrix(1, f2, f1); }   /* benchmark.c:295-296 */
  const double t_coh = now() - t0;
  const double n2 = square_norm(f2, N, 0);

  /* parity against the CPU restatement of the reference on the same host arrays */
  tmo_lattice *o = tmo_create(Tt, L, L, L, 1, 0);
  const double theta[4] = {0, 0, 0, 0};
  tmo_boundary(o, kappa, theta);
  tmo_set_gauge(o, (const tmo_su3 *)gauge);
  tmo_spinor *r1 = calloc(N, sizeof(spinor)), *r2 = calloc(N, sizeof(spinor));
  tmo_Hopping_Matrix(o, 0, r1, (const tmo_spinor *)f0);
  tmo_Hopping_Matrix(o, 1, r2, r1);
  double maxd = 0, maxr = 0;
  for (int i = 0; i < N * 24; i++) {
    const double d = fabs(((double *)f2)[i] - ((double *)r2)[i]), a = fabs(((double *)r2)[i]);
    if (d > maxd) maxd = d;
    if (a > maxr) maxr = a;
  }

  /* (2) the same loop with the fields resident in HBM (tmlqcd_hip_benchmark_loop = benchmark.c:291-300 on the device) */
  const double t_res = tmlqcd_hip_benchmark_loop(f0, f1, f2, j_max);
  const double sdt_coh = 1e6 * t_coh / ((double)j_max * VOLUME), sdt_res = 1e6 * t_res / ((double)j_max * VOLUME);   /* benchmark.c:318 */
  printf("{\"T\": %d, \"L\": %d, \"iterations\": %d, \"norm_out\": %.15e, \"max_rel_err_vs_oracle\": %.3e, "
         "\"mflops_coherent\": %.1f, \"mflops_resident\": %.1f}\n",
         Tt, L, j_max, n2, maxd / maxr, 1608.0 / sdt_coh, 1608.0 / sdt_res);   /* benchmark.c:327 */
  tmlqcd_hip_finalize();
  return maxd / maxr < 1e-13 ? 0 : 1;
}
