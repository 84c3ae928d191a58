/* TEST INFRASTRUCTURE ONLY -- CPU oracle for the tmLQCD hot path (see tm_oracle.h).
 *
 * Restates, in plain C99, the algorithm of the reference's generic-C code path
 * (_GAUGE_COPY build, no SSE/BG intrinsics).  Every function cites the reference
 * file:line it follows (paths relative to /root/reference).  Floating-point
 * operation order follows the reference macros so that, built without FMA
 * contraction, results agree with oracle/_ref/libtmref.so to the last bit or two.
 */
#include "tm_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TMO_MAX_THREADS 1024
static int tmo_threads = 1;

void tmo_set_threads(int n) {
  if (n < 1) n = 1;
  if (n > TMO_MAX_THREADS) n = TMO_MAX_THREADS;
  tmo_threads = n;
#ifdef _OPENMP
  omp_set_num_threads(n);
#endif
}
int tmo_get_threads(void) { return tmo_threads; }

/* ---------------------------------------------------------------- su3.h macros */
/* su3.h:197-217 */
static inline tmo_su3_vector v_add(tmo_su3_vector a, tmo_su3_vector b) {
  tmo_su3_vector r = {a.c0 + b.c0, a.c1 + b.c1, a.c2 + b.c2}; return r; }
static inline tmo_su3_vector v_sub(tmo_su3_vector a, tmo_su3_vector b) {
  tmo_su3_vector r = {a.c0 - b.c0, a.c1 - b.c1, a.c2 - b.c2}; return r; }
static inline tmo_su3_vector v_i_add(tmo_su3_vector a, tmo_su3_vector b) {
  tmo_su3_vector r = {a.c0 + I * b.c0, a.c1 + I * b.c1, a.c2 + I * b.c2}; return r; }
static inline tmo_su3_vector v_i_sub(tmo_su3_vector a, tmo_su3_vector b) {
  tmo_su3_vector r = {a.c0 - I * b.c0, a.c1 - I * b.c1, a.c2 - I * b.c2}; return r; }
/* su3.h:272-275 */
static inline tmo_su3_vector c_times_v(double _Complex c, tmo_su3_vector s) {
  tmo_su3_vector r = {c * s.c0, c * s.c1, c * s.c2}; return r; }
static inline tmo_su3_vector cc_times_v(double _Complex c, tmo_su3_vector s) {
  tmo_su3_vector r = {conj(c) * s.c0, conj(c) * s.c1, conj(c) * s.c2}; return r; }
/* su3.h:308-311 */
static inline tmo_su3_vector su3_mul(const tmo_su3 *u, tmo_su3_vector s) {
  tmo_su3_vector r;
  r.c0 = u->c00 * s.c0 + u->c01 * s.c1 + u->c02 * s.c2;
  r.c1 = u->c10 * s.c0 + u->c11 * s.c1 + u->c12 * s.c2;
  r.c2 = u->c20 * s.c0 + u->c21 * s.c1 + u->c22 * s.c2;
  return r;
}
/* su3.h:313-316 */
static inline tmo_su3_vector su3_inv_mul(const tmo_su3 *u, tmo_su3_vector s) {
  tmo_su3_vector r;
  r.c0 = conj(u->c00) * s.c0 + conj(u->c10) * s.c1 + conj(u->c20) * s.c2;
  r.c1 = conj(u->c01) * s.c0 + conj(u->c11) * s.c1 + conj(u->c21) * s.c2;
  r.c2 = conj(u->c02) * s.c0 + conj(u->c12) * s.c1 + conj(u->c22) * s.c2;
  return r;
}
#define V_ADD_ASSIGN(r, s)   do { (r).c0 += (s).c0; (r).c1 += (s).c1; (r).c2 += (s).c2; } while (0)
#define V_SUB_ASSIGN(r, s)   do { (r).c0 -= (s).c0; (r).c1 -= (s).c1; (r).c2 -= (s).c2; } while (0)
#define V_IADD_ASSIGN(r, s)  do { (r).c0 += I * (s).c0; (r).c1 += I * (s).c1; (r).c2 += I * (s).c2; } while (0)
#define V_ISUB_ASSIGN(r, s)  do { (r).c0 -= I * (s).c0; (r).c1 -= I * (s).c1; (r).c2 -= I * (s).c2; } while (0)

/* ---------------------------------------------------------------- geometry */
/* geometry_eo.c:279-299 (the "original" Index(): none / PARALLELT) */
int tmo_index(const tmo_lattice *lat, int x0, int x1, int x2, int x3) {
  const int T = lat->T, LX = lat->LX, LY = lat->LY, LZ = lat->LZ;
  int y0 = (x0 + T) % T, y1 = (x1 + LX) % LX, y2 = (x2 + LY) % LY, y3 = (x3 + LZ) % LZ;
  int ix = ((y0 * LX + y1) * LY + y2) * LZ + y3;
  if (lat->nproc_t > 1) {
    if (x0 == T) ix = lat->V + y3 + LZ * y2 + LZ * LY * y1;
    else if (x0 == -1) ix = lat->V + LX * LY * LZ + y3 + LZ * y2 + LZ * LY * y1;
  }
  return ix;
}

/* geometry_eo.c:743-885 (index tables) and :1470-1535 (Hopping_Matrix_Indices) */
static void tmo_build_geometry(tmo_lattice *lat) {
  const int T = lat->T, LX = lat->LX, LY = lat->LY, LZ = lat->LZ, V = lat->V, VR = lat->V + lat->RAND;
  const int st = lat->nproc_t > 1 ? 1 : 0;
  int *xeven = (int *)malloc(sizeof(int) * VR);
  for (int x0 = -st; x0 < T + st; x0++)
    for (int x1 = 0; x1 < LX; x1++)
      for (int x2 = 0; x2 < LY; x2++)
        for (int x3 = 0; x3 < LZ; x3++) {
          int ix = tmo_index(lat, x0, x1, x2, x3);
          int s = x0 + x1 + x2 + x3 + lat->proc_t * T;       /* geometry_eo.c:807-811 */
          xeven[ix] = (((s % 2) + 2) % 2 == 0);
          if (ix < V) {
            lat->iup[4 * ix + 0] = tmo_index(lat, x0 + 1, x1, x2, x3);
            lat->idn[4 * ix + 0] = tmo_index(lat, x0 - 1, x1, x2, x3);
            lat->iup[4 * ix + 1] = tmo_index(lat, x0, x1 + 1, x2, x3);
            lat->idn[4 * ix + 1] = tmo_index(lat, x0, x1 - 1, x2, x3);
            lat->iup[4 * ix + 2] = tmo_index(lat, x0, x1, x2 + 1, x3);
            lat->idn[4 * ix + 2] = tmo_index(lat, x0, x1, x2 - 1, x3);
            lat->iup[4 * ix + 3] = tmo_index(lat, x0, x1, x2, x3 + 1);
            lat->idn[4 * ix + 3] = tmo_index(lat, x0, x1, x2, x3 - 1);
          }
        }
  int i_even = 0, i_odd = 0;                                  /* geometry_eo.c:869-885 */
  for (int ix = 0; ix < VR; ix++) {
    if (xeven[ix]) {
      lat->lexic2eo[ix] = i_even; lat->lexic2eosub[ix] = i_even; lat->eo2lexic[i_even] = ix; i_even++;
    } else {
      lat->lexic2eo[ix] = VR / 2 + i_odd; lat->lexic2eosub[ix] = i_odd; lat->eo2lexic[VR / 2 + i_odd] = ix; i_odd++;
    }
  }
  free(xeven);
  for (int par = 0; par < 2; par++)                           /* geometry_eo.c:1470-1535 */
    for (int i = 0; i < V / 2; i++) {
      int ic = par * (VR / 2) + i, ix = lat->eo2lexic[ic];
      int *h = lat->hi + 16 * ic;
      for (int mu = 0; mu < 4; mu++) {
        h[4 * mu + 0] = lat->iup[4 * ix + mu];
        h[4 * mu + 1] = lat->lexic2eosub[lat->iup[4 * ix + mu]];
        h[4 * mu + 2] = lat->idn[4 * ix + mu];
        h[4 * mu + 3] = lat->lexic2eosub[lat->idn[4 * ix + mu]];
      }
      h[0] = ix;
    }
  lat->hi[16 * VR] = 0; lat->hi[16 * VR + 1] = 0;
}

tmo_lattice *tmo_create(int T, int LX, int LY, int LZ, int nproc_t, int proc_t) {
  tmo_lattice *lat = (tmo_lattice *)calloc(1, sizeof(*lat));
  lat->T = T; lat->LX = LX; lat->LY = LY; lat->LZ = LZ;
  lat->nproc_t = nproc_t < 1 ? 1 : nproc_t; lat->proc_t = proc_t;
  lat->V = T * LX * LY * LZ;
  lat->RAND = lat->nproc_t > 1 ? 2 * LX * LY * LZ : 0;       /* mpi_init.c:330-332 */
  lat->VPR = lat->V + lat->RAND;
  const int VR = lat->VPR;
  lat->iup = (int *)calloc(4 * (size_t)VR, sizeof(int));
  lat->idn = (int *)calloc(4 * (size_t)VR, sizeof(int));
  lat->lexic2eo = (int *)calloc(VR, sizeof(int));
  lat->lexic2eosub = (int *)calloc(VR, sizeof(int));
  lat->eo2lexic = (int *)calloc(VR, sizeof(int));
  lat->hi = (int *)calloc(16 * (size_t)VR + 2, sizeof(int));
  lat->gauge_copy = (tmo_su3 *)calloc(8 * (size_t)VR + 1, sizeof(tmo_su3));
  for (int i = 0; i < 3; i++) lat->scratch[i] = (tmo_spinor *)calloc((size_t)VR / 2 + 1, sizeof(tmo_spinor));
  tmo_build_geometry(lat);
  double th[4] = {0, 0, 0, 0};
  tmo_boundary(lat, 0.125, th);
  lat->gauge_dirty = 1;
  return lat;
}

void tmo_destroy(tmo_lattice *lat) {
  if (!lat) return;
  free(lat->iup); free(lat->idn); free(lat->lexic2eo); free(lat->lexic2eosub); free(lat->eo2lexic);
  free(lat->hi); free(lat->gauge_copy);
  for (int i = 0; i < 3; i++) free(lat->scratch[i]);
  free(lat);
}

/* boundary.c:36,40-55 */
void tmo_boundary(tmo_lattice *lat, double kappa, const double theta[4]) {
  const double PI_ = 3.14159265358979;
  const int ext[4] = {lat->T * lat->nproc_t, lat->LX, lat->LY, lat->LZ};
  lat->kappa = kappa;
  for (int mu = 0; mu < 4; mu++) {
    lat->theta[mu] = theta[mu];
    double x = theta[mu] * PI_ / ext[mu];
    lat->ka[mu] = kappa * cexp(x * I);
  }
}
void tmo_set_mu(tmo_lattice *lat, double mu) { lat->mu = mu; }
void tmo_set_mu3(tmo_lattice *lat, double mu3) { lat->mu3 = mu3; }
void tmo_set_gauge(tmo_lattice *lat, const tmo_su3 *g) { lat->gauge = g; lat->gauge_dirty = 1; }

/* update_backward_gauge.c:185-242 (plain _GAUGE_COPY layout [V+RAND][8]) */
static void tmo_update_backward_gauge(tmo_lattice *lat) {
  const int V = lat->V, VR = lat->VPR;
  const tmo_su3 *gf = lat->gauge;
  for (int par = 0; par < 2; par++) {
#pragma omp parallel for
    for (int i = 0; i < V / 2; i++) {
      int ix = par * (VR / 2) + i, kb2 = lat->eo2lexic[ix];
      for (int mu = 0; mu < 4; mu++) {
        int kb = lat->idn[4 * kb2 + mu];
        lat->gauge_copy[8 * (size_t)ix + 2 * mu] = gf[4 * (size_t)kb2 + mu];
        lat->gauge_copy[8 * (size_t)ix + 2 * mu + 1] = gf[4 * (size_t)kb + mu];
      }
    }
  }
  lat->gauge_dirty = 0;
}

/* ---------------------------------------------------------------- stencil */
enum { EPI_STORE = 0, EPI_TM_TIMES = 1, EPI_TM_SUB = 2 };

/* operator/hopping_body_dbl.c:27-181 with the generic macros of
   operator/hopping.h:574-694.  NB the body's "sp/up" pair serves the +mu hops
   and "sm/um" the -mu hops; with _GAUGE_COPY the links are the 8 consecutive
   entries of gauge_copy[icx]. */
static void tmo_hopping_generic(tmo_lattice *lat, int ieo, tmo_spinor *l, const tmo_spinor *p,
                                const tmo_spinor *k, double _Complex cfactor, int epi) {
  if (lat->gauge_dirty) tmo_update_backward_gauge(lat);     /* Hopping_Matrix.c:135-139 */
  const int ioff = ieo == 0 ? 0 : lat->VPR / 2;             /* hopping_body_dbl.c:43-48 */
  const int Vh = lat->V / 2;
  const double _Complex ka0 = lat->ka[0], ka1 = lat->ka[1], ka2 = lat->ka[2], ka3 = lat->ka[3];
#pragma omp parallel for
  for (int icx = ioff; icx < Vh + ioff; icx++) {
    const int *hi = lat->hi + 16 * (size_t)icx;
    const tmo_su3 *u = lat->gauge_copy + 8 * (size_t)icx;
    const tmo_spinor *s;
    tmo_su3_vector psi, chi;
    tmo_spinor temp;
    /* +t  hopping.h:578-588 */
    s = k + hi[1];
    psi = v_add(s->s0, s->s2); chi = su3_mul(&u[0], psi); psi = c_times_v(ka0, chi);
    temp.s0 = psi; temp.s2 = psi;
    psi = v_add(s->s1, s->s3); chi = su3_mul(&u[0], psi); psi = c_times_v(ka0, chi);
    temp.s1 = psi; temp.s3 = psi;
    /* -t  hopping.h:590-600 */
    s = k + hi[3];
    psi = v_sub(s->s0, s->s2); chi = su3_inv_mul(&u[1], psi); psi = cc_times_v(ka0, chi);
    V_ADD_ASSIGN(temp.s0, psi); V_SUB_ASSIGN(temp.s2, psi);
    psi = v_sub(s->s1, s->s3); chi = su3_inv_mul(&u[1], psi); psi = cc_times_v(ka0, chi);
    V_ADD_ASSIGN(temp.s1, psi); V_SUB_ASSIGN(temp.s3, psi);
    /* +x  hopping.h:602-612 */
    s = k + hi[5];
    psi = v_i_add(s->s0, s->s3); chi = su3_mul(&u[2], psi); psi = c_times_v(ka1, chi);
    V_ADD_ASSIGN(temp.s0, psi); V_ISUB_ASSIGN(temp.s3, psi);
    psi = v_i_add(s->s1, s->s2); chi = su3_mul(&u[2], psi); psi = c_times_v(ka1, chi);
    V_ADD_ASSIGN(temp.s1, psi); V_ISUB_ASSIGN(temp.s2, psi);
    /* -x  hopping.h:614-624 */
    s = k + hi[7];
    psi = v_i_sub(s->s0, s->s3); chi = su3_inv_mul(&u[3], psi); psi = cc_times_v(ka1, chi);
    V_ADD_ASSIGN(temp.s0, psi); V_IADD_ASSIGN(temp.s3, psi);
    psi = v_i_sub(s->s1, s->s2); chi = su3_inv_mul(&u[3], psi); psi = cc_times_v(ka1, chi);
    V_ADD_ASSIGN(temp.s1, psi); V_IADD_ASSIGN(temp.s2, psi);
    /* +y  hopping.h:626-636 */
    s = k + hi[9];
    psi = v_add(s->s0, s->s3); chi = su3_mul(&u[4], psi); psi = c_times_v(ka2, chi);
    V_ADD_ASSIGN(temp.s0, psi); V_ADD_ASSIGN(temp.s3, psi);
    psi = v_sub(s->s1, s->s2); chi = su3_mul(&u[4], psi); psi = c_times_v(ka2, chi);
    V_ADD_ASSIGN(temp.s1, psi); V_SUB_ASSIGN(temp.s2, psi);
    /* -y  hopping.h:638-648 */
    s = k + hi[11];
    psi = v_sub(s->s0, s->s3); chi = su3_inv_mul(&u[5], psi); psi = cc_times_v(ka2, chi);
    V_ADD_ASSIGN(temp.s0, psi); V_SUB_ASSIGN(temp.s3, psi);
    psi = v_add(s->s1, s->s2); chi = su3_inv_mul(&u[5], psi); psi = cc_times_v(ka2, chi);
    V_ADD_ASSIGN(temp.s1, psi); V_ADD_ASSIGN(temp.s2, psi);
    /* +z  hopping.h:650-660 */
    s = k + hi[13];
    psi = v_i_add(s->s0, s->s2); chi = su3_mul(&u[6], psi); psi = c_times_v(ka3, chi);
    V_ADD_ASSIGN(temp.s0, psi); V_ISUB_ASSIGN(temp.s2, psi);
    psi = v_i_sub(s->s1, s->s3); chi = su3_mul(&u[6], psi); psi = c_times_v(ka3, chi);
    V_ADD_ASSIGN(temp.s1, psi); V_IADD_ASSIGN(temp.s3, psi);
    /* -z  hopping.h:662-672 */
    s = k + hi[15];
    psi = v_i_sub(s->s0, s->s2); chi = su3_inv_mul(&u[7], psi); psi = cc_times_v(ka3, chi);
    V_ADD_ASSIGN(temp.s0, psi); V_IADD_ASSIGN(temp.s2, psi);
    psi = v_i_add(s->s1, s->s3); chi = su3_inv_mul(&u[7], psi); psi = cc_times_v(ka3, chi);
    V_ADD_ASSIGN(temp.s1, psi); V_ISUB_ASSIGN(temp.s3, psi);

    tmo_spinor *rn = l + (icx - ioff);
    if (epi == EPI_TM_TIMES) {                               /* hopping.h:674-678 */
      rn->s0 = c_times_v(cfactor, temp.s0); rn->s1 = c_times_v(cfactor, temp.s1);
      rn->s2 = cc_times_v(cfactor, temp.s2); rn->s3 = cc_times_v(cfactor, temp.s3);
    } else if (epi == EPI_TM_SUB) {                          /* hopping.h:680-688 */
      const tmo_spinor *pn = p + (icx - ioff);
      psi = c_times_v(cfactor, pn->s0); rn->s0 = v_sub(psi, temp.s0);
      chi = c_times_v(cfactor, pn->s1); rn->s1 = v_sub(chi, temp.s1);
      psi = cc_times_v(cfactor, pn->s2); rn->s2 = v_sub(temp.s2, psi);
      chi = cc_times_v(cfactor, pn->s3); rn->s3 = v_sub(temp.s3, chi);
    } else {                                                 /* hopping.h:690-694 */
      *rn = temp;
    }
  }
}

/* operator/Hopping_Matrix.c:131-156 (halo of k must already be filled: the oracle has no MPI) */
void tmo_Hopping_Matrix(tmo_lattice *lat, int ieo, tmo_spinor *l, const tmo_spinor *k) {
  tmo_hopping_generic(lat, ieo, l, NULL, k, 0, EPI_STORE);
}
/* operator/tm_times_Hopping_Matrix.c:72-153 */
void tmo_tm_times_Hopping_Matrix(tmo_lattice *lat, int ieo, tmo_spinor *l, const tmo_spinor *k,
                                 double cre, double cim) {
  tmo_hopping_generic(lat, ieo, l, NULL, k, cre + cim * I, EPI_TM_TIMES);
}
/* operator/tm_sub_Hopping_Matrix.c:73-157 */
void tmo_tm_sub_Hopping_Matrix(tmo_lattice *lat, int ieo, tmo_spinor *l, const tmo_spinor *p,
                               const tmo_spinor *k, double cre, double cim) {
  tmo_hopping_generic(lat, ieo, l, p, k, cre + cim * I, EPI_TM_SUB);
}

/* operator/D_psi_body.c:266-375 with the p?add/m?add helpers of :1-230; g_c_sw = 0 branch */
void tmo_D_psi(tmo_lattice *lat, tmo_spinor *P, const tmo_spinor *Q) {
  if (P == Q) {                                              /* D_psi_body.c:267-272 */
    printf("Error in D_psi (operator.c):\nArguments must be different spinor fields\nProgram aborted\n");
    exit(1);
  }
  const double _Complex ph0 = -lat->ka[0], ph1 = -lat->ka[1], ph2 = -lat->ka[2], ph3 = -lat->ka[3]; /* boundary.c:51-54 */
  const double _Complex rho1 = 1. + lat->mu * I, rho2 = conj(rho1);
  const tmo_su3 *g = lat->gauge;
#pragma omp parallel for
  for (int ix = 0; ix < lat->V; ix++) {
    const tmo_spinor *s = Q + ix;
    tmo_spinor t;
    tmo_su3_vector psi, chi;
    const tmo_su3 *u;
    int iy;
    t.s0 = c_times_v(rho1, s->s0); t.s1 = c_times_v(rho1, s->s1);
    t.s2 = c_times_v(rho2, s->s2); t.s3 = c_times_v(rho2, s->s3);
    /* +0 */
    iy = lat->iup[4 * ix + 0]; s = Q + iy; u = &g[4 * (size_t)ix + 0];
    psi = v_add(s->s0, s->s2); chi = su3_mul(u, psi); psi = c_times_v(ph0, chi);
    V_ADD_ASSIGN(t.s0, psi); V_ADD_ASSIGN(t.s2, psi);
    psi = v_add(s->s1, s->s3); chi = su3_mul(u, psi); psi = c_times_v(ph0, chi);
    V_ADD_ASSIGN(t.s1, psi); V_ADD_ASSIGN(t.s3, psi);
    /* -0 */
    iy = lat->idn[4 * ix + 0]; s = Q + iy; u = &g[4 * (size_t)iy + 0];
    psi = v_sub(s->s0, s->s2); chi = su3_inv_mul(u, psi); psi = cc_times_v(ph0, chi);
    V_ADD_ASSIGN(t.s0, psi); V_SUB_ASSIGN(t.s2, psi);
    psi = v_sub(s->s1, s->s3); chi = su3_inv_mul(u, psi); psi = cc_times_v(ph0, chi);
    V_ADD_ASSIGN(t.s1, psi); V_SUB_ASSIGN(t.s3, psi);
    /* +1 */
    iy = lat->iup[4 * ix + 1]; s = Q + iy; u = &g[4 * (size_t)ix + 1];
    psi = v_i_add(s->s0, s->s3); chi = su3_mul(u, psi); psi = c_times_v(ph1, chi);
    V_ADD_ASSIGN(t.s0, psi); V_ISUB_ASSIGN(t.s3, psi);
    psi = v_i_add(s->s1, s->s2); chi = su3_mul(u, psi); psi = c_times_v(ph1, chi);
    V_ADD_ASSIGN(t.s1, psi); V_ISUB_ASSIGN(t.s2, psi);
    /* -1 */
    iy = lat->idn[4 * ix + 1]; s = Q + iy; u = &g[4 * (size_t)iy + 1];
    psi = v_i_sub(s->s0, s->s3); chi = su3_inv_mul(u, psi); psi = cc_times_v(ph1, chi);
    V_ADD_ASSIGN(t.s0, psi); V_IADD_ASSIGN(t.s3, psi);
    psi = v_i_sub(s->s1, s->s2); chi = su3_inv_mul(u, psi); psi = cc_times_v(ph1, chi);
    V_ADD_ASSIGN(t.s1, psi); V_IADD_ASSIGN(t.s2, psi);
    /* +2 */
    iy = lat->iup[4 * ix + 2]; s = Q + iy; u = &g[4 * (size_t)ix + 2];
    psi = v_add(s->s0, s->s3); chi = su3_mul(u, psi); psi = c_times_v(ph2, chi);
    V_ADD_ASSIGN(t.s0, psi); V_ADD_ASSIGN(t.s3, psi);
    psi = v_sub(s->s1, s->s2); chi = su3_mul(u, psi); psi = c_times_v(ph2, chi);
    V_ADD_ASSIGN(t.s1, psi); V_SUB_ASSIGN(t.s2, psi);
    /* -2 */
    iy = lat->idn[4 * ix + 2]; s = Q + iy; u = &g[4 * (size_t)iy + 2];
    psi = v_sub(s->s0, s->s3); chi = su3_inv_mul(u, psi); psi = cc_times_v(ph2, chi);
    V_ADD_ASSIGN(t.s0, psi); V_SUB_ASSIGN(t.s3, psi);
    psi = v_add(s->s1, s->s2); chi = su3_inv_mul(u, psi); psi = cc_times_v(ph2, chi);
    V_ADD_ASSIGN(t.s1, psi); V_ADD_ASSIGN(t.s2, psi);
    /* +3 */
    iy = lat->iup[4 * ix + 3]; s = Q + iy; u = &g[4 * (size_t)ix + 3];
    psi = v_i_add(s->s0, s->s2); chi = su3_mul(u, psi); psi = c_times_v(ph3, chi);
    V_ADD_ASSIGN(t.s0, psi); V_ISUB_ASSIGN(t.s2, psi);
    psi = v_i_sub(s->s1, s->s3); chi = su3_mul(u, psi); psi = c_times_v(ph3, chi);
    V_ADD_ASSIGN(t.s1, psi); V_IADD_ASSIGN(t.s3, psi);
    /* -3 (m3addandstore, D_psi_body.c:206-230) */
    iy = lat->idn[4 * ix + 3]; s = Q + iy; u = &g[4 * (size_t)iy + 3];
    tmo_spinor *r = P + ix;
    psi = v_i_sub(s->s0, s->s2); chi = su3_inv_mul(u, psi); psi = cc_times_v(ph3, chi);
    r->s0 = v_add(t.s0, psi); r->s2 = v_i_add(t.s2, psi);
    psi = v_i_add(s->s1, s->s3); chi = su3_inv_mul(u, psi); psi = cc_times_v(ph3, chi);
    r->s1 = v_add(t.s1, psi); r->s3 = v_i_sub(t.s3, psi);
  }
}

/* ---------------------------------------------------------------- site-diagonal */
/* operator/mul_one_pm_imu_inv_body.c:1-41 */
void tmo_mul_one_pm_imu_inv(tmo_lattice *lat, tmo_spinor *l, double _sign, int N) {
  double nrm = 1. / (1. + lat->mu * lat->mu), sign = -1.;
  if (_sign < 0.) sign = 1.;
  double _Complex z = nrm + (sign * nrm * lat->mu) * I, w = conj(z);
#pragma omp parallel for
  for (int ix = 0; ix < N; ix++) {
    tmo_spinor *r = l + ix;
    r->s0 = c_times_v(z, r->s0); r->s1 = c_times_v(z, r->s1);
    r->s2 = c_times_v(w, r->s2); r->s3 = c_times_v(w, r->s3);
  }
}
/* operator/mul_one_pm_imu_inv_body.c:43-80 */
void tmo_assign_mul_one_pm_imu_inv(tmo_lattice *lat, tmo_spinor *l, const tmo_spinor *k, double _sign, int N) {
  double nrm = 1. / (1. + lat->mu * lat->mu), sign = -1.;
  if (_sign < 0.) sign = 1.;
  double _Complex z = nrm + (sign * nrm * lat->mu) * I, w = conj(z);
#pragma omp parallel for
  for (int ix = 0; ix < N; ix++) {
    const tmo_spinor *r = k + ix; tmo_spinor *s = l + ix;
    tmo_spinor o;
    o.s0 = c_times_v(z, r->s0); o.s1 = c_times_v(z, r->s1);
    o.s2 = c_times_v(w, r->s2); o.s3 = c_times_v(w, r->s3);
    *s = o;
  }
}
/* operator/tm_operators.c:669-720 */
void tmo_assign_mul_one_pm_imu(tmo_lattice *lat, tmo_spinor *l, const tmo_spinor *k, double _sign, int N) {
  double sign = 1.;
  if (_sign < 0.) sign = -1.;
  double _Complex z = 1. + (sign * lat->mu) * I, w = conj(z);
#pragma omp parallel for
  for (int ix = 0; ix < N; ix++) {
    const tmo_spinor *r = k + ix; tmo_spinor *s = l + ix;
    tmo_spinor o;
    o.s0 = c_times_v(z, r->s0); o.s1 = c_times_v(z, r->s1);
    o.s2 = c_times_v(w, r->s2); o.s3 = c_times_v(w, r->s3);
    *s = o;
  }
}
/* operator/mul_one_pm_imu_sub_mul_body.c:1-48 */
void tmo_mul_one_pm_imu_sub_mul(tmo_lattice *lat, tmo_spinor *l, const tmo_spinor *k, const tmo_spinor *j,
                                double _sign, int N) {
  double sign = 1.;
  if (_sign < 0.) sign = -1.;
  double _Complex z = 1. + (sign * lat->mu) * I, w = conj(z);
#pragma omp parallel for
  for (int ix = 0; ix < N; ix++) {
    const tmo_spinor *r = k + ix, *s = j + ix; tmo_spinor *t = l + ix;
    tmo_su3_vector p1 = c_times_v(z, r->s0), p2 = c_times_v(z, r->s1), p3 = c_times_v(w, r->s2), p4 = c_times_v(w, r->s3);
    tmo_spinor o;
    o.s0 = v_sub(p1, s->s0); o.s1 = v_sub(p2, s->s1); o.s2 = v_sub(p3, s->s2); o.s3 = v_sub(p4, s->s3);
    *t = o;
  }
}
/* operator/tm_operators.c:813-858 */
void tmo_mul_one_pm_imu_sub_mul_gamma5(tmo_lattice *lat, tmo_spinor *l, const tmo_spinor *k,
                                       const tmo_spinor *j, double _sign) {
  double sign = 1.;
  if (_sign < 0.) sign = -1.;
  double _Complex z = 1. + (sign * lat->mu) * I, w = conj(z);
  const int N = lat->V / 2;
#pragma omp parallel for
  for (int ix = 0; ix < N; ix++) {
    const tmo_spinor *r = k + ix, *s = j + ix; tmo_spinor *t = l + ix;
    tmo_su3_vector p1 = c_times_v(z, r->s0), p2 = c_times_v(z, r->s1), p3 = c_times_v(w, r->s2), p4 = c_times_v(w, r->s3);
    tmo_spinor o;
    o.s0 = v_sub(p1, s->s0); o.s1 = v_sub(p2, s->s1); o.s2 = v_sub(s->s2, p3); o.s3 = v_sub(s->s3, p4);
    *t = o;
  }
}
/* gamma.c:77-98 */
void tmo_gamma5(tmo_spinor *l, const tmo_spinor *k, int N) {
#pragma omp parallel for
  for (int ix = 0; ix < N; ix++) {
    tmo_spinor o = k[ix];
    o.s2.c0 = -o.s2.c0; o.s2.c1 = -o.s2.c1; o.s2.c2 = -o.s2.c2;
    o.s3.c0 = -o.s3.c0; o.s3.c1 = -o.s3.c1; o.s3.c2 = -o.s3.c2;
    l[ix] = o;
  }
}

/* ---------------------------------------------------------------- e/o compositions */
#define EO 0
#define OE 1
/* operator/tm_operators.c:508-526 (generic branch) */
void tmo_H_eo_tm_inv_psi(tmo_lattice *lat, tmo_spinor *l, const tmo_spinor *k, int ieo, double _sign) {
  double nrm = 1. / (1. + lat->mu * lat->mu), sign = -1.;
  if (_sign < 0.) sign = 1.;
  tmo_tm_times_Hopping_Matrix(lat, ieo, l, k, nrm, sign * nrm * lat->mu);
}
/* operator/tm_operators.c:528-546 */
static void tmo_tm_sub_H_eo_gamma5(tmo_lattice *lat, tmo_spinor *l, const tmo_spinor *p, const tmo_spinor *k,
                                   int ieo, double _sign) {
  double sign = 1.;
  if (_sign < 0.) sign = -1.;
  tmo_tm_sub_Hopping_Matrix(lat, ieo, l, p, k, 1., sign * lat->mu);
}
/* operator/tm_operators.c:172-177 */
void tmo_Qtm_plus_psi(tmo_lattice *lat, tmo_spinor *l, tmo_spinor *k) {
  tmo_Hopping_Matrix(lat, EO, lat->scratch[1], k);
  tmo_mul_one_pm_imu_inv(lat, lat->scratch[1], +1., lat->V / 2);
  tmo_Hopping_Matrix(lat, OE, lat->scratch[0], lat->scratch[1]);
  tmo_mul_one_pm_imu_sub_mul_gamma5(lat, l, k, lat->scratch[0], +1.);
}
/* operator/tm_operators.c:216-221 */
void tmo_Qtm_minus_psi(tmo_lattice *lat, tmo_spinor *l, tmo_spinor *k) {
  tmo_H_eo_tm_inv_psi(lat, lat->scratch[1], k, EO, -1);
  tmo_Hopping_Matrix(lat, OE, lat->scratch[2], lat->scratch[1]);
  tmo_mul_one_pm_imu_sub_mul_gamma5(lat, l, k, lat->scratch[2], -1);
}
/* operator/tm_operators.c:245-250 */
void tmo_Mtm_plus_psi(tmo_lattice *lat, tmo_spinor *l, tmo_spinor *k) {
  tmo_Hopping_Matrix(lat, EO, lat->scratch[1], k);
  tmo_mul_one_pm_imu_inv(lat, lat->scratch[1], +1., lat->V / 2);
  tmo_Hopping_Matrix(lat, OE, lat->scratch[0], lat->scratch[1]);
  tmo_mul_one_pm_imu_sub_mul(lat, l, k, lat->scratch[0], +1., lat->V / 2);
}
/* operator/tm_operators.c:289-294 */
void tmo_Mtm_minus_psi(tmo_lattice *lat, tmo_spinor *l, tmo_spinor *k) {
  tmo_Hopping_Matrix(lat, EO, lat->scratch[1], k);
  tmo_mul_one_pm_imu_inv(lat, lat->scratch[1], -1., lat->V / 2);
  tmo_Hopping_Matrix(lat, OE, lat->scratch[0], lat->scratch[1]);
  tmo_mul_one_pm_imu_sub_mul(lat, l, k, lat->scratch[0], -1., lat->V / 2);
}
/* operator/tm_operators.c:338-345 */
void tmo_Qtm_pm_psi(tmo_lattice *lat, tmo_spinor *l, tmo_spinor *k) {
  tmo_H_eo_tm_inv_psi(lat, lat->scratch[1], k, EO, -1);
  tmo_tm_sub_H_eo_gamma5(lat, lat->scratch[0], k, lat->scratch[1], OE, -1);
  tmo_H_eo_tm_inv_psi(lat, lat->scratch[1], lat->scratch[0], EO, +1);
  tmo_tm_sub_H_eo_gamma5(lat, l, lat->scratch[0], lat->scratch[1], OE, +1);
}
/* operator/tm_operators.c:781-810 : l = gamma5 (k - j) */
void tmo_mul_one_sub_mul_gamma5(tmo_lattice *lat, tmo_spinor *l, const tmo_spinor *k, const tmo_spinor *j) {
  const int N = lat->V / 2;
#pragma omp parallel for
  for (int ix = 0; ix < N; ix++) {
    const tmo_spinor *r = k + ix, *s = j + ix;
    tmo_spinor o;
    o.s0 = v_sub(r->s0, s->s0); o.s1 = v_sub(r->s1, s->s1); o.s2 = v_sub(s->s2, r->s2); o.s3 = v_sub(s->s3, r->s3);
    l[ix] = o;
  }
}
/* the "symmetric" family: scratch[0] <- A^-1 H_oe A^-1 H_eo k with A = 1 + sign i mu g5
 * (first four statements of operator/tm_operators.c:186-192, 223-229, 259-265, 296-302) */
static void tmo_sym_core(tmo_lattice *lat, tmo_spinor *k, double sign) {
  const int N = lat->V / 2;
  tmo_Hopping_Matrix(lat, EO, lat->scratch[1], k);
  tmo_mul_one_pm_imu_inv(lat, lat->scratch[1], sign, N);
  tmo_Hopping_Matrix(lat, OE, lat->scratch[0], lat->scratch[1]);
  tmo_mul_one_pm_imu_inv(lat, lat->scratch[0], sign, N);
}
/* operator/tm_operators.c:186-192 */
void tmo_Qtm_plus_sym_psi(tmo_lattice *lat, tmo_spinor *l, tmo_spinor *k) {
  tmo_sym_core(lat, k, +1.);
  tmo_mul_one_sub_mul_gamma5(lat, l, k, lat->scratch[0]);
}
/* operator/tm_operators.c:223-229 */
void tmo_Qtm_minus_sym_psi(tmo_lattice *lat, tmo_spinor *l, tmo_spinor *k) {
  tmo_sym_core(lat, k, -1.);
  tmo_mul_one_sub_mul_gamma5(lat, l, k, lat->scratch[0]);
}
/* operator/tm_operators.c:259-265 */
void tmo_Mtm_plus_sym_psi(tmo_lattice *lat, tmo_spinor *l, tmo_spinor *k) {
  tmo_sym_core(lat, k, +1.);
  tmo_diff(l, k, lat->scratch[0], lat->V / 2);
}
/* operator/tm_operators.c:296-302 */
void tmo_Mtm_minus_sym_psi(tmo_lattice *lat, tmo_spinor *l, tmo_spinor *k) {
  tmo_sym_core(lat, k, -1.);
  tmo_diff(l, k, lat->scratch[0], lat->V / 2);
}
/* operator/tm_operators.c:312-322 */
void tmo_Mtm_plus_sym_dagg_psi(tmo_lattice *lat, tmo_spinor *l, tmo_spinor *k) {
  const int N = lat->V / 2;
  tmo_gamma5(l, k, N);
  tmo_mul_one_pm_imu_inv(lat, l, -1., N);
  tmo_Hopping_Matrix(lat, EO, lat->scratch[1], l);
  tmo_mul_one_pm_imu_inv(lat, lat->scratch[1], -1., N);
  tmo_Hopping_Matrix(lat, OE, lat->scratch[0], lat->scratch[1]);
  tmo_gamma5(lat->scratch[1], lat->scratch[0], N);
  tmo_diff(l, k, lat->scratch[1], N);
}
/* operator/tm_operators.c:347-364, statement for statement -- including the second half, which applies its
 * stencils to scratch[0] but then rebuilds l from k and scratch[0] again (so the function does not return
 * Q_+ Q_- in the symmetric scheme; the oracle restates what the reference computes, not what it may have meant). */
void tmo_Qtm_pm_sym_psi(tmo_lattice *lat, tmo_spinor *l, tmo_spinor *k) {
  const int N = lat->V / 2;
  tmo_sym_core(lat, k, -1.);
  tmo_diff(l, k, lat->scratch[0], N);
  tmo_gamma5(l, l, N);
  tmo_Hopping_Matrix(lat, EO, l, lat->scratch[0]);
  tmo_mul_one_pm_imu_inv(lat, l, +1., N);
  tmo_Hopping_Matrix(lat, OE, lat->scratch[1], l);
  tmo_mul_one_pm_imu_inv(lat, lat->scratch[0], +1., N);
  tmo_diff(l, k, lat->scratch[0], N);
  tmo_gamma5(l, l, N);
}
/* operator/tm_operators.c:117-128 */
void tmo_M_full(tmo_lattice *lat, tmo_spinor *Even_new, tmo_spinor *Odd_new,
                const tmo_spinor *Even, const tmo_spinor *Odd) {
  const int N = lat->V / 2;
  tmo_Hopping_Matrix(lat, EO, lat->scratch[0], Odd);
  tmo_assign_mul_one_pm_imu(lat, Even_new, Even, 1., N);
  tmo_assign_add_mul_r(Even_new, lat->scratch[0], -1., N);
  tmo_Hopping_Matrix(lat, OE, lat->scratch[0], Even);
  tmo_assign_mul_one_pm_imu(lat, Odd_new, Odd, 1., N);
  tmo_assign_add_mul_r(Odd_new, lat->scratch[0], -1., N);
}

/* ---------------------------------------------------------------- clover twisted mass */
void tmo_set_clover(tmo_lattice *lat, const tmo_su3 *sw, const tmo_su3 *sw_inv) { lat->sw = sw; lat->sw_inv = sw_inv; }

/* operator/clovertm_operators.c:287-350: l = (1 + T_ee +- i mu g5)^-1 l on the V/2 sites the inverse was built for */
void tmo_clover_inv(tmo_lattice *lat, tmo_spinor *l, int tau3sign, double mu) {
  const int Vh = lat->V / 2;
  const int ioff = (tau3sign < 0 && fabs(mu) > 0) ? Vh : 0;
#pragma omp parallel for
  for (int icx = 0; icx < Vh; icx++) {
    const tmo_su3 *w = lat->sw_inv + 8 * (size_t)(ioff + icx);   /* [4][2]: w[2a + b] = sw_inv[icy][a][b] */
    tmo_spinor *rn = l + icx;
    tmo_su3_vector psi, chi, phi1 = rn->s0, phi3 = rn->s2;
    psi = su3_mul(&w[0], phi1); chi = su3_mul(&w[2], rn->s1); rn->s0 = v_add(psi, chi);
    psi = su3_mul(&w[6], phi1); chi = su3_mul(&w[4], rn->s1); rn->s1 = v_add(psi, chi);
    psi = su3_mul(&w[1], phi3); chi = su3_mul(&w[3], rn->s3); rn->s2 = v_add(psi, chi);
    psi = su3_mul(&w[7], phi3); chi = su3_mul(&w[5], rn->s3); rn->s3 = v_add(psi, chi);
  }
}

/* su3.h _vector_add_i_mul(r, c, s): r += i c s */
#define V_ADD_I_MUL(r, c, s) do { (r).c0 += I * (c) * (s).c0; (r).c1 += I * (c) * (s).c1; (r).c2 += I * (c) * (s).c2; } while (0)

/* operator/clovertm_operators.c:448-520 (g5 = 1) and :535-600 `clover` (g5 = 0):
 * l = [g5] ( (1 + T + i mu g5) k - j ) on the sites of parity ieo */
static void tmo_clover_generic(tmo_lattice *lat, int ieo, tmo_spinor *l, const tmo_spinor *k, const tmo_spinor *j,
                               double mu, int g5) {
  const int ioff = ieo == 0 ? 0 : lat->VPR / 2, Vh = lat->V / 2;
#pragma omp parallel for
  for (int icx = ioff; icx < Vh + ioff; icx++) {
    const int ix = lat->eo2lexic[icx];
    const tmo_su3 *w = lat->sw + 6 * (size_t)ix;                 /* [3][2]: w[2a + b] = sw[ix][a][b] */
    const tmo_spinor *s = k + (icx - ioff), *t = j + (icx - ioff);
    tmo_spinor o;
    tmo_su3_vector chi, psi1, psi2;
    psi1 = su3_mul(&w[0], s->s0); chi = su3_mul(&w[2], s->s1); V_ADD_ASSIGN(psi1, chi);
    psi2 = su3_inv_mul(&w[2], s->s0); chi = su3_mul(&w[4], s->s1); V_ADD_ASSIGN(psi2, chi);
    V_ADD_I_MUL(psi1, mu, s->s0); V_ADD_I_MUL(psi2, mu, s->s1);
    o.s0 = v_sub(psi1, t->s0); o.s1 = v_sub(psi2, t->s1);
    psi1 = su3_mul(&w[1], s->s2); chi = su3_mul(&w[3], s->s3); V_ADD_ASSIGN(psi1, chi);
    psi2 = su3_inv_mul(&w[3], s->s2); chi = su3_mul(&w[5], s->s3); V_ADD_ASSIGN(psi2, chi);
    V_ADD_I_MUL(psi1, -mu, s->s2); V_ADD_I_MUL(psi2, -mu, s->s3);
    if (g5) { o.s2 = v_sub(t->s2, psi1); o.s3 = v_sub(t->s3, psi2); }
    else    { o.s2 = v_sub(psi1, t->s2); o.s3 = v_sub(psi2, t->s3); }
    l[icx - ioff] = o;
  }
}
void tmo_clover_gamma5(tmo_lattice *lat, int ieo, tmo_spinor *l, const tmo_spinor *k, const tmo_spinor *j, double mu) {
  tmo_clover_generic(lat, ieo, l, k, j, mu, 1);
}
void tmo_clover(tmo_lattice *lat, int ieo, tmo_spinor *l, const tmo_spinor *k, const tmo_spinor *j, double mu) {
  tmo_clover_generic(lat, ieo, l, k, j, mu, 0);
}

/* operator/clovertm_operators.c:233-245 */
void tmo_Qsw_pm_psi(tmo_lattice *lat, tmo_spinor *l, tmo_spinor *k) {
  tmo_Hopping_Matrix(lat, EO, lat->scratch[1], k);
  tmo_clover_inv(lat, lat->scratch[1], -1, lat->mu);
  tmo_Hopping_Matrix(lat, OE, lat->scratch[0], lat->scratch[1]);
  tmo_clover_gamma5(lat, OE, lat->scratch[0], k, lat->scratch[0], -(lat->mu + lat->mu3));
  tmo_Hopping_Matrix(lat, EO, l, lat->scratch[0]);
  tmo_clover_inv(lat, l, +1, lat->mu);
  tmo_Hopping_Matrix(lat, OE, lat->scratch[1], l);
  tmo_clover_gamma5(lat, OE, l, lat->scratch[0], lat->scratch[1], +(lat->mu + lat->mu3));
}
/* the rest of the e/o clover family, operator/clovertm_operators.c:201-268: which = 0 Qsw_psi (mu = 0 in the
 * diagonal term), +1 Qsw_plus_psi / Msw_plus_psi, -1 Qsw_minus_psi / Msw_minus_psi; g5 selects clover_gamma5 vs clover */
static void tmo_sw_hat(tmo_lattice *lat, tmo_spinor *l, tmo_spinor *k, int which, int g5) {
  tmo_Hopping_Matrix(lat, EO, lat->scratch[1], k);
  tmo_clover_inv(lat, lat->scratch[1], which < 0 ? -1 : +1, lat->mu);
  tmo_Hopping_Matrix(lat, OE, lat->scratch[0], lat->scratch[1]);
  tmo_clover_generic(lat, OE, l, k, lat->scratch[0], which * (lat->mu + lat->mu3), g5);   /* +-(g_mu + g_mu3), :208,216,258,265 */
}
void tmo_Qsw_psi(tmo_lattice *lat, tmo_spinor *l, tmo_spinor *k) { tmo_sw_hat(lat, l, k, 0, 1); }          /* :201-206 */
void tmo_Qsw_minus_psi(tmo_lattice *lat, tmo_spinor *l, tmo_spinor *k) { tmo_sw_hat(lat, l, k, -1, 1); }   /* :209-214 */
void tmo_Qsw_plus_psi(tmo_lattice *lat, tmo_spinor *l, tmo_spinor *k) { tmo_sw_hat(lat, l, k, +1, 1); }    /* :217-222 */
void tmo_Msw_psi(tmo_lattice *lat, tmo_spinor *l, tmo_spinor *k) { tmo_sw_hat(lat, l, k, 0, 0); }          /* :247-252 */
void tmo_Msw_minus_psi(tmo_lattice *lat, tmo_spinor *l, tmo_spinor *k) { tmo_sw_hat(lat, l, k, -1, 0); }   /* :261-266 */
/* :225-237 */
void tmo_Qsw_sq_psi(tmo_lattice *lat, tmo_spinor *l, tmo_spinor *k) {
  tmo_Hopping_Matrix(lat, EO, lat->scratch[1], k);
  tmo_clover_inv(lat, lat->scratch[1], +1, lat->mu);
  tmo_Hopping_Matrix(lat, OE, lat->scratch[0], lat->scratch[1]);
  tmo_clover_gamma5(lat, OE, lat->scratch[0], k, lat->scratch[0], 0.);
  tmo_Hopping_Matrix(lat, EO, l, lat->scratch[0]);
  tmo_clover_inv(lat, l, +1, lat->mu);
  tmo_Hopping_Matrix(lat, OE, lat->scratch[1], l);
  tmo_clover_gamma5(lat, OE, l, lat->scratch[0], lat->scratch[1], 0.);
}
/* operator/assign_mul_one_sw_pm_imu_inv_block_body.c:1-72: k = (1 + T + i mu g5) l on parity ieo */
void tmo_assign_mul_one_sw_pm_imu(tmo_lattice *lat, int ieo, tmo_spinor *k, const tmo_spinor *l, double mu) {
  const int ioff = ieo == 0 ? 0 : lat->VPR / 2, Vh = lat->V / 2;
#pragma omp parallel for
  for (int icx = ioff; icx < Vh + ioff; icx++) {
    const tmo_su3 *w = lat->sw + 6 * (size_t)lat->eo2lexic[icx];
    const tmo_spinor *s = l + (icx - ioff);
    tmo_spinor o;
    tmo_su3_vector chi, psi1, psi2;
    psi1 = su3_mul(&w[0], s->s0); chi = su3_mul(&w[2], s->s1); V_ADD_ASSIGN(psi1, chi);
    psi2 = su3_inv_mul(&w[2], s->s0); chi = su3_mul(&w[4], s->s1); V_ADD_ASSIGN(psi2, chi);
    V_ADD_I_MUL(psi1, mu, s->s0); V_ADD_I_MUL(psi2, mu, s->s1);
    o.s0 = psi1; o.s1 = psi2;
    psi1 = su3_mul(&w[1], s->s2); chi = su3_mul(&w[3], s->s3); V_ADD_ASSIGN(psi1, chi);
    psi2 = su3_inv_mul(&w[3], s->s2); chi = su3_mul(&w[5], s->s3); V_ADD_ASSIGN(psi2, chi);
    V_ADD_I_MUL(psi1, -mu, s->s2); V_ADD_I_MUL(psi2, -mu, s->s3);
    o.s2 = psi1; o.s3 = psi2;
    k[icx - ioff] = o;
  }
}
/* operator/assign_mul_one_sw_pm_imu_inv_block_body.c:143-196: k = sw_inv(+mu set) l; like the reference, ieo and mu are not
 * looked at (the array holds the inverse for the parity / mu sw_invert was last called with) */
void tmo_assign_mul_one_sw_pm_imu_inv(tmo_lattice *lat, int ieo, tmo_spinor *k, const tmo_spinor *l, double mu) {
  (void)ieo; (void)mu;
  const int Vh = lat->V / 2;
#pragma omp parallel for
  for (int icx = 0; icx < Vh; icx++) {
    const tmo_su3 *w = lat->sw_inv + 8 * (size_t)icx;
    const tmo_spinor *rn = l + icx;
    tmo_spinor o;
    tmo_su3_vector psi, chi, phi1 = rn->s0, phi3 = rn->s2;
    psi = su3_mul(&w[0], phi1); chi = su3_mul(&w[2], rn->s1); o.s0 = v_add(psi, chi);
    psi = su3_mul(&w[6], phi1); chi = su3_mul(&w[4], rn->s1); o.s1 = v_add(psi, chi);
    psi = su3_mul(&w[1], phi3); chi = su3_mul(&w[3], rn->s3); o.s2 = v_add(psi, chi);
    psi = su3_mul(&w[7], phi3); chi = su3_mul(&w[5], rn->s3); o.s3 = v_add(psi, chi);
    k[icx] = o;
  }
}
/* operator/clovertm_operators.c:96-110 */
void tmo_Msw_full(tmo_lattice *lat, tmo_spinor *Even_new, tmo_spinor *Odd_new, const tmo_spinor *Even, const tmo_spinor *Odd) {
  const int N = lat->V / 2;
  tmo_Hopping_Matrix(lat, EO, lat->scratch[0], Odd);
  tmo_assign_mul_one_sw_pm_imu(lat, EO, Even_new, Even, +lat->mu);
  tmo_assign_add_mul_r(Even_new, lat->scratch[0], -1., N);
  tmo_Hopping_Matrix(lat, OE, lat->scratch[0], Even);
  tmo_assign_mul_one_sw_pm_imu(lat, OE, Odd_new, Odd, +lat->mu);
  tmo_assign_add_mul_r(Odd_new, lat->scratch[0], -1., N);
}
/* operator/clovertm_operators.c:256-261 */
void tmo_Msw_plus_psi(tmo_lattice *lat, tmo_spinor *l, tmo_spinor *k) {
  tmo_Hopping_Matrix(lat, EO, lat->scratch[1], k);
  tmo_clover_inv(lat, lat->scratch[1], +1, lat->mu);
  tmo_Hopping_Matrix(lat, OE, lat->scratch[0], lat->scratch[1]);
  tmo_clover(lat, OE, l, k, lat->scratch[0], +(lat->mu + lat->mu3));
}

/* ---------------------------------------------------------------- clover term and its inverse (host-side inputs in
 * the reference: operator/clover_term.c:88-200, operator/clover_invert.c:88-257).  3x3 blocks are handled as
 * m[row][col] views of tmo_su3 (su3.h:40-43 is row-major c00..c22).  */
typedef double _Complex c33[3][3];
#define M33(u) (*(c33 *)(u))
/* u (+)= op(v) op(w), op = identity or dagger; sums run left to right like su3.h:583-640 */
static inline void m33_mul(tmo_su3 *u, const tmo_su3 *v, int vdag, const tmo_su3 *w, int wdag, int acc) {
  tmo_su3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double _Complex a0 = vdag ? conj(M33(v)[0][i]) : M33(v)[i][0], a1 = vdag ? conj(M33(v)[1][i]) : M33(v)[i][1],
                      a2 = vdag ? conj(M33(v)[2][i]) : M33(v)[i][2];
      double _Complex b0 = wdag ? conj(M33(w)[j][0]) : M33(w)[0][j], b1 = wdag ? conj(M33(w)[j][1]) : M33(w)[1][j],
                      b2 = wdag ? conj(M33(w)[j][2]) : M33(w)[2][j];
      M33(&r)[i][j] = a0 * b0 + a1 * b1 + a2 * b2;
    }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      if (acc) M33(u)[i][j] += M33(&r)[i][j]; else M33(u)[i][j] = M33(&r)[i][j];
}
static inline const tmo_su3 *glink(const tmo_lattice *lat, int ix, int mu) { return lat->gauge + (size_t)4 * ix + mu; }

/* operator/clover_term.c:88-200.  sw is [V][3][2] (clover_term.c:60-87). */
void tmo_sw_term(tmo_lattice *lat, tmo_su3 *sw, double kappa, double c_sw) {
  const double ka_csw_8 = kappa * c_sw / 8.;
  const int *iup = lat->iup, *idn = lat->idn;
#pragma omp parallel for
  for (int x = 0; x < lat->V; x++) {
    tmo_su3 fkl[4][4], v1, v2, plaq, electric[4], magnetic[4];
    for (int k = 0; k < 4; k++)
      for (int l = k + 1; l < 4; l++) {
        const int xpk = iup[4 * x + k], xpl = iup[4 * x + l], xmk = idn[4 * x + k], xml = idn[4 * x + l];
        const int xpkml = idn[4 * xpk + l], xplmk = idn[4 * xpl + k], xmkml = idn[4 * xml + k];
        /* four leaves of the clover in the (k,l) plane, clover_term.c:120-151 */
        m33_mul(&v1, glink(lat, x, k), 0, glink(lat, xpk, l), 0, 0);
        m33_mul(&v2, glink(lat, x, l), 0, glink(lat, xpl, k), 0, 0);
        m33_mul(&plaq, &v1, 0, &v2, 1, 0);
        m33_mul(&v1, glink(lat, x, l), 0, glink(lat, xplmk, k), 1, 0);
        m33_mul(&v2, glink(lat, xmk, l), 1, glink(lat, xmk, k), 0, 0);
        m33_mul(&plaq, &v1, 0, &v2, 0, 1);
        m33_mul(&v1, glink(lat, xmkml, l), 0, glink(lat, xmk, k), 0, 0);
        m33_mul(&v2, glink(lat, xmkml, k), 0, glink(lat, xml, l), 0, 0);
        m33_mul(&plaq, &v1, 1, &v2, 0, 1);
        m33_mul(&v1, glink(lat, xml, l), 1, glink(lat, xml, k), 0, 0);
        m33_mul(&v2, glink(lat, xpkml, l), 0, glink(lat, x, k), 1, 0);
        m33_mul(&plaq, &v1, 0, &v2, 0, 1);
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++) M33(&fkl[k][l])[i][j] = M33(&plaq)[i][j] - conj(M33(&plaq)[j][i]);   /* :152-153 */
      }
    for (int k = 1; k < 4; k++) electric[k] = fkl[0][k];
    magnetic[1] = fkl[2][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M33(&magnetic[2])[i][j] = -M33(&fkl[1][3])[i][j];
    magnetic[3] = fkl[1][2];
    tmo_su3 *o = sw + (size_t)6 * x;    /* o[2*a+b] = sw[x][a][b] */
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        const double one = i == j ? 1. : 0.;
        const double _Complex e1 = M33(&electric[1])[i][j], e2 = M33(&electric[2])[i][j], e3 = M33(&electric[3])[i][j];
        const double _Complex m1 = M33(&magnetic[1])[i][j], m2 = M33(&magnetic[2])[i][j], m3 = M33(&magnetic[3])[i][j];
        double _Complex t;
        /* upper left 6x6, clover_term.c:174-183 */
        t = I * (e3 - m3);                     M33(&o[0])[i][j] = one; M33(&o[0])[i][j] += ka_csw_8 * t;
        t = I * (e1 - m1); t += (e2 - m2);     M33(&o[2])[i][j] = ka_csw_8 * t;
        t = I * (m3 - e3);                     M33(&o[4])[i][j] = one; M33(&o[4])[i][j] += ka_csw_8 * t;
        /* lower right 6x6, clover_term.c:187-196 */
        t = I * (e3 + m3);                     M33(&o[1])[i][j] = one; M33(&o[1])[i][j] += (-ka_csw_8) * t;
        t = I * (e1 + m1); t += (e2 + m2);     M33(&o[3])[i][j] = (-ka_csw_8) * t;
        t = I * (m3 + e3);                     M33(&o[5])[i][j] = one; M33(&o[5])[i][j] += ka_csw_8 * t;
      }
  }
}

/* operator/clover_invert.c:88-160: in-place inverse of a 6x6 complex matrix by Householder triangularisation
 * (no pivoting), back-substitution of the triangle, and the reflections applied from the right in reverse. */
static int tmo_six_invert(double _Complex a[6][6]) {
  const double tiny = 1.0e-20;            /* tiny_t, operator/clover_leaf.c */
  double _Complex d[6], u[6], sigma, z;
  double p[6], s, q;
  int fail = 0;
  for (int k = 0; k < 5; k++) {
    s = 0.0;
    for (int j = k + 1; j < 6; j++) s += conj(a[j][k]) * a[j][k];
    s = sqrt(1. + s / (conj(a[k][k]) * a[k][k]));
    sigma = s * a[k][k];
    a[k][k] += sigma;
    p[k] = conj(sigma) * a[k][k];
    q = conj(sigma) * sigma;
    if (q < tiny) fail++;
    d[k] = -conj(sigma) / q;
    for (int j = k + 1; j < 6; j++) {
      z = 0.0;
      for (int i = k; i < 6; i++) z += conj(a[i][k]) * a[i][j];
      z /= p[k];
      for (int i = k; i < 6; i++) a[i][j] -= z * a[i][k];
    }
  }
  sigma = a[5][5];
  q = conj(sigma) * sigma;
  if (q < tiny) fail++;
  d[5] = conj(sigma) / q;
  for (int k = 5; k >= 0; k--)
    for (int i = k - 1; i >= 0; i--) {
      z = 0.0;
      for (int j = i + 1; j < k; j++) z += a[i][j] * a[j][k];
      z += a[i][k] * d[k];
      a[i][k] = -z * d[i];
    }
  a[5][5] = d[5];
  for (int k = 4; k >= 0; k--) {
    for (int j = k; j < 6; j++) u[j] = a[j][k];
    a[k][k] = d[k];
    for (int j = k + 1; j < 6; j++) a[j][k] = 0.0;
    for (int i = 0; i < 6; i++) {
      z = 0.0;
      for (int j = k; j < 6; j++) z += a[i][j] * u[j];
      z /= p[k];
      for (int j = k; j < 6; j++) a[i][j] -= conj(u[j]) * z;
    }
  }
  return fail;
}

/* operator/clover_invert.c:170-257: sw_inv[icy] (+mu) and sw_inv[icy + V/2] (-mu, only when mu != 0) for the sites
 * of parity ieo; sw_inv is [V][4][2] with blocks 0: upper-left, 1: upper-right, 2: lower-right, 3: lower-left. */
int tmo_sw_invert(tmo_lattice *lat, tmo_su3 *sw_inv, const tmo_su3 *sw, int ieo, double mu) {
  const int Vh = lat->V / 2, ioff = ieo == 0 ? 0 : (lat->V + lat->RAND) / 2;
  int fails = 0;
#pragma omp parallel for reduction(+ : fails)
  for (int icy = 0; icy < Vh; icy++) {
    const int x = lat->eo2lexic[icy + ioff];
    const tmo_su3 *w = sw + (size_t)6 * x;
    for (int set = 0; set < (fabs(mu) > 0. ? 2 : 1); set++)
      for (int b = 0; b < 2; b++) {
        double _Complex a[6][6];
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++) {
            a[i][j] = M33(&w[0 + b])[i][j];
            a[i][j + 3] = M33(&w[2 + b])[i][j];
            a[i + 3][j] = conj(M33(&w[2 + b])[j][i]);
            a[i + 3][j + 3] = M33(&w[4 + b])[i][j];
          }
        const double m = (set == 0 ? 1. : -1.) * (b == 0 ? mu : -mu);
        for (int i = 0; i < 6; i++) a[i][i] += I * m;
        fails += tmo_six_invert(a);
        tmo_su3 *o = sw_inv + (size_t)8 * (icy + set * Vh);   /* o[2*a+b] = sw_inv[.][a][b] */
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++) {
            M33(&o[0 + b])[i][j] = a[i][j];
            M33(&o[2 + b])[i][j] = a[i][j + 3];
            M33(&o[4 + b])[i][j] = a[i + 3][j + 3];
            M33(&o[6 + b])[i][j] = a[i + 3][j];
          }
      }
  }
  return fails;
}

/* ---------------------------------------------------------------- fermion force, hopping part
 * deriv_Sb.c:401-700 (generic branch, _GAUGE_COPY): for every site x of parity ieo, left vector g5 l(x), right field k:
 *   df[x][mu]      += 2 factor trlambda( ka_mu U_mu(x)      [ (P+ g5 l(x)) (x) (P+ k(x+mu))^dagger ]^dagger )
 *   df[x-mu][mu]   += 2 factor trlambda( ka_mu U_mu(x-mu)   [ (P- k(x-mu)) (x) (P- g5 l(x))^dagger ]^dagger )
 * with the two-component projections of hopping.h and trlambda of su3adj.h:164-172.  Every link receives exactly one
 * contribution per call, so the site loop is race-free.  df is su3adj [VPR][4] = 8 doubles per link. */
static inline void tmo_project(const tmo_spinor *s, int mu, int plus, tmo_su3_vector *a, tmo_su3_vector *b) {
  switch (2 * mu + (plus ? 0 : 1)) {
    case 0: *a = v_add(s->s0, s->s2);   *b = v_add(s->s1, s->s3);   break;   /* deriv_Sb.c:470-474 */
    case 1: *a = v_sub(s->s0, s->s2);   *b = v_sub(s->s1, s->s3);   break;   /* :492-496 */
    case 2: *a = v_i_add(s->s0, s->s3); *b = v_i_add(s->s1, s->s2); break;   /* :515-519 */
    case 3: *a = v_i_sub(s->s0, s->s3); *b = v_i_sub(s->s1, s->s2); break;   /* :537-541 */
    case 4: *a = v_add(s->s0, s->s3);   *b = v_sub(s->s1, s->s2);   break;   /* :559-563 */
    case 5: *a = v_sub(s->s0, s->s3);   *b = v_add(s->s1, s->s2);   break;   /* :581-585 */
    case 6: *a = v_i_add(s->s0, s->s2); *b = v_i_sub(s->s1, s->s3); break;   /* :603-607 */
    default: *a = v_i_sub(s->s0, s->s2); *b = v_i_add(s->s1, s->s3); break;  /* :625-629 */
  }
}
/* t = u (x) v^dagger + w (x) z^dagger (su3.h:706-715); v2 = U t^dagger (su3.h:605-614); v1 = c v2; df += fac trlambda(v1) */
static inline void tmo_force_link(double *d, const tmo_su3 *U, double _Complex c, double fac, const tmo_su3_vector *u,
                                  const tmo_su3_vector *v, const tmo_su3_vector *w, const tmo_su3_vector *z) {
  const double _Complex *uu = &u->c0, *vv = &v->c0, *ww = &w->c0, *zz = &z->c0;
  tmo_su3 t, v2, a;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) M33(&t)[i][j] = uu[i] * conj(vv[j]) + ww[i] * conj(zz[j]);
  m33_mul(&v2, U, 0, &t, 1, 0);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) M33(&a)[i][j] = c * M33(&v2)[i][j];
  d[0] += fac * (-cimag(a.c10) - cimag(a.c01));
  d[1] += fac * (+creal(a.c10) - creal(a.c01));
  d[2] += fac * (-cimag(a.c00) + cimag(a.c11));
  d[3] += fac * (-cimag(a.c20) - cimag(a.c02));
  d[4] += fac * (+creal(a.c20) - creal(a.c02));
  d[5] += fac * (-cimag(a.c21) - cimag(a.c12));
  d[6] += fac * (+creal(a.c21) - creal(a.c12));
  d[7] += fac * ((-cimag(a.c00) - cimag(a.c11) + 2.0 * cimag(a.c22)) * 0.577350269189625);
}
void tmo_deriv_Sb(tmo_lattice *lat, int ieo, const tmo_spinor *l, const tmo_spinor *k, double *df, double factor) {
  if (lat->gauge_dirty) tmo_update_backward_gauge(lat);     /* deriv_Sb.c:416-420 */
  const int ioff = ieo == 0 ? 0 : lat->VPR / 2, Vh = lat->V / 2;
#pragma omp parallel for
  for (int icx = ioff; icx < Vh + ioff; icx++) {
    const int ix = lat->eo2lexic[icx];
    const tmo_su3 *u = lat->gauge_copy + 8 * (size_t)icx;   /* [2mu] = U_mu(x), [2mu+1] = U_mu(x-mu) */
    tmo_spinor rr = l[icx - ioff];
    rr.s2.c0 = -rr.s2.c0; rr.s2.c1 = -rr.s2.c1; rr.s2.c2 = -rr.s2.c2;   /* gamma5, deriv_Sb.c:455-456 */
    rr.s3.c0 = -rr.s3.c0; rr.s3.c1 = -rr.s3.c1; rr.s3.c2 = -rr.s3.c2;
    for (int mu = 0; mu < 4; mu++) {
      tmo_su3_vector psia, psib, phia, phib;
      const int iyp = lat->iup[4 * ix + mu], iym = lat->idn[4 * ix + mu];
      tmo_project(k + lat->lexic2eosub[iyp], mu, 1, &psia, &psib);
      tmo_project(&rr, mu, 1, &phia, &phib);
      tmo_force_link(df + ((size_t)4 * ix + mu) * 8, &u[2 * mu], lat->ka[mu], 2. * factor, &phia, &psia, &phib, &psib);
      tmo_project(k + lat->lexic2eosub[iym], mu, 0, &psia, &psib);
      tmo_project(&rr, mu, 0, &phia, &phib);
      tmo_force_link(df + ((size_t)4 * iym + mu) * 8, &u[2 * mu + 1], lat->ka[mu], 2. * factor, &psia, &phia, &psib, &phib);
    }
  }
}

/* ---------------------------------------------------------------- clover part of the fermion force
 * (monomial/cloverdet_monomial.c:110-147 calls, after the solves: sw_spinor_eo x2, sw_deriv, sw_all).
 * swm / swp: su3 [V][4] each, lexicographic site (operator/clover_leaf.c:141-172). */
/* operator/clover_deriv.c:252-318: insertion matrices from the spinor outer products on the sites of parity ieo */
void tmo_sw_spinor_eo(tmo_lattice *lat, int ieo, tmo_su3 *swm, tmo_su3 *swp, const tmo_spinor *kk, const tmo_spinor *ll, double fac) {
  const int ioff = ieo == 0 ? 0 : lat->VPR / 2, Vh = lat->V / 2;
#pragma omp parallel for
  for (int icx = ioff; icx < Vh + ioff; icx++) {
    const int x = lat->eo2lexic[icx];
    const tmo_spinor *r = kk + (icx - ioff), *s = ll + (icx - ioff);
    const tmo_su3_vector *ra[4] = {&r->s0, &r->s0, &r->s1, &r->s1}, *sa[4] = {&s->s0, &s->s1, &s->s1, &s->s0};   /* v0..v3 */
    const tmo_su3_vector *rb[4] = {&r->s2, &r->s2, &r->s3, &r->s3}, *sb[4] = {&s->s2, &s->s3, &s->s3, &s->s2};   /* u0..u3 */
    for (int n = 0; n < 4; n++) {
      const double _Complex *pr = &ra[n]->c0, *ps = &sa[n]->c0, *qr = &rb[n]->c0, *qs = &sb[n]->c0;
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
          const double _Complex v = pr[a] * conj(ps[b]), u = -qr[a] * conj(qs[b]);   /* su3.h:683-703 (u carries the gamma5 sign) */
          M33(&swm[4 * (size_t)x + n])[a][b] += fac * (u - v);
          M33(&swp[4 * (size_t)x + n])[a][b] += fac * (u + v);
        }
    }
  }
}
/* operator/clover_deriv.c:72-153: the tr log part, from sw_inv of the sites of parity ieo */
void tmo_sw_deriv(tmo_lattice *lat, int ieo, tmo_su3 *swm, tmo_su3 *swp, double mu) {
  const int ioff = ieo == 0 ? 0 : lat->VPR / 2, Vh = lat->V / 2;
  const double fac = fabs(mu) > 0. ? 0.5 : 1.0;
#pragma omp parallel for
  for (int icy = 0; icy < Vh; icy++) {
    const int x = lat->eo2lexic[icy + ioff];
    for (int set = 0; set < (fabs(mu) > 0. ? 2 : 1); set++) {
      const tmo_su3 *w = lat->sw_inv + 8 * (size_t)(icy + set * Vh);
      for (int n = 0; n < 4; n++)
        for (int a = 0; a < 3; a++)
          for (int b = 0; b < 3; b++) {
            const double _Complex lp = M33(&w[2 * n + 1])[a][b] + M33(&w[2 * n])[a][b], lm = M33(&w[2 * n + 1])[a][b] - M33(&w[2 * n])[a][b];
            M33(&swm[4 * (size_t)x + n])[a][b] += fac * lm;
            M33(&swp[4 * (size_t)x + n])[a][b] += fac * lp;
          }
    }
  }
}
/* su3adj.h:164-172 */
static inline void tmo_trace_lambda_add(double *d, double c, const tmo_su3 *a) {
  d[0] += c * (-cimag(a->c10) - cimag(a->c01));
  d[1] += c * (+creal(a->c10) - creal(a->c01));
  d[2] += c * (-cimag(a->c00) + cimag(a->c11));
  d[3] += c * (-cimag(a->c20) - cimag(a->c02));
  d[4] += c * (+creal(a->c20) - creal(a->c02));
  d[5] += c * (-cimag(a->c21) - cimag(a->c12));
  d[6] += c * (+creal(a->c21) - creal(a->c12));
  d[7] += c * ((-cimag(a->c00) - cimag(a->c11) + 2.0 * cimag(a->c22)) * 0.577350269189625);
}
static inline void m33_dag(tmo_su3 *u, const tmo_su3 *v) {
  tmo_su3 r;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M33(&r)[i][j] = conj(M33(v)[j][i]);
  *u = r;
}
/* operator/clover_accumulate_deriv.c:58-205: sixteen link derivatives per plane and site from the four clover leaves with the
 * insertion matrix vis[k][l] built from swm / swp.  The site loop scatters to neighbouring links, so it runs serially here
 * (the reference uses atomic updates under OpenMP); df is su3adj [VPR][4] as 8 doubles per link. */
void tmo_sw_all(tmo_lattice *lat, double *df, const tmo_su3 *swm, const tmo_su3 *swp, double kappa, double c_sw) {
  const double c = -2. * (kappa * c_sw / 8.);
  const int *iup = lat->iup, *idn = lat->idn;
#define DF(ix, mu) (df + ((size_t)4 * (ix) + (mu)) * 8)
  for (int x = 0; x < lat->V; x++) {
    tmo_su3 vis[4][4], v1, v2, vv1, vv2, plaq;
    const tmo_su3 *m = swm + 4 * (size_t)x, *p = swp + 4 * (size_t)x;
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        M33(&vis[0][1])[a][b] = -I * (M33(&m[1])[a][b] + M33(&m[3])[a][b]);
        M33(&vis[0][2])[a][b] = M33(&m[1])[a][b] - M33(&m[3])[a][b];
        M33(&vis[0][3])[a][b] = I * (M33(&m[2])[a][b] - M33(&m[0])[a][b]);
        M33(&vis[2][3])[a][b] = -I * (M33(&p[1])[a][b] + M33(&p[3])[a][b]);
        M33(&vis[1][3])[a][b] = M33(&p[3])[a][b] - M33(&p[1])[a][b];
        M33(&vis[1][2])[a][b] = I * (M33(&p[2])[a][b] - M33(&p[0])[a][b]);
      }
    for (int k = 0; k < 4; k++)
      for (int l = k + 1; l < 4; l++) {   /* anti-hermitian part, clover_accumulate_deriv.c:84-96 */
        m33_dag(&v1, &vis[k][l]);
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) M33(&vis[k][l])[a][b] -= M33(&v1)[a][b];
      }
    for (int k = 0; k < 4; k++)
      for (int l = k + 1; l < 4; l++) {
        const int xpk = iup[4 * x + k], xpl = iup[4 * x + l], xmk = idn[4 * x + k], xml = idn[4 * x + l];
        const int xpkml = idn[4 * xpk + l], xplmk = idn[4 * xpl + k], xmkml = idn[4 * xml + k];
        const tmo_su3 *V = &vis[k][l], *w1, *w2, *w3, *w4;
        /* leaf 1 */
        w1 = glink(lat, x, k); w2 = glink(lat, xpk, l); w3 = glink(lat, xpl, k); w4 = glink(lat, x, l);
        m33_mul(&v1, w1, 0, w2, 0, 0); m33_mul(&v2, w4, 0, w3, 0, 0); m33_mul(&plaq, &v1, 0, &v2, 1, 0);
        m33_mul(&vv1, &plaq, 0, V, 0, 0);                              tmo_trace_lambda_add(DF(x, k), c, &vv1);
        m33_mul(&vv2, w1, 1, &vv1, 0, 0); m33_mul(&vv1, &vv2, 0, w1, 0, 0);  tmo_trace_lambda_add(DF(xpk, l), c, &vv1);
        m33_mul(&vv2, V, 0, &plaq, 0, 0); m33_dag(&vv1, &vv2);         tmo_trace_lambda_add(DF(x, l), c, &vv1);
        m33_mul(&vv2, w4, 1, &vv1, 0, 0); m33_mul(&vv1, &vv2, 0, w4, 0, 0);  tmo_trace_lambda_add(DF(xpl, k), c, &vv1);
        /* leaf 2 */
        w1 = glink(lat, x, l); w2 = glink(lat, xplmk, k); w3 = glink(lat, xmk, l); w4 = glink(lat, xmk, k);
        m33_mul(&v1, w1, 0, w2, 1, 0); m33_mul(&v2, w3, 1, w4, 0, 0); m33_mul(&plaq, &v1, 0, &v2, 0, 0);
        m33_mul(&vv1, &plaq, 0, V, 0, 0);                              tmo_trace_lambda_add(DF(x, l), c, &vv1);
        m33_dag(&vv1, &v1); m33_mul(&vv2, &vv1, 0, V, 1, 0); m33_mul(&vv1, &vv2, 0, &v2, 1, 0);  tmo_trace_lambda_add(DF(xplmk, k), c, &vv1);
        m33_mul(&vv2, w3, 0, &vv1, 0, 0); m33_mul(&vv1, &vv2, 0, w3, 1, 0);  tmo_trace_lambda_add(DF(xmk, l), c, &vv1);
        m33_dag(&vv2, &vv1);                                           tmo_trace_lambda_add(DF(xmk, k), c, &vv2);
        /* leaf 3 */
        w1 = glink(lat, xmk, k); w2 = glink(lat, xmkml, l); w3 = glink(lat, xmkml, k); w4 = glink(lat, xml, l);
        m33_mul(&v1, w2, 0, w1, 0, 0); m33_mul(&v2, w3, 0, w4, 0, 0);
        m33_mul(&vv1, w1, 0, V, 1, 0); m33_mul(&vv2, &vv1, 0, &v2, 1, 0); m33_mul(&vv1, &vv2, 0, w2, 0, 0);  tmo_trace_lambda_add(DF(xmk, k), c, &vv1);
        m33_mul(&vv2, w2, 0, &vv1, 0, 0); m33_mul(&vv1, &vv2, 0, w2, 1, 0);  tmo_trace_lambda_add(DF(xmkml, l), c, &vv1);
        m33_dag(&vv2, &vv1);                                           tmo_trace_lambda_add(DF(xmkml, k), c, &vv2);
        m33_mul(&vv1, w3, 1, &vv2, 0, 0); m33_mul(&vv2, &vv1, 0, w3, 0, 0);  tmo_trace_lambda_add(DF(xml, l), c, &vv2);
        /* leaf 4 */
        w1 = glink(lat, xml, l); w2 = glink(lat, xml, k); w3 = glink(lat, xpkml, l); w4 = glink(lat, x, k);
        m33_mul(&v1, w1, 1, w2, 0, 0); m33_mul(&v2, w3, 0, w4, 1, 0);
        m33_mul(&vv1, w1, 0, V, 1, 0); m33_mul(&vv2, &vv1, 0, &v2, 1, 0); m33_mul(&vv1, &vv2, 0, w2, 1, 0);  tmo_trace_lambda_add(DF(xml, l), c, &vv1);
        m33_dag(&vv2, &vv1);                                           tmo_trace_lambda_add(DF(xml, k), c, &vv2);
        m33_mul(&vv1, w2, 1, &vv2, 0, 0); m33_mul(&vv2, &vv1, 0, w2, 0, 0);  tmo_trace_lambda_add(DF(xpkml, l), c, &vv2);
        m33_dag(&vv2, &v2); m33_mul(&vv1, &vv2, 0, &v1, 1, 0); m33_mul(&vv2, &vv1, 0, V, 1, 0);  tmo_trace_lambda_add(DF(x, k), c, &vv2);
      }
  }
#undef DF
}

/* ---------------------------------------------------------------- linalg */
/* Per-thread Kahan partials summed in thread order, as the reference does with
   g_omp_acc_re (linalg/square_norm.c:299-304). */
static double tmo_sum_partials(const double *acc, int n) {
  double res = 0.0;
  for (int i = 0; i < n; i++) res += acc[i];
  return res;
}

/* linalg/square_norm.c:253-320 */
double tmo_square_norm(const tmo_spinor *P, int N) {
  double acc[TMO_MAX_THREADS];
  int nthr = 1;
#pragma omp parallel
  {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#pragma omp single
    nthr = omp_get_num_threads();
#endif
    double ks = 0.0, kc = 0.0, ds, tr, ts, tt;
#pragma omp for
    for (int ix = 0; ix < N; ix++) {
      const tmo_spinor *s = P + ix;
      ds = conj(s->s0.c0) * s->s0.c0 + conj(s->s0.c1) * s->s0.c1 + conj(s->s0.c2) * s->s0.c2 +
           conj(s->s1.c0) * s->s1.c0 + conj(s->s1.c1) * s->s1.c1 + conj(s->s1.c2) * s->s1.c2 +
           conj(s->s2.c0) * s->s2.c0 + conj(s->s2.c1) * s->s2.c1 + conj(s->s2.c2) * s->s2.c2 +
           conj(s->s3.c0) * s->s3.c0 + conj(s->s3.c1) * s->s3.c1 + conj(s->s3.c2) * s->s3.c2;
      tr = ds + kc; ts = tr + ks; tt = ts - ks; ks = ts; kc = tr - tt;
    }
    kc = ks + kc;
    acc[tid] = kc;
  }
  return tmo_sum_partials(acc, nthr);
}

/* linalg/scalar_prod_r.c:135-197 */
double tmo_scalar_prod_r(const tmo_spinor *S, const tmo_spinor *R, int N) {
  double acc[TMO_MAX_THREADS];
  int nthr = 1;
#pragma omp parallel
  {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#pragma omp single
    nthr = omp_get_num_threads();
#endif
    double ks = 0.0, kc = 0.0, ds, tr, ts, tt;
#pragma omp for
    for (int ix = 0; ix < N; ix++) {
      const tmo_spinor *s = S + ix, *r = R + ix;
      ds = creal(r->s0.c0 * conj(s->s0.c0)) + creal(r->s0.c1 * conj(s->s0.c1)) + creal(r->s0.c2 * conj(s->s0.c2)) +
           creal(r->s1.c0 * conj(s->s1.c0)) + creal(r->s1.c1 * conj(s->s1.c1)) + creal(r->s1.c2 * conj(s->s1.c2)) +
           creal(r->s2.c0 * conj(s->s2.c0)) + creal(r->s2.c1 * conj(s->s2.c1)) + creal(r->s2.c2 * conj(s->s2.c2)) +
           creal(r->s3.c0 * conj(s->s3.c0)) + creal(r->s3.c1 * conj(s->s3.c1)) + creal(r->s3.c2 * conj(s->s3.c2));
      tr = ds + kc; ts = tr + ks; tt = ts - ks; ks = ts; kc = tr - tt;
    }
    kc = ks + kc;
    acc[tid] = kc;
  }
  return tmo_sum_partials(acc, nthr);
}

/* linalg/assign_add_mul_r.c:346-381  P += c Q */
void tmo_assign_add_mul_r(tmo_spinor *P, const tmo_spinor *Q, double c, int N) {
  double *p = (double *)P; const double *q = (const double *)Q;
#pragma omp parallel for
  for (long i = 0; i < 24L * N; i++) p[i] += c * q[i];
}
/* linalg/assign_mul_add_r.c:340-377  R = c R + S */
void tmo_assign_mul_add_r(tmo_spinor *R, double c, const tmo_spinor *S, int N) {
  double *r = (double *)R; const double *s = (const double *)S;
#pragma omp parallel for
  for (long i = 0; i < 24L * N; i++) r[i] = c * r[i] + s[i];
}
/* linalg/assign_mul_add_r_and_square.c:145-213 -- plain (non-Kahan) per-thread sum */
double tmo_assign_mul_add_r_and_square(tmo_spinor *R, double c, const tmo_spinor *S, int N) {
  double acc[TMO_MAX_THREADS];
  int nthr = 1;
#pragma omp parallel
  {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#pragma omp single
    nthr = omp_get_num_threads();
#endif
    double ds = 0.0;
#pragma omp for
    for (int ix = 0; ix < N; ix++) {
      double *r = (double *)(R + ix); const double *s = (const double *)(S + ix);
      for (int j = 0; j < 12; j++) {
        r[2 * j] = c * r[2 * j] + s[2 * j];
        r[2 * j + 1] = c * r[2 * j + 1] + s[2 * j + 1];
        ds += r[2 * j] * r[2 * j] + r[2 * j + 1] * r[2 * j + 1];
      }
    }
    acc[tid] = ds;
  }
  return tmo_sum_partials(acc, nthr);
}
/* linalg/diff.c:270-309  Q = R - S */
void tmo_diff(tmo_spinor *Q, const tmo_spinor *R, const tmo_spinor *S, int N) {
  double *q = (double *)Q; const double *r = (const double *)R, *s = (const double *)S;
#pragma omp parallel for
  for (long i = 0; i < 24L * N; i++) q[i] = r[i] - s[i];
}
/* linalg/add.c:45-80  Q = R + S */
void tmo_add(tmo_spinor *Q, const tmo_spinor *R, const tmo_spinor *S, int N) {
  double *q = (double *)Q; const double *r = (const double *)R, *s = (const double *)S;
#pragma omp parallel for
  for (long i = 0; i < 24L * N; i++) q[i] = r[i] + s[i];
}
/* linalg/mul_r.c:40-75  R = c S */
void tmo_mul_r(tmo_spinor *R, double c, const tmo_spinor *S, int N) {
  double *r = (double *)R; const double *s = (const double *)S;
#pragma omp parallel for
  for (long i = 0; i < 24L * N; i++) r[i] = c * s[i];
}
/* linalg/assign.c:42-46 */
void tmo_assign(tmo_spinor *R, const tmo_spinor *S, int N) { memcpy(R, S, (size_t)N * sizeof(tmo_spinor)); }

/* ---------------------------------------------------------------- solver */
/* solver/cg_her.c:62-141.  res_hist[i] (i < hist_len) receives err after iteration i+1. */
int tmo_cg_her(tmo_lattice *lat, tmo_spinor *P, tmo_spinor *Q, int max_iter, double eps_sq,
               int rel_prec, int N, tmo_matrix_mult f, double *res_hist, int hist_len) {
  const size_t Vf = (size_t)lat->VPR / 2 + 1;
  tmo_spinor *blk = (tmo_spinor *)calloc(3 * Vf, sizeof(tmo_spinor)); /* solver_field.c:31-71 */
  tmo_spinor *sf[3] = {blk, blk + Vf, blk + 2 * Vf}, *stmp;
  double normsq, pro, err, alpha_cg, beta_cg, squarenorm;
  int iteration;
  squarenorm = tmo_square_norm(Q, N);
  f(lat, sf[0], P);
  tmo_diff(sf[1], Q, sf[0], N);
  tmo_assign(sf[2], sf[1], N);
  normsq = tmo_square_norm(sf[1], N);
  for (iteration = 1; iteration <= max_iter; iteration++) {
    f(lat, sf[0], sf[2]);
    pro = tmo_scalar_prod_r(sf[2], sf[0], N);
    alpha_cg = normsq / pro;
    tmo_assign_add_mul_r(P, sf[2], alpha_cg, N);
    err = tmo_assign_mul_add_r_and_square(sf[0], -alpha_cg, sf[1], N);
    if (res_hist && iteration - 1 < hist_len) res_hist[iteration - 1] = err;
    if (((err <= eps_sq) && (rel_prec == 0)) || ((err <= eps_sq * squarenorm) && (rel_prec == 1))) break;
    beta_cg = err / normsq;
    tmo_assign_mul_add_r(sf[2], beta_cg, sf[0], N);
    stmp = sf[0]; sf[0] = sf[1]; sf[1] = stmp;
    normsq = err;
  }
  free(blk);
  if (iteration > max_iter) return -1;
  return iteration;
}

/* ---------------------------------------------------------------- molecular-dynamics link update (update_gauge.c:51-110)
 * gauge: g_gauge_field as [V][4] su3 (lexicographic), mom: hf->momenta as [V][4][8] doubles (su3adj d1..d8).
 * Per link: deriv = step * momentum (su3adj.h:237-245), w = exposu3(deriv) (expo.c:56-97: Cayley-Hamilton recursion, 13
 * steps), v = restoresu3(w) (expo.c:118-137), U <- v U (su3.h:583-592).  Every expression is written in the order of
 * the reference so that the result is bit-for-bit the reference's (tests/test_oracle_vs_ref.py). */
static void tmo_exposu3(tmo_su3 *vr, const double *p /* d1..d8 */) {
  tmo_su3 v, v2;
  double fac, r, a, b;
  double _Complex a0, a1, a2, a1p;
  const double d1 = p[0], d2 = p[1], d3 = p[2], d4 = p[3], d5 = p[4], d6 = p[5], d7 = p[6], d8 = p[7];
  /* _make_su3, su3adj.h:45-54 */
  v.c00 = 0.0 + (0.5773502691896258 * d8 + d3) * I;
  v.c01 = d2 + d1 * I;
  v.c02 = d5 + d4 * I;
  v.c10 = -d2 + d1 * I;
  v.c11 = 0.0 + (0.5773502691896258 * d8 - d3) * I;
  v.c12 = d7 + d6 * I;
  v.c20 = -d5 + d4 * I;
  v.c21 = -d7 + d6 * I;
  v.c22 = 0.0 - (1.154700538379252 * d8) * I;
  m33_mul(&v2, &v, 0, &v, 0, 0);                                               /* expo.c:66 */
  a = 0.5 * (creal(v2.c00) + creal(v2.c11) + creal(v2.c22));                   /* :68 */
  b = 0.33333333333333333 * cimag(v.c00 * v2.c00 + v.c01 * v2.c10 + v.c02 * v2.c20 +
                                  v.c10 * v2.c01 + v.c11 * v2.c11 + v.c12 * v2.c21 +
                                  v.c20 * v2.c02 + v.c21 * v2.c12 + v.c22 * v2.c22);   /* :70-72 */
  a0 = 0.16059043836821615e-9;
  a1 = 0.11470745597729725e-10;
  a2 = 0.76471637318198165e-12;
  fac = 0.20876756987868099e-8;
  r = 12.0;
  for (int i = 3; i <= 15; ++i) {                                              /* :78-86 */
    a1p = a0 + a * a2;
    a0 = fac + b * I * a2;
    a2 = a1;
    a1 = a1p;
    fac *= r;
    r -= 1.0;
  }
  vr->c00 = a0 + a1 * v.c00 + a2 * v2.c00;                                     /* :88-96 */
  vr->c01 = a1 * v.c01 + a2 * v2.c01;
  vr->c02 = a1 * v.c02 + a2 * v2.c02;
  vr->c10 = a1 * v.c10 + a2 * v2.c10;
  vr->c11 = a0 + a1 * v.c11 + a2 * v2.c11;
  vr->c12 = a1 * v.c12 + a2 * v2.c12;
  vr->c20 = a1 * v.c20 + a2 * v2.c20;
  vr->c21 = a1 * v.c21 + a2 * v2.c21;
  vr->c22 = a0 + a1 * v.c22 + a2 * v2.c22;
}
static void tmo_restoresu3(tmo_su3 *vr, const tmo_su3 *u) {                    /* expo.c:118-137 */
  const double n0 = 1.0 / sqrt(conj(u->c00) * u->c00 + conj(u->c01) * u->c01 + conj(u->c02) * u->c02);
  const double n1 = 1.0 / sqrt(conj(u->c10) * u->c10 + conj(u->c11) * u->c11 + conj(u->c12) * u->c12);
  vr->c00 = n0 * u->c00; vr->c01 = n0 * u->c01; vr->c02 = n0 * u->c02;
  vr->c10 = n1 * u->c10; vr->c11 = n1 * u->c11; vr->c12 = n1 * u->c12;
  vr->c20 = conj(vr->c01 * vr->c12 - vr->c02 * vr->c11);
  vr->c21 = conj(vr->c02 * vr->c10 - vr->c00 * vr->c12);
  vr->c22 = conj(vr->c00 * vr->c11 - vr->c01 * vr->c10);
}
void tmo_update_gauge(tmo_su3 *gauge, const double *mom, int V, double step) {
#pragma omp parallel for
  for (int i = 0; i < V; i++)
    for (int mu = 0; mu < 4; mu++) {
      double deriv[8];
      tmo_su3 v, w, *z = gauge + (size_t)4 * i + mu;
      for (int k = 0; k < 8; k++) deriv[k] = step * mom[((size_t)4 * i + mu) * 8 + k];   /* update_gauge.c:85 */
      tmo_exposu3(&w, deriv);                                                          /* :86 */
      tmo_restoresu3(&v, &w);                                                          /* :87 */
      m33_mul(&w, &v, 0, z, 0, 0);                                                     /* :88 */
      *z = w;                                                                          /* :89 */
    }
}
/* update_momenta.c:67-72: momenta -= step * derivative, both su3adj [V][4][8] */
void tmo_update_momenta(double *mom, const double *deriv, int V, double step) {
#pragma omp parallel for
  for (int i = 0; i < V; i++)
    for (int k = 0; k < 32; k++) mom[(size_t)32 * i + k] -= step * deriv[(size_t)32 * i + k];
}
