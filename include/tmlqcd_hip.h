/* tmlqcd_hip.h -- C-ABI of the MI355X-native tmLQCD hot path (libtmlqcd_hip.so).
 *
 * Plain C: opaque handles, raw pointers and sizes only.  Host arrays use the
 * reference's own AoS layouts (su3.h:40-43 `su3`, su3.h:60-63 `spinor`,
 * global.h:176 `g_gauge_field[ix][mu]`, lexicographic ix of geometry_eo.c:290);
 * device-side layouts are private.
 *
 * Two layers:
 *   1. this header: explicit context + device-resident fields.  Every entry point
 *      names the reference function (file:line under /root/reference) whose
 *      semantics it reproduces.
 *   2. include/tmlqcd_dropin.h: the reference's own symbol names and signatures
 *      (Hopping_Matrix, Qtm_pm_psi, square_norm, cg_her, ...) implemented on top
 *      of layer 1 so that benchmark / invert / hmc_tm link unchanged.
 *
 * All functions returning int return 0 on success; on failure they print a
 * diagnostic to stderr and return non-zero (the drop-in layer turns that into
 * exit(), the reference's own error convention, fatal_error.c).
 */
#ifndef TMLQCD_HIP_H
#define TMLQCD_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct tmhip_ctx tmhip_ctx;
typedef struct tmhip_field tmhip_field;

/* Local lattice of this rank.  Decomposition is T-only (the reference's
 * PARALLELT, mpi_init.c:240-242,330-332): global T = nproc_t * T. */
typedef struct {
  int T, LX, LY, LZ; /* local extents; all even, T >= 2 (mpi_init.c:784-799) */
  int nproc_t;       /* ranks along T (1 = single GPU, periodic wrap is local) */
  int proc_t;        /* this rank's coordinate along T */
} tmhip_geom;

enum { TMHIP_EO = 0, TMHIP_OE = 1 };           /* global.h EO/OE */
enum { TMHIP_FIELD_EO = 0, TMHIP_FIELD_FULL = 1 }; /* V/2 sites (one parity) | V sites */

/* ---- context ------------------------------------------------------------ */
/* device: HIP device ordinal.  Replaces the index-table / gauge-copy set-up of
 * geometry() (geometry_eo.c:743) + init_gauge_field (init/init_gauge_field.c:41). */
int tmhip_create(const tmhip_geom *geom, int device, tmhip_ctx **out);
void tmhip_destroy(tmhip_ctx *ctx);
int tmhip_sync(tmhip_ctx *ctx);
const char *tmhip_version(void);
int tmhip_device_count(void);

/* boundary(kappa) with X0..X3 = theta (boundary.c:40-55) */
int tmhip_set_boundary(tmhip_ctx *ctx, double kappa, const double theta[4]);
/* Same, but takes the host's ka0..ka3 verbatim (boundary.h:25) as {re0,im0,..,re3,im3}: the
 * drop-in layer re-reads the reference's globals at every call. */
int tmhip_set_ka(tmhip_ctx *ctx, const double ka[8]);
/* g_mu (global.h:198; = 2 kappa mu); read by the twisted-mass operators at call time */
int tmhip_set_mu(tmhip_ctx *ctx, double mu);
int tmhip_set_mu3(tmhip_ctx *ctx, double mu3);   /* g_mu3 (global.h:197), default 0: Qsw_plus/minus/pm_psi and Msw_plus/minus_psi twist their odd-odd
                                                  * clover term with +-(mu + mu3) (clovertm_operators.c:208,216,238,243,258,265) */

/* Upload g_gauge_field (host, su3[VOLUMEPLUSRAND][4], halo links included when
 * nproc_t > 1) and re-sort it into the device gauge copy.  Replaces
 * update_backward_gauge() (update_backward_gauge.c:185-242); call whenever
 * g_update_gauge_copy is set (Hopping_Matrix.c:135-139). */
int tmhip_set_gauge(tmhip_ctx *ctx, const void *gauge_field_lexic);

/* ---- device fields ------------------------------------------------------ */
int tmhip_field_alloc(tmhip_ctx *ctx, int kind, tmhip_field **out);
void tmhip_field_free(tmhip_ctx *ctx, tmhip_field *f);
/* host `spinor[nsites]` (AoS) <-> device.  EO fields: nsites <= V/2, e/o order of
 * geometry_eo.c:869-885.  FULL fields: nsites = V, lexicographic order. */
int tmhip_field_upload(tmhip_ctx *ctx, tmhip_field *f, const void *host_spinors, int nsites);
int tmhip_field_download(tmhip_ctx *ctx, tmhip_field *f, void *host_spinors, int nsites);
/* sites [first, first + count) of an fp64 field -> pinned_spinors[0 .. count), which must be page-locked memory (tmhip_pinned_alloc):
 * the kernel writes there directly and no staging buffer of the context is used, so this transfer alone may be issued from a second
 * host thread while another call is in progress (the fault handler of the drop-in's lazy mode).  FULL fields: first = 0, count = V. */
int tmhip_field_download_range(tmhip_ctx *ctx, tmhip_field *f, void *pinned_spinors, int first, int count);
/* page-locked host buffers (hipHostMalloc): staging for callers that keep the runtime away from their own pages */
int tmhip_pinned_alloc(unsigned long bytes, void **out);
int tmhip_pinned_free(void *p);
int tmhip_field_zero(tmhip_ctx *ctx, tmhip_field *f);
/* FULL field <-> its two parities (linalg/convert_eo_to_lexic.c) -- views, no copy */
tmhip_field *tmhip_field_even(tmhip_field *full);
tmhip_field *tmhip_field_odd(tmhip_field *full);

/* ---- stencil (EO fields) ------------------------------------------------ */
/* Hopping_Matrix(ieo, l, k)            operator/Hopping_Matrix.c:131-156 */
int tmhip_hopping_matrix(tmhip_ctx *ctx, int ieo, tmhip_field *l, tmhip_field *k);
/* Hopping_Matrix_nocom                  operator/Hopping_Matrix_nocom.c:48-56 (halo exchange skipped) */
int tmhip_hopping_matrix_nocom(tmhip_ctx *ctx, int ieo, tmhip_field *l, tmhip_field *k);
/* tm_times_Hopping_Matrix(ieo,l,k,c)    operator/tm_times_Hopping_Matrix.c:72 ; epilogue hopping.h:674-678 */
int tmhip_tm_times_hopping_matrix(tmhip_ctx *ctx, int ieo, tmhip_field *l, tmhip_field *k, double cre, double cim);
/* tm_sub_Hopping_Matrix(ieo,l,p,k,c)    operator/tm_sub_Hopping_Matrix.c:73 ; epilogue hopping.h:680-688 */
int tmhip_tm_sub_hopping_matrix(tmhip_ctx *ctx, int ieo, tmhip_field *l, tmhip_field *p, tmhip_field *k,
                                double cre, double cim);
/* D_psi(P,Q) on FULL fields             operator/D_psi.c:1133-1140, D_psi_body.c:266-375 */
int tmhip_D_psi(tmhip_ctx *ctx, tmhip_field *P, tmhip_field *Q);

/* ---- site-diagonal twisted-mass ops (operator/tm_operators.h:26-77) ------ */
int tmhip_mul_one_pm_imu_inv(tmhip_ctx *ctx, tmhip_field *l, double sign, int N);
int tmhip_assign_mul_one_pm_imu_inv(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k, double sign, int N);
int tmhip_assign_mul_one_pm_imu(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k, double sign, int N);
int tmhip_mul_one_pm_imu(tmhip_ctx *ctx, tmhip_field *l, double sign);
int tmhip_mul_one_pm_imu_sub_mul(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k, tmhip_field *j, double sign, int N);
int tmhip_mul_one_pm_imu_sub_mul_gamma5(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k, tmhip_field *j, double sign);
int tmhip_mul_one_sub_mul_gamma5(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k, tmhip_field *j); /* tm_operators.c:781-810 */
int tmhip_gamma5(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k, int N); /* gamma.c:77-98 */

/* ---- e/o compositions (operator/tm_operators.c) -------------------------- */
int tmhip_H_eo_tm_inv_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k, int ieo, double sign); /* :508-526 */
int tmhip_Qtm_plus_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k);   /* :172-177 */
int tmhip_Qtm_minus_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k);  /* :216-221 */
int tmhip_Mtm_plus_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k);   /* :245-250 */
int tmhip_Mtm_minus_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k);  /* :289-294 */
int tmhip_Qtm_pm_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k);     /* :338-345 */
int tmhip_M_full(tmhip_ctx *ctx, tmhip_field *Even_new, tmhip_field *Odd_new,
                 tmhip_field *Even, tmhip_field *Odd);                    /* :117-128 */
/* symmetric e/o preconditioning  1 - A^-1 H_oe A^-1 H_eo  (non-hermitian solvers of invert_eo.c:177-280) */
int tmhip_Qtm_plus_sym_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k);      /* :186-192 */
int tmhip_Qtm_minus_sym_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k);     /* :223-229 */
int tmhip_Mtm_plus_sym_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k);      /* :259-265 */
int tmhip_Mtm_minus_sym_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k);     /* :296-302 */
int tmhip_Mtm_plus_sym_dagg_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k); /* :312-322, l != k */
int tmhip_Qtm_pm_sym_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k);        /* :347-364 (returns what the reference returns) */

/* ---- spinor linalg (linalg/ of the reference) ---------------------------- */
/* `parallel` != 0 adds the cross-rank sum (MPI_Allreduce in the reference). */
int tmhip_square_norm(tmhip_ctx *ctx, tmhip_field *P, int N, int parallel, double *out);                       /* square_norm.c:253 */
int tmhip_scalar_prod_r(tmhip_ctx *ctx, tmhip_field *S, tmhip_field *R, int N, int parallel, double *out);     /* scalar_prod_r.c:135 */
int tmhip_assign_add_mul_r(tmhip_ctx *ctx, tmhip_field *P, tmhip_field *Q, double c, int N);                   /* assign_add_mul_r.c:346 */
int tmhip_assign_mul_add_r(tmhip_ctx *ctx, tmhip_field *R, double c, tmhip_field *S, int N);                   /* assign_mul_add_r.c:340 */
int tmhip_assign_mul_add_r_and_square(tmhip_ctx *ctx, tmhip_field *R, double c, tmhip_field *S, int N,
                                      int parallel, double *out);                                             /* assign_mul_add_r_and_square.c:145 */
int tmhip_diff(tmhip_ctx *ctx, tmhip_field *Q, tmhip_field *R, tmhip_field *S, int N);                         /* diff.c:270 */
int tmhip_assign(tmhip_ctx *ctx, tmhip_field *R, tmhip_field *S, int N);                                       /* assign.c:42 */
int tmhip_add(tmhip_ctx *ctx, tmhip_field *Q, tmhip_field *R, tmhip_field *S, int N);                          /* linalg/add.c:45   Q = R + S */
int tmhip_mul_r(tmhip_ctx *ctx, tmhip_field *R, double c, tmhip_field *S, int N);                              /* linalg/mul_r.c:40 R = c S */

/* ---- solver --------------------------------------------------------------- */
enum { TMHIP_OP_QTM_PM = 0, TMHIP_OP_QTM_PLUS = 1, TMHIP_OP_QTM_MINUS = 2, TMHIP_OP_MTM_PLUS = 3, TMHIP_OP_MTM_MINUS = 4,
       TMHIP_OP_QSW_PM = 5 /* clover: Qsw_pm_psi, needs tmhip_set_clover */ };
/* cg_her(P,Q,max_iter,eps_sq,rel_prec,N,f)   solver/cg_her.c:62-141.
 * Device-resident: P, Q and the three work fields never leave HBM.  Returns the
 * iteration count in *iters (-1 if not converged); res_hist (may be NULL) gets
 * err after each iteration, up to hist_len entries. */
int tmhip_cg_her(tmhip_ctx *ctx, tmhip_field *P, tmhip_field *Q, int max_iter, double eps_sq, int rel_prec,
                 int N, int op, int *iters, double *res_hist, int hist_len);

/* ---- fermion force, hopping part (deriv_Sb.c:401-700) ------------------------------
 * deriv_Sb(ieo, l, k, hf, factor) accumulates 2 factor trlambda(...) of the one-hop terms into hf->derivative.  Here the
 * accumulator is device-resident: zero it, call tmhip_deriv_Sb any number of times (one call per deriv_Sb call of the
 * reference, same ieo / l / k / factor), then fetch it in the host layout su3adj df[VOLUME][4] (8 doubles per link,
 * lexicographic sites), either overwriting or adding to the host array.  T-split ranks exchange the t=0 slices of both fields with the ring
 * neighbour first (xchange_2fields, deriv_Sb.c:102); tmhip_multi_deriv_Sb is the single-process ring of n contexts. */
int tmhip_derivative_zero(tmhip_ctx *ctx);
int tmhip_deriv_Sb(tmhip_ctx *ctx, int ieo, tmhip_field *l, tmhip_field *k, double factor);
int tmhip_derivative_download(tmhip_ctx *ctx, void *df, int accumulate);
int tmhip_multi_deriv_Sb(int n, tmhip_ctx **ctxs, int ieo, tmhip_field **l, tmhip_field **k, double factor);

/* Clover part of the force (monomial/cloverdet_monomial.c:110-147): the insertion matrices swm / swp (clover_leaf.c:141) are
 * device-resident; zero them, accumulate the spinor outer products (operator/clover_deriv.c:252) and the tr-log term
 * (clover_deriv.c:72; needs sw_inv of that parity), then tmhip_sw_all (operator/clover_accumulate_deriv.c:58) adds the sixteen
 * link derivatives per plane and site to the SAME derivative accumulator tmhip_deriv_Sb uses.  gauge_field = NULL reuses the
 * lexicographic copy kept by the last tmhip_sw_term.  tmhip_get_swpm returns su3 swm[VOLUME][4] / swp[VOLUME][4].
 * T-split ranks: sw_spinor_eo / sw_deriv are site-local; sw_all exchanges its contributions to the neighbours' links (see below). */
int tmhip_swpm_zero(tmhip_ctx *ctx);
int tmhip_sw_spinor_eo(tmhip_ctx *ctx, int ieo, tmhip_field *kk, tmhip_field *ll, double fac);
int tmhip_sw_deriv(tmhip_ctx *ctx, int ieo, double mu);
int tmhip_sw_all(tmhip_ctx *ctx, const void *gauge_field, double kappa, double c_sw);
/* sw_all on a T-split lattice held by n contexts of THIS process (peer copies instead of RCCL for the two-sided derivative halo
 * that xchange/xchange_deri.c ships); every context needs its own tmhip_sw_term (which keeps the links incl. halo slabs). */
int tmhip_multi_sw_all(int n, tmhip_ctx **ctxs, double kappa, double c_sw);
int tmhip_get_swpm(tmhip_ctx *ctx, void *swm, void *swp);

/* ---- clover twisted mass (invert_clover_eo.c:63-165) ------
 * The 6x6 site blocks are inputs like the gauge field: `sw` = su3 sw[VOLUME][3][2] from sw_term
 * (operator/clover_term.c:88), `sw_inv` = su3 sw_inv[VOLUME][4][2] from sw_invert(EE, mu)
 * (operator/clover_invert.c:170; +mu set in [0,V/2), -mu set in [V/2,V)).  Call again whenever they change. */
int tmhip_set_clover(tmhip_ctx *ctx, const void *sw, const void *sw_inv);
/* ... or computed on the device: sw_term(gf, kappa, c_sw) (operator/clover_term.c:88) from the host gauge field handed
 * over exactly as for tmhip_set_gauge, then sw_invert(ieo, mu) (operator/clover_invert.c:170; the operators below expect
 * ieo = 0 = EE as operator.c:364 uses it).  tmhip_get_clover copies the blocks back in the reference's host layouts
 * (either pointer may be NULL) for host code that still wants them (sw_trace, sw_deriv ...). */
int tmhip_sw_term(tmhip_ctx *ctx, const void *gauge_field, double kappa, double c_sw);
int tmhip_sw_invert(tmhip_ctx *ctx, int ieo, double mu);
int tmhip_get_clover(tmhip_ctx *ctx, void *sw, void *sw_inv);
int tmhip_clover_inv(tmhip_ctx *ctx, tmhip_field *l, int tau3sign, double mu);                                   /* clovertm_operators.c:287 */
int tmhip_clover_gamma5(tmhip_ctx *ctx, int ieo, tmhip_field *l, tmhip_field *k, tmhip_field *j, double mu);     /* :448 */
int tmhip_clover(tmhip_ctx *ctx, int ieo, tmhip_field *l, tmhip_field *k, tmhip_field *j, double mu);            /* :535 */
int tmhip_H_eo_sw_inv_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k, int ieo, int tau3sign, double mu);     /* :268 */
int tmhip_Qsw_pm_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k);                                            /* :233 */
int tmhip_Msw_plus_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k);                                          /* :256 */
int tmhip_Qsw_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k);        /* :201  Q-hat with mu = 0 in the diagonal term */
int tmhip_Qsw_plus_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k);   /* :217 */
int tmhip_Qsw_minus_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k);  /* :209; l may alias k (invert_clover_eo.c:128) */
int tmhip_Qsw_sq_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k);     /* :225 */
int tmhip_Msw_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k);        /* :247 */
int tmhip_Msw_minus_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k);  /* :261 */
int tmhip_Msw_full(tmhip_ctx *ctx, tmhip_field *Even_new, tmhip_field *Odd_new, tmhip_field *Even, tmhip_field *Odd);  /* :96-110 */
/* k = (1 + T + i mu g5) l on parity ieo / k = sw_inv l   (operator/assign_mul_one_sw_pm_imu_inv_block_body.c:1-72, 143-196) */
int tmhip_assign_mul_one_sw_pm_imu(tmhip_ctx *ctx, int ieo, tmhip_field *k, tmhip_field *l, double mu);
int tmhip_assign_mul_one_sw_pm_imu_inv(tmhip_ctx *ctx, int ieo, tmhip_field *k, tmhip_field *l, double mu);
int tmhip_Qsw_pm_psi_32(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k);                                         /* clovertm_operators_32.c */

/* ---- mixed precision -------------------------------------
 * fp32 one-parity fields hold the reference's `spinor32` (su3.h:65-68); the fp32 gauge copy is built
 * on first use from the links given to tmhip_set_gauge. */
int tmhip_field_alloc32(tmhip_ctx *ctx, tmhip_field **out);
int tmhip_field_upload32(tmhip_ctx *ctx, tmhip_field *f, const void *host_spinor32, int nsites);
int tmhip_field_download32(tmhip_ctx *ctx, tmhip_field *f, void *host_spinor32, int nsites);
int tmhip_assign_to_32(tmhip_ctx *ctx, tmhip_field *R32, tmhip_field *S64, int N);          /* linalg/assign_to_32.c */
int tmhip_assign_to_64(tmhip_ctx *ctx, tmhip_field *R64, tmhip_field *S32, int N);          /* linalg/assign_to_64.c */
int tmhip_add_from_32(tmhip_ctx *ctx, tmhip_field *P64, tmhip_field *X32, int N);           /* assign_to_64 + add, mixed_cg_her.c:158-159 */
int tmhip_hopping_matrix_32(tmhip_ctx *ctx, int ieo, tmhip_field *l, tmhip_field *k);       /* operator/Hopping_Matrix_32.c:97-127 */
int tmhip_Qtm_pm_psi_32(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k);                    /* operator/tm_operators_32.c Qtm_pm_psi_32 */
int tmhip_square_norm_32(tmhip_ctx *ctx, tmhip_field *P, int N, int parallel, double *out);
int tmhip_scalar_prod_r_32(tmhip_ctx *ctx, tmhip_field *S, tmhip_field *R, int N, int parallel, double *out);
int tmhip_assign_add_mul_r_32(tmhip_ctx *ctx, tmhip_field *P, tmhip_field *Q, float c, int N);
int tmhip_assign_mul_add_r_32(tmhip_ctx *ctx, tmhip_field *R, float c, tmhip_field *S, int N);
/* l = zc (.) k [- j] on HOST spinor32 arrays of any length N (zc = z on spin 0,1, conj(z) on spin 2,3; j may be NULL; l may alias
 * k or j): the fp32 instances mul_one_pm_imu_inv_32 / assign_mul_one_pm_imu_inv_32 / mul_one_pm_imu_sub_mul_32 that
 * operator/tm_operators.c:8-47 generates and solver/Msap.c calls on domain blocks.  Staged through the device, synchronous. */
int tmhip_diag32_host(tmhip_ctx *ctx, void *l, const void *k, const void *j, double zre, double zim, int N);
/* mixed_cg_her(P,Q,params,max_iter,eps_sq,rel_prec,N,f,f32)  solver/mixed_cg_her.c:65-202 with f = Qtm_pm_psi,
 * f32 = Qtm_pm_psi_32; innereps / max_inner_it are the reference's mixcg_innereps / mixcg_maxinnersolverit
 * (default_input_values.h:193-194: 5.0e-5, 5000).  *iters = the reference's return value (-1: not converged). */
int tmhip_mixed_cg_her(tmhip_ctx *ctx, tmhip_field *P, tmhip_field *Q, int max_iter, double eps_sq, int rel_prec, int N,
                       int op, double innereps, int max_inner_it, int *iters, int *outer_iters);
/* Restart points of the last tmhip_mixed_cg_her: inner_iters[i] = the reference's j of outer iteration i
 * (mixed_cg_her.c:152), at most `cap` of them; *n_outer = number of outer iterations run. */
int tmhip_mixed_cg_restarts(tmhip_ctx *ctx, int *inner_iters, int cap, int *n_outer);
/* solver/rg_mixed_cg_her.c:180-347: fp32 CG with reliable updates (restart when the iterated residual fell by `delta`
 * relative to its maximum since the last update) and an fp64 fail-safe.  *iters = iter_out + iter_in_sp + iter_in_dp
 * as the reference returns it, or -1; the three counters are reported separately when the pointers are non-NULL. */
int tmhip_rg_mixed_cg_her(tmhip_ctx *ctx, tmhip_field *P, tmhip_field *Q, int max_iter, double eps_sq, int rel_prec, int N,
                          int op, double delta, int *iters, int *iter_out, int *iter_in_sp, int *iter_in_dp);

/* ---- molecular-dynamics updates with the links resident in HBM -------------------------------------
 * tmhip_set_gauge keeps the lexicographic links it received on the device; per MD step the links are then updated in place
 * (update_gauge.c:51-110: U <- restoresu3(exposu3(step * P)) U), the halo slabs of a T-split rank are refreshed from the
 * ring neighbours (xchange_gauge) and the stencil's gauge copy is re-sorted (update_backward_gauge.c:185-242) without any
 * host <-> device copy of the gauge field.  Momenta: su3adj [VOLUME][4] = double [VOLUME][4][8] (hamiltonian_field_t::momenta).
 * After tmhip_update_gauge the clover blocks are stale: tmhip_sw_term(ctx, NULL, ...) recomputes them from the resident links. */
int tmhip_momenta_upload(tmhip_ctx *ctx, const void *host_momenta);
int tmhip_momenta_download(tmhip_ctx *ctx, void *host_momenta);
int tmhip_update_momenta(tmhip_ctx *ctx, double step);   /* update_momenta.c:67-72 from the device-resident derivative field */
int tmhip_update_gauge(tmhip_ctx *ctx, double step);     /* update_gauge.c:51-110 */
int tmhip_multi_update_gauge(int n, tmhip_ctx **ctxs, double step);   /* the same for n contexts of one process holding a T-split lattice (peer copies) */
int tmhip_gauge_download(tmhip_ctx *ctx, void *host_gauge);   /* [VOLUMEPLUSRAND][4] su3, e.g. at the end of a trajectory */

/* ---- ILDG gauge configurations (io/gauge_read.c:28-198, io/gauge_read_binary.c:140-200, io/gauge_write.c:22-59,
 *      io/gauge_write_binary.c:150-175, io/dml.c:49-60).  The "ildg-binary-data" record travels to / from the device as it lies in
 *      the file; byte swap, 32 <-> 64 bit conversion, site and link re-ordering and the SciDAC checksum happen in HBM (ildg.hip). -- */
typedef struct {
  int gauge_read;                       /* GaugeInfo.gaugeRead */
  unsigned suma, sumb;                  /* checksum calculated over the record (GaugeInfo.checksum) */
  unsigned suma_stored, sumb_stored;    /* ... and as stored in the "scidac-checksum" record */
  int prec, lx, ly, lz, lt;             /* "ildg-format" record */
  char xlf_info[1024];                  /* GaugeInfo.xlfInfo */
  char ildg_data_lfn[512];              /* GaugeInfo.ildg_data_lfn */
} tmhip_gauge_info;
/* this rank's part of the record (T LZ LY LX sites, x fastest; per site links x, y, z, t; big-endian; prec 64 | 32) -> resident links +
 * stencil gauge copy as after tmhip_set_gauge; sums[2] = SciDAC checksum A, B of this rank's sites (XOR over ranks = the file's) */
int tmhip_gauge_unpack_ildg(tmhip_ctx *ctx, const void *file_bytes, int prec, unsigned *sums);
int tmhip_gauge_pack_ildg(tmhip_ctx *ctx, void *file_bytes, int prec, unsigned *sums);
/* read_gauge_field(filename, gf) / write_gauge_field(filename, prec, xlfInfo), LIME framing included (both also on T-split ranks, collective:
 * every rank reads / writes its contiguous part of the record at its offset, the checksum is combined over the ranks, rank 0 writes the
 * framing records -- the file is byte for byte the one a single rank writes); the reader returns 0
 * or -1 with the reference's messages (io_checks = !g_disable_IO_checks; prec_expected = gauge_precision_read_flag);
 * host_gauge (may be NULL): the host's g_gauge_field to fill as well; xlf_info: the formatted "xlf-info" message or NULL */
int tmhip_read_gauge_field(tmhip_ctx *ctx, const char *filename, int prec_expected, int io_checks, void *host_gauge, tmhip_gauge_info *info);
int tmhip_write_gauge_field(tmhip_ctx *ctx, const char *filename, int prec, const char *xlf_info, unsigned *sums);

/* ---- multi-GPU halo exchange (replaces xchange_field / xchange_halffield,
 *      xchange/xchange_field.c:269-470, xchange/xchange_halffield.c:176-263) -- */
#define TMHIP_UNIQUE_ID_BYTES 128
int tmhip_comm_get_unique_id(char id[TMHIP_UNIQUE_ID_BYTES]);             /* rank 0, then broadcast by the host program */
int tmhip_comm_init(tmhip_ctx *ctx, const char id[TMHIP_UNIQUE_ID_BYTES]); /* ring of nproc_t ranks along T over RCCL */
/* The same ring WITHOUT RCCL: the ranks of a node meet in a POSIX shared-memory segment named after `job` (1 .. 64 characters, the same
 * string on every rank of the job, different between jobs -- e.g. the launcher's job id), and faces, halo slices and scalar sums travel
 * device -> page-locked host memory -> that segment -> device, stream-ordered on the compute stream with no overlap: the reference's own
 * MPI exchange on host memory (xchange/xchange_field.c:98-250, linalg/square_norm.c:299-316).  Not the fast path; it needs nothing but
 * the node's memory and lets several ranks share one GPU (how the multi-rank code is tested as real processes on a one-GPU box).
 * Collective over the ranks; replaces tmhip_comm_init for this context.  Sums are added in rank order: the same bits on every rank. */
int tmhip_comm_init_shm(tmhip_ctx *ctx, const char *job);
/* The direct face carrier, on top of either ring (collective, after tmhip_comm_init / tmhip_comm_init_shm): every rank maps its two
 * neighbours' receive buffers (hipIpcGetMemHandle / hipIpcOpenMemHandle; the handles travel over the communicator the context has) and
 * the kernels that produce the projected half-spinor faces store them straight into the neighbour's memory -- over xGMI between GPUs,
 * through the mapping between processes that share a GPU.  It replaces the MPI_Isend / MPI_Irecv / MPI_Waitall of
 * xchange/xchange_halffield.c:176-263 (operator/halfspinor_body.c:281-317) for the faces; sums and the other halos stay on the
 * communicator.  No copy, no kernel of a communication library, nothing on the receiver's compute units; on lattices whose boundary
 * waves fit the wait budget a stencil of a T-split rank is ONE kernel ("direct_form").  Non-zero (and the faces stay on the communicator,
 * on EVERY rank) when some rank cannot map a neighbour.  A single rank behind the one-rank RCCL communicator of
 * tmhip_comm_set_loopback(ctx, 2) may call it too: the collective set-up runs over RCCL with np = 1 and the rank becomes its own
 * neighbour (the single-GPU rehearsal of this function and of the direct sums). */
int tmhip_comm_init_ipc(tmhip_ctx *ctx);
/* 1 when the faces travel as direct stores (0: over the communicator); *sharers (may be NULL): ranks of the job on this rank's GPU */
int tmhip_comm_faces_direct(tmhip_ctx *ctx, int *sharers);
/* 1 when the scalar sums over the ranks travel as direct stores too (option "direct_sums", every rank's block mapped by every rank):
 * MPI_Allreduce of linalg/square_norm.c:314 as ONE wave per rank, added in rank order (the same bits on every rank) */
int tmhip_comm_sums_direct(tmhip_ctx *ctx);
/* tmhip_comm_init builds TWO communicators over the same ranks: one for the half-spinor faces (second HIP stream), one
 * (ncclCommSplit of the first) for the scalar all-reduces of the linalg (MPI_Allreduce in linalg/square_norm.c:314) and the
 * force halos on the main stream.  Ranks in each as RCCL reports them (ncclCommCount); 0, 0 before tmhip_comm_init. */
int tmhip_comm_count(tmhip_ctx *ctx, int *nranks_faces, int *nranks_reduce);
/* 1: the reductions have their own communicator; 0: ncclCommSplit was unavailable (or switched off with the "comm_split" option) and
 * they share the face communicator -- still correct, a face exchange is never in flight together with a reduction; -1: no communicator */
int tmhip_comm_is_split(tmhip_ctx *ctx);
/* Single-GPU self-test of the split-phase path: faces are packed, "exchanged"
 * with this rank itself and consumed by the exterior kernel.  on = 1: device-to-device
 * copies; on = 2: through a one-rank RCCL communicator (ncclSend/ncclRecv to self); on = 3: the direct carrier onto
 * oneself (the neighbours' receive buffers are this rank's own). */
int tmhip_comm_set_loopback(tmhip_ctx *ctx, int on);
/* Test hook: holds the comm stream back for `ms` milliseconds (<= 20000) in front of the next thing enqueued on it, i.e. the next halo
 * exchange -- what a late neighbour looks like from this rank (xchange_field's MPI_Waitall simply waits, xchange/xchange_field.c:98-250;
 * so does the split path here, up to "flag_timeout_ms").  Changes no result. */
int tmhip_comm_stream_delay_ms(tmhip_ctx *ctx, int ms);

/* Single-process ring of n contexts (ctxs[r] = rank r of an n-way T split; one per GPU, or several
 * on one GPU as a self-test): Hopping_Matrix on every slab with the faces moved by peer copies
 * (hipMemcpyPeerAsync) instead of RCCL.  Same pack / stencil / exterior kernels as the RCCL path. */
int tmhip_multi_hopping_matrix(int n, tmhip_ctx **ctxs, int ieo, tmhip_field **l, tmhip_field **k);

/* ---- measurement ---------------------------------------------------------- */
/* The benchmark.c:291-300 loop on device-resident fields: iters x {H(0,f1,f0); H(1,f2,f1)},
 * timed with HIP events on the context's stream.  ms_total = elapsed milliseconds. */
int tmhip_bench_hopping(tmhip_ctx *ctx, tmhip_field *f0, tmhip_field *f1, tmhip_field *f2, int iters, double *ms_total);
/* generic event slots (0..15) recorded on the context's compute stream */
int tmhip_event_record(tmhip_ctx *ctx, int slot);
int tmhip_event_elapsed_ms(tmhip_ctx *ctx, int slot_start, int slot_stop, double *ms);
/* Launch-shape and scheduling options; the defaults are the measured best.  Apart from "gauge_recon" (below) none of
 * them changes a result beyond rounding (reduction order, FMA contraction); unknown names are refused.
 * TMLQCD_HIP_OPTIONS="name=value,name=value" in the environment applies them when the context is created (executables linked against
 * the drop-in unmodified); a malformed or unknown entry fails tmhip_create.
 *   "block" 0|256|64 threads per block (0: automatic, 64 on local lattices of fewer than 131072 sites per parity)
 *   "xcd"   block order: 2 automatic (default; tile order up to L = 32, slab order above -- and on lattices of fewer than 24 time-slices
 *           whose time-slices are large (L >= 24): one rank's share of a T split, profiles/r04_shape_sweep.log), 0 none, 1 one chunk per XCD,
 *           3 slab, 4 tile;  "tgrp" time-slices per tile group (0 = automatic)
 *   "occ" / "occ32"  waves per SIMD allowed by a dynamic-LDS cap for the fp64 / fp32 stencil (3 / 0 = no cap)
 *   "minw" 4: __launch_bounds__(BS, 4)
 *   "split_sync" 0 (default) | 1: T-split ranks -- 0: the exterior kernel (main stream) and the pack kernel (comm stream) wait on the device for a
 *                flag of the other stream; 1: the two streams are ordered by HIP events, no wait on the device at all (slower: two events on the
 *                main stream per stencil)
 *                ("split_pipe" of round 3 is gone: slower than the default form since the stencil kernel has one store path, and the direct carrier
 *                does what it was for -- faces of a chain's next stencil on their way while the current one runs -- without an RCCL kernel;
 *                profiles/r04_split_forms.md)
 *   "direct_form" -1 (default) | 0 | 1: the direct carrier (tmhip_comm_init_ipc) -- 1: one kernel per stencil (the boundary waves wait for their
 *                neighbour's word after their seven local hops, add the hop across the cut, project their output and store the projection into the
 *                neighbour's buffer); 0: stencil kernel + exterior kernel (which waits and pushes); -1: one kernel while the boundary waves of a
 *                launch are at most 2048 (fp64 clover epilogues: 1792; ranks sharing a GPU in a rehearsal: 1024 / their number)
 *   "direct_sums" 1 (default) | 0 (before tmhip_comm_init_ipc): with the direct carrier every rank maps every rank's block and the scalar sums
 *                over the ranks (square_norm .. with parallel = 1, the alpha and the stopping test of cg_her) are one wave that stores this
 *                rank's partial sum into every rank's block and adds up its own row in rank order (the same bits on every rank) instead
 *                of an ncclAllReduce
 *   "direct_order" bit 0 / bit 1: one-kernel form -- boundary time-slices dispatched first (else last) in a stencil whose faces are packed now /
 *                were pushed ahead by the stencil before (default 3: first in both; with the slab order of short lattices the boundary slices are
 *                spread over all XCDs and going first is worth 2 - 3 % at T_local 8 / 16, profiles/r04_split_forms_ab.log)
 *   "prepack" 1 (default) | 0: T-split ranks -- the exterior kernel also projects the completed boundary slices of its output into the send buffers, so
 *                the next stencil of a chain (Qtm_pm_psi, a fused CG iteration) starts its exchange without a pack kernel
 *   "flag_timeout_ms" bound of those device-side waits (default 120 s, or TMLQCD_HIP_FLAG_TIMEOUT_S in the environment; 0 = none): a late
 *                neighbour is waited for, a dead one becomes an error of the next synchronising call (reported once, then cleared)
 *   "comm_split" 1|0 (before tmhip_comm_init / tmhip_comm_set_loopback(2)): 0 keeps the reductions on the face communicator (the fallback of an RCCL without ncclCommSplit)
 *   "cg_fused_dot" 2 (default: alpha / residual / norm in the stencil epilogues), 1 scalar product only, 0 plain linalg kernels
 *   "cg_self" 1 (default) | 0: small unsplit lattices (the hop-split stencil) -- the fused CG iteration adds up its partial sums inside the residual stencil
 *                (alpha) and the (P, p) kernel (stopping test, beta) instead of two one-block sum + scalar kernels in between
 *   "cg_sync" 1: host-side scalars as in the reference loop;  "cg_batch" n: iterations enqueued between two polls of `done`
 *   "gauge_cache" -1 (automatic) / 0 / 1: the 64-thread stencil launches of small unsplit lattices, and the stencil launches of a T-split rank,
 *                  load the links with (0) or without (1) the streaming hint; automatic = without while the gauge copy is <= 200 MB (it then
 *                  stays in the Infinity Cache between calls: 4 x 32^3 per rank 65 -> 80 % of the unsplit rate, profiles/r04_gcache_ab.log)
 *   "swall_order" 0 / 1 / 2: block order of the owner-computes sw_all (one chunk per XCD / slab order / sw_term's tile order, default: fabric reads
 *                6.6 -> 4.0 GB per launch at 32^4, 3 % faster -- the kernel is bound by its 43 dependent 3x3 products per link, not by bytes;
 *                profiles/r04_swall_ab.log)
 *   "swterm_order" 0 / 1: block order of sw_term (one chunk per XCD / small (x, y) tiles walked through all time-slices, default)
 * One option changes what is read from memory:
 * "gauge_recon" = 12 makes the twisted-mass stencil launches (fp64 and fp32) fetch only the first two rows of every link and
 * rebuild the third as conj(row0 x row1) in registers (the 12-real compression the reference exposes for its external
 * inverters, misc_types.h:29-33 COMPRESSION_12) -- 25 % fewer bytes per site.  It is opt-in and guarded: the links
 * of the resident gauge field must be SU(3) to 1e-13 (measured on the device at set_gauge / when the option is set),
 * otherwise the option is refused with a message and the full 18-real read stays in force. */
int tmhip_set_option(tmhip_ctx *ctx, const char *name, int value);
/* max |row2 - conj(row0 x row1)| over all links of the resident gauge field */
int tmhip_gauge_su3_deviation(tmhip_ctx *ctx, double *maxdev);

#ifdef __cplusplus
}
#endif
#endif
