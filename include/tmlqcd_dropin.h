/* tmlqcd_dropin.h -- the reference's own symbols, implemented on MI355X (libtmlqcd_dropin.so).
 *
 * Every function below has EXACTLY the name, signature and argument meaning of the
 * tmLQCD function cited next to it, so `benchmark`, `invert` and `hmc_tm` link against
 * this library instead of the corresponding objects of liboperator.a / liblinalg.a /
 * libsolver.a (link line configure.in:1074) with no source change.  Host arrays stay in
 * the reference's AoS `spinor` / `su3` layouts (su3.h:40-63).
 *
 * The library reads these tmLQCD globals AT CALL TIME (never cached across calls):
 *   T, LX, LY, LZ, VOLUME, RAND (global.h:82-84), g_nproc_t/x/y/z, g_proc_coords (global.h:206-207),
 *   g_gauge_field, g_update_gauge_copy (global.h:176,73), ka0..ka3 (boundary.h:25), g_mu (global.h:198),
 *   and -- weak references, a host program without them gets 0 / the plain branch -- g_mu3 (clover odd-odd twist,
 *   global.h:197), g_c_sw (clover branch of D_psi, global.h:198), sw / sw_inv (clovertm_operators.c:58-59),
 *   mixcg_innereps, mixcg_maxinnersolverit (read_input.h:112-113), g_prec_sequence_d_dagger_d, g_precWS, update_backward_gauge.
 *
 * Residency (SURVEY §7 "hard parts"): the reference API passes host pointers.  Two modes:
 *   COHERENT (default) every call uploads its inputs and downloads its outputs: always
 *            correct for unmodified callers, PCIe-bound.
 *   RESIDENT           outputs stay in HBM; a host array is uploaded only when it has no
 *            valid device mirror.  The host copy of an output is stale until
 *            tmlqcd_hip_sync_to_host(); after the host writes an array call
 *            tmlqcd_hip_host_modified().  cg_her() always runs device-resident
 *            internally and returns with P valid on the host.
 * Errors are fatal (print + exit), the reference's convention (fatal_error.c).
 */
#ifndef TMLQCD_DROPIN_H
#define TMLQCD_DROPIN_H
#ifdef __cplusplus
extern "C" {
#define TM_COMPLEX double _Complex
#else
#include <complex.h>
#define TM_COMPLEX double _Complex
#endif

/* su3.h:40-63 -- layout is ABI */
typedef struct { TM_COMPLEX c00, c01, c02, c10, c11, c12, c20, c21, c22; } su3;
typedef struct { TM_COMPLEX c0, c1, c2; } su3_vector;
typedef struct { su3_vector s0, s1, s2, s3; } spinor;
typedef void (*matrix_mult)(spinor *const, spinor *const); /* solver/matrix_mult_typedef.h:30 */

/* ---- stencil ------------------------------------------------------------- */
void Hopping_Matrix(const int ieo, spinor *const l, spinor *const k);        /* operator/Hopping_Matrix.h:30 */
void Hopping_Matrix_nocom(const int ieo, spinor *const l, spinor *const k);  /* operator/Hopping_Matrix_nocom.h */
void tm_times_Hopping_Matrix(const int ieo, spinor *const l, spinor *const k, TM_COMPLEX const cfactor); /* operator/tm_times_Hopping_Matrix.h */
void tm_sub_Hopping_Matrix(const int ieo, spinor *const l, spinor *p, spinor *const k, TM_COMPLEX const cfactor); /* operator/tm_sub_Hopping_Matrix.h */
void D_psi(spinor *const P, spinor *const Q);                                 /* operator/D_psi.h:27 */

/* ---- operator/tm_operators.h:26-77 ---------------------------------------- */
void Qtm_plus_psi(spinor *const l, spinor *const k);
void Qtm_plus_psi_nocom(spinor *const l, spinor *const k);
void Qtm_minus_psi(spinor *const l, spinor *const k);
void Mtm_plus_psi(spinor *const l, spinor *const k);
void Mtm_plus_psi_nocom(spinor *const l, spinor *const k);
void Mtm_minus_psi(spinor *const l, spinor *const k);
/* symmetric e/o preconditioning, operator/tm_operators.h:57-65 */
void Qtm_plus_sym_psi(spinor *const l, spinor *const k);
void Qtm_plus_sym_psi_nocom(spinor *const l, spinor *const k);
void Qtm_minus_sym_psi(spinor *const l, spinor *const k);
void Mtm_plus_sym_psi(spinor *const l, spinor *const k);
void Mtm_plus_sym_dagg_psi(spinor *const l, spinor *const k);
void Mtm_minus_sym_psi(spinor *const l, spinor *const k);
void Mtm_plus_sym_psi_nocom(spinor *const l, spinor *const k);
void Mtm_minus_sym_psi_nocom(spinor *const l, spinor *const k);
void Qtm_pm_sym_psi(spinor *const l, spinor *const k);
void Qtm_pm_psi(spinor *const l, spinor *const k);
void Qtm_pm_psi_nocom(spinor *const l, spinor *const k);
void H_eo_tm_inv_psi(spinor *const l, spinor *const k, const int ieo, const double sign);
void M_full(spinor *const Even_new, spinor *const Odd_new, spinor *const Even, spinor *const Odd);
void Q_full(spinor *const Even_new, spinor *const Odd_new, spinor *const Even, spinor *const Odd);
void M_minus_1_timesC(spinor *const Even_new, spinor *const Odd_new, spinor *const Even, spinor *const Odd);
void mul_one_pm_imu_inv(spinor *const l, const double _sign, const int N);
void assign_mul_one_pm_imu_inv(spinor *const l, spinor *const k, const double _sign, const int N);
void assign_mul_one_pm_imu(spinor *const l, spinor *const k, const double _sign, const int N);
void mul_one_pm_imu(spinor *const l, const double _sign);
void mul_one_pm_imu_sub_mul(spinor *const l, spinor *const k, spinor *const j, const double _sign, const int N);
void mul_one_pm_imu_sub_mul_gamma5(spinor *const l, spinor *const k, spinor *const j, const double _sign); /* tm_operators.c:813 */
void mul_one_sub_mul_gamma5(spinor *const l, spinor *const k, spinor *const j);                              /* tm_operators.c:781 */
void Mee_psi(spinor *const l, spinor *const k, const double mu);
void Mee_inv_psi(spinor *const l, spinor *const k, const double mu);
void Q_pm_psi(spinor *const l, spinor *const k);
void Q_plus_psi(spinor *const l, spinor *const k);
void Q_minus_psi(spinor *const l, spinor *const k);
void M_minus_psi(spinor *const l, spinor *const k);
void D_dagg_psi(spinor *const l, spinor *const k);
typedef struct { float c[24]; } spinor32;   /* su3.h:83: spinor32 = 4 x su3_vector32 = 12 complex float */
/* fp32 twins on HOST spinor32 arrays by their reference names (row f1).  Operands are copied in and results out on every call (the
 * coherent semantics; the fp32 solvers iterate device-resident through mixed_cg_her / rg_mixed_cg_her below).  N <= VOLUME/2. */
void Hopping_Matrix_32(const int ieo, spinor32 *const l, spinor32 *const k);              /* operator/Hopping_Matrix_32.h:31 */
void Hopping_Matrix_32_orphaned(const int ieo, spinor32 *const l, spinor32 *const k);     /* :30 -- called by EVERY thread of an enclosing OpenMP team: one issues the device call, all meet before and after */
void Qtm_pm_psi_32(spinor32 *const l, spinor32 *const k);                                 /* operator/tm_operators_32.c:94 */
float square_norm_32(const spinor32 *const P, const int N, const int parallel);           /* linalg/square_norm_32.c:95 */
float scalar_prod_r_32(const spinor32 *const S, const spinor32 *const R, const int N, const int parallel);   /* linalg/scalar_prod_r_32.c:109 */
void assign_add_mul_r_32(spinor32 *const R, spinor32 *const S, const float c, const int N);                  /* linalg/assign_add_mul_r_32.c:104 */
void assign_mul_add_r_32(spinor32 *const R, const float c, const spinor32 *const S, const int N);            /* linalg/assign_mul_add_r_32.c:81 */
void diff_32(spinor32 *const Q, const spinor32 *const R, const spinor32 *const S, const int N);              /* linalg/diff_32.c:39 */
void assign_to_32(spinor32 *const R, spinor *const S, const int N);                       /* linalg/assign_to_32.c:37 */
void assign_to_64(spinor *const R, spinor32 *const S, const int N);                       /* linalg/assign_to_32.c:84 (N = VOLUME/2) */
void mul_one_pm_imu_inv_32(spinor32 *const l, const double _sign, const int N);                                        /* tm_operators.c:74 */
void assign_mul_one_pm_imu_inv_32(spinor32 *const l, spinor32 *const k, const double _sign, const int N);              /* tm_operators.h */
void mul_one_pm_imu_sub_mul_32(spinor32 *const l, spinor32 *const k, spinor32 *const j, const double _sign, const int N); /* tm_operators.c:103 */
void Q_pm_psi_prec(spinor *const l, spinor *const k);   /* tm_operators.c:402; spinorPrecondition stays reference code (weak reference) */
void Q_pm_psi2(spinor *const l, spinor *const k);       /* tm_operators.c:453 */
void Q_pm_psi_gpu(spinor *const l, spinor *const k);    /* tm_operators.c:440; gamma5 is applied to k IN PLACE first */
void Q_minus_psi_gpu(spinor *const l, spinor *const k); /* tm_operators.c:476; likewise */
void Q_psi(spinor *const P, spinor *const Q);
void gamma5(spinor *const l, spinor *const k, const int V);                   /* gamma.h */

void tmlqcd_hip_comm_init(const char unique_id[128]); /* ranks along T: id from tmhip_comm_get_unique_id, broadcast (MPI_Bcast) by the host */
void tmlqcd_hip_comm_init_shm(const char *job);         /* the same ring without RCCL: host-staged through a shared-memory segment of the node (tmhip_comm_init_shm) */
int tmlqcd_hip_comm_init_ipc(void);                      /* after either: the half-spinor faces travel as direct stores into the ring neighbours' IPC-mapped receive buffers
                                                          * (tmhip_comm_init_ipc; replaces the MPI_Isend/Irecv/Waitall of xchange/xchange_halffield.c:176-263).  Collective.
                                                          * Non-zero when some rank cannot map a neighbour: every rank then keeps the communicator's exchange */
double square_norm(const spinor *const P, const int N, const int parallel);
double scalar_prod_r(const spinor *const S, const spinor *const R, const int N, const int parallel);
void assign_add_mul_r(spinor *const P, spinor *const Q, const double c, const int N);
void assign_mul_add_r(spinor *const R, const double c, const spinor *const S, const int N);
double assign_mul_add_r_and_square(spinor *const R, const double c, const spinor *const S, const int N, const int parallel);
void diff(spinor *const Q, const spinor *const R, const spinor *const S, const int N);
void assign(spinor *const R, spinor *const S, const int N);
void add(spinor *const Q, const spinor *const R, const spinor *const S, const int N);   /* linalg/add.h */
void mul_r(spinor *const R, const double c, spinor *const S, const int N);               /* linalg/mul_r.h */

/* ---- solver/cg_her.h ------------------------------------------------------- */
int cg_her(spinor *const P, spinor *const Q, const int max_iter, double eps_sq, const int rel_prec,
           const int N, matrix_mult f);

/* ---- operator/clovertm_operators.h ----------------------- */
/* The host keeps computing `sw` / `sw_inv` with its own sw_term / sw_invert (operator.c:329-330,364); the library
 * reads the globals `su3 ***sw, ***sw_inv` (clovertm_operators.c:58-59).  They carry no dirty flag in the reference,
 * so call tmlqcd_hip_update_clover() after every sw_term / sw_invert (first use uploads automatically). */
void tmlqcd_hip_update_clover(void);
void Qsw_pm_psi(spinor *const l, spinor *const k);                                                  /* clovertm_operators.c:233 */
void Msw_plus_psi(spinor *const l, spinor *const k);                                                /* :256 */
void Qsw_psi(spinor *const l, spinor *const k);                                                     /* :201 */
void Qsw_plus_psi(spinor *const l, spinor *const k);                                                /* :217 */
void Qsw_minus_psi(spinor *const l, spinor *const k);                                               /* :209 (in place in invert_clover_eo.c:128) */
void Qsw_sq_psi(spinor *const l, spinor *const k);                                                  /* :225 */
void Msw_psi(spinor *const l, spinor *const k);                                                     /* :247 */
void Msw_minus_psi(spinor *const l, spinor *const k);                                               /* :261 */
void Msw_full(spinor *const Even_new, spinor *const Odd_new, spinor *const Even, spinor *const Odd); /* :96 */
void assign_mul_one_sw_pm_imu(const int ieo, spinor *const k, spinor *const l, const double mu);     /* assign_mul_one_sw_pm_imu_inv_block_body.c:1 */
void assign_mul_one_sw_pm_imu_inv(const int ieo, spinor *const k, spinor *const l, const double mu); /* :143 */
void Mee_sw_psi(spinor *const k, spinor *const l, const double mu);                                 /* clovertm_operators.c:873 */
void Mee_sw_inv_psi(spinor *const k, spinor *const l, const double mu);                             /* :1098 */
void H_eo_sw_inv_psi(spinor *const l, spinor *const k, const int ieo, const int tau3sign, const double mu); /* :268 */
void clover_inv(spinor *const l, const int tau3sign, const double mu);                              /* :287 */
void clover_gamma5(const int ieo, spinor *const l, const spinor *const k, const spinor *const j, const double mu); /* :448 */
void clover(const int ieo, spinor *const l, const spinor *const k, const spinor *const j, const double mu);        /* :535 */

/* ---- solver/mixed_cg_her.h ------------------------------ */
/* `solver_params_t` (solver/solver_params.h:46-109) is passed BY VALUE.  Being larger than 16 bytes it travels in
 * memory under the SysV ABI, ahead of the stack-passed `f32`, so the callee must know its exact size and the offset
 * of the one field read on this path (mcg_delta, solver_params.h:68).  The mirror below restates that layout --
 * field order and types are ABI, like `spinor` and `su3` -- with neutral names for the fields this library never
 * reads.  tests/test_gpu_link_reference_caller.py has reference code build the struct and call through it. */
typedef void (*matrix_mult32)(void *const, void *const);   /* solver/matrix_mult_typedef.h:32 */
typedef struct {
  int eigcg_i[5];                  /* eigcg_nrhs .. eigcg_ldh */
  double eigcg_d[3];               /* eigcg_tolsq1, eigcg_tolsq, eigcg_restolsq */
  int eigcg_rand_guess_opt;
  float mcg_delta;                 /* reliable-update threshold of rg_mixed_cg_her */
  int type, max_iter, rel_prec, no_shifts, sdim;
  double squared_solver_prec;
  void (*M_psi)(spinor *const, spinor *const);
  matrix_mult32 M_psi32;
  void (*M_ndpsi)(spinor *const, spinor *const, spinor *const, spinor *const);
  void (*M_ndpsi32)(void *const, void *const, void *const, void *const);
  double *shifts;
  int solution_type, compression_type, sloppy_precision, external_inverter;   /* enums, misc_types.h */
} tmlqcd_solver_params;
/* solver/mixed_cg_her.c:65-202 reads the globals mixcg_innereps / mixcg_maxinnersolverit (read_input.h:112-113),
 * not the struct. */
int mixed_cg_her(spinor *const P, spinor *const Q, tmlqcd_solver_params solver_params, const int max_iter,
                 double eps_sq, const int rel_prec, const int N, matrix_mult f, matrix_mult32 f32);
/* solver/rg_mixed_cg_her.c:180 */
int rg_mixed_cg_her(spinor *const P, spinor *const Q, tmlqcd_solver_params solver_params, const int max_iter,
                    const double eps_sq, const int rel_prec, const int N, matrix_mult f, matrix_mult32 f32);

/* ---- deriv_Sb.h: hopping part of the fermion force -------- */
typedef struct { double d1, d2, d3, d4, d5, d6, d7, d8; } su3adj;           /* su3adj.h:23-26 */
typedef struct {                                                             /* hamiltonian_field.h:26-32 */
  su3 **gaugefield;
  su3adj **momenta;
  su3adj **derivative;
  int update_gauge_copy;
  int traj_counter;
} hamiltonian_field_t;
/* deriv_Sb.c:401.  Coherent mode: the contribution is added to hf->derivative before returning.  Resident mode: it
 * stays in the device accumulator until tmlqcd_hip_flush_derivative(hf) adds it to hf->derivative (call that once,
 * after the last deriv_Sb of a force computation, e.g. at the end of det_derivative, monomial/det_monomial.c:56-117). */
void deriv_Sb(const int ieo, spinor *const l, spinor *const k, hamiltonian_field_t *const hf, const double factor);
void tmlqcd_hip_flush_derivative(hamiltonian_field_t *const hf);
/* the clover part of cloverdet_derivative (monomial/cloverdet_monomial.c:67-72,125-147) with swm / swp resident in HBM: replace the
 * zeroing loop, sw_spinor_eo (operator/clover_deriv.c:252), sw_deriv (:72) and sw_all (operator/clover_accumulate_deriv.c:58) calls;
 * the derivative follows the same coherent / resident rule as deriv_Sb */
void tmlqcd_hip_swpm_zero(void);
void tmlqcd_hip_sw_spinor_eo(const int ieo, const spinor *const kk, const spinor *const ll, const double fac);
void tmlqcd_hip_sw_deriv(const int ieo, const double mu);
void tmlqcd_hip_sw_all(hamiltonian_field_t *const hf, const double kappa, const double c_sw);

/* Molecular-dynamics link update with the links resident in HBM: replaces the call update_gauge(step, hf)
 * (update_gauge.c:51-110, called from the integrators, integrator.c) -- under its own name because the reference's
 * update_gauge.o stays on the link line for programs that do not want it.  Coherent mode: hf->gaugefield is current when
 * the call returns.  Resident mode: the host links stay behind until tmlqcd_hip_sync_gauge_to_host(hf) (call it before
 * host code reads g_gauge_field: gauge-action force, measurements, I/O).  tmlqcd_hip_update_momenta is update_momenta.c:67-72
 * for a derivative accumulated on the device only (resident deriv_Sb / tmlqcd_hip_sw_all). */
void tmlqcd_hip_update_gauge(const double step, hamiltonian_field_t *const hf);
void tmlqcd_hip_sync_gauge_to_host(hamiltonian_field_t *const hf);
void tmlqcd_hip_update_momenta(const double step, hamiltonian_field_t *const hf);
void tmlqcd_hip_sync_momenta_to_host(hamiltonian_field_t *const hf);

/* ---- residency control (additions; not in the reference) ------------------- */
/* COHERENT (default): every call uploads its inputs and downloads its outputs -- exact drop-in, PCIe-bound.
 * RESIDENT: outputs stay in HBM until tmlqcd_hip_sync_to_host; the host program says when it touches a field.
 * LAZY: as RESIDENT, but the library finds out by itself: the pages of a host array whose current copy is in HBM are made
 *   inaccessible, the host's first load from one of them faults and fetches that page (or, if it keeps reading or stores, the field);
 *   arrays the device has read are write-protected, so a host store invalidates the mirror.  An UNMODIFIED host program then runs its
 *   stencil / operator loops at the HBM rate (also: environment TMLQCD_HIP_RESIDENCY=lazy).  Opt-in, for two reasons.  System calls
 *   do not fault: a field passed to write(2) / MPI while its host copy is stale must be synchronised first (tmlqcd_hip_sync_to_host).
 *   And arrays that go back to the allocator: a block the allocator unmaps (glibc: anything above its mmap threshold; the library
 *   leaves the program's malloc settings alone) is recognised when its address comes back -- a watched mirror is probed before it is
 *   trusted.  An array INSIDE a malloc arena (the main heap or a thread's: glibc serves a request from an arena whenever a free chunk
 *   fits), or in a shared / file-backed / named mapping, is never watched -- recognised from /proc/self/maps when it is first seen, it is
 *   copied on every call as in the coherent mode.  tmLQCD's fields, one calloc of hundreds of MB, are mappings of their own.  The SIGSEGV
 *   handler allocates nothing.  TMLQCD_HIP_LAZY_DEBUG=1 in the environment reports which arrays are not watched and why, and a SIGSEGV that
 *   is not the library's before it is passed on to the program's own handler. */
enum { TMLQCD_HIP_COHERENT = 0, TMLQCD_HIP_RESIDENT = 1, TMLQCD_HIP_LAZY = 2 };
/* Device versions of sw_term(g_gauge_field, kappa, c_sw) / sw_invert(ieo, mu) (operator/clover_term.c:88,
 * operator/clover_invert.c:170).  They carry their own names because the reference keeps other, unrelated functions in
 * the same objects (six_det, sw_invert_nd, sw_trace ...), so those objects stay on the link line; replace the two calls
 * in operator.c:329-330,364 / the clover monomials to use them.  The host's sw / sw_inv arrays receive copies. */
void tmlqcd_hip_sw_term(const double kappa, const double c_sw);
void tmlqcd_hip_sw_invert(const int ieo, const double mu);
void tmlqcd_hip_set_residency(int mode);
/* lazy mode counters: page faults served, pages fetched one by one, whole-field fetches, host stores noticed */
void tmlqcd_hip_lazy_stats(unsigned long out[4]);
/* Upper bound on the number of device mirrors kept for host arrays (default 64, also TMLQCD_HIP_MAX_MIRRORS): beyond it the
 * least recently used mirror whose host copy is current is freed.  Host programs that allocate work fields per solve
 * (solver/solver_field.c) hand in ever new addresses; mirrors holding device-only data (resident mode) are never dropped. */
void tmlqcd_hip_set_max_mirrors(int n);
unsigned long tmlqcd_hip_calls(void);   /* number of reference-named entry points served so far (integration checks) */
void tmlqcd_hip_sync_to_host(spinor *field);       /* download the device mirror of `field` if it is newer */
void tmlqcd_hip_sync_all_to_host(void);
void tmlqcd_hip_host_modified(spinor *field);      /* the host wrote `field`: drop its device mirror */
void tmlqcd_hip_forget(spinor *field);             /* host memory is being freed: drop the mirror */
void tmlqcd_hip_set_device(int device);            /* before the first call; default: $TMLQCD_HIP_DEVICE or 0 */
void tmlqcd_hip_comm_init(const char unique_id[128]); /* ranks along T: id from tmhip_comm_get_unique_id, MPI_Bcast by the host */
void tmlqcd_hip_finalize(void);
/* ---- ILDG gauge configurations: replaces io/gauge_read.o and io/gauge_write.o of libio.a -- */
typedef struct { unsigned int suma, sumb; } DML_Checksum;                        /* io/dml.h:34-37 */
typedef struct {                                                                 /* io/params.h:98-104 */
  double plaquetteEnergy;
  int gaugeRead;
  DML_Checksum checksum;
  char *xlfInfo;
  char *ildg_data_lfn;
} paramsGaugeInfo;
typedef struct {                                                                 /* io/params.h:71-88 */
  char date[64];
  char package_version[32];
  double beta, c2_rec, epsilonbar, kappa, mu, mubar, plaq;
  int counter;
  long int time;
} paramsXlfInfo;
extern paramsGaugeInfo GaugeInfo;                                                /* io/gauge_read.c:27 */
/* reads gauge_precision_read_flag and g_disable_IO_checks at call time (both optional: weak references, defaults 64 / checks on) */
int read_gauge_field(char *filename, su3 **const gf);                            /* io/gauge.h, io/gauge_read.c:28 */
int write_gauge_field(char *filename, const int prec, paramsXlfInfo const *xlfInfo);   /* io/gauge_write.c:22 */
/* benchmark.c:291-300 on device-resident mirrors; returns seconds for `iters` x {H(0,f1,f0); H(1,f2,f1)} */
double tmlqcd_hip_benchmark_loop(spinor *f0, spinor *f1, spinor *f2, int iters);

#ifdef __cplusplus
}
#endif
#endif
