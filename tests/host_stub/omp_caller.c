/* A caller in the shape of operator/tm_operators_32.c:94-112 (Qtm_pm_psi_32): it opens an OpenMP parallel region and EVERY thread of
 * the team calls the `_orphaned` operator, exactly once each.  Returns the team size.  Built by tests/test_gpu_dropin32.py with -fopenmp. */
#include <omp.h>
typedef void (*hop32_fn)(const int, void *, void *);
int omp_team_calls_orphaned(hop32_fn f, int nthreads, void *l1, void *k1, void *l2) {
  int team = 0;
#pragma omp parallel num_threads(nthreads)
  {
#pragma omp master
    team = omp_get_num_threads();
    f(0, l1, k1);        /* H_eo: every thread, as Hopping_Matrix_32_orphaned(EO, ...) in the reference */
    f(1, l2, l1);        /* H_oe of that result: needs the first call complete for ALL threads before anyone goes on */
  }
  return team;
}
