"""A leapfrog trajectory of the clover determinant monomial at L^4 (default 32) with everything resident in HBM (the class of
tests/test_gpu_md_trajectory.py): wall time per molecular-dynamics step, split into the solve and the rest, dH and the CG iterations.
Usage: python tools/md_trajectory.py [L] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_md_trajectory import CloverDetTrajectory  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
tr = CloverDetTrajectory(L=L, mu=0.05, c_sw=1.2)
h0 = tr.energy()
solve0 = tr.solve
t_solve = [0.0]


def timed_solve():
    tr.lat.sync(); t = time.perf_counter(); solve0(); tr.lat.sync(); t_solve[0] += time.perf_counter() - t


tr.solve = timed_solve
tr.iters = 0
tr.lat.sync()
t0 = time.perf_counter()
tr.leapfrog(steps, 0.1 / steps)
tr.lat.sync()
dt = time.perf_counter() - t0
nf = steps + 1
print("%d^4 clover determinant, %d leapfrog steps (tau = 0.1, %d force evaluations): %.1f ms per force evaluation = %.1f ms solve "
      "(cg_her to 1e-13, %d iterations each, incl. sw_term + sw_invert) + %.1f ms force kernels and link / momentum update"
      % (L, steps, nf, 1e3 * dt / nf, 1e3 * t_solve[0] / nf, tr.iters // nf, 1e3 * (dt - t_solve[0]) / nf), flush=True)
print("dH = %.3e on H = %.6e" % (tr.energy() - h0, h0))
tr.close()
