"""One rank of tests/test_gpu_multiprocess.py::test_md_trajectory_between_real_processes: `python mp_md_worker.py RANK WORLD JOB OUTDIR`.
A leapfrog trajectory of the clover determinant (the force statements of tests/test_gpu_md_trajectory.py::CloverDetTrajectory:
sw_term / sw_invert from the moving links, cg_her on Qsw_pm_psi, deriv_Sb twice, sw_spinor_eo, sw_deriv, sw_all, update_momenta,
update_gauge with its halo exchange) on this rank's slab of a 8 x 8^3 lattice, everything resident in HBM, the ranks meeting through the
shared-memory transport after every link update, force halo and reduction.  WORLD = 1: the unsplit lattice."""
import faulthandler
import os
import sys

import numpy as np

faulthandler.enable()
faulthandler.dump_traceback_later(int(os.environ.get("MP_WORKER_TIMEOUT", "240")), exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_gpu_md_trajectory import EO, CloverDetTrajectory  # noqa: E402
from tmlqcd_amd import Lattice  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402

rank, world, job, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
Tg, L = 8, 8
T = Tg // world


class SplitTrajectory(CloverDetTrajectory):
    def __init__(self, kappa=0.125, mu=0.25, c_sw=1.2, seed=7):
        self.c_sw, self.kappa, self.mu = c_sw, kappa, mu
        self.lat = lat = Lattice(T, L, L, L, kappa=kappa, mu=mu, nproc_t=world, proc_t=rank, device=0)
        if world > 1:
            lat.comm_init_shm(job)
            if os.environ.get("MP_FACES") == "direct":
                lat.comm_init_ipc()
        self.g0 = syn.gauge_field(seed, T, L, L, L, world, rank)
        XYZ = L ** 3
        self.p0 = np.concatenate([np.random.default_rng([seed + 1, rank * T + t]).standard_normal((XYZ, 4, 8)) for t in range(T)])
        self.R = lat.field(syn.spinor_field_eo(seed + 2, 1, T, L, L, L, world, rank))
        self.phi, self.X, self.Y, self.w2, self.w3 = (lat.field() for _ in range(5))
        self.reset()
        self.clover()
        lat.op("Qsw_plus_psi", self.phi, self.R)
        self.iters = 0


tr = SplitTrajectory()
tr.leapfrog(4, 0.05)
tr.solve()
res = {"links": tr.lat.gauge_download()[:tr.lat.V], "momenta": tr.lat.momenta_download(),
       "action": np.array([tr.lat.square_norm(tr.Y, tr.lat.Vh, 1)]), "iters": np.array([tr.iters])}
# and back: momenta flipped, the links return to where they started
tr.lat.momenta_upload(-res["momenta"])
tr.leapfrog(4, 0.05)
res["back"] = tr.lat.gauge_download()[:tr.lat.V] - tr.g0[:tr.lat.V]
np.savez(os.path.join(outdir, "md_%d_of_%d.npz" % (rank, world)), **res)
tr.close()
print("rank %d of %d done" % (rank, world), flush=True)
