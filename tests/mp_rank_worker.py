"""One rank of tests/test_gpu_multiprocess.py::test_next_rows_as_real_processes: `python mp_rank_worker.py RANK WORLD JOB OUTDIR`.
Every rank owns a T-slab of a global Tg x L^3 lattice, meets the others through the host-staged shared-memory transport
(tmhip_comm_init_shm) and runs the rows next to the stencil -- D_psi, device-side clover term + inverse and Qsw_pm_psi, the hopping
and clover parts of the fermion force, the molecular-dynamics link update with its halo exchange, cg_her / mixed_cg_her on both
operators -- leaving its slab of every result in OUTDIR/rank<r>.npz.  WORLD = 1: the unsplit lattice (the reference)."""
import os
import sys

import faulthandler

import numpy as np

faulthandler.enable()
faulthandler.dump_traceback_later(int(os.environ.get("MP_WORKER_TIMEOUT", "240")), exit=True)     # a hung rank says where, and ends

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tmlqcd_amd import Lattice  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402

rank, world, job, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
Tg, L = 12, 8
T = Tg // world
kappa, mu, csw, theta = 0.13, 0.02, 1.2, (1.0, 0.3, 0.0, -0.2)
lat = Lattice(T, L, L, L, kappa=kappa, mu=mu, theta=theta, nproc_t=world, proc_t=rank, device=0)
if world > 1:
    lat.comm_init_shm(job)
    if os.environ.get("MP_FACES") == "direct":          # faces as direct stores into the neighbours' IPC-mapped buffers; sums and the other halos stay on the ring
        lat.comm_init_ipc()
        assert lat.comm_faces_direct() == (True, world)  # (all ranks of the test share the one GPU)
g = syn.gauge_field(21, T, L, L, L, world, rank)
lat.set_gauge(g)
N, V = lat.Vh, lat.V
res = {}

# full-lattice operator (lexicographic local field; the parity of a site is global)
XYZ = L ** 3
full = np.concatenate([syn.spinor_slice(22, rank * T + t, L, L, L) for t in range(T)])
dQ, dP = lat.full_field(full), lat.full_field()
lat.D_psi(dP, dQ)
res["D_psi"] = dP.download()

# clover term and its inverse on the device from the resident links (neighbours' links across the cut from the halo slabs)
lat.sw_term(None, kappa, csw)
lat.sw_invert(0, mu)
src = syn.spinor_field_eo(23, 0, T, L, L, L, world, rank)
b = syn.spinor_field_eo(24, 1, T, L, L, L, world, rank)
k, q, l, x = lat.field(src), lat.field(b), lat.field(), lat.field()
lat.op("Qsw_pm_psi", l, k); res["Qsw_pm_psi"] = l.download()
lat.Qtm_pm_psi(l, k); res["Qtm_pm_psi"] = l.download()
res["norm"] = np.array([lat.square_norm(l, N, 1), lat.scalar_prod_r(l, k, N, 1)])

# solvers: iteration counts must agree (+-1 / a few for the mixed ones), the solutions are checked through their slabs
for name in ("Qtm_pm_psi", "Qsw_pm_psi"):
    x.zero(); it, _ = lat.cg_her(x, q, 3000, 1e-18, 1, N, op=name)
    res["cg_" + name] = x.download(); res["cg_it_" + name] = np.array([it])
    x.zero(); it, _ = lat.mixed_cg_her(x, q, 3000, 1e-18, 1, N, op=name)
    res["mixed_" + name] = x.download(); res["mixed_it_" + name] = np.array([it])

# fermion force of the clover determinant: hopping part (t = 0 slices of both fields from the up neighbour) + clover part
# (insertion matrices of the neighbours' boundary slices)
lat.derivative_zero(); lat.swpm_zero()
lat.deriv_Sb(1, q, k, 0.7); lat.deriv_Sb(0, k, q, -0.4)
lat.sw_spinor_eo(0, k, l, 0.3); lat.sw_spinor_eo(1, q, x, 0.3)
lat.sw_deriv(0, mu)                                         # the tr-log term, even sites (sw_inv holds the even sites)
lat.sw_all(kappa, csw)
res["derivative"] = lat.derivative()

# molecular dynamics: momenta -> links (exp of the momenta times the links, halo slabs refreshed from the neighbours), then a stencil
rng = np.random.default_rng([25, rank])
mom = np.concatenate([np.random.default_rng([25, rank * T + t]).standard_normal((XYZ, 4, 8)) for t in range(T)])
lat.momenta_upload(mom)
lat.update_gauge(0.05)
res["links"] = lat.gauge_download()[:V]
lat.Hopping_Matrix(0, l, k); res["hop_after_update"] = l.download()
# ILDG: every rank writes its part of ONE record at its offset (checksum words gathered over the ranks), then all read the file back
conf = os.path.join(outdir, "conf_%d.lime" % world)
sums = lat.write_gauge_field(conf, 64, "plaquette = 0.5")
rc, back, info = lat.read_gauge_field(conf, 64)
res["ildg"] = np.array([rc, sums[0], sums[1], info.suma, info.sumb, info.suma_stored, info.sumb_stored], dtype=np.float64)
res["links_read_back"] = back[:V]
lat.Hopping_Matrix(1, l, k); res["hop_after_read"] = l.download()
lat.update_momenta(0.1)
res["momenta"] = lat.momenta_download()
lat.sync()
np.savez(os.path.join(outdir, "rank%d_of_%d.npz" % (rank, world)), **res)
lat.close()
print("rank %d of %d done" % (rank, world), flush=True)
