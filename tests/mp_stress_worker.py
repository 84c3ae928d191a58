"""One rank of tests/test_gpu_multiprocess.py::test_random_sequences_between_real_processes:
`python mp_stress_worker.py RANK WORLD JOB OUTDIR SEED NOPS FORM`.  The seeded random operation sequence of tests/test_gpu_split_stress.py
on this rank's slab of a global lattice, the ranks meeting through the shared-memory transport -- with this rank falling behind at random
points (host-side sleeps, different on every rank: real skew between the processes).  WORLD = 1: the unsplit lattice."""
import os
import re
import sys
import time

import faulthandler

import numpy as np

faulthandler.enable()
faulthandler.dump_traceback_later(int(os.environ.get("MP_WORKER_TIMEOUT", "240")), exit=True)     # a hung rank says where, and ends

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_gpu_split_stress import DIRECT_FORMS, FORMS, _run  # noqa: E402
from tests.util import random_gauge  # noqa: E402, F401
from tmlqcd_amd import Lattice  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402

rank, world, job, outdir, seed, nops, form = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
Tg, L = int(os.environ.get("MP_TG", "16")), 16
T = Tg // world
lat = Lattice(T, L, L, L, kappa=0.131, mu=0.017, theta=(1.0, 0.2, 0.0, -0.3), nproc_t=world, proc_t=rank, device=0)
direct = form.startswith("direct: ")                     # "direct: <name of a form of DIRECT_FORMS>": the faces over the direct carrier
for k, v in (dict(DIRECT_FORMS)[form[len("direct: "):]] if direct else dict(FORMS)[form]).items():
    lat.set_option(k, v)
if world > 1:
    lat.comm_init_shm(job)
    if direct:
        lat.comm_init_ipc()
        assert lat.comm_faces_direct() == (True, world)
lat.set_gauge(syn.gauge_field(31, T, L, L, L, world, rank))
lag = np.random.default_rng([seed, 77, rank])


def skew(step):
    if world > 1 and lag.random() < 0.12:
        lat.sync() if lag.random() < 0.5 else None          # sometimes with the GPU idle, sometimes with work in flight
        time.sleep(float(lag.uniform(0.0, 0.03)))


out, scal = _run(lat, seed, nops, split=False, gen=lambda tag: syn.spinor_field_eo(tag, 0, T, L, L, L, world, rank), skew=skew)
np.savez(os.path.join(outdir, "stress_%s_%d_of_%d.npz" % (re.sub(r"[^A-Za-z0-9]+", "_", form), rank, world)), scal=np.array(scal), **{"f%d" % i: x for i, x in enumerate(out)})
lat.close()
print("rank %d of %d done" % (rank, world), flush=True)
