"""N ranks of a T-split lattice as N processes, one GPU each (rendezvous over gloo, faces and reductions over RCCL): Hopping_Matrix,
Qtm_pm_psi, a global norm and a cg_her solve, checked slab by slab against the unsplit lattice computed on rank 0's GPU.
For a node with N >= 2 GPUs -- RCCL refuses two ranks on one device ("invalid usage"), so a one-GPU box cannot run it; there the
multi-rank code is rehearsed by the loopback modes and the two-context tests.
FACES=direct in the environment: the half-spinor faces over the direct carrier (tmhip_comm_init_ipc: stores into the neighbour GPUs' IPC-mapped
receive buffers over xGMI) instead of ncclSend / ncclRecv -- the first thing to run on a node with >= 2 GPUs after the plain check.
Usage: [FACES=direct] python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 --master-port 29544 tools/multi_rank_check.py"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
T, L = 8, 8                                   # local T; global lattice (T * world) x L^3
dev = int(os.environ.get("LOCAL_RANK", "0"))
lat = Lattice(T, L, L, L, kappa=0.125, mu=0.01, nproc_t=world, proc_t=rank, device=dev)
uid = torch.zeros(128, dtype=torch.uint8)
if rank == 0:
    uid.copy_(torch.tensor(list(lat.comm_unique_id()), dtype=torch.uint8))
dist.broadcast(uid, 0)
lat.comm_init(bytes(uid.tolist()))
if os.environ.get("FACES") == "direct":
    lat.comm_init_ipc()
print("rank %d: communicator up, faces direct: %s" % (rank, lat.comm_faces_direct()), flush=True)
lat.set_gauge(syn.gauge_field(7, T, L, L, L, world, rank))
src = syn.spinor_field_eo(8, 0, T, L, L, L, world, rank)
k, l, q = lat.field(src), lat.field(), lat.field()
lat.Hopping_Matrix(0, l, k)
lat.Qtm_pm_psi(q, k)
nrm = lat.square_norm(q, lat.Vh, 1)            # global norm through ncclAllReduce
P = lat.field()
it, hist = lat.cg_her(P, k, 200, 1e-18, 1, lat.Vh)
print("rank %d: |Qtm_pm_psi k|^2 = %.15e, cg_her %d iterations, last err %.3e" % (rank, nrm, it, hist[-1]), flush=True)
mine = [l.download(), q.download(), P.download()]
lat.close()
dist.barrier()
# the same on the unsplit global lattice (rank 0 only), compared slab by slab
gathered = [None] * world
dist.gather_object(mine, gathered if rank == 0 else None, 0)
if rank == 0:
    G = Lattice(T * world, L, L, L, kappa=0.125, mu=0.01, device=dev)
    G.set_gauge(syn.gauge_field(7, T * world, L, L, L))
    gk = G.field(syn.spinor_field_eo(8, 0, T * world, L, L, L)); gl, gq, gP = G.field(), G.field(), G.field()
    G.Hopping_Matrix(0, gl, gk); G.Qtm_pm_psi(gq, gk)
    gn = G.square_norm(gq, G.Vh, 1)
    git, _ = G.cg_her(gP, gk, 200, 1e-18, 1, G.Vh)
    ref = [gl.download(), gq.download(), gP.download()]
    Vh = T * L ** 3 // 2
    worst = 0.0
    for r in range(world):
        for a, b in zip(gathered[r], ref):
            worst = max(worst, float(np.abs(a - b[r * Vh:(r + 1) * Vh]).max() / np.abs(b).max()))
    print("unsplit: norm %.15e, cg_her %d iterations; worst slab deviation %.2e" % (gn, git, worst), flush=True)
    assert abs(gn - nrm) <= 1e-12 * gn and abs(git - it) <= 1 and worst < 1e-9, "multi-rank result differs from the unsplit lattice"
    print("MULTI-RANK PATH OK (%d ranks)" % world, flush=True)
    G.close()
dist.destroy_process_group()
