"""CPU, world_size 2 over gloo: the T-split decomposition (PARALLELT, mpi_init.c:240-242) with the
halo exchange of xchange_field (xchange/xchange_field.c:308-309,347-348) restated on torch.distributed.
Each rank runs the oracle on its slab; the union must equal the single-rank result on the global
lattice.  This pins the conventions the GPU path shares: parity from global coordinates, halo slot
layout, which face goes to which neighbour, and the per-rank synthetic gauge slabs of bench.py."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def xchange_field(dist, rank, world, field, Vh, face):
    """Fill the halo of a one-parity field [Vh + 2*face]: t-up face at Vh, t-down face at Vh + face."""
    import torch
    up, dn = (rank + 1) % world, (rank - 1) % world
    send_dn = torch.from_numpy(field[0:face].copy())              # our t = 0 slice   -> down neighbour's t-up halo
    send_up = torch.from_numpy(field[Vh - face:Vh].copy())        # our t = T-1 slice -> up neighbour's t-down halo
    recv_up, recv_dn = torch.empty_like(send_dn), torch.empty_like(send_up)
    reqs = [dist.isend(send_dn, dn, tag=81), dist.irecv(recv_up, up, tag=81),
            dist.isend(send_up, up, tag=82), dist.irecv(recv_dn, dn, tag=82)]
    for r in reqs:
        r.wait()
    field[Vh:Vh + face] = recv_up.numpy()
    field[Vh + face:Vh + 2 * face] = recv_dn.numpy()


def _worker(rank, world, port, T, L, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import synthetic as syn
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    kappa, mu, theta = 0.13, 0.02, (1.0, 0.0, 0.0, 0.5)
    o = Oracle(T, L, L, L, kappa=kappa, mu=mu, theta=theta, nproc_t=world, proc_t=rank)
    o.set_gauge(syn.gauge_field(3, T, L, L, L, world, rank))
    face = L * L * L // 2
    res = {}
    for ieo in (0, 1):
        k = o.new_field()
        k[:o.Vh] = syn.spinor_field_eo(4, 1 - ieo, T, L, L, L, world, rank)
        xchange_field(dist, rank, world, k, o.Vh, face)
        l = o.new_field()
        o.Hopping_Matrix(ieo, l, k)
        res["H%d" % ieo] = l[:o.Vh].copy()
    # reductions: partial sums + all-reduce == MPI_Allreduce of square_norm.c:314
    import torch
    part = torch.tensor([o.square_norm(res["H0"], o.Vh)], dtype=torch.float64)
    dist.all_reduce(part)
    res["norm_H0_global"] = np.array([float(part[0])])
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("T,L", [(4, 4), (2, 4)])
def test_t_split_two_ranks_equals_single_rank(tmp_path, T, L):
    # stdlib spawn, not torch.multiprocessing: this process has libtmlqcd_hip.so (system ROCm runtime) loaded
    # and must not also import torch (which bundles its own HIP runtime); the workers import torch.
    import multiprocessing as mp
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import synthetic as syn
    world = 2
    port = 29600 + (os.getpid() % 200) + T
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, T, L, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, "rank exited with %r" % p.exitcode
    Tg = T * world
    g = Oracle(Tg, L, L, L, kappa=0.13, mu=0.02, theta=(1.0, 0.0, 0.0, 0.5))
    g.set_gauge(syn.gauge_field(3, Tg, L, L, L))
    Vh_loc = T * L ** 3 // 2
    for ieo in (0, 1):
        k = g.new_field()
        k[:g.Vh] = syn.spinor_field_eo(4, 1 - ieo, Tg, L, L, L)
        ref = g.new_field()
        g.Hopping_Matrix(ieo, ref, k)
        for rank in range(world):
            got = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))["H%d" % ieo]
            assert np.array_equal(got, ref[rank * Vh_loc:(rank + 1) * Vh_loc]), (ieo, rank)
        if ieo == 0:
            n = np.load(os.path.join(str(tmp_path), "rank0.npz"))["norm_H0_global"][0]
            assert abs(n - g.square_norm(ref, g.Vh)) <= 1e-13 * n


def test_geometry_tables_self_consistent():
    """What test/check_geometry.c:92-193 checks at start-up: iup/idn inverse of each other, e/o maps bijective."""
    from oracle.oraclebind import Oracle
    for dims, np_t, pt in (((4, 4, 4, 4), 1, 0), ((6, 4, 2, 8), 1, 0), ((4, 4, 4, 4), 2, 1)):
        o = Oracle(*dims, nproc_t=np_t, proc_t=pt)
        V, VR = o.V, o.VPR
        iup, idn = o.iup()[:V], o.idn()[:V]
        for mu in range(4):
            inner = iup[:, mu] < V
            assert np.array_equal(idn[iup[inner, mu], mu], np.arange(V)[inner])
            inner = idn[:, mu] < V
            assert np.array_equal(iup[idn[inner, mu], mu], np.arange(V)[inner])
        e2l = o.eo2lexic()
        assert sorted(e2l.tolist()) == list(range(VR))
        l2s = o.lexic2eosub()
        assert np.array_equal(l2s[e2l[:VR // 2]], np.arange(VR // 2))
        hi = o.hi()
        # odd entries address the OTHER parity's sub-index space and stay inside field + halo
        assert hi[:V // 2, 1::2].max() < VR // 2 and hi[:V // 2, 1::2].min() >= 0


def _ildg_worker(rank, world, port, T, L, prec, path, out_dir):
    """One rank of a T-split read of an ILDG file: seeks to ITS part of the ildg-binary-data record (the offset the reference computes
    in gauge_read_binary.c:158-163 for g_nproc_x = g_nproc_y = g_nproc_z = 1), unpacks it, and the ranks combine their checksum words
    with a bit-wise XOR (DML_checksum_combine, io/dml.c:63-66: MPI_Allreduce with MPI_BXOR)."""
    sys.path.insert(0, ROOT)
    import struct
    import torch
    import torch.distributed as dist
    from oracle import ildgbind as ib
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    raw = open(path, "rb").read()
    pos = 0
    while True:                                                        # walk the LIME records to the binary one
        n = struct.unpack(">Q", raw[pos + 8:pos + 16])[0]
        if raw[pos + 16:pos + 144].split(b"\0")[0] == b"ildg-binary-data":
            break
        pos += 144 + (n + 7) // 8 * 8
    sb = 576 if prec == 64 else 288
    Vloc = T * L ** 3
    assert n == world * Vloc * sb
    mine = np.frombuffer(raw, dtype=np.uint8, count=Vloc * sb, offset=pos + 144 + rank * Vloc * sb)
    gf, sums = ib.unpack(mine, prec, T, L, L, L, rank0=rank * Vloc)
    words = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(words, torch.tensor(sums, dtype=torch.int64))
    a = b = 0
    for w in words:
        a ^= int(w[0]); b ^= int(w[1])
    np.savez(os.path.join(out_dir, "ildg_rank%d.npz" % rank), gf=gf, sums=np.array([a, b], dtype=np.int64))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("prec", [64, 32])
def test_t_split_ranks_read_their_part_of_an_ildg_record(tmp_path, prec):
    import multiprocessing as mp
    from oracle import ildgbind as ib
    from tmlqcd_amd import synthetic as syn
    world, T, L = 2, 2, 4
    Tg = T * world
    g = syn.gauge_field(31, Tg, L, L, L)
    path = str(tmp_path / "conf.lime")
    rc, sums = ib.write_gauge_field(path, g, prec, Tg, L, L, L)
    assert rc == 0
    port = 29850 + (os.getpid() % 100) + prec
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_ildg_worker, args=(r, world, port, T, L, prec, path, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, "rank exited with %r" % p.exitcode
    want = g if prec == 64 else g.astype(np.float32).astype(np.float64)
    V = T * L ** 3
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "ildg_rank%d.npz" % r))
        assert np.array_equal(d["gf"], want[r * V:(r + 1) * V])
        assert (int(d["sums"][0]), int(d["sums"][1])) == sums          # every rank ends up with the file's checksum
