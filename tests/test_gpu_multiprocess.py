"""GPU: the multi-rank code as REAL PROCESSES.  The one-GPU test box cannot hold an RCCL communicator of more than one rank (RCCL
refuses two ranks on one device), so the ranks talk through the library's host-staged shared-memory transport
(tmhip_comm_init_shm, xfer_shm.hip: device -> page-locked host memory -> a POSIX shared-memory segment -> device) and share the GPU.
Everything above the transport is the code an N-GPU run executes: per-rank geometry (global parity, slab offsets), the split-phase
stencil with its exterior kernel, reductions over the ranks (square_norm, the fused CG's alpha and stopping test), `bench.py`'s
rank flow (rendezvous, per-step agreement, rank check against the unsplit lattice on rank 0, timing legs) -- with real skew between
the processes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(nranks, extra=(), rc_ok=True, **more_env):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TMLQCD_BENCH_TRANSPORT="shm", HSA_ENABLE_IPC_MODE_LEGACY="0", TMLQCD_HIP_FLAG_TIMEOUT_S="60")
    env.update(more_env)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--L", "16", "--steps", "20", "--warmup", "2", "--cg-iters", "20",
           "--no-cpu", "--no-rows"] + list(extra)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert (r.returncode == 0) == (rc_ok is True), r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0]), r.stderr


@pytest.mark.parametrize("nranks", [2, 3, 4])
def test_bench_runs_as_n_processes_and_agrees_with_the_unsplit_lattice(nranks):
    rec, err = _bench(nranks)
    chk = rec["rank_check"]
    assert chk["ok"] is True, chk
    assert chk["hopping_matrix_max_rel_dev"] <= 1e-13 and chk["qtm_pm_psi_max_rel_dev"] <= 1e-13 and chk["global_norm_rel_dev"] <= 1e-13
    assert abs(chk["cg_iters_split"] - chk["cg_iters_unsplit"]) <= 1 and chk["cg_solution_max_rel_dev"] <= 1e-8
    # an honest line: the ranks share ONE GPU -- n_gpus counts devices, n_ranks processes, and the line is marked as a rehearsal
    assert rec["n_ranks"] == nranks and rec["n_gpus"] == 1 and rec["rehearsal"] is True and rec["value"] and rec["value"] > 0
    assert rec["transport"] == "shm" and rec["ring_nranks"] == [nranks, nranks] and rec["rccl_nranks"] is None      # RCCL built nothing here
    assert set(rec["wall_s"]) >= {"configs[3]", "headline", "total"} and rec["wall_s"]["total"] < 600
    # default TMLQCD_BENCH_FACES=auto: after the communicator's legs the direct carrier (IPC-mapped neighbour buffers, faces stored by
    # the producing waves) runs configs[3] against the unsplit lattice and the headline loop against the communicator's output
    fd = rec["faces_direct"]
    assert fd["ok"] is True, fd
    assert fd["strong"]["rank_check"]["ok"] is True and fd["strong"]["faces"] == "direct" and fd["strong"]["ranks_sharing_a_gpu"] == nranks
    assert fd["headline"]["max_rel_dev_vs_communicator"] <= 1e-13 and fd["headline"]["faces"] == "direct"
    st = rec["strong"]
    assert st.get("ok", True) and st["value"] > 0 and st["cg_iters_per_s"] > 0 and st["nocom"]["value"] > 0
    if 16 % nranks == 0 and (16 // nranks) % 2 == 0:
        s32 = rec["strong_32"]
        assert s32["rank_check"]["ok"] is True, s32
    assert rec["cg"]["iters_per_s"] > 0 and abs(rec["hermiticity_rel_dev"]) < 1e-12


@pytest.mark.parametrize("who", ["abort_direct", "abort_direct_peer"])
def test_bench_line_survives_a_rank_dying_in_the_direct_legs(who):
    """The second-carrier phase of `bench.py --gpus N` must not be able to cost the line measured over the communicator: rank 0
    (or rank 1, whereupon the launcher ends rank 0) aborts at the head of the direct legs, as a GPU fault would end it -- the run
    fails, and its ONE line is the communicator's, complete, with `faces_direct.ok: false` (rank 0's guardian process prints it)."""
    rec, err = _bench(2, rc_ok=False, TMLQCD_BENCH_TEST_FAULT=who)
    assert rec["value"] and rec["value"] > 0 and rec["n_ranks"] == 2 and rec["rank_check"]["ok"] is True
    assert rec["faces"] != "direct" and rec["faces_direct"]["ok"] is False and "rank 0 ended" in rec["faces_direct"]["error"]
    assert rec["cg"]["iters_per_s"] > 0 and rec["strong"]["value"] > 0 and "total" in rec["wall_s"]


def test_bench_with_the_direct_carrier_from_the_start():
    """TMLQCD_BENCH_TRANSPORT=ipc: every leg -- rank checks against the unsplit lattice, configs[3], 16^4 / N, the headline with its
    reductions and solves -- with the faces stored straight into the neighbours' IPC-mapped buffers, three processes on one GPU."""
    rec, err = _bench(3, TMLQCD_BENCH_TRANSPORT="ipc")
    assert rec["rank_check"]["ok"] is True and rec["rank_check"]["faces"] == "direct", rec["rank_check"]
    assert rec["faces"] == "direct" and rec["sums"] == "direct" and rec["transport"] == "shm" and rec["ranks_sharing_a_gpu"] == 3 and "faces_direct" not in rec
    assert rec["strong"]["value"] > 0 and rec["cg"]["iters_per_s"] > 0 and abs(rec["hermiticity_rel_dev"]) < 1e-12
    assert "gave up" not in err


@pytest.mark.parametrize("world,faces", [(2, "ring"), (3, "ring"), (3, "direct")])
def test_next_rows_as_real_processes(world, faces, tmp_path):
    """D_psi, the device-side clover term / inverse and Qsw_pm_psi, cg_her and mixed_cg_her on both operators, both parts of the fermion
    force, the link update with its halo exchange and update_momenta -- every rank a process of its own (tests/mp_rank_worker.py), slab by
    slab against the unsplit lattice."""
    import numpy as np
    worker = os.path.join(ROOT, "tests", "mp_rank_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", TMLQCD_HIP_FLAG_TIMEOUT_S="60", MP_FACES=faces)
    job = "mp_%d_%d_%s" % (os.getpid(), world, faces)
    ref = subprocess.run([sys.executable, worker, "0", "1", job, str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert ref.returncode == 0, ref.stderr[-3000:]
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), job, str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=400) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    one = np.load(os.path.join(str(tmp_path), "rank0_of_1.npz"))
    parts = [np.load(os.path.join(str(tmp_path), "rank%d_of_%d.npz" % (r, world))) for r in range(world)]

    def slab(a, r):
        n = a.shape[0] // world
        return a[r * n:(r + 1) * n]
    # the configuration file written by `world` ranks is the file one rank writes, byte for byte (the date in the xlf record aside)
    fa, fb = (open(os.path.join(str(tmp_path), "conf_%d.lime" % w), "rb").read() for w in (1, world))
    assert len(fa) == len(fb) and sum(x != y for x, y in zip(fa, fb)) <= 32, "the split writers' file differs from the single writer's"
    for r in range(world):
        assert parts[r]["ildg"][0] == 0 and np.array_equal(parts[r]["ildg"][1:], one["ildg"][1:]), (r, parts[r]["ildg"], one["ildg"])   # read status, checksum words
    for key in ("D_psi", "Qsw_pm_psi", "Qtm_pm_psi", "derivative", "links", "links_read_back", "hop_after_update", "hop_after_read", "momenta"):
        sc = np.abs(one[key]).max()
        for r in range(world):
            dev = np.abs(parts[r][key] - slab(one[key], r)).max() / sc
            assert dev < 1e-12, (key, r, dev)
    for r in range(world):                                       # global sums: the same number on every rank
        assert np.allclose(parts[r]["norm"], one["norm"], rtol=1e-13), (r, parts[r]["norm"], one["norm"])
        assert np.array_equal(parts[r]["norm"], parts[0]["norm"])   # ... bit for bit (added in rank order on every rank)
    for name in ("Qtm_pm_psi", "Qsw_pm_psi"):
        it0 = int(one["cg_it_" + name][0])
        for r in range(world):
            assert abs(int(parts[r]["cg_it_" + name][0]) - it0) <= 1, name
            assert abs(int(parts[r]["mixed_it_" + name][0]) - int(one["mixed_it_" + name][0])) <= 3 + 0.03 * it0, name
            for kind in ("cg_", "mixed_"):
                sc = np.abs(one[kind + name]).max()
                assert np.abs(parts[r][kind + name] - slab(one[kind + name], r)).max() / sc < 1e-6, (kind, name, r)


@pytest.mark.parametrize("world,Tg", [(2, 16), (4, 16), (5, 20)])
def test_random_sequences_between_real_processes(world, Tg, tmp_path):
    """The seeded random operation sequence of test_gpu_split_stress.py (stencils with every epilogue, chains, linalg between them,
    short cg_her solves, uploads, the benchmark loop) on a Tg x 16^3 lattice cut into `world` slabs, every rank a process that falls
    behind at random points -- for every form of the split path, slab by slab against the unsplit lattice.  "direct: ...": the faces
    are stored by the producing waves into the NEIGHBOUR PROCESS's receive buffers (hipIpc mappings; the processes share the GPU), in
    the one-kernel form with the boundary waves waiting for the neighbour's word, and in the two-kernel form.  Five processes (with the
    test runner itself the most the test box admits on its GPU: six; T_local 4): the direct forms only."""
    import re
    import numpy as np
    from tests.test_gpu_split_stress import DIRECT_FORMS, FORMS
    worker = os.path.join(ROOT, "tests", "mp_stress_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", TMLQCD_HIP_FLAG_TIMEOUT_S="60", MP_TG=str(Tg))
    seed, nops = 5, 60
    ref = subprocess.run([sys.executable, worker, "0", "1", "none", str(tmp_path), str(seed), str(nops), "flags"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert ref.returncode == 0, ref.stderr[-3000:]
    one = np.load(os.path.join(str(tmp_path), "stress_flags_0_of_1.npz"))
    forms = ([f for f, _ in FORMS] if world < 5 else []) + ["direct: " + f for f, _ in DIRECT_FORMS]
    for form in forms:
        tag = re.sub(r"[^A-Za-z0-9]+", "_", form)
        job = "st_%d_%d_%s" % (os.getpid(), world, tag)
        procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), job, str(tmp_path), str(seed), str(nops), form], env=env,
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
        outs = [p.communicate(timeout=400) for p in procs]
        for p, (so, se) in zip(procs, outs):
            assert p.returncode == 0, (form, se[-3000:])
            assert "gave up" not in se, (form, se[-2000:])             # no bounded wait ran into its deadline
        for r in range(world):
            part = np.load(os.path.join(str(tmp_path), "stress_%s_%d_of_%d.npz" % (tag, r, world)))
            assert len(part["scal"]) == len(one["scal"]) and np.allclose(part["scal"], one["scal"], rtol=1e-11, atol=1e-11), (form, r)
            for i in range(5):
                full = one["f%d" % i]
                n = full.shape[0] // world
                dev = np.abs(part["f%d" % i] - full[r * n:(r + 1) * n]).max() / np.abs(full).max()
                assert dev < 1e-11, (form, r, i, dev)


@pytest.mark.parametrize("world,faces", [(2, "ring"), (4, "ring"), (4, "direct")])
def test_drop_in_symbols_on_t_split_ranks(world, faces, tmp_path):
    """The reference-named symbols on a T-split lattice, every rank a host process with tmLQCD's own globals (g_nproc_t,
    g_proc_coords, RAND halo slices of g_gauge_field): Hopping_Matrix, Qtm_pm_psi, square_norm / scalar_prod_r with parallel = 1
    (and 0: the local sum), cg_her -- in coherent and in lazy residency -- slab by slab against the unsplit host program."""
    import numpy as np
    worker = os.path.join(ROOT, "tests", "mp_dropin_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", TMLQCD_HIP_FLAG_TIMEOUT_S="60", MP_FACES=faces)
    ref = subprocess.run([sys.executable, worker, "0", "1", "none", str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert ref.returncode == 0, ref.stderr[-3000:]
    job = "di_%d_%d" % (os.getpid(), world)
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), job, str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=400) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    one = np.load(os.path.join(str(tmp_path), "dropin_0_of_1.npz"))
    parts = [np.load(os.path.join(str(tmp_path), "dropin_%d_of_%d.npz" % (r, world))) for r in range(world)]
    for tag in ("coherent", "lazy"):
        for key in ("_hop", "_qtm", "_cg"):
            full = one[tag + key]
            n = full.shape[0] // world
            for r in range(world):
                tol = 1e-8 if key == "_cg" else 1e-13
                assert np.abs(parts[r][tag + key] - full[r * n:(r + 1) * n]).max() / np.abs(full).max() < tol, (tag, key, r)
        local = 0.0
        for r in range(world):
            assert np.allclose(parts[r][tag + "_sums"][:2], one[tag + "_sums"][:2], rtol=1e-13), (tag, r)      # global sums on every rank
            assert abs(int(parts[r][tag + "_it"][0]) - int(one[tag + "_it"][0])) <= 1
            local += parts[r][tag + "_sums"][2]
        assert abs(local - one[tag + "_sums"][2]) <= 1e-13 * one[tag + "_sums"][2]                             # parallel = 0: each rank its own part
        assert np.array_equal(parts[0]["lazy_hop"], parts[0]["coherent_hop"])


@pytest.mark.parametrize("world,faces", [(2, "ring"), (4, "ring"), (4, "direct")])
def test_md_trajectory_between_real_processes(world, faces, tmp_path):
    """A leapfrog trajectory of the clover determinant with the lattice cut into `world` processes (tests/mp_md_worker.py): after four
    steps the links and momenta of every rank are the unsplit trajectory's slab, the action agrees, and the trajectory is reversible
    -- the link halo, the stencil copy and the clover term are refreshed from the neighbours after every update_gauge."""
    import numpy as np
    worker = os.path.join(ROOT, "tests", "mp_md_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", TMLQCD_HIP_FLAG_TIMEOUT_S="60", MP_FACES=faces)
    ref = subprocess.run([sys.executable, worker, "0", "1", "none", str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400)
    assert ref.returncode == 0, ref.stderr[-3000:]
    job = "md_%d_%d" % (os.getpid(), world)
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), job, str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=500) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    one = np.load(os.path.join(str(tmp_path), "md_0_of_1.npz"))
    for r in range(world):
        part = np.load(os.path.join(str(tmp_path), "md_%d_of_%d.npz" % (r, world)))
        for key in ("links", "momenta"):
            full = one[key]
            n = full.shape[0] // world
            dev = np.abs(part[key] - full[r * n:(r + 1) * n]).max() / np.abs(full).max()
            assert dev < 1e-10, (key, r, dev)
        assert abs(part["action"][0] - one["action"][0]) <= 1e-10 * abs(one["action"][0])
        assert abs(int(part["iters"][0]) - int(one["iters"][0])) <= 12          # (a dozen solves, +-1 iteration each)
        assert np.abs(part["back"]).max() < 1e-9
    assert np.abs(one["back"]).max() < 1e-9
