#!/usr/bin/env python3
"""bench.py -- benchmark.c's Hopping_Matrix loop (+ cg_her iterations/s) on N MI355X.

A "step" is one iteration of the reference's timed loop (benchmark.c:291-300):
    Hopping_Matrix(0, f1, f0); Hopping_Matrix(1, f2, f1)      (= VOLUME output sites per GPU)
on fields resident in HBM.  `value` follows benchmark.c:318,327: Mflop/s = nranks * 1608 / (us per
site-update).  N=1 workload: 32^4 fp64 (BASELINE.json configs[2]); N>1: weak scaling, every rank
holds a 32^4 slab of a 32^3 x (32 N) lattice split in T, half-spinor faces exchanged over RCCL
and overlapped with the stencil; the same run also measures BASELINE configs[3] (32^3 x 64 split
N ways, the `strong` object) and north_star's literal "32^4 at 1/2/4/8 GPUs" (32^4 split N ways,
`strong_32`), each after checking it slab by slab against the unsplit lattice computed on rank 0
BEFORE the split operations start (`rank_check`).  Every leg is a sequence of steps the ranks
agree on (class Phase): if any rank fails a step, all ranks leave that leg at that step -- nobody
goes on into a halo exchange or a collective the others will not reach -- and the line is still
printed.

    python bench.py --gpus N --steps K --warmup W

Launch: one process per GPU.  Under torchrun (WORLD_SIZE set) this process IS one rank.  Started
plainly with --gpus N > 1 it becomes a parent that never touches the GPU: it starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child process and relays
rank 0's single JSON line.
"""
import argparse
import datetime
import json
import os
import socket
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
METRIC = "Hopping_Matrix Mflop/s per site (benchmark.c) + CG iters/sec, 32^4 fp64"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="number of ranks = GPUs (default: WORLD_SIZE, else 1)")
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--L", type=int, default=32, help="spatial extent")
    ap.add_argument("--T", type=int, default=0, help="local time extent (default = L)")
    ap.add_argument("--strong", type=int, default=0, metavar="TGLOBAL",
                    help="make the HEADLINE a strong-scaling run: fixed global lattice TGLOBAL x L^3 split in T over the ranks "
                         "(default headline: weak scaling with L^4 per GPU; configs[3] is always reported in the `strong` object when N > 1)")
    ap.add_argument("--cg-iters", type=int, default=200, help="cg_her iterations timed for the CG part of the metric")
    ap.add_argument("--cpu-seconds", type=float, default=24.0, help="wall budget of the timed loops of the CPU-baseline leg (all thread counts together)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = sweep {16, 64, all physical cores}; n = that thread count only")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-rows", action="store_true", help="skip the informational next-row legs (MD step, ILDG record)")
    ap.add_argument("--no-rank-check", action="store_true", help="N > 1: skip the multi-rank parity check and the configs[3] leg")
    ap.add_argument("--rehearse-split", action="store_true",
                    help="one GPU, with --loopback: run the multi-rank legs (rank check against the unsplit lattice, configs[3], 32^4 / N) as a world of one rank "
                         "exchanging with itself -- the code path the N > 1 run takes, minus the rendezvous")
    ap.add_argument("--opt", action="append", default=[], help="library option name=value (A/B runs)")
    ap.add_argument("--loopback", type=int, default=0,
                    help="1-GPU rehearsal of the multi-GPU path: 1 = faces exchanged with self by a D2D copy, 2 = through a one-rank RCCL communicator")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------- launcher (parent, never touches the GPU)
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def parent_launch(n, args):
    """--gpus N without a torchrun environment: start N ranks as a CHILD process tree (never an exec from a process that has
    touched the GPU; this one has not even imported torch) and relay rank 0's JSON line.  Whatever happens to the child tree --
    it dies, it hangs past TMLQCD_BENCH_TIMEOUT_S (default 1500 s), it exits without a line -- ONE well-formed line leaves this
    process: `value` null and the reason in `error`."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("[bench] starting %d ranks: %s\n" % (n, " ".join(cmd)))
    limit = float(os.environ.get("TMLQCD_BENCH_TIMEOUT_S", "1500"))
    t0 = time.perf_counter()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, start_new_session=True)   # a session of its own: the whole tree can be ended
    why = None
    try:
        stdout, _ = p.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        why = "the %d-rank run did not finish within %.0f s (TMLQCD_BENCH_TIMEOUT_S)" % (n, limit)
        try:
            os.killpg(p.pid, 15)
            stdout, _ = p.communicate(timeout=20)
        except Exception:   # noqa: BLE001
            try:
                os.killpg(p.pid, 9)
            except Exception:   # noqa: BLE001
                pass
            stdout, _ = p.communicate()
    line = None
    for ln in (stdout or "").splitlines():
        if ln.startswith("{") and ("\"metric\"" in ln or "\"rendezvous\"" in ln or "\"selftest\"" in ln):
            line = ln
        else:
            sys.stderr.write(ln + "\n")
    if line is None:
        why = why or "the %d-rank run produced no result line (exit code %s)" % (n, p.returncode)
        sys.stderr.write("[bench] %s\n" % why)
        print(json.dumps({"metric": METRIC, "value": None, "unit": "Mflop/s", "n_gpus": n, "n_ranks": n, "steps": args.steps, "warmup": args.warmup,
                          "higher_is_better": True, "error": why, "wall_s": {"total": time.perf_counter() - t0}}), flush=True)
        return p.returncode or 1
    print(line, flush=True)
    return p.returncode


def rendezvous_only(world, rank):
    """TMLQCD_BENCH_RENDEZVOUS_ONLY=1 (the CPU test of the launcher): the ranks meet over gloo, agree on who is there, and rank 0
    prints one line -- nothing GPU-side is imported."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    # launcher tests: TMLQCD_BENCH_TEST_FAULT="die" -> rank 1 dies before the rendezvous, "hang" -> every rank sleeps past the parent's limit
    fault = os.environ.get("TMLQCD_BENCH_TEST_FAULT", "")
    if fault == "die" and rank == 1:
        os._exit(7)
    if fault == "hang":
        time.sleep(600)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=float(os.environ.get("TMLQCD_BENCH_PG_TIMEOUT_S", "180"))))
    t = torch.zeros(world, dtype=torch.int64)
    t[rank] = os.getpid()
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"rendezvous": "ok", "n_gpus": world, "pids": t.tolist(), "backend": "gloo"}), flush=True)
    dist.destroy_process_group()
    return 0


def agree_selftest(world, rank):
    """TMLQCD_BENCH_AGREE_SELFTEST=1 (CPU test, gloo): the phase / step / agree machinery of the multi-rank run without a GPU.
    The "rank check" phase has the real one's shape (local set-up, the reference on rank 0, the split operations, a gather, the
    comparison); TMLQCD_BENCH_INJECT_FAIL makes one rank fail in one step.  Whatever happens there, every rank leaves the phase
    at the same step, all of them run the headline phase, and rank 0 prints ONE well-formed line."""
    import numpy as np
    R = Ranks(world, rank, rank, True, backend="gloo")
    extra, trace = {}, []
    ph = Phase(R, "rank_check")
    try:
        ph.step("setup", lambda: trace.append("setup"))
        ph.step("reference", lambda: trace.append("reference") if rank == 0 else None)
        ph.step("split_ops", lambda: trace.append("split_ops"))
        got = ph.collective(lambda: R.gather0(np.full(4, float(rank))))
        ph.step("compare", lambda: trace.append("compare:%s" % (None if got is None else [float(g[0]) for g in got])))
        ph.step("timing", lambda: trace.append("timing"))
        extra["rank_check"] = {"ok": True}
    except PhaseAbort:
        extra["rank_check"] = dict(ph.error, ok=False)
    hl = Phase(R, "headline")
    hl.step("timing", lambda: trace.append("headline"))
    mx = R.allmax(float(len(trace)))
    traces = R.gather0(np.frombuffer(("|".join(trace)).ljust(256).encode(), dtype=np.uint8).copy())
    if rank == 0:
        print(json.dumps(dict(extra, selftest="agree", n_gpus=world, max_steps_run=mx,
                              traces=[bytes(t.tolist()).decode().strip() for t in traces])), flush=True)
    R.close()
    return 0


# ----------------------------------------------------------------------------------------------- CPU baseline (rank 0, N = 1)
def host_cpu_info():
    """Physical cores this process may run on, CPU model, and the cgroup CPU quota (a quota below the core count throttles
    an all-core run: the sweep shows it)."""
    aff = os.sched_getaffinity(0)
    cores, model = set(), None
    try:
        cur = {}
        for ln in open("/proc/cpuinfo"):
            if ":" in ln:
                k, v = [x.strip() for x in ln.split(":", 1)]
                cur[k] = v
            elif cur:
                if int(cur.get("processor", -1)) in aff:
                    cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                model = model or cur.get("model name")
                cur = {}
        if cur and int(cur.get("processor", -1)) in aff:
            cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
    except Exception:
        pass
    quota = None
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            quota = open(f).read().strip()
            break
        except Exception:
            continue
    return {"logical_cpus_visible": len(aff), "physical_cores": len(cores) or len(aff), "cpu_model": model, "cgroup_cpu_max": quota}


def cpu_baseline(args, T, L, gpu_out):
    """The reference CPU path (oracle/_ref: the reference's own objects, OpenMP build) timed on this box's host cores, on the
    SAME seeded arrays, after checking the GPU result against it.  One child process per thread count
    (oracle/cpu_baseline.py) with OMP_PROC_BIND=close OMP_PLACES=cores and the fields first-touched inside an OpenMP region;
    `value` is the best of the sweep, `cores` the threads it used."""
    import numpy as np
    info = host_cpu_info()
    phys = info["physical_cores"]
    counts = [args.cpu_threads] if args.cpu_threads else sorted({c for c in (16, 64) if c < phys} | {phys})
    per_leg = max(args.cpu_seconds / len(counts), 2.0)
    tmp = tempfile.NamedTemporaryFile(suffix=".npy", delete=False)
    tmp.close()
    np.save(tmp.name, gpu_out)
    sweep, parity = [], None
    try:
        for c in counts:
            env = dict(os.environ, OMP_NUM_THREADS=str(c), OMP_PROC_BIND="close", OMP_PLACES="cores")
            cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--T", str(T), "--L", str(L), "--threads", str(c),
                   "--seconds", "%.1f" % per_leg, "--gpu-out", tmp.name]
            t0 = time.perf_counter()
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True, timeout=600)
            rec = None
            for ln in r.stdout.splitlines():
                if ln.startswith("{"):
                    rec = json.loads(ln)
            if rec is None:
                sweep.append({"cores": c, "value": None, "error": (r.stderr or "")[-300:]})
                continue
            rec["leg_wall_s"] = time.perf_counter() - t0
            if rec.get("parity_max_rel_err_gpu_vs_cpu") is not None:
                parity = rec["parity_max_rel_err_gpu_vs_cpu"] if parity is None else max(parity, rec["parity_max_rel_err_gpu_vs_cpu"])
            sweep.append(rec)
    finally:
        os.unlink(tmp.name)
    good = [s for s in sweep if s.get("value")]
    if not good:
        raise RuntimeError("no CPU-baseline leg finished: %r" % sweep)
    best = max(good, key=lambda s: s["value"])
    out = {"value": best["value"], "unit": "Mflop/s", "cores": best["cores"], "kind": best["kind"],
           "sample": "%d iterations of {Hopping_Matrix(0);Hopping_Matrix(1)} on the same seeded %dx%d^3 arrays, %.1f s, %s; "
                     "OMP_PROC_BIND=close OMP_PLACES=cores, fields first-touched in an OpenMP region; best of the thread sweep"
                     % (best["iterations"], T, L, best["seconds"], best["what"]),
           "us_per_site": best["us_per_site"],
           "sweep": [{"cores": s.get("cores"), "Mflop/s": s.get("value"), "iterations": s.get("iterations"), "seconds": s.get("seconds")} for s in sweep]}
    out.update(info)
    return out, parity


# ----------------------------------------------------------------------------------------------- one rank
class Ranks:
    """torch.distributed plumbing of one rank (rendezvous, barrier, max over ranks, gather to rank 0).  backend "nccl" (= RCCL,
    tensors on the rank's GPU) for the real run, "gloo" (CPU tensors) for the CPU tests of the launcher and of `agree`."""

    def __init__(self, world, rank, local_rank, force, backend="nccl"):
        self.world, self.rank, self.local_rank = world, rank, local_rank
        self.on = world > 1 or force
        self.torch = self.dist = None
        self.dev = "cpu"
        self.device = local_rank          # GPU of this rank
        # TMLQCD_BENCH_TRANSPORT=shm: the ranks exchange through the library's host-staged shared-memory transport instead of RCCL
        # and may SHARE GPUs (rank r on GPU r mod #GPUs): the whole multi-rank run as real processes on a one-GPU box.
        # TMLQCD_BENCH_TRANSPORT=ipc: the same ring, but the half-spinor FACES travel over the direct carrier (tmhip_comm_init_ipc:
        # the producing waves store them into the neighbour's IPC-mapped buffers) from the start.
        tr = os.environ.get("TMLQCD_BENCH_TRANSPORT", "")
        self.shm = tr in ("shm", "ipc")
        self.ring = "shm" if self.shm else "rccl"             # what carries sums, force / gauge halos -- and the faces unless `faces` says direct
        # TMLQCD_BENCH_FACES: comm = faces over the ring's communicator only; direct = the direct carrier from the start (an error if it
        # cannot be set up); auto (default) = everything is measured over the communicator first, then the direct carrier is tried on the
        # same lattices under a watchdog, checked against the communicator's results, and the faster VERIFIED carrier gives `value`
        self.faces_mode = "direct" if tr == "ipc" else os.environ.get("TMLQCD_BENCH_FACES", "auto")
        if self.faces_mode not in ("auto", "comm", "direct"):
            raise SystemExit("[bench] TMLQCD_BENCH_FACES must be auto, comm or direct")
        self.transport = "the host-staged shared-memory transport (ranks may share a GPU: a rehearsal of the multi-rank run, not a scaling measurement)" if self.shm else "RCCL"
        # the ranks of one node share its CPUs: the synthetic-field generators take CPUs / ranks threads each (cgroup quota respected)
        ncpu = len(os.sched_getaffinity(0))
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            if q != "max":
                ncpu = max(1, min(ncpu, int(int(q) / int(per))))
        except Exception:   # noqa: BLE001
            pass
        os.environ.setdefault("TMLQCD_SYNTH_THREADS", str(max(1, ncpu // max(world, 1))))
        self.n_devices = 1
        if self.shm and backend == "nccl":
            backend = "gloo"
        if self.on:
            # torch first: its bundled HIP/RCCL runtime must be the one every later library binds to
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            # a rank that died before the rendezvous must not park the others for torch's default 10 minutes
            pg_timeout = datetime.timedelta(seconds=float(os.environ.get("TMLQCD_BENCH_PG_TIMEOUT_S", "180")))
            if backend == "nccl":
                ndev = torch.cuda.device_count()
                if local_rank >= ndev:
                    raise SystemExit("[bench] rank %d wants GPU %d but this node shows %d device(s)" % (rank, local_rank, ndev))
                torch.cuda.set_device(local_rank)
                self.dev = "cuda"
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), timeout=pg_timeout)
                self.n_devices = world
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world, timeout=pg_timeout)
                if self.shm:
                    ndev = max(torch.cuda.device_count(), 1)
                    self.device = local_rank % ndev
                    self.n_devices = min(world, ndev)

    def barrier(self, lat=None):
        if lat is not None:
            lat.sync()
        if self.on:
            if self.dev == "cuda":
                self.torch.cuda.synchronize()
            self.dist.barrier()

    def allmax(self, *vals):
        if not self.on:
            return vals if len(vals) > 1 else vals[0]
        t = self.torch.tensor(list(vals), dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        out = [float(x) for x in t]
        return out if len(out) > 1 else out[0]

    def agree(self, ok):
        """Every rank calls this at the same point with its own verdict; every rank gets the same answer: the list of ranks that
        reported a failure (empty = go on).  One all-reduce of a status word per rank -- the ranks of a multi-rank run must take
        the same branch, or they meet in mismatched collectives (benchmark.c has the same shape: every rank runs every phase)."""
        if not self.on:
            return [] if ok else [0]
        t = self.torch.zeros(self.world, dtype=self.torch.float64, device=self.dev)
        if not ok:
            t[self.rank] = 1.0
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [i for i, v in enumerate(t.tolist()) if v > 0.5]

    def bcast_uid(self, uid_bytes):
        if not self.on:
            return uid_bytes
        u = self.torch.zeros(128, dtype=self.torch.uint8, device=self.dev)
        if self.rank == 0:
            u.copy_(self.torch.tensor(list(uid_bytes), dtype=self.torch.uint8))
        self.dist.broadcast(u, 0)
        return bytes(u.cpu().tolist())

    def gather0(self, arr):
        """numpy array of every rank -> list on rank 0 (None elsewhere)."""
        import numpy as np
        if not self.on:
            return [arr]
        t = self.torch.from_numpy(np.ascontiguousarray(arr)).to(self.dev)
        bufs = [self.torch.empty_like(t) for _ in range(self.world)] if self.rank == 0 else None
        self.dist.gather(t, bufs, dst=0)
        return [b.cpu().numpy() for b in bufs] if self.rank == 0 else None

    def close(self):
        if self.on:
            self.dist.destroy_process_group()


class PhaseAbort(Exception):
    """Raised on EVERY rank at the same step when any rank reported a failure there."""


class Phase:
    """A leg of the run as a sequence of steps.  step(name, fn) runs fn on this rank (exceptions are caught), then all ranks
    agree on the outcome; if any rank failed, every rank leaves the phase at this very step (PhaseAbort) -- nobody goes on into
    a collective the others will not reach.  fn may call the library (whose halo exchanges are collective over the ranks: every
    rank runs the same fn) but no torch.distributed collective: those are steps of their own (`collective`).
    TMLQCD_BENCH_INJECT_FAIL="<rank>:<step name>" makes that rank raise in that step (the CPU test of this machinery)."""

    def __init__(self, R, name):
        self.R, self.name = R, name
        self.error = None          # (step, failed ranks, this rank's message) once aborted
        inj = os.environ.get("TMLQCD_BENCH_INJECT_FAIL", "")
        self.inject = inj.split(":", 1) if ":" in inj else None

    def step(self, name, fn):
        val, msg = None, None
        t0 = time.perf_counter()
        try:
            if self.inject and int(self.inject[0]) == self.R.rank and self.inject[1] == name:
                raise RuntimeError("injected failure in step %r on rank %d" % (name, self.R.rank))
            val = fn()
        except Exception as e:    # noqa: BLE001 -- every failure of a rank has to reach the agreement below
            msg = repr(e)
        if os.environ.get("TMLQCD_BENCH_TRACE"):
            sys.stderr.write("[bench] rank %d: %s / %s: %.3f s%s\n" % (self.R.rank, self.name, name, time.perf_counter() - t0, "" if msg is None else " FAILED " + msg))
        failed = self.R.agree(msg is None)
        if failed:
            self.error = {"phase": self.name, "step": name, "failed_ranks": failed, "error": msg}
            sys.stderr.write("[bench] rank %d: phase %r stops at step %r (failed ranks %s)%s\n"
                             % (self.R.rank, self.name, name, failed, ": " + msg if msg else ""))
            raise PhaseAbort(name)
        return val

    def collective(self, fn):
        """A torch.distributed collective: reached by all ranks (the step before it was agreed on), not wrapped."""
        return fn()


def make_lattice(ph, R, T, L, args, nproc_t, faces="comm"):
    """Lattice of this rank; on a T-split lattice the RCCL ring along T for the half-spinor faces (+ its split for the reductions):
    unique id from rank 0, broadcast by the host program (tmLQCD: MPI_Bcast).  Every rank runs every step.  faces "direct": the
    direct carrier on top of the ring (tmhip_comm_init_ipc, collective: all ranks or none)."""
    from tmlqcd_amd import Lattice
    box = {}

    def create():
        lat = Lattice(T, L, L, L, kappa=0.125, mu=0.01, nproc_t=nproc_t, proc_t=R.rank if nproc_t > 1 else 0, device=R.device)
        for kv in args.opt:
            k, v = kv.split("=")
            lat.set_option(k, int(v))
        box["lat"] = lat
        if R.on and nproc_t > 1 and R.rank == 0:
            if R.shm:
                make_lattice.jobs = getattr(make_lattice, "jobs", 0) + 1
                box["uid"] = ("bench_%d_%d_%x" % (os.getpid(), make_lattice.jobs, int.from_bytes(os.urandom(4), "little"))).encode().ljust(128, b"\0")
            else:
                box["uid"] = lat.comm_unique_id()
        else:
            box["uid"] = b"\0" * 128
    ph.step("create lattice", create)
    lat = box["lat"]
    if R.on and nproc_t > 1:
        uid = ph.collective(lambda: R.bcast_uid(box["uid"]))
        if R.shm:
            ph.step("comm_init", lambda: lat.comm_init_shm(uid.rstrip(b"\0").decode()))
        else:
            ph.step("comm_init", lambda: lat.comm_init(uid))
        if faces == "direct":
            ph.step("comm_init_ipc", lat.comm_init_ipc)
    if args.loopback and R.world == 1:
        lat.set_loopback(args.loopback)
        if faces == "direct" and args.loopback == 2:
            lat.comm_init_ipc()     # behind the one-rank RCCL communicator: the collective set-up over RCCL with np = 1, the rank its own neighbour
    return lat


def comm_labels(R, S, faces):
    """What the line says about the ring: `rccl_nranks` only when RCCL built it (null otherwise: the driver's "did RCCL see N ranks?"),
    `ring_nranks` whatever built it, `faces` = what carries the half-spinor faces."""
    nf, nr = S.comm_count()
    direct, sharers = S.comm_faces_direct()
    return {"transport": R.ring, "faces": "direct" if direct else R.ring, "sums": "direct" if S.comm_sums_direct() else R.ring, "ring_nranks": [nf, nr],
            "rccl_nranks": [nf, nr] if R.ring == "rccl" else None, "comm_split": S.comm_is_split() if R.ring == "rccl" else None,
            "ranks_sharing_a_gpu": sharers if direct else None}


CALIBRATION_STEPS = int(os.environ.get("TMLQCD_BENCH_CALIBRATION_STEPS", "512"))


def time_hopping(ph, R, lat, f0, f1, f2, steps, warmup, what="hopping"):
    """benchmark.c's calibration pass (benchmark.c:262-281: j_max = 512 applications of {H_eo, H_oe} before the measurement, which the
    reference uses to size its timed loop and which leaves the machine in its steady state), then `warmup` untimed steps, then exactly
    `steps` steps bracketed by a barrier + synchronisation on both sides (the closing one is the step's own agreement all-reduce); max over
    ranks of (wall seconds, HIP-event ms).  The pass is reported in the line (`config.calibration_steps`); with five warm-up steps alone the
    first timed launches of a fresh process run 2 % below the steady rate (profiles/r04_warmup_check.log)."""
    if CALIBRATION_STEPS > 0:
        ph.step(what + " calibration pass", lambda: (lat.bench_hopping(f0, f1, f2, CALIBRATION_STEPS), lat.sync()))
    ph.step(what + " warm-up", lambda: (lat.bench_hopping(f0, f1, f2, max(warmup, 1)), lat.sync()))
    ph.collective(lambda: R.barrier(lat))

    def timed():
        t0 = time.perf_counter()
        ev_ms = lat.bench_hopping(f0, f1, f2, steps)     # HIP events on the stream the kernels run on
        lat.sync()
        return time.perf_counter() - t0, ev_ms
    dt, ev = ph.step(what + " timed loop", timed)
    return ph.collective(lambda: R.allmax(dt, ev))


def time_nocom(ph, R, lat, f0, f1, f2, steps):
    """benchmark.c:336-374: the same loop with communication switched off (Hopping_Matrix_nocom: stale faces)."""
    def loop(n):
        for _ in range(n):
            lat.Hopping_Matrix_nocom(0, f1, f0)
            lat.Hopping_Matrix_nocom(1, f2, f1)
        lat.sync()
    ph.step("nocom warm-up", lambda: loop(2))
    ph.collective(lambda: R.barrier(lat))

    def timed():
        t0 = time.perf_counter()
        loop(steps)
        return time.perf_counter() - t0
    dt = ph.step("nocom timed loop", timed)
    return ph.collective(lambda: R.allmax(dt))


def time_cg(ph, R, lat, P, Q, total_iters, n_short=5, n_long=25):
    """cg_her iterations/s on a LIVE residual: every solve starts from P = 0 and stops after a fixed count, long before the
    residual of this well-conditioned system reaches the rounding floor (it needs ~24 iterations per 10 orders); the per-iteration time is
    (t(n_long) - t(n_short)) / (n_long - n_short), so the once-per-solve set-up (cg_her.c:82-88) is not counted as iterations."""
    reps = max(1, total_iters // (n_long - n_short))

    def solve(n):
        ph.step("cg zero", lambda: (P.zero(), lat.sync()))
        ph.collective(lambda: R.barrier(lat))

        def timed():
            t0 = time.perf_counter()
            lat.cg_her(P, Q, n, 0.0, 1, lat.Vh)
            lat.sync()
            return time.perf_counter() - t0
        dt = ph.step("cg_her %d iterations" % n, timed)
        return ph.collective(lambda: R.allmax(dt))
    solve(n_short)
    solve(n_long)
    ts = tl = 0.0
    for _ in range(reps):
        ts += solve(n_short)
        tl += solve(n_long)
    iters = reps * (n_long - n_short)
    return {"iters_per_s": iters / (tl - ts), "iters": iters, "ms_per_iter": 1e3 * (tl - ts) / iters,
            "method": "%d x (t(cg_her, %d iterations) - t(cg_her, %d iterations)), every solve from P = 0, eps_sq = 0" % (reps, n_long, n_short),
            "ms_per_solve_setup": 1e3 * (ts / reps - n_short * (tl - ts) / iters)}


REF_CACHE = {}   # rank 0: the unsplit lattice's results per global T (built once, reused by the second carrier's leg)


def split_leg(R, args, L, Tg, what, steps, faces="comm"):
    """One global Tg x L^3 lattice split in T over the ranks, inside the run that is about to be timed:
      1. every rank builds its slab; rank 0 ALSO computes the unsplit lattice (Hopping_Matrix, Qtm_pm_psi, a global norm, a
         cg_her solve) on its GPU -- BEFORE any split operation, while the other ranks wait in the step's agreement: nobody starts
         a halo exchange its neighbour is not ready for (tmLQCD's ranks arrive together the same way: MPI_Barrier, benchmark.c:263);
      2. the same operations on the split lattice, gathered to rank 0 and compared slab by slab (`rank_check`);
      3. the benchmark.c loop, its communication-free twin (benchmark.c:336-374) and cg_her on the split lattice.
    Returns (check, timing); either is a dict with ok = False and the step that failed when a rank reported a failure --
    every rank leaves the leg at that same step."""
    import numpy as np
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn
    w, r = R.world, R.rank
    Ts = Tg // w
    ph = Phase(R, what)
    chk = tim = None
    box = {}
    t_leg = time.perf_counter()
    try:
        S = make_lattice(ph, R, Ts, L, args, w, faces)
        box["S"] = S

        def upload():
            S.set_gauge(syn.gauge_field(7, Ts, L, L, L, w, r))
            box["k"] = S.field(syn.spinor_field_eo(8, 0, Ts, L, L, L, w, r))
            box["q9"] = S.field(syn.spinor_field_eo(9, 1, Ts, L, L, L, w, r))
            box["comm"] = comm_labels(R, S, faces)
            S.sync()
        ph.step("upload", upload)

        def reference():
            if r != 0:
                return
            if (Tg, L) in REF_CACHE:
                box["ref"] = REF_CACHE[(Tg, L)]
                return
            G = Lattice(Tg, L, L, L, kappa=0.125, mu=0.01, device=R.device)
            G.set_gauge(syn.gauge_field(7, Tg, L, L, L))
            gk = G.field(syn.spinor_field_eo(8, 0, Tg, L, L, L))
            gl, gq, gP = G.field(), G.field(), G.field()
            G.Hopping_Matrix(0, gl, gk)
            G.Qtm_pm_psi(gq, gk)
            gn = G.square_norm(gq, G.Vh, 1)
            git, _ = G.cg_her(gP, gk, 2000, 1e-20, 1, G.Vh)
            box["ref"] = REF_CACHE[(Tg, L)] = ([gl.download(), gq.download(), gP.download()], gn, git)
            G.close()
        ph.step("unsplit reference on rank 0", reference)      # the other ranks wait here, in the agreement

        def split_ops():
            k = box["k"]
            l, q, P = S.field(), S.field(), S.field()
            S.Hopping_Matrix(0, l, k)
            S.Qtm_pm_psi(q, k)
            nrm = S.square_norm(q, S.Vh, 1)
            it, _ = S.cg_her(P, k, 2000, 1e-20, 1, S.Vh)
            box["mine"] = ([f.download() for f in (l, q, P)], nrm, it)
            for f in (l, q, P):
                f.free()
        ph.step("split operators", split_ops)
        got = [ph.collective(lambda j=j: R.gather0(box["mine"][0][j])) for j in range(3)]

        def compare():
            if r != 0:
                return None
            ref, gn, git = box["ref"]
            _, nrm, it = box["mine"]
            Vh = Ts * L ** 3 // 2
            dev = []
            for slabs, b in zip(got, ref):
                sc = np.abs(b).max()
                dev.append(max(float(np.abs(slabs[j] - b[j * Vh:(j + 1) * Vh]).max() / sc) for j in range(w)))
            out = {"lattice": "%dx%d^3 split in T over %d ranks (T_local %d) vs the unsplit lattice on rank 0" % (Tg, L, w, Ts),
                   "hopping_matrix_max_rel_dev": dev[0], "qtm_pm_psi_max_rel_dev": dev[1], "cg_solution_max_rel_dev": dev[2],
                   "global_norm_rel_dev": abs(nrm - gn) / gn, "cg_iters_split": it, "cg_iters_unsplit": git,
                   "worst_operator_dev": max(dev[0], dev[1], abs(nrm - gn) / gn)}
            out["ok"] = bool(dev[0] <= 1e-13 and dev[1] <= 1e-13 and out["global_norm_rel_dev"] <= 1e-13 and abs(it - git) <= 1 and dev[2] <= 1e-8)
            out.update(box["comm"])
            sys.stderr.write("[bench] %s rank check: %s\n" % (what, json.dumps(out)))
            return out
        chk = ph.step("compare with the unsplit lattice", compare)

        f0, f1, f2 = box["k"], S.field(), S.field()
        dts, evs = time_hopping(ph, R, S, f0, f1, f2, steps, args.warmup, what="split hopping")
        dtn = time_nocom(ph, R, S, f0, f1, f2, steps)
        cgs = time_cg(ph, R, S, S.field(), box["q9"], min(args.cg_iters, 100))
        Vs = Ts * L ** 3
        tim = {"config": "global %dx%d^3 split in T over %d ranks (T_local %d), half-spinor faces %s" % (Tg, L, w, Ts, "as direct stores into the neighbours' memory (ring: %s)" % R.ring if box["comm"]["faces"] == "direct" else "over " + R.transport),
               "value": w * 1608.0 / (1e6 * dts / (steps * Vs)), "unit": "Mflop/s", "ms_per_step": 1e3 * dts / steps,
               "us_per_launch": 1e3 * evs / (2 * steps), "steps": steps, "cg_iters_per_s": cgs["iters_per_s"],
               "nocom": {"value": w * 1608.0 / (1e6 * dtn / (steps * Vs)), "ms_per_step": 1e3 * dtn / steps,
                         "exposed_comm_ms_per_step": 1e3 * (dts - dtn) / steps,
                         "note": "communication switched off (Hopping_Matrix_nocom), benchmark.c:336-374"}}
        tim.update(box["comm"])
    except PhaseAbort:
        err = dict(ph.error, ok=False)
        if chk is None:
            chk = err
        else:
            tim = err
    finally:
        if "S" in box:
            try:
                box["S"].close()
            except Exception:   # noqa: BLE001
                pass
    WALL[what] = time.perf_counter() - t_leg
    return chk, tim


WALL = {}        # wall seconds per leg of this rank's run (rank 0's go into the line)


GUARDIAN = r"""
import json, signal, sys
signal.signal(signal.SIGTERM, signal.SIG_IGN); signal.signal(signal.SIGINT, signal.SIG_IGN); signal.signal(signal.SIGHUP, signal.SIG_IGN)
saved, done = None, False
for ln in sys.stdin:                      # ends when rank 0 does, however it ends
    if ln.startswith("SAVE "):
        saved = ln[5:]
    elif ln.startswith("DONE"):
        done = True
if saved is not None and not done:
    line = json.loads(saved)
    line["faces_direct"] = {"ok": False, "error": "rank 0 ended during the direct-carrier legs; this is the line measured over the communicator before them"}
    print(json.dumps(line), flush=True)
"""


class Guardian:
    """The line measured over the communicator must survive whatever the direct-carrier legs behind it do to this process (a GPU fault
    aborts it; a fault on another rank has the launcher end this one).  Rank 0 starts a small child BEFORE anything touches the GPU and
    hands it the finished line in front of those legs; if rank 0 then ends without having printed its own line, the child prints the
    saved one (marked `faces_direct.ok: false`) to the same stdout.  It never touches the GPU and ends with rank 0."""

    def __init__(self, stdout_fd):
        # a session of its own: torchrun ends a worker by signalling its whole process GROUP (SubprocessHandler.close: os.killpg)
        self.p = subprocess.Popen([sys.executable, "-c", GUARDIAN], stdin=subprocess.PIPE, stdout=stdout_fd, text=True, start_new_session=True)

    def _say(self, what):
        try:
            self.p.stdin.write(what + "\n")
            self.p.stdin.flush()
        except (OSError, ValueError):
            pass

    def save(self, line):
        self._say("SAVE " + json.dumps(line))

    def done(self):
        self._say("DONE")
        try:
            self.p.stdin.close()
            self.p.wait(timeout=10)
        except Exception:
            pass


def rank_main(args, world, rank, local_rank):
    # stdout carries exactly ONE JSON line: RCCL prints a version banner to fd 1 when a communicator is
    # created, so everything before the final print goes to stderr.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    guard = Guardian(saved_stdout) if (rank == 0 and (world > 1 or args.rehearse_split)) else None
    os.dup2(2, 1)
    R = Ranks(world, rank, local_rank, os.environ.get("TMLQCD_BENCH_FORCE_TORCH") == "1")
    import numpy as np   # noqa: F401 (the next-row legs)
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn

    L = args.L
    extra = {}
    t_run = time.perf_counter()
    faces_a = "direct" if R.faces_mode == "direct" else "comm"        # what carries the faces in the legs below ("auto": the communicator first)
    LABELS = ("transport", "faces", "sums", "ring_nranks", "rccl_nranks", "comm_split", "ranks_sharing_a_gpu")
    # ---------------------------------------------------------------- N > 1: multi-rank parity check + BASELINE configs[3] (strong scaling)
    Tg = 64
    if Tg % world or (Tg // world) % 2 or Tg // world < 2:
        Tg = 8 * world                                      # odd rank counts: any even split serves the check
    steps_s = max(args.steps, 20)
    split_legs = (world > 1 or (args.rehearse_split and args.loopback)) and not args.no_rank_check
    if split_legs:
        chk, tim = split_leg(R, args, L, Tg, "configs[3]", steps_s, faces_a)
        extra["rank_check"] = chk
        if tim is not None:
            if tim.get("ok", True):
                tim["config"] = "BASELINE configs[3]: " + tim["config"]
            extra["strong"] = tim
        for key in LABELS:
            if isinstance(chk, dict) and key in chk:
                extra[key] = chk[key]
        # north_star's literal "32^4 at 1 / 2 / 4 / 8 GPUs": the headline lattice of ONE GPU cut N ways (T_local = 32 / N)
        if L % world == 0 and (L // world) % 2 == 0:
            chk32, tim32 = split_leg(R, args, L, L, "strong_32", steps_s, faces_a)
            if tim32 is not None and tim32.get("ok", True):
                tim32["config"] = "north_star: 32^4 cut in T over N GPUs -- " + tim32["config"]
            extra["strong_32"] = dict(tim32 or {}, rank_check=chk32)

    # ---------------------------------------------------------------- headline: weak scaling, L^4 per GPU (N = 1: BASELINE configs[2])
    T = args.T or L
    if args.strong:
        if args.strong % world or (args.strong // world) % 2:
            raise SystemExit("--strong %d cannot be split evenly (even local T) over %d ranks" % (args.strong, world))
        T = args.strong // world
    V = T * L ** 3
    hl = Phase(R, "headline")
    box = {}
    t_hl = time.perf_counter()
    try:
        lat = make_lattice(hl, R, T, L, args, world, faces_a)

        def upload():
            lat.set_gauge(syn.gauge_field(7, T, L, L, L, world, rank))
            box["src"] = syn.spinor_field_eo(8, 0, T, L, L, L, world, rank)
            box["f"] = (lat.field(box["src"]), lat.field(), lat.field())
            box["PQ"] = (lat.field(), lat.field(syn.spinor_field_eo(9, 1, T, L, L, L, world, rank)))
            lat.sync()
        hl.step("upload", upload)          # its agreement is the barrier in front of the first split stencil: every rank has its links and fields
        if world > 1:
            for key, v in comm_labels(R, lat, faces_a).items():
                extra.setdefault(key, v)
        src = box["src"]
        f0, f1, f2 = box["f"]
        dt, ev_ms = time_hopping(hl, R, lat, f0, f1, f2, args.steps, args.warmup)
        gpu_out = f2.download() if (rank == 0 and world == 1 and not args.no_cpu) else None
        try_direct = (world > 1 or (args.rehearse_split and args.loopback == 2)) and R.faces_mode == "auto"
        f2_ref = f2.download() if try_direct else None         # this rank's slab of H_oe H_eo f0: what the other carrier has to reproduce
        # --- CG part of the metric: cg_her on Qtm_pm_psi (solver/cg_her.c:91-126)
        P, Q = box["PQ"]
        cg = time_cg(hl, R, lat, P, Q, args.cg_iters)
        WALL["headline"] = time.perf_counter() - t_hl
    except PhaseAbort:
        # the headline itself failed on some rank: still ONE well-formed line (value null, the step and the ranks), non-zero exit
        if rank == 0:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            print(json.dumps(dict({"metric": METRIC, "value": None, "unit": "Mflop/s", "n_gpus": R.n_devices, "n_ranks": world, "steps": args.steps, "warmup": args.warmup,
                                   "higher_is_better": True, "error": hl.error, "wall_s": dict(WALL, total=time.perf_counter() - t_run)}, **extra)), flush=True)
            os.dup2(2, 1)
        R.close()
        return 1
    t_info = time.perf_counter()

    def leg(name, body):
        """An informational leg as a phase of its own: a failure (on any rank) costs this leg, never the headline line, and every
        rank leaves it at the same step."""
        ph = Phase(R, name)
        try:
            return body(ph)
        except PhaseAbort:
            return {"error": ph.error}

    # --- full-size property of the (multi-rank) operator: Q_+ = Q_-^dagger, i.e. Re<y, Q_+ x> = Re<Q_- y, x> with global sums
    def herm(ph):
        def fn():
            y, a, b = lat.field(src), lat.field(), lat.field()
            lat.op("Qtm_plus_psi", a, Q)
            lat.op("Qtm_minus_psi", b, y)
            s1, s2 = lat.scalar_prod_r(y, a, lat.Vh, 1), lat.scalar_prod_r(b, Q, lat.Vh, 1)
            for f in (y, a, b):
                f.free()
            return abs(s1 - s2) / max(abs(s1), 1e-300)
        return ph.step("Q_plus = Q_minus^dagger", fn)
    extra["hermiticity_rel_dev"] = leg("hermiticity", herm)

    # --- time to solution: cg_her vs mixed_cg_her (fp32 inner / fp64 restart) to |r|/|b| = 1e-10 (BASELINE configs[1])
    def solves(ph):
        out = {}
        ph.step("fp32 set-up", lambda: lat.mixed_cg_her(P, Q, 2, 1e-20, 1, lat.Vh))     # untimed: builds the fp32 gauge copy and work fields once per configuration
        for name in ("cg_her", "mixed_cg_her"):
            ph.step("zero", lambda: (P.zero(), lat.sync()))
            ph.collective(lambda: R.barrier(lat))

            def timed(name=name):
                t2 = time.perf_counter()
                if name == "cg_her":
                    its, outer = lat.cg_her(P, Q, 5000, 1e-20, 1, lat.Vh)[0], None
                else:
                    its, outer = lat.mixed_cg_her(P, Q, 5000, 1e-20, 1, lat.Vh)
                lat.sync()
                return its, outer, time.perf_counter() - t2
            its, outer, dts = ph.step("solve " + name, timed)
            dts = ph.collective(lambda: R.allmax(dts))

            def residual():
                Rf = lat.field()                             # true residual on the device in fp64
                lat.Qtm_pm_psi(Rf, P)
                lat.diff(Rf, Q, Rf, lat.Vh)
                res = lat.square_norm(Rf, lat.Vh, 1) / lat.square_norm(Q, lat.Vh, 1)
                Rf.free()
                return res
            out[name] = {"iters": its, "seconds": dts, "true_rel_res_sq": ph.step("true residual " + name, residual)}
            if outer is not None:
                out[name]["outer_iters"] = outer
        return out
    cg["solve_to_1e-10"] = leg("solve to 1e-10", solves)

    # --- benchmark.c:336-374: on a split lattice the reference also times the loop with communication switched off
    nocom = None
    if world > 1 or args.loopback:
        def nocom_leg(ph):
            dtn = time_nocom(ph, R, lat, f0, f1, f2, args.steps)
            return {"value": world * 1608.0 / (1e6 * dtn / (args.steps * V)), "unit": "Mflop/s", "ms_per_step": 1e3 * dtn / args.steps,
                    "exposed_comm_ms_per_step": 1e3 * (dt - dtn) / args.steps,
                    "note": "communication switched off (Hopping_Matrix_nocom), benchmark.c:336-374"}
        nocom = leg("nocom", nocom_leg)

    # --- informational: the same loop with the opt-in 12-real gauge read (third row of each link rebuilt in registers;
    # exact for SU(3) links, guarded on the device).  Never part of `value`: the headline is the plain 18-real path.
    recon = None
    if world == 1:                                      # keep the multi-rank run to the headline measurement
        def recon_leg(ph):
            lat.set_option("gauge_recon", 12)
            try:
                dt12, ev12 = time_hopping(ph, R, lat, f0, f1, f2, args.steps, 2, what="recon12 hopping")
                cg12 = time_cg(ph, R, lat, P, Q, min(args.cg_iters, 100))
            finally:
                lat.set_option("gauge_recon", 18)
            return {"value": world * 1608.0 / (1e6 * dt12 / (args.steps * V)), "unit": "Mflop/s", "cg_iters_per_s": cg12["iters_per_s"],
                    "us_per_launch": 1e3 * ev12 / (2 * args.steps), "alg_bytes_per_site": 1152,
                    "achieved_GBps": 1152.0 * (V // 2) / (ev12 * 1e-3 / (2 * args.steps)) / 1e9,
                    "note": "opt-in tmhip_set_option(gauge_recon, 12); COMPRESSION_12 of misc_types.h:29-33"}
        recon = leg("gauge_recon 12", recon_leg)

    # --- what this box's memory system delivers to plain streaming kernels ("measure a device-to-device copy / triad
    # on the box and quote the fraction of THAT too"): copy R = S and triad P += c Q over several pairs of fields in turn, so that the
    # working set (8 x 101 MB at 32^4) is beyond the 256 MB Infinity Cache
    stream = None
    if world == 1 and not args.loopback:
        def stream_leg(ph):
            def fn():
                Vh = V // 2
                fs = [lat.field() for _ in range(8)]           # four pairs of one-parity fields, taken in turn
                res = {}
                for name, call, nbytes in (("copy", lambda i: lat.assign(fs[2 * (i & 3)], fs[2 * (i & 3) + 1], Vh), 2 * 192.0 * Vh),
                                           ("triad", lambda i: lat.assign_add_mul_r(fs[2 * (i & 3)], fs[2 * (i & 3) + 1], 0.5, Vh), 3 * 192.0 * Vh)):
                    for i in range(4):
                        call(i)
                    lat.event_record(12)
                    for i in range(40):
                        call(i)
                    lat.event_record(13)
                    res[name + "_GBps"] = nbytes * 40 / (lat.event_elapsed_ms(12, 13) * 1e-3) / 1e9
                for x in fs:
                    x.free()
                res["note"] = "assign / assign_add_mul_r over four pairs of one-parity fp64 fields in turn (8 x %.0f MB: beyond the Infinity Cache), HIP events over 40 calls" % (192.0 * Vh / 1e6)
                return res
            return ph.step("stream kernels", fn)
        stream = leg("stream", stream_leg)

    # --- BASELINE configs[1] (16^4, one GPU): the launch-bound end of the path
    cg16 = None
    if world == 1 and not args.loopback and (T, L) == (32, 32):
        def cg16_leg(ph):
            l16 = Lattice(16, 16, 16, 16, kappa=0.125, mu=0.01, device=R.device)
            l16.set_gauge(syn.gauge_field(7, 16, 16, 16, 16))
            P16, Q16 = l16.field(), l16.field(syn.spinor_field_eo(9, 1, 16, 16, 16, 16))
            c = time_cg(ph, R, l16, P16, Q16, 200)
            g0, g1, g2 = l16.field(syn.spinor_field_eo(8, 0, 16, 16, 16, 16)), l16.field(), l16.field()
            d16, e16 = time_hopping(ph, R, l16, g0, g1, g2, 500, 50, what="16^4 hopping")
            l16.close()
            return {"lattice": "16^4", "iters_per_s": c["iters_per_s"], "ms_per_iter": c["ms_per_iter"],
                    "hopping_us_per_launch": 1e3 * e16 / 1000, "hopping_Mflop/s": 1608.0 / (1e6 * d16 / (500 * 16 ** 4))}
        cg16 = leg("16^4", cg16_leg)

    # --- informational: the rows next to the path (SURVEY section 8 f) on the headline lattice, everything resident in HBM.  One
    # molecular-dynamics step of the clover determinant without its solves (update_gauge.c, clover_term.c, clover_invert.c, deriv_Sb.c,
    # clover_deriv.c, clover_accumulate_deriv.c, update_momenta.c) and an ILDG record into / out of the resident links.
    rows = None
    if world == 1 and not args.loopback and not args.no_rows:
        try:
            kap, csw = 0.125, 1.5
            lat.momenta_upload(np.zeros((lat.V, 4, 8)))
            a, b = lat.field(src), lat.field()
            lat.Hopping_Matrix(0, b, a)

            def md_step():
                lat.update_gauge(0.0)                        # (zero momenta: the links stay SU(3) and the same)
                lat.sw_term(None, kap, csw); lat.sw_invert(0, lat_mu)
                lat.derivative_zero(); lat.swpm_zero()
                lat.deriv_Sb(1, a, b, 0.5); lat.deriv_Sb(0, b, a, 0.5)
                lat.sw_spinor_eo(0, b, b, 0.5); lat.sw_spinor_eo(1, a, a, 0.5)
                lat.sw_deriv(0, lat_mu); lat.sw_all(kap, csw)
                lat.update_momenta(0.0)
            lat_mu = 0.01
            md_step(); lat.sync()
            t2 = time.perf_counter()
            for _ in range(5):
                md_step()
            lat.sync()
            md_ms = 1e3 * (time.perf_counter() - t2) / 5
            rec, sums = lat.gauge_pack_ildg(64)
            t2 = time.perf_counter(); lat.gauge_unpack_ildg(rec, 64); t_un = time.perf_counter() - t2
            t2 = time.perf_counter(); lat.gauge_pack_ildg(64, out=rec); t_pk = time.perf_counter() - t2     # into a buffer whose pages exist, like the unpack leg's source
            rows = {"md_step_ms": md_ms, "md_step": "update_gauge, sw_term, sw_invert, 2 x deriv_Sb, 2 x sw_spinor_eo, sw_deriv, sw_all, update_momenta "
                                                    "on the links / momenta / derivative resident in HBM (no solves)",
                    "ildg_unpack_ms": 1e3 * t_un, "ildg_pack_ms": 1e3 * t_pk, "ildg_record_MB": rec.size / 1e6,
                    "ildg_note": "604 MB ildg-binary-data record <-> resident links incl. the PCIe copy of the record; checksum %08x %08x" % sums}
            for f in (a, b):
                f.free()
        except Exception as e:                            # informational legs never cost the headline line
            rows = {"error": repr(e)}

    WALL["informational legs"] = time.perf_counter() - t_info
    out = None
    if rank == 0:
        sdt = 1e6 * dt / (args.steps * V)                   # us per site-update, benchmark.c:318
        mflops = world * 1608.0 / sdt                       # benchmark.c:327 "Mflops(total)"
        launches = 2 * args.steps
        t_launch = ev_ms * 1e-3 / launches                  # average Hopping_Matrix launch duration (HIP events)
        alg_bytes = 1536.0 * (V // 2)                       # 1536 B per output site x sites per launch
        achieved = alg_bytes / t_launch / 1e9
        traffic, traffic_src = None, None
        pj = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(pj) and L == 32 and T == 32 and world == 1:
            try:
                traffic = json.load(open(pj)).get("bytes_per_launch")
                traffic_src = "profiles/hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (tools/profile.sh), not re-measured in this run"
            except Exception:
                traffic = None
        out = {
            "metric": METRIC,
            "value": mflops, "unit": "Mflop/s", "n_gpus": R.n_devices, "n_ranks": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "benchmark.c loop {Hopping_Matrix(0);Hopping_Matrix(1)}, local lattice %dx%d^3 per GPU, "
                                   "global %dx%d^3, fp64, kappa=0.125, periodic, random SU(3) gauge + Gaussian spinor"
                                   % (T, L, T * world, L),
                       "local_lattice": [T, L, L, L], "global_lattice": [T * world, L, L, L],
                       "calibration_steps": CALIBRATION_STEPS,      # untimed, before the `warmup` steps: benchmark.c:262-281 (see time_hopping)
                       "parallelism": ("T-split ring of %d, half-spinor faces %s" % (world, "as direct stores into the neighbours' memory (ring: %s)" % R.ring if faces_a == "direct" else "over " + R.transport) if world > 1 else
                                       ("single GPU, split-phase path rehearsed with self-exchange (loopback %d)" % args.loopback
                                        if args.loopback else "single GPU"))},
            "lattice_updates_per_s": args.steps / dt, "us_per_site": sdt,
            "cg": dict(cg, operator="Qtm_pm_psi", N="VOLUME/2"),
            "cg_16": cg16, "gauge_recon12": recon, "nocom": nocom, "next_rows": rows,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": "hop_kernel (Hopping_Matrix, one parity)",
                         "us_per_launch": 1e6 * t_launch, "achieved_2880B_model": achieved * 2880.0 / 1536.0},
        }
        if isinstance(stream, dict) and stream.get("triad_GBps"):
            # the vendor-nominal 8 TB/s stays `peak`; next to it what streaming kernels reach on this box, and the stencil against that
            out["roofline"]["measured_stream"] = dict(stream, frac_of_triad=achieved / stream["triad_GBps"], frac_of_copy=achieved / stream["copy_GBps"])
        out.update(extra)
        if world > 1 and R.n_devices < world:
            # several ranks per GPU (TMLQCD_BENCH_TRANSPORT=shm / ipc on a box with fewer GPUs than ranks): the multi-rank CODE runs as real
            # processes, the aggregate `value` is NOT a scaling measurement -- consumers of scaling curves filter on this flag
            out["rehearsal"] = True
        if gpu_out is not None:
            try:
                cb, parity = cpu_baseline(args, T, L, gpu_out)
                out["cpu_baseline"] = cb
                out["parity_max_rel_err_vs_cpu"] = parity
                out["gpu_over_cpu"] = mflops / cb["value"]
            except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU line
                out["cpu_baseline"] = {"value": None, "unit": "Mflop/s", "cores": 0, "kind": "unavailable", "sample": repr(e)}
    lat.close()

    def emit(line):
        line["wall_s"] = dict(WALL, total=time.perf_counter() - t_run)
        if guard is not None:
            guard.done()
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)

    # ---------------------------------------------------------------- the second carrier (N > 1, TMLQCD_BENCH_FACES=auto): the direct one
    if try_direct:
        budget = float(os.environ.get("TMLQCD_BENCH_DIRECT_BUDGET_S", "240"))

        def give_up():   # the line measured over the communicator is complete: nothing that happens to this attempt may cost it
            sys.stderr.write("[bench] rank %d: the direct-carrier legs did not finish within %.0f s: keeping the communicator's line\n" % (rank, budget))
            if rank == 0:
                out["faces_direct"] = {"ok": False, "error": "not finished within %.0f s (TMLQCD_BENCH_DIRECT_BUDGET_S)" % budget}
                emit(out)
            os._exit(0)
        dog = threading.Timer(budget, give_up)
        dog.daemon = True
        dog.start()
        if guard is not None:
            guard.save(dict(out, wall_s=dict(WALL, total=time.perf_counter() - t_run)))
        fault = os.environ.get("TMLQCD_BENCH_TEST_FAULT", "")      # tests: a rank that dies in these legs (as a GPU fault would end it)
        if (fault == "abort_direct" and rank == 0) or (fault == "abort_direct_peer" and rank == 1):
            os.abort()
        os.environ.setdefault("TMLQCD_HIP_FLAG_TIMEOUT_S", "20")       # a neighbour that never pushes ends this attempt, not the run
        res = direct_legs(R, args, L, T, Tg if split_legs else 0, steps_s, f2_ref, dt)
        dog.cancel()
        if rank == 0:
            out["faces_direct"] = res
            hd = res.get("headline") or {}
            if res.get("ok") and hd.get("value", 0.0) > out["value"]:
                # the faster carrier that reproduced the communicator's fields gives the headline; the other stays in the line
                out["carriers"] = {R.ring: {k: out[k] for k in ("value", "ms_per_step", "lattice_updates_per_s", "us_per_site")}, "direct": hd}
                out["carriers"][R.ring]["cg_iters_per_s"] = out["cg"]["iters_per_s"]
                for k in ("value", "ms_per_step", "lattice_updates_per_s", "us_per_site"):
                    out[k] = hd[k]
                out["cg"] = dict(out["cg"], iters_per_s=hd["cg_iters_per_s"], ms_per_iter=1e3 / hd["cg_iters_per_s"], carrier="direct")
                out["faces"] = "direct"
                out["config"]["parallelism"] = "T-split ring of %d, half-spinor faces as direct stores into the neighbours' memory (sums over %s); %s" % (world, R.transport, hd.get("check", ""))
                t_l = hd["us_per_launch"] * 1e-6
                out["roofline"].update(achieved=alg_bytes / t_l / 1e9, frac=alg_bytes / t_l / 1e9 / 8000.0, us_per_launch=hd["us_per_launch"],
                                       achieved_2880B_model=alg_bytes / t_l / 1e9 * 2880.0 / 1536.0)
                for key in ("strong", "strong_32"):
                    st = (res.get(key) or {})
                    if st.get("ok", True) and (st.get("rank_check") or {}).get("ok") and key in out and st.get("value", 0) > (out[key].get("value") or 0):
                        out[key + "_over_" + R.ring] = out[key]
                        out[key] = st
    if rank == 0:
        emit(out)
    R.close()
    return 0


def direct_legs(R, args, L, T, Tg, steps_s, f2_ref, dt_comm):
    """The direct face carrier (tmhip_comm_init_ipc) on the lattices just measured over the communicator: configs[3] against the
    unsplit lattice again (rank 0's reference is cached), the headline lattice against the communicator's own output on every rank
    (f2 = H_oe H_eo f0 of the benchmark loop, slab by slab), then the same timed loops.  Every rank takes every step; a failure
    anywhere ends the attempt on all ranks at that step."""
    import numpy as np
    from tmlqcd_amd import synthetic as syn
    world, rank = R.world, R.rank
    res = {"ok": False}
    if Tg:
        chk, tim = split_leg(R, args, L, Tg, "configs[3] direct", steps_s, "direct")
        res["strong"] = dict(tim or {}, rank_check=chk)
        if tim is not None and tim.get("ok", True):
            res["strong"]["config"] = "BASELINE configs[3]: " + tim["config"]
        # (the comparison lives on rank 0: every rank takes ITS verdict, or they part ways here)
        good = R.allmax(1.0 if (rank == 0 and isinstance(chk, dict) and chk.get("ok")) else 0.0)
        if good < 0.5:
            res["error"] = "configs[3] over the direct carrier did not reproduce the unsplit lattice"
            return res
        if L % world == 0 and (L // world) % 2 == 0:          # north_star's literal 32^4 over N GPUs: the smallest T_local of the run, where the carrier matters most
            chk32, tim32 = split_leg(R, args, L, L, "strong_32 direct", steps_s, "direct")
            res["strong_32"] = dict(tim32 or {}, rank_check=chk32)
            good = R.allmax(1.0 if (rank == 0 and isinstance(chk32, dict) and chk32.get("ok")) else 0.0)
            if good < 0.5:
                res["error"] = "32^4 / N over the direct carrier did not reproduce the unsplit lattice"
                return res
    ph = Phase(R, "headline direct")
    t0 = time.perf_counter()
    box = {}
    try:
        lat = make_lattice(ph, R, T, L, args, world, "direct")
        box["lat"] = lat

        def upload():
            lat.set_gauge(syn.gauge_field(7, T, L, L, L, world, rank))
            box["f"] = (lat.field(syn.spinor_field_eo(8, 0, T, L, L, L, world, rank)), lat.field(), lat.field())
            box["PQ"] = (lat.field(), lat.field(syn.spinor_field_eo(9, 1, T, L, L, L, world, rank)))
            lat.sync()
        ph.step("upload", upload)
        f0, f1, f2 = box["f"]
        dt, ev_ms = time_hopping(ph, R, lat, f0, f1, f2, args.steps, args.warmup, what="direct hopping")

        def compare():
            dev = float(np.abs(f2.download() - f2_ref).max() / np.abs(f2_ref).max())
            if not dev <= 1e-13:
                raise RuntimeError("rank %d: the benchmark loop's output differs from the communicator's by %.3e" % (rank, dev))
            return dev
        dev = ph.step("compare with the communicator's output", compare)
        dev = ph.collective(lambda: R.allmax(dev))
        P, Q = box["PQ"]
        cg = time_cg(ph, R, lat, P, Q, min(args.cg_iters, 100))
        V = T * L ** 3
        sdt = 1e6 * dt / (args.steps * V)
        res["headline"] = {"value": world * 1608.0 / sdt, "unit": "Mflop/s", "ms_per_step": 1e3 * dt / args.steps, "lattice_updates_per_s": args.steps / dt,
                           "us_per_site": sdt, "us_per_launch": 1e3 * ev_ms / (2 * args.steps), "cg_iters_per_s": cg["iters_per_s"],
                           "max_rel_dev_vs_communicator": dev, "speedup_vs_communicator": dt_comm / dt,
                           "check": "output of the timed loop equal to the communicator's on every rank (max rel. dev %.1e)" % dev}
        res["headline"].update(comm_labels(R, lat, "direct"))
        res["ok"] = True
    except PhaseAbort:
        res["error"] = ph.error
    finally:
        if "lat" in box:
            try:
                box["lat"].close()
            except Exception:   # noqa: BLE001
                pass
    WALL["headline direct"] = time.perf_counter() - t0
    return res


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None:
        n = args.gpus or 1
        if n < 1:
            raise SystemExit("--gpus must be >= 1")
        if n > 1:                                     # before torch or the HIP library is imported: this process stays off the GPU
            return parent_launch(n, args)
        world, rank, local_rank = 1, 0, int(os.environ.get("LOCAL_RANK", "0"))
    else:
        world, rank, local_rank = int(env_world), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
        if args.gpus is not None and args.gpus != world:
            sys.stderr.write("[bench] --gpus %d contradicts WORLD_SIZE=%d: start one rank per GPU (python -m torch.distributed.run "
                             "--nproc-per-node %d bench.py --gpus %d) or run `python bench.py --gpus %d` without a torchrun environment\n"
                             % (args.gpus, world, args.gpus, args.gpus, args.gpus))
            return 2
    if os.environ.get("TMLQCD_BENCH_RENDEZVOUS_ONLY") == "1":
        return rendezvous_only(world, rank)
    if os.environ.get("TMLQCD_BENCH_AGREE_SELFTEST") == "1":
        return agree_selftest(world, rank)
    return rank_main(args, world, rank, local_rank)


if __name__ == "__main__":
    sys.exit(main())
