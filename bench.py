#!/usr/bin/env python3
"""bench.py -- benchmark.c's Hopping_Matrix loop (+ cg_her iterations/s) on N MI355X.

A "step" is one iteration of the reference's timed loop (benchmark.c:291-300):
    Hopping_Matrix(0, f1, f0); Hopping_Matrix(1, f2, f1)      (= VOLUME output sites per GPU)
on fields resident in HBM.  `value` follows benchmark.c:318,327: Mflop/s = nranks * 1608 / (us per
site-update).  N=1 workload: 32^4 fp64 (BASELINE.json configs[2]); N>1: weak scaling, every rank
holds a 32^4 slab of a 32^3 x (32 N) lattice split in T, half-spinor faces exchanged over RCCL
and overlapped with the interior stencil; the same run also measures BASELINE configs[3]
(32^3 x 64 split N ways, the `strong` object) after checking it slab by slab against the unsplit
lattice computed on rank 0 (`rank_check`).

    python bench.py --gpus N --steps K --warmup W

Launch: one process per GPU.  Under torchrun (WORLD_SIZE set) this process IS one rank.  Started
plainly with --gpus N > 1 it becomes a parent that never touches the GPU: it starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child process and relays
rank 0's single JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
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
    ap.add_argument("--opt", action="append", default=[], help="library option name=value (A/B runs)")
    ap.add_argument("--loopback", type=int, default=0,
                    help="1-GPU rehearsal of the multi-GPU path: 1 = faces exchanged with self by D2D copies, 2 = through a one-rank RCCL communicator, "
                         "3 = written straight into the receive buffers by the pack kernel (diagnostic)")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------- launcher (parent, never touches the GPU)
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def parent_launch(n):
    """--gpus N without a torchrun environment: start N ranks as a CHILD process tree (never an exec from a process that has
    touched the GPU; this one has not even imported torch) and relay rank 0's JSON line."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("[bench] starting %d ranks: %s\n" % (n, " ".join(cmd)))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and ("\"metric\"" in ln or "\"rendezvous\"" in ln):
            line = ln
        else:
            sys.stderr.write(ln + "\n")
    if line is None:
        sys.stderr.write("[bench] the %d-rank run produced no result line (exit code %d)\n" % (n, r.returncode))
        return r.returncode or 1
    print(line, flush=True)
    return r.returncode


def rendezvous_only(world, rank):
    """TMLQCD_BENCH_RENDEZVOUS_ONLY=1 (the CPU test of the launcher): the ranks meet over gloo, agree on who is there, and rank 0
    prints one line -- nothing GPU-side is imported."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.zeros(world, dtype=torch.int64)
    t[rank] = os.getpid()
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"rendezvous": "ok", "n_gpus": world, "pids": t.tolist(), "backend": "gloo"}), flush=True)
    dist.destroy_process_group()
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
    """torch.distributed plumbing of one rank (rendezvous, barrier, max over ranks, gather to rank 0)."""

    def __init__(self, world, rank, local_rank, force):
        self.world, self.rank, self.local_rank = world, rank, local_rank
        self.on = world > 1 or force
        self.torch = self.dist = None
        if self.on:
            # torch first: its bundled HIP/RCCL runtime must be the one every later library binds to
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            ndev = torch.cuda.device_count()
            if local_rank >= ndev:
                raise SystemExit("[bench] rank %d wants GPU %d but this node shows %d device(s)" % (rank, local_rank, ndev))
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    def barrier(self, lat=None):
        if lat is not None:
            lat.sync()
        if self.on:
            self.torch.cuda.synchronize()
            self.dist.barrier()

    def allmax(self, *vals):
        if not self.on:
            return vals if len(vals) > 1 else vals[0]
        t = self.torch.tensor(list(vals), dtype=self.torch.float64, device="cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        out = [float(x) for x in t]
        return out if len(out) > 1 else out[0]

    def bcast_uid(self, uid_bytes):
        if not self.on:
            return uid_bytes
        u = self.torch.zeros(128, dtype=self.torch.uint8, device="cuda")
        if self.rank == 0:
            u.copy_(self.torch.tensor(list(uid_bytes), dtype=self.torch.uint8))
        self.dist.broadcast(u, 0)
        return bytes(u.cpu().tolist())

    def gather0(self, arr):
        """numpy array of every rank -> list on rank 0 (None elsewhere)."""
        import numpy as np
        if not self.on:
            return [arr]
        t = self.torch.from_numpy(np.ascontiguousarray(arr)).cuda()
        bufs = [self.torch.empty_like(t) for _ in range(self.world)] if self.rank == 0 else None
        self.dist.gather(t, bufs, dst=0)
        return [b.cpu().numpy() for b in bufs] if self.rank == 0 else None

    def close(self):
        if self.on:
            self.dist.destroy_process_group()


def make_lattice(R, T, L, args, nproc_t):
    from tmlqcd_amd import Lattice
    lat = Lattice(T, L, L, L, kappa=0.125, mu=0.01, nproc_t=nproc_t, proc_t=R.rank if nproc_t > 1 else 0, device=R.local_rank)
    if R.on and nproc_t > 1:
        # RCCL ring along T for the half-spinor faces (+ its split for the reductions): unique id from rank 0, broadcast by the host program
        lat.comm_init(R.bcast_uid(lat.comm_unique_id() if R.rank == 0 else b"\0" * 128))
    if args.loopback and R.world == 1:
        lat.set_loopback(args.loopback)
    for kv in args.opt:
        k, v = kv.split("=")
        lat.set_option(k, int(v))
    return lat


def time_hopping(R, lat, f0, f1, f2, steps, warmup):
    """warmup untimed steps, then exactly `steps` steps between two barriers; max over ranks of (wall seconds, HIP-event ms)."""
    lat.bench_hopping(f0, f1, f2, max(warmup, 1))
    R.barrier(lat)
    t0 = time.perf_counter()
    ev_ms = lat.bench_hopping(f0, f1, f2, steps)     # HIP events on the stream the kernels run on
    R.barrier(lat)
    dt = time.perf_counter() - t0
    return R.allmax(dt, ev_ms)


def time_nocom(R, lat, f0, f1, f2, steps):
    """benchmark.c:336-374: the same loop with communication switched off (Hopping_Matrix_nocom: stale faces)."""
    for _ in range(2):
        lat.Hopping_Matrix_nocom(0, f1, f0)
        lat.Hopping_Matrix_nocom(1, f2, f1)
    R.barrier(lat)
    t0 = time.perf_counter()
    for _ in range(steps):
        lat.Hopping_Matrix_nocom(0, f1, f0)
        lat.Hopping_Matrix_nocom(1, f2, f1)
    R.barrier(lat)
    return R.allmax(time.perf_counter() - t0)


def time_cg(R, lat, P, Q, total_iters, n_short=5, n_long=25):
    """cg_her iterations/s on a LIVE residual: every solve starts from P = 0 and stops after a fixed count, long before the
    residual of this well-conditioned system reaches the rounding floor (it needs ~24 iterations per 10 orders); the per-iteration time is
    (t(n_long) - t(n_short)) / (n_long - n_short), so the once-per-solve set-up (cg_her.c:82-88) is not counted as iterations."""
    reps = max(1, total_iters // (n_long - n_short))

    def solve(n):
        P.zero()
        R.barrier(lat)
        t0 = time.perf_counter()
        lat.cg_her(P, Q, n, 0.0, 1, lat.Vh)
        R.barrier(lat)
        return R.allmax(time.perf_counter() - t0)
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


def rank_check(R, S, T_loc, L, args):
    """The multi-rank path against the unsplit lattice, inside the run that is about to be timed (body of tools/multi_rank_check.py):
    Hopping_Matrix, Hopping_Matrix_nocom (shape only), Qtm_pm_psi, a global norm (ncclAllReduce) and a cg_her solve on the T-split
    lattice held by lattice S of every rank, gathered to rank 0 and compared slab by slab with the same operations on the
    unsplit (T_loc * world) x L^3 lattice computed on rank 0's GPU."""
    import numpy as np
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn
    w, r = R.world, R.rank
    k = S.field(syn.spinor_field_eo(8, 0, T_loc, L, L, L, w, r))
    l, q, P = S.field(), S.field(), S.field()
    S.Hopping_Matrix(0, l, k)
    S.Qtm_pm_psi(q, k)
    nrm = S.square_norm(q, S.Vh, 1)
    it, hist = S.cg_her(P, k, 2000, 1e-20, 1, S.Vh)
    got = [R.gather0(f.download()) for f in (l, q, P)]
    for f in (k, l, q, P):
        f.free()
    out = None
    if r == 0:
        Tg = T_loc * w
        G = Lattice(Tg, L, L, L, kappa=0.125, mu=0.01, device=R.local_rank)
        G.set_gauge(syn.gauge_field(7, Tg, L, L, L))
        gk = G.field(syn.spinor_field_eo(8, 0, Tg, L, L, L))
        gl, gq, gP = G.field(), G.field(), G.field()
        G.Hopping_Matrix(0, gl, gk)
        G.Qtm_pm_psi(gq, gk)
        gn = G.square_norm(gq, G.Vh, 1)
        git, _ = G.cg_her(gP, gk, 2000, 1e-20, 1, G.Vh)
        ref = [gl.download(), gq.download(), gP.download()]
        G.close()
        Vh = T_loc * L ** 3 // 2
        dev = []
        for slabs, b in zip(got, ref):
            sc = np.abs(b).max()
            dev.append(max(float(np.abs(slabs[j] - b[j * Vh:(j + 1) * Vh]).max() / sc) for j in range(w)))
        out = {"lattice": "%dx%d^3 split in T over %d ranks (T_local %d) vs the unsplit lattice on rank 0" % (Tg, L, w, T_loc),
               "hopping_matrix_max_rel_dev": dev[0], "qtm_pm_psi_max_rel_dev": dev[1], "cg_solution_max_rel_dev": dev[2],
               "global_norm_rel_dev": abs(nrm - gn) / gn, "cg_iters_split": it, "cg_iters_unsplit": git,
               "worst_operator_dev": max(dev[0], dev[1], abs(nrm - gn) / gn)}
        out["ok"] = bool(dev[0] <= 1e-13 and dev[1] <= 1e-13 and out["global_norm_rel_dev"] <= 1e-13 and abs(it - git) <= 1 and dev[2] <= 1e-8)
        sys.stderr.write("[bench] rank check: %s\n" % json.dumps(out))
    return out


def rank_main(args, world, rank, local_rank):
    # stdout carries exactly ONE JSON line: RCCL prints a version banner to fd 1 when a communicator is
    # created, so everything before the final print goes to stderr.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    R = Ranks(world, rank, local_rank, os.environ.get("TMLQCD_BENCH_FORCE_TORCH") == "1")
    import numpy as np
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn

    L = args.L
    extra = {}
    # ---------------------------------------------------------------- N > 1: multi-rank parity check + BASELINE configs[3] (strong scaling)
    if world > 1 and not args.no_rank_check:
        try:
            Tg = 64
            if Tg % world or (Tg // world) % 2 or Tg // world < 2:
                Tg = 8 * world                                      # odd rank counts: any even split serves the check
            Ts = Tg // world
            S = make_lattice(R, Ts, L, args, world)
            S.set_gauge(syn.gauge_field(7, Ts, L, L, L, world, rank))
            extra["rccl_nranks"] = list(S.comm_count())
            chk = rank_check(R, S, Ts, L, args)
            f0 = S.field(syn.spinor_field_eo(8, 0, Ts, L, L, L, world, rank))
            f1, f2 = S.field(), S.field()
            steps = max(args.steps, 20)
            dts, evs = time_hopping(R, S, f0, f1, f2, steps, args.warmup)
            dtn = time_nocom(R, S, f0, f1, f2, steps)
            P, Q = S.field(), S.field(syn.spinor_field_eo(9, 1, Ts, L, L, L, world, rank))
            cgs = time_cg(R, S, P, Q, min(args.cg_iters, 100))
            Vs = Ts * L ** 3
            extra["strong"] = {"config": "BASELINE configs[3]: global %dx%d^3 split in T over %d GPUs (T_local %d), half-spinor faces over RCCL" % (Tg, L, world, Ts),
                               "value": world * 1608.0 / (1e6 * dts / (steps * Vs)), "unit": "Mflop/s", "ms_per_step": 1e3 * dts / steps,
                               "us_per_launch": 1e3 * evs / (2 * steps), "steps": steps, "cg_iters_per_s": cgs["iters_per_s"],
                               "nocom": {"value": world * 1608.0 / (1e6 * dtn / (steps * Vs)), "ms_per_step": 1e3 * dtn / steps,
                                         "exposed_comm_ms_per_step": 1e3 * (dts - dtn) / steps,
                                         "note": "communication switched off (Hopping_Matrix_nocom), benchmark.c:336-374"}}
            extra["rank_check"] = chk
            S.close()
        except Exception as e:   # the check must never cost the headline line; its absence is visible in the line
            extra["rank_check"] = {"ok": False, "error": repr(e)}

    # ---------------------------------------------------------------- headline: weak scaling, L^4 per GPU (N = 1: BASELINE configs[2])
    T = args.T or L
    if args.strong:
        if args.strong % world or (args.strong // world) % 2:
            raise SystemExit("--strong %d cannot be split evenly (even local T) over %d ranks" % (args.strong, world))
        T = args.strong // world
    V = T * L ** 3
    lat = make_lattice(R, T, L, args, world)
    if world > 1:
        extra.setdefault("rccl_nranks", list(lat.comm_count()))
    gauge = syn.gauge_field(7, T, L, L, L, world, rank)
    lat.set_gauge(gauge)
    del gauge
    src = syn.spinor_field_eo(8, 0, T, L, L, L, world, rank)
    f0, f1, f2 = lat.field(src), lat.field(), lat.field()
    dt, ev_ms = time_hopping(R, lat, f0, f1, f2, args.steps, args.warmup)
    gpu_out = f2.download() if (rank == 0 and world == 1 and not args.no_cpu) else None

    # --- CG part of the metric: cg_her on Qtm_pm_psi (solver/cg_her.c:91-126)
    P, Q = lat.field(), lat.field(syn.spinor_field_eo(9, 1, T, L, L, L, world, rank))
    cg = time_cg(R, lat, P, Q, args.cg_iters)

    # --- full-size property of the (multi-rank) operator: Q_+ = Q_-^dagger, i.e. Re<y, Q_+ x> = Re<Q_- y, x> with global sums
    try:
        y, a, b = lat.field(src), lat.field(), lat.field()
        lat.op("Qtm_plus_psi", a, Q)
        lat.op("Qtm_minus_psi", b, y)
        s1, s2 = lat.scalar_prod_r(y, a, lat.Vh, 1), lat.scalar_prod_r(b, Q, lat.Vh, 1)
        extra["hermiticity_rel_dev"] = abs(s1 - s2) / max(abs(s1), 1e-300)
        for f in (y, a, b):
            f.free()
    except Exception as e:
        extra["hermiticity_rel_dev"] = repr(e)

    # --- time to solution: cg_her vs mixed_cg_her (fp32 inner / fp64 restart) to |r|/|b| = 1e-10 (BASELINE configs[1])
    solve = {}
    try:                                              # extra legs never cost the headline line
        lat.mixed_cg_her(P, Q, 2, 1e-20, 1, lat.Vh)     # untimed: builds the fp32 gauge copy and work fields once per configuration
        for name in ("cg_her", "mixed_cg_her"):
            P.zero()
            R.barrier(lat)
            t2 = time.perf_counter()
            if name == "cg_her":
                its, _ = lat.cg_her(P, Q, 5000, 1e-20, 1, lat.Vh)
                outer = None
            else:
                its, outer = lat.mixed_cg_her(P, Q, 5000, 1e-20, 1, lat.Vh)
            R.barrier(lat)
            dts = time.perf_counter() - t2
            Rf = lat.field()                             # true residual on the device in fp64
            lat.Qtm_pm_psi(Rf, P)
            lat.diff(Rf, Q, Rf, lat.Vh)
            res = lat.square_norm(Rf, lat.Vh, 1) / lat.square_norm(Q, lat.Vh, 1)
            Rf.free()
            solve[name] = {"iters": its, "seconds": dts, "true_rel_res_sq": res}
            if outer is not None:
                solve[name]["outer_iters"] = outer
    except Exception as e:
        solve["error"] = repr(e)
    cg["solve_to_1e-10"] = solve

    # --- benchmark.c:336-374: on a split lattice the reference also times the loop with communication switched off
    nocom = None
    try:
        if world > 1 or args.loopback:
            dtn = time_nocom(R, lat, f0, f1, f2, args.steps)
            nocom = {"value": world * 1608.0 / (1e6 * dtn / (args.steps * V)), "unit": "Mflop/s", "ms_per_step": 1e3 * dtn / args.steps,
                     "exposed_comm_ms_per_step": 1e3 * (dt - dtn) / args.steps,
                     "note": "communication switched off (Hopping_Matrix_nocom), benchmark.c:336-374"}
    except Exception as e:
        nocom = {"value": None, "note": repr(e)}

    # --- informational: the same loop with the opt-in 12-real gauge read (third row of each link rebuilt in registers;
    # exact for SU(3) links, guarded on the device).  Never part of `value`: the headline is the plain 18-real path.
    recon = None
    try:
        if world > 1:
            raise RuntimeError("single-GPU leg")      # keep the multi-rank run to the headline measurement
        lat.set_option("gauge_recon", 12)
        dt12, ev12 = time_hopping(R, lat, f0, f1, f2, args.steps, 2)
        cg12 = time_cg(R, lat, P, Q, min(args.cg_iters, 100))
        recon = {"value": world * 1608.0 / (1e6 * dt12 / (args.steps * V)), "unit": "Mflop/s", "cg_iters_per_s": cg12["iters_per_s"],
                 "us_per_launch": 1e3 * ev12 / (2 * args.steps), "alg_bytes_per_site": 1152,
                 "achieved_GBps": 1152.0 * (V // 2) / (ev12 * 1e-3 / (2 * args.steps)) / 1e9,
                 "note": "opt-in tmhip_set_option(gauge_recon, 12); COMPRESSION_12 of misc_types.h:29-33"}
    except Exception as e:                            # informational leg: never a reason to lose the headline line
        recon = {"value": None, "note": repr(e)}
    finally:
        lat.set_option("gauge_recon", 18)

    # --- BASELINE configs[1] (16^4, one GPU): the launch-bound end of the path
    cg16 = None
    if world == 1 and not args.loopback and (T, L) == (32, 32):
        try:
            l16 = Lattice(16, 16, 16, 16, kappa=0.125, mu=0.01, device=local_rank)
            l16.set_gauge(syn.gauge_field(7, 16, 16, 16, 16))
            P16, Q16 = l16.field(), l16.field(syn.spinor_field_eo(9, 1, 16, 16, 16, 16))
            c = time_cg(R, l16, P16, Q16, 200)
            g0, g1, g2 = l16.field(syn.spinor_field_eo(8, 0, 16, 16, 16, 16)), l16.field(), l16.field()
            d16, e16 = time_hopping(R, l16, g0, g1, g2, 500, 50)
            cg16 = {"lattice": "16^4", "iters_per_s": c["iters_per_s"], "ms_per_iter": c["ms_per_iter"],
                    "hopping_us_per_launch": 1e3 * e16 / 1000, "hopping_Mflop/s": 1608.0 / (1e6 * d16 / (500 * 16 ** 4))}
            l16.close()
        except Exception as e:
            cg16 = {"error": repr(e)}

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
            t2 = time.perf_counter(); lat.gauge_pack_ildg(64); t_pk = time.perf_counter() - t2
            rows = {"md_step_ms": md_ms, "md_step": "update_gauge, sw_term, sw_invert, 2 x deriv_Sb, 2 x sw_spinor_eo, sw_deriv, sw_all, update_momenta "
                                                    "on the links / momenta / derivative resident in HBM (no solves)",
                    "ildg_unpack_ms": 1e3 * t_un, "ildg_pack_ms": 1e3 * t_pk, "ildg_record_MB": rec.size / 1e6,
                    "ildg_note": "604 MB ildg-binary-data record <-> resident links incl. the PCIe copy of the record; checksum %08x %08x" % sums}
            for f in (a, b):
                f.free()
        except Exception as e:                            # informational legs never cost the headline line
            rows = {"error": repr(e)}

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
            "value": mflops, "unit": "Mflop/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "benchmark.c loop {Hopping_Matrix(0);Hopping_Matrix(1)}, local lattice %dx%d^3 per GPU, "
                                   "global %dx%d^3, fp64, kappa=0.125, periodic, random SU(3) gauge + Gaussian spinor"
                                   % (T, L, T * world, L),
                       "local_lattice": [T, L, L, L], "global_lattice": [T * world, L, L, L],
                       "parallelism": ("T-split ring of %d, half-spinor faces over RCCL" % world if world > 1 else
                                       ("single GPU, split-phase path rehearsed with self-exchange (loopback %d)" % args.loopback
                                        if args.loopback else "single GPU"))},
            "lattice_updates_per_s": args.steps / dt, "us_per_site": sdt,
            "cg": dict(cg, operator="Qtm_pm_psi", N="VOLUME/2"),
            "cg_16": cg16, "gauge_recon12": recon, "nocom": nocom, "next_rows": rows,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": "hop_kernel (Hopping_Matrix, one parity)",
                         "us_per_launch": 1e6 * t_launch, "achieved_2880B_model": achieved * 2880.0 / 1536.0},
        }
        out.update(extra)
        if gpu_out is not None:
            try:
                cb, parity = cpu_baseline(args, T, L, gpu_out)
                out["cpu_baseline"] = cb
                out["parity_max_rel_err_vs_cpu"] = parity
                out["gpu_over_cpu"] = mflops / cb["value"]
            except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU line
                out["cpu_baseline"] = {"value": None, "unit": "Mflop/s", "cores": 0, "kind": "unavailable", "sample": repr(e)}
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    lat.close()
    R.close()
    return 0


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None:
        n = args.gpus or 1
        if n < 1:
            raise SystemExit("--gpus must be >= 1")
        if n > 1:                                     # before torch or the HIP library is imported: this process stays off the GPU
            return parent_launch(n)
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
    return rank_main(args, world, rank, local_rank)


if __name__ == "__main__":
    sys.exit(main())
