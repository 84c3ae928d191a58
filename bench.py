#!/usr/bin/env python3
"""bench.py -- benchmark.c's Hopping_Matrix loop (+ cg_her iterations/s) on N MI355X.

A "step" is one iteration of the reference's timed loop (benchmark.c:291-300):
    Hopping_Matrix(0, f1, f0); Hopping_Matrix(1, f2, f1)      (= VOLUME output sites per GPU)
on fields resident in HBM.  `value` follows benchmark.c:318,327: Mflop/s = nranks * 1608 / (us per
site-update).  N=1 workload: 32^4 fp64 (BASELINE.json configs[2]); N>1: weak scaling, every rank
holds a 32^4 slab of a 32^3 x (32 N) lattice split in T, half-spinor faces exchanged over RCCL
and overlapped with the interior stencil.

    python bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--L", type=int, default=32, help="spatial extent")
    ap.add_argument("--T", type=int, default=0, help="local time extent (default = L)")
    ap.add_argument("--strong", type=int, default=0, metavar="TGLOBAL",
                    help="strong scaling: fixed global lattice TGLOBAL x L^3 split in T over the ranks "
                         "(BASELINE configs[3]: --strong 64); default is weak scaling with L^4 per GPU")
    ap.add_argument("--cg-iters", type=int, default=200, help="cg_her iterations timed for the CG part of the metric")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall budget of the CPU-baseline leg")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = min(available cores, 16)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--opt", action="append", default=[], help="library option name=value (A/B runs)")
    ap.add_argument("--loopback", type=int, default=0,
                    help="1-GPU rehearsal of the multi-GPU path: 1 = faces exchanged with self by D2D copies, 2 = through a one-rank RCCL communicator")
    return ap.parse_args()


def cpu_baseline(args, T, L, gauge, src, gpu_out):
    """Time the reference CPU path on this box's host cores (rank 0, N=1 only) on the SAME host
    arrays, after checking parity of the GPU result against it."""
    import numpy as np
    from oracle import refbind
    avail = len(os.sched_getaffinity(0))
    threads = args.cpu_threads or min(avail, 16)
    V = T * L ** 3
    N = V // 2
    if refbind.ref_available(omp=True):
        kind = "reference"
        ref = refbind.RefLattice(T, L, L, L, kappa=0.125, mu=0.01, nfields=6, omp=True, threads=threads)
        ref.gauge()[:] = gauge
        ref.mark_gauge_dirty()
        ref.spinor(0, N)[:] = src
        lib = ref.lib

        def step():
            lib.Hopping_Matrix(0, ref.sp(1), ref.sp(0))
            lib.Hopping_Matrix(1, ref.sp(2), ref.sp(1))

        def result():
            return ref.spinor(2, N)
        threads = ref.threads
        what = "oracle/_ref/libtmref_omp.so (reference sources, gcc -O3 -march=x86-64-v3 -fopenmp, _GAUGE_COPY)"
    else:
        kind = "port"
        from oracle.oraclebind import Oracle
        orc = Oracle(T, L, L, L, kappa=0.125, mu=0.01, threads=threads)
        orc.set_gauge(gauge)
        f = [orc.new_field() for _ in range(3)]
        f[0][:N] = src

        def step():
            orc.Hopping_Matrix(0, f[1], f[0])
            orc.Hopping_Matrix(1, f[2], f[1])

        def result():
            return f[2][:N]
        what = "oracle/libtmoracle.so (our C restatement of the reference algorithm, gcc -O3 -fopenmp)"
    step()  # warm-up, also refreshes the gauge copy (Hopping_Matrix.c:135-139)
    cpu_out = result()
    parity = float(np.abs(gpu_out - cpu_out).max() / np.abs(cpu_out).max())
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        dt = time.perf_counter() - t0
        if dt >= args.cpu_seconds or n >= 2000:
            break
    sdt = 1e6 * dt / (n * V)
    return {"value": 1608.0 / sdt, "unit": "Mflop/s", "cores": threads, "kind": kind,
            "sample": "%d iterations of {Hopping_Matrix(0);Hopping_Matrix(1)} on the same %dx%d^3 host arrays, %.1f s, %s"
                      % (n, T, L, dt, what),
            "us_per_site": sdt, "host_cpus_visible": avail}, parity


def main():
    args = parse()
    # stdout carries exactly ONE JSON line: RCCL prints a version banner to fd 1 when a communicator is
    # created, so everything before the final print goes to stderr.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or os.environ.get("TMLQCD_BENCH_FORCE_TORCH") == "1"
    dist = torch = None
    if use_dist:
        # torch first: its bundled HIP/RCCL runtime must be the one every later library binds to
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    import numpy as np
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn

    L = args.L
    T = args.T or L
    if args.strong:
        if args.strong % world or (args.strong // world) % 2:
            raise SystemExit("--strong %d cannot be split evenly (even local T) over %d ranks" % (args.strong, world))
        T = args.strong // world
    nproc_t = world
    V = T * L ** 3
    lat = Lattice(T, L, L, L, kappa=0.125, mu=0.01, nproc_t=nproc_t, proc_t=rank, device=local_rank)
    if use_dist:
        # RCCL ring along T for the half-spinor faces: unique id from rank 0, broadcast by the host program
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            uid.copy_(torch.tensor(list(lat.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(uid, 0)
        lat.comm_init(bytes(uid.cpu().tolist()))       # no-op for a single rank
    if args.loopback and world == 1:
        lat.set_loopback(args.loopback)
    for kv in args.opt:
        k, v = kv.split("=")
        lat.set_option(k, int(v))
    gauge = syn.gauge_field(7, T, L, L, L, nproc_t, rank)
    lat.set_gauge(gauge)
    src = syn.spinor_field_eo(8, 0, T, L, L, L, nproc_t, rank)
    f0, f1, f2 = lat.field(src), lat.field(), lat.field()

    def barrier():
        lat.sync()
        if use_dist:
            torch.cuda.synchronize()
            dist.barrier()

    lat.bench_hopping(f0, f1, f2, max(args.warmup, 1))
    barrier()
    t0 = time.perf_counter()
    ev_ms = lat.bench_hopping(f0, f1, f2, args.steps)     # HIP events on the stream the kernels run on
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt, ev_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, ev_ms = float(tt[0]), float(tt[1])
    gpu_out = f2.download() if (rank == 0 and world == 1 and not args.no_cpu) else None

    # --- CG part of the metric: cg_her on Qtm_pm_psi (solver/cg_her.c:91-126), fixed iteration count
    P, Q = lat.field(), lat.field(syn.spinor_field_eo(9, 1, T, L, L, L, nproc_t, rank))
    lat.cg_her(P, Q, 5, 0.0, 1, lat.Vh)
    P.zero()
    barrier()
    t1 = time.perf_counter()
    it, hist = lat.cg_her(P, Q, args.cg_iters, 0.0, 1, lat.Vh)
    barrier()
    cg_dt = time.perf_counter() - t1
    if use_dist:
        tt = torch.tensor([cg_dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        cg_dt = float(tt[0])

    # --- time to solution: cg_her vs mixed_cg_her (fp32 inner / fp64 restart) to |r|/|b| = 1e-10 (BASELINE configs[1])
    solve = {}
    try:                                              # extra legs never cost the headline line
        lat.mixed_cg_her(P, Q, 2, 1e-20, 1, lat.Vh)     # untimed: builds the fp32 gauge copy and work fields once per configuration
        for name in ("cg_her", "mixed_cg_her"):
            P.zero()
            barrier()
            t2 = time.perf_counter()
            if name == "cg_her":
                its, _ = lat.cg_her(P, Q, 5000, 1e-20, 1, lat.Vh)
                outer = None
            else:
                its, outer = lat.mixed_cg_her(P, Q, 5000, 1e-20, 1, lat.Vh)
            barrier()
            dts = time.perf_counter() - t2
            # true residual on the device in fp64
            R = lat.field()
            lat.Qtm_pm_psi(R, P)
            lat.diff(R, Q, R, lat.Vh)
            res = lat.square_norm(R, lat.Vh, 1) / lat.square_norm(Q, lat.Vh, 1)
            R.free()
            solve[name] = {"iters": its, "seconds": dts, "true_rel_res_sq": res}
            if outer is not None:
                solve[name]["outer_iters"] = outer
    except Exception as e:
        solve["error"] = repr(e)

    # --- benchmark.c:336-374: on a split lattice the reference also times the loop with communication switched off
    # (Hopping_Matrix_nocom: interior + boundary kernels on stale faces) and reports the difference as communication cost
    nocom = None
    try:
        if world > 1 or args.loopback:
            for _ in range(2):
                lat.Hopping_Matrix_nocom(0, f1, f0); lat.Hopping_Matrix_nocom(1, f2, f1)
            barrier()
            t5 = time.perf_counter()
            for _ in range(args.steps):
                lat.Hopping_Matrix_nocom(0, f1, f0); lat.Hopping_Matrix_nocom(1, f2, f1)
            barrier()
            dtn = time.perf_counter() - t5
            if use_dist:
                tt = torch.tensor([dtn], dtype=torch.float64, device="cuda")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dtn = float(tt[0])
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
        lat.bench_hopping(f0, f1, f2, 2)
        barrier()
        t3 = time.perf_counter()
        ev12 = lat.bench_hopping(f0, f1, f2, args.steps)
        barrier()
        dt12 = time.perf_counter() - t3
        if use_dist:
            tt = torch.tensor([dt12, ev12], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt12, ev12 = float(tt[0]), float(tt[1])
        P.zero()
        lat.sync()
        t4 = time.perf_counter()
        lat.cg_her(P, Q, args.cg_iters, 0.0, 1, lat.Vh)
        lat.sync()
        cg12 = args.cg_iters / (time.perf_counter() - t4)
        recon = {"value": world * 1608.0 / (1e6 * dt12 / (args.steps * V)), "unit": "Mflop/s", "cg_iters_per_s": cg12,
                 "us_per_launch": 1e3 * ev12 / (2 * args.steps), "alg_bytes_per_site": 1152,
                 "achieved_GBps": 1152.0 * (V // 2) / (ev12 * 1e-3 / (2 * args.steps)) / 1e9,
                 "note": "opt-in tmhip_set_option(gauge_recon, 12); COMPRESSION_12 of misc_types.h:29-33"}
    except Exception as e:                            # informational leg: never a reason to lose the headline line
        recon = {"value": None, "note": repr(e)}
    finally:
        lat.set_option("gauge_recon", 18)

    if rank == 0:
        sdt = 1e6 * dt / (args.steps * V)                   # us per site-update, benchmark.c:318
        mflops = world * 1608.0 / sdt                       # benchmark.c:327 "Mflops(total)"
        launches = 2 * args.steps
        t_launch = ev_ms * 1e-3 / launches                  # average Hopping_Matrix launch duration (HIP events)
        alg_bytes = 1536.0 * (V // 2)                       # 1536 B per output site x sites per launch
        achieved = alg_bytes / t_launch / 1e9
        traffic = None
        pj = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(pj) and L == 32 and T == 32 and world == 1:
            try:
                traffic = json.load(open(pj)).get("bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Hopping_Matrix Mflop/s per site (benchmark.c) + CG iters/sec, 32^4 fp64",
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
            "cg": {"iters_per_s": args.cg_iters / cg_dt, "iters": args.cg_iters, "operator": "Qtm_pm_psi", "N": "VOLUME/2",
                   "ms_per_iter": 1e3 * cg_dt / args.cg_iters,
                   "solve_to_1e-10": solve},
            "gauge_recon12": recon, "nocom": nocom,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": traffic, "kernel": "hop_kernel (Hopping_Matrix, one parity)",
                         "us_per_launch": 1e6 * t_launch, "achieved_2880B_model": achieved * 2880.0 / 1536.0},
        }
        if gpu_out is not None:
            try:
                cb, parity = cpu_baseline(args, T, L, gauge, src, gpu_out)
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
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
