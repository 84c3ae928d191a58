"""TEST / MEASUREMENT INFRASTRUCTURE ONLY (see oracle/README.md): one leg of bench.py's `cpu_baseline`.

Times the reference's own loop {Hopping_Matrix(0); Hopping_Matrix(1)} (benchmark.c:291-300) on the host cores with ONE
thread count, in a process of its own so that OMP_PROC_BIND / OMP_PLACES / OMP_NUM_THREADS (set by the caller) are seen
by libgomp when it starts and the fields are first-touched by exactly the threads that will use them
(oracle/ref_harness.c).  Inputs are the same seeded synthetic arrays bench.py gave the GPU; with --gpu-out the GPU's f2
is compared with the CPU's before anything is timed.

    OMP_NUM_THREADS=64 OMP_PROC_BIND=close OMP_PLACES=cores python oracle/cpu_baseline.py --T 32 --L 32 --threads 64 --seconds 8

Prints one JSON line.  Runs on the CPU only; never imports the HIP library.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, required=True)
    ap.add_argument("--L", type=int, required=True)
    ap.add_argument("--threads", type=int, required=True)
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--min-iters", type=int, default=100)
    ap.add_argument("--max-seconds", type=float, default=40.0)
    ap.add_argument("--gpu-out", default="", help=".npy file with the GPU's f2 for the parity check")
    ap.add_argument("--gauge-seed", type=int, default=7)
    ap.add_argument("--spinor-seed", type=int, default=8)
    args = ap.parse_args()
    import numpy as np
    from oracle import refbind
    from tmlqcd_amd import synthetic as syn
    T, L = args.T, args.L
    V = T * L ** 3
    N = V // 2
    gauge = syn.gauge_field(args.gauge_seed, T, L, L, L)
    src = syn.spinor_field_eo(args.spinor_seed, 0, T, L, L, L)
    if refbind.ref_available(omp=True):
        kind = "reference"
        ref = refbind.RefLattice(T, L, L, L, kappa=0.125, mu=0.01, nfields=6, omp=True, threads=args.threads)
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
        orc = Oracle(T, L, L, L, kappa=0.125, mu=0.01, threads=args.threads)
        orc.set_gauge(gauge)
        f = [orc.new_field() for _ in range(3)]
        f[0][:N] = src

        def step():
            orc.Hopping_Matrix(0, f[1], f[0])
            orc.Hopping_Matrix(1, f[2], f[1])

        def result():
            return f[2][:N]
        threads = args.threads
        what = "oracle/libtmoracle.so (our C restatement of the reference algorithm, gcc -O3 -fopenmp)"
    step()  # warm-up; also builds the gauge copy (Hopping_Matrix.c:135-139), first-touching it in parallel
    parity = None
    if args.gpu_out:
        gpu = np.load(args.gpu_out)
        cpu = result()
        parity = float(np.abs(gpu - cpu).max() / np.abs(cpu).max())
    step()
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        dt = time.perf_counter() - t0
        if (dt >= args.seconds and n >= args.min_iters) or dt >= args.max_seconds:
            break
    sdt = 1e6 * dt / (n * V)
    print(json.dumps({"value": 1608.0 / sdt, "unit": "Mflop/s", "cores": threads, "kind": kind, "iterations": n, "seconds": dt,
                      "us_per_site": sdt, "parity_max_rel_err_gpu_vs_cpu": parity, "what": what,
                      "omp_env": {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OMP_PROC_BIND", "OMP_PLACES")}}), flush=True)


if __name__ == "__main__":
    main()
