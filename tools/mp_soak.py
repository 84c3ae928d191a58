"""Soak of the direct face carrier between real processes on one GPU: the randomized operation sequences of
tests/test_gpu_multiprocess.py::test_random_sequences_between_real_processes with OTHER seeds, longer sequences and every direct form,
for 2, 4 and 5 ranks.  `python tools/mp_soak.py FIRST_SEED N_SEEDS [NOPS]`; one line per (seed, world, form); exit 1 at the first failure
(nothing is retried)."""
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_gpu_split_stress import DIRECT_FORMS  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
nops = int(sys.argv[3]) if len(sys.argv) > 3 else 120
worker = os.path.join(ROOT, "tests", "mp_stress_worker.py")
t_all = time.time()
with tempfile.TemporaryDirectory() as tmp:
    for seed in range(first, first + count):
        for world, Tg in ((2, 16), (4, 16), (5, 20)):
            env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", TMLQCD_HIP_FLAG_TIMEOUT_S="60", MP_TG=str(Tg))
            ref = subprocess.run([sys.executable, worker, "0", "1", "none", tmp, str(seed), str(nops), "flags"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
            if ref.returncode != 0:
                print("reference run failed:", ref.stderr[-2000:]); sys.exit(1)
            one = {k: v for k, v in np.load(os.path.join(tmp, "stress_flags_0_of_1.npz")).items()}
            for name, _ in DIRECT_FORMS:
                form = "direct: " + name
                tag = re.sub(r"[^A-Za-z0-9]+", "_", form)
                job = "soak_%d_%d_%d_%s" % (os.getpid(), seed, world, tag[:20])
                t0 = time.time()
                procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), job, tmp, str(seed), str(nops), form], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                         for r in range(world)]
                outs = [p.communicate(timeout=400) for p in procs]
                bad = [se[-1500:] for p, (so, se) in zip(procs, outs) if p.returncode != 0 or "gave up" in se]
                worst = 0.0
                if not bad:
                    for r in range(world):
                        part = np.load(os.path.join(tmp, "stress_%s_%d_of_%d.npz" % (tag, r, world)))
                        if len(part["scal"]) != len(one["scal"]) or not np.allclose(part["scal"], one["scal"], rtol=1e-11, atol=1e-11):
                            bad.append("rank %d: scalars differ" % r)
                        for i in range(5):
                            full = one["f%d" % i]
                            n = full.shape[0] // world
                            worst = max(worst, float(np.abs(part["f%d" % i] - full[r * n:(r + 1) * n]).max() / np.abs(full).max()))
                    if worst >= 1e-11:
                        bad.append("fields differ: %.3e" % worst)
                print("seed %3d world %d %-45s %s  max dev %.1e  %.1f s" % (seed, world, name, "FAILED" if bad else "ok", worst, time.time() - t0), flush=True)
                if bad:
                    print("\n".join(bad)); sys.exit(1)
print("soak done: %d seeds x 3 worlds x %d forms, %d operations each, %.0f s" % (count, len(DIRECT_FORMS), nops, time.time() - t_all))
