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


def _bench(nranks, extra=()):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TMLQCD_BENCH_TRANSPORT="shm", HSA_ENABLE_IPC_MODE_LEGACY="0", TMLQCD_HIP_FLAG_TIMEOUT_S="60")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--L", "16", "--steps", "20", "--warmup", "2", "--cg-iters", "20",
           "--no-cpu", "--no-rows"] + list(extra)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
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
    assert rec["n_gpus"] == nranks and rec["value"] and rec["value"] > 0
    assert rec["rccl_nranks"] == [nranks, nranks] and rec["comm_split"] is False      # (the transport reports its ring; no second communicator)
    st = rec["strong"]
    assert st.get("ok", True) and st["value"] > 0 and st["cg_iters_per_s"] > 0 and st["nocom"]["value"] > 0
    if 16 % nranks == 0 and (16 // nranks) % 2 == 0:
        s32 = rec["strong_32"]
        assert s32["rank_check"]["ok"] is True, s32
    assert rec["cg"]["iters_per_s"] > 0 and abs(rec["hermiticity_rel_dev"]) < 1e-12
