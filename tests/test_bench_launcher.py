"""bench.py's launcher (CPU): `--gpus N` without a torchrun environment must start N ranks as child processes that really
rendezvous, and a --gpus / WORLD_SIZE contradiction must fail loudly instead of silently running one rank (VERDICT r1
item 1).  The ranks stop after the rendezvous (TMLQCD_BENCH_RENDEZVOUS_ONLY=1, gloo): nothing GPU-side is imported."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_gpus_2_starts_two_ranks_that_meet():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1"], env=_env(TMLQCD_BENCH_RENDEZVOUS_ONLY="1"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                      # exactly ONE line on stdout, relayed from rank 0
    rec = json.loads(lines[0])
    assert rec["rendezvous"] == "ok" and rec["n_gpus"] == 2
    assert len(set(rec["pids"])) == 2 and os.getpid() not in rec["pids"]   # two distinct child processes


def test_gpus_contradicting_world_size_is_an_error():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4"], env=_env(TMLQCD_BENCH_RENDEZVOUS_ONLY="1", WORLD_SIZE="2", RANK="0"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 2
    assert "contradicts WORLD_SIZE" in r.stderr and not r.stdout.strip()


def test_under_torchrun_gpus_defaults_to_world_size():
    port = "29531"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", port, BENCH], env=_env(TMLQCD_BENCH_RENDEZVOUS_ONLY="1"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 2


def _selftest(inject=None):
    env = _env(TMLQCD_BENCH_AGREE_SELFTEST="1")
    if inject:
        env["TMLQCD_BENCH_INJECT_FAIL"] = inject
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                      # ONE well-formed line, whatever happened in the rank check
    return json.loads(lines[0]), r.stderr


def test_ranks_agree_all_steps_pass():
    rec, _ = _selftest()
    assert rec["rank_check"] == {"ok": True}
    assert rec["traces"][0] == "setup|reference|split_ops|compare:[0.0, 1.0]|timing|headline"
    assert rec["traces"][1] == "setup|split_ops|compare:None|timing|headline"      # (the reference is rank 0's job)


def test_rank_1_check_raises_and_both_ranks_take_the_same_branch():
    """rank 1's check raises -> BOTH ranks leave the rank-check leg at that step (no rank goes on into the
    timing collectives alone), both run the headline, rank 0 relays one line naming the step and the rank."""
    rec, err = _selftest("1:compare")
    rc = rec["rank_check"]
    assert rc["ok"] is False and rc["step"] == "compare" and rc["failed_ranks"] == [1] and rc["phase"] == "rank_check"
    for tr in rec["traces"]:
        assert tr.endswith("|headline") and "timing" not in tr      # neither rank ran the step behind the failed one
    assert "injected failure" in err


def test_rank_0_reference_raises_before_any_split_operation():
    rec, _ = _selftest("0:reference")
    rc = rec["rank_check"]
    assert rc["ok"] is False and rc["step"] == "reference" and rc["failed_ranks"] == [0]
    assert rec["traces"][0] == "setup|headline" and rec["traces"][1] == "setup|headline"     # no rank started the split operators


def test_a_rank_that_dies_still_yields_one_line():
    """the child tree dies (rank 1 exits before the rendezvous) -> the parent prints ONE well-formed line with
    value null and the reason, instead of nothing; the surviving rank's rendezvous is bounded (TMLQCD_BENCH_PG_TIMEOUT_S)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=_env(TMLQCD_BENCH_RENDEZVOUS_ONLY="1", TMLQCD_BENCH_TEST_FAULT="die", TMLQCD_BENCH_PG_TIMEOUT_S="20"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["value"] is None and rec["n_ranks"] == 2 and "no result line" in rec["error"] and rec["metric"].startswith("Hopping_Matrix")
    assert rec["steps"] == 3 and rec["warmup"] == 1 and rec["wall_s"]["total"] < 200


def test_a_run_that_hangs_is_ended_and_reported():
    """... and a child tree that hangs is ended after TMLQCD_BENCH_TIMEOUT_S (whole process group) and reported the same way."""
    import time
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"],
                       env=_env(TMLQCD_BENCH_RENDEZVOUS_ONLY="1", TMLQCD_BENCH_TEST_FAULT="hang", TMLQCD_BENCH_TIMEOUT_S="25"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and time.perf_counter() - t0 < 120
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["value"] is None and "did not finish within 25 s" in rec["error"]


def _guardian_child(ending):
    """A stand-in for rank 0: hands the guardian a finished line, then ends the way `ending` says."""
    code = ("import os, sys; sys.path.insert(0, %r); import bench\n"
            "g = bench.Guardian(1)\n"
            "g.save({'metric': 'm', 'value': 5.0, 'faces': 'rccl'})\n"
            "%s\n") % (ROOT, ending)
    return subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120, start_new_session=True)


def test_guardian_prints_the_saved_line_when_rank_0_dies():
    """bench.py's second-carrier phase: the communicator's line is handed to a child process first; rank 0 aborting (a GPU fault) or being
    ended by the launcher (SIGTERM to its process group: another rank died) must leave exactly that line on stdout, marked as such."""
    for ending in ("os.abort()", "os.killpg(os.getpgrp(), 15)"):      # (torchrun signals a worker's whole process group)
        r = _guardian_child(ending)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert r.returncode != 0 and len(lines) == 1, (ending, r.stdout, r.stderr[-500:])
        rec = json.loads(lines[0])
        assert rec["value"] == 5.0 and rec["faces"] == "rccl" and rec["faces_direct"]["ok"] is False and "rank 0 ended" in rec["faces_direct"]["error"]


def test_guardian_stays_silent_when_rank_0_prints_its_own_line():
    r = _guardian_child("g.done(); print('{\"own\": 1}', flush=True)")
    assert r.returncode == 0 and [ln for ln in r.stdout.splitlines() if ln.startswith("{")] == ['{"own": 1}'], (r.stdout, r.stderr[-500:])
