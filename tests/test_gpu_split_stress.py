"""GPU: the split-phase (T-split rank) stencil under a long seeded random sequence of operations -- stencils with every twisted-mass
epilogue, compositions (chains), linalg kernels that rewrite fields between stencils, short cg_her solves, communication-free
stencils (whose output is stale by construction and overwritten afterwards), uploads, and a comm stream that is held back at random
points (a neighbour arriving late) -- against the SAME sequence on the unsplit lattice.  Every form of the split path (flags, HIP
events; the direct carrier's one- and two-kernel forms) and all three self-exchanges (copies, one-rank RCCL communicator, direct stores) must reproduce the unsplit fields:
what this guards is the bookkeeping between stencils (sequence numbers, which field's faces sit in the send buffers, which faces
were exchanged ahead), which no single-operation test exercises."""
import numpy as np
import pytest

from tests.util import random_gauge, random_spinor

pytestmark = pytest.mark.gpu

T, L = 8, 16          # faces are a quarter of the local volume; face % 64 == 0 (all forms apply)
NF = 5


def _run(lat, seed, nops, split, gen=None, skew=None):
    """The sequence on one lattice; returns the final fields and the scalars computed on the way.  gen(tag): this rank's slab of the
    global random field `tag` (several ranks: tests/mp_stress_worker.py); skew(step): called before every operation."""
    rng = np.random.default_rng(seed)
    N = lat.Vh
    if gen is None:
        def gen(tag):
            return random_spinor(tag, N)
    f = [lat.field(gen(100 + i)) for i in range(NF)]
    scal = []
    for step in range(nops):
        if skew:
            skew(step)
        op = int(rng.integers(0, 13))
        a, b, c = (int(x) for x in rng.choice(NF, size=3, replace=False))
        ieo = int(rng.integers(0, 2))
        z = complex(rng.uniform(-1, 1), rng.uniform(-1, 1))
        r = float(rng.uniform(-0.9, 0.9))
        delay = int(rng.integers(0, 12)) if rng.random() < 0.15 else 0
        if split and delay:
            lat.comm_stream_delay_ms(delay)                  # the neighbour's faces arrive late
        if op == 0:
            lat.Hopping_Matrix(ieo, f[a], f[b])
        elif op == 1:
            lat.tm_times_Hopping_Matrix(ieo, f[a], f[b], z)
        elif op == 2:
            lat.tm_sub_Hopping_Matrix(ieo, f[a], f[c], f[b], z)
        elif op == 3:
            lat.Qtm_pm_psi(f[a], f[b])
        elif op == 4:
            lat.op("Qtm_plus_psi", f[a], f[b])
        elif op == 5:
            lat.op("Qtm_minus_psi", f[a], f[a])                # in place (invert_eo.c:270)
        elif op == 6:
            lat.assign_add_mul_r(f[a], f[b], r, N)
        elif op == 7:
            lat.mul_r(f[a], 0.5 + abs(r), f[b], N)
        elif op == 8:
            scal.append(lat.square_norm(f[a], N, 1))
        elif op == 9:
            f[a].zero()
            it, _ = lat.cg_her(f[a], f[b], 6, 0.0, 1, N)       # exactly six iterations of the fused loop
            scal.append(float(it))
        elif op == 10:
            if split:
                lat.Hopping_Matrix_nocom(ieo, f[a], f[b])      # stale faces: not comparable, ...
            lat.Hopping_Matrix(ieo, f[a], f[b])                # ... overwritten at once; what it did to the buffers must not matter
        elif op == 11:
            f[a].upload(gen(1000 + step))                      # the host rewrites a field that may have been a stencil output
        else:
            lat.bench_hopping(f[a], f[b], f[c], 2)             # the benchmark loop: every second stencil gathers the previous output
        # keep the numbers O(1): the operators have norm < 1 but the axpys can grow
        if step % 7 == 6:
            for g in f:
                n = lat.square_norm(g, N, 1)               # (global sum: the same number on every rank)
                if n > 0:
                    lat.mul_r(g, 1.0 / np.sqrt(n / (N * lat.nproc_t)), g, N)
    lat.sync()
    out = [g.download() for g in f]
    for g in f:
        g.free()
    return out, scal


FORMS = [("flags", {}), ("events", {"split_sync": 1}), ("no prepack", {"prepack": 0})]
# loopback 3 = the direct carrier (faces stored by the producing waves into "the neighbour's" buffers): its own two forms
DIRECT_FORMS = [("one kernel", {"direct_form": 1}), ("one kernel, boundary last / first", {"direct_form": 1, "direct_order": 2}), ("one kernel, boundary first / last", {"direct_form": 1, "direct_order": 1}),
                ("stencil + exterior kernel", {"direct_form": 0})]
def forms_of(loopback):
    return DIRECT_FORMS if loopback == 3 else FORMS


@pytest.mark.parametrize("loopback", [1, 2, 3])
def test_random_operation_sequences_on_the_split_path(loopback):
    from tmlqcd_amd import Lattice
    kappa, mu, theta = 0.131, 0.017, (1.0, 0.2, 0.0, -0.3)
    g = random_gauge(77, T * L ** 3)
    for seed, nops in ((1, 70), (2, 70), (3, 90)):
        ref_lat = Lattice(T, L, L, L, kappa=kappa, mu=mu, theta=theta)
        ref_lat.set_gauge(g)
        ref, ref_scal = _run(ref_lat, seed, nops, split=False)
        ref_lat.close()
        for name, opts in forms_of(loopback):
            lat = Lattice(T, L, L, L, kappa=kappa, mu=mu, theta=theta)
            lat.set_gauge(g)
            for k, v in opts.items():
                lat.set_option(k, v)
            lat.set_loopback(loopback)
            got, scal = _run(lat, seed, nops, split=True)
            lat.close()
            for i, (x, y) in enumerate(zip(got, ref)):
                dev = np.abs(x - y).max() / max(np.abs(y).max(), 1e-300)
                assert dev < 1e-11, (loopback, name, seed, i, dev)      # ~70 operations deep: rounding differences of the split sums accumulate
            assert len(scal) == len(ref_scal)
            for s1, s2 in zip(scal, ref_scal):
                assert abs(s1 - s2) <= 1e-11 * max(abs(s2), 1.0), (loopback, name, seed)


@pytest.mark.parametrize("loopback", [1, 2, 3])
def test_fp32_clover_and_solvers_on_every_form_of_the_split_path(loopback):
    """The other precisions and epilogues a T-split rank runs -- the fp32 stencil and Qtm_pm_psi_32, the clover operators in both
    precisions, cg_her / mixed_cg_her on Qtm_pm_psi and Qsw_pm_psi (fused CG iterations: reductions spread over stencil and exterior
    kernel, faces of fp32 fields) -- on every form of the split path against the unsplit lattice."""
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn
    kappa, mu = 0.13, 0.02
    g = random_gauge(78, T * L ** 3)
    sw, swi = syn.clover_blocks(79, T, L, L, L, mu)
    N = T * L ** 3 // 2
    src, b = random_spinor(81, N), random_spinor(82, N)

    def run(lat):
        res = {}
        k, q, l, x = lat.field(src), lat.field(b), lat.field(), lat.field()
        k32, l32 = lat.field32(src.astype(np.float32)), lat.field32()
        for ieo in (0, 1):
            lat.Hopping_Matrix_32(ieo, l32, k32); res["hop32_%d" % ieo] = l32.download().astype(np.float64)
        lat.Qtm_pm_psi_32(l32, k32); res["qtm32"] = l32.download().astype(np.float64)
        lat.Qsw_pm_psi_32(l32, k32); res["qsw32"] = l32.download().astype(np.float64)
        lat.op("Qsw_pm_psi", l, k); res["qsw"] = l.download()
        lat.op("Qsw_minus_psi", l, k); res["qsw_minus"] = l.download()
        for name in ("Qtm_pm_psi", "Qsw_pm_psi"):
            x.zero(); it, _ = lat.cg_her(x, q, 2000, 1e-18, 1, N, op=name)
            res["cg_" + name] = (it, x.download())
            x.zero(); it, outer = lat.mixed_cg_her(x, q, 2000, 1e-18, 1, N, op=name)
            res["mixed_" + name] = (it, x.download())
        lat.sync()
        for fld in (k, q, l, x, k32, l32):
            fld.free()
        return res

    def make():
        lat = Lattice(T, L, L, L, kappa=kappa, mu=mu)
        lat.set_gauge(g)
        lat.set_clover(sw, swi)
        return lat

    ref_lat = make()
    ref = run(ref_lat)

    def true_residual(name, sol):
        """|A x - b|^2 / |b|^2 with A applied on the UNSPLIT lattice"""
        x, y = ref_lat.field(sol), ref_lat.field()
        ref_lat.op(name, y, x)
        r = y.download() - b
        x.free(); y.free()
        return float((r * r).sum() / (b * b).sum())

    def rel(x, y):
        return float(np.abs(x - y).max() / np.abs(y).max())
    for name, opts in forms_of(loopback):
        lat = make()
        for kk, v in opts.items():
            lat.set_option(kk, v)
        lat.set_loopback(loopback)
        got = run(lat)
        lat.close()
        for key, val in got.items():
            if isinstance(val, tuple):
                (it, sol), (it0, _) = val, ref[key]
                # the same solve, rounded differently (split sums; fp32 restarts): about as many iterations, and a solution of the system
                assert it > 0 and abs(it - it0) <= 2 + 0.03 * it0, (loopback, name, key, it, it0)
                rr = true_residual(key.split("_", 1)[1], sol)
                assert rr <= 4e-18, (loopback, name, key, rr)
            else:
                tol = 5e-6 if key.endswith("32") or "32_" in key else 1e-13
                assert rel(val, ref[key]) < tol, (loopback, name, key, rel(val, ref[key]))
    ref_lat.close()
