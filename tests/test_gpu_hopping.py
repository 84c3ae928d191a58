"""GPU parity: hand-written HIP stencil (through the C-ABI) vs the CPU oracle, same inputs."""
import numpy as np
import pytest

from tests.util import TOL, random_gauge, random_spinor, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup16():
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    T, L = 16, 16
    kappa, mu, theta = 0.137, 0.011, (1.0, 0.3, -0.2, 0.5)
    orc = Oracle(T, L, L, L, kappa=kappa, mu=mu, theta=theta, threads=8)
    lat = Lattice(T, L, L, L, kappa=kappa, mu=mu, theta=theta)
    g = random_gauge(11, orc.VPR)
    orc.set_gauge(g)
    lat.set_gauge(g)
    yield orc, lat
    lat.close()


@pytest.mark.parametrize("ieo", [0, 1])
def test_hopping_matrix_16(setup16, ieo):
    orc, lat = setup16
    N = orc.Vh
    k = random_spinor(5 + ieo, N)
    ref = orc.new_field()
    orc.Hopping_Matrix(ieo, ref, k)
    dk, dl = lat.field(k), lat.field()
    lat.Hopping_Matrix(ieo, dl, dk)
    out = dl.download()
    assert rel_err(out, ref[:N]) < TOL
    dk.free(); dl.free()


@pytest.mark.parametrize("block,xcd,minw,occ", [(64, 0, 0, 0), (64, 2, 4, 2), (256, 1, 0, 3), (256, 2, 4, 0), (256, 3, 0, 3), (64, 3, 0, 0), (256, 4, 0, 3)])
def test_kernel_variants_agree(setup16, block, xcd, minw, occ):
    orc, lat = setup16
    lat.set_option("hopsplit", 0)          # the one-thread-per-site kernels (the automatic choice at 16^4 is the hop-split kernel)
    N = orc.Vh
    k = random_spinor(9, N)
    ref = orc.new_field()
    orc.Hopping_Matrix(1, ref, k)
    defaults = dict(block=0, xcd=2, minw=0, occ=3)
    for name, val in dict(block=block, xcd=xcd, minw=minw, occ=occ).items():
        lat.set_option(name, val)
    dk, dl = lat.field(k), lat.field()
    lat.Hopping_Matrix(1, dl, dk)
    assert rel_err(dl.download(), ref[:N]) < TOL
    for name, val in defaults.items():
        lat.set_option(name, val)
    lat.set_option("hopsplit", -1)
    dk.free(); dl.free()


@pytest.mark.parametrize("ieo", [0, 1])
def test_fused_epilogues(setup16, ieo):
    orc, lat = setup16
    N = orc.Vh
    k, p = random_spinor(21, N), random_spinor(22, N)
    c = 0.83 - 0.41j
    ref = orc.new_field()
    dk, dp, dl = lat.field(k), lat.field(p), lat.field()
    orc.tm_times_Hopping_Matrix(ieo, ref, k, c)
    lat.tm_times_Hopping_Matrix(ieo, dl, dk, c)
    assert rel_err(dl.download(), ref[:N]) < TOL
    orc.tm_sub_Hopping_Matrix(ieo, ref, p, k, c)
    lat.tm_sub_Hopping_Matrix(ieo, dl, dp, dk, c)
    assert rel_err(dl.download(), ref[:N]) < TOL
    for f in (dk, dp, dl):
        f.free()


@pytest.mark.parametrize("mode,split_sync,prepack,form", [(1, 0, 1, 0), (2, 0, 1, 0), (1, 1, 1, 0), (2, 1, 1, 0), (1, 0, 0, 0), (2, 1, 0, 0),
                                                          (3, 0, 1, "one/0"), (3, 0, 1, "one/1"), (3, 0, 1, "one/2"), (3, 0, 1, "one/3"), (3, 0, 1, "two/0")])
def test_loopback_split_path_matches(setup16, mode, split_sync, prepack, form):
    """Single-GPU self-test of the multi-GPU code path: faces packed, exchanged with self, the stencil over all sites with the hop
    across the cut added by the exterior kernel -- behind a flag (split_sync 0, default) or behind a HIP event (1) -- must equal the
    plain periodic stencil.  prepack 1 (default): the stencils of a chain (Qtm_pm_psi below) take their faces from the previous
    stencil's exterior kernel instead of a pack kernel.  mode 3: the direct carrier onto oneself (the
    producing waves store the faces into the receiver's buffers): "one/<order>" = one kernel per stencil, boundary waves wait for
    their neighbour's word, with every dispatch order of the boundary slices; "two/0" = stencil + exterior kernel."""
    orc, lat = setup16
    N = orc.Vh
    k = random_spinor(31, N)
    ref = orc.new_field()
    dk, dl = lat.field(k), lat.field()
    lat.set_option("split_sync", split_sync)
    lat.set_option("prepack", prepack)
    if mode == 3:
        lat.set_option("direct_form", 1 if form.startswith("one") else 0)
        lat.set_option("direct_order", int(form.split("/")[1]))
    lat.set_loopback(mode)  # 1: D2D copies, 2: one-rank RCCL communicator (ncclSend/Recv to self), 3: direct stores into "the neighbour's" buffers
    try:
        for rep in range(3):       # repeated calls re-use the face buffers: a stale-cache bug would show here
            for ieo in (0, 1):
                kk = k * (1.0 + rep)
                dk.upload(kk)
                orc.Hopping_Matrix(ieo, ref, kk)
                lat.Hopping_Matrix(ieo, dl, dk)
                assert rel_err(dl.download(), ref[:N]) < TOL
        # every fused epilogue pushes the exterior kernel's one term through its own factors
        p = random_spinor(32, N)
        dp = lat.field(p)
        c = 0.83 - 0.41j
        for ieo in (0, 1):
            orc.tm_times_Hopping_Matrix(ieo, ref, k, c)
            dk.upload(k)
            lat.tm_times_Hopping_Matrix(ieo, dl, dk, c)
            assert rel_err(dl.download(), ref[:N]) < TOL
            orc.tm_sub_Hopping_Matrix(ieo, ref, p, k, c)
            lat.tm_sub_Hopping_Matrix(ieo, dl, dp, dk, c)
            assert rel_err(dl.download(), ref[:N]) < TOL
        dp.free()
        # a dependent chain without host round trips in between (what Qtm_pm_psi does)
        dk.upload(k)
        q = lat.field()
        lat.Qtm_pm_psi(q, dk)
        qref = orc.new_field(); orc.op("Qtm_pm_psi", qref, k.copy())
        assert rel_err(q.download(), qref[:N]) < TOL
        q.free()
        # the benchmark loop (benchmark.c:291-300): f1 = H_eo f0, f2 = H_oe f1, many times back to back -- every second stencil gathers
        # the output of the one before (direct carrier: its faces are pushed by the waves that complete its boundary slices)
        dk.upload(k)
        f1, f2 = lat.field(), lat.field()
        lat.bench_hopping(dk, f1, f2, 7)
        r1, r2 = orc.new_field(), orc.new_field()
        orc.Hopping_Matrix(0, r1, k); orc.Hopping_Matrix(1, r2, r1)
        assert rel_err(f1.download(), r1[:N]) < TOL and rel_err(f2.download(), r2[:N]) < TOL
        f1.free(); f2.free()
        # a chain whose promise would be wrong if the library took it on faith: the field between two stencils is rewritten by another
        # kernel (the twists of Mtm_plus_sym_dagg_psi) -- those stencils must not be treated as chained
        for name in ("Mtm_plus_sym_dagg_psi", "Qtm_plus_sym_psi", "Qtm_minus_psi"):
            dk.upload(k)
            lat.op(name, dl, dk)
            orc.op(name, ref, k.copy())
            assert rel_err(dl.download(), ref[:N]) < TOL, name
        lat.sync()
    finally:
        lat.set_loopback(0)
        lat.set_option("split_sync", 0)
        lat.set_option("prepack", 1)
        lat.set_option("direct_form", -1); lat.set_option("direct_order", 3)
    dk.free(); dl.free()


@pytest.mark.parametrize("mode,split_sync", [(1, 0), (2, 0), (2, 1)])
def test_split_path_waits_for_a_late_neighbour(setup16, mode, split_sync):
    """xchange_field's MPI_Waitall waits as long as the neighbour needs (xchange/xchange_field.c:98-250).  So does the split path:
    the comm stream is held back 6 s in front of the exchange (what a neighbour arriving late looks like from this rank) and the
    result is still the oracle's, with default settings; nothing is left behind for the next call."""
    import time
    orc, lat = setup16
    N = orc.Vh
    k = random_spinor(33, N)
    ref = orc.new_field()
    dk, dl = lat.field(k), lat.field()
    lat.set_option("split_sync", split_sync)
    lat.set_loopback(mode)
    try:
        lat.Hopping_Matrix(0, dl, dk)          # warm: buffers, (mode 2) the communicator
        lat.sync()
        t0 = time.time()
        lat.comm_stream_delay_ms(6000)
        lat.Hopping_Matrix(0, dl, dk)
        out = dl.download()
        dt = time.time() - t0
        orc.Hopping_Matrix(0, ref, k)
        assert dt > 5.5, dt                    # the call really sat behind the delay
        assert rel_err(out, ref[:N]) < TOL
        lat.sync()                             # no error is reported ...
        lat.Hopping_Matrix(1, dl, dk)          # ... and the next call is an ordinary one
        orc.Hopping_Matrix(1, ref, k)
        assert rel_err(dl.download(), ref[:N]) < TOL
        # the same inside a solver: the delay lands in front of one of the exchanges of cg_her
        q = random_spinor(34, N)
        P = orc.new_field()
        it_ref, _ = orc.cg_her(P, q.copy(), 500, 1e-18, 1, N)
        dq, dp = lat.field(q), lat.field()
        lat.comm_stream_delay_ms(5000)
        it, _ = lat.cg_her(dp, dq, 500, 1e-18, 1, N)
        assert abs(it - it_ref) <= 1 and rel_err(dp.download(), P[:N]) < 1e-9
        dq.free(); dp.free()
    finally:
        lat.set_loopback(0)
        lat.set_option("split_sync", 0)
    dk.free(); dl.free()


def test_split_path_deadline_is_reported_once_and_cleared(setup16):
    """A neighbour that never answers must not hang the GPU: with a 50 ms bound on the device-side wait and the comm stream held
    back 2 s, the call that synchronises reports the error -- and the NEXT call on the same context is an ordinary, correct one
    (round 2's error word was sticky: one late neighbour poisoned the context)."""
    import tmlqcd_amd
    orc, lat = setup16
    N = orc.Vh
    k = random_spinor(35, N)
    ref = orc.new_field()
    dk, dl = lat.field(k), lat.field()
    lat.set_loopback(1)
    try:
        lat.Hopping_Matrix(0, dl, dk)
        lat.sync()
        lat.set_option("flag_timeout_ms", 50)
        lat.comm_stream_delay_ms(2000)
        lat.Hopping_Matrix(0, dl, dk)
        with pytest.raises(tmlqcd_amd.hip.TmHipError):
            lat.sync()
        lat.set_option("flag_timeout_ms", 20000)
        for ieo in (0, 1):
            lat.Hopping_Matrix(ieo, dl, dk)
            orc.Hopping_Matrix(ieo, ref, k)
            assert rel_err(dl.download(), ref[:N]) < TOL
        lat.sync()
    finally:
        lat.set_option("flag_timeout_ms", 20000)
        lat.set_loopback(0)
    dk.free(); dl.free()


@pytest.mark.parametrize("dims", [(2, 2, 2, 2), (4, 2, 2, 2), (2, 4, 6, 2), (24, 4, 4, 4), (4, 4, 4, 16), (6, 10, 2, 4), (2, 8, 8, 4), (2, 8, 8, 8)])
def test_small_and_ragged_lattices(dims):
    """The shapes of the reference's own operator regression (hopping_test: L in 4..16, T in 4..24,
    test/hopping_test_generate_script:17-31) plus the smallest legal lattice; every direction wraps."""
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    T, LX, LY, LZ = dims
    theta = (1.0, 0.3, -0.2, 0.5)
    orc = Oracle(T, LX, LY, LZ, kappa=0.14, mu=0.03, theta=theta)
    lat = Lattice(T, LX, LY, LZ, kappa=0.14, mu=0.03, theta=theta)
    g = random_gauge(sum(dims), orc.VPR)
    orc.set_gauge(g); lat.set_gauge(g)
    N = orc.Vh
    for ieo in (0, 1):
        k = random_spinor(3 + ieo, N)
        ref = orc.new_field()
        orc.Hopping_Matrix(ieo, ref, k)
        dk, dl = lat.field(k), lat.field()
        lat.Hopping_Matrix(ieo, dl, dk)
        assert rel_err(dl.download(), ref[:N]) < TOL
        for loop in (1, 3):      # 3: the direct carrier -- faces that are not whole waves take its two-kernel form with the per-lane stencil; (2, 8, 8, .): T_local = 2, every site a boundary site, in its one-kernel form
            lat.set_loopback(loop)
            lat.Hopping_Matrix(ieo, dl, dk)
            lat.set_loopback(0)
            assert rel_err(dl.download(), ref[:N]) < TOL, loop
    q = random_spinor(9, N)
    ref = orc.new_field(); orc.op("Qtm_pm_psi", ref, q.copy())
    dq, dl = lat.field(q), lat.field()
    for loop in (0, 3):
        lat.set_loopback(loop)
        lat.Qtm_pm_psi(dl, dq)
        assert rel_err(dl.download(), ref[:N]) < TOL, loop
        assert abs(lat.square_norm(dl, N) - orc.square_norm(ref, N)) <= TOL * orc.square_norm(ref, N)
    lat.set_loopback(0)
    lat.close()


def test_gauge_recon_12_is_exact_for_su3_links_and_refused_otherwise(setup16):
    """Opt-in 12-real gauge read (COMPRESSION_12 of misc_types.h:29-33): third row rebuilt in registers.  Same results
    to TOL on SU(3) links -- plain stencil, all fused epilogues, Qtm_pm_psi, cg_her with the fused dot -- and the option
    is refused (full read stays) when the resident links are not unitary to rounding."""
    orc, lat = setup16
    N = orc.Vh
    assert lat.gauge_su3_deviation() < 1e-14
    k, p = random_spinor(31, N), random_spinor(32, N)
    dk, dp, dl = lat.field(k), lat.field(p), lat.field()
    ref = orc.new_field()
    lat.set_option("gauge_recon", 12)
    try:
        for ieo in (0, 1):
            orc.Hopping_Matrix(ieo, ref, k); lat.Hopping_Matrix(ieo, dl, dk)
            assert rel_err(dl.download(), ref[:N]) < TOL
            c = -0.37 + 0.91j
            orc.tm_times_Hopping_Matrix(ieo, ref, k, c); lat.tm_times_Hopping_Matrix(ieo, dl, dk, c)
            assert rel_err(dl.download(), ref[:N]) < TOL
            orc.tm_sub_Hopping_Matrix(ieo, ref, p, k, c); lat.tm_sub_Hopping_Matrix(ieo, dl, dp, dk, c)
            assert rel_err(dl.download(), ref[:N]) < TOL
        for name in ("Qtm_pm_psi", "Mtm_plus_psi"):
            orc.op(name, ref, k.copy()); lat.op(name, dl, dk)
            assert rel_err(dl.download(), ref[:N]) < TOL
        P = orc.new_field(); it_ref, _ = orc.cg_her(P, k.copy(), 2000, 1e-18, 1, N)
        dl.zero(); it, _ = lat.cg_her(dl, dk, 2000, 1e-18, 1, N)
        assert abs(it - it_ref) <= 1 and rel_err(dl.download(), P[:N]) < 1e-8
    finally:
        lat.set_option("gauge_recon", 18)
    # a field that is not SU(3): one link scaled by 1 + 1e-6
    g = orc._gauge.copy()
    g[7, 2] *= 1.0 + 1e-6
    orc.set_gauge(g); lat.set_gauge(g)
    try:
        assert lat.gauge_su3_deviation() > 1e-7
        lat.set_option("gauge_recon", 12)        # refused with a message on stderr: full 18-real read stays
        orc.Hopping_Matrix(0, ref, k); lat.Hopping_Matrix(0, dl, dk)
        assert rel_err(dl.download(), ref[:N]) < TOL
        orc.Hopping_Matrix(1, ref, k); lat.Hopping_Matrix(1, dl, dk)
        assert rel_err(dl.download(), ref[:N]) < TOL
    finally:
        lat.set_option("gauge_recon", 18)
        g0 = random_gauge(11, orc.VPR)
        orc.set_gauge(g0); lat.set_gauge(g0)
    for f in (dk, dp, dl):
        f.free()


def test_options_from_the_environment(monkeypatch):
    """TMLQCD_HIP_OPTIONS applies tmhip_set_option at context creation (unmodified executables); a malformed or unknown entry
    fails the creation instead of being ignored."""
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    T = L = 8
    for bad in ("bogus=1", "occ", "occ=two"):
        monkeypatch.setenv("TMLQCD_HIP_OPTIONS", bad)
        with pytest.raises(RuntimeError):
            Lattice(T, L, L, L)
    monkeypatch.setenv("TMLQCD_HIP_OPTIONS", "gauge_recon=12, occ=2,block=64")
    orc = Oracle(T, L, L, L, kappa=0.13, mu=0.02, threads=4)
    lat = Lattice(T, L, L, L, kappa=0.13, mu=0.02)
    monkeypatch.delenv("TMLQCD_HIP_OPTIONS")
    g = random_gauge(5, orc.VPR)
    orc.set_gauge(g); lat.set_gauge(g)
    N = orc.Vh
    k = random_spinor(6, N)
    ref = orc.new_field()
    dk, dl = lat.field(k), lat.field()
    for ieo in (0, 1):
        orc.Hopping_Matrix(ieo, ref, k); lat.Hopping_Matrix(ieo, dl, dk)
        assert rel_err(dl.download(), ref[:N]) < TOL
    lat.close()


def test_slab_block_order_with_uneven_slabs():
    """xcd = 3 (XCD j owns the j-th eighth of every time-slice): 9 blocks per slice do not split into 8 equal slabs, so
    the last slab is short and its spare block slots exit -- also in the fused stencil+dot launch of cg_her, whose
    partial sums then include zero-filled slots."""
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    T, LX, LY, LZ = 8, 12, 16, 24            # face = 2304 sites = 9 blocks of 256; 72 blocks per launch
    kappa, mu, theta = 0.13, 0.02, (1.0, 0.0, 0.5, 0.0)
    orc = Oracle(T, LX, LY, LZ, kappa=kappa, mu=mu, theta=theta, threads=8)
    lat = Lattice(T, LX, LY, LZ, kappa=kappa, mu=mu, theta=theta)
    g = random_gauge(41, orc.VPR)
    orc.set_gauge(g); lat.set_gauge(g)
    N = orc.Vh
    k = random_spinor(42, N)
    dk, dl = lat.field(k), lat.field()
    ref = orc.new_field()
    for xcd in (3, 2, 4, 1):
        lat.set_option("xcd", xcd)
        for ieo in (0, 1):
            orc.Hopping_Matrix(ieo, ref, k); lat.Hopping_Matrix(ieo, dl, dk)
            assert rel_err(dl.download(), ref[:N]) < TOL, (xcd, ieo)
        orc.op("Qtm_pm_psi", ref, k.copy()); lat.op("Qtm_pm_psi", dl, dk)
        assert rel_err(dl.download(), ref[:N]) < TOL, xcd
    lat.set_option("xcd", 3)
    P = orc.new_field(); it_ref, _ = orc.cg_her(P, k.copy(), 2000, 1e-18, 1, N)
    dl.zero(); it, _ = lat.cg_her(dl, dk, 2000, 1e-18, 1, N)
    assert abs(it - it_ref) <= 1 and rel_err(dl.download(), P[:N]) < 1e-8
    k32 = k.astype(np.float32)
    d32, l32 = lat.field32(k32), lat.field32()
    orc.Hopping_Matrix(1, ref, k32.astype(np.float64)); lat.Hopping_Matrix_32(1, l32, d32)     # 512-site blocks: 4.5 per slice -> tile order
    assert rel_err(l32.download().astype(np.float64), ref[:N]) < 2e-6
    lat.close()


@pytest.mark.parametrize("dims", [(16, 16, 16, 16), (4, 8, 8, 8), (6, 8, 16, 8), (4, 16, 8, 32), (8, 32, 32, 32)])
def test_lds_staged_stencil_matches_the_gather_stencil(dims):
    """"lds" 1: the block's own 256 input spinors are staged in LDS once and every neighbour that falls inside the block is read from
    there (+-z always, +-y except one edge row per wave).  Same operations in the same order as the gather kernel, for every epilogue, the fp32 twin, the
    split path and cg_her."""
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn
    T, LX, LY, LZ = dims
    kappa, mu, theta = 0.131, 0.02, (1.0, 0.0, 0.4, 0.0)
    lat = Lattice(T, LX, LY, LZ, kappa=kappa, mu=mu, theta=theta)
    g = syn.gauge_field(91, T, LX, LY, LZ)
    lat.set_gauge(g)
    lat.set_option("block", 256)
    N = lat.Vh
    k, p = syn.spinor_field_eo(92, 1, T, LX, LY, LZ), syn.spinor_field_eo(93, 0, T, LX, LY, LZ)
    dk, dp, dl = lat.field(k), lat.field(p), lat.field()
    k32 = lat.field32(k.astype(np.float32)); l32 = lat.field32()
    lat.mixed_cg_her(dl, dk, 1, 1e-2, 1, N)          # builds the fp32 gauge copy

    def run_all():
        out = []
        for ieo in (0, 1):
            lat.Hopping_Matrix(ieo, dl, dk); out.append(dl.download())
            lat.tm_times_Hopping_Matrix(ieo, dl, dk, 0.7 - 0.2j); out.append(dl.download())
            lat.tm_sub_Hopping_Matrix(ieo, dl, dp, dk, 0.7 - 0.2j); out.append(dl.download())
            lat.Hopping_Matrix_32(ieo, l32, k32); out.append(l32.download())
        lat.Qtm_pm_psi(dl, dk); out.append(dl.download())
        dl.zero()
        it, hist = lat.cg_her(dl, dk, 300, 1e-18, 1, N)
        out.append(dl.download()); out.append(np.array([it], dtype=np.float64)); out.append(hist.copy())
        return out
    for loop in (0, 1):
        if loop and (T < 4 or (LX * LY * LZ // 2) % 256):
            continue
        lat.set_loopback(loop)
        lat.set_option("lds", 0); lat.set_option("lds32", 0); a = run_all()
        for mode in (1,):
            lat.set_option("lds", mode); lat.set_option("lds32", mode); b = run_all()
            for x, y in zip(a[:-2], b[:-2]):         # same operations on the same values; only FMA contraction may differ between the two instantiations
                assert rel_err(y.astype(np.float64), x.astype(np.float64)) < (2e-6 if x.dtype == np.float32 else 1e-14), (dims, loop, mode)
            assert abs(a[-2][0] - b[-2][0]) <= 1 and np.allclose(a[-1][:-2], b[-1][:-2], rtol=1e-6), (dims, loop, mode)
    if dims == (16, 16, 16, 16):      # and against the oracle
        orc = Oracle(T, LX, LY, LZ, kappa=kappa, mu=mu, theta=theta, threads=8)
        orc.set_gauge(g)
        ref = orc.new_field(); orc.Hopping_Matrix(0, ref, k)
        lat.set_loopback(0); lat.Hopping_Matrix(0, dl, dk)
        assert rel_err(dl.download(), ref[:N]) < TOL
    lat.close()


@pytest.mark.parametrize("dims", [(8, 8, 8, 8), (6, 4, 4, 4), (4, 6, 2, 12), (16, 16, 16, 16), (2, 2, 2, 2)])
def test_hop_split_kernel_on_small_lattices(dims):
    """"hopsplit" 1: 64 sites per 256-thread block, wave w does the +-mu hops for mu = w, the partial spinors meet in LDS; every
    component-wise epilogue and the fused CG iteration (four partials per block), against the oracle and against hopsplit 0."""
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn
    T, LX, LY, LZ = dims
    kappa, mu, theta = 0.133, 0.015, (1.0, 0.2, 0.0, -0.3)
    lat = Lattice(T, LX, LY, LZ, kappa=kappa, mu=mu, theta=theta)
    orc = Oracle(T, LX, LY, LZ, kappa=kappa, mu=mu, theta=theta, threads=4)
    g = syn.gauge_field(101, T, LX, LY, LZ)
    lat.set_gauge(g); orc.set_gauge(g)
    N = lat.Vh
    k, p = syn.spinor_field_eo(102, 1, T, LX, LY, LZ), syn.spinor_field_eo(103, 0, T, LX, LY, LZ)
    dk, dp, dl = lat.field(k), lat.field(p), lat.field()
    c = 0.6 + 0.3j
    res = {}
    for hs in (1, 0):
        lat.set_option("hopsplit", hs)
        out = []
        for ieo in (0, 1):
            ref = orc.new_field()
            lat.Hopping_Matrix(ieo, dl, dk); orc.Hopping_Matrix(ieo, ref, k); out.append(dl.download())
            assert rel_err(out[-1], ref[:N]) < TOL, (hs, ieo)
            lat.tm_times_Hopping_Matrix(ieo, dl, dk, c); orc.tm_times_Hopping_Matrix(ieo, ref, k, c); out.append(dl.download())
            assert rel_err(out[-1], ref[:N]) < TOL, (hs, ieo)
            lat.tm_sub_Hopping_Matrix(ieo, dl, dp, dk, c); orc.tm_sub_Hopping_Matrix(ieo, ref, p, k, c); out.append(dl.download())
            assert rel_err(out[-1], ref[:N]) < TOL, (hs, ieo)
        dl.zero()
        it, hist = lat.cg_her(dl, dk, 500, 1e-20, 1, N)
        res[hs] = (out, it, hist.copy(), dl.download())
    P = orc.new_field()
    it_ref, hist_ref = orc.cg_her(P, k.copy(), 500, 1e-20, 1, N)
    for hs in (1, 0):
        assert abs(res[hs][1] - it_ref) <= 1, (hs, res[hs][1], it_ref)
        assert rel_err(res[hs][3], P[:N]) < 1e-9, hs
    lat.close()
