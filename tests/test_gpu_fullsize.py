"""GPU: the BASELINE.json lattice sizes themselves.

configs[1] 32^4 and the unsplit configs[3] volume 32^3 x 64: the oracle is fast enough on the GPU box's host cores for a
full site-by-site comparison of the stencil and the fused operator; configs[4] 48^3 x 96 clover is checked site by site
too (device sw_term / sw_invert, Qsw_pm_psi and its fp32 twin against the oracle) and through size-independent properties (exact inverse of the clover block, hermiticity, Q_+ = Q_-^dagger, positivity, CG true residual,
fp32 vs fp64, 12-real gauge read vs full read)."""
import numpy as np
import pytest

from tests.util import TOL, rel_err

pytestmark = pytest.mark.gpu


def _mem_gb():
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable"):
            return int(line.split()[1]) / 1048576.0
    return 0.0


@pytest.mark.parametrize("T,L", [(32, 32), (64, 32)])
def test_full_size_site_by_site_against_oracle(T, L):
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn
    if _mem_gb() < 24:
        pytest.skip("needs ~20 GB of host memory for the oracle's gauge copy")
    kappa, mu, theta = 0.125, 0.01, (1.0, 0.0, 0.0, 0.0)
    orc = Oracle(T, L, L, L, kappa=kappa, mu=mu, theta=theta, threads=16)
    lat = Lattice(T, L, L, L, kappa=kappa, mu=mu, theta=theta)
    g = syn.gauge_field(31, T, L, L, L)
    orc.set_gauge(g); lat.set_gauge(g)
    del g
    N = orc.Vh
    k = syn.spinor_field_eo(32, 1, T, L, L, L)
    dk, dl = lat.field(k), lat.field()
    ref = orc.new_field()
    for ieo in (0, 1):
        orc.Hopping_Matrix(ieo, ref, k); lat.Hopping_Matrix(ieo, dl, dk)
        assert rel_err(dl.download(), ref[:N]) < TOL, ieo
    orc.op("Qtm_pm_psi", ref, k.copy()); lat.op("Qtm_pm_psi", dl, dk)
    out = dl.download()
    assert rel_err(out, ref[:N]) < TOL
    assert abs(lat.square_norm(dl, N) - orc.square_norm(ref, N)) <= 1e-13 * orc.square_norm(ref, N)
    # the opt-in 12-real gauge read and the fp32 twin at this size
    lat.set_option("gauge_recon", 12)
    lat.op("Qtm_pm_psi", dl, dk)
    lat.set_option("gauge_recon", 18)
    assert rel_err(dl.download(), ref[:N]) < TOL
    k32 = lat.field32(k.astype(np.float32)); l32 = lat.field32()
    lat.Qtm_pm_psi_32(l32, k32)
    assert rel_err(l32.download().astype(np.float64), ref[:N]) < 2e-5
    # solve to the BASELINE tolerance (|r|/|b| = 1e-10) and check the true residual on the CPU
    it, _ = lat.cg_her(dl, dk, 5000, 1e-20, 1, N)
    full = orc.new_field(); full[:N] = dl.download()
    orc.op("Qtm_pm_psi", ref, full)
    assert it > 0 and ((ref[:N] - k) ** 2).sum() / (k ** 2).sum() <= 1e-20
    lat.close()


def test_clover_48x96_properties():
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn
    if _mem_gb() < 40:
        pytest.skip("needs ~30 GB of host memory for the 48^3 x 96 gauge field and its temporaries")
    T, L = 96, 48
    kappa, mu, c_sw = 0.1394265, 0.0002, 1.69
    lat = Lattice(T, L, L, L, kappa=kappa, mu=mu)
    g = syn.gauge_field(41, T, L, L, L)
    lat.set_gauge(g)
    lat.sw_term(g, kappa, c_sw)
    lat.sw_invert(0, mu)
    del g
    N = lat.Vh
    x, y = lat.field(syn.spinor_field_eo(42, 1, T, L, L, L)), lat.field(syn.spinor_field_eo(43, 1, T, L, L, L))
    a, b, c = lat.field(), lat.field(), lat.field()

    def close(u, v, tol):
        lat.diff(c, u, v, N)
        return lat.square_norm(c, N) <= tol ** 2 * lat.square_norm(v, N)

    # (1 + T + i mu g5) sw_inv = 1 on the even sites: the device-computed inverse really inverts the device-computed term
    lat.assign_mul_one_sw_pm_imu_inv(0, a, x, mu); lat.assign_mul_one_sw_pm_imu(0, b, a, mu)
    assert close(b, x, 1e-12)
    # Q_+ = Q_-^dagger, Qsw_pm = Q_+ Q_- hermitian and positive
    lat.op("Qsw_plus_psi", a, x); lat.op("Qsw_minus_psi", b, y)
    s1, s2 = lat.scalar_prod_r(y, a, N), lat.scalar_prod_r(b, x, N)
    assert abs(s1 - s2) <= 1e-11 * max(abs(s1), np.sqrt(lat.square_norm(a, N) * lat.square_norm(y, N)) * 1e-3)
    lat.op("Qsw_pm_psi", a, x); lat.op("Qsw_pm_psi", b, y)
    h1, h2 = lat.scalar_prod_r(y, a, N), lat.scalar_prod_r(b, x, N)
    assert abs(h1 - h2) <= 1e-11 * abs(h1) and lat.scalar_prod_r(x, a, N) > 0
    # Q_+ Q_- composed from its halves
    lat.op("Qsw_minus_psi", b, x); lat.op("Qsw_plus_psi", c, b)
    lat.assign(b, c, N)
    assert close(b, a, 1e-12)
    # fp32 twin
    x32, a32 = lat.field32(x.download().astype(np.float32)), lat.field32()
    lat.Qsw_pm_psi_32(a32, x32)
    lat.assign_to_64(b, a32, N)
    assert close(b, a, 2e-5)
    # solvers: true residual in fp64 on the device
    for name, run in (("cg_her", lambda: (y.zero(), lat.cg_her(y, x, 20000, 1e-20, 1, N, op="Qsw_pm_psi")[0])[1]),
                      ("mixed_cg_her", lambda: lat.mixed_cg_her(y, x, 20000, 1e-20, 1, N, op="Qsw_pm_psi")[0]),
                      ("rg_mixed_cg_her", lambda: lat.rg_mixed_cg_her(y, x, 20000, 1e-20, 1, N, op="Qsw_pm_psi")[0])):
        assert run() > 0, name
        lat.op("Qsw_pm_psi", b, y); lat.diff(b, x, b, N)
        assert lat.square_norm(b, N) <= 1e-20 * lat.square_norm(x, N), name
    lat.close()


def test_clover_48x96_site_by_site_against_oracle():
    """BASELINE configs[4] at its full size against the oracle, site by site: sw_term and sw_invert computed on the device vs the
    OpenMP restatement of operator/clover_term.c:88 / clover_invert.c:170 (itself bit-exact against the reference objects on the
    fixtures), then one Qsw_pm_psi (operator/clovertm_operators.c:233-245) and one Qsw_pm_psi_32 application.  The arrays are
    compared one at a time (sw 9.2 GB, sw_inv 12.2 GB each side) to keep the host footprint near 50 GB."""
    import gc
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn
    if _mem_gb() < 72:
        pytest.skip("needs ~55 GB of host memory: links, the oracle's gauge copy, sw and sw_inv of both sides")
    T, L = 96, 48
    kappa, mu, c_sw = 0.1394265, 0.0002, 1.69
    orc = Oracle(T, L, L, L, kappa=kappa, mu=mu, threads=16)
    lat = Lattice(T, L, L, L, kappa=kappa, mu=mu)
    g = syn.gauge_field(41, T, L, L, L)
    orc.set_gauge(g); lat.set_gauge(g)
    lat.sw_term(None, kappa, c_sw)               # from the links resident on the device
    lat.sw_invert(0, mu)
    del g
    sw_ref = orc.sw_term(kappa, c_sw)
    sw, _ = lat.get_clover(want_sw=True, want_sw_inv=False)
    assert rel_err(sw, sw_ref) < TOL
    del sw
    gc.collect()
    swi_ref, fails = orc.sw_invert(sw_ref, 0, mu)
    assert fails == 0
    _, swi = lat.get_clover(want_sw=False, want_sw_inv=True)
    assert rel_err(swi, swi_ref) < TOL           # both sets: mu != 0
    del swi
    gc.collect()
    orc.set_clover(sw_ref, swi_ref)
    N = orc.Vh
    k = syn.spinor_field_eo(42, 1, T, L, L, L)
    ref = orc.new_field()
    orc.op("Qsw_pm_psi", ref, k.copy())
    dk, dl = lat.field(k), lat.field()
    lat.op("Qsw_pm_psi", dl, dk)
    assert rel_err(dl.download(), ref[:N]) < TOL
    k32, l32 = lat.field32(k.astype(np.float32)), lat.field32()
    lat.Qsw_pm_psi_32(l32, k32)
    assert rel_err(l32.download().astype(np.float64), ref[:N]) < 2e-5
    lat.close()


def test_t_split_32x64_over_two_contexts_full_size():
    """configs[3] volume (32^3 x 64) cut in T over two contexts of this process (32 time-slices each: halo gauge links, global
    parity offset, face pack -> peer copy -> boundary kernels, slab/tile block order on the 30 interior slices) == the unsplit
    lattice on the oracle, site by site."""
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn
    from tmlqcd_amd.hip import multi_Hopping_Matrix
    if _mem_gb() < 24:
        pytest.skip("needs ~20 GB of host memory for the oracle's gauge copy")
    T, L, world = 32, 32, 2
    Tg = T * world
    kappa, theta = 0.125, (1.0, 0.0, 0.0, 0.0)
    g = Oracle(Tg, L, L, L, kappa=kappa, theta=theta, threads=16)
    g.set_gauge(syn.gauge_field(51, Tg, L, L, L))
    lats = [Lattice(T, L, L, L, kappa=kappa, theta=theta, nproc_t=world, proc_t=r) for r in range(world)]
    for r, lat in enumerate(lats):
        lat.set_gauge(syn.gauge_field(51, T, L, L, L, world, r))
    Vh = lats[0].Vh
    for ieo in (0, 1):
        kg = g.new_field(); kg[:g.Vh] = syn.spinor_field_eo(52, 1 - ieo, Tg, L, L, L)
        ref = g.new_field()
        g.Hopping_Matrix(ieo, ref, kg)
        ks = [lat.field(syn.spinor_field_eo(52, 1 - ieo, T, L, L, L, world, r)) for r, lat in enumerate(lats)]
        ls = [lat.field() for lat in lats]
        multi_Hopping_Matrix(lats, ieo, ls, ks)
        for r in range(world):
            assert rel_err(ls[r].download(), ref[r * Vh:(r + 1) * Vh]) < TOL, (ieo, r)
        for f in ks + ls:
            f.free()
    for lat in lats:
        lat.close()


def test_direct_carrier_with_48_cubed_faces():
    """One rank's share of BASELINE configs[4] (48^3 x 96 over 8 GPUs: 12 x 48^3) over the direct carrier onto itself: 1728 boundary
    waves per launch -- the largest faces the one-kernel form takes (twisted-mass kernels; the clover epilogues fall back to stencil +
    exterior kernel at this size) -- against the same library's unsplit lattice, which is pinned against the oracle above: the plain
    stencil, a chain, the benchmark loop (faces pushed ahead), the fused CG and the clover operator."""
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn
    T, L = 12, 48
    kappa, mu, c_sw = 0.125, 0.01, 1.5
    gauge = syn.gauge_field(61, T, L, L, L)
    k0, k1 = syn.spinor_field_eo(62, 0, T, L, L, L), syn.spinor_field_eo(63, 1, T, L, L, L)
    out = {}
    for split in (False, True):
        lat = Lattice(T, L, L, L, kappa=kappa, mu=mu)
        lat.set_gauge(gauge)
        if split:
            lat.set_loopback(3)
        lat.sw_term(gauge, kappa, c_sw); lat.sw_invert(0, mu)
        d0, d1, a, b = lat.field(k0), lat.field(k1), lat.field(), lat.field()
        res = []
        lat.Hopping_Matrix(0, a, d1); res.append(a.download())
        lat.Hopping_Matrix(1, b, d0); res.append(b.download())
        lat.Qtm_pm_psi(a, d0); res.append(a.download())
        lat.bench_hopping(d1, a, b, 5); res.append(a.download()); res.append(b.download())
        lat.op("Qsw_pm_psi", a, d0); res.append(a.download())
        x = lat.field(); x.zero()
        it, hist = lat.cg_her(x, d0, 12, 1e-30, 1, lat.Vh)
        res.append(x.download())
        out[split] = (res, hist)
        lat.close()
    for i, (u, s) in enumerate(zip(out[False][0], out[True][0])):
        assert rel_err(s, u) < TOL, i
    assert np.allclose(out[True][1], out[False][1], rtol=1e-10)
