"""GPU: hopping part of the fermion force, deriv_Sb (deriv_Sb.c:401-700), against the reference's
4^4 fixture and the CPU oracle (which is bit-exact against the reference, tests/test_oracle_vs_ref.py)."""
import ctypes as C
import os

import numpy as np
import pytest

from tests.util import TOL, random_gauge, random_spinor, rel_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_deriv_Sb_fixture_from_reference():
    from tmlqcd_amd import Lattice
    f = np.load(os.path.join(GOLD, "ref_fields_4x4.npz"))
    want = np.load(os.path.join(GOLD, "ref_force_4x4.npz"))["derivative"]
    lat = Lattice(4, 4, 4, 4, kappa=0.125, mu=0.01)
    lat.set_gauge(np.ascontiguousarray(f["gauge"]))
    a, b = lat.field(np.ascontiguousarray(f["in"])), lat.field(np.ascontiguousarray(f["Heo"]))
    lat.derivative_zero()
    lat.deriv_Sb(1, a, b, 0.5)
    lat.deriv_Sb(0, b, a, -0.25)
    assert rel_err(lat.derivative(), want) < TOL
    lat.close()


@pytest.mark.parametrize("dims", [(8, 6, 4, 12), (4, 2, 6, 2), (2, 2, 2, 2)])
def test_deriv_Sb_against_oracle(dims):
    """Ragged extents incl. the 2-site wrap (x+mu == x-mu), twisted boundary phases in every direction, both parities,
    accumulation over calls, and the accumulate / overwrite download modes."""
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    T, LX, LY, LZ = dims
    kappa, mu, theta = 0.131, 0.02, (1.0, 0.5, -0.25, 0.125)
    orc = Oracle(T, LX, LY, LZ, kappa=kappa, mu=mu, theta=theta, threads=8)
    lat = Lattice(T, LX, LY, LZ, kappa=kappa, mu=mu, theta=theta)
    g = random_gauge(91, orc.VPR)
    orc.set_gauge(g); lat.set_gauge(g)
    N = orc.Vh
    l, k = random_spinor(92, N), random_spinor(93, N)
    lo, ko = orc.new_field(), orc.new_field(); lo[:N] = l; ko[:N] = k
    dl, dk = lat.field(l), lat.field(k)
    df = np.zeros((orc.VPR, 4, 8))
    lat.derivative_zero()
    for ieo, fac in ((0, 0.7), (1, -1.3), (0, 0.11)):
        orc.deriv_Sb(ieo, lo, ko, df, fac)
        lat.deriv_Sb(ieo, dl, dk, fac)
    got = lat.derivative()
    assert rel_err(got, df[:orc.V]) < TOL
    base = np.random.default_rng(94).standard_normal((orc.V, 4, 8))      # other monomials' forces already on the host
    acc = base.copy()
    lat.derivative(into=acc)
    assert rel_err(acc, base + df[:orc.V]) < TOL
    assert np.array_equal(dl.download(), l) and np.array_equal(dk.download(), k)   # inputs untouched
    lat.derivative_zero()
    assert not lat.derivative().any()
    lat.close()


@pytest.mark.parametrize("world", [2, 3])
def test_deriv_Sb_split_path_loopback_and_two_t_slabs(world):
    """T-split ranks: the +t neighbours of the last time-slice come from the exchanged t=0 slices of BOTH fields
    (xchange_2fields, deriv_Sb.c:102).  (1) one rank with the exchange looped back onto itself, (2) two contexts holding
    the two halves of the lattice (halo gauge links, global parity offset, peer copies) == the unsplit lattice."""
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn
    from tmlqcd_amd.hip import multi_deriv_Sb
    T, L = 2, 4
    Tg = T * world
    kappa, mu, theta = 0.13, 0.02, (1.0, 0.25, 0.0, 0.5)
    g = Oracle(Tg, L, L, L, kappa=kappa, mu=mu, theta=theta, threads=4)
    g.set_gauge(syn.gauge_field(6, Tg, L, L, L))
    Ng = g.Vh
    lg, kg = g.new_field(), g.new_field()
    # l lives on parity ieo, k on the other one; use ieo = 1 then 0 with the roles swapped, like det_derivative
    lg[:Ng] = syn.spinor_field_eo(7, 1, Tg, L, L, L); kg[:Ng] = syn.spinor_field_eo(8, 0, Tg, L, L, L)
    ref = np.zeros((g.VPR, 4, 8))
    g.deriv_Sb(1, lg, kg, ref, 0.6)
    g.deriv_Sb(0, kg, lg, ref, -0.3)
    # (1) loopback on the unsplit lattice
    one = Lattice(Tg, L, L, L, kappa=kappa, mu=mu, theta=theta)
    one.set_gauge(syn.gauge_field(6, Tg, L, L, L))
    dl, dk = one.field(np.ascontiguousarray(lg[:Ng])), one.field(np.ascontiguousarray(kg[:Ng]))
    one.set_loopback(1)
    one.derivative_zero()
    one.deriv_Sb(1, dl, dk, 0.6); one.deriv_Sb(0, dk, dl, -0.3)
    one.set_loopback(0)
    assert rel_err(one.derivative(), ref[:g.V]) < TOL
    one.close()
    # (2) two T-slabs
    lats = [Lattice(T, L, L, L, kappa=kappa, mu=mu, theta=theta, nproc_t=world, proc_t=r) for r in range(world)]
    for r, lat in enumerate(lats):
        lat.set_gauge(syn.gauge_field(6, T, L, L, L, world, r))
        lat.derivative_zero()
    ls = [lat.field(syn.spinor_field_eo(7, 1, T, L, L, L, world, r)) for r, lat in enumerate(lats)]
    ks = [lat.field(syn.spinor_field_eo(8, 0, T, L, L, L, world, r)) for r, lat in enumerate(lats)]
    multi_deriv_Sb(lats, 1, ls, ks, 0.6)
    multi_deriv_Sb(lats, 0, ks, ls, -0.3)
    V = lats[0].V
    for r, lat in enumerate(lats):
        assert rel_err(lat.derivative(), ref[r * V:(r + 1) * V]) < TOL, r
        lat.close()


def test_deriv_Sb_drop_in_symbol(host_stub):
    """deriv_Sb under its reference name: host AoS spinors, hamiltonian_field_t by pointer, contribution ADDED to
    hf->derivative (coherent mode) or held back until tmlqcd_hip_flush_derivative (resident mode)."""
    from oracle.oraclebind import Oracle
    stub, d = host_stub
    VP = C.c_void_p
    T, L = 4, 6
    kappa, mu, theta = 0.127, 0.01, (1.0, 0.0, 0.0, 0.0)
    V = T * L ** 3
    N = V // 2
    gptr = stub.stub_init(T, L, L, L)
    g = random_gauge(95, V)
    C.memmove(gptr, g.ctypes.data_as(VP), g.nbytes)
    stub.stub_boundary(kappa, *theta)
    stub.stub_set_mu(mu)
    orc = Oracle(T, L, L, L, kappa=kappa, mu=mu, theta=theta, threads=4)
    orc.set_gauge(g)

    class HF(C.Structure):        # hamiltonian_field.h:26-32
        _fields_ = [("gaugefield", VP), ("momenta", VP), ("derivative", VP), ("update_gauge_copy", C.c_int), ("traj_counter", C.c_int)]
    df_host = np.random.default_rng(96).standard_normal((V, 4, 8))
    start = df_host.copy()
    rows = (VP * V)(*[df_host.ctypes.data + 4 * 8 * 8 * i for i in range(V)])      # su3adj **derivative
    hf = HF(None, None, C.cast(rows, VP), 0, 0)
    d.deriv_Sb.argtypes = [C.c_int, VP, VP, C.POINTER(HF), C.c_double]
    d.deriv_Sb.restype = None
    d.tmlqcd_hip_flush_derivative.argtypes = [C.POINTER(HF)]
    d.tmlqcd_hip_set_residency.argtypes = [C.c_int]
    l, k = random_spinor(97, N), random_spinor(98, N)
    lo, ko = orc.new_field(), orc.new_field(); lo[:N] = l; ko[:N] = k
    ref = np.zeros((orc.VPR, 4, 8))
    d.deriv_Sb(1, l.ctypes.data_as(VP), k.ctypes.data_as(VP), C.byref(hf), 0.9)
    orc.deriv_Sb(1, lo, ko, ref, 0.9)
    assert rel_err(df_host, start + ref[:V]) < TOL
    d.tmlqcd_hip_set_residency(1)
    d.deriv_Sb(0, k.ctypes.data_as(VP), l.ctypes.data_as(VP), C.byref(hf), -0.4)
    d.deriv_Sb(1, l.ctypes.data_as(VP), k.ctypes.data_as(VP), C.byref(hf), 0.2)
    assert rel_err(df_host, start + ref[:V]) < TOL                      # nothing flushed yet
    orc.deriv_Sb(0, ko, lo, ref, -0.4); orc.deriv_Sb(1, lo, ko, ref, 0.2)
    d.tmlqcd_hip_flush_derivative(C.byref(hf))
    assert rel_err(df_host, start + ref[:V]) < TOL
    d.tmlqcd_hip_set_residency(0)
    d.tmlqcd_hip_finalize()


@pytest.mark.parametrize("dims,mu", [((4, 4, 6, 4), 0.02), ((2, 2, 2, 2), 0.0), ((8, 6, 4, 12), 0.3)])
def test_clover_force_chain_against_oracle(dims, mu):
    """The clover part of cloverdet_derivative (monomial/cloverdet_monomial.c:110-147) on the device: sw_spinor_eo on both
    parities, sw_deriv(EE, mu), then sw_all adding into the same derivative field deriv_Sb uses -- against the oracle, which
    is bit-exact against operator/clover_deriv.c and operator/clover_accumulate_deriv.c (tests/test_oracle_vs_ref.py)."""
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    T, LX, LY, LZ = dims
    kappa, c_sw, theta = 0.131, 1.37, (1.0, 0.5, -0.25, 0.125)
    orc = Oracle(T, LX, LY, LZ, kappa=kappa, mu=mu, theta=theta, threads=8)
    lat = Lattice(T, LX, LY, LZ, kappa=kappa, mu=mu, theta=theta)
    g = random_gauge(201, orc.VPR)
    orc.set_gauge(g); lat.set_gauge(g)
    sw = orc.sw_term(kappa, c_sw); swi, _ = orc.sw_invert(sw, 0, mu)
    orc.set_clover(sw, swi)
    lat.sw_term(g, kappa, c_sw); lat.sw_invert(0, mu)
    N, V = orc.Vh, orc.V
    fh = [random_spinor(210 + i, N) for i in range(4)]
    fo = []
    for a in fh:
        b = orc.new_field(); b[:N] = a; fo.append(b)
    fd = [lat.field(a) for a in fh]
    swm, swp = np.zeros((V, 4, 3, 3, 2)), np.zeros((V, 4, 3, 3, 2))
    orc.sw_spinor_eo(0, swm, swp, fo[2], fo[3], 0.7)
    orc.sw_spinor_eo(1, swm, swp, fo[0], fo[1], 0.7)
    orc.sw_deriv(0, swm, swp, mu)
    lat.swpm_zero()
    lat.sw_spinor_eo(0, fd[2], fd[3], 0.7)
    lat.sw_spinor_eo(1, fd[0], fd[1], 0.7)
    lat.sw_deriv(0, mu)
    gm, gp = lat.get_swpm()
    assert rel_err(gm, swm) < TOL and rel_err(gp, swp) < TOL
    df = np.zeros((orc.VPR, 4, 8))
    orc.deriv_Sb(1, fo[0], fo[2], df, 0.7)            # the hopping part goes into the same accumulator first
    orc.sw_all(df, swm, swp, kappa, c_sw)
    lat.derivative_zero()
    lat.deriv_Sb(1, fd[0], fd[2], 0.7)
    lat.sw_all(kappa, c_sw)                           # lexicographic links kept from sw_term
    assert rel_err(lat.derivative(), df[:V]) < 4 * TOL     # atomics: the sum order of the <= 16 contributions per link is not fixed
    lat.derivative_zero()
    lat.sw_all(kappa, c_sw, gauge=g)                  # ... or handed over again
    ref2 = np.zeros((orc.VPR, 4, 8)); orc.sw_all(ref2, swm, swp, kappa, c_sw)
    assert rel_err(lat.derivative(), ref2[:V]) < 4 * TOL
    lat.close()


def test_clover_force_and_clover_term_block_orders():
    """The block orders of the two plaquette-leaf kernels ("swall_order" 0 chunk / 1 slab / 2 tile (default) / 8 wider tiles, "swterm_order"
    0 / 1) on a lattice whose shape admits all of them (a 64-site block = whole z-rows of one (t, x) row): the order only changes
    which XCD computes a block, so the results must be bit-identical to each other -- and agree with the oracle."""
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    T, LX, LY, LZ = 4, 16, 8, 16
    kappa, c_sw, mu, theta = 0.131, 1.37, 0.02, (1.0, 0.5, -0.25, 0.125)
    orc = Oracle(T, LX, LY, LZ, kappa=kappa, mu=mu, theta=theta, threads=8)
    lat = Lattice(T, LX, LY, LZ, kappa=kappa, mu=mu, theta=theta)
    g = random_gauge(301, orc.VPR)
    orc.set_gauge(g); lat.set_gauge(g)
    sw = orc.sw_term(kappa, c_sw); swi, _ = orc.sw_invert(sw, 0, mu)
    orc.set_clover(sw, swi)
    got = []
    for order in (0, 1):
        lat.set_option("swterm_order", order)
        lat.sw_term(g, kappa, c_sw)
        got.append(lat.get_clover(True, False)[0])
    assert np.array_equal(got[0], got[1]) and rel_err(got[1], sw) < TOL
    lat.sw_invert(0, mu)
    N, V = orc.Vh, orc.V
    fh = [random_spinor(310 + i, N) for i in range(4)]
    fo = []
    for a in fh:
        b = orc.new_field(); b[:N] = a; fo.append(b)
    fd = [lat.field(a) for a in fh]
    swm, swp = np.zeros((V, 4, 3, 3, 2)), np.zeros((V, 4, 3, 3, 2))
    orc.sw_spinor_eo(0, swm, swp, fo[2], fo[3], 0.7); orc.sw_spinor_eo(1, swm, swp, fo[0], fo[1], 0.7); orc.sw_deriv(0, swm, swp, mu)
    lat.swpm_zero()
    lat.sw_spinor_eo(0, fd[2], fd[3], 0.7); lat.sw_spinor_eo(1, fd[0], fd[1], 0.7); lat.sw_deriv(0, mu)
    ref = np.zeros((orc.VPR, 4, 8)); orc.sw_all(ref, swm, swp, kappa, c_sw)
    first = None
    for order in (0, 1, 2, 8):
        lat.set_option("swall_order", order)
        lat.derivative_zero()
        lat.sw_all(kappa, c_sw)
        d = lat.derivative()
        assert rel_err(d, ref[:V]) < 4 * TOL, order
        if first is None:
            first = d
        assert np.array_equal(d, first), order
    with pytest.raises(Exception):
        lat.set_option("swall_order", 3)
    lat.close()


@pytest.mark.parametrize("T,L,world", [(2, 4, 2), (4, 4, 2), (2, 6, 3), (4, 8, 2), (4, 16, 2)])      # (4, 16, 2): the interior launch runs in tile order
def test_clover_force_on_two_t_slabs(T, L, world):
    """sw_all on T-split ranks: the leaves next to the t-faces reach links of BOTH ring neighbours (the two-sided derivative halo
    of xchange_deri.c).  Two contexts holding the two halves of the lattice (own sw_term / sw_invert from halo links, site-local
    sw_spinor_eo / sw_deriv, contributions exchanged by peer copies) == the unsplit oracle, together with deriv_Sb."""
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn
    from tmlqcd_amd.hip import multi_deriv_Sb, multi_sw_all
    Tg = T * world                                                # world = 3: the two ring neighbours are different ranks
    kappa, mu, c_sw, theta = 0.13, 0.02, 1.4, (1.0, 0.25, 0.0, 0.5)
    g = Oracle(Tg, L, L, L, kappa=kappa, mu=mu, theta=theta, threads=4)
    g.set_gauge(syn.gauge_field(16, Tg, L, L, L))
    sw = g.sw_term(kappa, c_sw); swi, _ = g.sw_invert(sw, 0, mu)
    g.set_clover(sw, swi)
    Ng, Vg = g.Vh, g.V
    fo = []
    for i, par in enumerate((1, 1, 0, 0)):                    # fields 0,1 on odd sites, 2,3 on even sites
        b = g.new_field(); b[:Ng] = syn.spinor_field_eo(20 + i, par, Tg, L, L, L); fo.append(b)
    swm, swp = np.zeros((Vg, 4, 3, 3, 2)), np.zeros((Vg, 4, 3, 3, 2))
    g.sw_spinor_eo(0, swm, swp, fo[2], fo[3], 0.7)
    g.sw_spinor_eo(1, swm, swp, fo[0], fo[1], 0.7)
    g.sw_deriv(0, swm, swp, mu)
    ref = np.zeros((g.VPR, 4, 8))
    g.deriv_Sb(1, fo[0], fo[2], ref, 0.7)
    g.sw_all(ref, swm, swp, kappa, c_sw)
    lats = [Lattice(T, L, L, L, kappa=kappa, mu=mu, theta=theta, nproc_t=world, proc_t=r) for r in range(world)]
    fd = []
    for r, lat in enumerate(lats):
        gr = syn.gauge_field(16, T, L, L, L, world, r)
        lat.set_gauge(gr)
        lat.sw_term(gr, kappa, c_sw); lat.sw_invert(0, mu)
        fr = [lat.field(syn.spinor_field_eo(20 + i, par, T, L, L, L, world, r)) for i, par in enumerate((1, 1, 0, 0))]
        fd.append(fr)
        lat.swpm_zero()
        lat.sw_spinor_eo(0, fr[2], fr[3], 0.7)
        lat.sw_spinor_eo(1, fr[0], fr[1], 0.7)
        lat.sw_deriv(0, mu)
        gm, gp = lat.get_swpm()
        V = lat.V
        assert rel_err(gm, swm[r * V:(r + 1) * V]) < TOL and rel_err(gp, swp[r * V:(r + 1) * V]) < TOL, r
        lat.derivative_zero()
    multi_deriv_Sb(lats, 1, [f[0] for f in fd], [f[2] for f in fd], 0.7)
    multi_sw_all(lats, kappa, c_sw)
    V = lats[0].V
    for r, lat in enumerate(lats):
        assert rel_err(lat.derivative(), ref[r * V:(r + 1) * V]) < 4 * TOL, r
        lat.close()


def test_clover_force_through_the_drop_in(host_stub):
    """cloverdet_derivative's clover statements through the drop-in helpers on host arrays: the contribution lands in
    hf->derivative next to deriv_Sb's (coherent mode)."""
    from oracle.oraclebind import Oracle
    stub, d = host_stub
    VP = C.c_void_p
    T, L = 4, 4
    kappa, mu, c_sw, theta = 0.127, 0.01, 1.5, (1.0, 0.0, 0.0, 0.0)
    V = T * L ** 3
    N = V // 2
    gptr = stub.stub_init(T, L, L, L)
    g = random_gauge(301, V)
    C.memmove(gptr, g.ctypes.data_as(VP), g.nbytes)
    stub.stub_boundary(kappa, *theta)
    stub.stub_set_mu(mu)
    stub.stub_init_clover.restype = VP
    stub.stub_init_clover.argtypes = [C.c_int]
    stub.stub_init_clover(0)
    orc = Oracle(T, L, L, L, kappa=kappa, mu=mu, theta=theta, threads=4)
    orc.set_gauge(g)
    sw = orc.sw_term(kappa, c_sw); swi, _ = orc.sw_invert(sw, 0, mu)
    orc.set_clover(sw, swi)

    class HF(C.Structure):
        _fields_ = [("gaugefield", VP), ("momenta", VP), ("derivative", VP), ("update_gauge_copy", C.c_int), ("traj_counter", C.c_int)]
    df_host = np.zeros((V, 4, 8))
    drows = (VP * V)(*[df_host.ctypes.data + 4 * 8 * 8 * i for i in range(V)])
    grows = (VP * V)(*[gptr + 4 * 144 * i for i in range(V)])
    hf = HF(C.cast(grows, VP), None, C.cast(drows, VP), 0, 0)
    d.tmlqcd_hip_sw_term.argtypes = [C.c_double, C.c_double]
    d.tmlqcd_hip_sw_invert.argtypes = [C.c_int, C.c_double]
    d.tmlqcd_hip_sw_spinor_eo.argtypes = [C.c_int, VP, VP, C.c_double]
    d.tmlqcd_hip_sw_deriv.argtypes = [C.c_int, C.c_double]
    d.tmlqcd_hip_sw_all.argtypes = [C.POINTER(HF), C.c_double, C.c_double]
    d.deriv_Sb.argtypes = [C.c_int, VP, VP, C.POINTER(HF), C.c_double]
    d.tmlqcd_hip_sw_term(kappa, c_sw)
    d.tmlqcd_hip_sw_invert(0, mu)
    w = [random_spinor(310 + i, N) for i in range(4)]
    wo = []
    for a in w:
        b = orc.new_field(); b[:N] = a; wo.append(b)
    p = lambda a: a.ctypes.data_as(VP)
    d.tmlqcd_hip_swpm_zero()                                       # cloverdet_monomial.c:67-72
    d.deriv_Sb(1, p(w[0]), p(w[2]), C.byref(hf), 0.8)              # :115
    d.deriv_Sb(0, p(w[3]), p(w[1]), C.byref(hf), 0.8)              # :121
    d.tmlqcd_hip_sw_spinor_eo(0, p(w[2]), p(w[3]), 0.8)            # :125
    d.tmlqcd_hip_sw_spinor_eo(1, p(w[0]), p(w[1]), 0.8)            # :128
    d.tmlqcd_hip_sw_deriv(0, mu)                                   # :134
    d.tmlqcd_hip_sw_all(C.byref(hf), kappa, c_sw)                  # :147
    ref = np.zeros((orc.VPR, 4, 8))
    swm, swp = np.zeros((V, 4, 3, 3, 2)), np.zeros((V, 4, 3, 3, 2))
    orc.deriv_Sb(1, wo[0], wo[2], ref, 0.8); orc.deriv_Sb(0, wo[3], wo[1], ref, 0.8)
    orc.sw_spinor_eo(0, swm, swp, wo[2], wo[3], 0.8); orc.sw_spinor_eo(1, swm, swp, wo[0], wo[1], 0.8)
    orc.sw_deriv(0, swm, swp, mu)
    orc.sw_all(ref, swm, swp, kappa, c_sw)
    assert rel_err(df_host, ref[:V]) < 4 * TOL
    d.tmlqcd_hip_finalize()


def test_update_gauge_and_momenta_on_the_device_against_reference_fixture():
    """update_gauge.c:51-110 with the links resident in HBM: fixture = the reference's own update_gauge (+ exposu3 / restoresu3,
    expo.c) run on the RANLUX gauge field of ref_fields_4x4.npz with seeded Gaussian momenta (oracle/make_golden.py md);
    the oracle restatement is bit-exact against it (tests/test_oracle_vs_ref.py), the GPU within fp64 tolerance.  Then the
    stencil must see the new links (gauge copy re-sorted on the device), and update_momenta must match update_momenta.c:67-72."""
    import os
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    f = np.load(os.path.join(gold, "ref_fields_4x4.npz"))
    m = np.load(os.path.join(gold, "ref_md_4x4.npz"))
    g0, mom, step = np.ascontiguousarray(f["gauge"]), np.ascontiguousarray(m["momenta"]), float(m["step"])
    lat = Lattice(4, 4, 4, 4, kappa=0.125, mu=0.01)
    orc = Oracle(4, 4, 4, 4, kappa=0.125, mu=0.01)
    lat.set_gauge(g0)
    lat.momenta_upload(mom)
    lat.update_gauge(step)
    g1 = lat.gauge_download()
    assert rel_err(g1, m["gauge_after"]) < TOL
    og = g0.copy(); orc.update_gauge(og, mom, step)
    assert np.array_equal(og, m["gauge_after"])                                   # (CPU) restatement == reference, bit for bit
    # the stencil works on the updated links without a new upload
    orc.set_gauge(m["gauge_after"])
    k = np.ascontiguousarray(f["in"])
    ref = orc.new_field(); orc.Hopping_Matrix(0, ref, k)
    dk, dl = lat.field(k), lat.field()
    lat.Hopping_Matrix(0, dl, dk)
    assert rel_err(dl.download(), ref[:lat.Vh]) < TOL
    # a second step, links still unitary
    lat.update_gauge(step)
    orc.update_gauge(og, mom, step)
    g2 = lat.gauge_download()
    assert rel_err(g2, og) < TOL
    u = g2[..., 0] + 1j * g2[..., 1]
    assert np.abs(np.einsum("nmij,nmkj->nmik", u, u.conj()) - np.eye(3)).max() < 1e-14
    # momenta: P -= step * derivative with the derivative accumulated on the device by deriv_Sb
    l = np.ascontiguousarray(f["Heo"])
    dl2 = lat.field(l)
    lat.derivative_zero()
    lat.deriv_Sb(0, dl2, dk, 0.7)
    df = lat.derivative()
    lat.update_momenta(0.05)
    mom2 = lat.momenta_download()
    exp = mom.copy(); orc.update_momenta(exp, df, 0.05)
    assert rel_err(mom2, exp) < TOL
    lat.close()


def test_update_gauge_at_scale_keeps_the_links_unitary_and_the_clover_term_follows():
    """16^4: the device update against the oracle, and sw_term recomputed from the RESIDENT links (gauge = None)."""
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn
    T = L = 16
    lat = Lattice(T, L, L, L, kappa=0.13, mu=0.02)
    orc = Oracle(T, L, L, L, kappa=0.13, mu=0.02, threads=8)
    g = syn.gauge_field(95, T, L, L, L)
    mom = np.random.default_rng(96).standard_normal((lat.V, 4, 8))
    lat.set_gauge(g); lat.momenta_upload(mom)
    for _ in range(3):
        lat.update_gauge(0.02); orc.update_gauge(g, mom, 0.02)
    assert rel_err(lat.gauge_download(), g) < TOL
    orc.set_gauge(g)
    lat.sw_term(None, 0.13, 1.7)
    sw_dev, _ = lat.get_clover(True, False)
    assert rel_err(sw_dev, orc.sw_term(0.13, 1.7)) < TOL
    lat.close()


@pytest.mark.parametrize("T,L,world", [(2, 4, 2), (4, 4, 3)])
def test_update_gauge_on_t_slabs_refreshes_the_halo_links(T, L, world):
    """update_gauge on T-split ranks (contexts of one process, peer copies): the updated t = 0 / T-1 slices reach the neighbours' halo
    slabs and the re-sorted stencil copies use them -- links and a stencil application afterwards == the unsplit oracle, slab by slab."""
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn
    from tmlqcd_amd.hip import multi_Hopping_Matrix, multi_update_gauge
    Tg = T * world
    kappa, mu, theta = 0.13, 0.02, (1.0, 0.25, 0.0, 0.5)
    orc = Oracle(Tg, L, L, L, kappa=kappa, mu=mu, theta=theta, threads=4)
    g = syn.gauge_field(16, Tg, L, L, L)
    rng = np.random.default_rng(77)
    mom = rng.standard_normal((Tg * L ** 3, 4, 8))
    V = T * L ** 3
    lats = [Lattice(T, L, L, L, kappa=kappa, mu=mu, theta=theta, nproc_t=world, proc_t=r) for r in range(world)]
    for r, lat in enumerate(lats):
        lat.set_gauge(syn.gauge_field(16, T, L, L, L, world, r))
        lat.momenta_upload(np.ascontiguousarray(mom[r * V:(r + 1) * V]))
    for step in (0.05, -0.02):
        multi_update_gauge(lats, step)
        orc.update_gauge(g, mom, step)
    orc.set_gauge(g)
    XYZ = L ** 3
    for r, lat in enumerate(lats):
        got = lat.gauge_download()
        assert rel_err(got[:V], g[r * V:(r + 1) * V]) < TOL
        up, dn = (r + 1) % world, (r - 1) % world
        assert np.array_equal(got[V:V + XYZ], lats[up].gauge_download()[:XYZ])                     # t = T slab = the up neighbour's t = 0
        assert np.array_equal(got[V + XYZ:V + 2 * XYZ], lats[dn].gauge_download()[V - XYZ:V])      # t = -1 slab = the down neighbour's t = T-1
    k = orc.new_field(); k[:orc.Vh] = syn.spinor_field_eo(21, 1, Tg, L, L, L)
    ref = orc.new_field(); orc.Hopping_Matrix(0, ref, k)
    ks = [lat.field(syn.spinor_field_eo(21, 1, T, L, L, L, world, r)) for r, lat in enumerate(lats)]
    ls = [lat.field() for lat in lats]
    multi_Hopping_Matrix(lats, 0, ls, ks)
    Vh = V // 2
    for r, lat in enumerate(lats):
        assert rel_err(ls[r].download(), ref[r * Vh:(r + 1) * Vh]) < TOL, r
        lat.close()


def test_update_gauge_drop_in_keeps_the_links_in_hbm(host_stub):
    """tmlqcd_hip_update_gauge(step, hf): coherent mode = the reference's update_gauge as the host sees it (links updated in
    hf->gaugefield, flags raised) with no second upload; resident mode = the host links stay behind until
    tmlqcd_hip_sync_gauge_to_host, while the stencil already works on the new ones."""
    from oracle.oraclebind import Oracle
    stub, d = host_stub
    VP = C.c_void_p
    T, L = 4, 6
    kappa, mu = 0.127, 0.01
    V = T * L ** 3
    N = V // 2
    gptr = stub.stub_init(T, L, L, L)
    g = random_gauge(195, V)
    C.memmove(gptr, g.ctypes.data_as(VP), g.nbytes)
    stub.stub_boundary(kappa, 1.0, 0.0, 0.0, 0.0)
    stub.stub_set_mu(mu)
    orc = Oracle(T, L, L, L, kappa=kappa, mu=mu, theta=(1.0, 0.0, 0.0, 0.0), threads=4)
    host_links = np.frombuffer((C.c_double * (V * 72)).from_address(gptr), dtype=np.float64).reshape(V, 4, 3, 3, 2)

    class HF(C.Structure):        # hamiltonian_field.h:26-32
        _fields_ = [("gaugefield", VP), ("momenta", VP), ("derivative", VP), ("update_gauge_copy", C.c_int), ("traj_counter", C.c_int)]
    mom = np.random.default_rng(196).standard_normal((V, 4, 8))
    grows = (VP * V)(*[gptr + 4 * 144 * i for i in range(V)])                       # su3 **gaugefield
    mrows = (VP * V)(*[mom.ctypes.data + 4 * 8 * 8 * i for i in range(V)])          # su3adj **momenta
    hf = HF(C.cast(grows, VP), C.cast(mrows, VP), None, 0, 0)
    for f in ("tmlqcd_hip_update_gauge",):
        getattr(d, f).argtypes = [C.c_double, C.POINTER(HF)]; getattr(d, f).restype = None
    d.tmlqcd_hip_sync_gauge_to_host.argtypes = [C.POINTER(HF)]
    d.tmlqcd_hip_set_residency.argtypes = [C.c_int]
    d.Hopping_Matrix.argtypes = [C.c_int, VP, VP]
    k = random_spinor(197, N); l = np.zeros_like(k)

    def check_stencil(links):
        orc.set_gauge(links)
        ref = orc.new_field(); orc.Hopping_Matrix(0, ref, k)
        d.Hopping_Matrix(0, l.ctypes.data_as(VP), k.ctypes.data_as(VP))
        assert rel_err(l, ref[:N]) < TOL
    want = g.copy()
    # coherent mode
    d.tmlqcd_hip_update_gauge(0.03, C.byref(hf))
    orc.update_gauge(want, mom, 0.03)
    # host and device links are in step and the flags are DOWN (the host's backward copy, if the program has one, was refreshed
    # right away): a flag raised from here on is the host program's own and forces an upload
    assert rel_err(host_links, want) < TOL and hf.update_gauge_copy == 0 and stub.stub_gauge_flag() == 0
    check_stencil(want)
    # the reject step restores the old links on the host and raises the flag (update_tm.c) -- the next operator call
    # must run on THOSE links, not on the ones the device update left in HBM
    host_links[:] = g
    stub.stub_mark_gauge_dirty()
    check_stencil(g)
    assert stub.stub_gauge_flag() == 0                                              # consumed by the stencil call, like Hopping_Matrix.c:135-139
    host_links[:] = want
    stub.stub_mark_gauge_dirty()
    check_stencil(want)
    # resident mode: two steps without the host seeing anything
    d.tmlqcd_hip_set_residency(1)
    before = host_links.copy()
    d.tmlqcd_hip_update_gauge(0.03, C.byref(hf)); d.tmlqcd_hip_update_gauge(-0.01, C.byref(hf))
    orc.update_gauge(want, mom, 0.03); orc.update_gauge(want, mom, -0.01)
    assert np.array_equal(host_links, before)
    d.tmlqcd_hip_set_residency(0)
    check_stencil(want)                                                             # the device is ahead of the host
    d.tmlqcd_hip_sync_gauge_to_host(C.byref(hf))
    assert rel_err(host_links, want) < TOL
    d.tmlqcd_hip_finalize()
