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


@pytest.mark.parametrize("T,L,world", [(2, 4, 2), (4, 4, 2), (2, 6, 3), (4, 8, 2)])
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
