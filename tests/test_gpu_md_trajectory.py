"""GPU: a molecular-dynamics trajectory of the determinant monomial with everything resident in HBM -- the size-independent property
that ties the pieces of SURVEY section 8 f3 together: the force the device accumulates (det_derivative's statements,
monomial/det_monomial.c:58-97: cg_her on Qtm_pm_psi, Qtm_minus_psi, H_eo_tm_inv_psi, deriv_Sb twice) is the derivative of the action
phi^+ (Q_+ Q_-)^-1 phi (det_acc, :236-242) in exactly the normalisation update_momenta (update_momenta.c:67-72) and update_gauge
(update_gauge.c:51-110) integrate it with, so a leapfrog trajectory (integrator.c) conserves H = p^2 / 2 + S up to O(eps^2) and is
reversible.  Links, momenta, pseudofermion and derivative never leave the device between the first upload and the checks."""
import numpy as np
import pytest

from tmlqcd_amd import synthetic as syn

pytestmark = pytest.mark.gpu
EO, OE = 0, 1


class DetTrajectory:
    def __init__(self, L=8, kappa=0.125, mu=0.25, seed=5):
        from tmlqcd_amd import Lattice
        self.lat = lat = Lattice(L, L, L, L, kappa=kappa, mu=mu)
        self.g0 = syn.gauge_field(seed, L, L, L, L)
        rng = np.random.default_rng(seed + 1)
        self.p0 = rng.standard_normal((lat.V, 4, 8))
        # det_heatbath (:177-180): phi = Q_+ R, R Gaussian
        self.R = lat.field(syn.spinor_field_eo(seed + 2, 1, L, L, L, L))
        self.phi, self.X, self.Y, self.w2, self.w3 = (lat.field() for _ in range(5))
        self.reset()
        lat.op("Qtm_plus_psi", self.phi, self.R)
        self.iters = 0

    def reset(self):
        self.lat.set_gauge(self.g0)
        self.lat.momenta_upload(self.p0)

    def solve(self):
        lat = self.lat
        self.X.zero()
        it, _ = lat.cg_her(self.X, self.phi, 2000, 1e-26, 1, lat.Vh)
        assert it > 0
        self.iters += it
        lat.op("Qtm_minus_psi", self.Y, self.X)

    def energy(self):
        self.solve()                                                    # det_acc: S = |Q_- (Q_+ Q_-)^-1 phi|^2
        p = self.lat.momenta_download()
        return 0.5 * float((p * p).sum()) + self.lat.square_norm(self.Y, self.lat.Vh, 1)

    def force(self, step):
        lat = self.lat
        lat.derivative_zero()                                           # update_momenta.c:46-50
        self.solve()                                                    # X_o, Y_o
        lat.H_eo_tm_inv_psi(self.w2, self.X, EO, -1.0)                  # X_e
        lat.deriv_Sb(OE, self.Y, self.w2, 1.0)
        lat.H_eo_tm_inv_psi(self.w3, self.Y, EO, +1.0)                  # Y_e
        lat.deriv_Sb(EO, self.w3, self.X, 1.0)
        lat.update_momenta(step)                                        # p -= step * derivative

    def leapfrog(self, nsteps, eps):
        self.force(0.5 * eps)
        for k in range(nsteps):
            self.lat.update_gauge(eps)
            self.force(eps if k < nsteps - 1 else 0.5 * eps)

    def close(self):
        self.lat.close()


def test_leapfrog_conserves_the_hamiltonian_to_second_order_and_is_reversible():
    tr = DetTrajectory()
    h0 = tr.energy()
    dh = {}
    for nsteps in (4, 8):                                               # trajectory length 0.2
        tr.reset()
        tr.leapfrog(nsteps, 0.2 / nsteps)
        dh[nsteps] = tr.energy() - h0
    # forward, flip the momenta, back again: the links return (update_gauge is exp(eps p) U, the force a function of U only)
    p = tr.lat.momenta_download()
    tr.lat.momenta_upload(-p)
    tr.leapfrog(8, 0.2 / 8)
    back = tr.lat.gauge_download()[:tr.lat.V]
    pend = tr.lat.momenta_download()
    tr.close()
    print("H0 = %.6f   dH(eps = 0.05) = %.3e   dH(eps = 0.025) = %.3e   ratio %.2f   CG iterations %d" % (h0, dh[4], dh[8], dh[4] / dh[8], tr.iters))
    assert abs(dh[4]) < 2e-4 * abs(h0) and abs(dh[8]) < abs(dh[4])
    assert 3.0 < dh[4] / dh[8] < 5.5                                    # O(eps^2): a wrong force normalisation leaves dH = O(1) * (1 - c) dS
    assert np.abs(back - tr.g0).max() < 1e-10
    assert np.abs(pend + tr.p0).max() < 1e-9


class CloverDetTrajectory(DetTrajectory):
    """The clover determinant: cloverdet_derivative's statements (monomial/cloverdet_monomial.c:60-147) as the force, and as the
    action |Q_- (Q_+ Q_-)^-1 phi|^2 of cloverdet_acc plus the -tr log (1 + T_ee +- i mu g5) of the clover tr-log monomial
    (monomial/clover_trlog_monomial.c:79 = -sw_trace(EO, mu), operator/clover_det.c:115-170), whose force is the sw_deriv call."""

    def __init__(self, L=8, kappa=0.125, mu=0.25, c_sw=1.2, seed=7):
        from tmlqcd_amd import Lattice
        self.c_sw, self.kappa, self.mu = c_sw, kappa, mu
        self.lat = lat = Lattice(L, L, L, L, kappa=kappa, mu=mu)
        self.g0 = syn.gauge_field(seed, L, L, L, L)
        rng = np.random.default_rng(seed + 1)
        self.p0 = rng.standard_normal((lat.V, 4, 8))
        self.R = lat.field(syn.spinor_field_eo(seed + 2, 1, L, L, L, L))
        self.phi, self.X, self.Y, self.w2, self.w3 = (lat.field() for _ in range(5))
        self.reset()
        self.clover()
        lat.op("Qsw_plus_psi", self.phi, self.R)                        # cloverdet_heatbath
        self.iters = 0
        c = np.indices((L, L, L, L)).sum(axis=0).reshape(-1)
        self.even = (c & 1) == 0

    def clover(self):
        self.lat.sw_term(None, self.kappa, self.c_sw)                   # from the links resident in HBM
        self.lat.sw_invert(EO, self.mu)

    def solve(self):
        lat = self.lat
        self.clover()
        self.X.zero()
        it, _ = lat.cg_her(self.X, self.phi, 2000, 1e-26, 1, lat.Vh, op="Qsw_pm_psi")
        assert it > 0
        self.iters += it
        lat.op("Qsw_minus_psi", self.Y, self.X)

    def trlog(self):
        """-sw_trace(EO, mu): - sum over even sites and the two chiralities of log |det(A + i mu)|^2, A = [[sw0, sw1], [sw1^+, sw2]]."""
        sw, _ = self.lat.get_clover(True, False)
        b = (sw[..., 0] + 1j * sw[..., 1])[self.even]                   # [Vh][3][2][3][3]
        tot = 0.0
        for i in range(2):
            a = np.zeros((b.shape[0], 6, 6), dtype=complex)
            a[:, :3, :3] = b[:, 0, i]; a[:, :3, 3:] = b[:, 1, i]
            a[:, 3:, :3] = np.conj(np.transpose(b[:, 1, i], (0, 2, 1))); a[:, 3:, 3:] = b[:, 2, i]
            a += 1j * self.mu * np.eye(6)
            tot += 2.0 * np.linalg.slogdet(a)[1].sum()
        return -tot

    def energy(self):
        self.solve()
        p = self.lat.momenta_download()
        return 0.5 * float((p * p).sum()) + self.lat.square_norm(self.Y, self.lat.Vh, 1) + self.trlog()

    def force(self, step):
        lat = self.lat
        lat.derivative_zero()
        self.solve()                                                    # sw_term, sw_invert(EE), X_o, Y_o
        lat.H_eo_sw_inv_psi(self.w2, self.X, EO, -1, self.mu)           # X_e
        lat.deriv_Sb(OE, self.Y, self.w2, 1.0)
        lat.H_eo_sw_inv_psi(self.w3, self.Y, EO, +1, self.mu)           # Y_e
        lat.deriv_Sb(EO, self.w3, self.X, 1.0)
        lat.swpm_zero()
        lat.sw_spinor_eo(0, self.w2, self.w3, 1.0)                      # EE
        lat.sw_spinor_eo(1, self.Y, self.X, 1.0)                        # OO
        lat.sw_deriv(0, self.mu)                                        # the tr-log term, even sites
        lat.sw_all(self.kappa, self.c_sw)
        lat.update_momenta(step)


def test_clover_determinant_trajectory_conserves_its_hamiltonian():
    """... with sw_term / sw_invert recomputed from the moving links, sw_spinor_eo, sw_deriv and the owner-computes sw_all in the force."""
    tr = CloverDetTrajectory()
    h0 = tr.energy()
    dh = {}
    for nsteps in (4, 8):
        tr.reset()
        tr.leapfrog(nsteps, 0.2 / nsteps)
        dh[nsteps] = tr.energy() - h0
    p = tr.lat.momenta_download()
    tr.lat.momenta_upload(-p)
    tr.leapfrog(8, 0.2 / 8)
    back = tr.lat.gauge_download()[:tr.lat.V]
    tr.close()
    print("clover: H0 = %.6f   dH(eps = 0.05) = %.3e   dH(eps = 0.025) = %.3e   ratio %.2f   CG iterations %d" % (h0, dh[4], dh[8], dh[4] / dh[8], tr.iters))
    assert abs(dh[4]) < 5e-4 * abs(h0) and abs(dh[8]) < abs(dh[4])          # (measured: -24.4, -6.25 on H0 = 1.13e5; -1.57 at eps = 0.0125)
    assert 3.0 < dh[4] / dh[8] < 5.5
    assert np.abs(back - tr.g0).max() < 1e-10
