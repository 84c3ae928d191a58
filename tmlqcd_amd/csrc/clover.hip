// Clover twisted mass on top of the stencil (invert_clover_eo.c:63-165).
//
// The site-local 6x6 blocks are INPUTS, exactly like the gauge field: the host computes
//   sw     = 1 + T            su3 sw[VOLUME][3][2]      (sw_term,   operator/clover_term.c:88-200)
//   sw_inv = (1+T+-i mu g5)^-1 su3 sw_inv[VOLUME][4][2]  (sw_invert, operator/clover_invert.c:170-257; even sites,
//                                                        +mu in [0,V/2), -mu in [V/2,V))
// and tmhip_set_clover() re-sorts them into SoA device arrays.  clover_inv / clover_gamma5 / clover are fused into the
// stencil epilogues (EPI_CLOVER_*), so Qsw_pm_psi = 4 launches; stand-alone site kernels serve the drop-in symbols.
#include "tmhip_internal.h"

// sw[ix][a][b] (lexicographic) -> swd[par][2a+b][e][i]
__global__ __launch_bounds__(256) void sw_sort_kernel(const v2d *__restrict__ raw, v2d *__restrict__ d, int gs, int Vh, int LX, int LY,
                                                      int LZ, int toff) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Vh) return;
  const int par = blockIdx.y;
  const int LZh = LZ / 2;
  int r = i / LZh;
  const int y = r % LY;
  r /= LY;
  const int x = r % LX, t = r / LX;
  const int o = (t + x + y + toff + par) & 1;
  const size_t ix = 2 * (size_t)i + o;
  const v2d *src = raw + ix * 54;
  v2d *dst = d + (size_t)par * 54 * gs + i;
#pragma unroll 6
  for (int e = 0; e < 54; e++) dst[(size_t)e * gs] = src[e];
}
// sw_inv[icy][a][b], icy = e/o index of the even site (+ V/2 for the -mu set) -> swinv[sign][2a+b][e][i]
__global__ __launch_bounds__(256) void swinv_sort_kernel(const v2d *__restrict__ raw, v2d *__restrict__ d, int gs, int Vh) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Vh) return;
  const int sign = blockIdx.y;
  const v2d *src = raw + ((size_t)sign * Vh + i) * 72;
  v2d *dst = d + (size_t)sign * 72 * gs + i;
#pragma unroll 6
  for (int e = 0; e < 72; e++) dst[(size_t)e * gs] = src[e];
}
__global__ void clover_to32_kernel(v2f *__restrict__ d, const v2d *__restrict__ s, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const v2d a = s[i]; d[i] = v2f{(float)a.x, (float)a.y}; }
}

__device__ __forceinline__ v2d cl_cfma(v2d a, v2d b, v2d c) { return v2d{c.x + a.x * b.x - a.y * b.y, c.y + a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ v2d cl_cfmac(v2d a, v2d b, v2d c) { return v2d{c.x + a.x * b.x + a.y * b.y, c.y + a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ void cl_block(v2d (&r)[3], const v2d *__restrict__ w, size_t gs, int i, int blk, const v2d *s, bool dagger, bool acc) {
  v2d u[9];
#pragma unroll
  for (int e = 0; e < 9; e++) u[e] = w[((size_t)blk * 9 + e) * gs + i];
#pragma unroll
  for (int row = 0; row < 3; row++) {
    v2d t = acc ? r[row] : v2d{0.0, 0.0};
    if (dagger) t = cl_cfmac(u[6 + row], s[2], cl_cfmac(u[3 + row], s[1], cl_cfmac(u[row], s[0], t)));
    else t = cl_cfma(u[3 * row + 2], s[2], cl_cfma(u[3 * row + 1], s[1], cl_cfma(u[3 * row], s[0], t)));
    r[row] = t;
  }
}

// MODE 0: l = W_inv l (clover_inv, in place)   1: l = g5((1+T+i mu g5) k - j)   2: same without g5
template <int MODE>
__global__ __launch_bounds__(256) void clover_site_kernel(v2d *L, const v2d *K, const v2d *J, const v2d *__restrict__ w, int ns, int gs,
                                                          int N, double mu) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
#pragma unroll
  for (int b = 0; b < 2; b++) {
    v2d sa[3], sb[3], r1[3], r2[3];
#pragma unroll
    for (int c = 0; c < 3; c++) { sa[c] = K[(size_t)(6 * b + c) * ns + i]; sb[c] = K[(size_t)(6 * b + 3 + c) * ns + i]; }
    if (MODE == 0) {
      cl_block(r1, w, gs, i, 0 * 2 + b, sa, false, false); cl_block(r1, w, gs, i, 1 * 2 + b, sb, false, true);
      cl_block(r2, w, gs, i, 3 * 2 + b, sa, false, false); cl_block(r2, w, gs, i, 2 * 2 + b, sb, false, true);
#pragma unroll
      for (int c = 0; c < 3; c++) { L[(size_t)(6 * b + c) * ns + i] = r1[c]; L[(size_t)(6 * b + 3 + c) * ns + i] = r2[c]; }
    } else {
      cl_block(r1, w, gs, i, 0 * 2 + b, sa, false, false); cl_block(r1, w, gs, i, 1 * 2 + b, sb, false, true);
      cl_block(r2, w, gs, i, 1 * 2 + b, sa, true, false);  cl_block(r2, w, gs, i, 2 * 2 + b, sb, false, true);
      const double m = b == 0 ? mu : -mu;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        r1[c] = v2d{r1[c].x - m * sa[c].y, r1[c].y + m * sa[c].x};
        r2[c] = v2d{r2[c].x - m * sb[c].y, r2[c].y + m * sb[c].x};
        const v2d zero = v2d{0.0, 0.0};
        const v2d j1 = J ? J[(size_t)(6 * b + c) * ns + i] : zero, j2 = J ? J[(size_t)(6 * b + 3 + c) * ns + i] : zero;
        const bool flip = MODE == 1 && b == 1;
        L[(size_t)(6 * b + c) * ns + i] = flip ? j1 - r1[c] : r1[c] - j1;
        L[(size_t)(6 * b + 3 + c) * ns + i] = flip ? j2 - r2[c] : r2[c] - j2;
      }
    }
  }
}


// ------------------------------------------------------------------ sw_term / sw_invert on the device
// (operator/clover_term.c:88-200, operator/clover_invert.c:88-257).  Once per gauge configuration (or per MD step
// in the HMC), so these favour clarity over the last GB/s: the leaf kernel walks the raw lexicographic gauge field
// [VPR][4][9] exactly as the reference indexes g_gauge_field, halo slabs included, which makes T-split ranks work
// without an edge exchange (only +-1 steps in two different directions are needed).
struct M3 { v2d e[9]; };
__device__ __forceinline__ v2d m3_cmul(v2d a, v2d b) { return v2d{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ v2d m3_conj(v2d a) { return v2d{a.x, -a.y}; }
__device__ __forceinline__ M3 m3_load(const v2d *__restrict__ raw, int ix, int mu) {
  M3 r;
  const v2d *p = raw + ((size_t)ix * 4 + mu) * 9;
#pragma unroll
  for (int e = 0; e < 9; e++) r.e[e] = p[e];
  return r;
}
// op(a) op(b), op = dagger when the flag is set
template <bool AD, bool BD>
__device__ __forceinline__ M3 m3_mul(const M3 &a, const M3 &b) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      v2d acc = v2d{0.0, 0.0};
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const v2d x = AD ? m3_conj(a.e[3 * k + i]) : a.e[3 * i + k];
        const v2d y = BD ? m3_conj(b.e[3 * j + k]) : b.e[3 * k + j];
        acc += m3_cmul(x, y);
      }
      r.e[3 * i + j] = acc;
    }
  return r;
}
__device__ __forceinline__ void m3_acc(M3 &a, const M3 &b) {
#pragma unroll
  for (int e = 0; e < 9; e++) a.e[e] += b.e[e];
}

struct LexGeom { int T, LX, LY, LZ, V, split; };
// geometry_eo.c:279-299: Index() of the PARALLELT / serial layouts; c[0] may be -1 or T
__device__ __forceinline__ int lex_index(const LexGeom &g, int t, int x, int y, int z) {
  x = (x + g.LX) % g.LX; y = (y + g.LY) % g.LY; z = (z + g.LZ) % g.LZ;
  const int sp = (x * g.LY + y) * g.LZ + z, XYZ = g.LX * g.LY * g.LZ;
  if (g.split) {
    if (t == g.T) return g.V + sp;
    if (t == -1) return g.V + XYZ + sp;
  }
  t = (t + g.T) % g.T;
  return t * XYZ + sp;
}

// clover_invert.c:88-160: 6x6 complex inverse by Householder triangularisation without pivoting, inversion of the
// triangle in place, then the reflections from the right in reverse order.  Fully unrolled => `a` lives in registers.
__device__ __forceinline__ int six_invert_dev(v2d (&a)[6][6]) {
  const double tiny = 1.0e-20;   // clover_leaf.c:55
  v2d d[6], u[6];
  double p[6];
  int fail = 0;
#pragma unroll
  for (int k = 0; k < 5; k++) {
    double s = 0.0;
#pragma unroll
    for (int j = k + 1; j < 6; j++) s += a[j][k].x * a[j][k].x + a[j][k].y * a[j][k].y;
    s = sqrt(1.0 + s / (a[k][k].x * a[k][k].x + a[k][k].y * a[k][k].y));
    const v2d sigma = v2d{s * a[k][k].x, s * a[k][k].y};
    a[k][k] += sigma;
    p[k] = sigma.x * a[k][k].x + sigma.y * a[k][k].y;
    const double q = sigma.x * sigma.x + sigma.y * sigma.y;
    if (q < tiny) fail++;
    d[k] = v2d{-sigma.x / q, sigma.y / q};
#pragma unroll
    for (int j = k + 1; j < 6; j++) {
      v2d z = v2d{0.0, 0.0};
#pragma unroll
      for (int i = k; i < 6; i++) z += m3_cmul(m3_conj(a[i][k]), a[i][j]);
      z = v2d{z.x / p[k], z.y / p[k]};
#pragma unroll
      for (int i = k; i < 6; i++) a[i][j] -= m3_cmul(z, a[i][k]);
    }
  }
  {
    const v2d sigma = a[5][5];
    const double q = sigma.x * sigma.x + sigma.y * sigma.y;
    if (q < tiny) fail++;
    d[5] = v2d{sigma.x / q, -sigma.y / q};
  }
#pragma unroll
  for (int k = 5; k >= 0; k--)
#pragma unroll
    for (int i = k - 1; i >= 0; i--) {
      v2d z = v2d{0.0, 0.0};
#pragma unroll
      for (int j = i + 1; j < k; j++) z += m3_cmul(a[i][j], a[j][k]);
      z += m3_cmul(a[i][k], d[k]);
      const v2d w = m3_cmul(z, d[i]);
      a[i][k] = v2d{-w.x, -w.y};
    }
  a[5][5] = d[5];
#pragma unroll
  for (int k = 4; k >= 0; k--) {
#pragma unroll
    for (int j = k; j < 6; j++) u[j] = a[j][k];
    a[k][k] = d[k];
#pragma unroll
    for (int j = k + 1; j < 6; j++) a[j][k] = v2d{0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 6; i++) {
      v2d z = v2d{0.0, 0.0};
#pragma unroll
      for (int j = k; j < 6; j++) z += m3_cmul(a[i][j], u[j]);
      z = v2d{z.x / p[k], z.y / p[k]};
#pragma unroll
      for (int j = k; j < 6; j++) a[i][j] -= m3_cmul(m3_conj(u[j]), z);
    }
  }
  return fail;
}

// clover_invert.c:170-257: thread = (site i of parity ieo, chirality block b, set: 0 -> +mu, 1 -> -mu)
__global__ __launch_bounds__(64) void sw_invert_kernel(const v2d *__restrict__ swp, v2d *__restrict__ swi, int gs, int Vh, double mu, int *fails) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= Vh) return;
  const int b = blockIdx.y, set = blockIdx.z;
  v2d a[6][6];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      a[r][c] = swp[((size_t)(0 + b) * 9 + 3 * r + c) * gs + i];
      const v2d off = swp[((size_t)(2 + b) * 9 + 3 * r + c) * gs + i];
      a[r][c + 3] = off;
      a[c + 3][r] = m3_conj(off);
      a[r + 3][c + 3] = swp[((size_t)(4 + b) * 9 + 3 * r + c) * gs + i];
    }
  const double m = (set == 0 ? 1.0 : -1.0) * (b == 0 ? mu : -mu);
#pragma unroll
  for (int r = 0; r < 6; r++) a[r][r].y += m;
  const int f = six_invert_dev(a);
  if (f) atomicAdd(fails, f);
  v2d *o = swi + (size_t)set * 72 * gs + i;
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      o[((size_t)(0 + b) * 9 + 3 * r + c) * gs] = a[r][c];
      o[((size_t)(2 + b) * 9 + 3 * r + c) * gs] = a[r][c + 3];
      o[((size_t)(4 + b) * 9 + 3 * r + c) * gs] = a[r + 3][c + 3];
      o[((size_t)(6 + b) * 9 + 3 * r + c) * gs] = a[r + 3][c];
    }
}

// inverse of sw_sort_kernel / swinv_sort_kernel: device layout -> the reference's host layout
__global__ __launch_bounds__(256) void sw_unsort_kernel(v2d *__restrict__ raw, const v2d *__restrict__ d, int gs, int Vh, int LX, int LY, int LZ,
                                                        int toff) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Vh) return;
  const int par = blockIdx.y;
  const int LZh = LZ / 2;
  int r = i / LZh;
  const int y = r % LY;
  r /= LY;
  const int x = r % LX, t = r / LX;
  const int o = (t + x + y + toff + par) & 1;
  v2d *dst = raw + (2 * (size_t)i + o) * 54;
  const v2d *src = d + (size_t)par * 54 * gs + i;
#pragma unroll 6
  for (int e = 0; e < 54; e++) dst[e] = src[(size_t)e * gs];
}
__global__ __launch_bounds__(256) void swinv_unsort_kernel(v2d *__restrict__ raw, const v2d *__restrict__ d, int gs, int Vh) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Vh) return;
  const int sign = blockIdx.y;
  v2d *dst = raw + ((size_t)sign * Vh + i) * 72;
  const v2d *src = d + (size_t)sign * 72 * gs + i;
#pragma unroll 6
  for (int e = 0; e < 72; e++) dst[e] = src[(size_t)e * gs];
}

// ------------------------------------------------------------------ clover part of the fermion force
// (cloverdet_monomial.c:110-147: sw_spinor_eo x2, sw_deriv, sw_all).  swm / swp live in ctx->swpm as [2][4][9][V] with
// site index s = parity * Vh + e/o index.
__device__ __forceinline__ v2d cf_cmulc(v2d a, v2d b) { return v2d{a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y}; }   // a conj(b)

// operator/clover_deriv.c:252-318: thread = site of parity ieo; u carries the gamma5 sign (_mvector_tensor_vector)
__global__ __launch_bounds__(128) void sw_spinor_eo_kernel(v2d *__restrict__ swpm, const v2d *__restrict__ kk, const v2d *__restrict__ ll, int ns,
                                                           int Vh, int V, int ieo, double fac) {
  const int i = blockIdx.x * 128 + threadIdx.x;
  if (i >= Vh) return;
  const size_t s = (size_t)ieo * Vh + i;
  v2d r[4][3], q[4][3];
#pragma unroll
  for (int sp = 0; sp < 4; sp++)
#pragma unroll
    for (int c = 0; c < 3; c++) { r[sp][c] = kk[(size_t)(3 * sp + c) * ns + i]; q[sp][c] = ll[(size_t)(3 * sp + c) * ns + i]; }
  const int ra[4] = {0, 0, 1, 1}, sa[4] = {0, 1, 1, 0};     // v_n = r_{ra} (x) s_{sa}^dagger, u_n = -r_{ra+2} (x) s_{sa+2}^dagger
#pragma unroll
  for (int n = 0; n < 4; n++)
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 3; b++) {
        const v2d v = cf_cmulc(r[ra[n]][a], q[sa[n]][b]), w = cf_cmulc(r[ra[n] + 2][a], q[sa[n] + 2][b]);
        const v2d u = v2d{-w.x, -w.y};
        v2d *pm = swpm + ((size_t)(0 * 4 + n) * 9 + 3 * a + b) * V + s, *pp = swpm + ((size_t)(1 * 4 + n) * 9 + 3 * a + b) * V + s;
        const v2d m0 = *pm, p0 = *pp;
        *pm = v2d{m0.x + fac * (u.x - v.x), m0.y + fac * (u.y - v.y)};
        *pp = v2d{p0.x + fac * (u.x + v.x), p0.y + fac * (u.y + v.y)};
      }
}

// operator/clover_deriv.c:72-153: thread = (site of parity ieo, n); sw_inv blocks [set][2n + b]
__global__ __launch_bounds__(256) void sw_deriv_kernel(v2d *__restrict__ swpm, const v2d *__restrict__ swi, int gs, int Vh, int V, int ieo, int nsets,
                                                       double fac) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Vh) return;
  const int n = blockIdx.y;
  const size_t s = (size_t)ieo * Vh + i;
#pragma unroll
  for (int e = 0; e < 9; e++) {
    v2d lm = v2d{0.0, 0.0}, lp = v2d{0.0, 0.0};
    for (int set = 0; set < nsets; set++) {
      const v2d w0 = swi[((size_t)set * 72 + (size_t)(2 * n) * 9 + e) * gs + i], w1 = swi[((size_t)set * 72 + (size_t)(2 * n + 1) * 9 + e) * gs + i];
      lp += w1 + w0; lm += w1 - w0;
    }
    v2d *pm = swpm + ((size_t)(0 * 4 + n) * 9 + e) * V + s, *pp = swpm + ((size_t)(1 * 4 + n) * 9 + e) * V + s;
    const v2d m0 = *pm, p0 = *pp;
    *pm = v2d{m0.x + fac * lm.x, m0.y + fac * lm.y};
    *pp = v2d{p0.x + fac * lp.x, p0.y + fac * lp.y};
  }
}

__device__ __forceinline__ M3 m3_dag(const M3 &a) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) r.e[3 * i + j] = m3_conj(a.e[3 * j + i]);
  return r;
}
// ------------------------------------------------------------------ sw_all, owner-computes (no atomics)
// The reference walks (site x, plane kl) and scatters sixteen su3adj contributions (clover_accumulate_deriv.c:100-201; written that
// way for the GPU -- 96 fp64 atomics per thread -- it took 4.3 ms at 32^4, profiles/r02_swall_ab.log; that form is not kept).  Turned inside out like deriv_Sb: one thread OWNS the link (y, mu) and
// collects everything that reaches it.  Every contribution of the reference is a closed plaquette loop that starts with U_mu(y)
// and carries the insertion matrix W_kl(z) = vis[k][l](z) - h.c. (:73-99) at one of its four corners z -- with +W when the loop runs
// k -> l -> -k -> -l and W^dagger = -W (W is anti-hermitian bit for bit) the other way round; the transports the reference writes
// as w^dagger (..) w differ from this form by w w^dagger = 1 + O(eps) only.  A link lies in two plaquettes of each of the three
// planes (mu, nu), so with  A = U_mu(y), B = U_nu(y+mu), C = U_mu(y+nu), D = U_nu(y)  (upper plaquette) and
// B' = U_nu(y+mu-nu), C' = U_mu(y-nu), D' = U_nu(y-nu)  (lower one), Yu = B C^+ D^+, Yl = B'^+ C'^+ D':
//   S_nu = +-[ W(y+mu) (Yu - Yl) + (Yu - Yl) W(y) + B (W(y+mu+nu) C^+ + C^+ W(y+nu)) D^+ - B'^+ (W(y+mu-nu) C'^+ + C'^+ W(y-nu)) D' ]
//   derivative(y, mu) += c * trace_lambda(A (S_nu1 + S_nu2 + S_nu3))                       (+ for mu < nu, - for mu > nu)
// 43 3x3 products per link instead of the 66 per link of the scatter form, one read-modify-write of the link's eight
// doubles, all loads coalesced SoA planes (the stencil's gauge copy, forward entries only; swm / swp as they lie in HBM).
// On a T-split rank the t = 0 and t = T-1 slices need links and insertion matrices of the ring neighbours: they are done by the
// same code through the raw lexicographic links (halo slabs included, as for sw_term) and the neighbours' swm / swp slices,
// exchanged before the launch -- instead of exchanging derivative contributions afterwards.
// Two passes: sw_insertion_kernel builds the six anti-hermitian insertion matrices of every site once, in a compact form
// (nine reals in five complex planes: 480 B per site instead of the 2 x 144 B per plane that each of a site's users would load from
// swm / swp), sw_all_gather_kernel then does the loops.
struct SwSite { int t, x, y, z; };
__device__ __forceinline__ SwSite sw_shift(SwSite c, int dir, int d) {
  c.t += dir == 0 ? d : 0; c.x += dir == 1 ? d : 0; c.y += dir == 2 ? d : 0; c.z += dir == 3 ? d : 0;
  return c;
}
// W = X - X^dagger from the compact planes: (W01, W02, W12, (Im W00, Im W11), (Im W22, -))
__device__ __forceinline__ M3 sw_expand(v2d w01, v2d w02, v2d w12, v2d d01, v2d d2) {
  M3 W;
  W.e[0] = v2d{0.0, d01.x}; W.e[1] = w01; W.e[2] = w02;
  W.e[3] = v2d{-w01.x, w01.y}; W.e[4] = v2d{0.0, d01.y}; W.e[5] = w12;
  W.e[6] = v2d{-w02.x, w02.y}; W.e[7] = v2d{-w12.x, w12.y}; W.e[8] = v2d{0.0, d2.x};
  return W;
}
#ifndef SW_FAST_MINB
#define SW_FAST_MINB 2   /* blocks of three waves per CU the fast kernel is compiled for: 2 <=> at most 256 VGPRs (A/B builds override) */
#endif
// interior / unsplit lattices: periodic in every direction, SoA arrays in e/o order
struct SwFastLd {
  static constexpr int min_blocks = SW_FAST_MINB;
  const v2d *__restrict__ G; unsigned gs;     // gauge copy [2][8][9][gs]
  const v2d *__restrict__ W; unsigned ws;     // insertion matrices [6 planes][5][ws], site = parity * Vh + e/o index
  int T, LX, LY, LZ, Vh;
  typedef unsigned Loc;                        // e/o index of the site; its parity is wave-uniform and travels separately (`par`)
  __device__ __forceinline__ Loc locate(SwSite c) const {
    const int t = c.t < 0 ? c.t + T : (c.t >= T ? c.t - T : c.t), x = c.x < 0 ? c.x + LX : (c.x >= LX ? c.x - LX : c.x);
    const int y = c.y < 0 ? c.y + LY : (c.y >= LY ? c.y - LY : c.y), z = c.z < 0 ? c.z + LZ : (c.z >= LZ ? c.z - LZ : c.z);
    return (unsigned)(((t * LX + x) * LY + y) * LZ + z) >> 1;
  }
  // every address = wave-uniform 64-bit base + the lane's 32-bit site index
  __device__ __forceinline__ M3 link(Loc l, int par, int dir) const {
    const v2d *base = G + (size_t)(unsigned)((par * 8 + 2 * dir) * 9) * gs;
    M3 r;
#pragma unroll
    for (int e = 0; e < 9; e++) { const v2d *pe = base + (size_t)((unsigned)e * gs); r.e[e] = pe[l]; }
    return r;
  }
  __device__ __forceinline__ M3 ins(Loc l, int par, int P) const {
    const v2d *base = W + (size_t)(unsigned)(P * 5) * ws + (unsigned)(par * Vh);
    v2d c[5];
#pragma unroll
    for (int e = 0; e < 5; e++) { const v2d *pe = base + (size_t)((unsigned)e * ws); c[e] = pe[l]; }
    return sw_expand(c[0], c[1], c[2], c[3], c[4]);
  }
};
// t-faces of a T-split rank: raw lexicographic links [VPR][4][9] with the t = T / t = -1 halo slabs; the insertion matrices of the
// halo slabs arrive in `halo` [2 slabs][30][XYZ]
struct SwEdgeLd {
  static constexpr int min_blocks = 1;         // (two t-slices only: all 512 registers rather than scratch)
  const v2d *__restrict__ raw; const v2d *__restrict__ W; unsigned ws; const v2d *__restrict__ halo;
  LexGeom g; int Vh;
  struct Loc { int ix; const v2d *w; unsigned wst; };   // lexicographic index (halo slabs behind V); base and plane stride of the site's insertion matrices
  __device__ __forceinline__ Loc locate(SwSite c) const {
    const int ix = lex_index(g, c.t, c.x, c.y, c.z), XYZ = g.LX * g.LY * g.LZ;
    if (ix >= g.V) {
      const int slab = (ix - g.V) / XYZ, sp = (ix - g.V) - slab * XYZ;
      return Loc{ix, halo + (size_t)slab * 30 * XYZ + sp, (unsigned)XYZ};
    }
    const int x = (c.x + g.LX) % g.LX, y = (c.y + g.LY) % g.LY, z = (c.z + g.LZ) % g.LZ;
    const int par = (c.t + x + y + z) & 1;      // 0 <= c.t < T here
    return Loc{ix, W + (size_t)par * Vh + (ix >> 1), ws};
  }
  __device__ __forceinline__ M3 link(Loc l, int, int dir) const { return m3_load(raw, l.ix, dir); }
  __device__ __forceinline__ M3 ins(Loc l, int, int P) const {
    const v2d *p = l.w + (size_t)(unsigned)(P * 5) * l.wst;
    v2d c[5];
#pragma unroll
    for (int e = 0; e < 5; e++) c[e] = p[(size_t)((unsigned)e * l.wst)];
    return sw_expand(c[0], c[1], c[2], c[3], c[4]);
  }
};
// W_kl(z) of plane P = (01, 02, 03, 12, 13, 23) for every site: clover_accumulate_deriv.c:73-99, the same arithmetic as in sw_all_kernel
//   P: 0 -i(m1+m3), 1 m1-m3, 2 i(m2-m0), 3 i(p2-p0), 4 p3-p1, 5 -i(p1+p3);  W = X - X^dagger
// thread = (site s = parity * Vh + e/o index, set blockIdx.y: swm -> planes 0..2, swp -> planes 3..5); [2][4][9][V] -> Wc [6][5][ws].
// Every matrix of swm / swp is read once (1152 B per site in, 480 B out).
__device__ __forceinline__ void sw_ins_store(v2d *__restrict__ o, size_t ws, const v2d (&X)[9]) {
  o[0] = X[1] - m3_conj(X[3]);
  o[ws] = X[2] - m3_conj(X[6]);
  o[2 * ws] = X[5] - m3_conj(X[7]);
  o[3 * ws] = v2d{2.0 * X[0].y, 2.0 * X[4].y};      // (X - X^dagger)_ii = 2 i Im X_ii, exactly
  o[4 * ws] = v2d{2.0 * X[8].y, 0.0};
}
__global__ __launch_bounds__(256) void sw_insertion_kernel(v2d *__restrict__ Wc, const v2d *__restrict__ swpm, int V, unsigned ws) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= V) return;
  const int set = blockIdx.y;
  const v2d *p = swpm + (size_t)(set * 36) * V + s;
  v2d m0[9], m1[9], m2[9], m3[9], X[9];
#pragma unroll
  for (int e = 0; e < 9; e++) { m0[e] = p[(size_t)e * V]; m1[e] = p[(size_t)(9 + e) * V]; m2[e] = p[(size_t)(18 + e) * V]; m3[e] = p[(size_t)(27 + e) * V]; }
  v2d *o = Wc + (size_t)(set * 15) * ws + s;
  // set 0: planes 01, 02, 03 = -i(m1+m3), m1-m3, i(m2-m0);   set 1: planes 12, 13, 23 = i(p2-p0), p3-p1, -i(p1+p3)
#pragma unroll
  for (int e = 0; e < 9; e++) { const v2d sum = m1[e] + m3[e]; X[e] = v2d{sum.y, -sum.x}; }
  sw_ins_store(o + (size_t)(set ? 10 : 0) * ws, ws, X);
#pragma unroll
  for (int e = 0; e < 9; e++) X[e] = set ? m3[e] - m1[e] : m1[e] - m3[e];
  sw_ins_store(o + (size_t)5 * ws, ws, X);
#pragma unroll
  for (int e = 0; e < 9; e++) { const v2d dif = m2[e] - m0[e]; X[e] = v2d{-dif.y, dif.x}; }
  sw_ins_store(o + (size_t)(set ? 0 : 10) * ws, ws, X);
}
__device__ __forceinline__ void m3_sub(M3 &a, const M3 &b) {
#pragma unroll
  for (int e = 0; e < 9; e++) a.e[e] -= b.e[e];
}
// op(a) op(b) with every complex multiply-add as four fused multiply-adds (m3_mul's mul / fma / add form is 1.5x the instructions)
template <bool AD, bool BD>
__device__ __forceinline__ M3 m3_mulf(const M3 &a, const M3 &b) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double re = 0.0, im = 0.0;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const v2d x = AD ? a.e[3 * k + i] : a.e[3 * i + k], y = BD ? b.e[3 * j + k] : b.e[3 * k + j];
        const double xi = AD ? -x.y : x.y, yi = BD ? -y.y : y.y;
        re = __builtin_fma(x.x, y.x, re); re = __builtin_fma(-xi, yi, re);
        im = __builtin_fma(x.x, yi, im);  im = __builtin_fma(xi, y.x, im);
      }
      r.e[3 * i + j] = v2d{re, im};
    }
  return r;
}
#ifndef SW_NOSTEP
#define SW_STEP() __builtin_amdgcn_sched_barrier(0)   /* keeps the compiler from hoisting the next step's loads over this one: register budget */
#else
#define SW_STEP()
#endif
// what plane (mu, nu) adds to link (y, mu): out[0..7] = +-c * trace_lambda(A S_nu)
template <class LD>
__device__ __forceinline__ void sw_all_plane(double (&out)[8], const LD &ld, SwSite y, int par, int mu, int nu, double c) {
  const int opp = 1 - par;   // parity of the one-hop neighbours (wave-uniform like par)
  const int K = mu < nu ? mu : nu, L = mu < nu ? nu : mu;
  const int P = K == 0 ? L - 1 : (K == 1 ? L + 1 : 5);
  const SwSite ym = sw_shift(y, mu, 1);
  const typename LD::Loc ly = ld.locate(y), lym = ld.locate(ym), lyn = ld.locate(sw_shift(y, nu, 1)), lyd = ld.locate(sw_shift(y, nu, -1));
  const typename LD::Loc lymn = ld.locate(sw_shift(ym, nu, 1)), lymd = ld.locate(sw_shift(ym, nu, -1));
  M3 Yd, R;
  {  // upper plaquette  y -> y+mu -> y+mu+nu -> y+nu -> y
    const M3 C = ld.link(lyn, opp, mu);
    M3 F = m3_mulf<false, true>(ld.ins(lymn, par, P), C);
    m3_acc(F, m3_mulf<true, false>(C, ld.ins(lyn, opp, P)));
    SW_STEP();
    const M3 D = ld.link(ly, par, nu);
    const M3 Gu = m3_mulf<false, true>(F, D);
    const M3 E = m3_mulf<true, true>(C, D);
    SW_STEP();
    const M3 B = ld.link(lym, opp, nu);
    Yd = m3_mulf<false, false>(B, E);
    R = m3_mulf<false, false>(B, Gu);
  }
  SW_STEP();
  {  // lower plaquette  y -> y+mu -> y+mu-nu -> y-nu -> y
    const M3 C = ld.link(lyd, opp, mu);
    M3 F = m3_mulf<false, true>(ld.ins(lymd, par, P), C);
    m3_acc(F, m3_mulf<true, false>(C, ld.ins(lyd, opp, P)));
    SW_STEP();
    const M3 D = ld.link(lyd, opp, nu);
    const M3 Gl = m3_mulf<false, false>(F, D);
    const M3 E = m3_mulf<true, false>(C, D);
    SW_STEP();
    const M3 B = ld.link(lymd, par, nu);
    m3_sub(Yd, m3_mulf<true, false>(B, E));
    m3_sub(R, m3_mulf<true, false>(B, Gl));
  }
  SW_STEP();
  m3_acc(R, m3_mulf<false, false>(ld.ins(lym, opp, P), Yd));
  m3_acc(R, m3_mulf<false, false>(Yd, ld.ins(ly, par, P)));
  SW_STEP();
  const M3 a = m3_mulf<false, false>(ld.link(ly, par, mu), R);
  const double cs = mu < nu ? c : -c;
  // su3adj.h:164-172 (_trace_lambda_mul_add_assign)
  out[0] = cs * (-a.e[3].y - a.e[1].y);
  out[1] = cs * (+a.e[3].x - a.e[1].x);
  out[2] = cs * (-a.e[0].y + a.e[4].y);
  out[3] = cs * (-a.e[6].y - a.e[2].y);
  out[4] = cs * (+a.e[6].x - a.e[2].x);
  out[5] = cs * (-a.e[7].y - a.e[5].y);
  out[6] = cs * (+a.e[7].x - a.e[5].x);
  out[7] = cs * ((-a.e[0].y - a.e[4].y + 2.0 * a.e[8].y) * 0.577350269189625);
}
// Tile order of the 64-site blocks of a launch (sw_term_kernel below has the reasoning; sw_all_gather_kernel shares it): tb = 0 plain order.
struct SwOrder { int tb, tx, tyb, nby, nty, ntiles, t0, nt; };
// site block of thread block blockIdx.x (r = its running number within the XCD and parity), or -1: beyond the last tile
__device__ __forceinline__ int sw_tile_block(const SwOrder &ord, int LX, int r) {
  const int bx = r % ord.tb; r /= ord.tb;
  const int tt = r % ord.nt, tile = (r / ord.nt) * 8 + (int)(blockIdx.x & 7);
  if (tile >= ord.ntiles) return -1;
  const int tile_x = tile / ord.nty, tile_y = tile - tile_x * ord.nty, bxx = bx / ord.tyb, bxy = bx - bxx * ord.tyb;
  return (tt * LX + tile_x * ord.tx + bxx) * ord.nby + tile_y * ord.tyb + bxy;      // (whole time-slices from i_begin on)
}
// block = 64 sites of one parity x 4 waves: wave w owns the links (site, mu = w), walks the three planes through them one after the
// other (one body with RUN-TIME directions: twelve compile-time instances measured the same, at twelve times the code) and does the
// link's one read-modify-write.  Sites: e/o index in [i_begin, i_end) of either parity.
// Block order: blocks b, b + 8, .. share an XCD (and its L2); each XCD gets a contiguous chunk of the site blocks (or, slab order,
// its eighth of every time-slice) and runs the two parities of a site block back to back, so the links and insertion matrices
// around those 64 + 64 sites are fetched once and re-used out of L2.
template <class LD>
__global__ __launch_bounds__(256, LD::min_blocks) void sw_all_gather_kernel(const LD ld, double *__restrict__ deriv, int LX, int LY, int LZ, int Vh,
                                                               int i_begin, int i_end, int chunk, int slab, int nbt, double c, const SwOrder ord) {
  const int lane = threadIdx.x & 63;
  const int mu = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform, in a scalar register
  const int q = blockIdx.x >> 3;
  int sb;
  if (ord.tb > 0) {  // tile order ("swall_order" 2): an XCD walks a tile of tx x-planes x tyb block rows through all time-slices
    sb = sw_tile_block(ord, LX, q >> 1);
    if (sb < 0) return;
  } else if (slab > 0) {    // slab order: XCD j owns the j-th eighth of EVERY time-slice (nbt site blocks each) and marches through t
    const int r = q >> 1, tt = r / slab, rr = (blockIdx.x & 7) * slab + (r - tt * slab);
    if (rr >= nbt) return;
    sb = tt * nbt + rr;
  } else {
    sb = (blockIdx.x & 7) * chunk + (q >> 1);
  }
  const int par = q & 1;
  const int i = i_begin + sb * 64 + lane;
  if (i >= i_end) return;
  const int LZh = LZ / 2;
  SwSite s;
  {
    int r = i / LZh;
    const int k = i - r * LZh;
    s.y = r % LY; r /= LY;
    s.x = r % LX; s.t = r / LX;
    s.z = 2 * k + ((s.t + s.x + s.y + par) & 1);
  }
  // the running su3adj sum lives in LDS (a private column per thread, no synchronisation): eight doubles too many for the 256-register budget
  __shared__ double acc[8][256];
  double o[8];
  sw_all_plane(o, ld, s, par, mu, mu == 0 ? 1 : 0, c);                // nu = the first, second, third direction other than mu
#pragma unroll
  for (int m = 0; m < 8; m++) acc[m][threadIdx.x] = o[m];
  SW_STEP();
  sw_all_plane(o, ld, s, par, mu, mu <= 1 ? 2 : 1, c);
#pragma unroll
  for (int m = 0; m < 8; m++) acc[m][threadIdx.x] += o[m];
  SW_STEP();
  sw_all_plane(o, ld, s, par, mu, mu <= 2 ? 3 : 2, c);
  double *d = deriv + ((size_t)par * 32 + (size_t)mu * 8) * Vh + i;
#pragma unroll
  for (int m = 0; m < 8; m++) d[(size_t)m * Vh] += acc[m][threadIdx.x] + o[m];
}
// the t = 0 / t = T-1 slices of the insertion matrices for the ring neighbours: out[w][30][XYZ], w = 0: our t = 0, w = 1: our t = T-1
__global__ __launch_bounds__(256) void sw_pack_slabs_kernel(v2d *__restrict__ out, const v2d *__restrict__ Wc, unsigned ws, LexGeom g, int Vh) {
  const int XYZ = g.LX * g.LY * g.LZ;
  const int sp = blockIdx.x * 256 + threadIdx.x;
  if (sp >= XYZ) return;
  const int w = blockIdx.y, t = w ? g.T - 1 : 0;
  const int z = sp % g.LZ, y = (sp / g.LZ) % g.LY, x = sp / (g.LZ * g.LY);
  const size_t s = (size_t)((t + x + y + z) & 1) * Vh + ((t * XYZ + sp) >> 1);
#pragma unroll 6
  for (int m = 0; m < 30; m++) out[((size_t)w * 30 + m) * XYZ + sp] = Wc[(size_t)m * ws + s];
}

// ------------------------------------------------------------------ sw_term in one kernel
// (operator/clover_term.c:88-200).  Block = 64 sites of one parity x 6 waves: wave p builds F_kl = Q_kl - Q_kl^dagger of plane
// p = (01, 02, 03, 12, 13, 23) -- Q = the four plaquette leaves around the site (:104-154), twelve forward links of eight sites,
// loaded as coalesced SoA planes of the stencil's gauge copy (interior / unsplit) or from the raw links with their halo slabs (the
// t-faces of a T-split rank) -- the six F meet in LDS (54 KB), and wave b writes block b of sw[x][3][2] = 1 + (kappa c_sw / 8) *
// combinations of E_k = F_0k, B_1 = F_23, B_2 = -F_13, B_3 = F_12 (:156-197) straight into the device layout swd[par][2a+b][e][i].
// Round 1 walked the raw AoS links with one thread per (site, plane) and went through a 906 MB F array in HBM: 2.25 + 0.5 ms at 32^4
// (3.0 ms per call with its allocation); this kernel: 1.26 ms per call.
#ifndef SWT_MINW
#define SWT_MINW 2   /* 3 waves per SIMD would fit the LDS but spills (168 VGPRs, 160 B of scratch): 1.77 vs 1.26 ms at 32^4 */
#endif
#ifdef SWT_NOSTEP
#define SWT_STEP()
#else
#define SWT_STEP() SW_STEP()
#endif
// Block order (round 4, review item 6).  The leaves of a site reach one step in every direction, so a link is wanted by the blocks of
// its own (t, x) row, of the rows x +- 1 and of the time-slices t +- 1.  In the plain order (a contiguous eighth of the sites per XCD,
// t slowest) a time-slice of links is 19 MB at 32^3 -- nothing of slice t is left in the 4 MB L2 when t + 1 comes by, and every link
// was fetched ~4.7 times (5447 instead of 1152 bytes per site of one parity: profiles/r03_summary.md).  TILE order: an XCD owns tiles of
// tx x-planes x tyb block-rows x all z (4 x 4 x 32 sites at 32^3: 0.3 MB of links per time-slice, 0.66 MB with the halo ring) and walks
// each tile through ALL time-slices before it takes the next one; slices t - 1, t, t + 1 of tile + halo (2 MB) stay in L2 while t
// advances, so a link comes in once per tile that owns or borders it.  tb = 0: the plain order.
// The four plaquette leaves around x in the (k, l) plane (clover_term.c:104-154), two at a time: A = the leaves in the +k+l and -k+l
// quadrants, B = those in -k-l and +k-l.  Q is set (init) or added to.
// A link that two consecutive leaves share is loaded ONCE and carried in registers (U_l(x) within A, U_l(x-l) within B, U_k(x-k) from A to
// B: 13 link loads per plane instead of 16: 0.88 -> 0.82 ms at 32^4.  Carrying U_k(x) from the first leaf to the last as well (12 loads, +10
// VGPRs) or dropping the scheduling barriers measured the same within noise, profiles/r04_swterm_ab.log; counters: the texture addressers are
// busy 79 % of the kernel, the fp64 VALUs 52 %, at two waves per SIMD).
template <class LD>
__device__ __forceinline__ void sw_leaves_a(M3 &Q, const LD &ld, const SwSite &x, int par, int k, int l, bool init, M3 *carry = nullptr) {
  const int opp = 1 - par;
  const SwSite xmk = sw_shift(x, k, -1);
  const typename LD::Loc l0 = ld.locate(x), lpk = ld.locate(sw_shift(x, k, 1)), lpl = ld.locate(sw_shift(x, l, 1)), lmk = ld.locate(xmk), lplmk = ld.locate(sw_shift(xmk, l, 1));
  M3 v1, v2;
  const M3 ulx = ld.link(l0, par, l);
  v1 = m3_mulf<false, false>(ld.link(l0, par, k), ld.link(lpk, opp, l));
  v2 = m3_mulf<false, false>(ulx, ld.link(lpl, opp, k));
  if (init) Q = m3_mulf<false, true>(v1, v2); else m3_acc(Q, m3_mulf<false, true>(v1, v2));
  SWT_STEP();
  v1 = m3_mulf<false, true>(ulx, ld.link(lplmk, par, k));
  const M3 ukm = ld.link(lmk, opp, k);
  v2 = m3_mulf<true, false>(ld.link(lmk, opp, l), ukm);
  m3_acc(Q, m3_mulf<false, false>(v1, v2));
  if (carry) *carry = ukm;
}
template <class LD>
__device__ __forceinline__ void sw_leaves_b(M3 &Q, const LD &ld, const SwSite &x, int par, int k, int l, bool init, const M3 *carry = nullptr) {
  const int opp = 1 - par;
  const SwSite xmk = sw_shift(x, k, -1);
  const typename LD::Loc l0 = ld.locate(x), lmk = ld.locate(xmk), lml = ld.locate(sw_shift(x, l, -1)), lmkml = ld.locate(sw_shift(xmk, l, -1)), lpkml = ld.locate(sw_shift(sw_shift(x, k, 1), l, -1));
  M3 v1, v2;
  v1 = m3_mulf<false, false>(ld.link(lmkml, par, l), carry ? *carry : ld.link(lmk, opp, k));
  const M3 ulm = ld.link(lml, opp, l);
  v2 = m3_mulf<false, false>(ld.link(lmkml, par, k), ulm);
  if (init) Q = m3_mulf<true, false>(v1, v2); else m3_acc(Q, m3_mulf<true, false>(v1, v2));
  SWT_STEP();
  v1 = m3_mulf<true, false>(ulm, ld.link(lml, opp, k));
  v2 = m3_mulf<false, true>(ld.link(lpkml, par, l), ld.link(l0, par, k));
  m3_acc(Q, m3_mulf<false, false>(v1, v2));
}
// Block = 64 sites of one parity x FOUR waves (round 4).  At 206-228 VGPRs a SIMD holds two waves, a CU eight: the six-wave block of
// rounds 2-3 (one plane per wave) left two of the eight slots empty and one block per CU.  Now the 6 planes x 2 leaf pairs = 12 half-works
// of a block are spread evenly: wave w does both halves of plane w (0 .. 3), then one half of plane 4 + (w >> 1); two blocks share a CU
// (2 x 54 KB of LDS).  The F_kl of planes 4 and 5 are completed in LDS by the second of their two waves, behind a barrier.
template <class LD>
__global__ __launch_bounds__(256, LD::min_blocks == 1 ? 1 : SWT_MINW) void sw_term_kernel(const LD ld, v2d *__restrict__ swd, unsigned gs, int LX, int LY, int LZ,
                                                                                   int i_begin, int i_end, int chunk, double ka_csw_8, const SwOrder ord) {
  __shared__ v2d F[6][9][64];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = blockIdx.x >> 3, par = q & 1;
  int sb;
  if (ord.tb > 0) {
    sb = sw_tile_block(ord, LX, q >> 1);
    if (sb < 0) return;
  } else {
    sb = (blockIdx.x & 7) * chunk + (q >> 1);
  }
  if (sb * 64 >= i_end - i_begin) return;
  const int i = i_begin + sb * 64 + lane;
  const bool active = i < i_end;
  const int LZh = LZ / 2;
  SwSite x;
  {
    const int ii = active ? i : i_begin;
    int r = ii / LZh;
    const int kz = ii - r * LZh;
    x.y = r % LY; r /= LY;
    x.x = r % LX; x.t = r / LX;
    x.z = 2 * kz + ((x.t + x.x + x.y + par) & 1);
  }
  auto plane_kl = [](int p, int &k, int &l) { k = p < 3 ? 0 : (p < 5 ? 1 : 2); l = p < 3 ? p + 1 : (p < 5 ? p - 1 : 3); };   // p = (01, 02, 03, 12, 13, 23)
  M3 Q;
  int k, l;
  plane_kl(w, k, l);
  {
    M3 ukm;
    sw_leaves_a(Q, ld, x, par, k, l, true, &ukm);
    SWT_STEP();
    sw_leaves_b(Q, ld, x, par, k, l, false, &ukm);
  }
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 3; b++) F[w][3 * a + b][lane] = Q.e[3 * a + b] - m3_conj(Q.e[3 * b + a]);
  SWT_STEP();
  const int p2 = 4 + (w >> 1), second = w & 1;       // wave-uniform
  plane_kl(p2, k, l);
  if (second) sw_leaves_b(Q, ld, x, par, k, l, true); else sw_leaves_a(Q, ld, x, par, k, l, true);
  if (!second) {
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 3; b++) F[p2][3 * a + b][lane] = Q.e[3 * a + b] - m3_conj(Q.e[3 * b + a]);
  }
  __syncthreads();
  if (second) {
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 3; b++) F[p2][3 * a + b][lane] += Q.e[3 * a + b] - m3_conj(Q.e[3 * b + a]);
  }
  __syncthreads();
  if (!active) return;
  auto itimes = [](v2d a) { return v2d{-a.y, a.x}; };
  for (int p = w; p < 6; p += 4) {                                     // block p = 2a + b of sw[x][a][b]: waves 0, 1 write two of them
    v2d *dst = swd + ((size_t)par * 54 + (size_t)p * 9) * gs + i;
#pragma unroll
    for (int e = 0; e < 9; e++) {
      const v2d e1 = F[0][e][lane], e2 = F[1][e][lane], e3 = F[2][e][lane], m3 = F[3][e][lane], f13 = F[4][e][lane], m1 = F[5][e][lane];
      const v2d m2 = v2d{-f13.x, -f13.y};
      const double one = (e == 0 || e == 4 || e == 8) ? 1.0 : 0.0;
      v2d a, r;
      switch (p) {
        case 0: a = itimes(e3 - m3);             r = v2d{one + ka_csw_8 * a.x, ka_csw_8 * a.y}; break;     // sw[x][0][0]
        case 1: a = itimes(e3 + m3);             r = v2d{one - ka_csw_8 * a.x, -ka_csw_8 * a.y}; break;    // sw[x][0][1]
        case 2: a = itimes(e1 - m1) + (e2 - m2); r = v2d{ka_csw_8 * a.x, ka_csw_8 * a.y}; break;           // sw[x][1][0]
        case 3: a = itimes(e1 + m1) + (e2 + m2); r = v2d{-ka_csw_8 * a.x, -ka_csw_8 * a.y}; break;         // sw[x][1][1]
        case 4: a = itimes(m3 - e3);             r = v2d{one + ka_csw_8 * a.x, ka_csw_8 * a.y}; break;     // sw[x][2][0]
        default: a = itimes(m3 + e3);            r = v2d{one + ka_csw_8 * a.x, ka_csw_8 * a.y}; break;     // sw[x][2][1]
      }
      __builtin_nontemporal_store(r, dst + (size_t)((unsigned)e * gs));     // written once, read by other kernels: must not push the links out of L2
    }
  }
}

static int need64(const tmhip_field *f, const char *who) {
  if (!f || f->kind != TMHIP_FIELD_EO || f->prec != 0) { fprintf(stderr, "[tmlqcd_hip] %s: needs a one-parity fp64 field\n", who); return 1; }
  return 0;
}
// nullptr (with a message) when the -mu set is wanted but the last tmhip_sw_invert / tmhip_set_clover built the +mu set only:
// every caller hands the pointer to a launcher that refuses a null clover array
static inline const v2d *swinv(tmhip_ctx *ctx, int tau3sign, double mu) {   /* clovertm_operators.c:298-300 */
  const int set = (tau3sign < 0 && fabs(mu) > 0) ? 1 : 0;
  if (set >= ctx->sw_inv_sets) {
    fprintf(stderr, "[tmlqcd_hip] clover operator needs the -mu set of sw_inv, but sw_inv holds %d set(s): call sw_invert with the current mu\n", ctx->sw_inv_sets);
    return nullptr;
  }
  return ctx->sw_inv + (size_t)set * 72 * ctx->gs;
}
static inline const v2d *swpar(tmhip_ctx *ctx, int ieo) { return ctx->sw + (size_t)(ieo ? 1 : 0) * 54 * ctx->gs; }

int tmhip_prepare_clover32(tmhip_ctx *ctx) {
  if (!ctx->clover_set) TMHIP_FAIL("clover operator called before tmhip_set_clover");
  if (tmhip_prepare_fp32(ctx)) return 1;
  const size_t n1 = (size_t)2 * 54 * ctx->gs, n2 = (size_t)2 * 72 * ctx->gs;
  if (!ctx->sw32) TMHIP_CHECK(hipMalloc((void **)&ctx->sw32, n1 * sizeof(v2f)));
  if (!ctx->sw_inv32) TMHIP_CHECK(hipMalloc((void **)&ctx->sw_inv32, n2 * sizeof(v2f)));
  if (!ctx->clover32_set) {   /* copy_32_sw_fields (operator.c:367) */
    hipLaunchKernelGGL(clover_to32_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, ctx->stream, ctx->sw32, ctx->sw, n1);
    hipLaunchKernelGGL(clover_to32_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, ctx->stream, ctx->sw_inv32, ctx->sw_inv, n2);
    TMHIP_CHECK(hipGetLastError());
    ctx->clover32_set = true;
  }
  return 0;
}

extern "C" {

int tmhip_set_clover(tmhip_ctx *ctx, const void *sw_host, const void *sw_inv_host) {
  if (!sw_host || !sw_inv_host) TMHIP_FAIL("tmhip_set_clover: null argument");
  TMHIP_CHECK(hipSetDevice(ctx->device));
  const size_t n1 = (size_t)2 * 54 * ctx->gs, n2 = (size_t)2 * 72 * ctx->gs;
  if (!ctx->sw) TMHIP_CHECK(hipMalloc((void **)&ctx->sw, n1 * sizeof(v2d)));
  if (!ctx->sw_inv) TMHIP_CHECK(hipMalloc((void **)&ctx->sw_inv, n2 * sizeof(v2d)));
  const size_t b1 = (size_t)ctx->V * 54 * sizeof(v2d), b2 = (size_t)ctx->V * 72 * sizeof(v2d);
  void *raw = nullptr;
  TMHIP_CHECK(hipMalloc(&raw, b2));
  TMHIP_CHECK(hipMemcpyAsync(raw, sw_host, b1, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(sw_sort_kernel, dim3((ctx->Vh + 255) / 256, 2), dim3(256), 0, ctx->stream, (const v2d *)raw, ctx->sw, ctx->gs, ctx->Vh,
                     ctx->g.LX, ctx->g.LY, ctx->g.LZ, ctx->g.proc_t * ctx->g.T);
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  TMHIP_CHECK(hipMemcpyAsync(raw, sw_inv_host, b2, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(swinv_sort_kernel, dim3((ctx->Vh + 255) / 256, 2), dim3(256), 0, ctx->stream, (const v2d *)raw, ctx->sw_inv, ctx->gs, ctx->Vh);
  TMHIP_CHECK(hipGetLastError());
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  TMHIP_CHECK(hipFree(raw));
  ctx->clover_set = true;
  ctx->sw_set = true;
  ctx->sw_inv_sets = 2;
  ctx->clover32_set = false;
  return 0;
}

static int clover_alloc(tmhip_ctx *ctx) {
  const size_t n1 = (size_t)2 * 54 * ctx->gs, n2 = (size_t)2 * 72 * ctx->gs;
  if (!ctx->sw) TMHIP_CHECK(hipMalloc((void **)&ctx->sw, n1 * sizeof(v2d)));
  if (!ctx->sw_inv) TMHIP_CHECK(hipMalloc((void **)&ctx->sw_inv, n2 * sizeof(v2d)));
  if (!ctx->sw_fail) TMHIP_CHECK(hipMalloc((void **)&ctx->sw_fail, sizeof(int)));
  return 0;
}

/* operator/clover_term.c:88 sw_term(gf, kappa, c_sw): gf is the host gauge field exactly as for tmhip_set_gauge
 * ([VOLUMEPLUSRAND][4] su3, halo slabs filled on T-split ranks).  Result stays in HBM (fetch with tmhip_get_clover). */
// Tile order of a launch over the e/o sites [ib, ie) (whole time-slices): fills *ord and returns the grid, or 0 when the shape does not
// allow it (a 64-site block must be whole z-rows of one (t, x) row).  txw: x-planes per tile.
static int sw_tile_order(const tmhip_ctx *ctx, int ib, int ie, int txw, SwOrder *ord) {
  const int LZh = ctx->g.LZ / 2, rpb = LZh > 0 && 64 % LZh == 0 ? 64 / LZh : 0;
  if (!(rpb > 0 && ctx->g.LY % rpb == 0 && ib % ctx->face == 0 && ie % ctx->face == 0 && ctx->face % 64 == 0)) return 0;
  ord->nby = ctx->g.LY / rpb;
  ord->tx = ctx->g.LX % txw == 0 ? txw : (ctx->g.LX % 4 == 0 ? 4 : (ctx->g.LX % 2 == 0 ? 2 : 1));
  ord->tyb = rpb >= 4 ? 1 : ((4 / rpb) <= ord->nby && ord->nby % (4 / rpb) == 0 ? 4 / rpb : 1);
  ord->nty = ord->nby / ord->tyb;
  ord->ntiles = (ctx->g.LX / ord->tx) * ord->nty;
  ord->tb = ord->tx * ord->tyb;
  ord->t0 = ib / ctx->face; ord->nt = (ie - ib) / ctx->face;
  return 8 * ((ord->ntiles + 7) / 8) * ord->nt * ord->tb * 2;
}
int tmhip_sw_term(tmhip_ctx *ctx, const void *gauge_host, double kappa, double c_sw) {
  if (!gauge_host && !(ctx->gauge_raw && ctx->gauge_raw_valid)) TMHIP_FAIL("tmhip_sw_term: null gauge field and no links resident on the device");
  TMHIP_CHECK(hipSetDevice(ctx->device));
  if (clover_alloc(ctx)) return 1;
  const size_t gbytes = (size_t)ctx->VPR * 4 * 9 * sizeof(v2d);
  if (!ctx->gauge_raw) TMHIP_CHECK(hipMalloc((void **)&ctx->gauge_raw, gbytes));   // kept: the clover force (tmhip_sw_all) and the t-faces of split ranks walk these links
  if (gauge_host) {   // NULL: the links tmhip_set_gauge / tmhip_update_gauge left in HBM
    TMHIP_CHECK(hipMemcpyAsync(ctx->gauge_raw, gauge_host, gbytes, hipMemcpyHostToDevice, ctx->stream));
    ctx->gauge_copy_current = false;
    ctx->gauge_raw_valid = true;
  }
  const bool split = ctx->g.nproc_t > 1;
  const int ib = split ? ctx->face : 0, ie = split ? ctx->Vh - ctx->face : ctx->Vh;      // sites whose plaquettes stay on this rank
  const double c = kappa * c_sw / 8.;
  if (ie > ib && (!split || ctx->g.T > 2)) {
    if (!ctx->gauge_copy_current && tmhip_resort_gauge(ctx)) return 1;     // the SoA planes the interior reads must come from these links
    const SwFastLd ld{ctx->gauge, (unsigned)ctx->gs, nullptr, 0u, ctx->g.T, ctx->g.LX, ctx->g.LY, ctx->g.LZ, ctx->Vh};
    const int chunk = ((ie - ib + 63) / 64 + 7) / 8;
    // tile order whenever a 64-site block is whole z-rows of one (t, x) row and the range is whole time-slices ("swterm_order" 0: plain order)
    SwOrder ord = {0, 0, 0, 0, 0, 0, 0, 0};
    int grid = chunk * 16;
    if (ctx->opt_swterm_order) { const int gt = sw_tile_order(ctx, ib, ie, 4, &ord); if (gt) grid = gt; }
    hipLaunchKernelGGL((sw_term_kernel<SwFastLd>), dim3(grid), dim3(256), 0, ctx->stream, ld, ctx->sw, (unsigned)ctx->gs, ctx->g.LX, ctx->g.LY, ctx->g.LZ,
                       ib, ie, chunk, c, ord);
  }
  if (split) {
    LexGeom g{ctx->g.T, ctx->g.LX, ctx->g.LY, ctx->g.LZ, ctx->V, 1};
    const SwEdgeLd ld{(const v2d *)ctx->gauge_raw, nullptr, 0u, nullptr, g, ctx->Vh};
    const int chunk = ((ctx->face + 63) / 64 + 7) / 8;
    for (int w = 0; w < (ctx->g.T > 1 ? 2 : 1); w++) {
      const int fb = w ? ctx->Vh - ctx->face : 0;
      hipLaunchKernelGGL((sw_term_kernel<SwEdgeLd>), dim3(chunk * 16), dim3(256), 0, ctx->stream, ld, ctx->sw, (unsigned)ctx->gs, ctx->g.LX, ctx->g.LY, ctx->g.LZ,
                         fb, fb + ctx->face, chunk, c, SwOrder{0, 0, 0, 0, 0, 0, 0, 0});
    }
  }
  TMHIP_CHECK(hipGetLastError());
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  ctx->sw_set = true;
  ctx->clover_set = false;     // sw_inv no longer matches
  ctx->clover32_set = false;
  return 0;
}

/* operator/clover_invert.c:170 sw_invert(ieo, mu): inverse of (1 + T +- i mu g5) on the sites of parity ieo, +mu set and
 * (mu != 0) -mu set, into the one sw_inv array -- like the reference's global, it holds one parity at a time and the
 * operators of this library expect ieo = EE (0), as operator.c:364 and invert_clover_eo.c use it. */
int tmhip_sw_invert(tmhip_ctx *ctx, int ieo, double mu) {
  if (!ctx->sw_set) TMHIP_FAIL("tmhip_sw_invert called before tmhip_sw_term / tmhip_set_clover");
  TMHIP_CHECK(hipSetDevice(ctx->device));
  if (clover_alloc(ctx)) return 1;
  TMHIP_CHECK(hipMemsetAsync(ctx->sw_fail, 0, sizeof(int), ctx->stream));
  const int nsets = fabs(mu) > 0. ? 2 : 1;   /* clover_invert.c:225 */
  hipLaunchKernelGGL(sw_invert_kernel, dim3((ctx->Vh + 63) / 64, 2, nsets), dim3(64), 0, ctx->stream, swpar(ctx, ieo), ctx->sw_inv, ctx->gs,
                     ctx->Vh, mu, ctx->sw_fail);
  TMHIP_CHECK(hipGetLastError());
  int fails = 0;
  TMHIP_CHECK(hipMemcpyAsync(&fails, ctx->sw_fail, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (fails > 0 && ctx->g.proc_t == 0) printf("# inversion failed in six_invert code %d\n", fails);   /* clover_invert.c:213-216 */
  ctx->sw_inv_sets = nsets;
  ctx->clover_set = true;
  ctx->clover32_set = false;
  return 0;
}

/* Copies the device-resident blocks back in the reference's layouts (either pointer may be NULL):
 * sw[VOLUME][3][2], sw_inv[VOLUME][4][2] su3. */
int tmhip_get_clover(tmhip_ctx *ctx, void *sw_host, void *sw_inv_host) {
  TMHIP_CHECK(hipSetDevice(ctx->device));
  if (sw_host && !ctx->sw_set) TMHIP_FAIL("tmhip_get_clover: no clover term on the device");
  if (sw_inv_host && !ctx->clover_set) TMHIP_FAIL("tmhip_get_clover: no inverse clover term on the device");
  const size_t b1 = (size_t)ctx->V * 54 * sizeof(v2d), b2 = (size_t)ctx->V * 72 * sizeof(v2d);
  void *raw = nullptr;
  TMHIP_CHECK(hipMalloc(&raw, b2));
  if (sw_host) {
    hipLaunchKernelGGL(sw_unsort_kernel, dim3((ctx->Vh + 255) / 256, 2), dim3(256), 0, ctx->stream, (v2d *)raw, (const v2d *)ctx->sw, ctx->gs, ctx->Vh,
                       ctx->g.LX, ctx->g.LY, ctx->g.LZ, ctx->g.proc_t * ctx->g.T);
    TMHIP_CHECK(hipMemcpyAsync(sw_host, raw, b1, hipMemcpyDeviceToHost, ctx->stream));
    TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  }
  if (sw_inv_host) {
    // like the reference, leave the -mu half of the host array alone when it was not computed (mu == 0)
    hipLaunchKernelGGL(swinv_unsort_kernel, dim3((ctx->Vh + 255) / 256, ctx->sw_inv_sets), dim3(256), 0, ctx->stream, (v2d *)raw,
                       (const v2d *)ctx->sw_inv, ctx->gs, ctx->Vh);
    TMHIP_CHECK(hipMemcpyAsync(sw_inv_host, raw, b2 / 2 * ctx->sw_inv_sets, hipMemcpyDeviceToHost, ctx->stream));
    TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  }
  TMHIP_CHECK(hipGetLastError());
  TMHIP_CHECK(hipFree(raw));
  return 0;
}

/* ---- clover part of the fermion force (SURVEY §8f rank 3 on top of the clover row): device-resident swm / swp -------------- */
static int swpm_alloc(tmhip_ctx *ctx) {
  if (ctx->swpm) return 0;
  const size_t bytes = (size_t)2 * 4 * 9 * ctx->V * sizeof(v2d);
  TMHIP_CHECK(hipMalloc((void **)&ctx->swpm, bytes));
  TMHIP_CHECK(hipMemsetAsync(ctx->swpm, 0, bytes, ctx->stream));
  return 0;
}
/* the loop that zeroes swm / swp at the start of cloverdet_derivative (monomial/cloverdet_monomial.c:67-72) */
int tmhip_swpm_zero(tmhip_ctx *ctx) {
  TMHIP_CHECK(hipSetDevice(ctx->device));
  if (!ctx->swpm) return swpm_alloc(ctx);
  TMHIP_CHECK(hipMemsetAsync(ctx->swpm, 0, (size_t)2 * 4 * 9 * ctx->V * sizeof(v2d), ctx->stream));
  return 0;
}
/* operator/clover_deriv.c:252 sw_spinor_eo(ieo, kk, ll, fac) */
int tmhip_sw_spinor_eo(tmhip_ctx *ctx, int ieo, tmhip_field *kk, tmhip_field *ll, double fac) {
  if (need64(kk, "sw_spinor_eo") || need64(ll, "sw_spinor_eo")) return 1;
  if (kk->ns != ll->ns) TMHIP_FAIL("sw_spinor_eo: fields with different strides");
  TMHIP_CHECK(hipSetDevice(ctx->device));
  if (swpm_alloc(ctx)) return 1;
  hipLaunchKernelGGL(sw_spinor_eo_kernel, dim3((ctx->Vh + 127) / 128), dim3(128), 0, ctx->stream, ctx->swpm, (const v2d *)kk->d, (const v2d *)ll->d,
                     kk->ns, ctx->Vh, ctx->V, ieo ? 1 : 0, fac);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}
/* operator/clover_deriv.c:72 sw_deriv(ieo, mu): needs the sw_inv of parity ieo (tmhip_sw_invert(ieo, mu) / tmhip_set_clover) */
int tmhip_sw_deriv(tmhip_ctx *ctx, int ieo, double mu) {
  if (!ctx->clover_set) TMHIP_FAIL("sw_deriv called before tmhip_sw_invert / tmhip_set_clover");
  const int nsets = fabs(mu) > 0. ? 2 : 1;
  if (nsets > ctx->sw_inv_sets) TMHIP_FAIL("sw_deriv: mu != 0 but sw_inv holds the +mu set only");
  TMHIP_CHECK(hipSetDevice(ctx->device));
  if (swpm_alloc(ctx)) return 1;
  hipLaunchKernelGGL(sw_deriv_kernel, dim3((ctx->Vh + 255) / 256, 4), dim3(256), 0, ctx->stream, ctx->swpm, (const v2d *)ctx->sw_inv, ctx->gs, ctx->Vh,
                     ctx->V, ieo ? 1 : 0, nsets, nsets == 2 ? 0.5 : 1.0);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}
static int sw_all_prepare(tmhip_ctx *ctx, const void *gauge_host) {
  if (!ctx->swpm) TMHIP_FAIL("sw_all called before sw_spinor_eo / sw_deriv");
  TMHIP_CHECK(hipSetDevice(ctx->device));
  const size_t gbytes = (size_t)ctx->VPR * 4 * 9 * sizeof(v2d);
  if (gauge_host) {
    if (!ctx->gauge_raw) TMHIP_CHECK(hipMalloc((void **)&ctx->gauge_raw, gbytes));
    TMHIP_CHECK(hipMemcpyAsync(ctx->gauge_raw, gauge_host, gbytes, hipMemcpyHostToDevice, ctx->stream));
    ctx->gauge_raw_valid = true;
    ctx->gauge_copy_current = false;
  } else if (!ctx->gauge_raw_valid) {
    TMHIP_FAIL("sw_all: no lexicographic gauge field on the device (pass the host field, or call tmhip_sw_term after tmhip_set_gauge)");
  }
  if (!ctx->deriv && tmhip_derivative_zero(ctx)) return 1;
  const size_t XYZ = (size_t)ctx->g.LX * ctx->g.LY * ctx->g.LZ;
  // owner-computes form: the interior (or the whole unsplit lattice) reads the stencil's gauge copy -- it must come from the same links
  if (!ctx->gauge_copy_current && (ctx->g.nproc_t == 1 || ctx->g.T > 2) && tmhip_resort_gauge(ctx)) return 1;
  // pass 1: the six insertion matrices of every site, compact (30 complex planes)
  const unsigned ws = (unsigned)ctx->V;
  if (!ctx->sw_ins) TMHIP_CHECK(hipMalloc((void **)&ctx->sw_ins, (size_t)30 * ws * sizeof(v2d)));
  hipLaunchKernelGGL(sw_insertion_kernel, dim3((ctx->V + 255) / 256, 2), dim3(256), 0, ctx->stream, ctx->sw_ins, (const v2d *)ctx->swpm, ctx->V, ws);
  TMHIP_CHECK(hipGetLastError());
  if (ctx->g.nproc_t > 1) {
    const size_t hb = (size_t)2 * 30 * XYZ * sizeof(v2d);
    if (!ctx->swpm_halo_send) TMHIP_CHECK(hipMalloc((void **)&ctx->swpm_halo_send, hb));
    if (!ctx->swpm_halo_recv) TMHIP_CHECK(hipMalloc((void **)&ctx->swpm_halo_recv, hb));
    LexGeom g{ctx->g.T, ctx->g.LX, ctx->g.LY, ctx->g.LZ, ctx->V, 1};
    hipLaunchKernelGGL(sw_pack_slabs_kernel, dim3((unsigned)((XYZ + 255) / 256), 2), dim3(256), 0, ctx->stream, ctx->swpm_halo_send, (const v2d *)ctx->sw_ins, ws, g, ctx->Vh);
    TMHIP_CHECK(hipGetLastError());
  }
  return 0;
}
static int sw_all_launch(tmhip_ctx *ctx, double kappa, double c_sw) {
  const SwOrder plain = {0, 0, 0, 0, 0, 0, 0, 0};
  LexGeom g{ctx->g.T, ctx->g.LX, ctx->g.LY, ctx->g.LZ, ctx->V, ctx->g.nproc_t > 1 ? 1 : 0};
  const double c = -2. * (kappa * c_sw / 8.);
  const bool split = ctx->g.nproc_t > 1;
  const int ib = split ? ctx->face : 0, ie = split ? ctx->Vh - ctx->face : ctx->Vh;      // sites whose plaquettes stay on this rank
  if (ie > ib && (!split || ctx->g.T > 2)) {
    const SwFastLd ld{ctx->gauge, (unsigned)ctx->gs, ctx->sw_ins, (unsigned)ctx->V, ctx->g.T, ctx->g.LX, ctx->g.LY, ctx->g.LZ, ctx->Vh};
    int chunk = ((ie - ib + 63) / 64 + 7) / 8, slab = 0, nbt = 0, grid = chunk * 16;
    if (ctx->opt_swall_order != 0 && ctx->face % 64 == 0 && ctx->face / 64 >= 8) {      // slab order (32^4: 1.92 vs 2.01 ms); fallback of the tile order
      nbt = ctx->face / 64; slab = (nbt + 7) / 8;
      grid = 8 * slab * ((ie - ib) / ctx->face) * 2;
    }
    SwOrder ord = plain;
    if (ctx->opt_swall_order >= 2) { const int gt = sw_tile_order(ctx, ib, ie, ctx->opt_swall_order >= 4 ? ctx->opt_swall_order : 4, &ord); if (gt) grid = gt; }
    hipLaunchKernelGGL((sw_all_gather_kernel<SwFastLd>), dim3(grid), dim3(256), 0, ctx->stream, ld, ctx->deriv, ctx->g.LX,
                       ctx->g.LY, ctx->g.LZ, ctx->Vh, ib, ie, chunk, slab, nbt, c, ord);
  }
  if (split) {     // the two t-faces, after the neighbours' swm / swp slices have arrived (same stream)
    const SwEdgeLd ld{(const v2d *)ctx->gauge_raw, (const v2d *)ctx->sw_ins, (unsigned)ctx->V, (const v2d *)ctx->swpm_halo_recv, g, ctx->Vh};
    const int nface = ctx->g.T > 1 ? 2 : 1;
    for (int w = 0; w < nface; w++) {
      const int fb = w ? ctx->Vh - ctx->face : 0;
      const int chunk = ((ctx->face + 63) / 64 + 7) / 8;
      hipLaunchKernelGGL((sw_all_gather_kernel<SwEdgeLd>), dim3(chunk * 16), dim3(256), 0, ctx->stream, ld, ctx->deriv, ctx->g.LX,
                         ctx->g.LY, ctx->g.LZ, ctx->Vh, fb, fb + ctx->face, chunk, 0, 0, c, plain);
    }
  }
  TMHIP_CHECK(hipGetLastError());
  return 0;
}
/* operator/clover_accumulate_deriv.c:58 sw_all(hf, kappa, c_sw): adds the clover-leaf derivatives to the device-resident derivative
 * field (the one tmhip_deriv_Sb accumulates into).  gauge_field: host links as for tmhip_set_gauge (with the halo slabs on a
 * T-split rank), or NULL to reuse the links resident in HBM.  Every link is owned by one thread that gathers its 24 contributions
 * (sw_all_gather_kernel).  On T-split ranks the plaquettes next to the t-faces contain links and insertion matrices of both ring
 * neighbours: the links are in the halo slabs, the neighbours' t-slices of swm / swp are exchanged over RCCL first (what
 * xchange_deri.c does for the derivative in one direction only, :88-89, is not needed: nothing is computed for foreign links). */
int tmhip_sw_all(tmhip_ctx *ctx, const void *gauge_host, double kappa, double c_sw) {
  if (sw_all_prepare(ctx, gauge_host)) return 1;
  const int np = ctx->g.nproc_t, up = (ctx->g.proc_t + 1) % np, dn = (ctx->g.proc_t + np - 1) % np;
  if (np > 1 && !ctx->comm_ready) TMHIP_FAIL("nproc_t > 1 but tmhip_comm_init was not called");
  if (np > 1) {
    const size_t n = (size_t)2 * 30 * ctx->g.LX * ctx->g.LY * ctx->g.LZ;   // doubles per slice of insertion matrices
    double *snd = (double *)ctx->swpm_halo_send, *rcv = (double *)ctx->swpm_halo_recv;
    if (ctx->shm) { if (tmhip_shm_ring(ctx, ctx->stream, snd, snd + n, rcv, rcv + n, n * sizeof(double))) return 1; }
    else {
    TMHIP_NCCL_CHECK(ncclGroupStart());
    TMHIP_NCCL_CHECK(ncclSend(snd, n, ncclDouble, dn, ctx->comm_red, ctx->stream));            // our t = 0 slice is the down neighbour's t = T
    TMHIP_NCCL_CHECK(ncclSend(snd + n, n, ncclDouble, up, ctx->comm_red, ctx->stream));        // our t = T-1 slice is the up neighbour's t = -1
    TMHIP_NCCL_CHECK(ncclRecv(rcv, n, ncclDouble, up, ctx->comm_red, ctx->stream));            // slab 0 = t = T
    TMHIP_NCCL_CHECK(ncclRecv(rcv + n, n, ncclDouble, dn, ctx->comm_red, ctx->stream));        // slab 1 = t = -1
    TMHIP_NCCL_CHECK(ncclGroupEnd());
    }
  }
  return sw_all_launch(ctx, kappa, c_sw);
}
/* The same on a T-split lattice held by n contexts of one process (peer copies instead of RCCL), as tmhip_multi_deriv_Sb. */
int tmhip_multi_sw_all(int n, tmhip_ctx **ctxs, double kappa, double c_sw) {
  if (n < 2) TMHIP_FAIL("tmhip_multi_sw_all needs >= 2 contexts");
  for (int r = 0; r < n; r++) {
    tmhip_ctx *c = ctxs[r];
    if (c->g.nproc_t != n || c->g.proc_t != r) TMHIP_FAIL("context %d is not rank %d of a %d-way T split", r, r, n);
    if (sw_all_prepare(c, nullptr)) return 1;
  }
  for (int r = 0; r < n; r++) { TMHIP_CHECK(hipSetDevice(ctxs[r]->device)); TMHIP_CHECK(hipStreamSynchronize(ctxs[r]->stream)); }
  const size_t XYZ = (size_t)ctxs[0]->g.LX * ctxs[0]->g.LY * ctxs[0]->g.LZ;
  const size_t sb = (size_t)30 * XYZ * sizeof(v2d);
  for (int r = 0; r < n; r++) {
    tmhip_ctx *c = ctxs[r], *up = ctxs[(r + 1) % n], *dn = ctxs[(r + n - 1) % n];
    TMHIP_CHECK(hipSetDevice(c->device));
    TMHIP_CHECK(hipMemcpyPeerAsync(c->swpm_halo_recv, c->device, up->swpm_halo_send, up->device, sb, c->stream));                                  // t = T  <- up's t = 0
    TMHIP_CHECK(hipMemcpyPeerAsync((char *)c->swpm_halo_recv + sb, c->device, (char *)dn->swpm_halo_send + sb, dn->device, sb, c->stream));        // t = -1 <- down's t = T-1
    if (sw_all_launch(c, kappa, c_sw)) return 1;
  }
  for (int r = 0; r < n; r++) { TMHIP_CHECK(hipSetDevice(ctxs[r]->device)); TMHIP_CHECK(hipStreamSynchronize(ctxs[r]->stream)); }
  return 0;
}
/* swm / swp in the reference's host layout: su3 [VOLUME][4] each (either pointer may be NULL) */
__global__ __launch_bounds__(256) void swpm_unsort_kernel(v2d *__restrict__ out, const v2d *__restrict__ swpm, int which, int Vh, int V, int LX, int LY, int LZ) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Vh) return;
  const int par = blockIdx.y;
  const int LZh = LZ / 2;
  int r = i / LZh;
  const int y = r % LY;
  r /= LY;
  const int x = r % LX, t = r / LX;
  const int o = (t + x + y + par) & 1;
  v2d *dst = out + (2 * (size_t)i + o) * 36;
  const size_t s = (size_t)par * Vh + i;
#pragma unroll 4
  for (int e = 0; e < 36; e++) dst[e] = swpm[((size_t)which * 36 + e) * V + s];
}
int tmhip_get_swpm(tmhip_ctx *ctx, void *swm_host, void *swp_host) {
  if (!ctx->swpm) TMHIP_FAIL("tmhip_get_swpm: no clover-force accumulators on the device");
  TMHIP_CHECK(hipSetDevice(ctx->device));
  const size_t bytes = (size_t)ctx->V * 36 * sizeof(v2d);
  if (tmhip_stage_reserve(ctx, bytes)) return 1;
  void *hosts[2] = {swm_host, swp_host};
  for (int w = 0; w < 2; w++) {
    if (!hosts[w]) continue;
    hipLaunchKernelGGL(swpm_unsort_kernel, dim3((ctx->Vh + 255) / 256, 2), dim3(256), 0, ctx->stream, (v2d *)ctx->stage, (const v2d *)ctx->swpm, w, ctx->Vh,
                       ctx->V, ctx->g.LX, ctx->g.LY, ctx->g.LZ);
    TMHIP_CHECK(hipMemcpyAsync(hosts[w], ctx->stage, bytes, hipMemcpyDeviceToHost, ctx->stream));
    TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  }
  TMHIP_CHECK(hipGetLastError());
  return 0;
}

/* clovertm_operators.c:287-350 */
int tmhip_clover_inv(tmhip_ctx *ctx, tmhip_field *l, int tau3sign, double mu) {
  if (need64(l, "clover_inv")) return 1;
  if (!ctx->clover_set) TMHIP_FAIL("clover_inv called before tmhip_set_clover");
  const v2d *wi = swinv(ctx, tau3sign, mu);
  if (!wi) return 1;
  hipLaunchKernelGGL(clover_site_kernel<0>, dim3((ctx->Vh + 255) / 256), dim3(256), 0, ctx->stream, l->d, (const v2d *)l->d, (const v2d *)nullptr,
                     wi, l->ns, ctx->gs, ctx->Vh, 0.0);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}
/* clovertm_operators.c:448-520 */
int tmhip_clover_gamma5(tmhip_ctx *ctx, int ieo, tmhip_field *l, tmhip_field *k, tmhip_field *j, double mu) {
  if (need64(l, "clover_gamma5") || need64(k, "clover_gamma5") || need64(j, "clover_gamma5")) return 1;
  if (!ctx->clover_set) TMHIP_FAIL("clover_gamma5 called before tmhip_set_clover");
  hipLaunchKernelGGL(clover_site_kernel<1>, dim3((ctx->Vh + 255) / 256), dim3(256), 0, ctx->stream, l->d, (const v2d *)k->d, (const v2d *)j->d,
                     swpar(ctx, ieo), l->ns, ctx->gs, ctx->Vh, mu);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}
/* clovertm_operators.c:535-600 */
int tmhip_clover(tmhip_ctx *ctx, int ieo, tmhip_field *l, tmhip_field *k, tmhip_field *j, double mu) {
  if (need64(l, "clover") || need64(k, "clover") || need64(j, "clover")) return 1;
  if (!ctx->clover_set) TMHIP_FAIL("clover called before tmhip_set_clover");
  hipLaunchKernelGGL(clover_site_kernel<2>, dim3((ctx->Vh + 255) / 256), dim3(256), 0, ctx->stream, l->d, (const v2d *)k->d, (const v2d *)j->d,
                     swpar(ctx, ieo), l->ns, ctx->gs, ctx->Vh, mu);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}
/* clovertm_operators.c:268-272 */
int tmhip_H_eo_sw_inv_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k, int ieo, int tau3sign, double mu) {
  if (need64(l, "H_eo_sw_inv_psi") || need64(k, "H_eo_sw_inv_psi")) return 1;
  if (!ctx->clover_set) TMHIP_FAIL("H_eo_sw_inv_psi called before tmhip_set_clover");
  return tmhip_launch_hopping(ctx, ieo, l->d, k->d, nullptr, EPI_CLOVER_INV, 0, 0, true, swinv(ctx, tau3sign, mu));
}
/* clovertm_operators.c:233-245: 4 stencil launches with the clover blocks applied in the epilogues; the odd-odd term twists
 * with mu + mu3 (:238,243), the even-even inverse is the one sw_invert built for mu */
int tmhip_Qsw_pm_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k) {
  if (need64(l, "Qsw_pm_psi") || need64(k, "Qsw_pm_psi")) return 1;
  if (!ctx->clover_set) TMHIP_FAIL("Qsw_pm_psi called before tmhip_set_clover");
  const double mu = ctx->mu, muo = ctx->mu + ctx->mu3;
  v2d *s0 = ctx->scratch[0]->d, *s1 = ctx->scratch[1]->d;
  return tmhip_launch_hopping(ctx, TMHIP_EO, s1, k->d, nullptr, EPI_CLOVER_INV, 0, 0, true, swinv(ctx, -1, mu)) ||
         tmhip_launch_hopping(ctx, TMHIP_OE, s0, s1, k->d, EPI_CLOVER_G5, 0, -muo, true, swpar(ctx, TMHIP_OE)) ||
         tmhip_launch_hopping(ctx, TMHIP_EO, s1, s0, nullptr, EPI_CLOVER_INV, 0, 0, true, swinv(ctx, +1, mu)) ||
         tmhip_launch_hopping(ctx, TMHIP_OE, l->d, s1, s0, EPI_CLOVER_G5, 0, +muo, true, swpar(ctx, TMHIP_OE));
}
/* clovertm_operators.c:256-261 */
int tmhip_Msw_plus_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k) {
  if (need64(l, "Msw_plus_psi") || need64(k, "Msw_plus_psi")) return 1;
  if (!ctx->clover_set) TMHIP_FAIL("Msw_plus_psi called before tmhip_set_clover");
  const double mu = ctx->mu;
  v2d *s1 = ctx->scratch[1]->d;
  return tmhip_launch_hopping(ctx, TMHIP_EO, s1, k->d, nullptr, EPI_CLOVER_INV, 0, 0, true, swinv(ctx, +1, mu)) ||
         tmhip_launch_hopping(ctx, TMHIP_OE, l->d, s1, k->d, EPI_CLOVER, 0, +(mu + ctx->mu3), true, swpar(ctx, TMHIP_OE));   /* :258 */
}
/* The rest of the e/o clover family (clovertm_operators.c:201-268), two launches each, clover blocks in the epilogues:
 * which = 0: mu = 0 in the diagonal term (Qsw_psi / Msw_psi), +-1: Qsw_plus/minus_psi, Msw_plus/minus_psi.  l may alias k
 * (invert_clover_eo.c:128 calls Qm(Odd_new, Odd_new)): k enters the last launch only through the element-wise epilogue. */
static int sw_hat(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k, int which, int epi, const char *who) {
  if (need64(l, who) || need64(k, who)) return 1;
  if (!ctx->clover_set) TMHIP_FAIL("%s called before tmhip_set_clover / tmhip_sw_invert", who);
  const double mu = ctx->mu;
  v2d *s1 = ctx->scratch[1]->d;
  return tmhip_launch_hopping(ctx, TMHIP_EO, s1, k->d, nullptr, EPI_CLOVER_INV, 0, 0, true, swinv(ctx, which < 0 ? -1 : +1, mu)) ||
         tmhip_launch_hopping(ctx, TMHIP_OE, l->d, s1, k->d, epi, 0, which * (mu + ctx->mu3), true, swpar(ctx, TMHIP_OE));   /* +-(g_mu + g_mu3), :208,216,265 */
}
int tmhip_Qsw_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k) { return sw_hat(ctx, l, k, 0, EPI_CLOVER_G5, "Qsw_psi"); }               /* :201-206 */
int tmhip_Qsw_minus_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k) { return sw_hat(ctx, l, k, -1, EPI_CLOVER_G5, "Qsw_minus_psi"); }  /* :209-214 */
int tmhip_Qsw_plus_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k) { return sw_hat(ctx, l, k, +1, EPI_CLOVER_G5, "Qsw_plus_psi"); }    /* :217-222 */
int tmhip_Msw_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k) { return sw_hat(ctx, l, k, 0, EPI_CLOVER, "Msw_psi"); }                  /* :247-252 */
int tmhip_Msw_minus_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k) { return sw_hat(ctx, l, k, -1, EPI_CLOVER, "Msw_minus_psi"); }     /* :261-266 */
/* :225-237 : (Qsw_psi)^2 */
int tmhip_Qsw_sq_psi(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k) {
  return sw_hat(ctx, ctx->scratch[0], k, 0, EPI_CLOVER_G5, "Qsw_sq_psi") || sw_hat(ctx, l, ctx->scratch[0], 0, EPI_CLOVER_G5, "Qsw_sq_psi");
}
/* assign_mul_one_sw_pm_imu_inv_block_body.c:1-72 : k = (1 + T + i mu g5) l on the sites of parity ieo */
int tmhip_assign_mul_one_sw_pm_imu(tmhip_ctx *ctx, int ieo, tmhip_field *k, tmhip_field *l, double mu) {
  if (need64(k, "assign_mul_one_sw_pm_imu") || need64(l, "assign_mul_one_sw_pm_imu")) return 1;
  if (!ctx->sw_set) TMHIP_FAIL("assign_mul_one_sw_pm_imu called before tmhip_sw_term / tmhip_set_clover");
  hipLaunchKernelGGL(clover_site_kernel<2>, dim3((ctx->Vh + 255) / 256), dim3(256), 0, ctx->stream, k->d, (const v2d *)l->d, (const v2d *)nullptr,
                     swpar(ctx, ieo), k->ns, ctx->gs, ctx->Vh, mu);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}
/* assign_mul_one_sw_pm_imu_inv_block_body.c:143-196 : k = sw_inv(+mu set) l; ieo and mu are ignored exactly as in the reference */
int tmhip_assign_mul_one_sw_pm_imu_inv(tmhip_ctx *ctx, int ieo, tmhip_field *k, tmhip_field *l, double mu) {
  (void)ieo; (void)mu;
  if (need64(k, "assign_mul_one_sw_pm_imu_inv") || need64(l, "assign_mul_one_sw_pm_imu_inv")) return 1;
  if (!ctx->clover_set) TMHIP_FAIL("assign_mul_one_sw_pm_imu_inv called before tmhip_set_clover / tmhip_sw_invert");
  hipLaunchKernelGGL(clover_site_kernel<0>, dim3((ctx->Vh + 255) / 256), dim3(256), 0, ctx->stream, k->d, (const v2d *)l->d, (const v2d *)nullptr,
                     (const v2d *)ctx->sw_inv, k->ns, ctx->gs, ctx->Vh, 0.0);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}
/* clovertm_operators.c:96-110 : X_new = (1 + T + i mu g5) X - H Y on both parities, clover blocks in the stencil epilogue */
int tmhip_Msw_full(tmhip_ctx *ctx, tmhip_field *En, tmhip_field *On, tmhip_field *E, tmhip_field *O) {
  if (need64(En, "Msw_full") || need64(On, "Msw_full") || need64(E, "Msw_full") || need64(O, "Msw_full")) return 1;
  if (!ctx->sw_set) TMHIP_FAIL("Msw_full called before tmhip_sw_term / tmhip_set_clover");
  return tmhip_launch_hopping(ctx, TMHIP_EO, En->d, O->d, E->d, EPI_CLOVER, 0, +ctx->mu, true, swpar(ctx, TMHIP_EO)) ||
         tmhip_launch_hopping(ctx, TMHIP_OE, On->d, E->d, O->d, EPI_CLOVER, 0, +ctx->mu, true, swpar(ctx, TMHIP_OE));
}
/* clovertm_operators_32.c Qsw_pm_psi_32 */
int tmhip_Qsw_pm_psi_32(tmhip_ctx *ctx, tmhip_field *l, tmhip_field *k) {
  if (!l || !k || l->prec != 1 || k->prec != 1) TMHIP_FAIL("Qsw_pm_psi_32 needs fp32 fields");
  if (tmhip_prepare_clover32(ctx)) return 1;
  const double mu = ctx->mu;
  const size_t gs = ctx->gs;
  const v2f *wim = ctx->sw_inv32 + (size_t)(fabs(mu) > 0 ? 1 : 0) * 72 * gs, *wip = ctx->sw_inv32, *wo = ctx->sw32 + (size_t)54 * gs;
  v2f *s0 = ctx->scratch32[0]->d32, *s1 = ctx->scratch32[1]->d32;
  return tmhip_launch_hopping32(ctx, TMHIP_EO, s1, k->d32, nullptr, EPI_CLOVER_INV, 0, 0, true, wim) ||
         tmhip_launch_hopping32(ctx, TMHIP_OE, s0, s1, k->d32, EPI_CLOVER_G5, 0, -(mu + ctx->mu3), true, wo) ||
         tmhip_launch_hopping32(ctx, TMHIP_EO, s1, s0, nullptr, EPI_CLOVER_INV, 0, 0, true, wip) ||
         tmhip_launch_hopping32(ctx, TMHIP_OE, l->d32, s1, s0, EPI_CLOVER_G5, 0, +(mu + ctx->mu3), true, wo);
}

}  // extern "C"
