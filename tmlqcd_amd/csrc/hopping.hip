// Even/odd Wilson twisted-mass hopping stencil for gfx950 (MI355X), fp64.
//
// Computes, for every site x of parity ieo,
//   l(x) = sum_mu [ ka_mu U_mu(x) (1+g_mu) k(x+mu) + conj(ka_mu) U_mu(x-mu)^dag (1-g_mu) k(x-mu) ]
// exactly as the reference's generic body (operator/hopping_body_dbl.c:27-181 with the
// macros of operator/hopping.h:574-694), optionally followed by the fused epilogues of
// tm_times_Hopping_Matrix (hopping.h:674-678) and tm_sub_Hopping_Matrix (hopping.h:680-688).
//
// MI355X mapping:
//  * one thread per output site, lanes consecutive in the e/o sub-index -> every global
//    load is a 16 B/lane, 1 KiB/wave coalesced read of one SoA plane (12 spinor planes,
//    72 gauge planes); the z/y/x/t neighbours of a wave are contiguous runs of the same
//    planes, so the 8-fold spinor re-use is served by L1/L2/Infinity Cache and HBM sees
//    each input spinor ~once.
//  * neighbour indices are computed arithmetically from (t,x,y,k): the reference's
//    g_hi gather table (64 B/site, geometry_eo.c:1470-1535) is never read.
//  * gauge links are used exactly once per call -> loaded non-temporally so they do
//    not evict the re-used spinor lines from L2.
//  * MFMA is not used: 3x3 complex mat-vec at 1 flop/B, HBM-bound by ~10x.
#include "tmhip_internal.h"

struct HopArgs {
  v2d *out;
  const v2d *in;
  const v2d *p;
  const v2d *gauge;  // already offset to the parity of the output sites
  const v2d *halo_up, *halo_dn;
  const v2d *dotv;   // EPI_TM_SUB_G5_DOT: field whose real scalar product with the output is accumulated
  double *partials;  // EPI_TM_SUB_G5_DOT: one partial per block
  int ns, gs;
  int T, LX, LY, LZh;
  int Vh, face, YZh;
  int i_begin, i_end;
  int par_off;  // (proc_t*T + ieo) & 1
  int nxcd_chunk;  // >0: XCD-aware block remap, blocks per XCD chunk
  int map_tc;      // >0: within an XCD chunk walk t fastest over map_tc time-slices (tile order)
  int map_bpt;     // blocks per time-slice (face / BS) for the tile order
  int map_nb;      // real number of blocks of this launch (the grid is padded to a multiple of 8)
  int nb_int_grid;           // TFACE 4: grid blocks [0, nb_int_grid) are interior, the rest walk the two t-faces
  int face_bpb;              // TFACE 4: blocks per face
  const unsigned int *halo_flag; unsigned int halo_seq; unsigned int *err_flag;  // TFACE 4: faces are valid once *halo_flag >= halo_seq
  unsigned gauge_bytes;      // experiment (GAUX >= 0): size of the gauge buffer descriptor; 0 drops every gauge load
  int shape_bx, shape_by;  // >1: a block covers shape_bx x-planes x shape_by y-rows x all k (instead of BS consecutive sites)
  double ka[4][2];
  double cre, cim;
};

__device__ __forceinline__ v2d cmul(v2d a, v2d b) { return v2d{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ v2d cmulc(v2d a, v2d b) {  // conj(a) * b
  return v2d{a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ v2d cfma(v2d a, v2d b, v2d c) {  // c + a*b
  return v2d{c.x + a.x * b.x - a.y * b.y, c.y + a.x * b.y + a.y * b.x};
}
__device__ __forceinline__ v2d cfmac(v2d a, v2d b, v2d c) {  // c + conj(a)*b
  return v2d{c.x + a.x * b.x + a.y * b.y, c.y + a.x * b.y - a.y * b.x};
}

template <bool NT>
__device__ __forceinline__ v2d ldg(const v2d *p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}

template <bool NT>
__device__ __forceinline__ void stg(v2d *p, v2d v) {
  if (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// One of the 8 hops.  D = 2*mu + (0: +mu, 1: -mu), mu = t,x,y,z.
// HALO: the projected half-spinor comes from an exchanged face buffer [6][face] instead of `in`.
typedef int v4i __attribute__((ext_vector_type(4)));

template <int D, bool HALO, bool NT, int GAUX = -1>
__device__ __forceinline__ void hop_dir(v2d (&acc)[12], const v2d *__restrict__ in, int ns, int j,
                                        const v2d *__restrict__ halo, int face,
                                        const v2d *__restrict__ g, size_t gs, int i, v2d ka,
                                        __amdgpu_buffer_rsrc_t rsrc = __amdgpu_buffer_rsrc_t()) {
  v2d pa[3], pb[3];
  if (HALO) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      pa[c] = halo[(size_t)c * face + j];
      pb[c] = halo[(size_t)(3 + c) * face + j];
    }
  } else {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const v2d s0 = in[(size_t)(0 + c) * ns + j], s1 = in[(size_t)(3 + c) * ns + j];
      const v2d s2 = in[(size_t)(6 + c) * ns + j], s3 = in[(size_t)(9 + c) * ns + j];
      if (D == 0) { pa[c] = s0 + s2; pb[c] = s1 + s3; }                                           // hopping.h:579,584
      if (D == 1) { pa[c] = s0 - s2; pb[c] = s1 - s3; }                                           // hopping.h:591,596
      if (D == 2) { pa[c] = v2d{s0.x - s3.y, s0.y + s3.x}; pb[c] = v2d{s1.x - s2.y, s1.y + s2.x}; }  // s0+i s3, s1+i s2
      if (D == 3) { pa[c] = v2d{s0.x + s3.y, s0.y - s3.x}; pb[c] = v2d{s1.x + s2.y, s1.y - s2.x}; }  // s0-i s3, s1-i s2
      if (D == 4) { pa[c] = s0 + s3; pb[c] = s1 - s2; }                                           // hopping.h:627,632
      if (D == 5) { pa[c] = s0 - s3; pb[c] = s1 + s2; }                                           // hopping.h:639,644
      if (D == 6) { pa[c] = v2d{s0.x - s2.y, s0.y + s2.x}; pb[c] = v2d{s1.x + s3.y, s1.y - s3.x}; }  // s0+i s2, s1-i s3
      if (D == 7) { pa[c] = v2d{s0.x + s2.y, s0.y - s2.x}; pb[c] = v2d{s1.x - s3.y, s1.y + s3.x}; }  // s0-i s2, s1+i s3
    }
  }
  const v2d *gd = g + (size_t)D * 9 * gs + i;
  v2d u[9];
  if (GAUX < 0) {
#pragma unroll
    for (int e = 0; e < 9; e++) u[e] = ldg<NT>(gd + (size_t)e * gs);
  } else {
    // experiment: gauge links through a buffer descriptor with an explicit cache policy (aux: 1 sc0, 2 nt, 16 sc1)
#pragma unroll
    for (int e = 0; e < 9; e++) {
      const unsigned off = (unsigned)((((size_t)(D * 9 + e)) * gs + i) * sizeof(v2d));
      const v4i raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, GAUX);
      u[e] = __builtin_bit_cast(v2d, raw);
    }
  }
  v2d ca[3], cb[3];
  if ((D & 1) == 0) {  // chi = U psi           (su3.h:308-311)
#pragma unroll
    for (int r = 0; r < 3; r++) {
      ca[r] = cfma(u[3 * r + 2], pa[2], cfma(u[3 * r + 1], pa[1], cmul(u[3 * r], pa[0])));
      cb[r] = cfma(u[3 * r + 2], pb[2], cfma(u[3 * r + 1], pb[1], cmul(u[3 * r], pb[0])));
    }
#pragma unroll
    for (int r = 0; r < 3; r++) { ca[r] = cmul(ka, ca[r]); cb[r] = cmul(ka, cb[r]); }
  } else {  // chi = U^dagger psi     (su3.h:313-316)
#pragma unroll
    for (int r = 0; r < 3; r++) {
      ca[r] = cfmac(u[6 + r], pa[2], cfmac(u[3 + r], pa[1], cmulc(u[r], pa[0])));
      cb[r] = cfmac(u[6 + r], pb[2], cfmac(u[3 + r], pb[1], cmulc(u[r], pb[0])));
    }
#pragma unroll
    for (int r = 0; r < 3; r++) { ca[r] = cmulc(ka, ca[r]); cb[r] = cmulc(ka, cb[r]); }
  }
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const v2d a = ca[c], b = cb[c];
    acc[c] += a;      // s0 += a
    acc[3 + c] += b;  // s1 += b
    if (D == 0) { acc[6 + c] += a; acc[9 + c] += b; }
    if (D == 1) { acc[6 + c] -= a; acc[9 + c] -= b; }
    if (D == 2) { acc[9 + c] += v2d{a.y, -a.x}; acc[6 + c] += v2d{b.y, -b.x}; }   // s3 -= i a ; s2 -= i b
    if (D == 3) { acc[9 + c] += v2d{-a.y, a.x}; acc[6 + c] += v2d{-b.y, b.x}; }   // s3 += i a ; s2 += i b
    if (D == 4) { acc[9 + c] += a; acc[6 + c] -= b; }
    if (D == 5) { acc[9 + c] -= a; acc[6 + c] += b; }
    if (D == 6) { acc[6 + c] += v2d{a.y, -a.x}; acc[9 + c] += v2d{-b.y, b.x}; }   // s2 -= i a ; s3 += i b
    if (D == 7) { acc[6 + c] += v2d{-a.y, a.x}; acc[9 + c] += v2d{b.y, -b.x}; }   // s2 += i a ; s3 -= i b
  }
}

// TFACE: 0 = t-neighbours are local (interior, or unsplit lattice with periodic wrap)
//        1 = sites of the t=0 slab:   -t half-spinors come from halo_dn
//        2 = sites of the t=T-1 slab: +t half-spinors come from halo_up
//        3 = both slabs in one launch (block-uniform choice between 1 and 2)
//        4 = interior AND both slabs in one launch; the face blocks wait in-kernel for the exchanged faces
template <int EPI, int TFACE, bool NTIO, int BS, int MINW, int GAUX = -1>
__global__ __launch_bounds__(BS, MINW) void hop_kernel(const HopArgs a) {
  constexpr bool NT = true;  // gauge links: used once per call -> non-temporal (measured 0.19 -> 0.16 ms at 32^4)
  int bid = blockIdx.x;
  int tf = (TFACE == 4) ? 0 : TFACE;
  int i = 0;
  bool face_block = false;
  if (TFACE == 4 && bid >= a.nb_int_grid) {
    // Face blocks of the single-launch split-phase kernel.  They carry the highest block ids, so they are
    // dispatched after the interior blocks; the exchanged half-spinors are published by the comm stream
    // through *halo_flag (flag_set_kernel after the exchange).  Consumer side of the hand-off
    // (cdna guide G16): one lane polls relaxed, then ONE agent-scope acquire, vmcnt(0), barrier, plain loads.
    // The face buffers are read nowhere else in this kernel, so no stale copy can sit in this XCD's L2.
    if (threadIdx.x == 0) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
      while ((int)(__hip_atomic_load(a.halo_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.halo_seq) < 0) {
        __builtin_amdgcn_s_sleep(4);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 300000000ull) {  // 3 s: give up, flag the error, never hang the GPU
          __hip_atomic_store(a.err_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const int bb = bid - a.nb_int_grid;
    const int second = bb >= a.face_bpb;
    const int jl = (bb - (second ? a.face_bpb : 0)) * BS + threadIdx.x;
    if (jl >= a.face) return;
    tf = second ? 2 : 1;
    i = second ? a.Vh - a.face + jl : jl;
    face_block = true;
  } else {
  if (a.nxcd_chunk > 0) {
    // XCD-aware remap: blocks b, b+8, b+16.. share an XCD (and its L2); give each XCD a
    // contiguous chunk of the lattice so neighbouring tiles hit the same L2.
    const int xcd = bid & 7;
    int q = bid >> 3;
    bid = xcd * a.nxcd_chunk + q;
    if (bid >= a.map_nb) {
      if (EPI == EPI_TM_SUB_G5_DOT && threadIdx.x == 0) a.partials[blockIdx.x] = 0.0;
      return;
    }
    if (a.map_tc > 0) {
      // tile order: time-slices are taken in groups of map_tc; inside a group the same spatial
      // tile at t, t+1, .. is dispatched back to back, so the +-t (and +-x) users of an input
      // line run close in time on the same XCD.
      const int per_group = a.map_tc * a.map_bpt;
      const int g = bid / per_group, rem = bid - g * per_group;
      const int sp = rem / a.map_tc, tl = rem - sp * a.map_tc;
      bid = (g * a.map_tc + tl) * a.map_bpt + sp;
    }
  }
  i = a.i_begin + bid * BS + threadIdx.x;
  if (TFACE == 0 && a.shape_bx > 1) {
    // compact block shape: fewer neighbour rows fall outside the block's own footprint
    const int tt = bid / a.map_bpt, sp = bid - tt * a.map_bpt;
    const int nyb = a.LY / a.shape_by;
    const int xb = sp / nyb, yb = sp - xb * nyb;
    const int rr = threadIdx.x / a.LZh, kk = threadIdx.x - rr * a.LZh;
    const int dx = rr / a.shape_by, dy = rr - dx * a.shape_by;
    i = ((tt * a.LX + xb * a.shape_bx + dx) * a.LY + yb * a.shape_by + dy) * a.LZh + kk;
  }
  }  // !face block
  if (face_block) {
    // index already set
  } else if (TFACE == 3) {
    // both t-faces in one launch: blocks [0, map_bpt) walk the t=0 slab, [map_bpt, 2 map_bpt) the t=T-1 slab
    const int second = bid >= a.map_bpt;
    const int jl = (bid - (second ? a.map_bpt : 0)) * BS + threadIdx.x;
    if (jl >= a.face) return;
    tf = second ? 2 : 1;
    i = second ? a.Vh - a.face + jl : jl;
  } else if (i >= a.i_end) {
    return;  // (the fused-reduction variant is only launched on lattices with V/2 % BS == 0: no partial blocks)
  }

  const int LZh = a.LZh;
  const int k = i % LZh;
  int r = i / LZh;
  const int y = r % a.LY;
  r /= a.LY;
  const int x = r % a.LX;
  const int t = r / a.LX;
  const int o = (t + x + y + a.par_off) & 1;  // z = 2k + o  (geometry_eo.c:807-811)
  const int XYZh = a.face, YZh = a.YZh;

  const int jtp = (t + 1 < a.T) ? i + XYZh : i - (a.T - 1) * XYZh;
  const int jtm = (t > 0) ? i - XYZh : i + (a.T - 1) * XYZh;
  const int jxp = (x + 1 < a.LX) ? i + YZh : i - (a.LX - 1) * YZh;
  const int jxm = (x > 0) ? i - YZh : i + (a.LX - 1) * YZh;
  const int jyp = (y + 1 < a.LY) ? i + LZh : i - (a.LY - 1) * LZh;
  const int jym = (y > 0) ? i - LZh : i + (a.LY - 1) * LZh;
  const int jzp = o ? ((k + 1 < LZh) ? i + 1 : i - (LZh - 1)) : i;
  const int jzm = o ? i : ((k > 0) ? i - 1 : i + (LZh - 1));
  const int jf = i - t * XYZh;  // index inside a t-face

  v2d acc[12];
#pragma unroll
  for (int c = 0; c < 12; c++) acc[c] = v2d{0.0, 0.0};

  const v2d ka0 = v2d{a.ka[0][0], a.ka[0][1]}, ka1 = v2d{a.ka[1][0], a.ka[1][1]};
  const v2d ka2 = v2d{a.ka[2][0], a.ka[2][1]}, ka3 = v2d{a.ka[3][0], a.ka[3][1]};
  const v2d *__restrict__ in = a.in;
  const v2d *__restrict__ g = a.gauge;

  __amdgpu_buffer_rsrc_t rsrc = __amdgpu_buffer_rsrc_t();
  if (GAUX >= 0) rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<v2d *>(g), 0, a.gauge_bytes, 0x00020000);
  if (TFACE != 0 && tf == 2) hop_dir<0, true, NT, GAUX>(acc, in, a.ns, jf, a.halo_up, a.face, g, a.gs, i, ka0, rsrc);
  else                       hop_dir<0, false, NT, GAUX>(acc, in, a.ns, jtp, nullptr, 0, g, a.gs, i, ka0, rsrc);
  if (TFACE != 0 && tf == 1) hop_dir<1, true, NT, GAUX>(acc, in, a.ns, jf, a.halo_dn, a.face, g, a.gs, i, ka0, rsrc);
  else                       hop_dir<1, false, NT, GAUX>(acc, in, a.ns, jtm, nullptr, 0, g, a.gs, i, ka0, rsrc);
  hop_dir<2, false, NT, GAUX>(acc, in, a.ns, jxp, nullptr, 0, g, a.gs, i, ka1, rsrc);
  hop_dir<3, false, NT, GAUX>(acc, in, a.ns, jxm, nullptr, 0, g, a.gs, i, ka1, rsrc);
  hop_dir<4, false, NT, GAUX>(acc, in, a.ns, jyp, nullptr, 0, g, a.gs, i, ka2, rsrc);
  hop_dir<5, false, NT, GAUX>(acc, in, a.ns, jym, nullptr, 0, g, a.gs, i, ka2, rsrc);
  hop_dir<6, false, NT, GAUX>(acc, in, a.ns, jzp, nullptr, 0, g, a.gs, i, ka3, rsrc);
  hop_dir<7, false, NT, GAUX>(acc, in, a.ns, jzm, nullptr, 0, g, a.gs, i, ka3, rsrc);

  v2d *__restrict__ out = a.out;
  const v2d cf = v2d{a.cre, a.cim};
  if (EPI == EPI_STORE) {  // hopping.h:690-694
#pragma unroll
    for (int c = 0; c < 12; c++) stg<NTIO>(out + (size_t)c * a.ns + i, acc[c]);
  } else if (EPI == EPI_TM_TIMES) {  // hopping.h:674-678
#pragma unroll
    for (int c = 0; c < 6; c++) stg<NTIO>(out + (size_t)c * a.ns + i, cmul(cf, acc[c]));
#pragma unroll
    for (int c = 6; c < 12; c++) stg<NTIO>(out + (size_t)c * a.ns + i, cmulc(cf, acc[c]));
  } else {
    // EPI_TM_SUB_G5[_DOT]: hopping.h:680-688  l = g5[(cf,cf*) p - H k];  EPI_TM_SUB: same without g5
    const v2d *__restrict__ p = a.p;
    double d = 0.0;
#pragma unroll
    for (int c = 0; c < 12; c++) {
      const v2d pv = ldg<NTIO>(p + (size_t)c * a.ns + i);
      v2d r;
      if (c < 6) r = cmul(cf, pv) - acc[c];
      else { const v2d zp = cmulc(cf, pv); r = (EPI == EPI_TM_SUB) ? zp - acc[c] : acc[c] - zp; }
      stg<NTIO>(out + (size_t)c * a.ns + i, r);
      if (EPI == EPI_TM_SUB_G5_DOT) {
        // fused scalar_prod_r(dotv, out) of cg_her.c:93: saves re-reading `out` (and a launch) per CG iteration
        const v2d w = ldg<NTIO>(a.dotv + (size_t)c * a.ns + i);
        d += w.x * r.x + w.y * r.y;
      }
    }
    if (EPI == EPI_TM_SUB_G5_DOT) {
      __shared__ double wsum[BS / 64];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
      if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = d;
      __syncthreads();
      if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < BS / 64; w++) t += wsum[w];
        a.partials[blockIdx.x] = t;
      }
    }
  }
}

// Cross-stream ordering without HIP events: a one-thread kernel publishes a sequence number, a
// one-thread kernel on the other stream waits for it.  Data hand-off itself still happens at kernel
// boundaries (producer kernel complete before the flag kernel runs; consumer kernel starts after the
// wait kernel), so only the flag word needs agent-scope atomics.  The spin is bounded: on timeout an
// error word is set (checked by tmhip_sync) and the wave exits.
__global__ void flag_set_kernel(unsigned int *flag, unsigned int seq) {
  __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void flag_wait_kernel(const unsigned int *flag, unsigned int seq, unsigned int *err) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
  while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - seq) < 0) {
    __builtin_amdgcn_s_sleep(8);
    if (__builtin_amdgcn_s_memrealtime() - t0 > 300000000ull) {  // 3 s
      __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// Project the two t-faces of the input field to half-spinors for the neighbours
// (what xchange_halffield ships, xchange/xchange_halffield.c:199-255; projections of
// operator/halfspinor_hopping.h:1279-1293):
//   send_dn[j] = (s0+s2, s1+s3)(t=0)      -> down neighbour, consumed by its +t hop at t=T-1
//   send_up[j] = (s0-s2, s1-s3)(t=T-1)    -> up neighbour,   consumed by its -t hop at t=0
__global__ __launch_bounds__(256) void pack_faces_kernel(const v2d *__restrict__ in, int ns, int Vh, int face,
                                                         v2d *__restrict__ send_dn, v2d *__restrict__ send_up) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= face) return;
  const int which = blockIdx.y;  // 0: t=0 face -> send_dn, 1: t=T-1 face -> send_up
  const int i = which ? Vh - face + j : j;
  v2d *dst = which ? send_up : send_dn;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const v2d s0 = in[(size_t)(0 + c) * ns + i], s1 = in[(size_t)(3 + c) * ns + i];
    const v2d s2 = in[(size_t)(6 + c) * ns + i], s3 = in[(size_t)(9 + c) * ns + i];
    dst[(size_t)c * face + j] = which ? s0 - s2 : s0 + s2;
    dst[(size_t)(3 + c) * face + j] = which ? s1 - s3 : s1 + s3;
  }
}

struct HopLaunch { int block; bool ntio; int minw; int xcd; int occ; int tgrp; int shape; int gaux; int gdrop; };

template <int EPI, int TFACE, bool NTIO, int BS, int MINW, int GAUX = -1>
static void launch_one(const HopArgs &a, hipStream_t st, const HopLaunch &o, bool allow_map) {
  const int n = a.i_end - a.i_begin;
  if (n <= 0 && TFACE != 4) return;
  int nb = n > 0 ? (n + BS - 1) / BS : 0;
  HopArgs b = a;
  b.nxcd_chunk = 0; b.map_tc = 0; b.map_bpt = 0; b.map_nb = nb; b.shape_bx = 0; b.shape_by = 0;
  if (allow_map && o.xcd && nb >= 64) {
    const int chunk = (nb + 7) / 8;
    b.nxcd_chunk = chunk;
    // tile order needs whole time-slices in whole blocks: range = k time-slices, face % BS == 0
    if (o.xcd >= 2 && a.face % BS == 0 && a.i_begin % a.face == 0 && n % a.face == 0) {
      const int nt = n / a.face;  // time-slices in this launch (T, or T-2 for the interior of a split lattice)
      int grp = 0;
      for (int cand : {4, 5, 6, 3, 2}) if (nt % cand == 0) { grp = cand; break; }
      if (o.tgrp > 0 && nt % o.tgrp == 0) grp = o.tgrp;
      if (grp) { b.map_tc = grp; b.map_bpt = a.face / BS; }
      // optional compact block shape (bx x-planes x by y-rows x LZ/2): needs the full lattice and exact tiling
      if (grp && o.shape > 1 && a.i_begin == 0 && n == a.Vh && BS % (a.LZh * o.shape) == 0) {
        const int by = BS / (a.LZh * o.shape);
        if (by >= 1 && a.LX % o.shape == 0 && a.LY % by == 0) { b.shape_bx = o.shape; b.shape_by = by; }
      }
    }
    nb = chunk * 8;  // blocks past i_end exit immediately
  }
  if (TFACE == 4) {  // face blocks ride behind the interior blocks of the same launch
    b.nb_int_grid = nb;
    b.face_bpb = (a.face + BS - 1) / BS;
    nb += 2 * b.face_bpb;
  }
  // occupancy cap for A/B runs: dynamic LDS sized so that only `occ` waves per SIMD fit on a CU
  size_t lds = 0;
  if (o.occ > 0) {
    const int blocks_per_cu = o.occ * 4 * 64 / BS;
    lds = (size_t)(163840 / (blocks_per_cu > 0 ? blocks_per_cu : 1)) / 256 * 256;
    if (lds > 65536) lds = 65536;  // default dynamic-LDS limit without an attribute opt-in
  }
  if (o.gdrop) b.gauge_bytes = 0;
  hipLaunchKernelGGL((hop_kernel<EPI, TFACE, NTIO, BS, MINW, GAUX>), dim3(nb), dim3(BS), lds, st, b);
}

template <int EPI, int TFACE>
static void launch_variant(const HopArgs &a, hipStream_t st, const HopLaunch &o, bool allow_map) {
  if (EPI == EPI_STORE && TFACE == 0 && o.gaux >= 0 && o.block == 256) {  // cache-policy experiment on the plain stencil only
    switch (o.gaux) {
      case 0: launch_one<EPI_STORE, 0, true, 256, 1, 0>(a, st, o, allow_map); return;
      case 1: launch_one<EPI_STORE, 0, true, 256, 1, 1>(a, st, o, allow_map); return;
      case 2: launch_one<EPI_STORE, 0, true, 256, 1, 2>(a, st, o, allow_map); return;
      case 3: launch_one<EPI_STORE, 0, true, 256, 1, 3>(a, st, o, allow_map); return;
      case 16: launch_one<EPI_STORE, 0, true, 256, 1, 16>(a, st, o, allow_map); return;
      case 17: launch_one<EPI_STORE, 0, true, 256, 1, 17>(a, st, o, allow_map); return;
      case 18: launch_one<EPI_STORE, 0, true, 256, 1, 18>(a, st, o, allow_map); return;
      case 19: launch_one<EPI_STORE, 0, true, 256, 1, 19>(a, st, o, allow_map); return;
      default: break;
    }
  }
#define TMHIP_L(NTIO, BS, MINW) launch_one<EPI, TFACE, NTIO, BS, MINW>(a, st, o, allow_map)
  if (o.block == 64) {
    if (o.ntio) { if (o.minw >= 4) TMHIP_L(true, 64, 4); else TMHIP_L(true, 64, 1); }
    else        { if (o.minw >= 4) TMHIP_L(false, 64, 4); else TMHIP_L(false, 64, 1); }
  } else {
    if (o.ntio) { if (o.minw >= 4) TMHIP_L(true, 256, 4); else TMHIP_L(true, 256, 1); }
    else        { if (o.minw >= 4) TMHIP_L(false, 256, 4); else TMHIP_L(false, 256, 1); }
  }
#undef TMHIP_L
}

template <int TFACE>
static void launch_epi(const HopArgs &a, int epi, hipStream_t st, const HopLaunch &o, bool allow_map) {
  switch (epi) {
    case EPI_STORE: launch_variant<EPI_STORE, TFACE>(a, st, o, allow_map); break;
    case EPI_TM_TIMES: launch_variant<EPI_TM_TIMES, TFACE>(a, st, o, allow_map); break;
    case EPI_TM_SUB_G5: launch_variant<EPI_TM_SUB_G5, TFACE>(a, st, o, allow_map); break;
    case EPI_TM_SUB_G5_DOT: launch_variant<EPI_TM_SUB_G5_DOT, TFACE>(a, st, o, allow_map); break;
    default: launch_variant<EPI_TM_SUB, TFACE>(a, st, o, allow_map); break;
  }
}

static void fill_args(HopArgs &a, tmhip_ctx *ctx, int ieo, v2d *out, const v2d *in, const v2d *p, double cre, double cim) {
  a.out = out; a.in = in; a.p = p; a.dotv = nullptr; a.partials = nullptr;
  a.gauge_bytes = (unsigned)((size_t)72 * ctx->gs * sizeof(v2d));
  a.nb_int_grid = 0; a.face_bpb = 0; a.halo_flag = nullptr; a.halo_seq = 0; a.err_flag = nullptr;
  a.gauge = ctx->gauge + (size_t)(ieo ? 1 : 0) * 72 * ctx->gs;
  a.halo_up = ctx->recv_up; a.halo_dn = ctx->recv_dn;
  a.ns = ctx->ns; a.gs = ctx->gs;
  a.T = ctx->g.T; a.LX = ctx->g.LX; a.LY = ctx->g.LY; a.LZh = ctx->g.LZ / 2;
  a.Vh = ctx->Vh; a.face = ctx->face; a.YZh = ctx->g.LY * ctx->g.LZ / 2;
  a.par_off = (ctx->g.proc_t * ctx->g.T + ieo) & 1;
  a.nxcd_chunk = 0; a.map_tc = 0; a.map_bpt = 0; a.map_nb = 0; a.shape_bx = 0; a.shape_by = 0;
  for (int m = 0; m < 4; m++) { a.ka[m][0] = ctx->ka[m][0]; a.ka[m][1] = ctx->ka[m][1]; }
  a.cre = cre; a.cim = cim;
}

static void launch_pack(tmhip_ctx *ctx, const v2d *in) {
  hipLaunchKernelGGL(pack_faces_kernel, dim3((ctx->face + 255) / 256, 2), dim3(256), 0, ctx->stream,
                     in, ctx->ns, ctx->Vh, ctx->face, ctx->send_dn, ctx->send_up);
}

static void launch_interior(tmhip_ctx *ctx, HopArgs &a, int epi, const HopLaunch &o) {
  a.i_begin = ctx->face; a.i_end = ctx->Vh - ctx->face;
  launch_epi<0>(a, epi, ctx->stream, o, true);
}

static void launch_boundary(tmhip_ctx *ctx, HopArgs &a, int epi, const HopLaunch &o, hipStream_t st) {
  // one launch for both faces, 64-thread blocks: 2*face/64 small blocks spread over the whole chip and
  // interleave with the interior kernel's blocks instead of forming two serial latency-bound launches
  HopArgs b = a;
  b.i_begin = 0; b.i_end = ctx->Vh;
  b.nxcd_chunk = 0; b.map_tc = 0; b.shape_bx = 0;
  b.map_bpt = (ctx->face + 63) / 64;
  b.map_nb = 2 * b.map_bpt;
  const dim3 grid(b.map_nb), blk(64);
#define TMHIP_B(EPI) if (o.ntio) hipLaunchKernelGGL((hop_kernel<EPI, 3, true, 64, 1>), grid, blk, 0, st, b); \
                     else hipLaunchKernelGGL((hop_kernel<EPI, 3, false, 64, 1>), grid, blk, 0, st, b)
  switch (epi) {
    case EPI_STORE: TMHIP_B(EPI_STORE); break;
    case EPI_TM_TIMES: TMHIP_B(EPI_TM_TIMES); break;
    case EPI_TM_SUB_G5: TMHIP_B(EPI_TM_SUB_G5); break;
    default: TMHIP_B(EPI_TM_SUB); break;
  }
#undef TMHIP_B
}

static void launch_fused_faces(tmhip_ctx *ctx, HopArgs &a, int epi, const HopLaunch &o) {
  a.i_begin = ctx->face; a.i_end = ctx->Vh - ctx->face;
#define TMHIP_F(EPI) if (o.ntio) launch_one<EPI, 4, true, 256, 1>(a, ctx->stream, o, true); \
                     else launch_one<EPI, 4, false, 256, 1>(a, ctx->stream, o, true)
  switch (epi) {
    case EPI_STORE: TMHIP_F(EPI_STORE); break;
    case EPI_TM_TIMES: TMHIP_F(EPI_TM_TIMES); break;
    case EPI_TM_SUB_G5: TMHIP_F(EPI_TM_SUB_G5); break;
    default: TMHIP_F(EPI_TM_SUB); break;
  }
#undef TMHIP_F
}

int tmhip_launch_hopping(tmhip_ctx *ctx, int ieo, v2d *out, const v2d *in, const v2d *p, int epi,
                         double cre, double cim, bool comm) {
  if (!ctx->gauge_set) TMHIP_FAIL("Hopping_Matrix called before tmhip_set_gauge");
  if (out == in) TMHIP_FAIL("Hopping_Matrix: l and k must differ (operator/D_psi_body.c:267-272 convention)");
  HopArgs a;
  fill_args(a, ctx, ieo, out, in, p, cre, cim);
  const HopLaunch o = {ctx->opt_block, ctx->opt_nt != 0, ctx->opt_minw, ctx->opt_xcd, ctx->opt_occ, ctx->opt_tgrp, ctx->opt_shape, ctx->opt_gaux, ctx->opt_gdrop};
  const bool split = ctx->g.nproc_t > 1 || ctx->loopback;
  if (!split) {
    a.i_begin = 0; a.i_end = ctx->Vh;
    launch_epi<0>(a, epi, ctx->stream, o, true);
  } else if (!comm) {
    launch_interior(ctx, a, epi, o);
    launch_boundary(ctx, a, epi, o, ctx->stream);
  } else {
    // Split-phase with the whole boundary pipeline on the second stream:
    //   comm stream : pack faces -> exchange (RCCL / copies) -> the two boundary kernels
    //   main stream : interior kernel (t in [1, T-2]), which needs no remote data
    // The GPU co-schedules the 2 x face/BS boundary blocks with the interior blocks, so pack,
    // exchange and boundary work hide behind the interior kernel (the reference's analogue is the
    // tsplit variant, operator/hopping_sse_dbl.c:79-161).
    const bool flags = ctx->opt_flagsync != 0;
    const unsigned int seq = ++ctx->hop_seq;
    if (ctx->opt_fusedface) {
      // ONE kernel on the main stream (interior blocks first, face blocks last, the latter wait in-kernel for
      // the faces); the comm stream packs, exchanges and publishes.  The main stream never waits on the host
      // side, so consecutive stencils run back to back as on an unsplit lattice.
      hipLaunchKernelGGL(flag_set_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->sync_flags + 0, seq);
      hipLaunchKernelGGL(flag_wait_kernel, dim3(1), dim3(1), 0, ctx->comm_stream, ctx->sync_flags + 0, seq, ctx->sync_flags + 2);
      hipLaunchKernelGGL(pack_faces_kernel, dim3((ctx->face + 255) / 256, 2), dim3(256), 0, ctx->comm_stream,
                         in, ctx->ns, ctx->Vh, ctx->face, ctx->send_dn, ctx->send_up);
      if (tmhip_halo_exchange(ctx)) return 1;
      hipLaunchKernelGGL(flag_set_kernel, dim3(1), dim3(1), 0, ctx->comm_stream, ctx->sync_flags + 1, seq);
      a.halo_flag = ctx->sync_flags + 1; a.halo_seq = seq; a.err_flag = ctx->sync_flags + 2;
      launch_fused_faces(ctx, a, epi, o);
      TMHIP_CHECK(hipGetLastError());
      return 0;
    }
    if (flags) {  // `in` (and `p`) are ready, `out` is free
      hipLaunchKernelGGL(flag_set_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->sync_flags + 0, seq);
      hipLaunchKernelGGL(flag_wait_kernel, dim3(1), dim3(1), 0, ctx->comm_stream, ctx->sync_flags + 0, seq, ctx->sync_flags + 2);
    } else {
      TMHIP_CHECK(hipEventRecord(ctx->ev_pack, ctx->stream));
      TMHIP_CHECK(hipStreamWaitEvent(ctx->comm_stream, ctx->ev_pack, 0));
    }
    hipLaunchKernelGGL(pack_faces_kernel, dim3((ctx->face + 255) / 256, 2), dim3(256), 0, ctx->comm_stream,
                       in, ctx->ns, ctx->Vh, ctx->face, ctx->send_dn, ctx->send_up);
    if (tmhip_halo_exchange(ctx)) return 1;
    HopArgs b = a;
    launch_boundary(ctx, b, epi, o, ctx->comm_stream);
    if (flags) hipLaunchKernelGGL(flag_set_kernel, dim3(1), dim3(1), 0, ctx->comm_stream, ctx->sync_flags + 1, seq);
    else TMHIP_CHECK(hipEventRecord(ctx->ev_comm, ctx->comm_stream));
    launch_interior(ctx, a, epi, o);
    if (flags) hipLaunchKernelGGL(flag_wait_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->sync_flags + 1, seq, ctx->sync_flags + 2);
    else TMHIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->ev_comm, 0));
  }
  TMHIP_CHECK(hipGetLastError());
  return 0;
}

// tm_sub_Hopping_Matrix with the real scalar product <dotv, l> accumulated in the epilogue (one partial
// per block in ctx->partials).  Unsplit lattices only; *npartials receives the number of blocks.
int tmhip_launch_hopping_dot(tmhip_ctx *ctx, int ieo, v2d *out, const v2d *in, const v2d *p, const v2d *dotv,
                             double cre, double cim, int *npartials) {
  if (!ctx->gauge_set) TMHIP_FAIL("Hopping_Matrix called before tmhip_set_gauge");
  if (ctx->g.nproc_t > 1 || ctx->loopback || ctx->Vh % 256 != 0) TMHIP_FAIL("fused scalar product needs an unsplit lattice with V/2 %% 256 == 0");
  if (out == in) TMHIP_FAIL("Hopping_Matrix: l and k must differ");
  HopArgs a;
  fill_args(a, ctx, ieo, out, in, p, cre, cim);
  a.dotv = dotv; a.partials = ctx->partials;
  HopLaunch o = {256, ctx->opt_nt != 0, 0, ctx->opt_xcd, ctx->opt_occ, ctx->opt_tgrp, ctx->opt_shape, ctx->opt_gaux, ctx->opt_gdrop};
  a.i_begin = 0; a.i_end = ctx->Vh;
  const int nb = (ctx->Vh + 255) / 256;
  *npartials = (o.xcd && nb >= 64) ? ((nb + 7) / 8) * 8 : nb;  // grid size chosen by launch_one
  if (*npartials > ctx->max_partials) TMHIP_FAIL("partials buffer too small");
  launch_epi<0>(a, EPI_TM_SUB_G5_DOT, ctx->stream, o, true);
  TMHIP_CHECK(hipGetLastError());
  return 0;
}

// Single-process ring: n contexts (one per GPU, or several on one GPU for the self-test) that
// together hold a T-split lattice; faces move by peer copies instead of RCCL.  Collective over
// all contexts because the host enqueues for every rank:
//   1. every rank packs its two faces            (after its neighbours finished reading the previous ones)
//   2. every rank pulls the neighbours' faces on its comm stream || runs its interior kernel
//   3. every rank runs its boundary kernels once its pulls have landed
extern "C" int tmhip_multi_hopping_matrix(int n, tmhip_ctx **ctxs, int ieo, tmhip_field **l, tmhip_field **k) {
  if (n < 2) TMHIP_FAIL("tmhip_multi_hopping_matrix needs >= 2 contexts");
  for (int r = 0; r < n; r++) {
    tmhip_ctx *c = ctxs[r];
    if (c->g.nproc_t != n || c->g.proc_t != r) TMHIP_FAIL("context %d is not rank %d of a %d-way T split", r, r, n);
    if (!c->gauge_set) TMHIP_FAIL("Hopping_Matrix called before tmhip_set_gauge");
    if (l[r]->kind != TMHIP_FIELD_EO || k[r]->kind != TMHIP_FIELD_EO || l[r]->d == k[r]->d) TMHIP_FAIL("bad fields for rank %d", r);
  }
  const size_t fb = (size_t)6 * ctxs[0]->face * sizeof(v2d);
  for (int r = 0; r < n; r++) {
    tmhip_ctx *c = ctxs[r], *up = ctxs[(r + 1) % n], *dn = ctxs[(r + n - 1) % n];
    TMHIP_CHECK(hipSetDevice(c->device));
    TMHIP_CHECK(hipStreamWaitEvent(c->stream, up->ev_comm, 0));  // neighbours still pulling the previous faces
    TMHIP_CHECK(hipStreamWaitEvent(c->stream, dn->ev_comm, 0));
    launch_pack(c, k[r]->d);
    TMHIP_CHECK(hipEventRecord(c->ev_pack, c->stream));
  }
  for (int r = 0; r < n; r++) {
    tmhip_ctx *c = ctxs[r], *up = ctxs[(r + 1) % n], *dn = ctxs[(r + n - 1) % n];
    TMHIP_CHECK(hipSetDevice(c->device));
    TMHIP_CHECK(hipStreamWaitEvent(c->comm_stream, up->ev_pack, 0));
    TMHIP_CHECK(hipStreamWaitEvent(c->comm_stream, dn->ev_pack, 0));
    TMHIP_CHECK(hipMemcpyPeerAsync(c->recv_up, c->device, up->send_dn, up->device, fb, c->comm_stream));
    TMHIP_CHECK(hipMemcpyPeerAsync(c->recv_dn, c->device, dn->send_up, dn->device, fb, c->comm_stream));
    TMHIP_CHECK(hipEventRecord(c->ev_comm, c->comm_stream));
    HopArgs a;
    fill_args(a, c, ieo, l[r]->d, k[r]->d, nullptr, 0, 0);
    const HopLaunch o = {c->opt_block, c->opt_nt != 0, c->opt_minw, c->opt_xcd, c->opt_occ, c->opt_tgrp, c->opt_shape, c->opt_gaux, c->opt_gdrop};
    launch_interior(c, a, EPI_STORE, o);
  }
  for (int r = 0; r < n; r++) {
    tmhip_ctx *c = ctxs[r];
    TMHIP_CHECK(hipSetDevice(c->device));
    TMHIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_comm, 0));
    HopArgs a;
    fill_args(a, c, ieo, l[r]->d, k[r]->d, nullptr, 0, 0);
    const HopLaunch o = {c->opt_block, c->opt_nt != 0, c->opt_minw, c->opt_xcd, c->opt_occ, c->opt_tgrp, c->opt_shape, c->opt_gaux, c->opt_gdrop};
    launch_boundary(c, a, EPI_STORE, o, c->stream);
    TMHIP_CHECK(hipGetLastError());
  }
  return 0;
}
