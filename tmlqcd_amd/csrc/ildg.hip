// ILDG gauge configurations in and out of HBM (io/gauge_read.c, io/gauge_write.c).
//
// The reference reads the "ildg-binary-data" record of a LIME file site by site on the host (io/gauge_read_binary.c:140-200:
// loops t, z, y, x; per site four links in the file's order x, y, z, t, big-endian, 64 or 32 bit; be_to_cpu_assign into
// g_gauge_field[g_ipt[t][x][y][z]][mu]) and accumulates the SciDAC checksum over the file bytes of every site (io/dml.c:49-60:
// CRC-32 of the site's bytes, rotated by rank % 29 and rank % 31, XOR-ed over sites).  Here the record's bytes go to the device as
// they lie in the file and ONE kernel does the rest in HBM: byte swap, precision conversion, the x <-> z transposition into the
// lexicographic order of g_gauge_field, the link rotation, and the checksum -- byte work bound by HBM: whole 1 KiB rows in and out,
// a lane owns one 144-byte link in between, the per-site CRC is assembled from the four link CRCs (see below).
// The write path is the mirror image.  The LIME container is framed on the host by a reader / writer written from the published
// format of c-lime 1.3.x (not part of the reference tree, not installed here): 144-byte record headers
//   0: magic 0x456789ab (be32)  4: version 1 (be16)  6: MB 0x80 | ME 0x40  8: data length (be64)  16: type, 128 bytes, NUL-padded
// followed by the data padded to a multiple of 8 bytes.  Record names, order and MB / ME bits follow io/gauge_write.c:22-59.
#include "tmhip_internal.h"

#include <string>
#include <vector>

namespace {

// ------------------------------------------------------------------ CRC-32 (io/DML_crc32.c = zlib's crc32: reflected 0xedb88320)
__host__ __device__ inline unsigned crc_table_entry(unsigned n) {
  unsigned c = n;
  for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
  return c;
}

#define ILDG_BS 256            /* threads per block = 4 waves x 16 file sites x 4 links */

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
struct IldgGeom { int T, LX, LY, LZ; unsigned long long rank0; };   // local extents; DML rank of this rank's first file site

// lexicographic index ((t LX + x) LY + y) LZ + z of the f-th site of the file order ((t LZ + z) LY + y) LX + x
__device__ __forceinline__ size_t ildg_site_index(const IldgGeom &g, size_t f) {
  const int x = (int)(f % g.LX); f /= g.LX;
  const int y = (int)(f % g.LY); f /= g.LY;
  const int z = (int)(f % g.LZ);
  const int t = (int)(f / g.LZ);
  return (((size_t)t * g.LX + x) * g.LY + y) * g.LZ + z;
}
__device__ __forceinline__ unsigned bswap32(unsigned v) { return __builtin_bswap32(v); }

// A wave owns 16 consecutive file sites = 64 links = 9216 contiguous bytes of a 64-bit record, a lane ONE link: file link j (x, y, z,
// t) of site f is the chunk of 144 bytes (72 for 32-bit data) at (4 f + j) chunks and becomes g_gauge_field[ix][mu], mu = (j + 1) % 4
// (gauge_read_binary.c:187-190).  Global memory is only touched in whole 1 KiB rows: the wave's part of the record goes through a
// private 9 KB LDS region (nine 16-byte steps in, each lane then takes its own 144 bytes out of it: lane stride 36 words, conflict-
// free for 16-byte reads), the converted numbers go back into the region in the order of the destination, and leave it in flat
// order again, so that the stores are whole 576-byte site records (the x <-> z transposition moves whole sites).
// Checksum: DML_crc32 over a site's bytes = its four link chunks in order.  CRC(A || B) = Z_|B|(CRC(A)) xor CRC(B), Z_n = the
// CRC register advanced by n zero bytes (linear over GF(2): a 32 x 32 bit matrix), so lane j computes the CRC of its chunk (four
// bytes per step, four independent table look-ups: "slicing by 4"), applies Z_chunk (3 - j) times and the four lanes of a site
// XOR their results; the site's rank rotations and the XOR over sites follow io/dml.c:49-60.
#define ILDG_WSITES 16                         /* file sites per wave */
#define ILDG_SLOTS 1024                        /* checksum accumulators (pairs of words) the waves spread their atomics over */
struct IldgLds { unsigned tab[4][256]; unsigned zop[32]; v4u region[ILDG_BS / 64][ILDG_WSITES * 36]; };
// the tables are compile-time constants (building them per block -- above all the 144 dependent steps of the zero operator -- cost a
// quarter of the kernel); a block copies them from constant memory into LDS, where 64 lanes can index them independently
struct IldgConst { unsigned tab[4][256]; unsigned zop[2][32]; };     // zop[0]: 144 zero bytes (64-bit links), zop[1]: 72 (32-bit)
constexpr unsigned crc_entry_c(unsigned n) {
  unsigned c = n;
  for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
  return c;
}
constexpr IldgConst ildg_make_const() {
  IldgConst t{};
  for (unsigned n = 0; n < 256; n++) {
    unsigned c = crc_entry_c(n);
    t.tab[0][n] = c;
    for (int k = 1; k < 4; k++) { c = crc_entry_c(c & 0xff) ^ (c >> 8); t.tab[k][n] = c; }
  }
  for (int w = 0; w < 2; w++)
    for (unsigned b = 0; b < 32; b++) {
      unsigned c = 1u << b;
      for (int n = 0; n < (w ? 72 : 144); n++) c = t.tab[0][c & 0xff] ^ (c >> 8);
      t.zop[w][b] = c;
    }
  return t;
}
__constant__ IldgConst c_ildg = ildg_make_const();
__device__ __forceinline__ void ildg_tables(IldgLds &L, int chunk_bytes) {
  for (int n = threadIdx.x; n < 1024; n += ILDG_BS) (&L.tab[0][0])[n] = (&c_ildg.tab[0][0])[n];
  if (threadIdx.x < 32) L.zop[threadIdx.x] = c_ildg.zop[chunk_bytes == 144 ? 0 : 1][threadIdx.x];
  __syncthreads();
}
__device__ __forceinline__ unsigned ildg_crc_step(unsigned c, unsigned w, const IldgLds &L) {
  const unsigned v = c ^ w;
  return L.tab[3][v & 0xff] ^ L.tab[2][(v >> 8) & 0xff] ^ L.tab[1][(v >> 16) & 0xff] ^ L.tab[0][v >> 24];
}
__device__ __forceinline__ unsigned ildg_zero_op(unsigned c, const IldgLds &L) {
  unsigned r = 0;
#pragma unroll 8
  for (int b = 0; b < 32; b++) r ^= ((c >> b) & 1u) ? L.zop[b] : 0u;
  return r;
}
// chunk CRC of lane j (already final, i.e. with DML_crc32's pre / post inversion) -> site CRC in the site's four lanes -> checksum words
__device__ __forceinline__ void ildg_checksum_accum(unsigned crc, int j, unsigned long long rank64, bool active, const IldgLds &L, unsigned *sums) {
  if (j < 3) crc = ildg_zero_op(crc, L);
  if (j < 2) crc = ildg_zero_op(crc, L);
  if (j < 1) crc = ildg_zero_op(crc, L);
  crc ^= __shfl_xor(crc, 1, 64);
  crc ^= __shfl_xor(crc, 2, 64);               // every lane of the site now holds DML_crc32(0, site bytes)
  const unsigned rank = (unsigned)rank64;      // DML_SiteRank is uint32_t (io/dml.h:40)
  const unsigned r29 = rank % 29, r31 = rank % 31;
  const bool lead = active && j == 0;
  unsigned a = lead ? (r29 ? (crc << r29 | crc >> (32 - r29)) : crc) : 0u, b = lead ? (r31 ? (crc << r31 | crc >> (32 - r31)) : crc) : 0u;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { a ^= __shfl_xor(a, off, 64); b ^= __shfl_xor(b, off, 64); }
  // XOR is associative and commutative: any order gives the same words.  One address would serialise the atomics of all waves
  // (measured: 131 072 atomics on one word = 1.5 ms, the whole kernel): ILDG_SLOTS pairs, folded on the host.
  if ((threadIdx.x & 63) == 0) {
    unsigned *slot = sums + 2 * ((blockIdx.x * (ILDG_BS / 64) + (threadIdx.x >> 6)) & (ILDG_SLOTS - 1));
    atomicXor(slot, a); atomicXor(slot + 1, b);
  }
}

// file bytes -> lexicographic links [ix][4][9] complex double.  WPS = 32-bit words per file site: 144 (64-bit data) or 72.
template <int WPS>
__global__ __launch_bounds__(ILDG_BS) void ildg_unpack_kernel(const unsigned *__restrict__ file, v2d *__restrict__ raw, IldgGeom g, size_t nsites, unsigned *sums) {
  __shared__ IldgLds L;
  ildg_tables(L, WPS);                         // chunk = WPS / 4 words = WPS bytes
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 3;
  const size_t f0 = ((size_t)blockIdx.x * (ILDG_BS / 64) + wv) * ILDG_WSITES, f = f0 + (lane >> 2);
  const int ns = f0 >= nsites ? 0 : (int)(nsites - f0 < ILDG_WSITES ? nsites - f0 : ILDG_WSITES);
  const bool active = (lane >> 2) < ns;
  v4u *R = L.region[wv];
  // 1. the wave's part of the record, whole rows: ns * WPS words
  {
    const int nq = ns * WPS / 4;               // 16-byte pieces
    const v4u *src = reinterpret_cast<const v4u *>(file + f0 * WPS);
    for (int q = lane; q < nq; q += 64) R[q] = __builtin_nontemporal_load(src + q);
  }
  __syncthreads();
  // 2. every lane its link: CRC over the file bytes, conversion
  unsigned crc = 0xffffffffu;
  v2d u[9];
  if (active) {
    if (WPS == 144) {
      const v4u *p = R + lane * 9;
#pragma unroll
      for (int e = 0; e < 9; e++) {
        const v4u w = p[e];
        crc = ildg_crc_step(ildg_crc_step(ildg_crc_step(ildg_crc_step(crc, w.x, L), w.y, L), w.z, L), w.w, L);
        const unsigned long long re = ((unsigned long long)bswap32(w.x) << 32) | bswap32(w.y), im = ((unsigned long long)bswap32(w.z) << 32) | bswap32(w.w);
        u[e] = v2d{__longlong_as_double((long long)re), __longlong_as_double((long long)im)};
      }
    } else {
      const v2u *p = reinterpret_cast<const v2u *>(R) + lane * 9;
#pragma unroll
      for (int e = 0; e < 9; e++) {
        const v2u w = p[e];
        crc = ildg_crc_step(ildg_crc_step(crc, w.x, L), w.y, L);
        u[e] = v2d{(double)__uint_as_float(bswap32(w.x)), (double)__uint_as_float(bswap32(w.y))};   // be_to_cpu_assign_single2double
      }
    }
  }
  ildg_checksum_accum(crc ^ 0xffffffffu, j, g.rank0 + f, active, L, sums);
  __syncthreads();                             // every lane has taken its bytes: the region now receives the destination order
  if (active) {
    v2d *o = reinterpret_cast<v2d *>(R) + ((lane >> 2) * 4 + ((j + 1) & 3)) * 9;
#pragma unroll
    for (int e = 0; e < 9; e++) o[e] = u[e];
  }
  __syncthreads();
  // 3. whole 576-byte site records out
  for (int k = lane; k < ns * 36; k += 64) {
    const int s = k / 36, e = k - s * 36;
    raw[ildg_site_index(g, f0 + s) * 36 + e] = reinterpret_cast<const v2d *>(R)[k];
  }
}

// lexicographic links -> file bytes (io/gauge_write_binary.c:150-175) and their checksum
template <int WPS>
__global__ __launch_bounds__(ILDG_BS) void ildg_pack_kernel(unsigned *__restrict__ file, const v2d *__restrict__ raw, IldgGeom g, size_t nsites, unsigned *sums) {
  __shared__ IldgLds L;
  ildg_tables(L, WPS);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 3;
  const size_t f0 = ((size_t)blockIdx.x * (ILDG_BS / 64) + wv) * ILDG_WSITES, f = f0 + (lane >> 2);
  const int ns = f0 >= nsites ? 0 : (int)(nsites - f0 < ILDG_WSITES ? nsites - f0 : ILDG_WSITES);
  const bool active = (lane >> 2) < ns;
  v4u *R = L.region[wv];
  // 1. whole site records in, into the order of the file (link mu -> file link (mu + 3) % 4)
  for (int k = lane; k < ns * 36; k += 64) {
    const int s = k / 36, e = k - s * 36, mu = e / 9, c = e - mu * 9;
    reinterpret_cast<v2d *>(R)[s * 36 + ((mu + 3) & 3) * 9 + c] = raw[ildg_site_index(g, f0 + s) * 36 + e];
  }
  __syncthreads();
  // 2. every lane its link: conversion, CRC over the bytes that go to the file
  unsigned crc = 0xffffffffu;
  v4u w4[9];
  v2u w2[9];
  if (active) {
    const v2d *p = reinterpret_cast<const v2d *>(R) + lane * 9;
#pragma unroll
    for (int e = 0; e < 9; e++) {
      const v2d val = p[e];
      if (WPS == 144) {
        const unsigned long long re = (unsigned long long)__double_as_longlong(val.x), im = (unsigned long long)__double_as_longlong(val.y);
        w4[e] = v4u{bswap32((unsigned)(re >> 32)), bswap32((unsigned)re), bswap32((unsigned)(im >> 32)), bswap32((unsigned)im)};
        crc = ildg_crc_step(ildg_crc_step(ildg_crc_step(ildg_crc_step(crc, w4[e].x, L), w4[e].y, L), w4[e].z, L), w4[e].w, L);
      } else {
        w2[e] = v2u{bswap32(__float_as_uint((float)val.x)), bswap32(__float_as_uint((float)val.y))};   // be_to_cpu_assign_double2single
        crc = ildg_crc_step(ildg_crc_step(crc, w2[e].x, L), w2[e].y, L);
      }
    }
  }
  ildg_checksum_accum(crc ^ 0xffffffffu, j, g.rank0 + f, active, L, sums);
  __syncthreads();                             // (32-bit data: the packed words of a lane land where another lane's input lay)
  if (active) {
#pragma unroll
    for (int e = 0; e < 9; e++) {
      if (WPS == 144) R[lane * 9 + e] = w4[e];
      else reinterpret_cast<v2u *>(R)[lane * 9 + e] = w2[e];
    }
  }
  __syncthreads();
  // 3. the wave's part of the record, whole rows
  {
    const int nq = ns * WPS / 4;
    v4u *dst = reinterpret_cast<v4u *>(file + f0 * WPS);
    for (int q = lane; q < nq; q += 64) dst[q] = R[q];
  }
}

// ------------------------------------------------------------------ LIME framing (host)
const unsigned LIME_MAGIC = 0x456789abu;
struct LimeRecord { std::string type; unsigned long long bytes; long data_pos; int mb, me; };

unsigned long long be64(const unsigned char *p) { unsigned long long v = 0; for (int i = 0; i < 8; i++) v = (v << 8) | p[i]; return v; }
void put_be(unsigned char *p, unsigned long long v, int n) { for (int i = n - 1; i >= 0; i--) { p[i] = (unsigned char)(v & 0xff); v >>= 8; } }

// the records of a LIME file in order; -1 on a malformed header
int lime_scan(FILE *fp, std::vector<LimeRecord> &recs) {
  unsigned char h[144];
  long pos = 0;
  for (;;) {
    if (fseek(fp, pos, SEEK_SET)) return -1;
    const size_t n = fread(h, 1, 144, fp);
    if (n == 0) return 0;                       // LIME_EOF
    if (n != 144) return -1;
    const unsigned magic = ((unsigned)h[0] << 24) | ((unsigned)h[1] << 16) | ((unsigned)h[2] << 8) | h[3];
    const unsigned version = ((unsigned)h[4] << 8) | h[5];
    if (magic != LIME_MAGIC || version != 1) return -1;
    LimeRecord r;
    r.mb = (h[6] & 0x80) != 0; r.me = (h[6] & 0x40) != 0;
    r.bytes = be64(h + 8);
    h[143] = 0;
    r.type = std::string(reinterpret_cast<const char *>(h + 16));
    r.data_pos = pos + 144;
    recs.push_back(r);
    pos += 144 + (long)((r.bytes + 7) / 8 * 8);
  }
}
int lime_write_header(FILE *fp, int mb, int me, const char *type, unsigned long long bytes) {
  unsigned char h[144];
  memset(h, 0, sizeof(h));
  put_be(h, LIME_MAGIC, 4);
  put_be(h + 4, 1, 2);
  h[6] = (unsigned char)((mb ? 0x80 : 0) | (me ? 0x40 : 0));
  put_be(h + 8, bytes, 8);
  strncpy(reinterpret_cast<char *>(h + 16), type, 127);
  return fwrite(h, 1, 144, fp) == 144 ? 0 : -1;
}
int lime_write_data(FILE *fp, const void *data, unsigned long long bytes) {
  static const unsigned char zero[8] = {0};
  if (bytes && fwrite(data, 1, bytes, fp) != bytes) return -1;
  const unsigned long long pad = (8 - bytes % 8) % 8;
  if (pad && fwrite(zero, 1, pad, fp) != pad) return -1;
  return 0;
}
int lime_write_message(FILE *fp, int mb, int me, const char *type, const std::string &msg) {
  return lime_write_header(fp, mb, me, type, msg.size()) || lime_write_data(fp, msg.data(), msg.size());
}
int lime_read_string(FILE *fp, const LimeRecord &r, std::string &out) {
  out.resize(r.bytes);
  if (fseek(fp, r.data_pos, SEEK_SET)) return -1;
  return r.bytes == 0 || fread(&out[0], 1, r.bytes, fp) == r.bytes ? 0 : -1;
}
// io/utils_parse_ildgformat_xml.c / utils_parse_checksum_xml.c: tokens separated by "<> \n\t", the value is the token behind the tag
bool xml_value(const std::string &msg, const char *tag, std::string &val) {
  std::vector<std::string> tok;
  size_t i = 0;
  const std::string sep = "<> \n\t";
  while (i < msg.size()) {
    const size_t b = msg.find_first_not_of(sep, i);
    if (b == std::string::npos) break;
    size_t e = msg.find_first_of(sep, b);
    if (e == std::string::npos) e = msg.size();
    tok.push_back(msg.substr(b, e - b));
    i = e;
  }
  const size_t n = strlen(tag);
  for (size_t k = 0; k + 1 < tok.size(); k++)
    if (!strncmp(tok[k].c_str(), tag, n)) { val = tok[k + 1]; return true; }
  return false;
}

int sums_reserve(tmhip_ctx *ctx) {
  if (!ctx->io_sums) TMHIP_CHECK(hipMalloc((void **)&ctx->io_sums, 2 * ILDG_SLOTS * sizeof(unsigned)));
  TMHIP_CHECK(hipMemsetAsync(ctx->io_sums, 0, 2 * ILDG_SLOTS * sizeof(unsigned), ctx->stream));
  return 0;
}
// the checksum words of the launch that has just been enqueued (synchronises the stream)
int sums_fetch(tmhip_ctx *ctx, unsigned out[2]) {
  static unsigned h[2 * ILDG_SLOTS];
  TMHIP_CHECK(hipMemcpyAsync(h, ctx->io_sums, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  out[0] = out[1] = 0;
  for (int k = 0; k < ILDG_SLOTS; k++) { out[0] ^= h[2 * k]; out[1] ^= h[2 * k + 1]; }
  return 0;
}
// XOR of the checksum words over the ranks (io/dml.c:63-66): RCCL has no bit-wise XOR reduction, so the words are gathered
// any_nonzero: the words are error flags, not checksums -- combined as "did ANY rank report one" (an XOR lets two ranks' flags cancel)
int sums_combine(tmhip_ctx *ctx, unsigned sums[2], bool any_nonzero = false) {
  int n = ctx->g.nproc_t;
  if (!ctx->shm) TMHIP_NCCL_CHECK(ncclCommCount(ctx->comm_red, &n));
  if (n < 1 || n > ILDG_SLOTS) TMHIP_FAIL("sums_combine: %d ranks", n);
  TMHIP_CHECK(hipMemcpyAsync(ctx->io_sums, sums, 2 * sizeof(unsigned), hipMemcpyHostToDevice, ctx->stream));
  if (ctx->shm) { if (tmhip_shm_allgather(ctx, ctx->stream, ctx->io_sums, ctx->io_sums + 2, 2 * sizeof(unsigned))) return 1; }
  else TMHIP_NCCL_CHECK(ncclAllGather(ctx->io_sums, ctx->io_sums + 2, 2, ncclUint32, ctx->comm_red, ctx->stream));   // (io_sums holds 2 * ILDG_SLOTS words)
  std::vector<unsigned> h(2 * (size_t)n);
  TMHIP_CHECK(hipMemcpyAsync(h.data(), ctx->io_sums + 2, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  sums[0] = sums[1] = 0;
  for (int k = 0; k < n; k++) {
    if (any_nonzero) { sums[0] |= h[2 * k]; sums[1] |= h[2 * k + 1]; }
    else { sums[0] ^= h[2 * k]; sums[1] ^= h[2 * k + 1]; }
  }
  return 0;
}
IldgGeom io_geom(const tmhip_ctx *ctx) {
  return IldgGeom{ctx->g.T, ctx->g.LX, ctx->g.LY, ctx->g.LZ, (unsigned long long)ctx->g.proc_t * ctx->g.T * ctx->g.LX * ctx->g.LY * ctx->g.LZ};
}
}  // namespace

extern "C" {

/* io/gauge_read_binary.c:140-200 + io/dml.c:49-60: this rank's part of an "ildg-binary-data" record (host pointer: T LZ LY LX
 * sites x 4 links x 9 complex, big-endian, prec = 64 or 32 bits) -> the device-resident lexicographic links and the stencil's
 * gauge copy, as after tmhip_set_gauge; sums[0..1] = SciDAC checksum A, B of this rank's sites (XOR over the ranks gives the file's,
 * io/dml.c:63-66).  On a T-split rank the halo slabs are exchanged with the ring neighbours (xchange_gauge). */
int tmhip_gauge_unpack_ildg(tmhip_ctx *ctx, const void *file_bytes, int prec, unsigned *sums) {
  if (!file_bytes || (prec != 32 && prec != 64)) TMHIP_FAIL("tmhip_gauge_unpack_ildg: null data or precision %d (32 or 64)", prec);
  if (ctx->g.nproc_t > 1 && !ctx->comm_ready) TMHIP_FAIL("tmhip_gauge_unpack_ildg: T-split rank without tmhip_comm_init (the halo slabs of the new links come from the ring neighbours)");
  TMHIP_CHECK(hipSetDevice(ctx->device));
  const size_t nsites = (size_t)ctx->V, fbytes = nsites * 4 * 9 * (prec == 64 ? 16 : 8), gbytes = (size_t)ctx->VPR * 36 * sizeof(v2d);
  if (tmhip_stage_reserve(ctx, fbytes) || sums_reserve(ctx)) return 1;
  if (!ctx->gauge_raw) TMHIP_CHECK(hipMalloc((void **)&ctx->gauge_raw, gbytes));
  TMHIP_CHECK(hipMemcpyAsync(ctx->stage, file_bytes, fbytes, hipMemcpyHostToDevice, ctx->stream));
  const dim3 grid((unsigned)((nsites + ILDG_BS / 4 - 1) / (ILDG_BS / 4)));
  if (prec == 64) hipLaunchKernelGGL(ildg_unpack_kernel<144>, grid, dim3(ILDG_BS), 0, ctx->stream, (const unsigned *)ctx->stage, ctx->gauge_raw, io_geom(ctx), nsites, ctx->io_sums);
  else hipLaunchKernelGGL(ildg_unpack_kernel<72>, grid, dim3(ILDG_BS), 0, ctx->stream, (const unsigned *)ctx->stage, ctx->gauge_raw, io_geom(ctx), nsites, ctx->io_sums);
  TMHIP_CHECK(hipGetLastError());
  ctx->gauge_raw_valid = true;
  if (ctx->g.nproc_t > 1 && tmhip_exchange_gauge_halo(ctx)) return 1;
  ctx->sw_set = false; ctx->clover_set = false; ctx->clover32_set = false;      // clover blocks belong to the old links
  if (tmhip_resort_gauge(ctx)) return 1;
  unsigned h[2];
  if (sums_fetch(ctx, h)) return 1;
  if (sums) { sums[0] = h[0]; sums[1] = h[1]; }
  if (ctx->opt_recon == 12) return tmhip_check_gauge_recon(ctx);
  return 0;
}

/* io/gauge_write_binary.c:150-175: the device-resident links of this rank as they go into the "ildg-binary-data" record */
int tmhip_gauge_pack_ildg(tmhip_ctx *ctx, void *file_bytes, int prec, unsigned *sums) {
  if (!file_bytes || (prec != 32 && prec != 64)) TMHIP_FAIL("tmhip_gauge_pack_ildg: null buffer or precision %d (32 or 64)", prec);
  if (!ctx->gauge_raw || !ctx->gauge_raw_valid) TMHIP_FAIL("tmhip_gauge_pack_ildg: no links resident on the device (tmhip_set_gauge first)");
  TMHIP_CHECK(hipSetDevice(ctx->device));
  const size_t nsites = (size_t)ctx->V, fbytes = nsites * 4 * 9 * (prec == 64 ? 16 : 8);
  if (tmhip_stage_reserve(ctx, fbytes) || sums_reserve(ctx)) return 1;
  const dim3 grid((unsigned)((nsites + ILDG_BS / 4 - 1) / (ILDG_BS / 4)));
  if (prec == 64) hipLaunchKernelGGL(ildg_pack_kernel<144>, grid, dim3(ILDG_BS), 0, ctx->stream, (unsigned *)ctx->stage, (const v2d *)ctx->gauge_raw, io_geom(ctx), nsites, ctx->io_sums);
  else hipLaunchKernelGGL(ildg_pack_kernel<72>, grid, dim3(ILDG_BS), 0, ctx->stream, (unsigned *)ctx->stage, (const v2d *)ctx->gauge_raw, io_geom(ctx), nsites, ctx->io_sums);
  TMHIP_CHECK(hipGetLastError());
  unsigned h[2];
  TMHIP_CHECK(hipMemcpyAsync(file_bytes, ctx->stage, fbytes, hipMemcpyDeviceToHost, ctx->stream));
  if (sums_fetch(ctx, h)) return 1;
  if (sums) { sums[0] = h[0]; sums[1] = h[1]; }
  return 0;
}

/* io/gauge_read.c:28-198 read_gauge_field(filename, gf) for a single rank: walks the LIME records ("ildg-format", "ildg-binary-data",
 * "scidac-checksum", "xlf-info", "ildg-data-lfn"), checks record multiplicity, lattice size, precision and the checksum exactly as the
 * reference does (io_checks = 0 is g_disable_IO_checks), leaves the links in HBM and -- host_gf != NULL -- in the host's
 * g_gauge_field layout ([VOLUMEPLUSRAND][4] su3).  prec_expected = gauge_precision_read_flag (64 / 32).  Returns 0, or -1 like the
 * reference (messages on stderr); info (may be NULL) receives what GaugeInfo would hold. */
int tmhip_read_gauge_field(tmhip_ctx *ctx, const char *filename, int prec_expected, int io_checks, void *host_gf, tmhip_gauge_info *info) {
  // T-split ranks: every rank walks the records itself and reads ITS contiguous part of the binary record (the offset of
  // gauge_read_binary.c:158-163 for a decomposition in T only); the checksum words are combined over the ranks (DML_checksum_combine)
  if (ctx->g.nproc_t > 1 && !ctx->comm_ready) TMHIP_FAIL("tmhip_read_gauge_field: T-split rank without tmhip_comm_init");
  FILE *fp = fopen(filename, "rb");
  if (!fp) { fprintf(stderr, "[tmlqcd_hip] read_gauge_field: cannot open %s\n", filename); return -1; }
  std::vector<LimeRecord> recs;
  if (lime_scan(fp, recs)) { fprintf(stderr, "[tmlqcd_hip] read_gauge_field: %s is not a LIME file (bad record header)\n", filename); fclose(fp); return -1; }
  const int L[4] = {ctx->g.LX, ctx->g.LY, ctx->g.LZ, ctx->g.T * ctx->g.nproc_t};
  int n_bin = 0, n_sum = 0, n_fmt = 0, fmt_ok = 0, sum_ok = 0;
  int fprec = 0, fl[4] = {0, 0, 0, 0};
  unsigned calc[2] = {0, 0}, stored[2] = {0, 0};
  std::string xlf, lfn;
  for (const LimeRecord &r : recs) {
    if (r.type == "ildg-binary-data") {
      if (n_bin++ && io_checks) {
        fprintf(stderr, "In gauge file %s, multiple LIME records with name: \"ildg-binary-data\" found.\nUnable to verify integrity of the gauge field data.\n", filename);
        fclose(fp); return -1;
      }
      const unsigned long long mine = (unsigned long long)ctx->V * 4 * 144 / (prec_expected == 64 ? 1 : 2), want = mine * ctx->g.nproc_t;
      if (r.bytes != want) {     // gauge_read_binary.c:148-155
        fprintf(stderr, "Lattice size and precision found in data file do not match those requested at input.\nExpected LX = %d, LY = %d, LZ = %d, LT = %d, and %s precision.\n"
                        "Expected %llu bytes, found %llu bytes.\nGauge file reading failed at binary part, unable to proceed.\n",
                L[0], L[1], L[2], L[3], prec_expected == 64 ? "double" : "single", want, r.bytes);
        fclose(fp); return -1;
      }
      void *buf = nullptr;
      if (hipHostMalloc(&buf, mine, hipHostMallocDefault) != hipSuccess) { fprintf(stderr, "[tmlqcd_hip] read_gauge_field: no pinned buffer of %llu bytes\n", mine); fclose(fp); return -1; }
      const bool ok = !fseek(fp, r.data_pos + (long)(mine * ctx->g.proc_t), SEEK_SET) && fread(buf, 1, mine, fp) == mine;
      int rc = ok ? tmhip_gauge_unpack_ildg(ctx, buf, prec_expected, calc) : 1;
      (void)hipHostFree(buf);
      if (!rc && tmhip_reduce_over_ranks(ctx)) rc = sums_combine(ctx, calc);
      if (rc) { fprintf(stderr, "Gauge file reading failed at binary part, unable to proceed.\n"); fclose(fp); return -1; }
    } else if (r.type == "scidac-checksum") {
      if (n_sum++) {
        if (io_checks) { fprintf(stderr, "In gauge file %s, multiple LIME records with name: \"scidac-checksum\" found.\nUnable to verify integrity of the gauge field data.\n", filename); fclose(fp); return -1; }
        continue;
      }
      std::string msg, a, b;
      if (lime_read_string(fp, r, msg)) { fclose(fp); return -1; }
      sum_ok = xml_value(msg, "suma", a) && xml_value(msg, "sumb", b) && sscanf(a.c_str(), "%x", &stored[0]) == 1 && sscanf(b.c_str(), "%x", &stored[1]) == 1;
    } else if (r.type == "ildg-format") {
      if (n_fmt++) {
        if (io_checks) { fprintf(stderr, "In gauge file %s, multiple LIME records with name: \"ildg-format\" found.\nUnable to verify integrity of the gauge field data.\n", filename); fclose(fp); return -1; }
        continue;
      }
      std::string msg, v;
      if (lime_read_string(fp, r, msg)) { fclose(fp); return -1; }
      const char *tags[5] = {"precision", "lx", "ly", "lz", "lt"};
      int *dst[5] = {&fprec, &fl[0], &fl[1], &fl[2], &fl[3]};
      fmt_ok = 1;
      for (int k = 0; k < 5; k++) fmt_ok = fmt_ok && xml_value(msg, tags[k], v) && sscanf(v.c_str(), "%d", dst[k]) == 1;
    } else if (r.type == "xlf-info") {
      if (lime_read_string(fp, r, xlf)) { fclose(fp); return -1; }
    } else if (r.type == "ildg-data-lfn") {
      if (lime_read_string(fp, r, lfn)) { fclose(fp); return -1; }
    }
  }
  fclose(fp);
  if (io_checks) {     // gauge_read.c:121-170
    if (!fmt_ok) { fprintf(stderr, "LIME record with name: \"ildg-format\", in gauge file %s either missing or malformed.\nUnable to verify gauge field size or precision.\n", filename); return -1; }
    if (!n_bin) { fprintf(stderr, "LIME record with name: \"ildg-binary-data\", in gauge file %s either missing or malformed.\nNo gauge field was read, unable to proceed.\n", filename); return -1; }
    if (!sum_ok) { fprintf(stderr, "LIME record with name: \"scidac-checksum\", in gauge file %s either missing or malformed.\nUnable to verify integrity of gauge field data.\n", filename); return -1; }
    if (calc[0] != stored[0]) { fprintf(stderr, "For gauge file %s, calculated and stored values for SciDAC checksum A do not match.\n", filename); return -1; }
    if (calc[1] != stored[1]) { fprintf(stderr, "For gauge file %s, calculated and stored values for SciDAC checksum B do not match.\n", filename); return -1; }
  }
  if (info) {
    memset(info, 0, sizeof(*info));
    info->gauge_read = n_bin > 0; info->suma = calc[0]; info->sumb = calc[1]; info->suma_stored = stored[0]; info->sumb_stored = stored[1];
    info->prec = fprec; info->lx = fl[0]; info->ly = fl[1]; info->lz = fl[2]; info->lt = fl[3];
    strncpy(info->xlf_info, xlf.c_str(), sizeof(info->xlf_info) - 1);
    strncpy(info->ildg_data_lfn, lfn.c_str(), sizeof(info->ildg_data_lfn) - 1);
  }
  if (n_bin && host_gf) return tmhip_gauge_download(ctx, host_gf) ? -1 : 0;
  return 0;
}

/* io/gauge_write.c:22-59 write_gauge_field(filename, prec, xlfInfo): "xlf-info" (MB 1, ME 1; the message is formatted by the caller,
 * io/utils_write_xlf_xml.c:30-63, NULL or "" leaves the record out), "ildg-format" (1, 0), "ildg-binary-data" (0, 0) from the links
 * resident in HBM, "scidac-checksum" (0, 1).  sums (may be NULL) receives the checksum written.
 * T-split ranks (collective, as the reference's parallel writer io/gauge_write_binary.c:38-170): the record is the global lattice in
 * its site order t, z, y, x -- the ranks' parts follow one another -- so rank 0 writes the framing records and every rank its part
 * at its offset; the checksum words are combined over the ranks first (which is also the barrier behind rank 0 creating the file).
 * The file is complete when the call has returned on every rank. */
int tmhip_write_gauge_field(tmhip_ctx *ctx, const char *filename, int prec, const char *xlf_info, unsigned *sums) {
  if (prec != 32 && prec != 64) TMHIP_FAIL("tmhip_write_gauge_field: precision %d (32 or 64)", prec);
  const int np = ctx->g.nproc_t, rk = ctx->g.proc_t;
  if (np > 1 && !tmhip_reduce_over_ranks(ctx)) TMHIP_FAIL("tmhip_write_gauge_field on a T-split lattice needs the communicator (tmhip_comm_init)");
  const unsigned long long part = (unsigned long long)ctx->V * 4 * 144 * prec / 64, bytes = part * np;
  void *buf = nullptr;
  TMHIP_CHECK(hipHostMalloc(&buf, part, hipHostMallocDefault));
  unsigned cs[2] = {0, 0};
  if (tmhip_gauge_pack_ildg(ctx, buf, prec, cs)) { (void)hipHostFree(buf); return 1; }
  char fmt[512], chk[512];
  snprintf(fmt, sizeof(fmt), "<?xml version=\"1.0\" encoding=\"UTF-8\"?>\n<ildgFormat xmlns=\"http://www.lqcd.org/ildg\"\n"
           "            xmlns:xsi=\"http://www.w3.org/2001/XMLSchema-instance\"\n            xsi:schemaLocation=\"http://www.lqcd.org/ildg/filefmt.xsd\">\n"
           "  <version>1.0</version>\n  <field>su3gauge</field>\n  <precision>%d</precision>\n  <lx>%d</lx>\n  <ly>%d</ly>\n  <lz>%d</lz>\n  <lt>%d</lt>\n</ildgFormat>",
           prec, ctx->g.LX, ctx->g.LY, ctx->g.LZ, ctx->g.T * np);                             // io/utils_write_ildg_format.c:30-43
  const bool have_xlf = xlf_info && xlf_info[0];
  auto padded = [](unsigned long long n) { return (n + 7) / 8 * 8; };
  const unsigned long long data_pos = (have_xlf ? 144 + padded(strlen(xlf_info)) : 0) + 144 + padded(strlen(fmt)) + 144;   // where the binary record's data begins
  FILE *fp = nullptr;
  int bad = 0;
  // A rank that cannot open the file does NOT leave: both collectives below are reached by every rank, whatever happened to it (a rank
  // that returned early left the others waiting in the gather without a bound), and its error reaches everybody through the second one.
  if (rk == 0) {
    fp = fopen(filename, "wb");
    if (!fp) { fprintf(stderr, "[tmlqcd_hip] write_gauge_field: cannot create %s\n", filename); bad = 1; }
    else {
      if (have_xlf) bad = bad || lime_write_message(fp, 1, 1, "xlf-info", xlf_info);
      bad = bad || lime_write_message(fp, 1, 0, "ildg-format", fmt);
      bad = bad || lime_write_header(fp, 0, 0, "ildg-binary-data", bytes);
      bad = bad || ftell(fp) != (long)data_pos || fflush(fp);
    }
  }
  if (np > 1 && sums_combine(ctx, cs)) { if (fp) fclose(fp); (void)hipHostFree(buf); return 1; }   // XOR over the ranks; nobody gets here before rank 0 has created the file
  if (rk != 0) {
    fp = fopen(filename, "r+b");
    if (!fp) { fprintf(stderr, "[tmlqcd_hip] write_gauge_field: rank %d cannot open %s\n", rk, filename); bad = 1; }
  }
  if (fp) bad = bad || fseek(fp, (long)(data_pos + (unsigned long long)rk * part), SEEK_SET) || (fwrite(buf, 1, part, fp) != part);
  if (rk == 0 && fp) {
    snprintf(chk, sizeof(chk), "<?xml version=\"1.0\" encoding=\"UTF-8\"?>\n<scidacChecksum>\n  <version>1.0</version>\n  <suma>%08x</suma>\n  <sumb>%08x</sumb>\n</scidacChecksum>",
             cs[0], cs[1]);                                                                     // io/utils_write_checksum.c:30-35
    static const unsigned char zero[8] = {0};
    bad = bad || fseek(fp, (long)(data_pos + bytes), SEEK_SET);
    if (!bad && bytes % 8) bad = fwrite(zero, 1, 8 - bytes % 8, fp) != 8 - bytes % 8;
    bad = bad || lime_write_message(fp, 0, 1, "scidac-checksum", chk);
  }
  if (fp) bad = fclose(fp) || bad;
  (void)hipHostFree(buf);
  if (np > 1) { unsigned done[2] = {bad ? 1u : 0u, 0}; if (sums_combine(ctx, done, true)) return 1; bad = bad || done[0]; }   // every part is in the file (and any rank's error is everybody's: OR, not XOR)
  if (bad) TMHIP_FAIL("write_gauge_field: error while writing %s", filename);
  if (sums) { sums[0] = cs[0]; sums[1] = cs[1]; }
  return 0;
}

}  // extern "C"
