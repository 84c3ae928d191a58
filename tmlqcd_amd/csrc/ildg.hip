// ILDG gauge configurations in and out of HBM (io/gauge_read.c, io/gauge_write.c).
//
// The reference reads the "ildg-binary-data" record of a LIME file site by site on the host (io/gauge_read_binary.c:140-200:
// loops t, z, y, x; per site four links in the file's order x, y, z, t, big-endian, 64 or 32 bit; be_to_cpu_assign into
// g_gauge_field[g_ipt[t][x][y][z]][mu]) and accumulates the SciDAC checksum over the file bytes of every site (io/dml.c:49-60:
// CRC-32 of the site's bytes, rotated by rank % 29 and rank % 31, XOR-ed over sites).  Here the record's bytes go to the device as
// they lie in the file and ONE kernel does the rest in HBM: byte swap, precision conversion, the x <-> z transposition into the
// lexicographic order of g_gauge_field, the link rotation, and the checksum -- byte work bound by HBM, so a wave stages 64 file
// sites (36 KB, read as 1 KiB coalesced rows) in LDS, every lane then walks ITS site's bytes for the CRC (site stride 145 words:
// conflict-free), and the 64 x 36 complex numbers leave LDS in flat order so that the stores are whole 576-byte link records.
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

#define ILDG_SITES 64          /* file sites per block = one wave */
#define ILDG_STRIDE 145        /* words per staged site (144 data words + 1: odd stride, lanes hit different banks) */

typedef unsigned v4u __attribute__((ext_vector_type(4)));
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

// CRC-32 of `words` 32-bit words of LDS (memory byte order), DML_crc32(0, buf, 4 * words), four bytes per step ("slicing by 4"):
// tab[0] is the byte table of DML_crc32.c, tab[k][n] = tab[0][n] advanced by k more zero bytes, so one step is four INDEPENDENT
// look-ups instead of a chain of four -- the dependent chain of a 576-byte site is 144 LDS round trips, not 576
__device__ __forceinline__ unsigned ildg_crc(const unsigned *w, int words, const unsigned (*tab)[256]) {
  unsigned c = 0xffffffffu;
  for (int i = 0; i < words; i++) {
    const unsigned v = c ^ w[i];
    c = tab[3][v & 0xff] ^ tab[2][(v >> 8) & 0xff] ^ tab[1][(v >> 16) & 0xff] ^ tab[0][v >> 24];
  }
  return c ^ 0xffffffffu;
}
__device__ __forceinline__ void ildg_crc_tables(unsigned (*tab)[256], int lane) {
  for (int n = lane; n < 256; n += ILDG_SITES) {
    unsigned c = crc_table_entry((unsigned)n);
    tab[0][n] = c;
    for (int k = 1; k < 4; k++) { c = crc_table_entry(c & 0xff) ^ (c >> 8); tab[k][n] = c; }
  }
}
// io/dml.c:49-60 for the lane's site, then XOR over the wave and into the two global words
__device__ __forceinline__ void ildg_checksum_accum(unsigned crc, unsigned long long rank64, bool active, unsigned *sums) {
  const unsigned rank = (unsigned)rank64;      // DML_SiteRank is uint32_t (io/dml.h:40)
  const unsigned r29 = rank % 29, r31 = rank % 31;
  unsigned a = active ? (crc << r29 | crc >> (32 - r29)) : 0u, b = active ? (crc << r31 | crc >> (32 - r31)) : 0u;   // (rank % 29 == 0: x >> 32 is x on this
  if (active && r29 == 0) a = crc;                                                                                 //  target too, but do not rely on it)
  if (active && r31 == 0) b = crc;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { a ^= __shfl_xor(a, off, 64); b ^= __shfl_xor(b, off, 64); }
  if ((threadIdx.x & 63) == 0) { atomicXor(sums, a); atomicXor(sums + 1, b); }
}

// file bytes -> lexicographic links [ix][4][9] complex double.  WPS = 32-bit words per file site: 144 (64-bit data) or 72.
template <int WPS>
__global__ __launch_bounds__(ILDG_SITES) void ildg_unpack_kernel(const unsigned *__restrict__ file, v2d *__restrict__ raw, IldgGeom g, size_t nsites, unsigned *sums) {
  __shared__ unsigned tab[4][256];
  __shared__ unsigned st[ILDG_SITES * ILDG_STRIDE];
  const int lane = threadIdx.x;
  ildg_crc_tables(tab, lane);
  const size_t s0 = (size_t)blockIdx.x * ILDG_SITES;
  const int ns = (int)(nsites - s0 < ILDG_SITES ? nsites - s0 : ILDG_SITES);
  // 1. stage: the block's ns * WPS words are contiguous in the file; 16 bytes per lane and step
  const v4u *src = reinterpret_cast<const v4u *>(file + s0 * WPS);
  const int nq = ns * WPS / 4;
  for (int q = lane; q < nq; q += ILDG_SITES) {
    const v4u v = __builtin_nontemporal_load(src + q);
    const int w = 4 * q, s = w / WPS, o = w - s * WPS;     // (WPS % 4 == 0: the four words stay in one site)
    unsigned *d = st + s * ILDG_STRIDE + o;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  __syncthreads();
  // 2. checksum: every lane its own site, over the bytes as they lie in the file
  {
    const bool active = lane < ns;
    const unsigned crc = active ? ildg_crc(st + lane * ILDG_STRIDE, WPS, tab) : 0u;
    ildg_checksum_accum(crc, g.rank0 + s0 + lane, active, sums);
  }
  // 3. unpack in flat order: 36 complex numbers per site; file link j (x, y, z, t) is mu = (j + 1) % 4 (gauge_read_binary.c:187-190)
  for (int k = lane; k < ns * 36; k += ILDG_SITES) {
    const int s = k / 36, e = k - s * 36, j = e / 9, c = e - j * 9;
    const unsigned *w = st + s * ILDG_STRIDE;
    v2d val;
    if (WPS == 144) {
      const unsigned *p = w + 4 * e;
      const unsigned long long re = ((unsigned long long)bswap32(p[0]) << 32) | bswap32(p[1]), im = ((unsigned long long)bswap32(p[2]) << 32) | bswap32(p[3]);
      val = v2d{__longlong_as_double((long long)re), __longlong_as_double((long long)im)};
    } else {
      const unsigned *p = w + 2 * e;
      val = v2d{(double)__uint_as_float(bswap32(p[0])), (double)__uint_as_float(bswap32(p[1]))};   // be_to_cpu_assign_single2double
    }
    const size_t ix = ildg_site_index(g, s0 + s);
    raw[(ix * 4 + ((j + 1) & 3)) * 9 + c] = val;
  }
}

// lexicographic links -> file bytes (io/gauge_write_binary.c:150-175) and their checksum
template <int WPS>
__global__ __launch_bounds__(ILDG_SITES) void ildg_pack_kernel(unsigned *__restrict__ file, const v2d *__restrict__ raw, IldgGeom g, size_t nsites, unsigned *sums) {
  __shared__ unsigned tab[4][256];
  __shared__ unsigned st[ILDG_SITES * ILDG_STRIDE];
  const int lane = threadIdx.x;
  ildg_crc_tables(tab, lane);
  const size_t s0 = (size_t)blockIdx.x * ILDG_SITES;
  const int ns = (int)(nsites - s0 < ILDG_SITES ? nsites - s0 : ILDG_SITES);
  for (int k = lane; k < ns * 36; k += ILDG_SITES) {
    const int s = k / 36, e = k - s * 36, j = e / 9, c = e - j * 9;
    const size_t ix = ildg_site_index(g, s0 + s);
    const v2d val = raw[(ix * 4 + ((j + 1) & 3)) * 9 + c];
    unsigned *w = st + s * ILDG_STRIDE;
    if (WPS == 144) {
      const unsigned long long re = (unsigned long long)__double_as_longlong(val.x), im = (unsigned long long)__double_as_longlong(val.y);
      unsigned *p = w + 4 * e;
      p[0] = bswap32((unsigned)(re >> 32)); p[1] = bswap32((unsigned)re); p[2] = bswap32((unsigned)(im >> 32)); p[3] = bswap32((unsigned)im);
    } else {
      unsigned *p = w + 2 * e;
      p[0] = bswap32(__float_as_uint((float)val.x)); p[1] = bswap32(__float_as_uint((float)val.y));   // be_to_cpu_assign_double2single
    }
  }
  __syncthreads();
  {
    const bool active = lane < ns;
    const unsigned crc = active ? ildg_crc(st + lane * ILDG_STRIDE, WPS, tab) : 0u;
    ildg_checksum_accum(crc, g.rank0 + s0 + lane, active, sums);
  }
  v4u *dst = reinterpret_cast<v4u *>(file + s0 * WPS);
  const int nq = ns * WPS / 4;
  for (int q = lane; q < nq; q += ILDG_SITES) {
    const int w = 4 * q, s = w / WPS, o = w - s * WPS;
    const unsigned *d = st + s * ILDG_STRIDE + o;
    dst[q] = v4u{d[0], d[1], d[2], d[3]};
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
  if (!ctx->io_sums) TMHIP_CHECK(hipMalloc((void **)&ctx->io_sums, 2 * sizeof(unsigned)));
  TMHIP_CHECK(hipMemsetAsync(ctx->io_sums, 0, 2 * sizeof(unsigned), ctx->stream));
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
  TMHIP_CHECK(hipSetDevice(ctx->device));
  const size_t nsites = (size_t)ctx->V, fbytes = nsites * 4 * 9 * (prec == 64 ? 16 : 8), gbytes = (size_t)ctx->VPR * 36 * sizeof(v2d);
  if (tmhip_stage_reserve(ctx, fbytes) || sums_reserve(ctx)) return 1;
  if (!ctx->gauge_raw) TMHIP_CHECK(hipMalloc((void **)&ctx->gauge_raw, gbytes));
  TMHIP_CHECK(hipMemcpyAsync(ctx->stage, file_bytes, fbytes, hipMemcpyHostToDevice, ctx->stream));
  const dim3 grid((unsigned)((nsites + ILDG_SITES - 1) / ILDG_SITES));
  if (prec == 64) hipLaunchKernelGGL(ildg_unpack_kernel<144>, grid, dim3(ILDG_SITES), 0, ctx->stream, (const unsigned *)ctx->stage, ctx->gauge_raw, io_geom(ctx), nsites, ctx->io_sums);
  else hipLaunchKernelGGL(ildg_unpack_kernel<72>, grid, dim3(ILDG_SITES), 0, ctx->stream, (const unsigned *)ctx->stage, ctx->gauge_raw, io_geom(ctx), nsites, ctx->io_sums);
  TMHIP_CHECK(hipGetLastError());
  ctx->gauge_raw_valid = true;
  if (ctx->g.nproc_t > 1 && tmhip_exchange_gauge_halo(ctx)) return 1;
  ctx->sw_set = false; ctx->clover_set = false; ctx->clover32_set = false;      // clover blocks belong to the old links
  if (tmhip_resort_gauge(ctx)) return 1;
  unsigned h[2];
  TMHIP_CHECK(hipMemcpyAsync(h, ctx->io_sums, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
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
  const dim3 grid((unsigned)((nsites + ILDG_SITES - 1) / ILDG_SITES));
  if (prec == 64) hipLaunchKernelGGL(ildg_pack_kernel<144>, grid, dim3(ILDG_SITES), 0, ctx->stream, (unsigned *)ctx->stage, (const v2d *)ctx->gauge_raw, io_geom(ctx), nsites, ctx->io_sums);
  else hipLaunchKernelGGL(ildg_pack_kernel<72>, grid, dim3(ILDG_SITES), 0, ctx->stream, (unsigned *)ctx->stage, (const v2d *)ctx->gauge_raw, io_geom(ctx), nsites, ctx->io_sums);
  TMHIP_CHECK(hipGetLastError());
  unsigned h[2];
  TMHIP_CHECK(hipMemcpyAsync(file_bytes, ctx->stage, fbytes, hipMemcpyDeviceToHost, ctx->stream));
  TMHIP_CHECK(hipMemcpyAsync(h, ctx->io_sums, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  TMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (sums) { sums[0] = h[0]; sums[1] = h[1]; }
  return 0;
}

/* io/gauge_read.c:28-198 read_gauge_field(filename, gf) for a single rank: walks the LIME records ("ildg-format", "ildg-binary-data",
 * "scidac-checksum", "xlf-info", "ildg-data-lfn"), checks record multiplicity, lattice size, precision and the checksum exactly as the
 * reference does (io_checks = 0 is g_disable_IO_checks), leaves the links in HBM and -- host_gf != NULL -- in the host's
 * g_gauge_field layout ([VOLUMEPLUSRAND][4] su3).  prec_expected = gauge_precision_read_flag (64 / 32).  Returns 0, or -1 like the
 * reference (messages on stderr); info (may be NULL) receives what GaugeInfo would hold. */
int tmhip_read_gauge_field(tmhip_ctx *ctx, const char *filename, int prec_expected, int io_checks, void *host_gf, tmhip_gauge_info *info) {
  if (ctx->g.nproc_t != 1) TMHIP_FAIL("tmhip_read_gauge_field: single rank only (T-split ranks hand their part of the record to tmhip_gauge_unpack_ildg)");
  FILE *fp = fopen(filename, "rb");
  if (!fp) { fprintf(stderr, "[tmlqcd_hip] read_gauge_field: cannot open %s\n", filename); return -1; }
  std::vector<LimeRecord> recs;
  if (lime_scan(fp, recs)) { fprintf(stderr, "[tmlqcd_hip] read_gauge_field: %s is not a LIME file (bad record header)\n", filename); fclose(fp); return -1; }
  const int L[4] = {ctx->g.LX, ctx->g.LY, ctx->g.LZ, ctx->g.T};
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
      const unsigned long long want = (unsigned long long)ctx->V * 4 * 144 / (prec_expected == 64 ? 1 : 2);
      if (r.bytes != want) {     // gauge_read_binary.c:148-155
        fprintf(stderr, "Lattice size and precision found in data file do not match those requested at input.\nExpected LX = %d, LY = %d, LZ = %d, LT = %d, and %s precision.\n"
                        "Expected %llu bytes, found %llu bytes.\nGauge file reading failed at binary part, unable to proceed.\n",
                L[0], L[1], L[2], L[3], prec_expected == 64 ? "double" : "single", want, r.bytes);
        fclose(fp); return -1;
      }
      void *buf = nullptr;
      if (hipHostMalloc(&buf, r.bytes, hipHostMallocDefault) != hipSuccess) { fprintf(stderr, "[tmlqcd_hip] read_gauge_field: no pinned buffer of %llu bytes\n", r.bytes); fclose(fp); return -1; }
      const bool ok = !fseek(fp, r.data_pos, SEEK_SET) && fread(buf, 1, r.bytes, fp) == r.bytes;
      const int rc = ok ? tmhip_gauge_unpack_ildg(ctx, buf, prec_expected, calc) : 1;
      (void)hipHostFree(buf);
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

/* io/gauge_write.c:22-59 write_gauge_field(filename, prec, xlfInfo) for a single rank: "xlf-info" (MB 1, ME 1; the message is
 * formatted by the caller, io/utils_write_xlf_xml.c:30-63, NULL or "" leaves the record out), "ildg-format" (1, 0), "ildg-binary-data"
 * (0, 0) from the links resident in HBM, "scidac-checksum" (0, 1).  sums (may be NULL) receives the checksum written. */
int tmhip_write_gauge_field(tmhip_ctx *ctx, const char *filename, int prec, const char *xlf_info, unsigned *sums) {
  if (ctx->g.nproc_t != 1) TMHIP_FAIL("tmhip_write_gauge_field: single rank only (T-split ranks take their part of the record from tmhip_gauge_pack_ildg)");
  if (prec != 32 && prec != 64) TMHIP_FAIL("tmhip_write_gauge_field: precision %d (32 or 64)", prec);
  const unsigned long long bytes = (unsigned long long)ctx->V * 4 * 144 * prec / 64;
  void *buf = nullptr;
  TMHIP_CHECK(hipHostMalloc(&buf, bytes, hipHostMallocDefault));
  unsigned cs[2] = {0, 0};
  if (tmhip_gauge_pack_ildg(ctx, buf, prec, cs)) { (void)hipHostFree(buf); return 1; }
  FILE *fp = fopen(filename, "wb");
  if (!fp) { (void)hipHostFree(buf); TMHIP_FAIL("write_gauge_field: cannot create %s", filename); }
  char fmt[512], chk[512];
  snprintf(fmt, sizeof(fmt), "<?xml version=\"1.0\" encoding=\"UTF-8\"?>\n<ildgFormat xmlns=\"http://www.lqcd.org/ildg\"\n"
           "            xmlns:xsi=\"http://www.w3.org/2001/XMLSchema-instance\"\n            xsi:schemaLocation=\"http://www.lqcd.org/ildg/filefmt.xsd\">\n"
           "  <version>1.0</version>\n  <field>su3gauge</field>\n  <precision>%d</precision>\n  <lx>%d</lx>\n  <ly>%d</ly>\n  <lz>%d</lz>\n  <lt>%d</lt>\n</ildgFormat>",
           prec, ctx->g.LX, ctx->g.LY, ctx->g.LZ, ctx->g.T);                                  // io/utils_write_ildg_format.c:30-43
  snprintf(chk, sizeof(chk), "<?xml version=\"1.0\" encoding=\"UTF-8\"?>\n<scidacChecksum>\n  <version>1.0</version>\n  <suma>%08x</suma>\n  <sumb>%08x</sumb>\n</scidacChecksum>",
           cs[0], cs[1]);                                                                       // io/utils_write_checksum.c:30-35
  int bad = 0;
  if (xlf_info && xlf_info[0]) bad = bad || lime_write_message(fp, 1, 1, "xlf-info", xlf_info);
  bad = bad || lime_write_message(fp, 1, 0, "ildg-format", fmt);
  bad = bad || lime_write_header(fp, 0, 0, "ildg-binary-data", bytes) || lime_write_data(fp, buf, bytes);
  bad = bad || lime_write_message(fp, 0, 1, "scidac-checksum", chk);
  bad = fclose(fp) || bad;
  (void)hipHostFree(buf);
  if (bad) TMHIP_FAIL("write_gauge_field: error while writing %s", filename);
  if (sums) { sums[0] = cs[0]; sums[1] = cs[1]; }
  return 0;
}

}  // extern "C"
