/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's ILDG gauge I/O for the parity tests.  Nothing under tmlqcd_amd/
 * uses it.
 *
 * What is restated, and how it is pinned:
 *   tmo_crc32, tmo_checksum_accum   io/DML_crc32.c (zlib's crc32), io/dml.c:49-60.  PINNED: bit for bit against the reference's own
 *                                   two files compiled in place (oracle/_ref/libtmref_dml.so) and against zlib.crc32.
 *   tmo_ildg_unpack / _pack         io/gauge_read_binary.c:140-200 / io/gauge_write_binary.c:150-175 (non-LEMON branches): loops
 *                                   t, z, y, x, links x, y, z, t -> mu 1, 2, 3, 0, be_to_cpu_assign[_single2double] of io/utils.c.
 *                                   The two files include c-lime's lime.h and cannot be compiled here; the loops are restated and
 *                                   the fixture tests/golden/ildg_4x4_prec{64,32}.lime pins byte order and site / link order
 *                                   (a file written by this restatement is read back by the device path and vice versa).
 *   LIME framing                    c-lime 1.3.2 (third-party, not in /root/reference, not installed: PARITY UNPINNED for the
 *                                   container itself).  Restated from its published record format: 144-byte header = magic
 *                                   0x456789ab (be32), version 1 (be16), MB/ME bits 0x80/0x40 in byte 6, data length (be64) at
 *                                   byte 8, record type (128 bytes, NUL padded) at byte 16; data padded to 8 bytes.  Record names,
 *                                   order and MB/ME bits: io/gauge_write.c:22-59, io/utils_write_*.c; reader checks: io/gauge_read.c.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* io/DML_crc32.c: table for the reflected polynomial 0xedb88320, crc = crc ^ ~0 before and after */
static uint32_t crc_tab[256];
static int crc_tab_ready = 0;
static void make_tab(void) {
  for (uint32_t n = 0; n < 256; n++) {
    uint32_t c = n;
    for (int k = 0; k < 8; k++) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
    crc_tab[n] = c;
  }
  crc_tab_ready = 1;
}
uint32_t tmo_crc32(uint32_t crc, const unsigned char *buf, size_t len) {
  if (!crc_tab_ready) make_tab();
  crc ^= 0xffffffffu;
  for (size_t i = 0; i < len; i++) crc = crc_tab[(crc ^ buf[i]) & 0xff] ^ (crc >> 8);
  return crc ^ 0xffffffffu;
}
/* io/dml.c:49-60 */
void tmo_checksum_accum(uint32_t *sums, uint32_t rank, const unsigned char *buf, size_t size) {
  const uint32_t work = tmo_crc32(0, buf, size);
  const uint32_t r29 = rank % 29, r31 = rank % 31;
  sums[0] ^= r29 ? (work << r29 | work >> (32 - r29)) : work;
  sums[1] ^= r31 ? (work << r31 | work >> (32 - r31)) : work;
}

static double be_double(const unsigned char *p) {
  uint64_t v = 0;
  for (int i = 0; i < 8; i++) v = (v << 8) | p[i];
  double d; memcpy(&d, &v, 8); return d;
}
static float be_float(const unsigned char *p) {
  uint32_t v = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
  float f; memcpy(&f, &v, 4); return f;
}
static void put_be_double(unsigned char *p, double d) {
  uint64_t v; memcpy(&v, &d, 8);
  for (int i = 7; i >= 0; i--) { p[i] = (unsigned char)(v & 0xff); v >>= 8; }
}
static void put_be_float(unsigned char *p, float f) {
  uint32_t v; memcpy(&v, &f, 4);
  for (int i = 3; i >= 0; i--) { p[i] = (unsigned char)(v & 0xff); v >>= 8; }
}

/* io/gauge_read_binary.c:157-192: gf = double [T LX LY LZ][4][18] (g_gauge_field layout, ix = ((t LX + x) LY + y) LZ + z);
 * rank0 = DML rank of the first file site of this rank (T-split: proc_t T LZ LY LX) */
void tmo_ildg_unpack(const unsigned char *file, int prec, int T, int LX, int LY, int LZ, uint32_t rank0, double *gf, uint32_t *sums) {
  const size_t sb = prec == 64 ? 576 : 288;
  sums[0] = sums[1] = 0;
  for (int t = 0; t < T; t++) for (int z = 0; z < LZ; z++) for (int y = 0; y < LY; y++) for (int x = 0; x < LX; x++) {
    const size_t f = (((size_t)t * LZ + z) * LY + y) * LX + x;
    const unsigned char *cur = file + f * sb;
    tmo_checksum_accum(sums, rank0 + (uint32_t)f, cur, sb);
    const size_t ix = (((size_t)t * LX + x) * LY + y) * LZ + z;
    for (int j = 0; j < 4; j++) {
      double *dst = gf + (ix * 4 + (size_t)((j + 1) & 3)) * 18;          /* tmp[0..2] -> gf[..][1..3], tmp[3] -> gf[..][0] */
      for (int k = 0; k < 18; k++) dst[k] = prec == 64 ? be_double(cur + ((size_t)j * 18 + k) * 8) : (double)be_float(cur + ((size_t)j * 18 + k) * 4);
    }
  }
}
/* io/gauge_write_binary.c:150-175 */
void tmo_ildg_pack(unsigned char *file, int prec, int T, int LX, int LY, int LZ, uint32_t rank0, const double *gf, uint32_t *sums) {
  const size_t sb = prec == 64 ? 576 : 288;
  sums[0] = sums[1] = 0;
  for (int t = 0; t < T; t++) for (int z = 0; z < LZ; z++) for (int y = 0; y < LY; y++) for (int x = 0; x < LX; x++) {
    const size_t f = (((size_t)t * LZ + z) * LY + y) * LX + x;
    unsigned char *cur = file + f * sb;
    const size_t ix = (((size_t)t * LX + x) * LY + y) * LZ + z;
    for (int j = 0; j < 4; j++) {
      const double *src = gf + (ix * 4 + (size_t)((j + 1) & 3)) * 18;
      for (int k = 0; k < 18; k++) {
        if (prec == 64) put_be_double(cur + ((size_t)j * 18 + k) * 8, src[k]);
        else put_be_float(cur + ((size_t)j * 18 + k) * 4, (float)src[k]);
      }
    }
    tmo_checksum_accum(sums, rank0 + (uint32_t)f, cur, sb);
  }
}

/* ---- LIME framing ---------------------------------------------------------------------------------------------------------- */
static int lime_put_header(FILE *fp, int mb, int me, const char *type, uint64_t bytes) {
  unsigned char h[144];
  memset(h, 0, sizeof(h));
  h[0] = 0x45; h[1] = 0x67; h[2] = 0x89; h[3] = 0xab;
  h[4] = 0; h[5] = 1;
  h[6] = (unsigned char)((mb ? 0x80 : 0) | (me ? 0x40 : 0));
  for (int i = 0; i < 8; i++) h[8 + i] = (unsigned char)(bytes >> (56 - 8 * i));
  strncpy((char *)h + 16, type, 127);
  return fwrite(h, 1, 144, fp) == 144 ? 0 : -1;
}
static int lime_put_data(FILE *fp, const void *d, uint64_t bytes) {
  static const unsigned char zero[8];
  if (bytes && fwrite(d, 1, bytes, fp) != bytes) return -1;
  const uint64_t pad = (8 - bytes % 8) % 8;
  return pad && fwrite(zero, 1, pad, fp) != pad ? -1 : 0;
}
/* io/gauge_write.c:22-59 for one rank; xlf may be NULL */
int tmo_write_gauge_field(const char *filename, int prec, int T, int LX, int LY, int LZ, const double *gf, const char *xlf, uint32_t *sums) {
  const uint64_t bytes = (uint64_t)T * LX * LY * LZ * (prec == 64 ? 576 : 288);
  unsigned char *buf = malloc(bytes);
  if (!buf) return -1;
  tmo_ildg_pack(buf, prec, T, LX, LY, LZ, 0, gf, sums);
  FILE *fp = fopen(filename, "wb");
  if (!fp) { free(buf); return -1; }
  char fmt[512], chk[512];
  snprintf(fmt, sizeof(fmt), "<?xml version=\"1.0\" encoding=\"UTF-8\"?>\n<ildgFormat xmlns=\"http://www.lqcd.org/ildg\"\n"
           "            xmlns:xsi=\"http://www.w3.org/2001/XMLSchema-instance\"\n            xsi:schemaLocation=\"http://www.lqcd.org/ildg/filefmt.xsd\">\n"
           "  <version>1.0</version>\n  <field>su3gauge</field>\n  <precision>%d</precision>\n  <lx>%d</lx>\n  <ly>%d</ly>\n  <lz>%d</lz>\n  <lt>%d</lt>\n</ildgFormat>",
           prec, LX, LY, LZ, T);
  snprintf(chk, sizeof(chk), "<?xml version=\"1.0\" encoding=\"UTF-8\"?>\n<scidacChecksum>\n  <version>1.0</version>\n  <suma>%08x</suma>\n  <sumb>%08x</sumb>\n</scidacChecksum>",
           sums[0], sums[1]);
  int bad = 0;
  if (xlf && xlf[0]) bad |= lime_put_header(fp, 1, 1, "xlf-info", strlen(xlf)) || lime_put_data(fp, xlf, strlen(xlf));
  bad |= lime_put_header(fp, 1, 0, "ildg-format", strlen(fmt)) || lime_put_data(fp, fmt, strlen(fmt));
  bad |= lime_put_header(fp, 0, 0, "ildg-binary-data", bytes) || lime_put_data(fp, buf, bytes);
  bad |= lime_put_header(fp, 0, 1, "scidac-checksum", strlen(chk)) || lime_put_data(fp, chk, strlen(chk));
  bad |= fclose(fp);
  free(buf);
  return bad ? -1 : 0;
}
/* io/gauge_read.c:28-198 for one rank, checks on: 0 and the field, or -1.  out[0..1] calculated, out[2..3] stored checksum. */
int tmo_read_gauge_field(const char *filename, int prec, int T, int LX, int LY, int LZ, double *gf, uint32_t *out) {
  FILE *fp = fopen(filename, "rb");
  if (!fp) return -1;
  unsigned char h[144];
  long pos = 0;
  int have_bin = 0, have_sum = 0, have_fmt = 0;
  uint32_t calc[2] = {0, 0}, stored[2] = {0, 0};
  for (;;) {
    if (fseek(fp, pos, SEEK_SET)) break;
    const size_t n = fread(h, 1, 144, fp);
    if (n == 0) break;
    if (n != 144 || h[0] != 0x45 || h[1] != 0x67 || h[2] != 0x89 || h[3] != 0xab || h[4] != 0 || h[5] != 1) { fclose(fp); return -1; }
    uint64_t bytes = 0;
    for (int i = 0; i < 8; i++) bytes = (bytes << 8) | h[8 + i];
    h[143] = 0;
    const char *type = (const char *)h + 16;
    if (!strcmp(type, "ildg-binary-data")) {
      if (have_bin++) { fclose(fp); return -1; }
      if (bytes != (uint64_t)T * LX * LY * LZ * (prec == 64 ? 576 : 288)) { fclose(fp); return -1; }
      unsigned char *buf = malloc(bytes);
      if (!buf || fread(buf, 1, bytes, fp) != bytes) { free(buf); fclose(fp); return -1; }
      tmo_ildg_unpack(buf, prec, T, LX, LY, LZ, 0, gf, calc);
      free(buf);
    } else if (!strcmp(type, "scidac-checksum") || !strcmp(type, "ildg-format")) {
      char *msg = calloc(bytes + 1, 1);
      if (!msg || fread(msg, 1, bytes, fp) != bytes) { free(msg); fclose(fp); return -1; }
      if (type[0] == 's') {
        if (have_sum++) { free(msg); fclose(fp); return -1; }
        const char *a = strstr(msg, "<suma>"), *b = strstr(msg, "<sumb>");
        if (!a || !b || sscanf(a + 6, "%x", &stored[0]) != 1 || sscanf(b + 6, "%x", &stored[1]) != 1) have_sum = -1000;
      } else {
        if (have_fmt++) { free(msg); fclose(fp); return -1; }
        int p = 0, l[4] = {0, 0, 0, 0};
        const char *tags[5] = {"<precision>", "<lx>", "<ly>", "<lz>", "<lt>"};
        int *dst[5] = {&p, &l[0], &l[1], &l[2], &l[3]};
        for (int k = 0; k < 5; k++) { const char *q = strstr(msg, tags[k]); if (!q || sscanf(q + strlen(tags[k]), "%d", dst[k]) != 1) have_fmt = -1000; }
      }
      free(msg);
    }
    pos += 144 + (long)((bytes + 7) / 8 * 8);
  }
  fclose(fp);
  out[0] = calc[0]; out[1] = calc[1]; out[2] = stored[0]; out[3] = stored[1];
  if (have_fmt != 1 || have_bin != 1 || have_sum != 1) return -1;
  if (calc[0] != stored[0] || calc[1] != stored[1]) return -1;
  return 0;
}
