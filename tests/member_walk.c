/* member_walk.c -- test helper of tests/test_gpu_fullsize.py (compiled there with the host cc, linked against zlib).
 * Walks a buffer of whole BGZF / gzip members (RFC 1952 + the 'BC' extra subfield of SAMv1 4.1), folds the members' own
 * CRC-32 / ISIZE trailers into the CRC-32 and length of the concatenated text (zlib's crc32_combine), and inflates every
 * sample_every-th member to check that its trailer describes what it holds.  Returns the number of members, or -1 - the
 * offset of the first byte that is wrong. */
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* crc32_combine(a, b, 32768) for the full-size members: zlib before 1.2.12 squares 32 x 32 GF(2) matrices on every call
 * (~20 us; two million members per record).  The operator is linear in a: four 256-entry tables of it, built once. */
static unsigned long shift_tab[4][256];
static int shift_tab_ready = 0;
static void build_shift_tab(void) {
  for (int k = 0; k < 4; k++)
    for (unsigned long b = 0; b < 256; b++) shift_tab[k][b] = crc32_combine(b << (8 * k), 0, 32768);
  shift_tab_ready = 1;
}
static unsigned long combine(unsigned long a, unsigned long b, long len_b) {
  if (len_b != 32768) return crc32_combine(a, b, len_b);
  return shift_tab[0][a & 255] ^ shift_tab[1][(a >> 8) & 255] ^ shift_tab[2][(a >> 16) & 255] ^ shift_tab[3][(a >> 24) & 255] ^ b;
}

long walk_members(const unsigned char *buf, long n, unsigned long *crc_out, long *len_out, long first_index, long sample_every) {
  if (!shift_tab_ready) build_shift_tab();
  unsigned long crc = 0;
  long total = 0, at = 0, count = 0;
  unsigned char *tmp = NULL;
  while (at < n) {
    if (n - at < 26 || buf[at] != 31 || buf[at + 1] != 139 || buf[at + 2] != 8 || buf[at + 3] != 4 || buf[at + 12] != 'B' ||
        buf[at + 13] != 'C')
      return -1 - at;
    const long size = ((long)buf[at + 16] | ((long)buf[at + 17] << 8)) + 1;
    if (at + size > n) return -1 - at;
    const unsigned char *t = buf + at + size - 8;
    const unsigned long mcrc = (unsigned long)t[0] | ((unsigned long)t[1] << 8) | ((unsigned long)t[2] << 16) | ((unsigned long)t[3] << 24);
    const long isize = (long)t[4] | ((long)t[5] << 8) | ((long)t[6] << 16) | ((long)t[7] << 24);
    if (isize > 65536) return -1 - at;
    if (sample_every > 0 && (first_index + count) % sample_every == 0) {
      if (!tmp) tmp = malloc(65536);
      z_stream z;
      memset(&z, 0, sizeof z);
      if (inflateInit2(&z, -15) != Z_OK) return -1 - at;
      z.next_in = (unsigned char *)buf + at + 18;
      z.avail_in = (unsigned)(size - 26);
      z.next_out = tmp;
      z.avail_out = 65536;
      const int rc = inflate(&z, Z_FINISH);
      const long got = 65536 - (long)z.avail_out;
      inflateEnd(&z);
      if (rc != Z_STREAM_END || got != isize || crc32(0L, tmp, (unsigned)got) != mcrc) {
        free(tmp);
        return -1 - at;
      }
    }
    crc = combine(crc, mcrc, isize);
    total += isize;
    at += size;
    count++;
  }
  free(tmp);
  *crc_out = crc;
  *len_out = total;
  return count;
}

unsigned long fold(unsigned long crc_a, unsigned long crc_b, long len_b) { return crc32_combine(crc_a, crc_b, len_b); }
