/* crcsum -- stands in for `gzip` / `samtools view -b -o FILE -` behind the reference's popen() when a run's output is too
 * large to keep (tests/golden/make_fullsize.py): reads stdin to the end and prints "<crc32 hex> <bytes>\n" of what it read,
 * to stdout (gzip stub: the shell redirects it into the output file) or to the file named by "-o FILE" (samtools stub).
 * Test infrastructure; zlib's crc32. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

int main(int argc, char **argv) {
  const char *out = NULL;
  for (int i = 1; i + 1 < argc; i++)
    if (strcmp(argv[i], "-o") == 0) out = argv[i + 1];
  static unsigned char buf[1 << 22];
  unsigned long crc = crc32(0L, Z_NULL, 0);
  unsigned long long n = 0;
  size_t k;
  while ((k = fread(buf, 1, sizeof buf, stdin)) > 0) {
    crc = crc32(crc, buf, (uInt)k);
    n += k;
  }
  FILE *f = out ? fopen(out, "w") : stdout;
  if (!f) return 1;
  fprintf(f, "%08lx %llu\n", crc, n);
  if (out) fclose(f);
  return 0;
}
