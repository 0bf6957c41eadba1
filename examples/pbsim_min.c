/* examples/pbsim_min.c -- the smallest C host of the ABI in include/pbsim3_amd.h.
 *
 *   cc -std=c99 -I include examples/pbsim_min.c -L pbsim3_amd/lib -lpbsim3_amd -Wl,-rpath,$PWD/pbsim3_amd/lib -o pbsim_min
 *   ./pbsim_min ERRHMM-ONT.model genome.fa 20 1 > reads.fq 2> reads.maf
 *
 * WGS / ERRHMM only, one FASTA record per '>' line, FASTQ on stdout and MAF on stderr: what a maintainer's
 * call site looks like without the reference's option parsing around it.  The record is handed over
 * raw; upper-casing and homopolymer lengths are computed on the GPU (pbsim.cpp:1035-1065). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pbsim3_amd.h"

static int to_stdout(void *user, const char *text, int64_t n) {
  (void)user;
  return fwrite(text, 1, (size_t)n, stdout) == (size_t)n;
}
static int to_stderr(void *user, const char *text, int64_t n) {
  (void)user;
  return fwrite(text, 1, (size_t)n, stderr) == (size_t)n;
}

static int fail(const char *what) {
  fprintf(stderr, "ERROR: %s: %s\n", what, pbsim_last_error());
  return 255; /* the reference's exit(-1) */
}

int main(int argc, char **argv) {
  if (argc < 5) {
    fprintf(stderr, "usage: %s MODEL GENOME.fa DEPTH SEED\n", argv[0]);
    return 255;
  }
  pbsim_params p;
  pbsim_params_default(&p);
  p.strategy = PBSIM_STRATEGY_WGS;
  p.method = PBSIM_METHOD_ERR;
  p.depth = atof(argv[3]);
  p.seed = (uint32_t)atoi(argv[4]);
  pbsim_ctx *ctx = pbsim_create(&p, 0);
  if (!ctx) return fail("pbsim_create");
  if (!pbsim_load_errhmm(ctx, argv[1])) return fail("pbsim_load_errhmm");

  FILE *fp = fopen(argv[2], "r");
  if (!fp) {
    fprintf(stderr, "ERROR: Cannot open file: %s\n", argv[2]);
    return 255;
  }
  size_t cap = 1 << 20, len = 0;
  char *seq = malloc(cap), line[10240];
  int64_t record = 0;
  pbsim_sink sink = {NULL, to_stdout, to_stderr};
  int at_eof = 0;
  while (!at_eof) {
    char *got = fgets(line, sizeof line, fp);
    at_eof = got == NULL;
    if (at_eof || line[0] == '>') {
      if (record > 0 && len > 0) { /* a record is complete: simulate it */
        if (!pbsim_set_reference(ctx, (const uint8_t *)seq, (int64_t)len, record)) return fail("pbsim_set_reference");
        if (!pbsim_simulate_wgs(ctx, &sink)) return fail("pbsim_simulate_wgs");
      }
      if (!at_eof) {
        record++;
        len = 0;
        while (!strchr(line, '\n') && fgets(line, sizeof line, fp)) {} /* rest of a long header line */
      }
      continue;
    }
    size_t n = strcspn(line, "\n");
    if (len + n > cap) seq = realloc(seq, cap = (len + n) * 2);
    memcpy(seq + len, line, n);
    len += n;
  }
  fclose(fp);
  free(seq);
  pbsim_destroy(ctx);
  return 0;
}
