/* examples/pbsim_ranks.c -- the whole-genome job of include/pbsim3_amd.h on several ranks, in C99.
 *
 *   cc -std=c99 -I include examples/pbsim_ranks.c -L pbsim3_amd/lib -lpbsim3_amd -lpthread -Wl,-rpath,$PWD/pbsim3_amd/lib -o pbsim_ranks
 *   ./pbsim_ranks ERRHMM-ONT.model genome.fa DEPTH SEED RANKS OUT_PREFIX      ->  OUT_PREFIX_0001.fq / .maf, ...
 *
 * One context per rank (here: RANKS host threads, all on GPU 0 -- on a multi-GPU node give every thread its own device, or
 * run one process per GPU and put MPI / RCCL behind the same two callbacks), every rank adds every FASTA record and runs
 * the same job; the ranks exchange integers only, through a pbsim_comm whose collectives are a pthread barrier over shared
 * memory.  Every rank pwrite()s its own byte ranges of the final files: their concatenation in offset order is byte for
 * byte what one rank produces, and on_record_done reports the same merged statistics on every rank. */
#define _XOPEN_SOURCE 600
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "pbsim3_amd.h"

#define MAX_RANKS 16
#define MAX_RECORDS 64

/* ---- the communicator: all-gather and all-reduce of int64 over a barrier -------------------------------------------- */
/* A barrier that can be given up (pbsim_comm.abort): a rank whose job failed between two exchanges will never enter the
 * next collective; it calls abort, every waiter wakes, and this and all later collectives return 0 -- the other ranks'
 * pbsim_job_run then fails too instead of waiting for ever (a pthread_barrier_t has no such exit). */
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_cv = PTHREAD_COND_INITIALIZER;
static int g_world, g_count, g_aborted;
static unsigned long g_gen;
static const int64_t *g_ptr[MAX_RANKS];

static int barrier(void) { /* 1: everybody arrived, 0: aborted */
  pthread_mutex_lock(&g_mu);
  int ok = !g_aborted;
  if (ok) {
    const unsigned long gen = g_gen;
    if (++g_count == g_world) {
      g_count = 0;
      g_gen++;
      pthread_cond_broadcast(&g_cv);
    } else {
      while (gen == g_gen && !g_aborted) pthread_cond_wait(&g_cv, &g_mu);
      ok = gen != g_gen;
    }
  }
  pthread_mutex_unlock(&g_mu);
  return ok;
}
static int comm_abort(void *user) {
  (void)user;
  pthread_mutex_lock(&g_mu);
  g_aborted = 1;
  pthread_cond_broadcast(&g_cv);
  pthread_mutex_unlock(&g_mu);
  return 1;
}

static int gather(void *user, const int64_t *send, int64_t n, int64_t *recv) {
  const int rank = *(int *)user;
  g_ptr[rank] = send;
  if (!barrier()) return 0;
  for (int q = 0; q < g_world; q++) memcpy(recv + (size_t)q * (size_t)n, g_ptr[q], (size_t)n * 8);
  return barrier(); /* nobody's send buffer goes away before everybody has read it */
}
static int reduce(void *user, int64_t *buf, int64_t n, int32_t op) {
  const int rank = *(int *)user;
  int64_t *out = malloc((size_t)(n ? n : 1) * 8);
  g_ptr[rank] = buf;
  if (!barrier()) {
    free(out);
    return 0;
  }
  for (int64_t i = 0; i < n; i++) {
    int64_t v = g_ptr[0][i];
    for (int q = 1; q < g_world; q++) {
      const int64_t x = g_ptr[q][i];
      v = op == PBSIM_OP_SUM ? v + x : op == PBSIM_OP_MIN ? (x < v ? x : v) : (x > v ? x : v);
    }
    out[i] = v;
  }
  const int ok = barrier();
  if (ok) memcpy(buf, out, (size_t)n * 8);
  free(out);
  return ok;
}

/* ---- the sink: every rank writes its pieces at their offsets ---------------------------------------------------------- */
struct files {
  const char *prefix;
  int rank;
  int fq[MAX_RECORDS + 1], maf[MAX_RECORDS + 1];
};
static int put(int fd, const char *t, int64_t n, int64_t off) {
  while (n > 0) {
    const ssize_t k = pwrite(fd, t, (size_t)n, (off_t)off);
    if (k <= 0) return 0;
    t += k, n -= k, off += k;
  }
  return 1;
}
static int on_read(void *u, int64_t rec, const char *t, int64_t n, int64_t off) { return put(((struct files *)u)->fq[rec], t, n, off); }
static int on_maf(void *u, int64_t rec, const char *t, int64_t n, int64_t off) { return put(((struct files *)u)->maf[rec], t, n, off); }
static int on_done(void *u, int64_t rec, const pbsim_stats *st, int64_t fq_bytes, int64_t maf_bytes) {
  const struct files *f = u;
  if (f->rank == 0) /* the same numbers on every rank: one of them reports */
    printf("record %lld: %lld reads, %lld bases, mean accuracy %.6f, %lld + %lld bytes\n", (long long)rec, (long long)st->res_num,
           (long long)st->res_len_total, st->res_accuracy_mean, (long long)fq_bytes, (long long)maf_bytes);
  return 1;
}

/* ---- one rank ------------------------------------------------------------------------------------------------------------ */
struct job {
  int rank;
  const char *model, *prefix;
  double depth;
  uint32_t seed;
  int n_records;
  char *seq[MAX_RECORDS];
  int64_t len[MAX_RECORDS];
  int rc;
};

static void *run_rank(void *arg) {
  struct job *j = arg;
  j->rc = 255;
  pbsim_params p;
  pbsim_params_default(&p);
  p.strategy = PBSIM_STRATEGY_WGS;
  p.method = PBSIM_METHOD_ERR;
  p.depth = j->depth;
  p.seed = j->seed;
  pbsim_ctx *ctx = pbsim_create(&p, 0);
  int ok = ctx != NULL && pbsim_load_errhmm(ctx, j->model);
  for (int r = 0; ok && r < j->n_records; r++) ok = pbsim_job_add_record(ctx, (const uint8_t *)j->seq[r], j->len[r]);
  struct files f = {j->prefix, j->rank, {0}, {0}};
  char name[4096];
  for (int r = 1; r <= j->n_records; r++) { /* rank 0 created (and truncated) the files before the threads started */
    snprintf(name, sizeof name, "%s_%04d.fq", j->prefix, r);
    f.fq[r] = open(name, O_WRONLY);
    snprintf(name, sizeof name, "%s_%04d.maf", j->prefix, r);
    f.maf[r] = open(name, O_WRONLY);
    ok = ok && f.fq[r] >= 0 && f.maf[r] >= 0;
  }
  pbsim_comm comm = {&j->rank, j->rank, g_world, gather, reduce, NULL, comm_abort};
  pbsim_record_sink sink = {&f, on_read, on_maf, on_done};
  /* every rank MUST reach the job together (the collectives inside would wait forever for a rank that gave up) */
  int64_t all_ok = ok;
  reduce(&j->rank, &all_ok, 1, PBSIM_OP_MIN);
  if (all_ok && pbsim_job_run(ctx, g_world > 1 ? &comm : NULL, &sink)) j->rc = 0;
  else fprintf(stderr, "ERROR (rank %d): %s\n", j->rank, ok ? pbsim_last_error() : "set-up failed");
  for (int r = 1; r <= j->n_records; r++) {
    if (f.fq[r] >= 0) close(f.fq[r]);
    if (f.maf[r] >= 0) close(f.maf[r]);
  }
  if (ctx) pbsim_destroy(ctx);
  return NULL;
}

int main(int argc, char **argv) {
  if (argc < 7) {
    fprintf(stderr, "usage: %s MODEL GENOME.fa DEPTH SEED RANKS OUT_PREFIX\n", argv[0]);
    return 255;
  }
  g_world = atoi(argv[5]);
  if (g_world < 1 || g_world > MAX_RANKS) return 255;
  /* the FASTA: one record per '>' line, sequence lines concatenated (get_genome_seq, pbsim.cpp:1014-1033) */
  FILE *fp = fopen(argv[2], "r");
  if (!fp) {
    fprintf(stderr, "ERROR: Cannot open file: %s\n", argv[2]);
    return 255;
  }
  static struct job jobs[MAX_RANKS];
  int n_records = 0;
  size_t cap = 0, len = 0;
  char *seq = NULL, line[10240];
  while (fgets(line, sizeof line, fp)) {
    if (line[0] == '>') {
      if (n_records == MAX_RECORDS) return 255;
      n_records++;
      seq = NULL, cap = len = 0;
      while (!strchr(line, '\n') && fgets(line, sizeof line, fp)) {} /* rest of a long header line */
      continue;
    }
    if (n_records == 0) continue;
    const size_t k = strcspn(line, "\n");
    if (len + k + 1 > cap) seq = realloc(seq, cap = (len + k + 1) * 2);
    memcpy(seq + len, line, k);
    len += k;
    jobs[0].seq[n_records - 1] = seq, jobs[0].len[n_records - 1] = (int64_t)len;
  }
  fclose(fp);
  char name[4096];
  for (int r = 1; r <= n_records; r++) {
    snprintf(name, sizeof name, "%s_%04d.fq", argv[6], r);
    close(open(name, O_WRONLY | O_CREAT | O_TRUNC, 0666));
    snprintf(name, sizeof name, "%s_%04d.maf", argv[6], r);
    close(open(name, O_WRONLY | O_CREAT | O_TRUNC, 0666));
  }
  pthread_t th[MAX_RANKS];
  for (int q = 0; q < g_world; q++) {
    jobs[q] = jobs[0];
    jobs[q].rank = q;
    jobs[q].model = argv[1];
    jobs[q].prefix = argv[6];
    jobs[q].depth = atof(argv[3]);
    jobs[q].seed = (uint32_t)atoi(argv[4]);
    jobs[q].n_records = n_records;
  }
  for (int q = 1; q < g_world; q++) pthread_create(&th[q], NULL, run_rank, &jobs[q]);
  run_rank(&jobs[0]);
  int rc = jobs[0].rc;
  for (int q = 1; q < g_world; q++) {
    pthread_join(th[q], NULL);
    rc |= jobs[q].rc;
  }
  return rc;
}
