/* examples/pbsim_rccl_rank.c -- ONE RANK of a one-process-per-GPU job in C99, on the library's own RCCL communicator.
 *
 *   cc -std=c99 -I include examples/pbsim_rccl_rank.c -L pbsim3_amd/lib -lpbsim3_amd -Wl,-rpath,$PWD/pbsim3_amd/lib -o pbsim_rccl_rank
 *   for r in 0 1 2 3 4 5 6 7; do ./pbsim_rccl_rank ERRHMM-ONT.model genome.fa 20 1 $r 8 /dev/shm/rdv.$$ out & done; wait
 *
 * Every process is started with its rank and the world size (by a shell loop, mpirun, srun ..), uses GPU `rank`, and meets the
 * others in pbsim_rccl_comm_create_file: rank 0 makes the id (ncclGetUniqueId) and publishes it in the rendezvous file, the
 * others wait for it, all enter ncclCommInitRank.  Rank 0 loads the FASTA; pbsim_job_add_record_comm broadcasts every record
 * GPU to GPU (C1, ncclBroadcast) -- the other ranks pass NULL.  pbsim_job_run then shards every round of the pipeline by read
 * block over the ranks (integers only between them: C3 ncclAllGather per round, C2 per record), and every rank pwrite()s its
 * own byte ranges of OUT_0001.fq / OUT_0001.maf .. -- the files are byte for byte what one GPU writes.  The reference has no
 * analogue (one process, pbsim.cpp:4-14); what this replaces is its record loop (pbsim.cpp:667-759). */
#define _XOPEN_SOURCE 700
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "pbsim3_amd.h"

#define MAX_RECORDS 64

struct files {
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
  if (((struct files *)u)->rank == 0) /* the same numbers on every rank: one of them reports */
    printf("record %lld: %lld reads, %lld bases, mean accuracy %.6f, %lld + %lld bytes\n", (long long)rec, (long long)st->res_num,
           (long long)st->res_len_total, st->res_accuracy_mean, (long long)fq_bytes, (long long)maf_bytes);
  return 1;
}

static int fail(int rank, const char *what) {
  fprintf(stderr, "ERROR (rank %d): %s: %s\n", rank, what, pbsim_last_error());
  return 255;
}

int main(int argc, char **argv) {
  if (argc < 9) {
    fprintf(stderr, "usage: %s MODEL GENOME.fa DEPTH SEED RANK WORLD RENDEZVOUS_FILE OUT_PREFIX\n", argv[0]);
    return 255;
  }
  const int rank = atoi(argv[5]), world = atoi(argv[6]);
  const char *prefix = argv[8];
  char what[256];
  pbsim_bind_host_to_device(rank, what, sizeof what); /* this rank's threads and pinned staging next to its GPU; before any HIP call */
  pbsim_comm *comm = pbsim_rccl_comm_create_file(argv[7], rank, world, rank);
  if (!comm) return fail(rank, "pbsim_rccl_comm_create_file");

  pbsim_params p;
  pbsim_params_default(&p);
  p.strategy = PBSIM_STRATEGY_WGS;
  p.method = PBSIM_METHOD_ERR;
  p.depth = atof(argv[3]);
  p.seed = (uint32_t)atoi(argv[4]);
  pbsim_ctx *ctx = pbsim_create(&p, rank);
  if (!ctx) return fail(rank, "pbsim_create");
  if (!pbsim_load_errhmm(ctx, argv[1])) return fail(rank, "pbsim_load_errhmm");

  /* rank 0 reads the FASTA (get_genome_seq, pbsim.cpp:1014-1033); the record count and lengths reach the others through the
   * communicator's own all-reduce, the bases GPU to GPU through its broadcast */
  static char *seq[MAX_RECORDS];
  int64_t lens[MAX_RECORDS + 1];
  memset(lens, 0, sizeof lens);
  if (rank == 0) {
    FILE *fp = fopen(argv[2], "r");
    if (!fp) {
      fprintf(stderr, "ERROR: Cannot open file: %s\n", argv[2]);
      lens[MAX_RECORDS] = -1; /* the others must not wait for records that will never come */
    } else {
      int n = 0;
      size_t cap = 0, len = 0;
      char line[10240];
      while (fgets(line, sizeof line, fp)) {
        if (line[0] == '>') {
          if (n == MAX_RECORDS) break;
          n++, cap = len = 0;
          while (!strchr(line, '\n') && fgets(line, sizeof line, fp)) {} /* rest of a long header line */
          continue;
        }
        if (n == 0) continue;
        const size_t k = strcspn(line, "\n");
        if (len + k + 1 > cap) {
          char *old = cap ? seq[n - 1] : NULL;
          cap = (len + k + 1) * 2;
          seq[n - 1] = realloc(old, cap);
        }
        memcpy(seq[n - 1] + len, line, k);
        len += k;
        lens[n - 1] = (int64_t)len;
      }
      fclose(fp);
      lens[MAX_RECORDS] = n;
    }
  }
  if (world > 1 && !comm->all_reduce_i64(comm->user, lens, MAX_RECORDS + 1, PBSIM_OP_SUM)) return fail(rank, "all_reduce_i64");
  const int n_records = (int)lens[MAX_RECORDS];
  if (n_records < 1) return 255;
  for (int r = 0; r < n_records; r++)
    if (!pbsim_job_add_record_comm(ctx, (const uint8_t *)seq[r], lens[r], comm, 0)) return fail(rank, "pbsim_job_add_record_comm");

  /* rank 0 creates (and truncates) the files; an exchange tells the others that they exist */
  struct files f;
  memset(&f, 0, sizeof f);
  f.rank = rank;
  char name[4096];
  int64_t made = 1;
  for (int r = 1; r <= n_records && rank == 0; r++) {
    snprintf(name, sizeof name, "%s_%04d.fq", prefix, r);
    const int a = open(name, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    snprintf(name, sizeof name, "%s_%04d.maf", prefix, r);
    const int b = open(name, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (a < 0 || b < 0) made = 0;
    if (a >= 0) close(a);
    if (b >= 0) close(b);
  }
  if (world > 1 && !comm->all_reduce_i64(comm->user, &made, 1, PBSIM_OP_MIN)) return fail(rank, "all_reduce_i64");
  if (!made) return 255;
  for (int r = 1; r <= n_records; r++) {
    snprintf(name, sizeof name, "%s_%04d.fq", prefix, r);
    f.fq[r] = open(name, O_WRONLY);
    snprintf(name, sizeof name, "%s_%04d.maf", prefix, r);
    f.maf[r] = open(name, O_WRONLY);
    if (f.fq[r] < 0 || f.maf[r] < 0) return 255; /* (a rank that leaves here is noticed by the others' watchdog: PBSIM_COMM_TIMEOUT_S) */
  }
  pbsim_record_sink sink = {&f, on_read, on_maf, on_done};
  const int ok = pbsim_job_run(ctx, world > 1 ? comm : NULL, &sink);
  if (!ok) fail(rank, "pbsim_job_run");
  for (int r = 1; r <= n_records; r++) close(f.fq[r]), close(f.maf[r]);
  pbsim_destroy(ctx);
  pbsim_rccl_comm_destroy(comm);
  return ok ? 0 : 255;
}
