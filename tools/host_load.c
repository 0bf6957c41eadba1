/* host_load -- the host-memory traffic of the OTHER ranks of an N-GPU node, emulated on a one-GPU box.
 *
 * A rank of the job receives ~47 GB/s of gzip members from its GPU into page-locked memory on its GPU's NUMA node and a sink
 * reads them back.  On a node with eight GPUs eight such streams share the host's memory controllers; a box with
 * one GPU has one.  This program stands in for the other N - 1: per virtual rank
 *     writers   stream `rate` GB/s into a ring of buffers with NON-TEMPORAL stores (what a DMA engine's writes are to the
 *               memory controller: no cache allocation), throttled to the rate;
 *     readers   read every buffer back once behind the writers (the sink: cached loads, summed so the loop is not elided);
 * threads and pages of virtual rank v bound to the NUMA node GPU v would be attached to (the box's nodes in equal shares, as on
 * an 8-GPU node: the first half of the GPUs on the first socket).  Runs until stdin closes or `seconds` have passed, then prints
 * what it moved per virtual rank.
 *
 *   cc -O2 -pthread -o host_load tools/host_load.c
 *   ./host_load RANKS RATE_GBS [SECONDS] [REAL_RANK] [REAL_NODE]
 *        e.g. ./host_load 8 47 30 0 1   (ranks 1..7 are emulated; the real rank's GPU sits on NUMA node 1: with two nodes three
 *        virtual ranks share that node with it and four take the other, as on an 8-GPU node)
 */
#define _GNU_SOURCE
#include <emmintrin.h>
#include <pthread.h>
#include <signal.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#define CHUNK (8u << 20)
#define RING 64                /* 512 MiB per virtual rank */
#define MAX_T 8
static int WRITERS = 4, READERS = 3;   /* per virtual rank; HOST_LOAD_WRITERS / HOST_LOAD_READERS (a box with a CPU quota: fewer, faster threads) */

static volatile int g_stop = 0;
static int g_read_only = 0;

static double now_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + t.tv_nsec * 1e-9;
}

typedef struct {
  int rank, node, n_nodes;
  double rate;                 /* bytes per second for this virtual rank */
  char *ring;
  volatile long written[MAX_T], read_[MAX_T];   /* chunks */
  cpu_set_t cpus;
  int have_cpus;
} VRank;

typedef struct {
  VRank *v;
  int k;
} Arg;

static int node_cpus(int node, cpu_set_t *set) {
  char path[128], buf[4096];
  snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
  FILE *f = fopen(path, "r");
  if (!f) return 0;
  if (!fgets(buf, sizeof buf, f)) {
    fclose(f);
    return 0;
  }
  fclose(f);
  CPU_ZERO(set);
  int n = 0;
  for (char *p = buf; *p && *p != '\n';) {
    char *e;
    long a = strtol(p, &e, 10), b = a;
    if (e == p) break;
    if (*e == '-') b = strtol(e + 1, &e, 10);
    for (long c = a; c <= b; c++) {
      CPU_SET((int)c, set);
      n++;
    }
    p = (*e == ',') ? e + 1 : e;
  }
  return n;
}

static int count_nodes(void) {
  int n = 0;
  char path[128];
  for (;; n++) {
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d", n);
    if (access(path, F_OK) != 0) break;
  }
  return n > 0 ? n : 1;
}

static void bind_here(const VRank *v) {
  if (v->have_cpus) (void)sched_setaffinity(0, sizeof v->cpus, &v->cpus);
#ifdef SYS_set_mempolicy
  if (v->n_nodes > 1) {
    unsigned long mask = 1ul << v->node;
    (void)syscall(SYS_set_mempolicy, 2 /* MPOL_BIND */, &mask, sizeof mask * 8);
  }
#endif
}

static void *writer(void *p) {
  Arg *a = (Arg *)p;
  VRank *v = a->v;
  bind_here(v);
  const double per = v->rate / WRITERS;      /* this thread's share */
  const __m128i pat = _mm_set1_epi32(0x41434754 + a->k);
  double t0 = now_s();
  long done = 0;
  while (!g_stop) {
    char *dst = v->ring + (size_t)((done * WRITERS + a->k) % RING) * CHUNK;
    for (size_t i = 0; i < CHUNK; i += 64) {
      _mm_stream_si128((__m128i *)(dst + i), pat);
      _mm_stream_si128((__m128i *)(dst + i + 16), pat);
      _mm_stream_si128((__m128i *)(dst + i + 32), pat);
      _mm_stream_si128((__m128i *)(dst + i + 48), pat);
    }
    _mm_sfence();
    done++;
    v->written[a->k] = done;
    const double due = t0 + (double)done * CHUNK / per;   /* throttle to the rate */
    double now = now_s();
    while (now < due && !g_stop) {
      struct timespec ts = {0, 50000};
      nanosleep(&ts, NULL);
      now = now_s();
    }
  }
  return NULL;
}

static void *reader(void *p) {
  Arg *a = (Arg *)p;
  VRank *v = a->v;
  bind_here(v);
  long done = 0;
  uint64_t sum = 0;
  while (!g_stop) {
    long avail = 0;
    for (int k = 0; k < WRITERS; k++) avail += v->written[k];
    if (g_read_only) avail = done * READERS + a->k + 1;   /* HOST_LOAD_READ_ONLY: the ring over and over, no writers */
    if (done * READERS + a->k >= avail) {   /* behind the writers, never ahead of them */
      struct timespec ts = {0, 50000};
      nanosleep(&ts, NULL);
      continue;
    }
    const uint64_t *src = (const uint64_t *)(v->ring + (size_t)((done * READERS + a->k) % RING) * CHUNK);
    for (size_t i = 0; i < CHUNK / 8; i += 8) sum += src[i] + src[i + 1] + src[i + 2] + src[i + 3] + src[i + 4] + src[i + 5] + src[i + 6] + src[i + 7];
    done++;
    v->read_[a->k] = done;
  }
  if (sum == 42) fprintf(stderr, "\n");
  return NULL;
}

static void *stdin_watch(void *p) {
  (void)p;
  char c;
  while (read(0, &c, 1) > 0) {
  }
  g_stop = 1;
  return NULL;
}

static void on_signal(int sig) {
  (void)sig;
  g_stop = 1;
}

int main(int argc, char **argv) {
  signal(SIGINT, on_signal);
  signal(SIGTERM, on_signal);
  if (argc < 3) {
    fprintf(stderr, "usage: host_load RANKS RATE_GBS [SECONDS] [REAL_RANK] [REAL_NODE]\n");
    return 2;
  }
  const int ranks = atoi(argv[1]);
  const double rate = atof(argv[2]) * 1e9, seconds = argc > 3 ? atof(argv[3]) : 0;
  const int skip = argc > 4 ? atoi(argv[4]) : 0;
  const int real_node = argc > 5 ? atoi(argv[5]) : 0;
  const int nodes = count_nodes();
  g_read_only = getenv("HOST_LOAD_READ_ONLY") != NULL;
  if (getenv("HOST_LOAD_WRITERS")) WRITERS = atoi(getenv("HOST_LOAD_WRITERS"));
  if (getenv("HOST_LOAD_READERS")) READERS = atoi(getenv("HOST_LOAD_READERS"));
  if (WRITERS < 1 || WRITERS > MAX_T || READERS < 1 || READERS > MAX_T) return 2;
  VRank *vs = calloc((size_t)ranks, sizeof *vs);
  static pthread_t th[64 * 2 * MAX_T];
  static Arg args[64 * 2 * MAX_T];
  int nt = 0;
  for (int r = 0; r < ranks && r < 64; r++) {
    if (r == skip) continue;
    VRank *v = &vs[r];
    v->rank = r;
    v->n_nodes = nodes;
    v->node = (real_node + (int)((long)((r - skip + ranks) % ranks) * nodes / ranks)) % nodes;
    v->rate = rate;
    v->have_cpus = node_cpus(v->node, &v->cpus) > 0;
    v->ring = mmap(NULL, (size_t)RING * CHUNK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (v->ring == MAP_FAILED) {
      perror("mmap");
      return 1;
    }
    if (g_read_only) memset(v->ring, 1, (size_t)RING * CHUNK);
    for (int k = 0; k < (g_read_only ? 0 : WRITERS); k++, nt++) {
      args[nt].v = v;
      args[nt].k = k;
      pthread_create(&th[nt], NULL, writer, &args[nt]);   /* (first touch happens in the bound writer threads) */
    }
    for (int k = 0; k < (getenv("HOST_LOAD_NO_READERS") ? 0 : READERS); k++, nt++) {
      args[nt].v = v;
      args[nt].k = k;
      pthread_create(&th[nt], NULL, reader, &args[nt]);
    }
  }
  pthread_t w;
  if (seconds <= 0) pthread_create(&w, NULL, stdin_watch, NULL);   /* no time limit: until the caller closes our stdin */
  printf("host_load: %d virtual ranks x %.1f GB/s in + read back, %d NUMA node(s), %d threads; running\n", ranks - 1, rate / 1e9, nodes, nt);
  fflush(stdout);
  const double t0 = now_s();
  while (!g_stop && (seconds <= 0 || now_s() - t0 < seconds)) usleep(20000);
  g_stop = 1;
  const double dt = now_s() - t0;
  for (int i = 0; i < nt; i++) pthread_join(th[i], NULL);
  double tw = 0, tr = 0;
  for (int r = 0; r < ranks && r < 64; r++) {
    if (r == skip) continue;
    long wsum = 0, rsum = 0;
    for (int k = 0; k < WRITERS; k++) wsum += vs[r].written[k];
    for (int k = 0; k < READERS; k++) rsum += vs[r].read_[k];
    printf("  virtual rank %d (node %d): wrote %.1f GB/s, read back %.1f GB/s\n", r, vs[r].node, wsum * (double)CHUNK / dt / 1e9,
           rsum * (double)CHUNK / dt / 1e9);
    tw += wsum * (double)CHUNK / dt / 1e9;
    tr += rsum * (double)CHUNK / dt / 1e9;
  }
  printf("host_load: %.1f s, total %.1f GB/s written + %.1f GB/s read\n", dt, tw, tr);
  return 0;
}
