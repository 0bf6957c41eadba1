/* oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.  See ref_shim.h.
 *
 * Maps each rand() call site (by __LINE__ of /root/reference/src/pbsim.cpp,
 * v3.0.5) to a (stream, sub-block, slot) of the keyed Philox contract.
 * PBSHIM_MODE=glibc forwards to libc rand() (proves the shim is transparent);
 * PBSHIM_MODE=philox (default) serves the keyed stream; PBSHIM_CENSUS=<file>
 * additionally dumps per-call-site draw counts at exit.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "philox4x32.h"

struct site_t { int line; int kind; int sub; int slot; };
enum { K_HDR = 1, K_WALK = 2, K_SERIAL = 3 };

/* kind, sub, slot per call site.  HDR sites are evaluated before sim.res_num++
 * (pbsim.cpp:3813, 2194, 4508, 2829...), so their read index is res_num+1. */
static const site_t SITES[] = {
  /* simulate_by_qshmm (WGS)              pbsim.cpp:2174-2270 */
  {2174,K_HDR,0,0},{2183,K_HDR,0,1},{2189,K_HDR,0,2},
  {2216,K_WALK,0,0},{2219,K_WALK,0,0},{2222,K_WALK,0,1},{2225,K_WALK,0,1},
  {2232,K_WALK,0,2},{2235,K_WALK,0,3},{2245,K_WALK,1,0},{2252,K_WALK,0,3},{2270,K_WALK,2,0},
  /* simulate_by_qshmm_trans              pbsim.cpp:2809-2906 */
  {2809,K_HDR,0,0},{2812,K_HDR,0,1},{2816,K_HDR,0,2},
  {2852,K_WALK,0,0},{2855,K_WALK,0,0},{2858,K_WALK,0,1},{2861,K_WALK,0,1},
  {2868,K_WALK,0,2},{2871,K_WALK,0,3},{2881,K_WALK,1,0},{2888,K_WALK,0,3},{2906,K_WALK,2,0},
  /* simulate_by_qshmm_templ              pbsim.cpp:3361-3430 */
  {3361,K_HDR,0,1},
  {3376,K_WALK,0,0},{3379,K_WALK,0,0},{3382,K_WALK,0,1},{3385,K_WALK,0,1},
  {3392,K_WALK,0,2},{3395,K_WALK,0,3},{3405,K_WALK,1,0},{3412,K_WALK,0,3},{3430,K_WALK,2,0},
  /* simulate_by_errhmm (WGS)             pbsim.cpp:3793-3958 */
  {3793,K_HDR,0,0},{3802,K_HDR,0,1},{3808,K_HDR,0,2},
  {3854,K_WALK,0,0},{3857,K_WALK,0,0},{3861,K_WALK,0,1},{3866,K_WALK,0,2},{3868,K_WALK,0,2},
  {3874,K_WALK,0,0},{3877,K_WALK,0,0},{3881,K_WALK,0,1},{3886,K_WALK,0,2},{3888,K_WALK,0,2},
  {3893,K_WALK,1,0},{3895,K_WALK,1,1},
  {3902,K_WALK,0,0},{3905,K_WALK,0,0},{3909,K_WALK,0,1},{3914,K_WALK,0,2},{3916,K_WALK,0,2},
  {3921,K_WALK,1,0},
  {3938,K_WALK,0,3},{3948,K_WALK,1,2},{3958,K_WALK,0,3},
  /* simulate_by_errhmm_trans             pbsim.cpp:4488-4653 */
  {4488,K_HDR,0,0},{4491,K_HDR,0,1},{4495,K_HDR,0,2},
  {4549,K_WALK,0,0},{4552,K_WALK,0,0},{4556,K_WALK,0,1},{4561,K_WALK,0,2},{4563,K_WALK,0,2},
  {4569,K_WALK,0,0},{4572,K_WALK,0,0},{4576,K_WALK,0,1},{4581,K_WALK,0,2},{4583,K_WALK,0,2},
  {4588,K_WALK,1,0},{4590,K_WALK,1,1},
  {4597,K_WALK,0,0},{4600,K_WALK,0,0},{4604,K_WALK,0,1},{4609,K_WALK,0,2},{4611,K_WALK,0,2},
  {4616,K_WALK,1,0},
  {4633,K_WALK,0,3},{4643,K_WALK,1,2},{4653,K_WALK,0,3},
  /* simulate_by_errhmm_templ             pbsim.cpp:5092-5228 */
  {5092,K_HDR,0,1},
  {5124,K_WALK,0,0},{5127,K_WALK,0,0},{5131,K_WALK,0,1},{5136,K_WALK,0,2},{5138,K_WALK,0,2},
  {5144,K_WALK,0,0},{5147,K_WALK,0,0},{5151,K_WALK,0,1},{5156,K_WALK,0,2},{5158,K_WALK,0,2},
  {5163,K_WALK,1,0},{5165,K_WALK,1,1},
  {5172,K_WALK,0,0},{5175,K_WALK,0,0},{5179,K_WALK,0,1},{5184,K_WALK,0,2},{5186,K_WALK,0,2},
  {5191,K_WALK,1,0},
  {5208,K_WALK,0,3},{5218,K_WALK,1,2},{5228,K_WALK,0,3},
  /* simulate_by_sample                   pbsim.cpp:1732-1820
   * 1732 is the per-round `sample_value`: header slot 3 of the round's first read (res_num+1);
   * 1758 the read's offset (header slot 2); the walk uses the QSHMM slots (error class z, nucleotide w,
   * sub-block 1 for a non-ACGT substitution, sub-block 2 for the deletion test of a column) */
  {1732,K_HDR,0,3},{1758,K_HDR,0,2},{1782,K_WALK,0,2},{1785,K_WALK,0,3},
  {1795,K_WALK,1,0},{1802,K_WALK,0,3},{1820,K_WALK,2,0},
};
#define NSITES ((int)(sizeof(SITES)/sizeof(SITES[0])))

static int g_init = 0, g_glibc = 0;
static uint32_t g_seed = 0;
static const site_t *g_by_line[6000];
static unsigned long long g_count[6000];
static const char *g_census = 0;

static void dump_census(void) {
  if (!g_census) return;
  FILE *fp = fopen(g_census, "w");
  if (!fp) return;
  for (int i = 0; i < 6000; i++)
    if (g_count[i]) fprintf(fp, "%d\t%llu\n", i, g_count[i]);
  fclose(fp);
}

static void shim_init(void) {
  g_init = 1;
  const char *m = getenv("PBSHIM_MODE");
  g_glibc = (m && strcmp(m, "glibc") == 0);
  for (int i = 0; i < NSITES; i++) g_by_line[SITES[i].line] = &SITES[i];
  g_census = getenv("PBSHIM_CENSUS");
  if (g_census) atexit(dump_census);
}

/* The reference seeds with srand(sim.seed) (pbsim.cpp:543); the shim reads the
 * same value from the `--seed` argument via PBSHIM_SEED (set by the runner),
 * because srand() is not a macro-able expression we want to touch. */
extern "C" int pbshim_draw(int line, long unit, long read, long pass, long event) {
  if (!g_init) {
    shim_init();
    const char *s = getenv("PBSHIM_SEED");
    if (!g_glibc && !s) { fprintf(stderr, "pbshim: PBSHIM_SEED is not set\n"); exit(97); }
    g_seed = s ? (uint32_t)(unsigned int)atoi(s) : 0;
  }
  if (line < 0 || line >= 6000 || !g_by_line[line]) {
    fprintf(stderr, "pbshim: unknown rand() call site at line %d\n", line);
    exit(98);
  }
  g_count[line]++;
  const site_t *s = g_by_line[line];
  if (g_glibc || s->kind == K_SERIAL) return (rand)();
  if (s->kind == K_HDR)
    return (int)orc_keyed_draw(g_seed, ORC_STREAM_HDR, (uint32_t)unit, (uint32_t)(read + 1),
                               0, 0, 0, (uint32_t)s->slot);
  if (s->sub == 2) /* the deletion test of a column: block of event column >> 2, word column & 3 */
    return (int)orc_keyed_draw(g_seed, ORC_STREAM_WALK, (uint32_t)unit, (uint32_t)read,
                               (uint32_t)pass, (uint32_t)event >> 2, 2u, (uint32_t)event & 3u);
  return (int)orc_keyed_draw(g_seed, ORC_STREAM_WALK, (uint32_t)unit, (uint32_t)read,
                             (uint32_t)pass, (uint32_t)event, (uint32_t)s->sub, (uint32_t)s->slot);
}
