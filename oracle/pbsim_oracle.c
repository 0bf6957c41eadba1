/* oracle/pbsim_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the pbsim3 hot path named by BASELINE.json:
 * "draw read header -> walk the FIC-HMM base by base -> emit read/qual/MAF",
 * for strategies wgs / trans / templ and methods errhmm / qshmm.  It exists
 * to CHECK the HIP product (tests/, __graft_entry__.smoke(), bench.py's
 * cpu_baseline leg).  Nothing in pbsim3_amd/ may include, link or execute it.
 *
 * Every function cites the reference lines it restates (paths relative to
 * /root/reference, `pbsim.cpp` = src/pbsim.cpp, v3.0.5).
 *
 * Pinning (see oracle/README.md, tests/test_oracle_vs_reference.py):
 *   --rng glibc   : byte-identical to the unmodified reference binary
 *                   (oracle/_ref/pbsim_ref) at a fixed --seed;
 *   --rng philox  : byte-identical to the reference source compiled with
 *                   oracle/ref_shim.h (oracle/_ref/pbsim_ref_philox), i.e. the
 *                   reference's own control flow fed with the keyed Philox
 *                   stream of DESIGN.md "RNG contract".
 * Golden outputs of both are committed under tests/golden/.
 *
 * Output files are the uncompressed text the reference pipes into gzip /
 * samtools: <prefix>_NNNN.{ref,fq,maf,sam} (wgs), <prefix>.{fq,maf,sam}.
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <getopt.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "philox4x32.h"

#define BUF_SIZE 10240          /* pbsim.cpp:20 */
#define REF_ID_LEN_MAX 128      /* pbsim.cpp:21 */
#define REF_SEQ_NUM_MAX 9999    /* pbsim.cpp:23 */
#define REF_SEQ_LEN_MAX 1000000000L
#define REF_SEQ_LEN_MIN 100
#define FASTQ_LEN_MAX 1000000
#define ACC_MAX 100             /* ACCURACY_MAX pbsim.cpp:42 */
#define STATE_MAX 50            /* pbsim.cpp:43 */
#define TR_RANK_MAX 1000

enum { ST_WGS = 1, ST_TRANS = 2, ST_TEMPL = 3 };
enum { ME_QS = 1, ME_ERR = 2, ME_SAM = 3, ME_SAM_REUSE = 4, ME_SAM_STORE = 5 }; /* pbsim.cpp:37-41 */
enum { RNG_GLIBC = 0, RNG_PHILOX = 1 };

/* ---------------------------------------------------------------- state -- */
static struct {
  int set_flg[30];
  int strategy, method, rng;
  unsigned int seed;
  double depth, accuracy_mean;
  long len_min, len_max;
  double len_mean, len_sd;
  long sub_ratio, ins_ratio, del_ratio;
  double sub_rate, ins_rate, del_rate;
  const char *prefix, *id_prefix, *genome_file, *transcript_file, *templ_file, *model_file;
  const char *sample_file, *profile_id;
  double accuracy_min, accuracy_max; /* sampling filter, pbsim.cpp:1622-1634 */
  int pass_num;
  double hp_del_bias;
  /* results (pbsim.cpp:63-70) */
  long res_num;
  long long res_len_total;
  long res_len_min, res_len_max;
  long res_sub_num, res_ins_num, res_del_num;
  double res_len_mean, res_len_sd, res_accuracy_mean, res_accuracy_sd;
  long res_pass_num;
} sim;

static double qc_prob[94];                 /* pbsim.cpp:546-549 */
static double uni_ep[ACC_MAX + 1][94];     /* pbsim.cpp:558-578 */
static long sub_thre[94], ins_thre[94], del_thre[94]; /* pbsim.cpp:5474-5479 */

/* models */
static double e_ip[ACC_MAX + 1][STATE_MAX + 1], e_ep[ACC_MAX + 1][STATE_MAX + 1][4],
    e_tp[ACC_MAX + 1][STATE_MAX + 1][STATE_MAX + 1];
static int e_exist[ACC_MAX + 1], e_state_max[ACC_MAX + 1], e_acc_min, e_acc_max;
/* struct qshmm_t (pbsim.cpp:160-166) lays ip[101][51], ep[101][51][94], tp[101][51][51] out back to back, and set_qshmm
 * indexes them with whatever state / column numbers the file holds.  QSHMM-ONT-HQ.model has classes with up to 56 states
 * (SURVEY Q7): in the compiled reference a state above STATE_MAX simply lands STATE_MAX+1 slots further on in the same
 * block (the next state's row, the next class, the next member).  That is deterministic, and it is what the goldens
 * `wgs_qshmm_onthq*` were produced with, so the block is restated as ONE flat array with the reference's strides and no
 * per-dimension check; only a write past the end of tp[] (it would hit exist_hmm[]) is refused. */
#define Q_IP_N ((ACC_MAX + 1) * (STATE_MAX + 1))
#define Q_EP_N ((ACC_MAX + 1) * (STATE_MAX + 1) * 94)
#define Q_TP_N ((ACC_MAX + 1) * (STATE_MAX + 1) * (STATE_MAX + 1))
static double q_block[Q_IP_N + Q_EP_N + Q_TP_N];
#define Q_IP_AT(a, s) ((long)(a) * (STATE_MAX + 1) + (s))
#define Q_EP_AT(a, s, k) (Q_IP_N + ((long)(a) * (STATE_MAX + 1) + (s)) * 94 + (k))
#define Q_TP_AT(a, s, k) (Q_IP_N + Q_EP_N + ((long)(a) * (STATE_MAX + 1) + (s)) * (STATE_MAX + 1) + (k))
#define q_ip(a, s) q_block[Q_IP_AT(a, s)]
#define q_ep(a, s, k) q_block[Q_EP_AT(a, s, k)]
#define q_tp(a, s, k) q_block[Q_TP_AT(a, s, k)]
static int q_exist[ACC_MAX + 1];

/* lookup tables (expanded, as the reference builds them) */
static long prob2len[100001], prob2accuracy[100001];
static long len_rand_value, accuracy_rand_value;
static int accuracy_min, accuracy_max;
static unsigned char init2state[ACC_MAX + 1][1001];
static unsigned char emis2err[ACC_MAX + 1][STATE_MAX + 1][1001];
static long emis2del[ACC_MAX + 1][STATE_MAX + 1];
static unsigned char tran2state[ACC_MAX + 1][STATE_MAX + 1][1001];
static long rv_init[ACC_MAX + 1], rv_emis[ACC_MAX + 1][STATE_MAX + 1], rv_tran[ACC_MAX + 1][STATE_MAX + 1];
static unsigned char freq2qc[ACC_MAX + 1][1001];
static long rv_freq[ACC_MAX + 1];
static long prob2ssp[TR_RANK_MAX + 1][1001], ssp_rand_value[TR_RANK_MAX + 1];

/* reference sequence of the current unit */
static char *ref_seq;
static short *ref_hp;
static long ref_len, ref_num, ref_num_seq;
static char ref_id[REF_ID_LEN_MAX + 1];
/* struct genome_t / transcript_t tail (pbsim.cpp:101-102, 115-116):
 * `long hpfreq[11]; double hp_del_bias[11];` -- hpfreq[11] IS hp_del_bias[0]
 * (Q1/Q15).  Modelled as one 12-slot array each with the aliasing explicit. */
static long hpfreq[12];
static double hp_bias[12]; /* [0] aliased from hpfreq[11]; [11] pinned 0.0 */

static long freq_len[FASTQ_LEN_MAX + 1];
static long freq_accuracy[100001];
static double accuracy_total;

/* scratch of one read (pbsim.cpp:5488-5531: 2*len_max+1 each) */
static char *m_seq, *m_read, *m_maf, *m_mafref, *m_qc, *m_newqc;
static short *m_hp;

static FILE *fp_fq, *fp_sam, *fp_maf;

/* ------------------------------------------------------------------ RNG -- */
static uint32_t g_unit, g_read, g_pass, g_event;
static unsigned long long g_draws;

static inline long R_hdr(int slot) {
  g_draws++;
  if (sim.rng == RNG_GLIBC) return rand();
  return (long)orc_keyed_draw(sim.seed, ORC_STREAM_HDR, g_unit, g_read, 0, 0, 0, (uint32_t)slot);
}
static inline long R_walk(int sub, int slot) {
  g_draws++;
  if (sim.rng == RNG_GLIBC) return rand();
  /* sub-block 2 (the deletion test of a column, one draw): the block of event column >> 2 serves four columns, word
   * column & 3 (contract of round 2) */
  if (sub == 2)
    return (long)orc_keyed_draw(sim.seed, ORC_STREAM_WALK, g_unit, g_read, g_pass, g_event >> 2, 2u, g_event & 3u);
  return (long)orc_keyed_draw(sim.seed, ORC_STREAM_WALK, g_unit, g_read, g_pass, g_event,
                              (uint32_t)sub, (uint32_t)slot);
}

static void die(const char *msg) {
  fprintf(stderr, "ERROR: %s\n", msg);
  exit(255);
}

static int trim(char *line) { /* pbsim.cpp:882-890 */
  long end_pos = (long)strlen(line) - 1;
  if (end_pos >= 0 && line[end_pos] == '\n') {
    line[end_pos] = '\0';
    return 1;
  }
  return 0;
}

static int count_digit(long num) { /* pbsim.cpp:5823-5835 */
  int digit = 1;
  int quotient = (int)(num / 10);
  while (quotient != 0) {
    digit++;
    quotient = (int)(quotient / 10);
  }
  return digit;
}

static void revcomp(char *str) { /* pbsim.cpp:5841-5864 */
  long len = (long)strlen(str);
  for (long i = 0; i < len / 2; i++) {
    char c = str[i];
    str[i] = str[len - i - 1];
    str[len - i - 1] = c;
  }
  for (long i = 0; i < len; i++) {
    if (str[i] == 'A') str[i] = 'T';
    else if (str[i] == 'T') str[i] = 'A';
    else if (str[i] == 'G') str[i] = 'C';
    else if (str[i] == 'C') str[i] = 'G';
  }
}

static void revshort(short *s, long len) { /* pbsim.cpp:5870-5879 */
  for (long i = 0; i < len / 2; i++) {
    short t = s[i];
    s[i] = s[len - i - 1];
    s[len - i - 1] = t;
  }
}

/* --------------------------------------------------------- model parsers -- */
static void set_errhmm(void) { /* pbsim.cpp:5640-5714 */
  FILE *fp = fopen(sim.model_file, "r");
  char line[BUF_SIZE], *tp;
  if (!fp) { fprintf(stderr, "ERROR: Cannot open file: %s\n", sim.model_file); exit(255); }
  e_acc_min = 100;
  e_acc_max = 0;
  while (fgets(line, BUF_SIZE, fp) != NULL) {
    trim(line);
    tp = strtok(line, " ");
    int accuracy = atoi(tp);
    e_exist[accuracy] = 1;
    if (e_acc_min > accuracy) e_acc_min = accuracy;
    if (e_acc_max < accuracy) e_acc_max = accuracy;
    tp = strtok(NULL, " ");
    if (strcmp(tp, "IP") == 0) {
      int state = atoi(strtok(NULL, " "));
      e_ip[accuracy][state] = atof(strtok(NULL, " "));
      e_state_max[accuracy] = state;
    } else if (strcmp(tp, "EP") == 0) {
      int state = atoi(strtok(NULL, " "));
      int num = 0;
      while ((tp = strtok(NULL, " ")) != NULL) e_ep[accuracy][state][num++] = atof(tp);
    } else if (strcmp(tp, "TP") == 0) {
      int state = atoi(strtok(NULL, " "));
      int num = 0;
      while ((tp = strtok(NULL, " ")) != NULL) e_tp[accuracy][state][++num] = atof(tp);
    }
  }
  fclose(fp);
}

static void q_store(long at, double v) {
  if (at < 0 || at >= (long)(Q_IP_N + Q_EP_N + Q_TP_N))
    die("oracle: QSHMM model writes past qshmm.tp[] (the reference would overwrite exist_hmm[])");
  q_block[at] = v;
}

static void set_qshmm(void) { /* pbsim.cpp:5570-5634 */
  FILE *fp = fopen(sim.model_file, "r");
  char line[BUF_SIZE], *tp;
  if (!fp) { fprintf(stderr, "ERROR: Cannot open file: %s\n", sim.model_file); exit(255); }
  while (fgets(line, BUF_SIZE, fp) != NULL) {
    trim(line);
    tp = strtok(line, " ");
    int accuracy = atoi(tp);
    if (accuracy < 0 || accuracy > ACC_MAX) die("oracle: QSHMM accuracy class outside 0-100");
    q_exist[accuracy] = 1;
    tp = strtok(NULL, " ");
    if (strcmp(tp, "IP") == 0) {
      int state = atoi(strtok(NULL, " "));
      q_store(Q_IP_AT(accuracy, state), atof(strtok(NULL, " ")));
    } else if (strcmp(tp, "EP") == 0) {
      int state = atoi(strtok(NULL, " "));
      int num = 0;
      while ((tp = strtok(NULL, " ")) != NULL) q_store(Q_EP_AT(accuracy, state, num++), atof(tp));
    } else if (strcmp(tp, "TP") == 0) {
      int state = atoi(strtok(NULL, " "));
      int num = 0;
      while ((tp = strtok(NULL, " ")) != NULL) {
        num++;
        q_store(Q_TP_AT(accuracy, state, num), atof(tp));
      }
    }
  }
  fclose(fp);
}

/* ------------------------------------------------------ table construction -- */
static void init_common_tables(void) {
  /* pbsim.cpp:546-578 */
  for (int i = 0; i <= 93; i++) qc_prob[i] = pow(10, (double)i / -10);
  for (int i = 0; i <= ACC_MAX; i++) {
    for (int j = 0; j <= 93; j++) uni_ep[i][j] = 0;
    if (i == ACC_MAX) { uni_ep[i][93] = 1.0; continue; }
    double prob = 1.0 - i / 100.0;
    for (int j = 0; j <= 93; j++) {
      if (prob == qc_prob[j]) { uni_ep[i][j] = 1.0; break; }
      else if (prob > qc_prob[j]) {
        double rate = (prob - qc_prob[j]) / (qc_prob[j - 1] - qc_prob[j]);
        uni_ep[i][j - 1] = rate;
        uni_ep[i][j] = 1 - rate;
        break;
      }
    }
  }
  /* pbsim.cpp:5474-5479 */
  for (int i = 0; i <= 93; i++) {
    sub_thre[i] = (int)((qc_prob[i] * sim.sub_rate) * 1000000 + 0.5);
    ins_thre[i] = (int)((qc_prob[i] * (sim.sub_rate + sim.ins_rate)) * 1000000 + 0.5);
    del_thre[i] = (int)((qc_prob[i] * sim.del_rate) / (1 + qc_prob[i] * sim.del_rate) * 1000000 + 0.5);
  }
}

static void build_len_table(void) { /* pbsim.cpp:3634-3662 (=1992-2020, 4167-4193) */
  double variance = pow(sim.len_sd, 2);
  double kappa = pow(sim.len_mean, 2) / variance;
  double theta = variance / sim.len_mean;
  double gamma = tgamma(kappa);
  long start_wk, end_wk = 0, i, j;
  if (sim.len_sd == 0.0) {
    prob2len[1] = (int)(sim.len_mean + 0.5);
    len_rand_value = 1;
  } else {
    double len_prob_total = 0.0;
    start_wk = 1;
    for (i = sim.len_min; i <= sim.len_max; i++) {
      len_prob_total += pow(i, kappa - 1) * exp(-1 * i / theta) / pow(theta, kappa) / gamma;
      end_wk = (int)(len_prob_total * 100000 + 0.5);
      if (end_wk > 100000) end_wk = 100000;
      for (j = start_wk; j <= end_wk; j++) prob2len[j] = i;
      if (end_wk >= 100000) break;
      start_wk = end_wk + 1;
    }
    len_rand_value = end_wk;
  }
  /* pbsim.cpp:3664-3669: the WGS variants only check when pass_num==1;
   * trans always (4195).  A zero modulus would trap either way. */
  if (len_rand_value < 1) die("length parameters are not appropriate.");
}

static void build_acc_table(void) { /* pbsim.cpp:3672-3706 */
  double mean = sim.accuracy_mean * 100;
  long start_wk, end_wk = 0, i, j;
  accuracy_max = (int)floor(mean * 1.05);
  accuracy_min = (int)floor(mean * 0.75);
  if (accuracy_max > 100) accuracy_max = 100;
  double freq_total = 0.0;
  for (i = accuracy_min; i <= accuracy_max; i++) freq_total += exp(0.22 * i);
  start_wk = 1;
  double accuracy_prob_total = 0.0;
  for (i = accuracy_min; i <= accuracy_max; i++) {
    accuracy_prob_total += exp(0.22 * i) / freq_total;
    end_wk = (int)(accuracy_prob_total * 100000 + 0.5);
    if (end_wk > 100000) end_wk = 100000;
    for (j = start_wk; j <= end_wk; j++) prob2accuracy[j] = i;
    if (end_wk >= 100000) break;
    start_wk = end_wk + 1;
  }
  accuracy_rand_value = end_wk;
  if (accuracy_rand_value < 1) die("accuracy parameters are not appropriate.");
}

/* pbsim.cpp:3709-3789 (wgs; emis skip `<= 0`), 4264-4340 / 4889-4965 (trans,
 * templ; emis skip `== 0`).  `end_wk` deliberately carries over
 * between rows exactly as in the reference. */
static void build_errhmm_tables(int emis_skip_le) {
  long i, j, k, l, start_wk, end_wk = 0;
  for (i = accuracy_min; i <= accuracy_max; i++) {
    if (e_exist[i] == 0) continue;
    start_wk = 1;
    double tot = 0.0;
    for (j = 1; j <= e_state_max[i]; j++) {
      if (e_ip[i][j] == 0) continue;
      tot += e_ip[i][j];
      end_wk = (int)(tot * 1000 + 0.5);
      if (end_wk > 1000) end_wk = 1000;
      for (k = start_wk; k <= end_wk; k++) init2state[i][k] = (unsigned char)j;
      if (end_wk >= 1000) break;
      start_wk = end_wk + 1;
    }
    rv_init[i] = end_wk;
    for (j = 1; j <= e_state_max[i]; j++) {
      start_wk = 1;
      tot = 0.0;
      emis2del[i][j] = (int)(e_ep[i][j][3] * 1000 + 0.5);
      for (k = 0; k <= 2; k++) {
        if (emis_skip_le ? (e_ep[i][j][k] <= 0) : (e_ep[i][j][k] == 0)) continue;
        tot += e_ep[i][j][k];
        end_wk = (int)(tot * 1000 + 0.5);
        if (end_wk > 1000) end_wk = 1000;
        for (l = start_wk; l <= end_wk; l++) emis2err[i][j][l] = (unsigned char)k;
        if (end_wk >= 1000) break;
        start_wk = end_wk + 1;
      }
      rv_emis[i][j] = end_wk;
    }
    for (j = 1; j <= e_state_max[i]; j++) {
      start_wk = 1;
      tot = 0.0;
      for (k = 1; k <= STATE_MAX; k++) {
        if (e_tp[i][j][k] == 0) continue;
        tot += e_tp[i][j][k];
        end_wk = (int)(tot * 1000 + 0.5);
        if (end_wk > 1000) end_wk = 1000;
        for (l = start_wk; l <= end_wk; l++) tran2state[i][j][l] = (unsigned char)k;
        if (end_wk >= 1000) break;
        start_wk = end_wk + 1;
      }
      rv_tran[i][j] = end_wk;
    }
  }
}

/* pbsim.cpp:2066-2170: resolution 100 for init/emis/tran, 1000 for freq2qc */
static void build_qshmm_tables(void) {
  long i, j, k, l, start_wk, end_wk = 0;
  for (i = accuracy_min; i <= accuracy_max; i++) {
    if (q_exist[i] == 1) {
      start_wk = 1;
      double tot = 0.0;
      for (j = 1; j <= STATE_MAX; j++) {
        if (q_ip(i, j) == 0) continue;
        tot += q_ip(i, j);
        end_wk = (int)(tot * 100 + 0.5);
        if (end_wk > 100) end_wk = 100;
        for (k = start_wk; k <= end_wk; k++) init2state[i][k] = (unsigned char)j;
        if (end_wk >= 100) break;
        start_wk = end_wk + 1;
      }
      rv_init[i] = end_wk;
      for (j = 1; j <= STATE_MAX; j++) {
        start_wk = 1;
        tot = 0.0;
        for (k = 0; k <= 93; k++) {
          if (q_ep(i, j, k) == 0) continue;
          tot += q_ep(i, j, k);
          end_wk = (int)(tot * 100 + 0.5);
          if (end_wk > 100) end_wk = 100;
          for (l = start_wk; l <= end_wk; l++) emis2err[i][j][l] = (unsigned char)k; /* emis2qc */
          if (end_wk >= 100) break;
          start_wk = end_wk + 1;
        }
        rv_emis[i][j] = end_wk;
      }
      for (j = 1; j <= STATE_MAX; j++) {
        start_wk = 1;
        tot = 0.0;
        for (k = 1; k <= STATE_MAX; k++) {
          if (q_tp(i, j, k) == 0) continue;
          tot += q_tp(i, j, k);
          end_wk = (int)(tot * 100 + 0.5);
          if (end_wk > 100) end_wk = 100;
          for (l = start_wk; l <= end_wk; l++) tran2state[i][j][l] = (unsigned char)k;
          if (end_wk >= 100) break;
          start_wk = end_wk + 1;
        }
        rv_tran[i][j] = end_wk;
      }
    } else {
      start_wk = 1;
      double tot = 0.0;
      for (j = 0; j <= 93; j++) {
        if (uni_ep[i][j] == 0) continue;
        tot += uni_ep[i][j];
        end_wk = (int)(tot * 1000 + 0.5);
        if (end_wk > 1000) end_wk = 1000;
        for (k = start_wk; k <= end_wk; k++) freq2qc[i][k] = (unsigned char)j;
        if (end_wk >= 1000) break;
        start_wk = end_wk + 1;
      }
      rv_freq[i] = end_wk;
    }
  }
}

static void build_ssp_table(int rank_max) { /* pbsim.cpp:4200-4224 */
  for (long i = 1; i <= rank_max; i++) {
    double sum = 0, value = (double)1 / i, tot = 0.0;
    long start_wk = 1, end_wk = 0;
    for (long j = 1; j <= 21; j++) sum += value / pow(j, (1 + value));
    for (long j = 1; j <= 21; j++) {
      tot += (value / pow(j, (1 + value))) / sum;
      end_wk = (int)(tot * 1000 + 0.5);
      if (end_wk > 1000) end_wk = 1000;
      for (long k = start_wk; k <= end_wk; k++) prob2ssp[i][k] = (j - 1) * 5;
      if (end_wk >= 1000) break;
      start_wk = end_wk + 1;
    }
    ssp_rand_value[i] = end_wk;
  }
}

/* homopolymer length per base with the 10/11 oscillation (Q1): pbsim.cpp:1039-1065
 * (count=1 adds to hpfreq as get_genome_seq does; weight>0 is the trans bias
 * pre-pass 4389-4409 which adds read_num*(run length) and assigns nothing). */
static void compute_hp(const char *seq, long len, short *hp, int count, long weight) {
  long nstart = 0, nend = 0;
  short nnum = 1;
  for (long i = 1; i <= len; i++) {
    if ((i < len) && (seq[i - 1] == seq[i])) {
      nend = i;
      nnum++;
      if (nnum > 11) nnum = 10;
    } else {
      short v = (seq[i - 1] == 'N') ? 1 : nnum;
      if (weight > 0) {
        hpfreq[v] += weight * (nend - nstart + 1);
      } else {
        for (long j = nstart; j <= nend; j++) {
          hp[j] = v;
          if (count) hpfreq[v]++;
        }
      }
      nstart = i;
      nend = nstart;
      nnum = 1;
    }
  }
}

/* hpfreq[11] aliases hp_del_bias[0] (struct layout, pbsim.cpp:101-102) */
static void sync_bias_alias(void) {
  memcpy(&hp_bias[0], &hpfreq[11], sizeof(double));
  hp_bias[11] = 0.0; /* out-of-struct read observed as 0.0 */
}

static void set_bias_default(void) { /* pbsim.cpp:673-676 */
  for (int i = 1; i <= 10; i++) hp_bias[i] = 1;
}

static void normalise_bias(void) { /* pbsim.cpp:686-696 (sum1 is a long: truncates each add) */
  long sum1 = 0, sum2 = 0;
  for (int i = 1; i <= 10; i++) {
    hp_bias[i] = 1 + (sim.hp_del_bias - 1) / 9 * (i - 1);
    sum1 += hpfreq[i] * hp_bias[i];
    sum2 += hpfreq[i];
  }
  double rate = (double)sum2 / sum1;
  for (int i = 1; i <= 10; i++) hp_bias[i] *= rate;
}

/* ------------------------------------------------------------- emission -- */
static char sub_nt(char nt, long idx, int *need_n) {
  *need_n = 0;
  if (nt == 'A') return "TGC"[idx];
  if (nt == 'T') return "AGC"[idx];
  if (nt == 'G') return "ATC"[idx];
  if (nt == 'C') return "ATG"[idx];
  *need_n = 1;
  return 0;
}

/* FASTQ/SAM + MAF text of one pass: pbsim.cpp:4012-4078 (=2318-2383, 4702-4768) */
static void emit_record(long len, const char *qual, long pass, const char *ref_name, int ref_name_digits,
                        long seq_left, long seq_right, char strand, int wgs_id) {
  char id[192];
  long i;
  if (sim.pass_num == 1) {
    if (wgs_id) sprintf(id, "%s%ld_%ld", sim.id_prefix, ref_num, sim.res_num);
    else sprintf(id, "%s_%ld", sim.id_prefix, sim.res_num);
    fprintf(fp_fq, "@%s\n%s\n+%s\n%s\n", id, m_read, id, qual);
  } else {
    if (wgs_id) sprintf(id, "%s%ld/%ld/%ld", sim.id_prefix, ref_num, sim.res_num, pass);
    else sprintf(id, "%s/%ld/%ld", sim.id_prefix, sim.res_num, pass);
    fprintf(fp_sam, "%s\t4\t*\t0\t255\t*\t*\t0\t0\t%s\t%s", id, m_read, qual);
    fprintf(fp_sam, "\tcx:i:3\tip:B:C");
    for (i = 0; i < len; i++) fprintf(fp_sam, ",9");
    fprintf(fp_sam, "\tnp:i:1\tpw:B:C");
    for (i = 0; i < len; i++) fprintf(fp_sam, ",9");
    long qeval = (int)(len - 1);
    fprintf(fp_sam, "\tqs:i:0\tqe:i:%ld\trq:f:%f\tsn:B:f,10.0,10.0,10.0,10.0\tzm:i:%ld\tRG:Z:ffffffff\n",
            qeval, sim.accuracy_mean, sim.res_num);
  }
  int d1[4], d2[4], d[4];
  d1[0] = ref_name_digits;
  d2[0] = 1 + count_digit(sim.res_num);
  d1[1] = count_digit(seq_left - 1);
  d2[1] = 1;
  d1[2] = count_digit(seq_right - seq_left + 1);
  d2[2] = count_digit(len);
  d1[3] = count_digit(ref_len);
  d2[3] = count_digit(len);
  for (int k = 0; k < 4; k++) d[k] = d1[k] >= d2[k] ? d1[k] : d2[k];
  fprintf(fp_maf, "a\ns %s", ref_name);
  while (d1[0]++ < d[0]) fprintf(fp_maf, " ");
  while (d1[1]++ < d[1]) fprintf(fp_maf, " ");
  fprintf(fp_maf, " %ld", seq_left - 1);
  while (d1[2]++ < d[2]) fprintf(fp_maf, " ");
  fprintf(fp_maf, " %ld +", seq_right - seq_left + 1);
  while (d1[3]++ < d[3]) fprintf(fp_maf, " ");
  fprintf(fp_maf, " %ld %s\n", ref_len, m_mafref);
  fprintf(fp_maf, "s %s", id);
  while (d2[0]++ < d[0]) fprintf(fp_maf, " ");
  while (d2[1]++ < d[1]) fprintf(fp_maf, " ");
  fprintf(fp_maf, " %d", 0);
  while (d2[2]++ < d[2]) fprintf(fp_maf, " ");
  fprintf(fp_maf, " %ld %c", len, strand);
  while (d2[3]++ < d[3]) fprintf(fp_maf, " ");
  fprintf(fp_maf, " %ld %s\n\n", len, m_maf);
}

static void account(long len, double value) { /* pbsim.cpp:3986-4005 */
  sim.res_len_total += len;
  freq_len[len]++;
  if (len > sim.res_len_max) sim.res_len_max = len;
  if (len < sim.res_len_min) sim.res_len_min = len;
  accuracy_total += value;
  int acc_wk = (int)(value * 100000 + 0.5);
  freq_accuracy[acc_wk]++;
}

/* One ERRHMM pass over m_seq[0..L) -- pbsim.cpp:3836-3984 (=4531-4679, 5106-5254).
 * Returns the read length; fills m_read/m_maf/m_mafref. */
static long walk_errhmm(long L, int acc, int rate_mag, char strand) {
  long ref_offset = 0, read_offset = 0, maf_offset = 0, state = 0, index, err_num = 0;
  if (acc == 100) { /* Q8 */
    for (long i = 0; i < L; i++) m_read[i] = m_mafref[i] = m_maf[i] = m_seq[i];
    read_offset = maf_offset = L;
  } else {
    int m = e_exist[acc] ? acc : (acc < e_acc_min ? e_acc_min : e_acc_max);
    int below = !e_exist[acc] && acc < e_acc_min;
    int above = !e_exist[acc] && !below;
    while (ref_offset < L) {
      char nt = m_seq[ref_offset];
      g_event = (uint32_t)maf_offset;
      if (read_offset == 0) { /* Q2 */
        index = R_walk(0, 0) % rv_init[m] + 1;
        state = init2state[m][index];
      } else {
        index = R_walk(0, 0) % rv_tran[m][state] + 1;
        state = tran2state[m][state][index];
      }
      int hp = m_hp[ref_offset];
      index = R_walk(0, 1) % 1000 + 1;
      if (index <= emis2del[m][state] * hp_bias[hp]) {
        index = 3;
      } else if (rv_emis[m][state] == 0) {
        index = R_walk(0, 2) % 3;
      } else {
        index = R_walk(0, 2) % rv_emis[m][state] + 1;
        index = emis2err[m][state][index];
      }
      if (below && index == 0) { /* Q3: pbsim.cpp:3892-3899 */
        index = R_walk(1, 0) % 100 + 1;
        if (index <= rate_mag) index = R_walk(1, 1) % 3 + 1;
        else index = 0;
      } else if (above && index != 0) { /* pbsim.cpp:3920-3925 */
        long index2 = R_walk(1, 0) % 100 + 1;
        if (index2 <= rate_mag) index = 0;
      }
      if (index == 0) {
        m_read[read_offset] = nt;
        m_maf[maf_offset] = nt;
        m_mafref[maf_offset] = nt;
        ref_offset++;
        read_offset++;
      } else if (index == 1) {
        int need_n;
        err_num++;
        sim.res_sub_num++;
        char b = sub_nt(nt, R_walk(0, 3) % 3, &need_n);
        if (need_n) b = "ATGC"[R_walk(1, 2) % 4];
        m_read[read_offset] = b;
        m_maf[maf_offset] = b;
        m_mafref[maf_offset] = nt;
        ref_offset++;
        read_offset++;
      } else if (index == 2) {
        err_num++;
        sim.res_ins_num++;
        index = R_walk(0, 3) % 8;
        m_read[read_offset] = (index >= 4) ? nt : "ATGC"[index];
        m_maf[maf_offset] = m_read[read_offset];
        m_mafref[maf_offset] = '-';
        read_offset++;
      } else {
        err_num++;
        sim.res_del_num++;
        m_maf[maf_offset] = '-';
        m_mafref[maf_offset] = nt;
        ref_offset++;
      }
      maf_offset++;
    }
  }
  m_read[read_offset] = '\0';
  m_maf[maf_offset] = '\0';
  m_mafref[maf_offset] = '\0';
  if (strand == '-') {
    revcomp(m_maf);
    revcomp(m_mafref);
  }
  long len = (long)strlen(m_read);
  account(len, 1.0 - ((double)err_num / len));
  for (long i = 0; i < len; i++) m_newqc[i] = '!';
  m_newqc[len] = '\0';
  return len;
}

/* One QSHMM pass -- pbsim.cpp:2210-2316 (=2846-2952, 3370-3476). */
static long walk_qshmm(long L, int acc, char strand) {
  long ref_offset = 0, read_offset = 0, maf_offset = 0, state = 0, index, rand_value;
  while (ref_offset < L) {
    g_event = (uint32_t)maf_offset;
    if (q_exist[acc] == 1) {
      if (read_offset == 0) {
        index = R_walk(0, 0) % rv_init[acc] + 1;
        state = init2state[acc][index];
      } else {
        index = R_walk(0, 0) % rv_tran[acc][state] + 1;
        state = tran2state[acc][state][index];
      }
      index = R_walk(0, 1) % rv_emis[acc][state] + 1;
      index = emis2err[acc][state][index];
    } else {
      index = R_walk(0, 1) % rv_freq[acc] + 1;
      index = freq2qc[acc][index];
    }
    m_qc[read_offset] = (char)(index + 33);
    char nt = m_seq[ref_offset];
    int qv = (int)m_qc[read_offset] - 33;
    rand_value = R_walk(0, 2) % 1000000;
    if (rand_value < sub_thre[qv]) {
      int need_n;
      sim.res_sub_num++;
      char b = sub_nt(nt, R_walk(0, 3) % 3, &need_n);
      if (need_n) b = "ATGC"[R_walk(1, 0) % 4];
      m_read[read_offset] = b;
      m_mafref[maf_offset] = nt;
      ref_offset++;
    } else if (rand_value < ins_thre[qv]) {
      sim.res_ins_num++;
      index = R_walk(0, 3) % 8;
      m_read[read_offset] = (index >= 4) ? nt : "ATGC"[index];
      m_mafref[maf_offset] = '-';
    } else {
      m_read[read_offset] = nt;
      m_mafref[maf_offset] = nt;
      ref_offset++;
    }
    m_maf[maf_offset] = m_read[read_offset];
    maf_offset++;
    read_offset++;
    while (ref_offset < L) {
      /* Q15: hp = mut.hp[-1] when ref_offset==0 -> observed 0 -> hp_del_bias[0] */
      int hp = (ref_offset > 0) ? m_hp[ref_offset - 1] : 0;
      g_event = (uint32_t)maf_offset;
      rand_value = R_walk(2, 0) % 1000000;
      qv = (int)m_qc[read_offset - 1] - 33;
      if (rand_value < del_thre[qv] * hp_bias[hp]) {
        sim.res_del_num++;
        m_maf[maf_offset] = '-';
        m_mafref[maf_offset] = m_seq[ref_offset];
        maf_offset++;
        ref_offset++;
      } else {
        break;
      }
    }
  }
  m_qc[read_offset] = '\0';
  m_read[read_offset] = '\0';
  m_maf[maf_offset] = '\0';
  m_mafref[maf_offset] = '\0';
  if (strand == '-') {
    revcomp(m_maf);
    revcomp(m_mafref);
  }
  long len = (long)strlen(m_read);
  double prob = 0.0;
  for (long i = 0; i < len; i++) prob += qc_prob[(int)m_qc[i] - 33];
  account(len, 1.0 - (prob / len));
  return len;
}

static int errhmm_rate_mag(int acc, int prev) { /* pbsim.cpp:3829-3833 (value persists otherwise) */
  if (acc < e_acc_min) return (int)((double)(e_acc_min - acc) / e_acc_min * 100);
  if (acc > e_acc_max) return (int)((double)(acc - e_acc_max) / (100 - e_acc_max) * 100);
  return prev;
}

static void finish_stats(void) { /* pbsim.cpp:4082-4105 */
  sim.res_pass_num = sim.res_num * sim.pass_num;
  sim.res_len_mean = (double)sim.res_len_total / sim.res_pass_num;
  sim.res_accuracy_mean = accuracy_total / sim.res_pass_num;
  if (sim.res_pass_num == 1) {
    sim.res_len_sd = 0.0;
    sim.res_accuracy_sd = 0.0;
  } else {
    double variance = 0.0;
    for (long i = 0; i <= sim.len_max; i++)
      if (freq_len[i] > 0) variance += pow((sim.res_len_mean - i), 2) * freq_len[i];
    sim.res_len_sd = sqrt(variance / sim.res_pass_num);
    variance = 0.0;
    for (long i = 0; i <= 100000; i++)
      if (freq_accuracy[i] > 0) variance += pow((sim.res_accuracy_mean - i * 0.00001), 2) * freq_accuracy[i];
    sim.res_accuracy_sd = sqrt(variance / sim.res_pass_num);
  }
}

static void init_sim_res(void) { /* pbsim.cpp:1437-1445 + the per-call zeroing 3626-3631 */
  sim.res_num = 0;
  sim.res_len_total = 0;
  sim.res_sub_num = sim.res_ins_num = sim.res_del_num = 0;
  sim.res_len_min = LONG_MAX;
  sim.res_len_max = 0;
  accuracy_total = 0.0;
  for (long i = 0; i <= sim.len_max; i++) freq_len[i] = 0;
  for (long i = 0; i <= 100000; i++) freq_accuracy[i] = 0;
}

static void print_simulation_stats(void) { /* pbsim.cpp:5541-5564 */
  if (sim.strategy == ST_WGS) {
    double res_depth = (double)sim.res_len_total / ref_len / sim.pass_num;
    fprintf(stderr, ":::: Simulation stats (ref.%ld) ::::\n\n", ref_num);
    fprintf(stderr, "read num. : %ld\n", sim.res_num);
    fprintf(stderr, "depth : %lf\n", res_depth);
  } else {
    fprintf(stderr, ":::: Simulation stats ::::\n\n");
    fprintf(stderr, "read num. : %ld\n", sim.res_num);
  }
  fprintf(stderr, "read length mean (SD) : %f (%f)\n", sim.res_len_mean, sim.res_len_sd);
  fprintf(stderr, "read length min : %ld\n", sim.res_len_min);
  fprintf(stderr, "read length max : %ld\n", sim.res_len_max);
  fprintf(stderr, "read accuracy mean (SD) : %f (%f)\n", sim.res_accuracy_mean, sim.res_accuracy_sd);
  fprintf(stderr, "substitution rate. : %f\n", (double)sim.res_sub_num / sim.res_len_total);
  fprintf(stderr, "insertion rate. : %f\n", (double)sim.res_ins_num / sim.res_len_total);
  fprintf(stderr, "deletion rate. : %f\n", (double)sim.res_del_num / sim.res_len_total);
  fprintf(stderr, "\n");
}

static void print_sim_param(void) { /* pbsim.cpp:5397-5465 */
  fprintf(stderr, ":::: Simulation parameters :::\n\n");
  fprintf(stderr, "strategy : %s\n", sim.strategy == ST_WGS ? "wgs" : sim.strategy == ST_TRANS ? "trans" : "templ");
  if (sim.method == ME_QS) fprintf(stderr, "method : qshmm\nqshmm : %s\n", sim.model_file);
  else if (sim.method == ME_ERR) fprintf(stderr, "method : errhmm\nerrhmm : %s\n", sim.model_file);
  else fprintf(stderr, "method : sample\n");
  if (sim.strategy == ST_WGS) fprintf(stderr, "genome : %s\n", sim.genome_file);
  else if (sim.strategy == ST_TRANS) fprintf(stderr, "transcript : %s\n", sim.transcript_file);
  else fprintf(stderr, "template : %s\n", sim.templ_file);
  fprintf(stderr, "prefix : %s\n", sim.prefix);
  fprintf(stderr, "id-prefix : %s\n", sim.id_prefix);
  if (sim.strategy == ST_WGS) fprintf(stderr, "depth : %lf\n", sim.depth);
  if (sim.strategy == ST_TEMPL) {
  } else if (sim.method == ME_QS || sim.method == ME_ERR) {
    fprintf(stderr, "length-mean : %f\n", sim.len_mean);
    fprintf(stderr, "length-sd : %f\n", sim.len_sd);
    fprintf(stderr, "length-min : %ld\n", sim.len_min);
    fprintf(stderr, "length-max : %ld\n", sim.len_max);
  } else {
    fprintf(stderr, "length-mean : (sample FASTQ)\nlength-sd : (sample FASTQ)\n");
    fprintf(stderr, "length-min : %ld\n", sim.len_min);
    fprintf(stderr, "length-max : %ld\n", sim.len_max);
  }
  if (sim.method != ME_ERR)
    fprintf(stderr, "difference-ratio : %ld:%ld:%ld\n", sim.sub_ratio, sim.ins_ratio, sim.del_ratio);
  fprintf(stderr, "seed : %d\n", sim.seed);
  if (sim.method == ME_QS || sim.method == ME_ERR) {
    fprintf(stderr, "accuracy-mean : %f\n", sim.accuracy_mean);
  } else { /* glibc prints "(null)" for the option that was not given */
    fprintf(stderr, "sample : %s\n", sim.sample_file ? sim.sample_file : "(null)");
    fprintf(stderr, "sample-profile-id : %s\n", sim.profile_id ? sim.profile_id : "(null)");
    fprintf(stderr, "accuracy-mean : (sample FASTQ)\naccuracy-sd : (sample FASTQ)\n");
    fprintf(stderr, "accuracy-min : %f\n", sim.accuracy_min);
    fprintf(stderr, "accuracy-max : %f\n", sim.accuracy_max);
  }
  fprintf(stderr, "pass_num : %d\n", sim.pass_num);
  fprintf(stderr, "hp-del-bias : %f\n", sim.hp_del_bias);
  fprintf(stderr, "\n");
}

/* ------------------------------------------------------ sampling method -- */
/* pbsim.cpp:120-137 (sample_t), 1155-1330 (get_sample_inf), 1336-1360 (print_sample_stats),
 * 1694-1949 (simulate_by_sample).  Only the quality strings of the sample FASTQ are used. */
static struct {
  long num, len_min, len_max;
  long long len_total;
  long num_filtered, len_min_filtered, len_max_filtered;
  long long len_total_filtered;
  double len_mean_filtered, len_sd_filtered, accuracy_mean_filtered, accuracy_sd_filtered;
} sample;
static FILE *fp_filtered, *fp_stats;
static char profile_fq[4096], profile_stats[4096];

static void get_sample_inf(void) { /* pbsim.cpp:1155-1330 */
  char line[BUF_SIZE];
  for (long i = 0; i <= sim.len_max; i++) freq_len[i] = 0;
  for (long i = 0; i <= 100000; i++) freq_accuracy[i] = 0;
  memset(&sample, 0, sizeof sample);
  sample.len_min = sample.len_min_filtered = LONG_MAX;
  if (sim.method == ME_SAM_REUSE) { /* the stored stats file, "key\tvalue" lines (:1185-1210) */
    while (fgets(line, BUF_SIZE, fp_stats) != NULL) {
      trim(line);
      char *item = strtok(line, "\t"), *val = strtok(NULL, "\t");
      if (strcmp(item, "num") == 0) sample.num_filtered = atol(val);
      else if (strcmp(item, "len_total") == 0) sample.len_total_filtered = atol(val);
      else if (strcmp(item, "len_min") == 0) sample.len_min_filtered = atol(val);
      else if (strcmp(item, "len_max") == 0) sample.len_max_filtered = atol(val);
      else if (strcmp(item, "len_mean") == 0) sample.len_mean_filtered = atof(val);
      else if (strcmp(item, "len_sd") == 0) sample.len_sd_filtered = atof(val);
      else if (strcmp(item, "accuracy_mean") == 0) sample.accuracy_mean_filtered = atof(val);
      else if (strcmp(item, "accuracy_sd") == 0) sample.accuracy_sd_filtered = atof(val);
    }
    return;
  }
  FILE *fp = fopen(sim.sample_file, "r");
  if (!fp) { fprintf(stderr, "ERROR: Cannot open file: %s\n", sim.sample_file); exit(255); }
  char *qc_tmp = malloc(FASTQ_LEN_MAX + BUF_SIZE);
  double acc_total = 0;
  long len = 0;
  int line_num = 0;
  qc_tmp[0] = '\0';
  /* records are counted by line feeds: the 4th one closes a read; a line longer than the buffer arrives in
   * chunks without a line feed, of which only those of the quality line (line_num == 3) are kept (:1216-1283) */
  while (fgets(line, BUF_SIZE, fp) != NULL) {
    if (trim(line) == 1) {
      if (++line_num < 4) continue;
      len += (long)strlen(line);
      if (len > FASTQ_LEN_MAX) { fprintf(stderr, "ERROR: fastq is too long. Max acceptable length is %d.\n", FASTQ_LEN_MAX); exit(255); }
      sample.num++;
      sample.len_total += len;
      if (len > sample.len_max) sample.len_max = len;
      if (len < sample.len_min) sample.len_min = len;
      if (len >= sim.len_min && len <= sim.len_max) {
        strcat(qc_tmp, line);
        double prob = 0.0;
        for (long i = 0; i < len; i++) prob += qc_prob[(int)qc_tmp[i] - 33];
        const double accuracy = 1.0 - (prob / len);
        if (accuracy >= sim.accuracy_min && accuracy <= sim.accuracy_max) {
          acc_total += accuracy;
          sample.num_filtered++;
          sample.len_total_filtered += len;
          freq_len[len]++;
          freq_accuracy[(int)(accuracy * 100000 + 0.5)]++;
          fprintf(fp_filtered, "%s\n", qc_tmp);
          if (len > sample.len_max_filtered) sample.len_max_filtered = len;
          if (len < sample.len_min_filtered) sample.len_min_filtered = len;
        }
      }
      line_num = 0;
      qc_tmp[0] = '\0';
      len = 0;
    } else if (line_num == 3) {
      len += (long)strlen(line);
      if (len > FASTQ_LEN_MAX) { fprintf(stderr, "ERROR: fastq is too long. Max acceptable length is %d.\n", FASTQ_LEN_MAX); exit(255); }
      strcat(qc_tmp, line);
    }
  }
  fclose(fp);
  free(qc_tmp);
  if (sample.num_filtered < 1) die("there is no sample in the valid range of length and accuracy.");
  sample.len_mean_filtered = (double)sample.len_total_filtered / sample.num_filtered;
  sample.accuracy_mean_filtered = acc_total / sample.num_filtered;
  double variance = 0.0;
  for (long i = 0; i <= sim.len_max; i++)
    if (freq_len[i] > 0) variance += pow((sample.len_mean_filtered - i), 2) * freq_len[i];
  sample.len_sd_filtered = sqrt(variance / sample.num_filtered);
  variance = 0.0;
  for (long i = 0; i <= 100000; i++)
    if (freq_accuracy[i] > 0) variance += pow((sample.accuracy_mean_filtered - i * 0.00001), 2) * freq_accuracy[i];
  sample.accuracy_sd_filtered = sqrt(variance / sample.num_filtered);
  if (sim.method == ME_SAM_STORE) { /* :1317-1326 */
    fprintf(fp_stats, "num\t%ld\nlen_total\t%lld\nlen_min\t%ld\nlen_max\t%ld\n", sample.num_filtered,
            sample.len_total_filtered, sample.len_min_filtered, sample.len_max_filtered);
    fprintf(fp_stats, "len_mean\t%f\nlen_sd\t%f\naccuracy_mean\t%f\naccuracy_sd\t%f\n", sample.len_mean_filtered,
            sample.len_sd_filtered, sample.accuracy_mean_filtered, sample.accuracy_sd_filtered);
  }
}

static void print_sample_stats(void) { /* pbsim.cpp:1336-1360 */
  fprintf(stderr, ":::: sample reads stats ::::\n\n");
  if (sim.method == ME_SAM_REUSE) {
    fprintf(stderr, "file name : %s\n", profile_fq);
  } else {
    fprintf(stderr, "file name : %s\n", sim.sample_file);
    fprintf(stderr, "\n:: all reads ::\n");
    fprintf(stderr, "read num. : %ld\n", sample.num);
    fprintf(stderr, "read total length : %lld\n", sample.len_total);
    fprintf(stderr, "read min length : %ld\n", sample.len_min);
    fprintf(stderr, "read max length : %ld\n", sample.len_max);
  }
  fprintf(stderr, "\n:: filtered reads ::\n");
  fprintf(stderr, "read num. : %ld\n", sample.num_filtered);
  fprintf(stderr, "read total length : %lld\n", sample.len_total_filtered);
  fprintf(stderr, "read min length : %ld\n", sample.len_min_filtered);
  fprintf(stderr, "read max length : %ld\n", sample.len_max_filtered);
  fprintf(stderr, "read length mean (SD) : %f (%f)\n", sample.len_mean_filtered, sample.len_sd_filtered);
  fprintf(stderr, "read accuracy mean (SD) : %f (%f)\n", sample.accuracy_mean_filtered, sample.accuracy_sd_filtered);
  fprintf(stderr, "\n");
}

static void open_sample_profile(void) { /* pbsim.cpp:580-617 */
  if (sim.method == ME_SAM) {
    if (!(fp_filtered = tmpfile())) die("Cannot open temporary file");
  } else {
    const char *mode = sim.method == ME_SAM_STORE ? "w+" : "r";
    if (!(fp_filtered = fopen(profile_fq, mode)) || !(fp_stats = fopen(profile_stats, mode)))
      die("Cannot open sample_profile");
  }
  get_sample_inf();
  print_sample_stats();
}

/* One record under the sampling method (pbsim.cpp:1694-1949).  Every filtered quality string is used
 * sample_num (or sample_num + 1) times per sweep over the profile; the string is cut to the length of the read it
 * just produced (:1834), so the copies of one string form a chain of non-increasing lengths. */
static void simulate_sample_record(void) {
  const long long len_quota = (long long)(sim.depth * ref_len);
  long long len_total = 0;
  for (long i = 0; i <= sim.len_max; i++) freq_len[i] = 0;
  for (long i = 0; i <= 100000; i++) freq_accuracy[i] = 0;
  long sample_num = (long)(len_quota / sample.len_total_filtered);
  const long sample_residue = (long)(len_quota % sample.len_total_filtered);
  long sample_interval = 1;
  if (sample_residue != 0) {
    sample_interval = (long)((double)(sample.len_total_filtered / sample_residue) * 2 + 0.5); /* integer quotient first */
    if (sample_interval > (long)(sample.num_filtered * 0.5)) sample_interval = (long)(sample.num_filtered * 0.5);
  }
  g_unit = (uint32_t)ref_num;
  g_pass = 0;
  while (len_total < len_quota) {
    rewind(fp_filtered);
    g_read = (uint32_t)(sim.res_num + 1);
    long sample_value = R_hdr(3) % sample.num_filtered; /* :1732 */
    while (fgets(m_qc, (int)(sample.len_max_filtered + 2), fp_filtered) != NULL) {
      if (len_total >= len_quota) break;
      trim(m_qc);
      const long num = (sample_value % sample_interval == 0) ? sample_num + 1 : sample_num;
      sample_value++;
      for (long i = 0; i < num; i++) {
        if (len_total >= len_quota) break;
        long L = (long)strlen(m_qc), offset;
        g_read = (uint32_t)(sim.res_num + 1);
        if (L >= ref_len) {
          offset = 0;
          L = ref_len;
        } else {
          offset = R_hdr(2) % (ref_len - L + 1);
        }
        sim.res_num++;
        for (long j = 0; j < L; j++) {
          m_seq[j] = ref_seq[offset + j];
          m_hp[j] = ref_hp[offset + j];
        }
        m_seq[L] = '\0';
        char strand = '+';
        if (sim.res_num % 2 != 1) {
          strand = '-';
          revcomp(m_seq);
          revshort(m_hp, L);
        }
        long ref_offset = 0, read_offset = 0, maf_offset = 0;
        while (ref_offset < L && read_offset < L) { /* :1777-1832 */
          const char nt = m_seq[ref_offset];
          int qv = (int)m_qc[read_offset] - 33;
          g_event = (uint32_t)maf_offset;
          long rand_value = R_walk(0, 2) % 1000000;
          if (rand_value < sub_thre[qv]) {
            int need_n;
            sim.res_sub_num++;
            char b = sub_nt(nt, R_walk(0, 3) % 3, &need_n);
            if (need_n) b = "ATGC"[R_walk(1, 0) % 4];
            m_read[read_offset] = b;
            m_mafref[maf_offset] = nt;
            ref_offset++;
          } else if (rand_value < ins_thre[qv]) {
            sim.res_ins_num++;
            const long index = R_walk(0, 3) % 8;
            m_read[read_offset] = (index >= 4) ? nt : "ATGC"[index];
            m_mafref[maf_offset] = '-';
          } else {
            m_read[read_offset] = nt;
            m_mafref[maf_offset] = nt;
            ref_offset++;
          }
          m_maf[maf_offset] = m_read[read_offset];
          maf_offset++;
          read_offset++;
          while (ref_offset < L && read_offset < L) {
            /* mut.hp[-1] when no reference base has been consumed yet: observed 0 (see walk_qshmm, Q15) */
            const int hp = (ref_offset > 0) ? m_hp[ref_offset - 1] : 0;
            g_event = (uint32_t)maf_offset;
            rand_value = R_walk(2, 0) % 1000000;
            qv = (int)m_qc[read_offset - 1] - 33;
            if (!(rand_value < del_thre[qv] * hp_bias[hp])) break;
            sim.res_del_num++;
            m_maf[maf_offset] = '-';
            m_mafref[maf_offset] = m_seq[ref_offset];
            maf_offset++;
            ref_offset++;
          }
        }
        m_qc[read_offset] = '\0'; /* the chain: the next copy of this string is this much shorter */
        m_read[read_offset] = '\0';
        m_maf[maf_offset] = '\0';
        m_mafref[maf_offset] = '\0';
        if (strand == '-') {
          revcomp(m_maf);
          revcomp(m_mafref);
        }
        const long len = (long)strlen(m_read);
        len_total += len;
        double prob = 0.0;
        for (long j = 0; j < len; j++) prob += qc_prob[(int)m_qc[j] - 33];
        account(len, 1.0 - (prob / len));
        emit_record(len, m_qc, 0, "ref", 3, offset + 1, offset + ref_offset, strand, 1);
      }
    }
    sample_num = 0;
  }
  finish_stats();
}

/* --------------------------------------------------------- WGS strategy -- */
static long ref_max_len;

static void get_genome_inf(void) { /* pbsim.cpp:896-991 */
  FILE *fp, *fp_out = NULL;
  char line[BUF_SIZE], name[4096];
  int ret;
  fprintf(stderr, ":::: Reference stats ::::\n\n");
  fprintf(stderr, "file name : %s\n", sim.genome_file);
  fprintf(stderr, "\n");
  if ((fp = fopen(sim.genome_file, "r")) == NULL) {
    fprintf(stderr, "ERROR: Cannot open file: %s\n", sim.genome_file);
    exit(255);
  }
  ref_num_seq = 0;
  ref_len = 0;
  while (fgets(line, BUF_SIZE, fp) != NULL) {
    ret = trim(line);
    if (line[0] == '>') {
      if (ref_num_seq != 0) {
        if (ref_len < REF_SEQ_LEN_MIN) {
          fprintf(stderr, "ERROR: Reference is too short. Acceptable length >= %d.\n", REF_SEQ_LEN_MIN);
          exit(255);
        }
        fprintf(stderr, "ref.%ld (len:%ld) : %s\n", ref_num_seq, ref_len, ref_id);
        fclose(fp_out);
        if (ref_len > ref_max_len) ref_max_len = ref_len;
      }
      ref_num_seq++;
      if (ref_num_seq > REF_SEQ_NUM_MAX) {
        fprintf(stderr, "ERROR: References are too many. Max number of reference is %d.\n", REF_SEQ_NUM_MAX);
        exit(255);
      }
      strncpy(ref_id, line + 1, REF_ID_LEN_MAX);
      ref_id[REF_ID_LEN_MAX] = '\0';
      snprintf(name, sizeof name, "%s_%04ld.ref", sim.prefix, ref_num_seq);
      if ((fp_out = fopen(name, "w")) == NULL) {
        fprintf(stderr, "ERROR: Cannot open output file: %s\n", name);
        exit(255);
      }
      ref_len = 0;
      while (ret != 1) {
        if (fgets(line, BUF_SIZE, fp) == NULL) break;
        ret = trim(line);
      }
      fprintf(fp_out, ">%s\n", ref_id);
    } else {
      if (!fp_out) die("oracle: sequence data before the first FASTA header (reference UB)");
      ref_len += (long)strlen(line);
      if (ref_len > REF_SEQ_LEN_MAX) {
        fprintf(stderr, "ERROR: Reference is too long. Acceptable length <= %ld.\n", REF_SEQ_LEN_MAX);
        exit(255);
      }
      fprintf(fp_out, "%s\n", line);
    }
  }
  fclose(fp);
  if (ref_len < REF_SEQ_LEN_MIN) {
    fprintf(stderr, "ERROR: Reference is too short. Acceptable length >= %d.\n", REF_SEQ_LEN_MIN);
    exit(255);
  }
  fprintf(stderr, "ref.%ld (len:%ld) : %s\n", ref_num_seq, ref_len, ref_id);
  fclose(fp_out);
  if (ref_len > ref_max_len) ref_max_len = ref_len;
  fprintf(stderr, "\n");
  ref_seq = (char *)malloc(ref_max_len + 1);
  ref_hp = (short *)malloc(ref_max_len * sizeof(short) + 2);
  if (!ref_seq || !ref_hp) die("Cannot allocate memory.");
}

static void get_genome_seq(void) { /* pbsim.cpp:997-1068 */
  FILE *fp;
  char line[BUF_SIZE], name[4096];
  long offset = 0;
  int ret;
  snprintf(name, sizeof name, "%s_%04ld.ref", sim.prefix, ref_num);
  if ((fp = fopen(name, "r")) == NULL) {
    fprintf(stderr, "ERROR: Cannot open file: %s\n", name);
    exit(255);
  }
  while (fgets(line, BUF_SIZE, fp) != NULL) {
    ret = trim(line);
    if (line[0] == '>') {
      while (ret != 1) {
        if (fgets(line, BUF_SIZE, fp) == NULL) break;
        ret = trim(line);
      }
    } else {
      long n = (long)strlen(line);
      memcpy(ref_seq + offset, line, n);
      offset += n;
    }
  }
  fclose(fp);
  ref_seq[offset] = '\0';
  ref_len = (long)strlen(ref_seq);
  for (long i = 0; i < ref_len; i++) ref_seq[i] = (char)toupper(ref_seq[i]);
  compute_hp(ref_seq, ref_len, ref_hp, 1, 0);
  sync_bias_alias();
}

static FILE *open_out(const char *suffix, int wgs) {
  char name[4096];
  if (wgs) snprintf(name, sizeof name, "%s_%04ld.%s", sim.prefix, ref_num, suffix);
  else snprintf(name, sizeof name, "%s.%s", sim.prefix, suffix);
  FILE *fp = fopen(name, "w");
  if (!fp) { fprintf(stderr, "ERROR: Cannot open output file: %s\n", name); exit(255); }
  return fp;
}

static void open_outputs(int wgs) { /* pbsim.cpp:707-730 */
  if (sim.pass_num == 1) {
    fp_fq = open_out("fq", wgs);
  } else {
    fp_sam = open_out("sam", wgs);
    fprintf(fp_sam, "@HD\tVN:1.5\tSO:unknown\tpb:3.0.7\n");
    if (wgs)
      fprintf(fp_sam, "@RG\tID:ffffffff\tPL:PACBIO\tDS:READTYPE=SUBREAD;Ipd:CodecV1=ip;PulseWidth:CodecV1=pw;BINDINGKIT=101-789-500;SEQUENCINGKIT=101-826-100;BASECALLERVERSION=5.0.0;FRAMERATEHZ=100.000000\tPU:%s%ld\tPM:SEQUELII\n", sim.id_prefix, ref_num);
    else
      fprintf(fp_sam, "@RG\tID:ffffffff\tPL:PACBIO\tDS:READTYPE=SUBREAD;Ipd:CodecV1=ip;PulseWidth:CodecV1=pw;BINDINGKIT=101-789-500;SEQUENCINGKIT=101-826-100;BASECALLERVERSION=5.0.0;FRAMERATEHZ=100.000000\tPU:%s\tPM:SEQUELII\n", sim.id_prefix);
  }
  fp_maf = open_out("maf", wgs);
}

static void close_outputs(void) {
  if (sim.pass_num == 1) fclose(fp_fq); else fclose(fp_sam);
  fclose(fp_maf);
}

/* pbsim.cpp:3792-4080 (errhmm) and 2173-2385 (qshmm): one FASTA record */
static void simulate_wgs_record(void) {
  long long len_total = 0, len_quota = (long long)(sim.depth * ref_len); /* pbsim.cpp:705 */
  int rate_mag = 0;
  while (len_total < len_quota) {
    g_unit = (uint32_t)ref_num;
    g_read = (uint32_t)(sim.res_num + 1);
    long L = prob2len[R_hdr(0) % len_rand_value + 1];
    if (len_total + L > len_quota) { /* Q10 */
      L = (long)(len_quota - len_total);
      if (L < sim.len_min) L = sim.len_min;
    }
    int acc = (int)prob2accuracy[R_hdr(1) % accuracy_rand_value + 1];
    long offset;
    if (L >= ref_len) {
      offset = 0;
      L = ref_len;
    } else {
      offset = R_hdr(2) % (ref_len - L + 1);
    }
    long seq_left = offset + 1, seq_right = offset + L;
    sim.res_num++;
    for (long i = 0; i < L; i++) {
      m_seq[i] = ref_seq[offset + i];
      m_hp[i] = ref_hp[offset + i];
    }
    m_seq[L] = '\0';
    char strand = '+';
    if (sim.res_num % 2 != 1) { /* Q9 */
      strand = '-';
      revcomp(m_seq);
      revshort(m_hp, L);
    }
    if (sim.method == ME_ERR) rate_mag = errhmm_rate_mag(acc, rate_mag);
    for (long h = 0; h < sim.pass_num; h++) {
      g_pass = (uint32_t)h;
      long len = (sim.method == ME_ERR) ? walk_errhmm(L, acc, rate_mag, strand) : walk_qshmm(L, acc, strand);
      if (h == 0) len_total += len;
      emit_record(len, sim.method == ME_ERR ? m_newqc : m_qc, h, "ref", 3, seq_left, seq_right, strand, 1);
    }
  }
  finish_stats();
}

static void run_wgs(void) { /* pbsim.cpp:667-759 */
  get_genome_inf();
  if (sim.hp_del_bias == 1) {
    set_bias_default();
  } else {
    for (int i = 0; i <= 10; i++) hpfreq[i] = 0;
    for (ref_num = 1; ref_num <= ref_num_seq; ref_num++) get_genome_seq();
    normalise_bias();
  }
  const int sampling = sim.method >= ME_SAM;
  if (!sampling) {
    build_len_table();
    build_acc_table();
    if (sim.method == ME_ERR) build_errhmm_tables(1); else build_qshmm_tables();
  }
  for (ref_num = 1; ref_num <= ref_num_seq; ref_num++) {
    get_genome_seq();
    init_sim_res();
    open_outputs(1);
    if (sampling) simulate_sample_record(); else simulate_wgs_record();
    print_simulation_stats();
    close_outputs();
  }
  if (sim.method == ME_SAM_STORE || sim.method == ME_SAM_REUSE) { /* pbsim.cpp:756-759 */
    fclose(fp_filtered);
    fclose(fp_stats);
  }
}

/* ------------------------------------------------------- trans strategy -- */
/* Streams the TSV with the reference's fgets chunking (pbsim.cpp:1095-1120,
 * 4429-4451).  cb(mode) is called once per completed transcript line. */
static char tr_id[REF_ID_LEN_MAX + 1];
static long tr_plus, tr_minus;

static long scan_transcripts(int mode);


static void simulate_trans_unit(void) { /* pbsim.cpp:4453-4769 (errhmm), 2774-3017 (qshmm) */
  long read_num = (int)(tr_plus + tr_minus);
  int rate_mag = 0;
  static int rate_mag_persist = 0;
  rate_mag = rate_mag_persist;
  for (long i = 1; i <= read_num; i++) {
    g_unit = 0;
    g_read = (uint32_t)(sim.res_num + 1);
    long L = prob2len[R_hdr(0) % len_rand_value + 1];
    int acc = (int)prob2accuracy[R_hdr(1) % accuracy_rand_value + 1];
    int rank = (int)ceil((double)ref_len / 1000);
    long index = R_hdr(2) % ssp_rand_value[rank] + 1;
    double value = (prob2ssp[rank][index] == 0) ? 0.0 : ((double)prob2ssp[rank][index] - 2.5) / 100;
    long offset = (int)((double)ref_len * value + 0.5);
    if (offset + L > ref_len) L = ref_len - offset;
    long seq_left = offset + 1, seq_right = offset + L;
    sim.res_num++;
    for (long j = 0; j < L; j++) {
      m_seq[j] = ref_seq[offset + j];
      m_hp[j] = ref_hp[offset + j];
    }
    m_seq[L] = '\0';
    char strand = '+';
    if (i > tr_plus) {
      strand = '-';
      revcomp(m_seq);
      revshort(m_hp, L);
    }
    if (sim.method == ME_ERR) rate_mag = errhmm_rate_mag(acc, rate_mag);
    for (long h = 0; h < sim.pass_num; h++) {
      g_pass = (uint32_t)h;
      long len = (sim.method == ME_ERR) ? walk_errhmm(L, acc, rate_mag, strand) : walk_qshmm(L, acc, strand);
      emit_record(len, sim.method == ME_ERR ? m_newqc : m_qc, h, tr_id, (int)strlen(tr_id), seq_left, seq_right,
                  strand, 0);
    }
    /* Q5: the verbatim copy of an accuracy-100 read, `for (i=0; i<mut.len; i++)` (pbsim.cpp:4533), runs on the SAME `i` as
     * the per-transcript read loop (pbsim.cpp:4487): behind such a read the loop continues at i = mut.len + 1, so a
     * transcript makes fewer (or, with mut.len < i, repeated) reads than its expression value says. */
    if (sim.method == ME_ERR && acc == 100) i = L > 0 ? L : 0;
  }
  rate_mag_persist = rate_mag;
}

/* mode 0: stats pass (get_transcript_inf, pbsim.cpp:1075-1136) -> returns max_len
 * mode 1: hp-bias pre-pass (4356-4412)
 * mode 2: simulation pass (4428-4770) */
static long tr_num_seq, tr_total_exp;
static long scan_transcripts(int mode) {
  FILE *fp = fopen(sim.transcript_file, "r");
  char line[BUF_SIZE], *tp;
  int flg1 = 1, flg2;
  long offset = 0, max_len = 0, cur_len = 0;
  if (!fp) { fprintf(stderr, "ERROR: Cannot open file: %s\n", sim.transcript_file); exit(255); }
  while (fgets(line, BUF_SIZE, fp) != NULL) {
    flg2 = (trim(line) == 1);
    if (flg1 == 1) {
      tp = strtok(line, "\t");
      if (mode == 0) {
        tr_num_seq++;
        tr_total_exp += atoi(strtok(NULL, "\t"));
        tr_total_exp += atoi(strtok(NULL, "\t"));
        tp = strtok(NULL, "\t");
        cur_len = (long)strlen(tp);
      } else {
        strncpy(tr_id, tp, REF_ID_LEN_MAX);
        tr_id[REF_ID_LEN_MAX] = '\0';
        tr_plus = atoi(strtok(NULL, "\t"));
        tr_minus = atoi(strtok(NULL, "\t"));
        tp = strtok(NULL, "\t");
        long n = (long)strlen(tp);
        memcpy(ref_seq, tp, n);
        offset = n;
      }
    } else {
      long n = (long)strlen(line);
      if (mode == 0) cur_len += n;
      else { memcpy(ref_seq + offset, line, n); offset += n; }
    }
    if (flg2 == 1) {
      if (mode == 0) {
        if (cur_len > max_len) max_len = cur_len;
      } else {
        ref_seq[offset] = '\0';
        ref_len = (long)strlen(ref_seq);
        /* Q6: errhmm trans upper-cases seq[1..len]; qshmm trans seq[0..len-1]
         * (pbsim.cpp:4457-4459 vs 2778-2780) */
        if (sim.method == ME_ERR) { for (long i = 1; i <= ref_len; i++) ref_seq[i] = (char)toupper(ref_seq[i]); }
        else { for (long i = 0; i < ref_len; i++) ref_seq[i] = (char)toupper(ref_seq[i]); }
        if (mode == 1) {
          compute_hp(ref_seq, ref_len, ref_hp, 0, (long)(int)(tr_plus + tr_minus));
        } else {
          compute_hp(ref_seq, ref_len, ref_hp, 0, 0);
          simulate_trans_unit();
        }
      }
    }
    flg1 = flg2;
  }
  fclose(fp);
  return max_len;
}

static void run_trans(void) { /* pbsim.cpp:761-812 */
  long max_len = scan_transcripts(0);
  ref_seq = (char *)malloc(max_len + 2);
  ref_hp = (short *)malloc((max_len + 2) * sizeof(short));
  int rank_max = (int)ceil((float)max_len / 1000);
  fprintf(stderr, ":::: transcript stats ::::\n\n");
  fprintf(stderr, "file name : %s\n", sim.transcript_file);
  fprintf(stderr, "transcript num : %ld\n", tr_num_seq);
  fprintf(stderr, "total expression value : %ld\n", tr_total_exp);
  fprintf(stderr, "\n");
  init_sim_res();
  open_outputs(0);
  build_len_table();
  build_ssp_table(rank_max);
  build_acc_table();
  if (sim.method == ME_ERR) build_errhmm_tables(0); else build_qshmm_tables();
  if (sim.hp_del_bias == 1) {
    set_bias_default();
  } else {
    for (int i = 0; i <= 10; i++) hpfreq[i] = 0;
    scan_transcripts(1);
    normalise_bias();
  }
  sync_bias_alias();
  scan_transcripts(2);
  finish_stats();
  print_simulation_stats();
  close_outputs();
}

/* ------------------------------------------------------- templ strategy -- */
/* get_templ_inf (pbsim.cpp:1366-1418) + simulate_by_{errhmm,qshmm}_templ
 * (:4807-5392, :3055-3587): every FASTA record is one full-length template,
 * one read each, '+' strand, offset 0; the header draws the accuracy only. */
static long tp_num;
static long long tp_len_total;

static void templ_unit(int mode) {
  ref_len = (long)strlen(ref_seq);
  for (long i = 1; i <= ref_len; i++) ref_seq[i] = (char)toupper(ref_seq[i]); /* pbsim.cpp:5062-5064 (index 0 keeps its case) */
  if (mode == 1) { /* bias pre-pass: hpfreq[v] += run length (pbsim.cpp:4995-5016) */
    compute_hp(ref_seq, ref_len, ref_hp, 0, 1);
    return;
  }
  compute_hp(ref_seq, ref_len, ref_hp, 0, 0);
  static int rate_mag = 0;
  g_unit = 0;
  g_read = (uint32_t)(sim.res_num + 1);
  int acc = (int)prob2accuracy[R_hdr(1) % accuracy_rand_value + 1];
  long L = ref_len;
  sim.res_num++;
  memcpy(m_seq, ref_seq, (size_t)L);
  memcpy(m_hp, ref_hp, (size_t)L * sizeof(short));
  m_seq[L] = '\0';
  if (sim.method == ME_ERR) rate_mag = errhmm_rate_mag(acc, rate_mag);
  for (long h = 0; h < sim.pass_num; h++) {
    g_pass = (uint32_t)h;
    long len = (sim.method == ME_ERR) ? walk_errhmm(L, acc, rate_mag, '+') : walk_qshmm(L, acc, '+');
    /* the MAF line names the template by its id but pads as if it were "ref" (digit_num1[0] = 3, pbsim.cpp:5290) */
    emit_record(len, sim.method == ME_ERR ? m_newqc : m_qc, h, tr_id, 3, 1, L, '+', 0);
  }
}

/* mode 0: stats; 1: hp census; 2: simulate.  Control flow of pbsim.cpp:5055-5362. */
static void scan_templates(int mode) {
  FILE *fp = fopen(sim.templ_file, "r");
  char line[BUF_SIZE];
  long offset = 0, seqlen = 0;
  if (!fp) { fprintf(stderr, "ERROR: Cannot open file: %s\n", sim.templ_file); exit(255); }
  while (1) {
    char *rp = fgets(line, BUF_SIZE, fp);
    if (mode != 0 && ((rp == NULL) || (line[0] == '>')) && (offset != 0)) {
      ref_seq[offset] = '\0';
      templ_unit(mode);
    }
    if (rp == NULL) break;
    int ret = trim(line);
    if (line[0] == '>') {
      if (mode == 0) { tp_num++; seqlen = 0; }
      strncpy(tr_id, line + 1, REF_ID_LEN_MAX); /* pbsim.cpp:5344-5345 */
      tr_id[REF_ID_LEN_MAX] = '\0';
      offset = 0;
      while (ret != 1) {
        if (fgets(line, BUF_SIZE, fp) == NULL) break;
        ret = trim(line);
      }
    } else {
      long n = (long)strlen(line);
      if (mode == 0) {
        seqlen += n;
        tp_len_total += n;
        if (seqlen > 1000000) die("template is too long. Max acceptable length is 1000000.");
      } else {
        memcpy(ref_seq + offset, line, n);
        offset += n;
      }
    }
  }
  fclose(fp);
}

static void run_templ(void) { /* pbsim.cpp:814-866 */
  scan_templates(0);
  ref_seq = (char *)malloc(1000000 + 2);
  ref_hp = (short *)malloc((1000000 + 2) * sizeof(short));
  fprintf(stderr, ":::: Template stats ::::\n\n");
  fprintf(stderr, "file name : %s\n", sim.templ_file);
  fprintf(stderr, "template num. : %ld\n", tp_num);
  fprintf(stderr, "template total length : %lld\n", tp_len_total);
  fprintf(stderr, "\n");
  init_sim_res();
  open_outputs(0);
  build_acc_table();
  if (sim.method == ME_ERR) build_errhmm_tables(0); else build_qshmm_tables();
  if (sim.hp_del_bias == 1) {
    set_bias_default();
  } else {
    for (int i = 0; i <= 10; i++) hpfreq[i] = 0;
    scan_templates(1);
    normalise_bias();
  }
  sync_bias_alias();
  scan_templates(2);
  finish_stats();
  print_simulation_stats();
  close_outputs();
}

/* ------------------------------------------------------------------ CLI -- */
static void set_sim_param(void) { /* pbsim.cpp:1451-1688 */
  if (!sim.set_flg[0] || !sim.set_flg[1]) die("--strategy and --method must be set.");
  if (sim.strategy != ST_WGS && sim.method == ME_SAM) die("sampling-based simulation is possible only for wgs strategy.");
  if (sim.strategy == ST_WGS && !sim.set_flg[2]) die("for --strategy wgs, --genome must be set.");
  if (sim.strategy == ST_TRANS && !sim.set_flg[3]) die("for --strategy trans, --transcript must be set.");
  if (sim.strategy == ST_TEMPL && !sim.set_flg[21]) die("for --strategy templ, --template must be set.");
  if (!sim.set_flg[4]) sim.prefix = "sd";
  if (!sim.set_flg[5]) sim.id_prefix = "S";
  if (!sim.set_flg[6]) sim.depth = 20.0;
  if (!sim.set_flg[7]) sim.len_min = 100;
  if (!sim.set_flg[8]) sim.len_max = 1000000;
  if (!sim.set_flg[9]) { sim.sub_ratio = 6; sim.ins_ratio = 55; sim.del_ratio = 39; }
  long sum = sim.sub_ratio + sim.ins_ratio + sim.del_ratio;
  sim.sub_rate = (double)sim.sub_ratio / sum;
  sim.ins_rate = (double)sim.ins_ratio / sum;
  sim.del_rate = (double)sim.del_ratio / sum;
  if (sim.method == ME_SAM) { /* pbsim.cpp:1567-1620 */
    if (sim.set_flg[11]) {
      if (sim.set_flg[12]) sim.method = ME_SAM_STORE;
    } else if (sim.set_flg[12]) {
      sim.method = ME_SAM_REUSE;
    } else {
      die("for --method sample, --sample (and/or --sample-profile-id) must be set.");
    }
  }
  if (sim.set_flg[12]) {
    snprintf(profile_fq, sizeof profile_fq, "sample_profile_%s.fastq", sim.profile_id);
    snprintf(profile_stats, sizeof profile_stats, "sample_profile_%s.stats", sim.profile_id);
  }
  if (sim.method == ME_SAM_STORE || sim.method == ME_SAM_REUSE) {
    const char *names[2] = {profile_fq, profile_stats};
    for (int k = 0; k < 2; k++) {
      FILE *fp = fopen(names[k], "r");
      if (fp && sim.method == ME_SAM_STORE) { fprintf(stderr, "ERROR: %s exists.\n", names[k]); exit(255); }
      if (!fp && sim.method == ME_SAM_REUSE) { fprintf(stderr, "ERROR: %s does not exist.\n", names[k]); exit(255); }
      if (fp) fclose(fp);
    }
  }
  sim.accuracy_min = sim.set_flg[13] ? (int)(sim.accuracy_min * 100) * 0.01 : 0.75;
  sim.accuracy_max = sim.set_flg[14] ? (int)(sim.accuracy_max * 100) * 0.01 : 1.0;
  if (sim.method == ME_QS && !sim.set_flg[15]) die("for --method qshmm, --qshmm must be set.");
  if (sim.method == ME_ERR && !sim.set_flg[16]) die("for --method errhmm, --errhmm must be set.");
  if (!sim.set_flg[17]) sim.len_mean = 9000;
  if (!sim.set_flg[18]) sim.len_sd = 7000;
  if (sim.set_flg[19]) sim.accuracy_mean = (int)(sim.accuracy_mean * 100) * 0.01;
  else sim.accuracy_mean = 0.85;
  if (sim.len_min > sim.len_max) {
    fprintf(stderr, "ERROR: length min(%ld) is greater than max(%ld).\n", sim.len_min, sim.len_max);
    exit(255);
  }
  if (!sim.set_flg[20]) sim.pass_num = 1;
  if (sim.pass_num > 1 && sim.method >= ME_SAM) die("sampling-based simulation supports only single-pass.");
  if (!sim.set_flg[22]) sim.hp_del_bias = 1;
}

int main(int argc, char **argv) {
  static struct option long_options[] = {
      {"strategy", 1, NULL, 0}, {"method", 1, NULL, 0}, {"genome", 1, NULL, 0}, {"transcript", 1, NULL, 0},
      {"prefix", 1, NULL, 0}, {"id-prefix", 1, NULL, 0}, {"depth", 1, NULL, 0}, {"length-min", 1, NULL, 0},
      {"length-max", 1, NULL, 0}, {"difference-ratio", 1, NULL, 0}, {"seed", 1, NULL, 0}, {"sample", 1, NULL, 0},
      {"sample-profile-id", 1, NULL, 0}, {"accuracy-min", 1, NULL, 0}, {"accuracy-max", 1, NULL, 0},
      {"qshmm", 1, NULL, 0}, {"errhmm", 1, NULL, 0}, {"length-mean", 1, NULL, 0}, {"length-sd", 1, NULL, 0},
      {"accuracy-mean", 1, NULL, 0}, {"pass-num", 1, NULL, 0}, {"template", 1, NULL, 0}, {"hp-del-bias", 1, NULL, 0},
      {"rng", 1, NULL, 0}, {"draw-count", 0, NULL, 0}, {0, 0, 0, 0}};
  int opt, option_index = 0, want_count = 0;
  sim.seed = 1;
  sim.rng = RNG_GLIBC;
  while ((opt = getopt_long(argc, argv, "", long_options, &option_index)) != -1) {
    if (opt != 0) exit(255);
    sim.set_flg[option_index] = 1;
    switch (option_index) {
    case 0:
      if (strncmp(optarg, "wgs", 3) == 0) sim.strategy = ST_WGS;
      else if (strncmp(optarg, "trans", 5) == 0) sim.strategy = ST_TRANS;
      else if (strncmp(optarg, "templ", 5) == 0) sim.strategy = ST_TEMPL;
      else die("strategy: Acceptable value: wgs, trans, templ.");
      break;
    case 1:
      if (strncmp(optarg, "qshmm", 5) == 0) sim.method = ME_QS;
      else if (strncmp(optarg, "errhmm", 6) == 0) sim.method = ME_ERR;
      else if (strncmp(optarg, "sample", 6) == 0) sim.method = ME_SAM;
      else die("method: Acceptable value: qshmm, errhmm, sample.");
      break;
    case 2: sim.genome_file = optarg; break;
    case 3: sim.transcript_file = optarg; break;
    case 4: sim.prefix = optarg; break;
    case 5: sim.id_prefix = optarg; break;
    case 6: sim.depth = atof(optarg); if (sim.depth <= 0.0) die("depth: Acceptable range is more than 0."); break;
    case 7: sim.len_min = atoi(optarg); if (strlen(optarg) >= 8 || sim.len_min < 1 || sim.len_min > FASTQ_LEN_MAX) die("length-min: Acceptable range is 1-1000000."); break;
    case 8: sim.len_max = atoi(optarg); if (strlen(optarg) >= 8 || sim.len_max < 1 || sim.len_max > FASTQ_LEN_MAX) die("length-max: Acceptable range is 1-1000000."); break;
    case 9: {
      char *buf = strdup(optarg), *tp = strtok(buf, ":");
      for (int num = 0; num < 3; num++) {
        if (!tp) die("difference-ratio: Format is sub:ins:del.");
        long ratio = atoi(tp);
        if (strlen(tp) >= 5 || ratio < 0 || ratio > 1000) die("difference-ratio: Acceptable range is 0-1000.");
        if (num == 0) sim.sub_ratio = ratio; else if (num == 1) sim.ins_ratio = ratio; else sim.del_ratio = ratio;
        tp = strtok(NULL, ":");
      }
      break;
    }
    case 10: sim.seed = (unsigned int)atoi(optarg); break;
    case 11: sim.sample_file = optarg; break;
    case 12: sim.profile_id = optarg; break;
    case 13: sim.accuracy_min = atof(optarg); if (sim.accuracy_min < 0.0 || sim.accuracy_min > 1.0) die("accuracy-min: Acceptable range is 0.0-1.0."); break;
    case 14: sim.accuracy_max = atof(optarg); if (sim.accuracy_max < 0.0 || sim.accuracy_max > 1.0) die("accuracy-max: Acceptable range is 0.0-1.0."); break;
    case 15: case 16: sim.model_file = optarg; break;
    case 17: sim.len_mean = atof(optarg); if (sim.len_mean < 1 || sim.len_mean > FASTQ_LEN_MAX) die("length-mean: Acceptable range is 1-1000000."); break;
    case 18: sim.len_sd = atof(optarg); if (sim.len_sd < 0 || sim.len_sd > FASTQ_LEN_MAX) die("length-sd: Acceptable range is 0-1000000."); break;
    case 19: sim.accuracy_mean = atof(optarg); if (sim.accuracy_mean < 0.0 || sim.accuracy_mean > 1.0) die("accuracy-mean: Acceptable range is 0.0-1.0."); break;
    case 20: sim.pass_num = atoi(optarg); if (sim.pass_num < 1) die("pass_num: Acceptable range is more than 1."); break;
    case 21: sim.templ_file = optarg; break;
    case 22: sim.hp_del_bias = atof(optarg); if (strlen(optarg) >= 8 || sim.hp_del_bias < 1 || sim.hp_del_bias > 10) die("hp-del-bias: Acceptable range is 1-10."); break;
    case 23:
      if (strcmp(optarg, "glibc") == 0) sim.rng = RNG_GLIBC;
      else if (strcmp(optarg, "philox") == 0) sim.rng = RNG_PHILOX;
      else die("rng: glibc or philox");
      break;
    case 24: want_count = 1; break;
    default: break;
    }
  }
  set_sim_param();
  print_sim_param();
  srand(sim.seed); /* pbsim.cpp:543 */
  init_common_tables();
  if (sim.method >= ME_SAM) open_sample_profile(); /* pbsim.cpp:580-617, before the models */
  if (sim.method == ME_QS) set_qshmm(); else if (sim.method == ME_ERR) set_errhmm();
  size_t cap = (size_t)sim.len_max * 2 + 1; /* pbsim.cpp:5488-5531 */
  m_seq = malloc(cap); m_read = malloc(cap); m_maf = malloc(cap); m_mafref = malloc(cap);
  m_qc = malloc(cap); m_newqc = malloc(cap); m_hp = malloc(cap * sizeof(short));
  if (!m_seq || !m_read || !m_maf || !m_mafref || !m_qc || !m_newqc || !m_hp) die("Cannot allocate memory.");
  if (sim.strategy == ST_WGS) run_wgs();
  else if (sim.strategy == ST_TRANS) run_trans();
  else run_templ();
  if (want_count) fprintf(stderr, "oracle draws : %llu\n", g_draws);
  return 0;
}
