// host_tables.cpp -- see host_tables.h.  One-time host work: parse the model
// text, run the reference's double-precision table arithmetic (libm pow / exp /
// tgamma and the `int(x*res + 0.5)` rounding), and pack integer tables for LDS.
#include "host_tables.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <memory>

namespace pbsim {

namespace {

constexpr int kLineMax = 10240;  // BUF_SIZE, pbsim.cpp:20

void chomp(char *s) {
  size_t n = strlen(s);
  if (n && s[n - 1] == '\n') s[n - 1] = '\0';
}

// Running-total inverse-CDF expansion (the idiom of pbsim.cpp:3715-3735 and
// its 20-odd siblings): walk the items in order, skip per `skip(i)`, add p,
// end = min(int(total*res+0.5), res), give slots start..end to the item, stop
// when end reaches res.  `end` is in/out because the reference's `end_wk` is
// one variable shared by consecutive expansions: an expansion that skips every
// item reports the previous one's end.
template <class Prob, class Skip, class Fill>
void expand_cdf(long res, int first, int last, Prob prob, Skip skip, Fill fill, long *end) {
  long start = 1;
  double total = 0.0;
  for (int i = first; i <= last; i++) {
    if (skip(i)) continue;
    total += prob(i);
    long e = (long)(int)(total * (double)res + 0.5);
    if (e > res) e = res;
    for (long s = start; s <= e; s++) fill(s, i);
    *end = e;
    if (e >= res) break;
    start = e + 1;
  }
}

bool token_to_class(const char *tok, int *acc, std::string *err) {
  if (!tok) {
    *err = "malformed model line";
    return false;
  }
  *acc = atoi(tok);
  if (*acc < 0 || *acc > kAccMax) {
    *err = "model accuracy class out of range 0-100";
    return false;
  }
  return true;
}

}  // namespace

bool parse_errhmm(const char *path, ErrModel *m, std::string *err) {
  FILE *fp = fopen(path, "r");
  if (!fp) {
    *err = std::string("Cannot open file: ") + path;
    return false;
  }
  memset(m, 0, sizeof(*m));
  m->acc_min = 100;
  m->acc_max = 0;
  std::unique_ptr<char[]> line(new char[kLineMax]);
  bool ok = true;
  char *tok_save = nullptr;  // strtok_r: several ranks of one process parse their models at the same time
  while (ok && fgets(line.get(), kLineMax, fp)) {
    chomp(line.get());
    int acc;
    char *tok = strtok_r(line.get(), " ", &tok_save);
    if (!tok) continue;
    if (!(ok = token_to_class(tok, &acc, err))) break;
    m->exist[acc] = 1;
    m->acc_min = std::min(m->acc_min, acc);
    m->acc_max = std::max(m->acc_max, acc);
    const char *kind = strtok_r(NULL, " ", &tok_save);
    const char *st = strtok_r(NULL, " ", &tok_save);
    if (!kind || !st) {
      *err = "malformed model line";
      ok = false;
      break;
    }
    const int state = atoi(st);
    if (state < 0 || state > kStateMax) {
      *err = "model state exceeds STATE_MAX (50)";
      ok = false;
      break;
    }
    if (!strcmp(kind, "IP")) {
      const char *v = strtok_r(NULL, " ", &tok_save);
      m->ip[acc][state] = v ? atof(v) : 0.0;
      m->state_max[acc] = state;
    } else if (!strcmp(kind, "EP")) {
      int n = 0;
      for (char *v = strtok_r(NULL, " ", &tok_save); v; v = strtok_r(NULL, " ", &tok_save)) {
        if (n >= 4) {
          *err = "ERRHMM EP row has more than 4 columns";
          ok = false;
          break;
        }
        m->ep[acc][state][n++] = atof(v);
      }
    } else if (!strcmp(kind, "TP")) {
      int n = 0;
      for (char *v = strtok_r(NULL, " ", &tok_save); v; v = strtok_r(NULL, " ", &tok_save)) {
        if (++n > kStateMax) {
          *err = "model TP row exceeds STATE_MAX (50) columns";
          ok = false;
          break;
        }
        m->tp[acc][state][n] = atof(v);
      }
    }
  }
  fclose(fp);
  return ok;
}

bool parse_qshmm(const char *path, QsModel *m, std::string *err) {
  FILE *fp = fopen(path, "r");
  if (!fp) {
    *err = std::string("Cannot open file: ") + path;
    return false;
  }
  memset(m, 0, sizeof(*m));
  std::unique_ptr<char[]> line(new char[kLineMax]);
  bool ok = true;
  char *tok_save = nullptr;  // strtok_r: several ranks of one process parse their models at the same time
  // the reference's unchecked `qshmm.xx[accuracy][state][num] = atof(tp)` on the flat block (see QsModel)
  auto store = [&](long at, double v) {
    if (at < 0 || at >= QsModel::kIpN + QsModel::kEpN + QsModel::kTpN) {
      *err = "QSHMM model writes past qshmm.tp[] (the reference would overwrite exist_hmm[] here)";
      return false;
    }
    m->blk[at] = v;
    return true;
  };
  while (ok && fgets(line.get(), kLineMax, fp)) {
    chomp(line.get());
    int acc;
    char *tok = strtok_r(line.get(), " ", &tok_save);
    if (!tok) continue;
    if (!(ok = token_to_class(tok, &acc, err))) break;
    m->exist[acc] = 1;
    const char *kind = strtok_r(NULL, " ", &tok_save);
    const char *st = strtok_r(NULL, " ", &tok_save);
    if (!kind || !st) {
      *err = "malformed model line";
      ok = false;
      break;
    }
    const int state = atoi(st);
    if (state < 0) {
      *err = "negative model state";
      ok = false;
      break;
    }
    if (!strcmp(kind, "IP")) {
      const char *v = strtok_r(NULL, " ", &tok_save);
      ok = store(QsModel::ip_at(acc, state), v ? atof(v) : 0.0);
    } else if (!strcmp(kind, "EP")) {
      int n = 0;
      for (char *v = strtok_r(NULL, " ", &tok_save); v && ok; v = strtok_r(NULL, " ", &tok_save)) ok = store(QsModel::ep_at(acc, state, n++), atof(v));
    } else if (!strcmp(kind, "TP")) {
      int n = 0;
      for (char *v = strtok_r(NULL, " ", &tok_save); v && ok; v = strtok_r(NULL, " ", &tok_save)) ok = store(QsModel::tp_at(acc, state, ++n), atof(v));
    }
  }
  fclose(fp);
  return ok;
}

bool build_header_tables(const pbsim_params &p, HeaderTables *t, std::string *err) {
  // ---- read length: discretised gamma, resolution 1e5 (pbsim.cpp:3634-3662)
  t->prob2len.assign(100001, 0);
  if (p.len_sd == 0.0) {
    t->prob2len[1] = (int)(p.len_mean + 0.5);
    t->len_rv = 1;
  } else {
    const double variance = pow(p.len_sd, 2);
    const double kappa = pow(p.len_mean, 2) / variance;
    const double theta = variance / p.len_mean;
    const double gam = tgamma(kappa);
    long start = 1, end = 0;
    double total = 0.0;
    for (long i = p.len_min; i <= p.len_max; i++) {
      total += pow((double)i, kappa - 1) * exp((double)(-1 * i) / theta) / pow(theta, kappa) / gam;
      end = (long)(int)(total * 100000 + 0.5);
      if (end > 100000) end = 100000;
      for (long s = start; s <= end; s++) t->prob2len[s] = (int32_t)i;
      if (end >= 100000) break;
      start = end + 1;
    }
    t->len_rv = end;
  }
  if (t->len_rv < 1) {
    *err = "length parameters are not appropriate.";
    return false;
  }
  t->prob2len.resize(t->len_rv + 1);
  double sum = 0;
  for (long s = 1; s <= t->len_rv; s++) sum += t->prob2len[s];
  t->mean_len = sum / (double)t->len_rv;

  // ---- accuracy class: weight exp(0.22 a), resolution 1e5 (pbsim.cpp:3672-3706)
  const double mean = p.accuracy_mean * 100;
  t->acc_hi = (int)floor(mean * 1.05);
  t->acc_lo = (int)floor(mean * 0.75);
  if (t->acc_hi > 100) t->acc_hi = 100;
  double freq_total = 0.0;
  for (int a = t->acc_lo; a <= t->acc_hi; a++) freq_total += exp(0.22 * a);
  t->prob2acc.assign(100001, 0);
  long end = 0;
  expand_cdf(
      100000, t->acc_lo, t->acc_hi, [&](int a) { return exp(0.22 * a) / freq_total; }, [](int) { return false; },
      [&](long s, int a) { t->prob2acc[s] = (uint8_t)a; }, &end);
  t->acc_rv = end;
  if (t->acc_rv < 1) {
    *err = "accuracy parameters are not appropriate.";
    return false;
  }
  t->prob2acc.resize(t->acc_rv + 1);
  return true;
}

void build_ssp_tables(int rank_max, SspTables *t) {
  t->rank_max = rank_max;
  t->value.assign((size_t)(rank_max + 1) * 1001, 0);
  t->rv.assign(rank_max + 1, 0);
  for (int rank = 1; rank <= rank_max; rank++) {
    const double v = (double)1 / rank;
    double sum = 0;
    for (int j = 1; j <= 21; j++) sum += v / pow((double)j, 1 + v);
    long end = 0;
    expand_cdf(
        1000, 1, 21, [&](int j) { return (v / pow((double)j, 1 + v)) / sum; }, [](int) { return false; },
        [&](long s, int j) { t->value[(size_t)rank * 1001 + s] = (j - 1) * 5; }, &end);
    t->rv[rank] = (int32_t)end;
  }
}

void hp_bias_default(HpBias *b) {
  b->bias[0] = 0.0;
  for (int i = 1; i <= 10; i++) b->bias[i] = 1;
  b->bias[11] = 0.0;  // hp_del_bias[11] reads past the struct -> 0.0
}

void hp_bias_from_census(double hp_del_bias, const int64_t hpfreq[kHpSlots], HpBias *b) {
  // pbsim.cpp:686-696.  sum1 is a `long` there, so `sum1 += long*double`
  // truncates toward zero at each of the ten additions.
  long sum1 = 0, sum2 = 0;
  b->bias[0] = 0.0;
  b->bias[11] = 0.0;
  for (int i = 1; i <= 10; i++) {
    b->bias[i] = 1 + (hp_del_bias - 1) / 9 * (i - 1);
    sum1 = (long)((double)sum1 + (double)hpfreq[i] * b->bias[i]);
    sum2 += (long)hpfreq[i];
  }
  const double rate = (double)sum2 / (double)sum1;
  for (int i = 1; i <= 10; i++) b->bias[i] *= rate;
}

void hp_census_weighted(const uint8_t *seq, int64_t len, int64_t weight, int keep_first_case, int64_t hpfreq[kHpSlots]) {
  auto at = [&](int64_t i) -> int {
    int ch = seq[i];
    if (!(keep_first_case && i == 0) && ch >= 'a' && ch <= 'z') ch -= 32;
    return ch;
  };
  int64_t start = 0;
  for (int64_t i = 1; i <= len; i++) {
    if (i < len && at(i - 1) == at(i)) continue;
    const int64_t run = i - start;
    int v = (run <= 11) ? (int)run : ((run & 1) ? 11 : 10);  // nnum oscillation, pbsim.cpp:4396-4398
    if (at(i - 1) == 'N') v = 1;
    hpfreq[v] += weight * run;
    start = i;
  }
}

// x % d for x < 2^31, 2 <= d <= 65535 as  x - (mulhi(x, magic) >> shift) * d :
// shift = floor(log2(d - 1)), magic = ceil(2^(32 + shift) / d) < 2^32; the error term magic*d - 2^(32+shift) < d <= 2^(shift+1)
// times x < 2^31 stays below 2^(32+shift), so the quotient is exact.
void emission_magic(uint32_t d, uint32_t *magic, uint32_t *shift) {
  uint32_t s = 0;
  while ((2u << s) <= d - 1) s++;
  const unsigned __int128 num = (unsigned __int128)1 << (32 + s);
  *magic = (uint32_t)((num + d - 1) / d);
  *shift = s;
}

bool build_err_class_tables(const ErrModel &m, const HeaderTables &h, const HpBias &b, bool emis_skip_le,
                            ErrClassTables *t, std::string *err) {
  t->acc_lo = h.acc_lo;
  t->acc_hi = h.acc_hi;
  // states that can be reached: IP rows define state_max; TP columns may name more
  int smax = 1;
  for (int a = 0; a <= kAccMax; a++) {
    if (!m.exist[a]) continue;
    smax = std::max(smax, m.state_max[a]);
    for (int j = 1; j <= m.state_max[a]; j++)
      for (int k = 1; k <= kStateMax; k++)
        if (m.tp[a][j][k] != 0) smax = std::max(smax, k);
  }
  t->smax = smax;
  t->rows_off = 64;
  t->emis_off = t->rows_off + 32u * (uint32_t)(smax + 1);
  t->init_off = t->emis_off + 16u * (uint32_t)(smax + 1);
  t->tran_off = t->init_off + 1000u;  // contiguous: the initial-state table is row 0 of the transition table (k_walk_errhmm)
  t->stride = (t->tran_off + 1000u * (uint32_t)smax + 15u) & ~15u;
  const int ncls = h.acc_hi - h.acc_lo + 1;
  t->blob.assign((size_t)ncls * t->stride, 0);
  t->all_rv_1000 = true;

  // `end_wk` of pbsim.cpp:3615 lives across all rows and classes
  long end_wk = 0;
  // the reference builds tables only for classes acc_lo..acc_hi that have a
  // model (pbsim.cpp:3709-3713); out-of-range classes borrow acc_min/acc_max
  std::vector<uint8_t> built((size_t)(kAccMax + 1), 0);
  struct ClassRows {
    long init_rv;
    std::vector<uint8_t> init;                 // [1001]
    std::vector<long> tran_rv, emis_rv, del;   // [smax+1]
    std::vector<std::vector<uint8_t>> tran;    // [smax+1][1001]
    std::vector<std::vector<uint8_t>> emis;    // [smax+1][1001]
  };
  std::vector<ClassRows> rows(kAccMax + 1);
  for (int a = h.acc_lo; a <= h.acc_hi; a++) {
    if (!m.exist[a]) continue;
    ClassRows &r = rows[a];
    built[a] = 1;
    r.init.assign(1001, 0);
    r.tran_rv.assign(smax + 1, 0);
    r.emis_rv.assign(smax + 1, 0);
    r.del.assign(smax + 1, 0);
    r.tran.assign(smax + 1, std::vector<uint8_t>(1001, 0));
    r.emis.assign(smax + 1, std::vector<uint8_t>(1001, 0));
    expand_cdf(
        1000, 1, m.state_max[a], [&](int j) { return m.ip[a][j]; }, [&](int j) { return m.ip[a][j] == 0; },
        [&](long s, int j) { r.init[s] = (uint8_t)j; }, &end_wk);
    r.init_rv = end_wk;
    for (int j = 1; j <= m.state_max[a]; j++) {
      r.del[j] = (long)(int)(m.ep[a][j][3] * 1000 + 0.5);
      expand_cdf(
          1000, 0, 2, [&](int k) { return m.ep[a][j][k]; },
          [&](int k) { return emis_skip_le ? (m.ep[a][j][k] <= 0) : (m.ep[a][j][k] == 0); },
          [&](long s, int k) { r.emis[j][s] = (uint8_t)k; }, &end_wk);
      r.emis_rv[j] = end_wk;
    }
    for (int j = 1; j <= m.state_max[a]; j++) {
      expand_cdf(
          1000, 1, kStateMax, [&](int k) { return m.tp[a][j][k]; }, [&](int k) { return m.tp[a][j][k] == 0; },
          [&](long s, int k) { r.tran[j][s] = (uint8_t)k; }, &end_wk);
      r.tran_rv[j] = end_wk;
    }
  }

  for (int a = h.acc_lo; a <= h.acc_hi; a++) {
    uint8_t *dst = t->blob.data() + (size_t)(a - h.acc_lo) * t->stride;
    uint32_t *hdr = reinterpret_cast<uint32_t *>(dst);
    uint32_t mode, rate_mag = 0;
    int mc;
    if (a == 100) {
      mode = kModeVerbatim;  // pbsim.cpp:3837-3845, no draws
      mc = -1;
    } else if (m.exist[a]) {
      mode = kModeInRange;
      mc = a;
    } else if (a < m.acc_min) {
      mode = kModeBelow;  // pbsim.cpp:3829-3830, 3872-3899
      mc = m.acc_min;
      rate_mag = (uint32_t)(int)((double)(m.acc_min - a) / m.acc_min * 100);
    } else if (a > m.acc_max) {
      mode = kModeAbove;  // pbsim.cpp:3831-3832, 3900-3925
      mc = m.acc_max;
      rate_mag = (uint32_t)(int)((double)(a - m.acc_max) / (100 - m.acc_max) * 100);
    } else {
      // a hole inside [acc_min, acc_max]: the reference falls into the "above"
      // branch with whatever rate_mag the previous read left behind
      // (pbsim.cpp:3829-3833 assigns nothing) -- order dependent, not keyed.
      *err = "ERRHMM model has no table for an accuracy class inside its own range; unsupported";
      return false;
    }
    hdr[0] = (uint32_t)smax;
    hdr[2] = mode;
    hdr[3] = rate_mag;
    hdr[4] = (uint32_t)a;
    hdr[5] = (uint32_t)(mc < 0 ? 0 : mc);
    if (mc < 0) continue;
    if (!built[mc]) {
      *err = "ERRHMM model class needed by the accuracy range was not built";
      return false;
    }
    const ClassRows &r = rows[mc];
    hdr[1] = (uint32_t)r.init_rv;
    if (r.init_rv != 1000) t->all_rv_1000 = false;
    if (r.init_rv < 1) {
      *err = "ERRHMM initial-state table is empty";
      return false;
    }
    for (long s = 1; s <= 1000; s++) dst[t->init_off + (s - 1)] = r.init[s];
    {  // the highest state the class's tables can reach (k_walk_errhmm_coop walks every possible start state of a group)
      uint32_t reach = 0;
      for (long s = 1; s <= 1000; s++) reach = std::max<uint32_t>(reach, r.init[s]);
      for (int j = 1; j <= smax; j++)
        for (long s = 1; s <= 1000; s++) reach = std::max<uint32_t>(reach, r.tran[j][s]);
      hdr[6] = reach;
    }
    for (int j = 1; j <= smax; j++) {
      uint16_t *row = reinterpret_cast<uint16_t *>(dst + t->rows_off + 32u * (uint32_t)j);
      long e0 = 0, e1 = 0;
      for (long s = 1; s <= r.emis_rv[j]; s++) {
        e0 += (r.emis[j][s] == 0);
        e1 += (r.emis[j][s] <= 1);
      }
      row[0] = (uint16_t)r.tran_rv[j];
      row[1] = (uint16_t)r.emis_rv[j];
      row[2] = (uint16_t)e0;
      row[3] = (uint16_t)e1;
      {
        // Emission class without a division: e = (r >= E0') + (r >= E1') with r = z % d by multiply-high
        // (emission_magic()).  emis_rv == 0 (pbsim.cpp:3865: `rand() % 3`) is d = 3 with thresholds 1, 2;
        // emis_rv == 1 (the index is always 1) is a constant class, also carried by d = 3.
        uint32_t d = (uint32_t)r.emis_rv[j], t0 = (uint32_t)e0, t1 = (uint32_t)e1;
        if (d == 0) {
          d = 3;
          t0 = 1;
          t1 = 2;
        } else if (d == 1) {
          const int c = (1 > e0) + (1 > e1);
          d = 3;
          t0 = (c >= 1) ? 0 : 3;
          t1 = (c >= 2) ? 0 : 3;
        }
        uint32_t magic, shift;
        emission_magic(d, &magic, &shift);
        uint8_t *er = dst + t->emis_off + 16u * (uint32_t)j;
        // (the walks take the remainder as z + quo * (2^24 - d), low 24 bits: one 24-bit multiply-add)
        const uint32_t w1 = shift | ((0x1000000u - d) << 8), w2 = t0 | (t1 << 16);
        memcpy(er, &magic, 4);
        memcpy(er + 4, &w1, 4);
        memcpy(er + 8, &w2, 4);
      }
      for (int hp = 0; hp < kHpSlots; hp++) {
        // `index <= emis2del * bias[hp]` (pbsim.cpp:3862) with integral index
        // == `index <= floor(emis2del*bias[hp])`
        double v = floor((double)r.del[j] * b.bias[hp]);
        if (v > 65535.0) v = 65535.0;
        row[4 + hp] = (uint16_t)v;
      }
      {
        // the two thresholds the default --hp-del-bias 1 can reach (hp != 11 | hp == 11, Q1) ride in the emission row, so
        // that the walk gets everything a step needs of its state with one 16-byte LDS read
        const uint32_t w3 = (uint32_t)row[4 + 1] | ((uint32_t)row[4 + 11] << 16);
        memcpy(dst + t->emis_off + 16u * (uint32_t)j + 12, &w3, 4);
      }
      if (j <= m.state_max[mc]) {
        if (r.tran_rv[j] != 1000) t->all_rv_1000 = false;
      }
      for (long s = 1; s <= 1000; s++) dst[t->tran_off + 1000u * (uint32_t)(j - 1) + (s - 1)] = r.tran[j][s];
    }
  }
  return true;
}

// qc[].prob (pbsim.cpp:546-549) and set_mut's thresholds (:5474-5479), shared by the QSHMM and sampling walks
void build_mut_tables(const pbsim_params &p, const HpBias &b, QsClassTables *t) {
  for (int q = 0; q < kQcNum; q++) t->qprob[q] = pow(10, (double)q / -10);
  const long sum = (long)(p.sub_ratio + p.ins_ratio + p.del_ratio);
  const double sub_rate = (double)p.sub_ratio / sum, ins_rate = (double)p.ins_ratio / sum,
               del_rate = (double)p.del_ratio / sum;
  for (int q = 0; q < kQcNum; q++) {
    const double pr = t->qprob[q];
    t->sub_thre[q] = (uint32_t)(int)((pr * sub_rate) * 1000000 + 0.5);
    t->ins_thre[q] = (uint32_t)(int)((pr * (sub_rate + ins_rate)) * 1000000 + 0.5);
    const long del = (long)(int)((pr * del_rate) / (1 + pr * del_rate) * 1000000 + 0.5);
    for (int hp = 0; hp < kHpSlots; hp++) {
      // `rand_value < del_thre * bias[hp]` (pbsim.cpp:2272, 1821), rand_value integral
      // hp==0 (Q15) reads hp_del_bias[0] = the bits of hpfreq[11]: a positive
      // denormal once any hp==11 base has been counted, else 0.0
      double v;
      if (hp == 0) v = (b.hp11_seen && del > 0) ? 4.9406564584124654e-324 : 0.0;
      else v = (double)del * b.bias[hp];
      double c = ceil(v);
      if (c > 4294967295.0) c = 4294967295.0;
      t->del_thr[q][hp] = (uint32_t)c;
    }
  }
}

bool build_qs_class_tables(const QsModel &m, const HeaderTables &h, const HpBias &b, const pbsim_params &p,
                           QsClassTables *t, std::string *err) {
  (void)err;
  // ---- class-independent: qc[] (pbsim.cpp:546-549), uni_ep (:558-578), set_mut (:5474-5479)
  double uni[kAccMax + 1][kQcNum];
  for (int q = 0; q < kQcNum; q++) t->qprob[q] = pow(10, (double)q / -10);
  for (int a = 0; a <= kAccMax; a++) {
    for (int q = 0; q < kQcNum; q++) uni[a][q] = 0;
    if (a == kAccMax) {
      uni[a][93] = 1.0;
      continue;
    }
    const double prob = 1.0 - a / 100.0;
    for (int q = 0; q < kQcNum; q++) {
      if (prob == t->qprob[q]) {
        uni[a][q] = 1.0;
        break;
      } else if (prob > t->qprob[q]) {
        const double rate = (prob - t->qprob[q]) / (t->qprob[q - 1] - t->qprob[q]);
        uni[a][q - 1] = rate;
        uni[a][q] = 1 - rate;
        break;
      }
    }
  }
  build_mut_tables(p, b, t);

  // ---- per class
  t->acc_lo = h.acc_lo;
  t->acc_hi = h.acc_hi;
  int smax = 1;
  for (int a = 0; a <= kAccMax; a++) {
    if (!m.exist[a]) continue;
    for (int j = 1; j <= kStateMax; j++) {
      if (m.ip(a, j) != 0) smax = std::max(smax, j);
      for (int k = 1; k <= kStateMax; k++)
        if (m.tp(a, j, k) != 0) smax = std::max(smax, std::max(j, k));
      for (int k = 0; k < kQcNum; k++)
        if (m.ep(a, j, k) != 0) smax = std::max(smax, j);
    }
  }
  t->smax = smax;
  t->rv_off = 64;
  t->init_off = (t->rv_off + 4u * (uint32_t)(smax + 1) + 15u) & ~15u;
  t->tran_off = t->init_off + 100u;  // contiguous: the initial-state table is row 0 of the transition table (k_walk_qshmm)
  t->emis_off = t->tran_off + 100u * (uint32_t)smax;
  t->freq_off = (t->emis_off + 100u * (uint32_t)smax + 15u) & ~15u;
  t->stride = (t->freq_off + 1000u + 15u) & ~15u;
  const int ncls = h.acc_hi - h.acc_lo + 1;
  t->blob.assign((size_t)ncls * t->stride, 0);
  long end_wk = 0;  // shared like pbsim.cpp:1974
  t->all_rv_100 = true;
  for (int a = h.acc_lo; a <= h.acc_hi; a++) {
    uint8_t *dst = t->blob.data() + (size_t)(a - h.acc_lo) * t->stride;
    uint32_t *hdr = reinterpret_cast<uint32_t *>(dst);
    hdr[0] = (uint32_t)smax;
    hdr[4] = (uint32_t)a;
    if (m.exist[a] == 1) {
      hdr[2] = 1;
      expand_cdf(
          100, 1, kStateMax, [&](int j) { return m.ip(a, j); }, [&](int j) { return m.ip(a, j) == 0; },
          [&](long s, int j) { dst[t->init_off + (s - 1)] = (uint8_t)j; }, &end_wk);
      hdr[1] = (uint32_t)end_wk;
      uint16_t *rv = reinterpret_cast<uint16_t *>(dst + t->rv_off);
      std::vector<long> emis_rv(kStateMax + 1, 0), tran_rv(kStateMax + 1, 0);
      for (int j = 1; j <= kStateMax; j++) {
        expand_cdf(
            100, 0, kQcNum - 1, [&](int k) { return m.ep(a, j, k); }, [&](int k) { return m.ep(a, j, k) == 0; },
            [&](long s, int k) {
              if (j <= smax) dst[t->emis_off + 100u * (uint32_t)(j - 1) + (s - 1)] = (uint8_t)k;
            },
            &end_wk);
        emis_rv[j] = end_wk;
      }
      for (int j = 1; j <= kStateMax; j++) {
        expand_cdf(
            100, 1, kStateMax, [&](int k) { return m.tp(a, j, k); }, [&](int k) { return m.tp(a, j, k) == 0; },
            [&](long s, int k) {
              if (j <= smax) dst[t->tran_off + 100u * (uint32_t)(j - 1) + (s - 1)] = (uint8_t)k;
            },
            &end_wk);
        tran_rv[j] = end_wk;
      }
      if (hdr[1] != 100) t->all_rv_100 = false;
      // states that can be visited: named by an IP entry or a TP column of this class
      std::vector<char> reach(kStateMax + 1, 0);
      for (int j = 1; j <= kStateMax; j++) {
        if (m.ip(a, j) != 0) reach[j] = 1;
        for (int k = 1; k <= kStateMax; k++)
          if (m.tp(a, j, k) != 0) reach[k] = 1;
      }
      for (int j = 1; j <= smax; j++) {
        rv[2 * j] = (uint16_t)tran_rv[j];
        rv[2 * j + 1] = (uint16_t)emis_rv[j];
        if (reach[j] && (tran_rv[j] != 100 || emis_rv[j] != 100)) t->all_rv_100 = false;
      }
      {  // the highest state the class's expanded tables name (k_walk_qshmm_coop walks every possible start state of a group)
        uint32_t top = 0;
        for (uint32_t q = 0; q < 100u * (uint32_t)(smax + 1); q++) top = std::max<uint32_t>(top, dst[t->init_off + q]);
        hdr[6] = top;
      }
    } else {
      hdr[2] = 0;
      expand_cdf(
          1000, 0, kQcNum - 1, [&](int q) { return uni[a][q]; }, [&](int q) { return uni[a][q] == 0; },
          [&](long s, int q) { dst[t->freq_off + (s - 1)] = (uint8_t)q; }, &end_wk);
      hdr[3] = (uint32_t)end_wk;
    }
  }
  return true;
}

}  // namespace pbsim
