// stats.cpp -- the statistics block of a unit (pbsim.cpp:3986-4005 per task, :4082-4105 + :5541-5562 at the end) and its
// merge across the ranks of a multi-GPU job.  Host arithmetic only, no device: a tables-only context (device -1) can
// run all of it, which is how the CPU tests cover the N > 1 reduction path (tests/test_multi_gloo.py).
//
// The reference accumulates `accuracy_total += value` task by task (pbsim.cpp:4003, 2313), an order-dependent double
// sum.  Reads are sharded over ranks by blocks of the read numbering, so a rank keeps the per-task values of its blocks
// and the merge folds the blocks of all ranks in read order: the merged mean equals the single-GPU one bit for bit.
#include <math.h>
#include <string.h>

#include "ctx.h"

namespace pbsim {

void stats_add_task(StatsAcc *st, int64_t len_max, bool quality, long len, long nsub, long nins, long ndel, double qsum,
                    std::vector<double> *values) {
  if (st->freq_len.empty()) st->freq_len.assign((size_t)len_max * 2 + 2, 0);
  if (st->freq_acc.empty()) st->freq_acc.assign(100001, 0);
  st->res_len_total += len;
  if ((size_t)len < st->freq_len.size()) st->freq_len[(size_t)len]++;
  if (len > st->res_len_max) st->res_len_max = len;
  if (len < st->res_len_min) st->res_len_min = len;
  st->res_sub += nsub;
  st->res_ins += nins;
  st->res_del += ndel;
  double value;
  if (quality) value = 1.0 - (qsum / len);                               // pbsim.cpp:2309-2313
  else value = 1.0 - ((double)(nsub + nins + ndel) / len);               // pbsim.cpp:4002
  st->accuracy_total += value;
  if (values) values->push_back(value);
  const double w = value * 100000 + 0.5;                                 // pbsim.cpp:4004-4005
  if (w >= 0 && w < 100001) st->freq_acc[(size_t)(int)w]++;
}

void stats_finish(const StatsAcc &st, const pbsim_params &p, int64_t ref_len, pbsim_stats *o) {
  memset(o, 0, sizeof *o);
  o->res_num = st.res_num;
  o->res_pass_num = st.res_num * p.pass_num;
  o->res_len_total = st.res_len_total;
  o->res_len_min = st.res_len_min;
  o->res_len_max = st.res_len_max;
  o->res_sub_num = st.res_sub;
  o->res_ins_num = st.res_ins;
  o->res_del_num = st.res_del;
  o->res_len_mean = (double)st.res_len_total / o->res_pass_num;
  o->res_accuracy_mean = st.accuracy_total / o->res_pass_num;
  if (o->res_pass_num == 1) {
    o->res_len_sd = 0.0;
    o->res_accuracy_sd = 0.0;
  } else {
    double variance = 0.0;
    const long lmax = std::min<long>((long)p.len_max, (long)st.freq_len.size() - 1);
    for (long i = 0; i <= lmax; i++)
      if (st.freq_len[(size_t)i] > 0) variance += pow((o->res_len_mean - i), 2) * st.freq_len[(size_t)i];
    o->res_len_sd = sqrt(variance / o->res_pass_num);
    variance = 0.0;
    for (long i = 0; i <= 100000 && (size_t)i < st.freq_acc.size(); i++)
      if (st.freq_acc[(size_t)i] > 0) variance += pow((o->res_accuracy_mean - i * 0.00001), 2) * st.freq_acc[(size_t)i];
    o->res_accuracy_sd = sqrt(variance / o->res_pass_num);
  }
  if (ref_len > 0) o->res_depth = (double)st.res_len_total / ref_len / p.pass_num;
  o->res_sub_rate = (double)st.res_sub / st.res_len_total;
  o->res_ins_rate = (double)st.res_ins / st.res_len_total;
  o->res_del_rate = (double)st.res_del / st.res_len_total;
}

#define COMM_OK(expr)                                                                        \
  do {                                                                                       \
    if (!(expr)) return fail(std::string("pbsim_comm callback failed: " #expr));             \
  } while (0)

int stats_merge(StatsAcc *st, const pbsim_params &p, const pbsim_comm *comm, int64_t *extra, int n_extra) {
  if (!comm || comm->world <= 1) return PBSIM_SUCCEEDED;
  if (!comm->all_gather_i64 || !comm->all_reduce_i64) return fail("pbsim_comm: all_gather_i64 and all_reduce_i64 must be set");
  const int W = comm->world;
  if (st->freq_len.empty()) st->freq_len.assign((size_t)p.len_max * 2 + 2, 0);
  if (st->freq_acc.empty()) st->freq_acc.assign(100001, 0);
  // ---- C2: counters (sum), extremes (min / max)
  std::vector<int64_t> sums = {st->res_num, st->res_len_total, st->res_sub, st->res_ins, st->res_del};
  for (int i = 0; i < n_extra; i++) sums.push_back(extra[i]);
  COMM_OK(comm->all_reduce_i64(comm->user, sums.data(), (int64_t)sums.size(), PBSIM_OP_SUM));
  int64_t mn = st->res_len_min, mx = st->res_len_max;
  COMM_OK(comm->all_reduce_i64(comm->user, &mn, 1, PBSIM_OP_MIN));
  COMM_OK(comm->all_reduce_i64(comm->user, &mx, 1, PBSIM_OP_MAX));
  st->res_num = sums[0];
  st->res_len_total = sums[1];
  st->res_sub = sums[2];
  st->res_ins = sums[3];
  st->res_del = sums[4];
  for (int i = 0; i < n_extra; i++) extra[i] = sums[(size_t)5 + i];
  st->res_len_min = mn;
  st->res_len_max = mx;
  // ---- C2: the two histograms (pbsim.cpp:195-196); freq_len only as far as any rank has counted
  const int64_t nlen = std::min<int64_t>((int64_t)st->freq_len.size(), mx + 1);
  if (nlen > 0) COMM_OK(comm->all_reduce_i64(comm->user, st->freq_len.data(), nlen, PBSIM_OP_SUM));
  COMM_OK(comm->all_reduce_i64(comm->user, st->freq_acc.data(), (int64_t)st->freq_acc.size(), PBSIM_OP_SUM));
  // ---- accuracy_total in read order: gather every rank's blocks and fold them by first task
  int64_t mine[2] = {(int64_t)st->blocks.size(), 0};
  for (const StatsAcc::Block &b : st->blocks) mine[1] += (int64_t)b.values.size();
  std::vector<int64_t> meta((size_t)W * 2);
  COMM_OK(comm->all_gather_i64(comm->user, mine, 2, meta.data()));
  int64_t max_blocks = 0, max_values = 0;
  for (int r = 0; r < W; r++) {
    max_blocks = std::max(max_blocks, meta[(size_t)r * 2]);
    max_values = std::max(max_values, meta[(size_t)r * 2 + 1]);
  }
  if (max_blocks > 0) {
    std::vector<int64_t> desc((size_t)max_blocks * 2, 0), all_desc((size_t)W * max_blocks * 2);
    for (size_t i = 0; i < st->blocks.size(); i++) {
      desc[i * 2] = st->blocks[i].first_task;
      desc[i * 2 + 1] = (int64_t)st->blocks[i].values.size();
    }
    COMM_OK(comm->all_gather_i64(comm->user, desc.data(), max_blocks * 2, all_desc.data()));
    static_assert(sizeof(double) == sizeof(int64_t), "values travel as their bit patterns");
    // (8 B per task of the record on every rank -- 40 MB for a 750 Mbp record at depth 60: not value-initialised, the
    // collective writes all of it, and what lies behind a rank's own count is never read)
    const size_t n_vals = (size_t)std::max<int64_t>(max_values, 1);
    std::unique_ptr<int64_t[]> vals(new int64_t[n_vals]), all_vals(new int64_t[(size_t)W * n_vals]);
    size_t at = 0;
    for (const StatsAcc::Block &b : st->blocks) {
      if (!b.values.empty()) memcpy(&vals[at], b.values.data(), b.values.size() * 8);
      at += b.values.size();
    }
    COMM_OK(comm->all_gather_i64(comm->user, vals.get(), (int64_t)n_vals, all_vals.get()));
    struct Piece {
      int64_t first, n;
      const int64_t *v;
    };
    std::vector<Piece> pieces;
    for (int r = 0; r < W; r++) {
      const int64_t *v = &all_vals[(size_t)r * n_vals];
      for (int64_t i = 0; i < meta[(size_t)r * 2]; i++) {
        const int64_t first = all_desc[((size_t)r * max_blocks + i) * 2], n = all_desc[((size_t)r * max_blocks + i) * 2 + 1];
        pieces.push_back(Piece{first, n, v});
        v += n;
      }
    }
    std::stable_sort(pieces.begin(), pieces.end(), [](const Piece &a, const Piece &b) { return a.first < b.first; });
    double total = 0.0;
    for (const Piece &pc : pieces)
      for (int64_t i = 0; i < pc.n; i++) {
        double v;
        memcpy(&v, &pc.v[i], 8);
        total += v;  // pbsim.cpp:4003, in read order
      }
    st->accuracy_total = total;
  }
  st->blocks.clear();
  return PBSIM_SUCCEEDED;
}

}  // namespace pbsim

extern "C" {

// Statistics primitives for callers that shard a unit over several contexts themselves (pbsim_simulate_units_range per rank,
// pbsim3_amd/run_multi.py): keep the per-task accuracy values while accounting, then merge.
int pbsim_stats_keep_values(pbsim_ctx *c, int on) {
  if (!c) return fail("bad argument");
  c->st.keep_values = on != 0;
  return PBSIM_SUCCEEDED;
}

int pbsim_stats_merge(pbsim_ctx *c, const pbsim_comm *comm) {
  if (!c) return fail("bad argument");
  return stats_merge(&c->st, c->p, comm, nullptr, 0);
}

// Accounts n finished tasks given as plain arrays (what the kernels leave per task: pbsim.cpp:3986-4005); first_task is the
// 0-based global task index of element 0.  The engine's own batches go through pbsim_batch_account; this entry exists for
// device-free callers (tests of the multi-rank merge).
int pbsim_stats_add_tasks(pbsim_ctx *c, int64_t first_task, int64_t n, const int32_t *out_len, const int32_t *nsub,
                          const int32_t *nins, const int32_t *ndel, const double *qsum) {
  if (!c || n < 0 || !out_len || !nsub || !nins || !ndel) return fail("pbsim_stats_add_tasks: bad argument");
  const bool quality = c->p.method == PBSIM_METHOD_QS || c->p.method == PBSIM_METHOD_SAMPLE;
  if (quality && !qsum) return fail("pbsim_stats_add_tasks: qsum is needed for the quality-score methods");
  if (first_task % c->p.pass_num != 0 || n % c->p.pass_num != 0) return fail("pbsim_stats_add_tasks: whole reads only");
  std::vector<double> *values = nullptr;
  if (c->st.keep_values) {
    c->st.blocks.emplace_back();
    c->st.blocks.back().first_task = first_task;
    values = &c->st.blocks.back().values;
  }
  c->st.res_num += n / c->p.pass_num;
  for (int64_t t = 0; t < n; t++)
    stats_add_task(&c->st, c->p.len_max, quality, out_len[t], nsub[t], nins[t], ndel[t], quality ? qsum[t] : 0.0, values);
  return PBSIM_SUCCEEDED;
}

}  // extern "C"
