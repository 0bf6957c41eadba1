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
  // Three collectives per record (round 5; eight until then -- every one of them waits for the slowest rank):
  // ---- C2 (1): counters, extremes and the shape of the rank's accuracy blocks in ONE all-gather, reduced here
  int64_t mine_blocks = (int64_t)st->blocks.size(), mine_values = 0;
  for (const StatsAcc::Block &b : st->blocks) mine_values += (int64_t)b.values.size();
  std::vector<int64_t> head = {st->res_num, st->res_len_total, st->res_sub, st->res_ins, st->res_del};
  for (int i = 0; i < n_extra; i++) head.push_back(extra[i]);
  const size_t n_sum = head.size();
  head.push_back(st->res_len_min);
  head.push_back(st->res_len_max);
  head.push_back(mine_blocks);
  head.push_back(mine_values);
  const size_t hw = head.size();
  std::vector<int64_t> heads((size_t)W * hw);
  COMM_OK(comm->all_gather_i64(comm->user, head.data(), (int64_t)hw, heads.data()));
  std::vector<int64_t> sums(n_sum, 0);
  int64_t mn = st->res_len_min, mx = st->res_len_max, max_blocks = 0, max_values = 0;
  std::vector<int64_t> meta((size_t)W * 2);
  for (int r = 0; r < W; r++) {
    const int64_t *h = &heads[(size_t)r * hw];
    for (size_t i = 0; i < n_sum; i++) sums[i] += h[i];
    mn = std::min(mn, h[n_sum]);
    mx = std::max(mx, h[n_sum + 1]);
    meta[(size_t)r * 2] = h[n_sum + 2];
    meta[(size_t)r * 2 + 1] = h[n_sum + 3];
    max_blocks = std::max(max_blocks, h[n_sum + 2]);
    max_values = std::max(max_values, h[n_sum + 3]);
  }
  st->res_num = sums[0];
  st->res_len_total = sums[1];
  st->res_sub = sums[2];
  st->res_ins = sums[3];
  st->res_del = sums[4];
  for (int i = 0; i < n_extra; i++) extra[i] = sums[(size_t)5 + i];
  st->res_len_min = mn;
  st->res_len_max = mx;
  // ---- C2 (2): the two histograms (pbsim.cpp:195-196) in ONE all-reduce; freq_len only as far as any rank has counted
  {
    const int64_t nlen = std::max<int64_t>(0, std::min<int64_t>((int64_t)st->freq_len.size(), mx + 1));
    const size_t nacc = st->freq_acc.size();
    std::unique_ptr<int64_t[]> hist(new int64_t[(size_t)nlen + nacc]);
    if (nlen > 0) memcpy(hist.get(), st->freq_len.data(), (size_t)nlen * 8);
    memcpy(hist.get() + nlen, st->freq_acc.data(), nacc * 8);
    COMM_OK(comm->all_reduce_i64(comm->user, hist.get(), nlen + (int64_t)nacc, PBSIM_OP_SUM));
    if (nlen > 0) memcpy(st->freq_len.data(), hist.get(), (size_t)nlen * 8);
    memcpy(st->freq_acc.data(), hist.get() + nlen, nacc * 8);
  }
  // ---- C2 (3): accuracy_total in read order: every rank's blocks (first task, size) and values in ONE all-gather, folded by
  // first task
  if (max_blocks > 0) {
    static_assert(sizeof(double) == sizeof(int64_t), "values travel as their bit patterns");
    // (8 B per task of the record on every rank -- 40 MB for a 750 Mbp record at depth 60: not value-initialised, the
    // collective writes all of it, and what lies behind a rank's own count is never read)
    const size_t n_desc = (size_t)max_blocks * 2, n_vals = (size_t)std::max<int64_t>(max_values, 1), n_msg = n_desc + n_vals;
    std::unique_ptr<int64_t[]> msg(new int64_t[n_msg]), all(new int64_t[(size_t)W * n_msg]);
    memset(msg.get(), 0, n_desc * 8);
    size_t at = n_desc;
    for (size_t i = 0; i < st->blocks.size(); i++) {
      const StatsAcc::Block &b = st->blocks[i];
      msg[i * 2] = b.first_task;
      msg[i * 2 + 1] = (int64_t)b.values.size();
      if (!b.values.empty()) memcpy(&msg[at], b.values.data(), b.values.size() * 8);
      at += b.values.size();
    }
    COMM_OK(comm->all_gather_i64(comm->user, msg.get(), (int64_t)n_msg, all.get()));
    struct Piece {
      int64_t first, n;
      const int64_t *v;
    };
    std::vector<Piece> pieces;
    for (int r = 0; r < W; r++) {
      const int64_t *d = &all[(size_t)r * n_msg], *v = d + n_desc;
      for (int64_t i = 0; i < meta[(size_t)r * 2]; i++) {
        const int64_t first = d[(size_t)i * 2], n = d[(size_t)i * 2 + 1];
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
