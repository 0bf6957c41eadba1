// units.cpp -- the trans and templ strategies behind the C ABI (simulate_by_*_trans pbsim.cpp:4428-4770 / 2738-3017,
// simulate_by_*_templ :5055-5362): all units resident as one buffer, a fixed read count per unit (no quota), reads numbered
// globally; the drivers and the unit-file loaders.  Split out of engine.cpp in round 5; the batch machinery stays there.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <functional>
#include <future>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "ctx.h"
#include "engine_internal.h"
#include "unit_io.h"
#include "philox.h"

extern "C" {

// Replaces get_transcript_inf (pbsim.cpp:1075-1136) + the streaming reader and
// per-transcript hp of simulate_by_*_trans (:4428-4485): all units are concatenated
// with '\n' separators (a byte no sequence contains, so homopolymer runs never
// join across units) and made resident once.
static int set_units(pbsim_ctx *c, int64_t n, const char *const *ids, const int64_t *plus_exp,
                     const int64_t *minus_exp, const uint8_t *const *seqs, const int64_t *lens, bool templ) {
  HIP_OK(hipSetDevice(c->device));
  // errhmm trans and both templ variants upper-case seq[1..len] only (pbsim.cpp:4457, 3332, 5063);
  // qshmm trans upper-cases seq[0..len-1] (:2778)
  const int keep_first = templ || c->p.method == PBSIM_METHOD_ERR;
  int64_t total = 0, reads = 0, max_len = 0;
  for (int64_t u = 0; u < n; u++) {
    if (lens[u] < 1 || lens[u] > 1000000) return fail("transcript length outside 1-1000000");
    total += lens[u] + 1;
    reads += (int64_t)(int)(plus_exp[u] + minus_exp[u]);  // `int read_num` (pbsim.cpp:4149)
    max_len = std::max(max_len, lens[u]);
  }
  // zero total expression is legal: the reference simply simulates nothing and prints its report (with NaN means)
  if (reads > 0x7fffff00LL) return fail("too many reads");
  // start-position buckets per rank (pbsim.cpp:4200-4224); unused by templ
  const int rank_max = (int)ceil((float)max_len / 1000);  // pbsim.cpp:1133
  SspTables st;
  build_ssp_tables(rank_max, &st);
  std::vector<uint8_t> ssp((size_t)(rank_max + 1) * 1000, 0);
  for (int k = 1; k <= rank_max; k++)
    for (int i = 1; i <= 1000; i++) ssp[(size_t)k * 1000 + (i - 1)] = (uint8_t)(st.value[(size_t)k * 1001 + i] / 5);
  // in simulate_by_errhmm_trans the verbatim copy of an accuracy-100 read, `for (i=0; i<mut.len; i++)`
  // (pbsim.cpp:4533), runs on the same `i` as the per-transcript read loop (:4487), which therefore continues at
  // i = mut.len + 1 behind such a read.  Which reads a transcript makes then depends on the header draws of the reads
  // before it -- a serial chain, but over header draws only (one Philox block and three table lookups per read, the
  // arithmetic of k_header_trans), so the host walks it once here and the kernels see an ordinary read -> unit map.
  const bool q5 = !templ && c->p.method == PBSIM_METHOD_ERR && c->hdr.acc_hi == 100;
  std::vector<uint8_t> cat((size_t)total);
  std::vector<int64_t> ubase(n), ulen(n);
  std::vector<int32_t> urank(n), offt((size_t)n * 21), runit;
  std::vector<int64_t> rbase;
  std::vector<uint8_t> rminus;
  runit.reserve((size_t)reads);
  rbase.reserve((size_t)reads);
  rminus.reserve((size_t)reads);
  std::vector<char> names((size_t)n * 132, 0);
  int64_t pos = 0;
  for (int64_t u = 0; u < n; u++) {
    memcpy(cat.data() + pos, seqs[u], (size_t)lens[u]);
    cat[pos + lens[u]] = '\n';
    ubase[u] = pos;
    ulen[u] = lens[u];
    urank[u] = (int32_t)ceil((double)lens[u] / 1000);  // pbsim.cpp:4494
    for (int k = 0; k < 21; k++) {                      // pbsim.cpp:4496-4501
      const double value = (k == 0) ? 0.0 : ((double)(k * 5) - 2.5) / 100;
      offt[(size_t)u * 21 + k] = (int32_t)(int)((double)lens[u] * value + 0.5);
    }
    strncpy(&names[(size_t)u * 132], ids[u], 128);
    const int64_t rn = (int64_t)(int)(plus_exp[u] + minus_exp[u]);
    for (int64_t i = 1; i <= rn; i++) {
      if (q5) {
        if ((int64_t)runit.size() >= 0x7fffff00LL) return fail("too many reads");
        const U4 w = header_block(c->p.seed, 0u, (uint32_t)(runit.size() + 1));
        int64_t L = c->hdr.prob2len[(size_t)(w.x % (uint32_t)c->hdr.len_rv) + 1];
        const int acc = c->hdr.prob2acc[(size_t)(w.y % (uint32_t)c->hdr.acc_rv) + 1];
        const uint32_t rv = (uint32_t)st.rv[(size_t)urank[u]];
        const int64_t off = offt[(size_t)u * 21 + ssp[(size_t)urank[u] * 1000 + w.z % (rv ? rv : 1u)]];
        if (off + L > lens[u]) L = lens[u] - off;
        runit.push_back((int32_t)u);
        rbase.push_back(pos);
        rminus.push_back((i > plus_exp[u]) ? 1 : 0);
        if (acc == 100) i = std::max<int64_t>(L, 0);  // the clobbered counter; the loop's i++ follows
        continue;
      }
      runit.push_back((int32_t)u);
      rbase.push_back(pos);
      rminus.push_back((i > plus_exp[u]) ? 1 : 0);  // pbsim.cpp:4516-4522
    }
    pos += lens[u] + 1;
  }
  reads = (int64_t)runit.size();

  // hp-del-bias census weighted by expression (pbsim.cpp:4352-4426)
  hp_bias_default(&c->bias);
  c->bias.hp11_seen = false;
  if (c->p.hp_del_bias != 1) {
    int64_t freq[kHpSlots] = {0};
    for (int64_t u = 0; u < n; u++)
      hp_census_weighted(seqs[u], lens[u], (int64_t)(int)(plus_exp[u] + minus_exp[u]), keep_first, freq);
    hp_bias_from_census(c->p.hp_del_bias, freq, &c->bias);
    c->bias.hp11_seen = freq[11] > 0;  // hpfreq[11] aliases hp_del_bias[0] (Q15)
  }
  c->class_tables_dirty = true;
  c->coop_wg_errhmm[0] = c->coop_wg_errhmm[1] = 0;

  HIP_OK(c->d_seq_own.ensure((size_t)total + 64));
  HIP_OK(hipMemcpyAsync(c->d_seq_own.p, cat.data(), (size_t)total, hipMemcpyHostToDevice, c->stream));
  HIP_OK(hipMemsetAsync(c->d_seq_own.as<uint8_t>() + total, 0, 64, c->stream));
  int64_t census[kHpSlots] = {0};
  if (!prepare_reference(c, c->d_seq_own.as<uint8_t>(), total, keep_first, census)) return PBSIM_FAILED;
  if (reads > 0) {
    if (!upload(c->d_read_unit, runit.data(), runit.size() * 4, c->stream)) return PBSIM_FAILED;
    if (!upload(c->d_read_base, rbase.data(), rbase.size() * 8, c->stream)) return PBSIM_FAILED;
    if (!upload(c->d_read_minus, rminus.data(), rminus.size(), c->stream)) return PBSIM_FAILED;
  }
  if (!upload(c->d_unit_len, ulen.data(), ulen.size() * 8, c->stream)) return PBSIM_FAILED;
  if (!upload(c->d_unit_rank, urank.data(), urank.size() * 4, c->stream)) return PBSIM_FAILED;
  if (!upload(c->d_unit_names, names.data(), names.size(), c->stream)) return PBSIM_FAILED;
  if (!upload(c->d_off_table, offt.data(), offt.size() * 4, c->stream)) return PBSIM_FAILED;
  if (!upload(c->d_ssp, ssp.data(), ssp.size(), c->stream)) return PBSIM_FAILED;
  if (!upload(c->d_ssp_rv, st.rv.data(), st.rv.size() * 4, c->stream)) return PBSIM_FAILED;
  HIP_OK(hipStreamSynchronize(c->stream));
  c->d_seq = c->d_seq_own.as<uint8_t>();
  c->ref_len = total;  // only sizes the batches; record lengths come from unit_len
  c->unit = 0;
  c->n_units = n;
  c->trans_reads = reads;
  for (Slot &sl : c->slots) sl.b_enqueued = sl.b_walked = sl.b_finalized = false;
  return PBSIM_SUCCEEDED;
}

int pbsim_set_transcripts(pbsim_ctx *c, int64_t n, const char *const *ids, const int64_t *plus_exp,
                          const int64_t *minus_exp, const uint8_t *const *seqs, const int64_t *lens) {
  if (!c || n < 1 || !ids || !plus_exp || !minus_exp || !seqs || !lens) return fail("pbsim_set_transcripts: bad argument");
  NEED_DEVICE(c);
  if (c->p.strategy != PBSIM_STRATEGY_TRANS) return fail("pbsim_set_transcripts: strategy is not trans");
  return set_units(c, n, ids, plus_exp, minus_exp, seqs, lens, false);
}

// get_templ_inf (pbsim.cpp:1366-1418) + the per-template loop of simulate_by_*_templ (:5055-5103):
// every template is one unit with exactly one '+' read over its whole length
int pbsim_set_templates(pbsim_ctx *c, int64_t n, const char *const *ids, const uint8_t *const *seqs,
                        const int64_t *lens) {
  if (!c || n < 1 || !ids || !seqs || !lens) return fail("pbsim_set_templates: bad argument");
  NEED_DEVICE(c);
  if (c->p.strategy != PBSIM_STRATEGY_TEMPL) return fail("pbsim_set_templates: strategy is not templ");
  std::vector<int64_t> one((size_t)n, 1), zero((size_t)n, 0);
  return set_units(c, n, ids, one.data(), zero.data(), seqs, lens, true);
}

int pbsim_simulate_templ(pbsim_ctx *c, const pbsim_sink *sink) { return pbsim_simulate_trans(c, sink); }

// simulate_by_errhmm_trans / simulate_by_qshmm_trans (pbsim.cpp:4428-4770, 2738-3017): fixed read
// count per transcript, no quota; reads are numbered globally like sim.res_num.  The templ
// strategy (simulate_by_*_templ) runs through the same driver.
int pbsim_simulate_trans(pbsim_ctx *c, const pbsim_sink *sink) {
  if (!c) return fail("bad argument");
  return pbsim_simulate_units_range(c, 1, c->trans_reads, sink);
}

int64_t pbsim_unit_reads(pbsim_ctx *c) { return c ? c->trans_reads : -1; }

static int load_unit_file(pbsim_ctx *c, const char *path, int64_t stats[2], bool templ) {
  if (!c || !path) return fail("bad argument");
  std::vector<Transcript> tr;
  std::string err;
  long a = 0;
  long long b = 0;
  if (templ ? !read_templates(path, &tr, &a, &b, &err) : !read_transcripts(path, &tr, &a, &err)) return fail(err);
  std::vector<const char *> ids;
  std::vector<int64_t> plus, minus, lens;
  std::vector<const uint8_t *> seqs;
  for (auto &t : tr) {
    ids.push_back(t.id.c_str());
    plus.push_back(t.plus);
    minus.push_back(t.minus);
    seqs.push_back((const uint8_t *)t.seq.data());
    lens.push_back((int64_t)strlen(t.seq.c_str()));
  }
  if (stats) {
    stats[0] = templ ? (int64_t)a : (int64_t)tr.size();
    stats[1] = templ ? (int64_t)b : (int64_t)a;
  }
  return templ ? pbsim_set_templates(c, (int64_t)tr.size(), ids.data(), seqs.data(), lens.data())
               : pbsim_set_transcripts(c, (int64_t)tr.size(), ids.data(), plus.data(), minus.data(), seqs.data(), lens.data());
}
int pbsim_load_transcript_file(pbsim_ctx *c, const char *path, int64_t stats[2]) { return load_unit_file(c, path, stats, false); }
int pbsim_load_template_file(pbsim_ctx *c, const char *path, int64_t stats[2]) { return load_unit_file(c, path, stats, true); }

int pbsim_simulate_units_range(pbsim_ctx *c, int64_t first_read, int64_t n_reads, const pbsim_sink *sink) {
  if (!c) return fail("bad argument");
  NEED_DEVICE(c);
  if (c->p.strategy == PBSIM_STRATEGY_WGS) return fail("pbsim_simulate_trans: strategy is wgs");
  if (!c->d_seq || c->n_units < 1) return fail("no transcripts/templates set");
  if (first_read < 1 || n_reads < 0 || first_read - 1 + n_reads > c->trans_reads)
    return fail("pbsim_simulate_units_range: reads outside 1 .. pbsim_unit_reads()");
  pbsim_reset_stats(c);
  // No quota here: every read is final, so the batches simply pipeline over the slots (the walk of one beside the text
  // emission of the other): one batch per slot when everything fits (small batches waste the GPU on their tails).
  // The text emission is not waited for (as in the job pipeline): a batch's statistics are added on the host while its text is
  // being written and the next batch's emission queues up behind it -- the GPU sat idle for 5 of the 39 ms of configs[3]
  // while the host added 1 M reads' counters.  deliver() waits for the emission before it hands text to a sink.
  DeferTextSync defer_guard(c);
  // (a delivering job keeps three slots busy: one whose bytes are leaving, one finalized behind it, one walking)
  const bool delivering = sink && c->deflate != 0 && (sink->on_read_text || sink->on_maf_text);
  const int n_slots = std::max(1, std::min(kMaxSlots, delivering ? std::max(3, c->pipeline_depth) : c->pipeline_depth));
  const int64_t R = first_read - 1 + n_reads;  // last read of the range
  int64_t cap = batch_capacity(c);
  struct Pending {
    int slot;
    int64_t first, n;
  };
  std::vector<Pending> fifo;
  auto drop_pending = [&]() {
    for (const Pending &pd : fifo) {
      c->cur = pd.slot;
      (void)hipStreamSynchronize(c->s().stream);
      c->s().b_enqueued = false;
    }
    fifo.clear();
    c->cur = 0;
  };
  // A sink that receives its bytes through the GPU's compression (pbsim_set_deflate) is bound by the link, not by the walks
  // (configs[3]: 37 ms of walks, 4.3 GB of members): a batch's text is emitted (pbsim_batch_finalize) and the first piece of its
  // two streams compressed BEFORE the batch in front of it is delivered, so the link does not wait for the emission kernels nor
  // for a call's first deflate kernels between two batches; and the batch's statistics are added on a thread beside that
  // delivery instead of behind it.  The sink still receives the batches in read order.  Round 6, same box
  // (tools/closed_ab/units_ab*.sh, profiles/r06_units_delivery_ab.txt): 121 -> 112 -> 102.5 ms per job.  MORE batches do not
  // help (every batch's walk lasts as long as its longest read: twelve batches walk at 40 instead of 120 Gbases/s), nor does a
  // ramp of batch sizes, nor walking the batches one after the other on the GPU (105 -> 108 ms).
  const char *up = exp_env("PBSIM_UNITS_PARTS");  // experiment knobs: batches per job; prelaunch of a batch's first deflate piece
  const char *upre = exp_env("PBSIM_UNITS_PRELAUNCH");
  const int n_parts = up && atoi(up) > 0 ? atoi(up) : n_slots;
  const bool prelaunch = delivering && (c->deflate & 3) == 3 && (upre ? atoi(upre) != 0 : true);
  int64_t next_begin = first_read, next_read = first_read;
  int next_slot = 0;
  bool have_prev = false;  // a batch that is finalized and not yet delivered (it keeps its slot)
  int prev_slot = 0;
  // The statistics of a batch (two million tasks per job: 10-15 ms of host loop, pbsim.cpp:3986-4005 in read order) are added on a
  // thread of their own beside the delivery of the batch in front, not behind every delivery with the link idle; batch after
  // batch, so the order-dependent accuracy sum keeps its order.
  struct DeferAccount {
    pbsim_ctx *c;
    bool was;
    DeferAccount(pbsim_ctx *cc, bool on) : c(cc), was(cc->defer_account) { c->defer_account = on; }
    ~DeferAccount() { c->defer_account = was; }
  } account_guard(c, delivering);
  std::future<int> acct;
  auto join_acct = [&]() -> int {
    if (!acct.valid()) return PBSIM_SUCCEEDED;
    return acct.get() ? PBSIM_SUCCEEDED : fail("the statistics of a batch could not be added");
  };
  auto deliver_prev = [&]() -> int {
    if (!have_prev) return PBSIM_SUCCEEDED;
    have_prev = false;
    c->cur = prev_slot;
    return deliver(c, sink);
  };
  auto give_up = [&]() {
    if (acct.valid()) (void)acct.get();
    drop_pending();
    if (have_prev) {
      c->cur = prev_slot;
      (void)hipStreamSynchronize(c->s().stream);
      have_prev = false;
      c->cur = 0;
    }
  };
  while (next_read <= R) {
    while ((int)fifo.size() + (have_prev ? 1 : 0) < n_slots && next_begin <= R) {
      const int64_t part = std::max<int64_t>(65536, (n_reads + n_parts - 1) / n_parts);
      const int64_t n = std::min(std::min(cap, part), R - next_begin + 1);
      c->cur = next_slot;
      if (!pbsim_batch_walk_begin(c, next_begin, n, -1)) {
        give_up();
        return PBSIM_FAILED;
      }
      fifo.push_back(Pending{next_slot, next_begin, n});
      next_slot = (next_slot + 1) % n_slots;
      next_begin += n;
    }
    if (fifo.empty()) {  // (one slot: the finalized batch must leave before the next one can begin)
      if (!deliver_prev()) {
        give_up();
        return PBSIM_FAILED;
      }
      continue;
    }
    const Pending pd = fifo.front();
    fifo.erase(fifo.begin());
    c->cur = pd.slot;
    if (!pbsim_batch_walk_end(c, nullptr)) {
      const bool budget = g_err.rfind("scratch budget exceeded", 0) == 0 && pd.n > 1;
      const std::string keep = g_err;
      drop_pending();
      if (!budget) {
        give_up();
        g_err = keep;
        return PBSIM_FAILED;
      }
      cap = std::max<int64_t>(1, pd.n / 2);  // retry from this batch with smaller ones
      next_begin = pd.first;
      next_slot = have_prev ? (prev_slot + 1) % n_slots : 0;
      continue;
    }
    pbsim_batch_info bi;
    if (!pbsim_batch_finalize(c, 0, &bi)) {
      give_up();
      return PBSIM_FAILED;
    }
    next_read += bi.n_final;
    // the table fit and the first piece of this batch's two streams go onto the slot's stream right behind its text emission:
    // when the batch's turn comes its first copies start at once instead of waiting for their kernels (deflate_host.cpp df_begin)
    if (prelaunch && !deflate_prelaunch(c, c->s(), sink->on_read_text != nullptr, sink->on_maf_text != nullptr, true)) {
      give_up();
      return PBSIM_FAILED;
    }
    if (delivering) {
      if (!join_acct()) {
        give_up();
        return PBSIM_FAILED;
      }
      Slot *sl = &c->s();
      acct = std::async(std::launch::async, [c, sl]() { return account_of(c, *sl, &c->st); });
    }
    if (!deliver_prev()) {  // the batch in front: its bytes leave while this batch's text is being emitted
      give_up();
      return PBSIM_FAILED;
    }
    have_prev = true;
    prev_slot = pd.slot;
  }
  if (!deliver_prev() || !join_acct()) {
    give_up();
    return PBSIM_FAILED;
  }
  return PBSIM_SUCCEEDED;
}

}  // extern "C"
