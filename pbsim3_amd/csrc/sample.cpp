// sample.cpp -- the sampling method (pbsim.cpp:1155-1330 get_sample_inf's product, :1694-1949 simulate_by_sample) behind the
// C ABI: the filtered profile in HBM, the chunk planner and launcher, the one-GPU driver and the one sharded over ranks by
// string blocks.  Split out of engine.cpp in round 5; the batch machinery it drives (walk kernels' launch, prefix,
// cut, text, statistics) stays there.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "ctx.h"
#include "engine_internal.h"
#include "philox.h"

extern "C" {

// get_sample_inf's product (pbsim.cpp:1155-1330): the quality strings that passed the length and accuracy
// filter, in file order.  Parsing, filtering, the statistics and the stored-profile files are the caller's.
int pbsim_set_sample_profile(pbsim_ctx *c, int64_t n, const uint8_t *const *quals, const int64_t *lens) {
  if (!c || n < 1 || !quals || !lens) return fail("pbsim_set_sample_profile: bad argument");
  NEED_DEVICE(c);
  if (c->p.method != PBSIM_METHOD_SAMPLE) return fail("pbsim_set_sample_profile: method is not sample");
  if (n > 0x7fffffffLL) return fail("too many sample reads");
  HIP_OK(hipSetDevice(c->device));
  c->sq_len.resize((size_t)n);
  c->sq_off.resize((size_t)n);
  int64_t total = 0, bytes = 0;
  for (int64_t i = 0; i < n; i++) {
    if (lens[i] < 1 || lens[i] > 1000000) return fail("sample read length outside 1-1000000");
    c->sq_len[(size_t)i] = (int32_t)lens[i];
    c->sq_off[(size_t)i] = bytes;
    total += lens[i];
    bytes += (lens[i] + 7) & ~7LL;
  }
  std::vector<uint8_t> pool((size_t)bytes + 8, 0);
  for (int64_t i = 0; i < n; i++) memcpy(pool.data() + c->sq_off[(size_t)i], quals[i], (size_t)lens[i]);
  if (!upload(c->d_sq, pool.data(), pool.size(), c->stream)) return PBSIM_FAILED;
  HIP_OK(hipStreamSynchronize(c->stream));
  c->sq_total = total;
  return PBSIM_SUCCEEDED;
}

// simulate_by_sample (pbsim.cpp:1694-1949) for the current record.  Each sweep over the profile uses string s
// `sample_num` or `sample_num + 1` times ((sample_value + s) % sample_interval == 0), reads are numbered in that
// order, and a read is made as long as len_total < quota at its start.  A chunk = a run of consecutive strings
// with all their copies; its strings walk in parallel, one lane each (k_walk_sample), and everything after the walk
// (prefix in read order, the cut, text, statistics) is the ordinary batch machinery.
extern "C++" {
namespace {

// Host side of the chunks: which strings a chunk holds and how their copies are laid out in the scratch pool (plan),
// and the upload + walk + pass-0 prefix of a planned chunk on slot 0 (launch).  Shared by the one-GPU driver and the
// sharded one; where a chunk ends never changes a byte (tests/test_gpu_sample.py).
struct SampleChunks {
  struct Ent {
    int64_t line, num;  // string index, copies to make in this chunk
    int32_t len;        // its current length (shorter than the file's once a chain is carried over)
  };
  struct Chunk {
    std::vector<Ent> cand;
    size_t n_c = 0;              // strings of cand[] the chunk takes
    int64_t n_tasks = 0;
    int64_t next_probe = 0;      // the string after the last one that was looked at
    std::vector<int32_t> h_len, h_vbase, h_cap, h_tos, h_sot;
    std::vector<int64_t> h_qoff, h_woff;
    int32_t n_coop_waves = 0;    // leading line waves (the longest strings) whose strings get a wave each
  };
  pbsim_ctx *c;
  int64_t F, G, interval = 1, sample_num = 0;
  // a string whose copies do not fit one chunk continues in the next one: copies done so far, current length
  int64_t carry_line = -1, carry_done = 0;
  int32_t carry_len = 0;
  std::vector<int32_t> order, order_tmp;

  int init(int64_t quota) {
    F = (int64_t)c->sq_len.size();
    G = c->ref_len;
    sample_num = quota / c->sq_total;            // :1718-1728
    const int64_t residue = quota % c->sq_total;
    interval = 1;
    if (residue != 0) {
      interval = (int64_t)((double)(c->sq_total / residue) * 2 + 0.5);
      if (interval > (int64_t)(F * 0.5)) interval = (int64_t)(F * 0.5);
    }
    if (interval < 1) return fail("sample profile holds a single read: the reference divides by zero here (pbsim.cpp:1741)");
    return PBSIM_SUCCEEDED;
  }
  // How many of the chunk's line waves (64 strings each, longest first) are walked one WAVE per string (k_walk_sample's
  // scoop_walk_string): a string's copies are a serial chain, a lane takes 0.6 us per column, a wave 0.03, and a chunk holds at
  // most 2^18 strings -- four lane waves per SIMD, which cannot hide the lanes' latencies.  Default: every string (measured on
  // 200 000 strings, 2 Gbases: 42-53 ms against 60 with the strings below twice the mean length on lanes and 78-152 with all of
  // them there).  PBSIM_COOP_LEN as for the HMM walks: -1 none, 0 all, n = the line waves whose strings all have >= n
  // characters.  Depends on the chunk alone: every rank of a sharded run decides alike.
  int32_t coop_waves(const std::vector<Ent> &cand, size_t n_c) const {
    if (n_c == 0) return 0;
    const int32_t n_w = (int32_t)((n_c + 63) / 64);
    const char *e = getenv("PBSIM_COOP_LEN");
    int64_t thr = e ? atoll(e) : -2;
    if (thr == -1) return 0;
    if (thr == 0) return n_w;
    if (thr < 0) return n_w;
    int32_t n = 0;
    while (n < n_w && cand[(size_t)order[std::min(n_c, (size_t)(n + 1) * 64) - 1]].len >= thr) n++;
    return n;
  }
  // order[] = the first n_c strings by length, longest first, ties in file order: a stable LSD radix sort over the 20 bits a
  // length has (<= 1 000 000, pbsim_set_sample_profile) -- std::stable_sort took 15 ms of a 65 ms job for 200 000 strings
  void sort_by_length(const std::vector<Ent> &cand, size_t n_c) {
    order.resize(n_c);
    order_tmp.resize(n_c);
    uint32_t cnt[1025];
    for (int pass = 0; pass < 2; pass++) {
      const int shift = pass * 10;
      memset(cnt, 0, sizeof cnt);
      auto key = [&](int32_t i) { return ((0xfffffu - (uint32_t)cand[(size_t)i].len) >> shift) & 1023u; };
      if (pass == 0) for (size_t i = 0; i < n_c; i++) cnt[key((int32_t)i) + 1]++;
      else for (size_t i = 0; i < n_c; i++) cnt[key(order_tmp[i]) + 1]++;
      for (int b = 0; b < 1024; b++) cnt[b + 1] += cnt[b];
      if (pass == 0) for (size_t i = 0; i < n_c; i++) order_tmp[cnt[key((int32_t)i)]++] = (int32_t)i;
      else for (size_t i = 0; i < n_c; i++) order[cnt[key(order_tmp[i])]++] = order_tmp[i];
    }
  }
  int64_t copies_of(int64_t sv, int64_t line) const { return sample_num + (((sv + line) % interval == 0) ? 1 : 0); }
  // strings of [line, F) that have copies in this sweep (the sharded driver deals them out in equal runs)
  int64_t count_candidates(int64_t sv, int64_t line) const {
    int64_t n = 0;
    for (int64_t l = line; l < F; l++) n += (copies_of(sv, l) - (l == carry_line ? carry_done : 0)) > 0;
    return n;
  }

  // the chunk that starts at `line`: at most max_cand strings, shrunk until its scratch fits the pool.  ck->cand empty: no
  // string from `line` on has copies (ck->next_probe == F).
  int plan(int64_t sv, int64_t line, size_t max_cand, Chunk *ck) {
    ck->cand.clear();
    ck->cand.reserve((size_t)std::min<int64_t>((int64_t)max_cand, std::max<int64_t>(F - line, 0)));
    int64_t probe = line;
    int64_t phase = interval > 0 ? (sv + probe) % interval : 0;  // (sv + probe) % interval, kept up to date without a division per string
    while (probe < F && ck->cand.size() < max_cand) {
      int64_t k = sample_num + (phase == 0 ? 1 : 0);  // = copies_of(sv, probe)
      if (++phase == interval) phase = 0;
      int32_t len = c->sq_len[(size_t)probe];
      if (probe == carry_line) {
        k -= carry_done;
        len = carry_len;
      }
      if (k > 0) ck->cand.push_back(Ent{probe, k, len});
      probe++;
    }
    ck->next_probe = probe;
    ck->n_c = 0;
    ck->n_tasks = 0;
    if (ck->cand.empty()) return PBSIM_SUCCEEDED;
    // ---- lay the chunk out.  Reads stay in file order; LANES are dealt by length (the longest strings share a
    // wave), one virtual wave of scratch per copy.  Shrink the chunk until it fits the pool.
    std::vector<Ent> &cand = ck->cand;
    size_t n_c = cand.size();
    int64_t need = 0, n_tasks = 0;
    for (;;) {
      sort_by_length(cand, n_c);
      ck->h_vbase.assign(1, 0);
      ck->h_cap.clear();
      ck->h_woff.clear();
      need = 0;
      n_tasks = 0;
      ck->n_coop_waves = coop_waves(cand, n_c);
      for (size_t w0 = 0; w0 < n_c; w0 += 64) {
        int64_t kmax = 0, lmax = 0;
        for (size_t i = w0; i < std::min(n_c, w0 + 64); i++) {
          const Ent &e = cand[(size_t)order[i]];
          kmax = std::max(kmax, e.num);
          lmax = std::max<int64_t>(lmax, std::min<int64_t>(e.len, G));
          n_tasks += e.num;
        }
        const int32_t transposed = (int64_t)(w0 / 64) < ck->n_coop_waves ? kWaveTransposed : 0;  // rows task by task
        int64_t cap_dw = (2 * lmax + kScratchPad + 3) / 4;
        if (transposed) cap_dw = (cap_dw + 3) & ~3LL;  // ... each on a 16-byte boundary (k_sample_qsum reads them 16 bytes at a time)
        for (int64_t k = 0; k < kmax; k++) {
          ck->h_cap.push_back((int32_t)cap_dw | transposed);
          ck->h_woff.push_back(need);
          need += cap_dw * 256 * 3;
        }
        ck->h_vbase.push_back((int32_t)ck->h_cap.size());
      }
      if (need <= c->scratch_budget && n_tasks <= 0x3fffffff && ck->h_cap.size() <= 0x1ffffff) break;
      if (n_c > 1) {
        n_c = (n_c + 1) / 2;
        continue;
      }
      // one string alone: make as many of its copies as fit, the chain continues in the next chunk
      const int64_t per_copy = need / cand[0].num;
      const int64_t kfit = c->scratch_budget / std::max<int64_t>(per_copy, 1);
      if (kfit < 1) return fail("scratch pool too small for a single sampled read (pbsim_set_scratch_bytes)");
      cand[0].num = std::min(cand[0].num, kfit);
    }
    ck->h_len.assign(((n_c + 63) / 64) * 64, 0);
    ck->h_qoff.assign(ck->h_len.size(), 0);
    ck->h_tos.assign(ck->h_cap.size() * 64, -1);
    ck->h_sot.resize((size_t)n_tasks);
    {
      std::vector<int32_t> pos_of(n_c);
      for (size_t i = 0; i < n_c; i++) pos_of[(size_t)order[i]] = (int32_t)i;
      int64_t t = 0;
      for (size_t e = 0; e < n_c; e++) {  // tasks in file order, lanes in length order
        const size_t pos = (size_t)pos_of[e];
        ck->h_len[pos] = cand[e].len;
        ck->h_qoff[pos] = c->sq_off[(size_t)cand[e].line];
        const int64_t v0 = ck->h_vbase[pos / 64];
        for (int64_t k = 0; k < cand[e].num; k++) {
          const int64_t slot = (v0 + k) * 64 + (int64_t)(pos % 64);
          ck->h_tos[(size_t)slot] = (int32_t)t;
          ck->h_sot[(size_t)t] = (int32_t)slot;
          t++;
        }
      }
    }
    ck->n_c = n_c;
    ck->n_tasks = n_tasks;
    return PBSIM_SUCCEEDED;
  }
  // where a chunk leaves the sweep: past its last string, unless that string still has copies to make
  bool last_unfinished(int64_t sv, const Chunk &ck, int64_t *last_done) const {
    const Ent &last = ck.cand[ck.n_c - 1];
    *last_done = (last.line == carry_line ? carry_done : 0) + last.num;
    return *last_done < copies_of(sv, last.line);
  }

  // upload + walk + pass-0 prefix of the chunk on the selected slot, not waited for; its reads are first_read .. first_read + n_tasks - 1
  int enqueue(const Chunk &ck, int64_t first_read) {
    Slot &sl = c->s();
    const int64_t n_tasks = ck.n_tasks;
    const int64_t n_lines = (int64_t)ck.h_len.size(), n_lw = (int64_t)ck.h_vbase.size() - 1, V = (int64_t)ck.h_cap.size();
    if (first_read - 1 + n_tasks > 0xfffffff0LL) return fail("read index exceeds 32 bits");
    // ---- device state of the batch
    HIP_OK(sl.d_flags.ensure(sizeof(DeviceFlags)));
    HIP_OK(sl.d_len.ensure(n_tasks * 4));
    HIP_OK(sl.d_off.ensure(n_tasks * 4));
    HIP_OK(sl.d_task_of_slot.ensure(V * 64 * 4));
    HIP_OK(sl.d_slot_of_task.ensure(n_tasks * 4));
    HIP_OK(sl.d_wave_cap.ensure(V * 4));
    HIP_OK(sl.d_wave_off.ensure(V * 8));
    HIP_OK(sl.d_out_len.ensure(n_tasks * 4));
    HIP_OK(sl.d_maf_len.ensure(n_tasks * 4));
    HIP_OK(sl.d_nsub.ensure(n_tasks * 4));
    HIP_OK(sl.d_nins.ensure(n_tasks * 4));
    HIP_OK(sl.d_ndel.ensure(n_tasks * 4));
    HIP_OK(sl.d_qsum.ensure(n_tasks * 8));
    HIP_OK(sl.d_cum.ensure((n_tasks + 1) * 8));
    HIP_OK(sl.d_scan_tmp.ensure((n_tasks / 1024 + 8) * 8));
    HIP_OK(sl.d_scratch.ensure((size_t)c->scratch_budget + kScratchSlack, true));
    HIP_OK(c->d_sq_line_len.ensure(n_lines * 4));
    HIP_OK(c->d_sq_line_qoff.ensure(n_lines * 8));
    HIP_OK(c->d_sq_vbase.ensure((n_lw + 1) * 4));
    DeviceFlags f0;
    memset(&f0, 0, sizeof f0);
    f0.total_slots = V * 64;
    f0.n_final = n_tasks;
    HIP_OK(hipMemcpyAsync(sl.d_flags.p, &f0, sizeof f0, hipMemcpyHostToDevice, sl.stream));
    HIP_OK(hipMemcpyAsync(c->d_sq_line_len.p, ck.h_len.data(), n_lines * 4, hipMemcpyHostToDevice, sl.stream));
    HIP_OK(hipMemcpyAsync(c->d_sq_line_qoff.p, ck.h_qoff.data(), n_lines * 8, hipMemcpyHostToDevice, sl.stream));
    HIP_OK(hipMemcpyAsync(c->d_sq_vbase.p, ck.h_vbase.data(), (n_lw + 1) * 4, hipMemcpyHostToDevice, sl.stream));
    HIP_OK(hipMemcpyAsync(sl.d_task_of_slot.p, ck.h_tos.data(), V * 64 * 4, hipMemcpyHostToDevice, sl.stream));
    HIP_OK(hipMemcpyAsync(sl.d_slot_of_task.p, ck.h_sot.data(), n_tasks * 4, hipMemcpyHostToDevice, sl.stream));
    HIP_OK(hipMemcpyAsync(sl.d_wave_cap.p, ck.h_cap.data(), V * 4, hipMemcpyHostToDevice, sl.stream));
    HIP_OK(hipMemcpyAsync(sl.d_wave_off.p, ck.h_woff.data(), V * 8, hipMemcpyHostToDevice, sl.stream));
    DeviceFlags *flags = sl.d_flags.as<DeviceFlags>();
    SampleArgs a;
    memset(&a, 0, sizeof a);
    a.seed = c->p.seed;
    a.unit = (uint32_t)c->unit;
    a.first_read = first_read;
    a.n_lines = (int32_t)n_lines;
    a.n_line_waves = (int32_t)n_lw;
    a.n_coop_waves = ck.n_coop_waves;
    {
      // persistent workgroups of the wave path: four per CU = the four waves per SIMD the kernel's 103 VGPRs allow (measured:
      // 512 / 768 / 1024 / 1280 / 2048 workgroups -> 43.2 / 37.4 / 34.9 / 39.6 / 34.9 ms for the 2-Gbase bench; capping the
      // kernel at 96 VGPRs for a fifth wave bought nothing).  PBSIM_SAMPLE_COOP_WG: experiment knob
      const char *cb = exp_env("PBSIM_SAMPLE_COOP_WG");
      a.n_coop_blocks = (int32_t)std::min<int64_t>(((int64_t)ck.n_coop_waves * 64 + 3) / 4, cb && atoi(cb) > 0 ? atoi(cb) : 1024);
    }
    a.n_coop_slots = (int64_t)ck.h_vbase[(size_t)ck.n_coop_waves] * 64;
    a.ref.seq = c->d_seq;
    a.ref.hp = c->d_hp.as<uint8_t>();
    a.ref.len = G;
    a.quals = c->d_sq.as<uint8_t>();
    a.line_qoff = c->d_sq_line_qoff.as<int64_t>();
    a.line_len = c->d_sq_line_len.as<int32_t>();
    a.vbase = c->d_sq_vbase.as<int32_t>();
    a.task_of_slot = sl.d_task_of_slot.as<int32_t>();
    a.wave_cap = sl.d_wave_cap.as<int32_t>();
    a.wave_off = sl.d_wave_off.as<int64_t>();
    a.scratch = sl.d_scratch.as<uint8_t>();
    a.span = sl.d_len.as<int32_t>();
    a.off = sl.d_off.as<int32_t>();
    a.out_len = sl.d_out_len.as<int32_t>();
    a.maf_len = sl.d_maf_len.as<int32_t>();
    a.nsub = sl.d_nsub.as<int32_t>();
    a.nins = sl.d_nins.as<int32_t>();
    a.ndel = sl.d_ndel.as<int32_t>();
    a.qsum = sl.d_qsum.as<double>();
    const uint8_t *t = c->d_qs_tabs_v[c->bias.hp11_seen].as<uint8_t>();
    a.sub_thre = reinterpret_cast<const uint32_t *>(t);
    a.ins_thre = reinterpret_cast<const uint32_t *>(t + 94 * 4);
    a.del_thr = reinterpret_cast<const uint32_t *>(t + 94 * 8);
    a.qprob = reinterpret_cast<const double *>(t + 94 * 8 + 94 * 48);
    a.flags = flags;
    if (sl.sq_pending) {  // (a chunk whose statistics were never fetched: its sums still read the pool)
      HIP_OK(hipStreamWaitEvent(sl.stream, sl.ev_sq_done, 0));
      sl.sq_pending = false;
    }
    launch_walk_sample(a, c->seq_hp_flag, sl.stream);
    if (a.n_coop_slots > 0) {
      if (!c->sq_stream) HIP_OK(hipStreamCreateWithFlags(&c->sq_stream, hipStreamNonBlocking));
      if (!sl.ev_sq_walk) {
        HIP_OK(hipEventCreateWithFlags(&sl.ev_sq_walk, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&sl.ev_sq_done, hipEventDisableTiming));
      }
      HIP_OK(hipEventRecord(sl.ev_sq_walk, sl.stream));
      HIP_OK(hipStreamWaitEvent(c->sq_stream, sl.ev_sq_walk, 0));
      launch_sample_qsum(a, c->sq_stream);
      HIP_OK(hipEventRecord(sl.ev_sq_done, c->sq_stream));
      sl.sq_pending = true;
    }
    launch_gather_pass0_scan(a.out_len, n_tasks, 1, sl.d_cum.as<int64_t>(), sl.d_scan_tmp.as<int64_t>(),
                             &flags->sums[0], sl.stream);
    HIP_OK(hipGetLastError());
    return PBSIM_SUCCEEDED;
  }
  // ... and the wait for it: the chunk's flags and pass-0 bases, the slot's batch state for pbsim_batch_finalize
  int finish(const Chunk &ck, int64_t first_read) {
    Slot &sl = c->s();
    const int64_t n_tasks = ck.n_tasks, V = (int64_t)ck.h_cap.size();
    sl.b_enqueued = false;
    DeviceFlags f;
    if (!read_flags(c, &f)) return PBSIM_FAILED;
    if (f.error & kErrScratchOverflow) return fail("a sampled read produced more MAF columns than its scratch holds");
    sl.b_first = first_read;
    sl.b_n = n_tasks;
    sl.b_slots_max = V * 64;
    sl.b_truncated = false;
    sl.b_enqueued = false;
    sl.b_walked = true;
    sl.b_finalized = false;
    sl.b_pass0 = f.sums[0];
    return PBSIM_SUCCEEDED;
  }
  int launch(const Chunk &ck, int64_t first_read) { return enqueue(ck, first_read) && finish(ck, first_read); }
};

int sample_common_checks(pbsim_ctx *c) {
  if (c->p.method != PBSIM_METHOD_SAMPLE) return fail("pbsim_simulate_sample: method is not sample");
  if (!c->d_seq) return fail("no reference set (pbsim_set_reference)");
  if (c->sq_len.empty()) return fail("no sample profile set (pbsim_set_sample_profile)");
  HIP_OK(hipSetDevice(c->device));
  if (!ensure_class_tables(c) || !ensure_qs_tabs(c, c->bias.hp11_seen)) return PBSIM_FAILED;
  return PBSIM_SUCCEEDED;
}

}  // namespace
}  // extern "C++"

int pbsim_simulate_sample(pbsim_ctx *c, const pbsim_sink *sink) {
  if (!c) return fail("bad argument");
  NEED_DEVICE(c);
  if (!sample_common_checks(c)) return PBSIM_FAILED;
  pbsim_reset_stats(c);
  const int64_t quota = pbsim_unit_quota(c);
  SampleChunks S;
  S.c = c;
  if (!S.init(quota)) return PBSIM_FAILED;
  const int64_t F = S.F;
  for (int s = 0; s < 2; s++) c->slots[(size_t)s].ref = current_ref(c);
  // Where the next chunk starts: a sweep (its sample_value, pbsim.cpp:1732, drawn from the number of reads made so far) and a
  // string of it.  next_chunk plans the next chunk that holds anything, opening sweeps as it goes (:1922: from the second
  // sweep on a string is used once or not at all).
  struct Pos {
    int64_t sv = 0, line = 0;
    bool open = false;
  } pos;
  auto next_chunk = [&](int64_t res_now, Pos *at, SampleChunks::Chunk *ck) -> int {
    for (;;) {
      if (!at->open) {
        at->sv = (int64_t)(header_block(c->p.seed, (uint32_t)c->unit, (uint32_t)(res_now + 1)).w % (uint32_t)F);
        at->line = 0;
        at->open = true;
      }
      while (at->line < F) {
        if (!S.plan(at->sv, at->line, (size_t)1 << 18, ck)) return PBSIM_FAILED;
        if (!ck->cand.empty()) return PBSIM_SUCCEEDED;
        at->line = ck->next_probe;
      }
      S.sample_num = 0;
      S.carry_line = -1;
      at->open = false;
    }
  };
  // Two slots: while a chunk's text emission, delivery and statistics are under way, the NEXT chunk is planned and walks --
  // whenever the next chunk is determined by then (see `look` below).
  DeferTextSync defer_guard(c);
  SampleChunks::Chunk cks[2];
  int cur = 0;
  bool have = false;  // cks[cur] is planned and enqueued already
  const bool trace = getenv("PBSIM_TRACE") != nullptr;
  auto wall_ms = []() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
  };
  int64_t len_total = 0, res = 0;
  bool done = false;
  auto drain = [&]() {
    for (int s = 0; s < 2; s++)
      if (c->slots[(size_t)s].stream) (void)hipStreamSynchronize(c->slots[(size_t)s].stream);
    c->cur = 0;
  };
  while (len_total < quota && !done) {
    SampleChunks::Chunk &ck = cks[cur];
    c->cur = cur;
    if (!have) {
      const double t0 = trace ? wall_ms() : 0;
      if (!next_chunk(res, &pos, &ck)) {
        drain();
        return PBSIM_FAILED;
      }
      const double t1 = trace ? wall_ms() : 0;
      if (!S.enqueue(ck, res + 1)) {
        drain();
        return PBSIM_FAILED;
      }
      if (trace) fprintf(stderr, "[pbsim sample] chunk of %zu strings, %lld reads: planned in %.2f ms, enqueued in %.2f ms\n", ck.n_c, (long long)ck.n_tasks, t1 - t0, wall_ms() - t1);
    }
    have = false;
    const SampleChunks::Ent last = ck.cand[ck.n_c - 1];
    int64_t last_done = 0;
    const bool last_unfinished = S.last_unfinished(pos.sv, ck, &last_done);
    if (!S.finish(ck, res + 1)) {
      drain();
      return PBSIM_FAILED;
    }
    // the quota cut and the text sizes; the text emission is enqueued, not waited for (DeferTextSync)
    pbsim_batch_info bi;
    if (!pbsim_batch_finalize(c, len_total, &bi)) {
      drain();
      return PBSIM_FAILED;
    }
    // every read of the chunk was made, its last string is finished and the quota is not reached: the next chunk is determined.
    // It is planned, uploaded and set walking now, beside this chunk's text emission, delivery and statistics.
    bool look = !last_unfinished && bi.n_final == ck.n_tasks && bi.len_total_after < quota;
    if (look && c->slots[(size_t)(1 - cur)].d_scratch.bytes < (size_t)c->scratch_budget) {
      // the other slot has no pool yet: a second pool (and the text of a second chunk) must fit what the GPU has left, else
      // the chunks simply follow each other on this slot as they did before round 3
      size_t free_b = 0, total_b = 0;
      HIP_OK(hipMemGetInfo(&free_b, &total_b));
      const double text_now = (double)c->s().d_read_text.bytes + (double)c->s().d_maf_text.bytes;
      if ((double)free_b < 1.1 * (double)c->scratch_budget + 1.5 * text_now + (double)(2ull << 30)) look = false;
    }
    Pos pos2 = pos;
    if (look) {
      S.carry_line = -1;
      S.carry_done = 0;
      pos2.line = last.line + 1;
      c->cur = 1 - cur;
      if (!next_chunk(res + bi.n_final, &pos2, &cks[1 - cur]) || !S.enqueue(cks[1 - cur], res + bi.n_final + 1)) {
        drain();
        return PBSIM_FAILED;
      }
      c->cur = cur;
    }
    if (!deliver(c, sink)) {
      drain();
      return PBSIM_FAILED;
    }
    len_total = bi.len_total_after;
    res += bi.n_final;
    if (bi.n_final < ck.n_tasks) done = true;  // the quota was reached inside this chunk (:1735, :1749)
    if (look) {
      pos = pos2;
      cur = 1 - cur;
      have = true;
      continue;
    }
    if (last_unfinished && !done) {
      S.carry_line = last.line;
      S.carry_done = last_done;
      HIP_OK(hipMemcpy(&S.carry_len, c->s().d_out_len.as<int32_t>() + (ck.n_tasks - 1), 4, hipMemcpyDeviceToHost));
      pos.line = last.line;
    } else {
      S.carry_line = -1;
      S.carry_done = 0;
      pos.line = last.line + 1;
    }
  }
  if (have) {  // (cannot happen: a chunk is only enqueued ahead when the total stays below the quota) -- nothing is left in flight
    drain();
    return fail("internal: a sampled chunk was left in flight");
  }
  c->cur = 0;
  return PBSIM_SUCCEEDED;
}

// The same record on several ranks (one context per GPU, every rank holds the record and the profile).  The copies of ONE
// string are a chain (each copy is as long as the read the previous one produced), but strings are independent, and the
// quota test at a read's start (`len_total < quota`, pbsim.cpp:1749) is the same prefix dependence as the wgs quota rule: a
// round = W chunks of consecutive strings of one sweep, rank r walks chunk r, and three small all-gathers per round place
// the quota prefix (A: pass-0 bases), the cut (B: reads made, bases behind them) and every rank's byte range in the
// record's two streams (C).  Chunks in front of the cut are delivered, the rest of the round is void.  Every number the
// planner uses is the same on all ranks (the pool size is agreed first), so all ranks plan the same chunks.  A string whose
// copies do not fit one chunk's pool (the carry-over of the one-GPU driver) is refused here: give the ranks a larger pool.
// `*agreed`: the failure was learned through a collective's status word (or is the same on every rank by construction), so
// every rank leaves at the same exchange; any other failure is this rank's alone and the caller releases the others (abort).
static int sample_comm_run(pbsim_ctx *c, const pbsim_comm *comm, const pbsim_record_sink *sink, bool *agreed) {
  const int W = comm->world, rank = comm->rank;
  int ok = sample_common_checks(c);
  SampleChunks S;
  S.c = c;
  const int64_t quota = ok ? pbsim_unit_quota(c) : 0;
  if (ok) ok = S.init(quota);
  {  // every rank is ready, and plans with the same pool
    std::string keep = g_err;
    int64_t v[2] = {ok ? 0 : 1, -c->scratch_budget};
    if (!comm->all_reduce_i64(comm->user, v, 2, PBSIM_OP_MAX)) return fail("pbsim_comm.all_reduce_i64 failed");
    if (v[0]) {
      *agreed = true;
      return ok ? fail("another rank failed") : fail(keep);
    }
    c->scratch_budget = -v[1];
  }
  pbsim_reset_stats(c);
  c->st.keep_values = true;
  c->cur = 0;
  Slot &sl = c->s();
  sl.ref = current_ref(c);
  const int64_t F = S.F;
  int64_t len_total = 0, res = 0, read_off = 0, maf_off = 0;
  bool done = false;
  std::vector<SampleChunks::Chunk> cks((size_t)W);
  std::string buf_r, buf_m;
  struct Keep {
    std::string *r, *m;
  } keep = {&buf_r, &buf_m};
  const pbsim_sink collect = {&keep,
                              [](void *u, const char *t, int64_t k) { ((Keep *)u)->r->append(t, (size_t)k); return 1; },
                              [](void *u, const char *t, int64_t k) { ((Keep *)u)->m->append(t, (size_t)k); return 1; }};
  auto gather = [&](const int64_t *send, int n, std::vector<int64_t> *recv) -> int {
    recv->assign((size_t)W * n, 0);
    return comm->all_gather_i64(comm->user, send, n, recv->data()) ? PBSIM_SUCCEEDED : fail("pbsim_comm.all_gather_i64 failed");
  };
  std::vector<int64_t> A, B, Cs;
  const bool trace = getenv("PBSIM_TRACE") != nullptr;
  while (len_total < quota && !done) {
    const int64_t sv = (int64_t)(header_block(c->p.seed, (uint32_t)c->unit, (uint32_t)(res + 1)).w % (uint32_t)F);  // :1732
    int64_t line = 0;
    while (line < F && len_total < quota && !done) {
      // ---- the round's chunks: the sweep's remaining strings in W equal runs (at most; the pool may cut a run short)
      const int64_t left = S.count_candidates(sv, line);
      if (left == 0) break;
      const size_t per = (size_t)std::min<int64_t>((left + W - 1) / W, (int64_t)1 << 18);
      int n_chunks = 0;
      int64_t at = line, first = res + 1, my_first = 0;
      int local = PBSIM_SUCCEEDED;
      std::string local_err;
      for (int q = 0; q < W && at < F; q++) {
        SampleChunks::Chunk &ck = cks[(size_t)q];
        if (!S.plan(sv, at, per, &ck)) {
          local = PBSIM_FAILED;  // (the same on every rank: the plan depends on nothing local)
          local_err = g_err;
          break;
        }
        if (ck.cand.empty()) break;
        int64_t last_done = 0;
        if (S.last_unfinished(sv, ck, &last_done)) {
          local = PBSIM_FAILED;
          local_err = "the copies of one sampled read do not fit a rank's scratch pool: the sharded sampling method needs a larger "
                      "pool (pbsim_set_scratch_bytes / PBSIM_SCRATCH_MB), or run this profile on one GPU";
          break;
        }
        if (q == rank) my_first = first;
        first += ck.n_tasks;
        at = ck.cand[ck.n_c - 1].line + 1;
        n_chunks++;
      }
      if (!local) {
        *agreed = true;  // (the plan depends on nothing local: every rank refuses alike)
        return fail(local_err);
      }
      const bool mine = rank < n_chunks;
      if (trace)
        fprintf(stderr, "[pbsim sample r%d] sweep sv=%lld line=%lld left=%lld chunks=%d first=%lld len_total=%lld\n", rank, (long long)sv,
                (long long)line, (long long)left, n_chunks, (long long)(res + 1), (long long)len_total);
      // ---- walk, A: pass-0 bases of every chunk -> the quota prefix
      int64_t sendA[2] = {0, 0};
      if (mine) {
        if (S.launch(cks[(size_t)rank], my_first)) sendA[0] = sl.b_pass0;
        else sendA[1] = 1, local_err = g_err;
      }
      if (!gather(sendA, 2, &A)) return PBSIM_FAILED;
      int64_t before = len_total, bad = 0;
      for (int q = 0; q < W; q++) {
        bad += A[(size_t)q * 2 + 1];
        if (q < rank) before += A[(size_t)q * 2];
      }
      if (bad) {
        *agreed = true;
        return sendA[1] ? fail(local_err) : fail("another rank failed");
      }
      // ---- the cut inside my chunk, B: reads made and bases behind them -> the first chunk that stops short
      pbsim_batch_info bi;
      memset(&bi, 0, sizeof bi);
      int64_t sendB[3] = {0, before, 0};
      if (mine) {
        if (finalize_cut(c, before, &bi)) sendB[0] = bi.n_final, sendB[1] = bi.len_total_after;
        else sendB[2] = 1, local_err = g_err;
      }
      if (!gather(sendB, 3, &B)) return PBSIM_FAILED;
      bad = 0;
      for (int q = 0; q < W; q++) bad += B[(size_t)q * 3 + 2];
      if (bad) {
        *agreed = true;
        return sendB[2] ? fail(local_err) : fail("another rank failed");
      }
      int cut = -1;
      for (int q = 0; q < n_chunks && cut < 0; q++)
        if (B[(size_t)q * 3] < cks[(size_t)q].n_tasks) cut = q;
      const int last_valid = cut < 0 ? n_chunks - 1 : cut;
      // ---- text of the valid chunks, C: byte counts -> every rank's range in the record's streams
      buf_r.clear();
      buf_m.clear();
      int64_t sendC[3] = {0, 0, 0};
      if (mine && rank <= last_valid && bi.n_final > 0) {
        if (finalize_text(c, &bi) && deliver(c, &collect)) sendC[0] = (int64_t)buf_r.size(), sendC[1] = (int64_t)buf_m.size();
        else sendC[2] = 1, local_err = g_err;
      }
      if (!gather(sendC, 3, &Cs)) return PBSIM_FAILED;
      bad = 0;
      int64_t r_at = read_off, m_at = maf_off;
      for (int q = 0; q < W; q++) {
        bad += Cs[(size_t)q * 3 + 2];
        if (q < rank) r_at += Cs[(size_t)q * 3], m_at += Cs[(size_t)q * 3 + 1];
        read_off += Cs[(size_t)q * 3];
        maf_off += Cs[(size_t)q * 3 + 1];
      }
      if (bad) {
        *agreed = true;
        return sendC[2] ? fail(local_err) : fail("another rank failed");
      }
      if (sink && sink->on_read_text && !buf_r.empty() && !sink->on_read_text(sink->user, c->unit, buf_r.data(), (int64_t)buf_r.size(), r_at))
        return fail("sink aborted (read text)");
      if (sink && sink->on_maf_text && !buf_m.empty() && !sink->on_maf_text(sink->user, c->unit, buf_m.data(), (int64_t)buf_m.size(), m_at))
        return fail("sink aborted (MAF text)");
      // ---- the record's state, identical on every rank
      for (int q = 0; q <= last_valid; q++) res += B[(size_t)q * 3];
      len_total = B[(size_t)last_valid * 3 + 1];
      if (cut >= 0) done = true;  // the quota was reached inside this round (:1735, :1749)
      line = cks[(size_t)last_valid].cand[cks[(size_t)last_valid].n_c - 1].line + 1;
      if (trace)
        fprintf(stderr, "[pbsim sample r%d]   round done: cut=%d res=%lld len_total=%lld next line=%lld\n", rank, cut, (long long)res,
                (long long)len_total, (long long)line);
    }
    S.sample_num = 0;  // :1922
  }
  int64_t extra[2] = {0, 0};
  if (!stats_merge(&c->st, c->p, comm, extra, 0)) return PBSIM_FAILED;
  c->st.keep_values = false;
  if (sink && sink->on_record_done) {
    pbsim_stats st;
    stats_finish(c->st, c->p, c->ref_len, &st);
    if (!sink->on_record_done(sink->user, c->unit, &st, read_off, maf_off)) return fail("sink aborted (record done)");
  }
  return PBSIM_SUCCEEDED;
}

int pbsim_simulate_sample_comm(pbsim_ctx *c, const pbsim_comm *comm, const pbsim_record_sink *sink) {
  if (!c) return fail("bad argument");
  NEED_DEVICE(c);
  if (!comm || comm->world <= 1) return fail("pbsim_simulate_sample_comm: a communicator of at least two ranks (else pbsim_simulate_sample)");
  if (!comm->all_gather_i64 || !comm->all_reduce_i64) return fail("pbsim_comm: all_gather_i64 and all_reduce_i64 must be set");
  bool agreed = false;
  const int ok = sample_comm_run(c, comm, sink, &agreed);
  if (!ok && !agreed && comm->abort) {
    // a sink callback, the statistics merge, a HIP error between two exchanges: the other ranks cannot know and would wait
    // in their next all-gather (for ever with a host barrier, until the watchdog with RCCL) -- ADVICE r3
    const std::string keep = g_err;
    comm->abort(comm->user);
    g_err = keep;
  }
  return ok;
}

}  // extern "C"
