// engine.cpp -- host orchestration of the gfx950 kernels and the C ABI of
// include/pbsim3_amd.h.  One context = one GPU, one HIP stream, one unit
// (FASTA record or transcript set) resident in HBM at a time.
//
// Data in HBM per context:
//   reference  seq u8[len+pad] (upper-cased in place by K0), hp u8[len+pad]
//   tables     prob2len i32[<=100001], prob2acc u8[<=100001], class blobs
//   batch      per-read header arrays, per-task result arrays, task<->slot maps,
//              wave scratch pool (wave-transposed rows), text buffers
// The product has NO CPU fallback: every compute entry point needs the device.
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
#include "unit_io.h"
#include "philox.h"

namespace pbsim {
thread_local std::string g_err;
int fail(const std::string &m) {
  g_err = m;
  return PBSIM_FAILED;
}
}  // namespace pbsim

namespace {

bool has_quality_row(const pbsim_ctx *c) { return c->p.method == PBSIM_METHOD_QS || c->p.method == PBSIM_METHOD_SAMPLE; }
int regions_of(const pbsim_ctx *c) { return has_quality_row(c) ? 3 : 2; }
int ncls_of(const pbsim_ctx *c) { return c->hdr.acc_hi - c->hdr.acc_lo + 1; }

}  // namespace

// ---- shared with sample.cpp (engine_internal.h) ----
int pbsim::upload(DevBuf &b, const void *src, size_t n, hipStream_t s) {
  HIP_OK(b.ensure(n));
  HIP_OK(hipMemcpyAsync(b.p, src, n, hipMemcpyHostToDevice, s));
  return PBSIM_SUCCEEDED;
}

int pbsim::ensure_header_tables(pbsim_ctx *c) {
  if (c->header_uploaded) return PBSIM_SUCCEEDED;
  if (!upload(c->d_prob2len, c->hdr.prob2len.data(), c->hdr.prob2len.size() * sizeof(int32_t), c->stream))
    return PBSIM_FAILED;
  if (!upload(c->d_prob2acc, c->hdr.prob2acc.data(), c->hdr.prob2acc.size(), c->stream)) return PBSIM_FAILED;
  HIP_OK(hipStreamSynchronize(c->stream));
  c->header_uploaded = true;
  return PBSIM_SUCCEEDED;
}

// [sub 94 u32 | ins 94 u32 | del 94*12 u32 | qprob 94 f64]: qc[].prob and set_mut's thresholds.  Only del_thr[q][0] depends on
// whether an hp == 11 base has been counted yet (Q15), so both variants can be resident and a batch picks its record's.
int pbsim::ensure_qs_tabs(pbsim_ctx *c, bool hp11) {
  if (c->qs_tabs_ready[hp11]) return PBSIM_SUCCEEDED;
  HpBias b = c->bias;
  b.hp11_seen = hp11;
  QsClassTables q;
  build_mut_tables(c->p, b, &q);
  std::vector<uint8_t> t(94 * 4 * 2 + 94 * 12 * 4 + 94 * 8);
  memcpy(t.data(), q.sub_thre, 94 * 4);
  memcpy(t.data() + 94 * 4, q.ins_thre, 94 * 4);
  memcpy(t.data() + 94 * 8, q.del_thr, 94 * 12 * 4);
  memcpy(t.data() + 94 * 8 + 94 * 48, q.qprob, 94 * 8);
  if (!upload(c->d_qs_tabs_v[hp11], t.data(), t.size(), c->stream)) return PBSIM_FAILED;
  HIP_OK(hipStreamSynchronize(c->stream));
  c->qs_tabs_ready[hp11] = true;
  return PBSIM_SUCCEEDED;
}

int pbsim::ensure_class_tables(pbsim_ctx *c) {
  if (!c->class_tables_dirty) return PBSIM_SUCCEEDED;
  std::string e;
  c->qs_tabs_ready[0] = c->qs_tabs_ready[1] = false;  // the bias may have changed
  if (c->p.method == PBSIM_METHOD_SAMPLE) {  // no model: only qc[].prob and set_mut's thresholds
  } else if (c->p.method == PBSIM_METHOD_ERR) {
    if (!c->err) return fail("no ERRHMM model loaded (pbsim_load_errhmm)");
    const bool wgs = c->p.strategy == PBSIM_STRATEGY_WGS;
    if (!build_err_class_tables(*c->err, c->hdr, c->bias, wgs, &c->ect, &e)) return fail(e);
    if (!upload(c->d_cls, c->ect.blob.data(), c->ect.blob.size(), c->stream)) return PBSIM_FAILED;
  } else {
    if (!c->qs) return fail("no QSHMM model loaded (pbsim_load_qshmm)");
    if (!build_qs_class_tables(*c->qs, c->hdr, c->bias, c->p, &c->qct, &e)) return fail(e);
    if (!upload(c->d_cls, c->qct.blob.data(), c->qct.blob.size(), c->stream)) return PBSIM_FAILED;
  }
  HIP_OK(hipStreamSynchronize(c->stream));
  c->class_tables_dirty = false;
  return PBSIM_SUCCEEDED;
}

// Through PINNED staging: a device-to-host copy into pageable memory (a stack variable) is not asynchronous -- the runtime
// stages it and waits in ways that depend on everything else the device is doing; with three walks in flight such a
// "small" read took tens of milliseconds.
int pbsim::read_flags(pbsim_ctx *c, DeviceFlags *f) {
  Slot &sl = c->s();
  HIP_OK(sl.h_flags.ensure(sizeof(DeviceFlags) + 64));
  HIP_OK(hipMemcpyAsync(sl.h_flags.p, sl.d_flags.p, sizeof(DeviceFlags), hipMemcpyDeviceToHost, sl.stream));
  HIP_OK(hipStreamSynchronize(sl.stream));
  memcpy(f, sl.h_flags.p, sizeof(DeviceFlags));
  return PBSIM_SUCCEEDED;
}

namespace {

// upper-case + homopolymer lengths on the GPU: enqueue on `stream` ...
static int enqueue_prepare(pbsim_ctx *c, uint8_t *d_seq, DevBuf &hp, DevBuf &tiles, DevBuf &flags, int64_t len, int keep_first_case,
                           hipStream_t stream) {
  const int64_t n_tiles = (len + kHpTile - 1) / kHpTile;
  HIP_OK(hp.ensure((size_t)len + 64));
  HIP_OK(tiles.ensure((size_t)(n_tiles + 1) * 4 * sizeof(int64_t)));
  HIP_OK(flags.ensure(sizeof(DeviceFlags)));
  HIP_OK(hipMemsetAsync(flags.p, 0, sizeof(DeviceFlags), stream));
  HIP_OK(hipMemsetAsync(hp.as<uint8_t>() + len, 0, 64, stream));
  int64_t *t = tiles.as<int64_t>();
  launch_prepare_reference(d_seq, hp.as<uint8_t>(), c->p.hp_del_bias == 1, len, t, t + (n_tiles + 1), t + 2 * (n_tiles + 1),
                           t + 3 * (n_tiles + 1), keep_first_case, flags.as<DeviceFlags>(), stream);
  HIP_OK(hipGetLastError());
  return PBSIM_SUCCEEDED;
}
// ... and collect: adds the unit's hp census to `census_out`, notes whether the sequence bytes carry the hp == 11 flag
static int finish_prepare(pbsim_ctx *c, DevBuf &flags, hipStream_t stream, int64_t census_out[kHpSlots]) {
  DeviceFlags f;
  HIP_OK(hipMemcpyAsync(&f, flags.p, sizeof f, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));
  for (int i = 0; i < kHpSlots; i++) census_out[i] += (int64_t)f.hpfreq[i];
  c->seq_hp_flag = c->p.hp_del_bias == 1 && !f.high_bytes;
  return PBSIM_SUCCEEDED;
}
}  // namespace
int pbsim::prepare_reference(pbsim_ctx *c, uint8_t *d_seq, int64_t len, int keep_first_case, int64_t census_out[kHpSlots]) {
  if (!enqueue_prepare(c, d_seq, c->d_hp, c->d_tiles, c->d_ref_flags, len, keep_first_case, c->stream)) return PBSIM_FAILED;
  return finish_prepare(c, c->d_ref_flags, c->stream, census_out);
}
namespace {

void note_hp11(pbsim_ctx *c, const int64_t census[kHpSlots]) {
  // hpfreq[11]++ in get_genome_seq (pbsim.cpp:1058) lands in hp_del_bias[0]:
  // from then on the Q15 deletion test can fire when the draw is exactly 0
  if (census[11] > 0 && !c->bias.hp11_seen) {
    c->bias.hp11_seen = true;  // batches pick the matching set_mut table variant through RefDesc::hp11 (ensure_qs_tabs)
  }
}

}  // namespace

int64_t pbsim::batch_capacity(const pbsim_ctx *c) { return batch_capacity_for(c, c->ref_len); }

double pbsim::scratch_factor_of(const pbsim_ctx *c) { return std::min(2.0, std::max(1.0, c->scratch_factor)); }

int64_t pbsim::batch_capacity_for(const pbsim_ctx *c, int64_t ref_len) {
  const double mean = std::min<double>(c->hdr.mean_len, (double)std::max<int64_t>(ref_len, 1));
  const double per_task = (double)regions_of(c) * (scratch_factor_of(c) * mean + kScratchPad) * 1.12 + 64.0;
  int64_t n = (int64_t)((double)c->scratch_budget / (per_task * c->p.pass_num));
  n = std::max<int64_t>(n, 1);
  n = std::min<int64_t>(n, (int64_t)(0x7fffff00 / std::max(1, c->p.pass_num)));
  return n;
}

extern "C" {

const char *pbsim_last_error(void) { return g_err.c_str(); }
const char *pbsim_version(void) { return "pbsim3_amd 0.1 (gfx950)"; }

void pbsim_params_default(pbsim_params *p) {  // pbsim.cpp:1539-1685
  memset(p, 0, sizeof(*p));
  p->strategy = PBSIM_STRATEGY_WGS;
  p->method = PBSIM_METHOD_ERR;
  p->seed = 1;
  p->pass_num = 1;
  p->depth = 20.0;
  p->accuracy_mean = 0.85;
  p->len_mean = 9000;
  p->len_sd = 7000;
  p->hp_del_bias = 1;
  p->len_min = 100;
  p->len_max = 1000000;
  p->sub_ratio = 6;
  p->ins_ratio = 55;
  p->del_ratio = 39;
  strcpy(p->id_prefix, "S");
}

void pbsim_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  const U4 r = philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
  out[0] = r.x;
  out[1] = r.y;
  out[2] = r.z;
  out[3] = r.w;
}

pbsim_ctx *pbsim_create(const pbsim_params *p, int device) {
  if (!p) {
    fail("pbsim_create: params is NULL");
    return nullptr;
  }
  if (p->strategy < 1 || p->strategy > 3 ||
      (p->method != PBSIM_METHOD_QS && p->method != PBSIM_METHOD_ERR && p->method != PBSIM_METHOD_SAMPLE)) {
    fail("--strategy and --method must be set.");
    return nullptr;
  }
  if (p->method == PBSIM_METHOD_SAMPLE && p->strategy != PBSIM_STRATEGY_WGS) {  // pbsim.cpp:1461-1464
    fail("sampling-based simulation is possible only for wgs strategy.");
    return nullptr;
  }
  if (p->method == PBSIM_METHOD_SAMPLE && p->pass_num > 1) {  // pbsim.cpp:1675-1680
    fail("sampling-based simulation supports only single-pass.");
    return nullptr;
  }
  if (p->len_min > p->len_max || p->len_min < 1 || p->len_max > 1000000) {
    fail("length min is greater than max, or outside 1-1000000.");
    return nullptr;
  }
  if (p->pass_num < 1) {
    fail("pass_num: Acceptable range is more than 1.");
    return nullptr;
  }
  if (strnlen(p->id_prefix, sizeof p->id_prefix) >= sizeof p->id_prefix) {
    fail("id-prefix is too long (max 63)");
    return nullptr;
  }
  std::unique_ptr<pbsim_ctx> c(new pbsim_ctx);
  c->p = *p;
  c->device = device;
  std::string e;
  if (!build_header_tables(c->p, &c->hdr, &e)) {
    fail(e);
    return nullptr;
  }
  hp_bias_default(&c->bias);
  if (device == -1) return c.release();  // tables-only context: every compute entry point refuses
  int n = 0;
  hipError_t he = hipGetDeviceCount(&n);
  if (he != hipSuccess || n <= 0) {
    fail("no HIP device available: this library is the gfx950 product path and has no CPU fallback");
    return nullptr;
  }
  if (device < 0 || device >= n) {
    fail("device index out of range");
    return nullptr;
  }
  if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess) {
    fail("cannot initialise the HIP device/stream");
    return nullptr;
  }
  // The long walk kernel goes to a LOW priority stream of its slot, everything short (header, sort, scans, text
  // emission) to a HIGH priority one: workgroups of an older kernel are otherwise dispatched before those of a
  // younger one, and the other slot's half-millisecond kernels waited 15-20 ms behind a walk's pending workgroups.
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  for (Slot &sl : c->slots) {
    if (hipStreamCreateWithPriority(&sl.stream, hipStreamNonBlocking, prio_greatest) != hipSuccess)
      sl.stream = nullptr;
    if (hipStreamCreateWithPriority(&sl.walk_stream, hipStreamNonBlocking, prio_least) != hipSuccess)
      sl.walk_stream = nullptr;
    // no priorities on this runtime: plain streams still give the right answers, only the overlap is worse
    if ((!sl.stream && hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking) != hipSuccess) ||
        (!sl.walk_stream && hipStreamCreateWithFlags(&sl.walk_stream, hipStreamNonBlocking) != hipSuccess)) {
      fail("cannot create a HIP stream");
      return nullptr;
    }
    (void)hipEventCreateWithFlags(&sl.ev_prep, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&sl.ev_coop, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&sl.ev_text, hipEventDisableTiming);
    (void)hipEventCreate(&sl.ev_t0);
    (void)hipEventCreate(&sl.ev_t1);
    // (normal priority: streams of one priority share a few hardware queues, and a packet waits for the one in front of it in
    // its queue -- a 10 ms walk of long reads among the HIGH priority streams held up other slots' flag reads for 30 ms)
    if (hipStreamCreateWithFlags(&sl.coop_stream, hipStreamNonBlocking) != hipSuccess) sl.coop_stream = nullptr;
    (void)hipEventCreate(&sl.ev0);
    (void)hipEventCreate(&sl.ev1);
    (void)hipEventCreate(&sl.ev2);
    (void)hipEventCreate(&sl.ev3);
  }
  if (const char *sf = getenv("PBSIM_SCRATCH_FACTOR")) {  // fixes the rows' factor (ctx.h); 2 = the reference's own bound
    if (atof(sf) >= 1.0) {
      c->scratch_factor = std::min(2.0, atof(sf));
      c->scratch_factor_fixed = true;
    }
  }
  const char *mb = getenv("PBSIM_SCRATCH_MB");
  c->scratch_budget = (mb && atoll(mb) > 0) ? atoll(mb) * (1LL << 20) : (8LL << 30);
  c->scratch_auto = !(mb && atoll(mb) > 0);
  const char *pd = exp_env("PBSIM_PIPELINE_DEPTH");
  if (pd && atoi(pd) >= 1) c->pipeline_depth = std::min(kMaxSlots, atoi(pd));
  return c.release();
}

void pbsim_destroy(pbsim_ctx *c) {
  if (!c) return;
  if (c->device >= 0) (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (Slot &sl : c->slots) {
    if (sl.stream) (void)hipStreamSynchronize(sl.stream);
    if (sl.ev0) (void)hipEventDestroy(sl.ev0);
    if (sl.ev1) (void)hipEventDestroy(sl.ev1);
    if (sl.ev2) (void)hipEventDestroy(sl.ev2);
    if (sl.ev3) (void)hipEventDestroy(sl.ev3);
    if (sl.stream) (void)hipStreamDestroy(sl.stream);
    for (DfLane &L : sl.df) {
      for (int i = 0; i < kDfBuffers; i++) {
        if (L.ev_df[i]) (void)hipEventDestroy(L.ev_df[i]);
        if (L.ev_cp[i]) (void)hipEventDestroy(L.ev_cp[i]);
        if (L.ev_k0[i]) (void)hipEventDestroy(L.ev_k0[i]);
        if (L.ev_k1[i]) (void)hipEventDestroy(L.ev_k1[i]);
      }
      for (hipStream_t &st : L.own)
        if (st) (void)hipStreamDestroy(st);
    }
    if (sl.walk_stream) (void)hipStreamDestroy(sl.walk_stream);
    if (sl.ev_prep) (void)hipEventDestroy(sl.ev_prep);
    if (sl.coop_stream) (void)hipStreamDestroy(sl.coop_stream);
    if (sl.ev_coop) (void)hipEventDestroy(sl.ev_coop);
    if (sl.ev_text) (void)hipEventDestroy(sl.ev_text);
    if (sl.ev_t0) (void)hipEventDestroy(sl.ev_t0);
    if (sl.ev_t1) (void)hipEventDestroy(sl.ev_t1);
  }
  for (auto &lane : c->df_streams)
    for (hipStream_t &st : lane)
      if (st) (void)hipStreamDestroy(st);
  if (c->ev_prof_base) (void)hipEventDestroy(c->ev_prof_base);
  if (c->prefetch_stream) (void)hipStreamDestroy(c->prefetch_stream);
  if (c->sq_stream) (void)hipStreamDestroy(c->sq_stream);
  for (Slot &sl : c->slots) {
    if (sl.ev_sq_walk) (void)hipEventDestroy(sl.ev_sq_walk);
    if (sl.ev_sq_done) (void)hipEventDestroy(sl.ev_sq_done);
  }
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int pbsim_load_errhmm(pbsim_ctx *c, const char *path) {
  if (!c || !path) return fail("pbsim_load_errhmm: bad argument");
  std::unique_ptr<ErrModel> m(new ErrModel);
  std::string e;
  if (!parse_errhmm(path, m.get(), &e)) return fail(e);
  c->err = std::move(m);
  c->class_tables_dirty = true;
  c->coop_wg_errhmm[0] = c->coop_wg_errhmm[1] = 0;
  return PBSIM_SUCCEEDED;
}

int pbsim_load_qshmm(pbsim_ctx *c, const char *path) {
  if (!c || !path) return fail("pbsim_load_qshmm: bad argument");
  std::unique_ptr<QsModel> m(new QsModel);
  std::string e;
  if (!parse_qshmm(path, m.get(), &e)) return fail(e);
  c->qs = std::move(m);
  c->class_tables_dirty = true;
  c->coop_wg_errhmm[0] = c->coop_wg_errhmm[1] = 0;
  return PBSIM_SUCCEEDED;
}

// gives the slots' large device buffers (scratch pools, text, deflate staging) back; the next batch re-allocates what it needs
int pbsim_release_pools(pbsim_ctx *c) {
  if (!c) return fail("bad argument");
  NEED_DEVICE(c);
  HIP_OK(hipSetDevice(c->device));
  for (Slot &sl : c->slots) {
    if (sl.b_enqueued) return fail("pbsim_release_pools: a batch is in flight");
    if (sl.stream) HIP_OK(hipStreamSynchronize(sl.stream));
  }
  HIP_OK(hipDeviceSynchronize());
  for (Slot &sl : c->slots) {
    sl.d_scratch.release();
    sl.d_read_text.release();
    sl.d_maf_text.release();
    for (DfLane &L : sl.df) {
      for (DevBuf &b : L.d_df_dense) b.release();
      L.arena_release();  // the page-locked blocks of several-rank jobs (their automatic trim waits for sixteen idle jobs)
    }
    sl.b_walked = sl.b_finalized = false;
  }
  {
    std::lock_guard<std::mutex> lk(c->job_mu);
    c->job_spare.clear();  // the buffers pbsim_job_begin kept of the records it dropped
  }
  return PBSIM_SUCCEEDED;
}

int pbsim_set_scratch_bytes(pbsim_ctx *c, int64_t bytes) {
  if (!c || bytes < (1 << 20)) return fail("pbsim_set_scratch_bytes: bad argument");
  c->scratch_budget = bytes;
  c->scratch_auto = false;
  return PBSIM_SUCCEEDED;
}

int pbsim_add_hp_census(pbsim_ctx *c, const uint8_t *seq, int64_t len) {
  if (!c || !seq || len < 1) return fail("pbsim_add_hp_census: bad argument");
  NEED_DEVICE(c);
  HIP_OK(hipSetDevice(c->device));
  HIP_OK(c->d_seq_own.ensure((size_t)len + 64));
  HIP_OK(hipMemcpyAsync(c->d_seq_own.p, seq, (size_t)len, hipMemcpyHostToDevice, c->stream));
  HIP_OK(hipMemsetAsync(c->d_seq_own.as<uint8_t>() + len, 0, 64, c->stream));
  if (!prepare_reference(c, c->d_seq_own.as<uint8_t>(), len, 0, c->census)) return PBSIM_FAILED;
  note_hp11(c, c->census);
  if (c->census[11] > 0) c->hp11_explicit = c->hp11_before_job = true;  // seen by the pre-pass: in front of every record of the genome
  return PBSIM_SUCCEEDED;
}

int pbsim_finish_hp_census(pbsim_ctx *c) {
  if (!c) return fail("pbsim_finish_hp_census: bad argument");
  if (c->p.hp_del_bias != 1) {
    const bool seen = c->bias.hp11_seen;
    hp_bias_from_census(c->p.hp_del_bias, c->census, &c->bias);
    c->bias.hp11_seen = seen;
    c->class_tables_dirty = true;
    c->coop_wg_errhmm[0] = c->coop_wg_errhmm[1] = 0;
  }
  c->census_done = true;
  c->census_from_job = false;
  return PBSIM_SUCCEEDED;
}

// pbsim_prefetch_reference*: upload + prepare the NEXT record on a stream of its own while the current one is simulated; the
// following pbsim_set_reference* call with the same pointer and length adopts it instead of doing the work again
static int prefetch_reference(pbsim_ctx *c, const void *seq, int64_t len, hipMemcpyKind kind) {
  if (!c || !seq) return fail("pbsim_prefetch_reference: bad argument");
  NEED_DEVICE(c);
  HIP_OK(hipSetDevice(c->device));
  if (len < 1 || len > 1000000000LL) return fail("pbsim_prefetch_reference: bad length");
  if (c->p.strategy != PBSIM_STRATEGY_WGS) return fail("pbsim_prefetch_reference: strategy is not wgs");
  if (!c->prefetch_stream) HIP_OK(hipStreamCreateWithFlags(&c->prefetch_stream, hipStreamNonBlocking));
  c->pf_src = nullptr;
  HIP_OK(hipStreamSynchronize(c->prefetch_stream));  // an earlier prefetch that was never adopted
  HIP_OK(c->d_seq_next.ensure((size_t)len + 64));
  HIP_OK(hipMemcpyAsync(c->d_seq_next.p, seq, (size_t)len, kind, c->prefetch_stream));
  HIP_OK(hipMemsetAsync(c->d_seq_next.as<uint8_t>() + len, 0, 64, c->prefetch_stream));
  if (!enqueue_prepare(c, c->d_seq_next.as<uint8_t>(), c->d_hp_next, c->d_tiles_next, c->d_ref_flags_next, len, 0, c->prefetch_stream))
    return PBSIM_FAILED;
  c->pf_src = seq;
  c->pf_len = len;
  return PBSIM_SUCCEEDED;
}
int pbsim_prefetch_reference(pbsim_ctx *c, const uint8_t *seq, int64_t len) {
  return prefetch_reference(c, seq, len, hipMemcpyHostToDevice);
}
int pbsim_prefetch_reference_device(pbsim_ctx *c, const void *seq_device, int64_t len) {
  return prefetch_reference(c, seq_device, len, hipMemcpyDeviceToDevice);
}
static bool take_prefetch(pbsim_ctx *c, const void *seq, int64_t len) {
  const bool hit = c->pf_src && c->pf_src == seq && c->pf_len == len;
  c->pf_src = nullptr;
  return hit;
}

// `prefetched`: the record is already in d_seq_next / d_hp_next, its preparation enqueued on the prefetch stream
static int set_reference_common(pbsim_ctx *c, uint8_t *d_seq, int64_t len, int64_t record_index, bool prefetched = false) {
  if (len < 1) return fail("Reference is too short.");
  if (len > 1000000000LL) return fail("Reference is too long. Acceptable length <= 1000000000.");
  if (c->p.hp_del_bias != 1 && !c->census_done)
    return fail("--hp-del-bias != 1 needs pbsim_add_hp_census() for every record and pbsim_finish_hp_census() first");
  int64_t census[kHpSlots] = {0};
  if (prefetched) {
    if (!finish_prepare(c, c->d_ref_flags_next, c->prefetch_stream, census)) return PBSIM_FAILED;
    std::swap(c->d_seq_own.p, c->d_seq_next.p);
    std::swap(c->d_seq_own.bytes, c->d_seq_next.bytes);
    std::swap(c->d_hp.p, c->d_hp_next.p);
    std::swap(c->d_hp.bytes, c->d_hp_next.bytes);
    d_seq = c->d_seq_own.as<uint8_t>();
  } else if (!prepare_reference(c, d_seq, len, 0, census)) {
    return PBSIM_FAILED;
  }
  note_hp11(c, census);
  c->d_seq = d_seq;
  c->ref_len = len;
  c->unit = record_index;
  for (Slot &sl : c->slots) sl.b_enqueued = sl.b_walked = sl.b_finalized = false;
  return PBSIM_SUCCEEDED;
}

// descriptor of the context's current unit (what pbsim_batch_walk_begin hands to the slot)
extern "C++" RefDesc pbsim::current_ref(const pbsim_ctx *c) {
  RefDesc r;
  r.seq = c->d_seq;
  r.hp = c->d_hp.as<uint8_t>();
  r.len = c->ref_len;
  r.unit = c->unit;
  r.hp_flag = c->seq_hp_flag;
  r.hp11 = c->bias.hp11_seen;
  return r;
}

int pbsim_set_reference(pbsim_ctx *c, const uint8_t *seq, int64_t len, int64_t record_index) {
  if (!c || !seq) return fail("pbsim_set_reference: bad argument");
  NEED_DEVICE(c);
  HIP_OK(hipSetDevice(c->device));
  if (len < 1) return fail("Reference is too short.");
  if (take_prefetch(c, seq, len)) return set_reference_common(c, nullptr, len, record_index, true);
  HIP_OK(c->d_seq_own.ensure((size_t)len + 64));
  HIP_OK(hipMemcpyAsync(c->d_seq_own.p, seq, (size_t)len, hipMemcpyHostToDevice, c->stream));
  HIP_OK(hipMemsetAsync(c->d_seq_own.as<uint8_t>() + len, 0, 64, c->stream));
  return set_reference_common(c, c->d_seq_own.as<uint8_t>(), len, record_index);
}

int pbsim_set_reference_device(pbsim_ctx *c, const void *seq_device, int64_t len, int64_t record_index) {
  if (!c || !seq_device) return fail("pbsim_set_reference_device: bad argument");
  NEED_DEVICE(c);
  HIP_OK(hipSetDevice(c->device));
  if (len < 1) return fail("Reference is too short.");
  if (take_prefetch(c, seq_device, len)) return set_reference_common(c, nullptr, len, record_index, true);
  // the kernels read whole dwords and upper-case in place: keep an owned, padded copy
  HIP_OK(c->d_seq_own.ensure((size_t)len + 64));
  HIP_OK(hipMemcpyAsync(c->d_seq_own.p, seq_device, (size_t)len, hipMemcpyDeviceToDevice, c->stream));
  HIP_OK(hipMemsetAsync(c->d_seq_own.as<uint8_t>() + len, 0, 64, c->stream));
  return set_reference_common(c, c->d_seq_own.as<uint8_t>(), len, record_index);
}

extern "C++" int64_t pbsim::quota_of(const pbsim_ctx *c, int64_t ref_len) {  // pbsim.cpp:705
  return (int64_t)(long long)(c->p.depth * (double)ref_len);
}
int64_t pbsim_unit_quota(pbsim_ctx *c) { return c ? quota_of(c, c->ref_len) : 0; }

int64_t pbsim_batch_capacity(pbsim_ctx *c) { return c ? batch_capacity(c) : 0; }

int pbsim_reset_stats(pbsim_ctx *c) {  // init_sim_res, pbsim.cpp:1437-1445 + 3626-3631
  if (!c) return fail("bad argument");
  c->st.reset();
  return PBSIM_SUCCEEDED;
}

int pbsim_batch_walk(pbsim_ctx *c, int64_t first_read, int64_t n_reads, int64_t truncate_remaining,
                     int64_t *pass0_bases) {
  if (!pbsim_batch_walk_begin(c, first_read, n_reads, truncate_remaining)) return PBSIM_FAILED;
  return pbsim_batch_walk_end(c, pass0_bases);
}

int pbsim_select_slot(pbsim_ctx *c, int slot) {
  if (!c || slot < 0 || slot >= kMaxSlots) return fail("pbsim_select_slot: slot out of range");
  c->cur = slot;
  return PBSIM_SUCCEEDED;
}

int pbsim_slot_count(void) { return kMaxSlots; }

int pbsim_batch_walk_begin(pbsim_ctx *c, int64_t first_read, int64_t n_reads, int64_t truncate_remaining) {
  if (!c) return fail("pbsim_batch_walk: bad argument");
  if (!c->d_seq)
    return fail(c->p.strategy != PBSIM_STRATEGY_WGS ? "no transcripts/templates set (pbsim_set_transcripts, pbsim_set_templates)"
                                                    : "no reference set (pbsim_set_reference)");
  return walk_begin(c, current_ref(c), first_read, n_reads, truncate_remaining);
}

// Reads of at least this length are walked by a whole wave each (k_walk_errhmm_coop) instead of one lane.  A lane needs
// ~0.4 us per column whatever else the GPU does, so a launch lasts as long as its longest read (~30 ms for the default
// length distribution) unless its bulk lasts longer; the wave walker takes a read through in ~40 cycles per column but
// moves a third of the lane walker's columns per second when the GPU is full (100 against 286 G columns/s).  The split that
// ends both at the same time grows with the batch (same-box sweep, one launch at a time, profiles/r02z_coop_split.txt:
// 50 k reads 34 -> 5.8 ms from half a mean length, 100 k 34 -> 8.1 ms from one, 200 k 36 -> 12 ms from 1.5, 450 k
// 36 -> 19 ms from 2-3; the job pipeline, three rounds of 450 k reads in flight, is flat from 2.5 to 4).  A launch alone
// is not what a job runs, though: its rounds overlap each other and, when it delivers, the compression, whose workgroups
// the wave walker's keep off the CUs (four of them hold 118 of a CU's 160 KB of LDS) -- jobs want the split later than a
// lone launch does.  Round 4, after the wave walker's step was cut (profiles/r04_coop_split_ab.txt): mean length x reads /
// 80 k, between 0.5 and 4 (round 2: / 150 k) -- configs[1] in HBM 319 -> 305-309 ms, delivered 1137 -> 1108-1134 ms, a rank
// of eight 172 -> 167 ms (configs[4]: 471 -> 452 ms); / 60 k and later lose again (the rounds then wait for their longest
// lanes).  Small batches (top-up rounds, the truncated tail reads a record's completion waits for) go to the wave walker
// entirely, and a batch of a million reads hides its longest lane behind its own bulk.
// A multiple of 256 (the sort's length bucket).  PBSIM_COOP_LEN overrides: -1 never, n >= 0 that length.
constexpr int kCoopWorkgroups = 4096 / kCoopWaves, kCoopSmallBatch = 20000, kQCoopSmallBatch = 20000, kCoopHugeBatch = 1000000;  // 4096 persistent waves
static int32_t coop_min_len(const pbsim_ctx *c, int64_t n_reads, bool hp_flag) {
  const bool qs = c->p.method == PBSIM_METHOD_QS;
  if (qs) {  // k_walk_qshmm_coop: moduli of 100, the hp == 11 flag in the sequence bytes (default --hp-del-bias), <= 63 states
    if (!c->qct.all_rv_100 || !hp_flag || c->qct.smax > kQCoopMaxStates) return INT32_MAX;
  } else if (c->p.method != PBSIM_METHOD_ERR || !c->ect.all_rv_1000 || c->ect.smax > kCoopMaxStates) {
    return INT32_MAX;
  }
  const char *env = getenv("PBSIM_COOP_LEN");
  const int64_t n_tasks = n_reads * c->p.pass_num;
  const char *sr = getenv("PBSIM_COOP_SPLIT_READS");  // experiment knob: the batch size at which the split is one mean length
  const double split_reads = sr && atof(sr) > 0 ? atof(sr) : 80000.0;
  int64_t len = (int64_t)(std::min(4.0, std::max(0.5, (double)n_tasks / split_reads)) * c->hdr.mean_len);
  // QSHMM (round 5: its wave walker repairs its chains instead of re-walking them and runs four waves per SIMD, 34 -> 41 G
  // columns/s alone): the same kind of split -- the long tasks by waves beside the lane walk of the rest.  One launch at a time,
  // QSHMM-RSII x 10 passes (tools/walk_solo.py, profiles/r05_qshmm_split.txt): 60 000 tasks 29.6 ms by lanes, 20.7 by waves,
  // 14.9 split at two mean lengths; 100 000: 31.5 / 28.1 / 16.7 at three; 200 000: 32.2 / 47.5 / 20.6 at three; 400 000: 33.2
  // lanes, 25.1 at four, 26.1 at five.  (Rounds 3-4 kept every task of a QSHMM batch with one walker: the wave walker was
  // too slow for the long tasks of a large batch.)
  // In a JOB whose rounds are large (configs[2] on one GPU: 540 000 tasks a round, three rounds in flight) the wave walker's
  // register-heavy workgroups beside three lane walks cost more than the long lanes they remove -- 167 G subread bases/s by
  // lanes only, 153 split at four mean lengths, 159 at five, 167 at seven -- so batches beyond 250 000 tasks stay with the lane
  // walker; below that (the rounds of a rank of eight, top-up rounds) the split is a lone launch's.
  if (qs) len = n_tasks > 250000 ? -1 : (int64_t)(std::min(3.5, std::max(0.5, 0.5 + (double)n_tasks / 40000.0)) * c->hdr.mean_len);
  if (n_tasks <= (qs ? kQCoopSmallBatch : kCoopSmallBatch)) len = 0;
  if (n_tasks >= kCoopHugeBatch) len = -1;
  if (env) len = atoll(env);
  if (len < 0) return INT32_MAX;
  len = (len + 255) / 256 * 256;
  return len >= (int64_t)kLenBuckets << kLenShift ? INT32_MAX : (int32_t)len;
}

// header draw -> bucketing -> walk -> pass-0 prefix of reads [first_read, first_read + n_reads) of `ref`, on the selected slot
extern "C++" int pbsim::walk_begin(pbsim_ctx *c, const RefDesc &ref, int64_t first_read, int64_t n_reads, int64_t truncate_remaining,
                                   bool chain, double factor) {
  if (factor <= 0) factor = scratch_factor_of(c);
  if (!c || n_reads < 1 || first_read < 1) return fail("pbsim_batch_walk: bad argument");
  if (chain && (truncate_remaining < 0 || c->p.strategy != PBSIM_STRATEGY_WGS)) return fail("internal: a chain of truncated reads needs a quota");
  NEED_DEVICE(c);
  if (c->s().b_enqueued) return fail("pbsim_batch_walk_begin: this slot still has a batch in flight (pbsim_batch_walk_end)");
  if (c->p.method == PBSIM_METHOD_SAMPLE) return fail("the sampling method runs through pbsim_simulate_sample");
  const bool trans = c->p.strategy != PBSIM_STRATEGY_WGS;  // trans and templ share the unit machinery
  if (!ref.seq) return fail("no reference set (pbsim_set_reference)");
  if (trans && (truncate_remaining >= 0 || first_read + n_reads - 1 > c->trans_reads))
    return fail("pbsim_batch_walk: read range outside the transcript set");
  if (truncate_remaining >= 0 && n_reads != 1 && !chain) return fail("a truncated batch holds exactly one read");
  if (first_read + n_reads > 0xffffffffLL) return fail("read index exceeds 32 bits");
  HIP_OK(hipSetDevice(c->device));
  if (!ensure_header_tables(c) || !ensure_class_tables(c)) return PBSIM_FAILED;
  if (c->p.method == PBSIM_METHOD_QS && !ensure_qs_tabs(c, ref.hp11)) return PBSIM_FAILED;
  c->s().ref = ref;
  const int P = c->p.pass_num;
  const int ncls = ncls_of(c);
  if (ncls > kMaxClasses) return fail("too many accuracy classes");
  const int64_t n_tasks = n_reads * P;
  if (n_tasks > 0x7fffff00LL) return fail("batch too large");
  const int64_t slots_max = ((n_tasks + (int64_t)ncls * (kWG - 1)) / kWG + 1) * kWG;
  const int64_t waves_max = slots_max / 64;
  const size_t nbins = (size_t)ncls * kLenBuckets;

  HIP_OK(c->s().d_flags.ensure(sizeof(DeviceFlags)));
  HIP_OK(c->s().d_rawlen.ensure(n_reads * 4));
  HIP_OK(c->s().d_len.ensure(n_reads * 4));
  HIP_OK(c->s().d_off.ensure(n_reads * 4));
  HIP_OK(c->s().d_acc.ensure(n_reads));
  HIP_OK(c->s().d_hist.ensure(nbins * kBinPad * 4));
  HIP_OK(c->s().d_bin_start.ensure(nbins * 4));
  HIP_OK(c->s().d_bin_cursor.ensure(nbins * kBinPad * 4));
  HIP_OK(c->s().d_class_start.ensure((2 * ncls + 2) * 4));  // class_start[ncls + 1] | coop_end[ncls]
  HIP_OK(c->s().d_task_of_slot.ensure(slots_max * 4));
  HIP_OK(c->s().d_slot_of_task.ensure(n_tasks * 4));
  HIP_OK(c->s().d_wave_cap.ensure(waves_max * 4));
  HIP_OK(c->s().d_wave_off.ensure(waves_max * 8));
  HIP_OK(c->s().d_wg_tmp.ensure((size_t)(kLenBuckets + 1) * 2 * 4));
  HIP_OK(c->s().d_wg_order.ensure((size_t)(slots_max / kWG) * 4));
  HIP_OK(c->s().d_out_len.ensure(n_tasks * 4));
  HIP_OK(c->s().d_maf_len.ensure(n_tasks * 4));
  HIP_OK(c->s().d_nsub.ensure(n_tasks * 4));
  HIP_OK(c->s().d_nins.ensure(n_tasks * 4));
  HIP_OK(c->s().d_ndel.ensure(n_tasks * 4));
  HIP_OK(c->s().d_qsum.ensure(n_tasks * 8));
  HIP_OK(c->s().d_cum.ensure((n_reads + 1) * 8));
  HIP_OK(c->s().d_scan_tmp.ensure((n_tasks / 1024 + 8) * 8));
  if (chain) {
    HIP_OK(c->s().d_chain.ensure(64));
    HIP_OK(c->s().d_chain_mask.ensure(slots_max * 4));
  }
  // The pool is the budget, unless this batch cannot need that much: a wave's rows hold at most 2 * Lmax + pad columns
  // (Lmax = the longest read the header can draw), so a handful of reads (the truncated tail reads) gets by with little.
  double lmax = (double)std::min<int64_t>(c->p.len_max, std::max<int64_t>(ref.len, 1));
  if (truncate_remaining >= 0) lmax = std::min(lmax, (double)std::max<int64_t>(truncate_remaining, c->p.len_min));  // (pbsim.cpp:3795-3800)
  // (only waves that hold a task take scratch: at most one per task)
  const double worst = (double)std::min<int64_t>(waves_max, n_tasks) * regions_of(c) * (factor * lmax + kScratchPad + 64) * 64.0;
  const int64_t pool = std::max<int64_t>((int64_t)c->s().d_scratch.bytes - (int64_t)kScratchSlack,
                                         (int64_t)std::min<double>((double)c->scratch_budget, worst));
  HIP_OK(c->s().d_scratch.ensure((size_t)pool + kScratchSlack, true));

  HIP_OK(hipEventRecord(c->s().ev0, c->s().stream));
  HIP_OK(hipMemsetAsync(c->s().d_flags.p, 0, sizeof(DeviceFlags), c->s().stream));
  DeviceFlags *flags = c->s().d_flags.as<DeviceFlags>();

  HeaderArgs h;
  h.seed = c->p.seed;
  h.unit = (uint32_t)ref.unit;
  h.first_read = first_read;
  h.n_reads = n_reads;
  h.prob2len = c->d_prob2len.as<int32_t>();
  h.len_rv = c->hdr.len_rv;
  h.prob2acc = c->d_prob2acc.as<uint8_t>();
  h.acc_rv = c->hdr.acc_rv;
  h.ref_len = ref.len;
  h.len_min = c->p.len_min;
  h.truncate_remaining = truncate_remaining;
  h.rawlen = c->s().d_rawlen.as<int32_t>();
  h.len = c->s().d_len.as<int32_t>();
  h.off = c->s().d_off.as<int32_t>();
  h.acc = c->s().d_acc.as<uint8_t>();
  h.read_unit = nullptr;
  h.is_templ = 0;
  h.max_rawlen = trans ? nullptr : &flags->max_rawlen;
  if (trans) {
    h.is_templ = c->p.strategy == PBSIM_STRATEGY_TEMPL;
    h.read_unit = c->d_read_unit.as<int32_t>() + (first_read - 1);
    h.unit_len = c->d_unit_len.as<int64_t>();
    h.unit_rank = c->d_unit_rank.as<int32_t>();
    h.off_table = c->d_off_table.as<int32_t>();
    h.ssp = c->d_ssp.as<uint8_t>();
    h.ssp_rv = c->d_ssp_rv.as<int32_t>();
    launch_header_trans(h, c->s().stream);
  } else {
    launch_header_wgs(h, c->s().stream);
  }

  SortArgs s;
  s.n_reads = n_reads;
  s.pass_num = P;
  s.acc_lo = c->hdr.acc_lo;
  s.ncls = ncls;
  s.len = h.len;
  s.acc = h.acc;
  s.hist = c->s().d_hist.as<int32_t>();
  s.bin_start = c->s().d_bin_start.as<int32_t>();
  s.bin_cursor = c->s().d_bin_cursor.as<int32_t>();
  s.class_start = c->s().d_class_start.as<int32_t>();
  s.coop_end = s.class_start + ncls + 1;
  int32_t coop_len = coop_min_len(c, n_reads, ref.hp_flag);
  // (a chain's layout stands for upper bounds of its reads' lengths, its walks see the lengths they end up with: a split by
  // length would put a read on one walker's side of the layout and the other walker's side of the walk -- all or none)
  if (chain && coop_len != INT32_MAX) coop_len = 0;
  s.coop_bucket = coop_len == INT32_MAX ? kLenBuckets : coop_len >> kLenShift;
  s.coop_classes = 0;
  if (coop_len != INT32_MAX)  // verbatim classes (ERRHMM) stay with the lane walker
    for (int i = 0; i < ncls; i++) {
      uint32_t mode;  // hdr[2]: ERRHMM mode | QSHMM has_model
      if (c->p.method == PBSIM_METHOD_ERR) {
        memcpy(&mode, c->ect.blob.data() + (size_t)i * c->ect.stride + 8, 4);
        if (mode != kModeVerbatim) s.coop_classes |= 1ull << i;
      } else {
        memcpy(&mode, c->qct.blob.data() + (size_t)i * c->qct.stride + 8, 4);
        (void)mode;  // classes without a model (quality from the accuracy alone) are walked by waves too since round 4
        s.coop_classes |= 1ull << i;
      }
    }
  s.task_of_slot = c->s().d_task_of_slot.as<int32_t>();
  s.slot_of_task = c->s().d_slot_of_task.as<int32_t>();
  s.wave_cap = c->s().d_wave_cap.as<int32_t>();
  s.wave_off = c->s().d_wave_off.as<int64_t>();
  s.n_slots_max = slots_max;
  s.wg_hist = c->s().d_wg_tmp.as<int32_t>();
  s.wg_start = s.wg_hist + (kLenBuckets + 1);
  s.wg_order = c->s().d_wg_order.as<int32_t>();
  s.regions = regions_of(c);
  s.cap_q8 = (int32_t)std::min(512.0, ceil(factor * 256.0));
  s.scratch_bytes = std::min<int64_t>(pool, c->scratch_budget);
  s.flags = flags;
  launch_task_sort(s, c->s().stream);

  WalkArgs w;
  memset(&w, 0, sizeof w);
  w.seed = c->p.seed;
  w.unit = trans ? 0u : (uint32_t)ref.unit;
  w.first_read = first_read;
  w.pass_num = P;
  w.ncls = ncls;
  w.ref.seq = ref.seq;
  w.ref.hp = ref.hp;
  w.ref.len = ref.len;
  w.len = h.len;
  w.off = h.off;
  if (trans) {
    w.read_base = c->d_read_base.as<int64_t>() + (first_read - 1);
    w.read_minus = c->d_read_minus.as<uint8_t>() + (first_read - 1);
  }
  w.cls_blob = c->d_cls.as<uint8_t>();
  w.class_start = s.class_start;
  w.task_of_slot = s.task_of_slot;
  w.wg_order = s.wg_order;
  w.mean_len = (int32_t)c->hdr.mean_len;
  w.cap_q8 = s.cap_q8;
  w.coop_min_len = coop_len;
  {
    // the wave walkers' units: drawn from a counter in batches of 16 k - 250 k tasks, where the wave walk decides the batch's
    // duration and a workgroup gets several units (default split, walk kernels alone: 20 000 reads 3.17 -> 2.59 ms, 50 000
    // 5.91 -> 4.27, 100 000 8.57 -> 7.59, 200 000 12.6 -> 12.2 -- the rounds of a job on 4-8 ranks); dealt round-robin
    // otherwise: beside the lane walk of a full round there is no gain (400 000 reads 17.0 -> 17.2 ms, whole job -1 %: the
    // draw costs two barriers per unit), and one unit per workgroup has nothing to balance (QSHMM 500 reads x 10 passes
    // 12.6 -> 13.3 ms).  tools/closed_ab/coop_dynamic_ab.sh; PBSIM_COOP_DYNAMIC=0/1 forces either.
    const char *cd = getenv("PBSIM_COOP_DYNAMIC");
    w.coop_dynamic = cd ? atoi(cd) == 1 : (coop_len != INT32_MAX && n_tasks >= 4LL * kCoopWaves * kCoopWorkgroups && n_tasks <= 250000);
    if (chain) w.coop_dynamic = 0;  // (the steps of a chain share the batch's flags: the units' counter is not theirs to draw from)
  }
  w.coop_end = s.coop_end;
  w.wave_cap = s.wave_cap;
  w.wave_off = s.wave_off;
  w.scratch = c->s().d_scratch.as<uint8_t>();
  w.out_len = c->s().d_out_len.as<int32_t>();
  w.maf_len = c->s().d_maf_len.as<int32_t>();
  w.nsub = c->s().d_nsub.as<int32_t>();
  w.nins = c->s().d_nins.as<int32_t>();
  w.ndel = c->s().d_ndel.as<int32_t>();
  w.qsum = c->s().d_qsum.as<double>();
  w.flags = flags;
  // the walk of a batch goes to the slot's LOW priority stream; a single truncated tail read is the opposite case -- one
  // workgroup whose latency a record's completion waits for -- and must not queue behind the pending workgroups of the batches
  hipStream_t ws = (truncate_remaining >= 0) ? c->s().stream : c->s().walk_stream;
  HIP_OK(hipEventRecord(c->s().ev_prep, c->s().stream));
  HIP_OK(hipStreamWaitEvent(ws, c->s().ev_prep, 0));
  HIP_OK(hipEventRecord(c->s().ev1, ws));
  // A chain of truncated reads (kernels.h ChainState): the header and the layout above stand for upper bounds -- every read
  // at most what is left of the quota NOW -- and the steps below re-draw and walk one read after the other, all on this one
  // stream, without a host round trip in between.  The walks of step k see a task map that holds read k's tasks only.
  ChainState *d_chain = chain ? c->s().d_chain.as<ChainState>() : nullptr;
  if (chain) launch_chain_init(d_chain, truncate_remaining, ws);
  const int n_steps = chain ? (int)n_reads : 1;
  for (int step = 0; step < n_steps; step++) {
  if (chain) {
    launch_chain_prepare(h, step, P, s.task_of_slot, c->s().d_chain_mask.as<int32_t>(), slots_max, d_chain, ws);
    w.task_of_slot = c->s().d_chain_mask.as<int32_t>();
  }
  if (c->p.method == PBSIM_METHOD_ERR) {
    w.stride = c->ect.stride;
    w.rows_off = c->ect.rows_off;
    w.emis_off = c->ect.emis_off;
    w.init_off = c->ect.init_off;
    w.tran_off = c->ect.tran_off;
    if (coop_len != INT32_MAX) {
      // the long reads first, so that their workgroups are resident before the lane walk fills the CUs; beside a batch
      // on a stream of their own, a lone tail read simply in front of the (then empty) lane walk
      // 512 persistent workgroups of eight waves (two per CU: 63 KB of LDS) beside a lane walk and the compression, which
      // need the rest of the CUs' LDS; a batch walked by waves only (a small batch, a record's tail) gets as many as the GPU
      // holds at once: three per CU = the six waves per SIMD the kernel is compiled for.  Round 4 (profiles/
      // r04_coop_split_ab.txt): four-wave workgroups, four per CU, held 118 KB for the same 4096 waves -- a rank of eight
      // 167 -> 164 ms with eight-wave ones, 100 000 reads by waves only 139 (1024 x 4 waves) -> 156 G columns/s (768 x 8).
      const char *cw = getenv("PBSIM_COOP_WG");  // experiment knob
      const bool waves_only = coop_len == 0;
      if (c->coop_wg_errhmm[ref.hp_flag] == 0) {
        hipDeviceProp_t pr;
        int cus = 256;
        if (hipGetDeviceProperties(&pr, c->device) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount;
        const int res = walk_errhmm_coop_resident(c->ect.stride + 512 + 1024, ref.hp_flag);
        c->coop_wg_errhmm[ref.hp_flag] = std::max(kCoopWorkgroups, cus * std::min(res, 8));
      }
      const int n_wg = (int)std::max<int64_t>(1, std::min<int64_t>(cw && atoi(cw) > 0 ? atoi(cw) : waves_only ? c->coop_wg_errhmm[ref.hp_flag] : kCoopWorkgroups,
                                                                   (n_tasks + kCoopWaves - 1) / kCoopWaves));
      hipStream_t cs = (ws == c->s().walk_stream && c->s().coop_stream) ? c->s().coop_stream : ws;
      if (cs != ws) HIP_OK(hipStreamWaitEvent(cs, c->s().ev_prep, 0));
      launch_walk_errhmm_coop(w, n_wg, c->ect.stride + 512 + 1024, ref.hp_flag, cs);
      c->prof_wave_launches++;
      if (cs != ws) HIP_OK(hipEventRecord(c->s().ev_coop, cs));
      // (every read on the wave walker -- a small batch of a model without verbatim classes, e.g. a truncated tail read --
      // leaves the lane walker nothing to do: no empty launch, and the kernel's profile holds its bulk launches only)
      const bool lanes_idle = coop_len == 0 && s.coop_classes == (ncls >= 64 ? ~0ull : (1ull << ncls) - 1);
      if (!lanes_idle)
        launch_walk_errhmm(w, slots_max, c->ect.stride + 512 + 1024, c->ect.all_rv_1000, ref.hp_flag, ws, c->walk_lds_kb);
      if (cs != ws) HIP_OK(hipStreamWaitEvent(ws, c->s().ev_coop, 0));
    } else {
      launch_walk_errhmm(w, slots_max, c->ect.stride + 512 + 1024, c->ect.all_rv_1000, ref.hp_flag, ws, c->walk_lds_kb);
    }
  } else {
    w.stride = c->qct.stride;
    w.rv_off = c->qct.rv_off;
    w.init_off = c->qct.init_off;
    w.tran_off = c->qct.tran_off;
    w.emis_off = c->qct.emis_off;
    w.freq_off = c->qct.freq_off;
    const uint8_t *t = c->d_qs_tabs_v[ref.hp11].as<uint8_t>();
    w.sub_thre = reinterpret_cast<const uint32_t *>(t);
    w.ins_thre = reinterpret_cast<const uint32_t *>(t + 94 * 4);
    w.del_thr = reinterpret_cast<const uint32_t *>(t + 94 * 8);
    w.qprob = reinterpret_cast<const double *>(t + 94 * 8 + 94 * 48);
    bool lanes_idle = false;
    if (coop_len != INT32_MAX) {  // the wave walker first, beside a batch on a stream of its own (as for ERRHMM above)
      const char *cw = getenv("PBSIM_COOP_WG");
      // (k_walk_qshmm_coop's workgroups hold four waves: 1024 of them are the 4096 persistent waves)
      const int n_wg = (int)std::max<int64_t>(1, std::min<int64_t>(cw && atoi(cw) > 0 ? atoi(cw) : 4096 / (kWG / 64), (n_tasks + 3) / 4));
      hipStream_t cs = (ws == c->s().walk_stream && c->s().coop_stream) ? c->s().coop_stream : ws;
      if (cs != ws) HIP_OK(hipStreamWaitEvent(cs, c->s().ev_prep, 0));
      launch_walk_qshmm_coop(w, n_wg, c->qct.stride + 512 + 1024, cs);
      launch_qshmm_coop_qsum(w, slots_max, cs);
      c->prof_wave_launches++;
      if (cs != ws) HIP_OK(hipEventRecord(c->s().ev_coop, cs));
      lanes_idle = coop_len == 0 && s.coop_classes == (ncls >= 64 ? ~0ull : (1ull << ncls) - 1);
      if (!lanes_idle)
        launch_walk_qshmm(w, slots_max, c->qct.stride + 1536 + 96 * 8 + 94 * 48 + 94 * 8 + 94 * 32, c->qct.all_rv_100,
                          ref.hp_flag, ws, c->walk_lds_kb);
      if (cs != ws) HIP_OK(hipStreamWaitEvent(ws, c->s().ev_coop, 0));
    } else {
      launch_walk_qshmm(w, slots_max, c->qct.stride + 1536 + 96 * 8 + 94 * 48 + 94 * 8 + 94 * 32, c->qct.all_rv_100,
                        ref.hp_flag, ws, c->walk_lds_kb);
    }
    (void)lanes_idle;
  }
  if (chain) launch_chain_update(step, P, w.out_len, d_chain, ws);
  }  // steps
  HIP_OK(hipEventRecord(c->s().ev2, ws));
  HIP_OK(hipStreamWaitEvent(c->s().stream, c->s().ev2, 0));
  launch_gather_pass0_scan(w.out_len, n_reads, P, c->s().d_cum.as<int64_t>(), c->s().d_scan_tmp.as<int64_t>(),
                           &flags->sums[0], c->s().stream);
  HIP_OK(hipEventRecord(c->s().ev3, c->s().stream));
  HIP_OK(hipGetLastError());
  c->s().b_first = first_read;
  c->s().b_n = n_reads;
  c->s().b_slots_max = slots_max;
  c->s().b_chain = chain;
  c->s().b_trunc = truncate_remaining;
  c->s().b_factor = factor;
  c->s().b_truncated = truncate_remaining >= 0 || trans;  // trans has no quota: every read is final
  c->s().b_enqueued = true;
  c->s().b_walked = false;
  c->s().b_finalized = false;
  c->s().stats_fetched = false;
  return PBSIM_SUCCEEDED;
}

int pbsim_batch_walk_end(pbsim_ctx *c, int64_t *pass0_bases) {
  if (!c) return fail("pbsim_batch_walk_end: bad argument");
  NEED_DEVICE(c);
  if (!c->s().b_enqueued) return fail("pbsim_batch_walk_end: no batch was begun on this slot");
  HIP_OK(hipSetDevice(c->device));
  c->s().b_enqueued = false;
  DeviceFlags f;
  if (!read_flags(c, &f)) return PBSIM_FAILED;
  float ms = 0;
  // the profile is about the walk kernel at work: a launch that carries a single truncated tail read (latency of one lane,
  // no bytes to speak of) is counted apart
  const bool bulk = c->s().b_n > 1 && !c->s().b_chain;
  if (hipEventElapsedTime(&ms, c->s().ev1, c->s().ev2) == hipSuccess) (bulk ? c->prof_walk_ms : c->prof_tail_ms) += ms;
  if (!bulk) c->prof_tail_launches++;
  if (bulk && c->ev_prof_base) {
    float a = 0, b = 0;
    if (hipEventElapsedTime(&a, c->ev_prof_base, c->s().ev1) == hipSuccess &&
        hipEventElapsedTime(&b, c->ev_prof_base, c->s().ev2) == hipSuccess)
      c->prof_intervals.emplace_back(a, b);
  }
  if (hipEventElapsedTime(&ms, c->s().ev0, c->s().ev3) == hipSuccess) c->prof_total_ms += ms;
  if (bulk) c->prof_walk_launches++;
  if (f.error & kErrScratchBudget) {
    char buf[160];
    snprintf(buf, sizeof buf, "scratch budget exceeded: batch needs %lld bytes, pool holds %lld",
             (long long)f.scratch_need, (long long)c->scratch_budget);
    return fail(buf);
  }
  // what the batch's reads needed of their rows: the next batches are laid out for the largest need seen so far + 0.08
  if (f.need_q10) {
    c->need_seen = std::max(c->need_seen, (double)f.need_q10 / 1024.0);
    if (!c->scratch_factor_fixed) c->scratch_factor = std::min(2.0, std::max(1.0, c->need_seen + 0.08));
  }
  if (f.error & kErrScratchOverflow) {
    if (c->s().b_factor < 2.0) {
      // a read ran out of row: the batch again, laid out with the reference's own bound (and the factor follows what it needed)
      c->rewalked_batches++;
      if (!c->scratch_factor_fixed) c->scratch_factor = std::min(2.0, c->scratch_factor + 0.25);
      Slot &sl = c->s();
      if (!walk_begin(c, sl.ref, sl.b_first, sl.b_n, sl.b_trunc, sl.b_chain, 2.0)) return PBSIM_FAILED;
      return pbsim_batch_walk_end(c, pass0_bases);
    }
    return fail("a read produced more MAF columns than 2*len+64 (the reference's buffers are 2*len_max+1)");
  }
  c->s().b_walked = true;
  c->s().b_pass0 = f.sums[0];
  c->s().b_max_raw = (int64_t)f.max_rawlen;
  if (pass0_bases) *pass0_bases = f.sums[0];
  return PBSIM_SUCCEEDED;
}

int pbsim_batch_fetch_lengths(pbsim_ctx *c, int32_t *rawlen, int32_t *len, int32_t *out_len_pass0) {
  if (!c || !c->s().b_walked) return fail("pbsim_batch_fetch_lengths: no walked batch on this slot");
  NEED_DEVICE(c);
  HIP_OK(hipSetDevice(c->device));
  Slot &sl = c->s();
  const size_t n = (size_t)sl.b_n;
  if (rawlen) HIP_OK(hipMemcpyAsync(rawlen, sl.d_rawlen.p, n * 4, hipMemcpyDeviceToHost, sl.stream));
  if (len) HIP_OK(hipMemcpyAsync(len, sl.d_len.p, n * 4, hipMemcpyDeviceToHost, sl.stream));
  if (out_len_pass0)  // out_len is per task (read-major, pass minor): every pass_num-th value
    HIP_OK(hipMemcpy2DAsync(out_len_pass0, 4, sl.d_out_len.p, (size_t)c->p.pass_num * 4, 4, n, hipMemcpyDeviceToHost, sl.stream));
  HIP_OK(hipStreamSynchronize(sl.stream));
  return PBSIM_SUCCEEDED;
}

// Reads a chain may hold (<= kChainReads): the layout of a chain stands for upper bounds -- every read as long as what is left of
// the quota now -- and in the worst case every read sits in a scratch block of its own (64 lanes wide whatever it holds), so
// a small pool (tests, PBSIM_SCRATCH_MB) takes fewer steps per chain; one read is what a single truncated batch needs.
extern "C++" int pbsim::chain_reads_for(const pbsim_ctx *c, int64_t ref_len, int64_t remaining) {
  const double lub = (double)std::min<int64_t>(std::min<int64_t>(c->p.len_max, std::max<int64_t>(ref_len, 1)),
                                               std::max<int64_t>(remaining, c->p.len_min));
  const double per_block = (double)regions_of(c) * 64.0 * (scratch_factor_of(c) * lub + kScratchPad + 64.0);
  const int64_t fit = (int64_t)((double)c->scratch_budget / (1.05 * per_block));
  return (int)std::max<int64_t>(1, std::min<int64_t>(kChainReads, fit));
}

// The chain of truncated reads begun with walk_begin(.., chain) on the selected slot: waits for its steps, says how far it
// got (reads made -- all final --, their pass-0 bases, whether the quota is reached) and emits the text of the reads made.
extern "C++" int pbsim::chain_end_finalize(pbsim_ctx *c, int64_t len_total_before, pbsim_batch_info *out) {
  if (!c || !c->s().b_enqueued || !c->s().b_chain) return fail("internal: no chain of truncated reads on this slot");
  if (!pbsim_batch_walk_end(c, nullptr)) return PBSIM_FAILED;
  Slot &sl = c->s();
  ChainState *pin = reinterpret_cast<ChainState *>((char *)sl.h_flags.p + sizeof(DeviceFlags) + 16);  // pinned (read_flags)
  HIP_OK(hipMemcpyAsync(pin, sl.d_chain.p, sizeof(ChainState), hipMemcpyDeviceToHost, sl.stream));
  HIP_OK(hipStreamSynchronize(sl.stream));
  const ChainState cs = *pin;
  pbsim_batch_info bi;
  memset(&bi, 0, sizeof bi);
  bi.first_read = sl.b_first;
  bi.n_reads = sl.b_n;
  bi.n_final = cs.made;
  bi.len_total_after = len_total_before + cs.total;
  bi.quota_reached = cs.done != 0;
  bi.need_truncated_read = !cs.done;
  sl.b_info = bi;
  if (!finalize_text(c, &bi)) return PBSIM_FAILED;
  *out = bi;
  return PBSIM_SUCCEEDED;
}

static void fill_text_args(pbsim_ctx *c, TextArgs *t, int64_t n_emit) {
  memset(t, 0, sizeof *t);
  t->first_read = c->s().b_first;
  t->n_reads = n_emit;
  t->pass_num = c->p.pass_num;
  t->is_wgs = c->p.strategy == PBSIM_STRATEGY_WGS;
  t->is_qs = has_quality_row(c);
  t->unit = (uint32_t)c->s().ref.unit;
  t->ref_len = c->s().ref.len;
  t->len = c->s().d_len.as<int32_t>();
  t->off = c->s().d_off.as<int32_t>();
  t->out_len = c->s().d_out_len.as<int32_t>();
  t->maf_len = c->s().d_maf_len.as<int32_t>();
  t->slot_of_task = c->s().d_slot_of_task.as<int32_t>();
  t->task_of_slot = c->s().d_task_of_slot.as<int32_t>();
  t->row_dst = c->s().d_row_dst.as<int64_t>();
  t->wave_cap = c->s().d_wave_cap.as<int32_t>();
  t->wave_off = c->s().d_wave_off.as<int64_t>();
  t->scratch = c->s().d_scratch.as<uint8_t>();
  t->read_text_len = c->s().d_rt_len.as<int64_t>();
  t->maf_text_len = c->s().d_mt_len.as<int64_t>();
  t->read_text_off = t->read_text_len;
  t->maf_text_off = t->maf_text_len;
  t->id_prefix_len = (int)strlen(c->p.id_prefix);
  memcpy(t->id_prefix, c->p.id_prefix, sizeof t->id_prefix);
  t->rq_len = snprintf(t->rq_text, sizeof t->rq_text, "%f", c->p.accuracy_mean);  // pbsim.cpp:4027
  t->bam = c->bam_output && c->p.pass_num > 1;
  {
    const float f = strtof(t->rq_text, nullptr);
    memcpy(&t->rq_bits, &f, 4);
  }
  if (c->p.strategy != PBSIM_STRATEGY_WGS) {
    t->name_pad3 = c->p.strategy == PBSIM_STRATEGY_TEMPL;
    t->read_unit = c->d_read_unit.as<int32_t>() + (c->s().b_first - 1);
    t->read_minus = c->d_read_minus.as<uint8_t>() + (c->s().b_first - 1);
    t->unit_len = c->d_unit_len.as<int64_t>();
    t->unit_names = c->d_unit_names.as<char>();
  }
}

// First half of pbsim_batch_finalize: the quota cut (which reads of the batch are final, len_total behind them).
extern "C++" int pbsim::finalize_cut(pbsim_ctx *c, int64_t len_total_before, pbsim_batch_info *out) {
  if (!c || !c->s().b_walked) return fail("pbsim_batch_finalize: no walked batch");
  NEED_DEVICE(c);
  HIP_OK(hipSetDevice(c->device));
  DeviceFlags *flags = c->s().d_flags.as<DeviceFlags>();
  const int64_t quota = quota_of(c, c->s().ref.len);
  // the sampling method never truncates a read: it only stops before the first read that starts at or past the quota
  launch_quota_cut(c->s().d_cum.as<int64_t>(),
                   c->p.method == PBSIM_METHOD_SAMPLE ? nullptr : c->s().d_rawlen.as<int32_t>(), c->s().b_n, len_total_before, quota,
                   c->s().b_truncated ? 1 : 0, flags, c->s().stream);
  DeviceFlags f;
  if (!read_flags(c, &f)) return PBSIM_FAILED;
  const int64_t n_final = f.n_final;
  int64_t cum_final = c->s().b_pass0;
  if (n_final < c->s().b_n) {
    int64_t *pin = reinterpret_cast<int64_t *>((char *)c->s().h_flags.p + sizeof(DeviceFlags));  // pinned (read_flags)
    HIP_OK(hipMemcpyAsync(pin, c->s().d_cum.as<int64_t>() + n_final, 8, hipMemcpyDeviceToHost, c->s().stream));
    HIP_OK(hipStreamSynchronize(c->s().stream));
    cum_final = *pin;
  }
  pbsim_batch_info bi;
  memset(&bi, 0, sizeof bi);
  bi.first_read = c->s().b_first;
  bi.n_reads = c->s().b_n;
  bi.n_final = n_final;
  bi.len_total_after = len_total_before + cum_final;
  if (c->p.strategy != PBSIM_STRATEGY_WGS) {
    bi.quota_reached = 0;
    bi.need_truncated_read = 0;
  } else if (c->s().b_truncated) {
    bi.quota_reached = bi.len_total_after >= quota;
    bi.need_truncated_read = !bi.quota_reached;
  } else {
    bi.quota_reached = (n_final < c->s().b_n) || (bi.len_total_after >= quota);
    bi.need_truncated_read = (n_final < c->s().b_n) && (bi.len_total_after < quota);
  }
  c->s().b_info = bi;
  *out = bi;
  return PBSIM_SUCCEEDED;
}

// The cut of a batch that cannot touch the quota -- len_total_before + its pass-0 bases + its largest raw length <= quota, so
// pbsim.cpp:3792-3800 neither stops nor truncates in it -- needs no kernel and no flag read: every read is final.  The job's
// rounds take this path before they know len_total_before (job.cpp: it arrives with the round's one exchange); the caller
// patches len_total_after once it does.
extern "C++" int pbsim::finalize_uncut(pbsim_ctx *c, pbsim_batch_info *out) {
  if (!c || !c->s().b_walked) return fail("pbsim_batch_finalize: no walked batch");
  NEED_DEVICE(c);
  pbsim_batch_info bi;
  memset(&bi, 0, sizeof bi);
  bi.first_read = c->s().b_first;
  bi.n_reads = c->s().b_n;
  bi.n_final = c->s().b_n;
  bi.len_total_after = c->s().b_pass0;  // + len_total_before (the caller's, once known)
  c->s().b_info = bi;
  *out = bi;
  return PBSIM_SUCCEEDED;
}

// Second half: text sizes, their scans, and the text itself into the slot's device buffers.
static int fetch_stats(pbsim_ctx *c, Slot &sl, int64_t n_tasks);
// adds a finished text emission's event time to the profile (no wait: an emission still running is left for later)
static void collect_text_timing(pbsim_ctx *c, Slot &sl) {
  if (!sl.text_timed || hipEventQuery(sl.ev_t1) != hipSuccess) return;
  float ms = 0;
  if (hipEventElapsedTime(&ms, sl.ev_t0, sl.ev_t1) == hipSuccess) {
    std::lock_guard<std::mutex> lk(c->prof_mu);
    c->prof_text_ms += ms;
    c->prof_text_launches++;
    c->prof_text_in += sl.text_in;
    c->prof_text_out += sl.text_out;
  }
  sl.text_timed = false;
}
extern "C++" int pbsim::finalize_text(pbsim_ctx *c, pbsim_batch_info *info) {
  DeviceFlags *flags = c->s().d_flags.as<DeviceFlags>();
  DeviceFlags f;
  pbsim_batch_info bi = c->s().b_info;
  const int P = c->p.pass_num;
  const int64_t n_final = bi.n_final;
  const int64_t n_tasks = n_final * P;
  for (DfLane &L : c->s().df) L.pre_valid = false;  // (pieces launched ahead for the slot's previous text are void)
  if (n_tasks > 0) {
    HIP_OK(c->s().d_rt_len.ensure(n_tasks * 8));
    HIP_OK(c->s().d_mt_len.ensure(n_tasks * 8));
    HIP_OK(c->s().d_row_dst.ensure(n_tasks * 6 * 8));
    // deferred mode: the counters the statistics need travel now, in front of the text emission this call no longer waits
    // for (the read_flags below waits for them): pbsim_batch_account then finds them in place
    c->s().stats_fetched = c->defer_text_sync;
    if (c->defer_text_sync && !fetch_stats(c, c->s(), n_tasks)) return PBSIM_FAILED;
    HIP_OK(hipMemsetAsync(&flags->sums[1], 0, 5 * sizeof(int64_t), c->s().stream));
    TextArgs t;
    fill_text_args(c, &t, n_final);
    launch_text_sizes(t, flags, c->s().stream);
    launch_exclusive_scan_i64(t.read_text_len, t.read_text_len, n_tasks, c->s().d_scan_tmp.as<int64_t>(), &flags->sums[1],
                              c->s().stream);
    launch_exclusive_scan_i64(t.maf_text_len, t.maf_text_len, n_tasks, c->s().d_scan_tmp.as<int64_t>(), &flags->sums[2],
                              c->s().stream);
    if (!read_flags(c, &f)) return PBSIM_FAILED;
    bi.read_text_bytes = f.sums[1];
    bi.maf_text_bytes = f.sums[2];
    bi.bases = f.sums[3];
    bi.ref_bases = f.sums[4];
    bi.maf_columns = f.sums[5];
    // (8 MiB at least: the text of any single read fits, so the slot of the truncated tail reads never reallocates)
    HIP_OK(c->s().d_read_text.ensure(std::max<size_t>((size_t)bi.read_text_bytes + 16, 8u << 20)));
    HIP_OK(c->s().d_maf_text.ensure(std::max<size_t>((size_t)bi.maf_text_bytes + 16, 8u << 20)));
    t.read_text = c->s().d_read_text.as<char>();
    t.maf_text = c->s().d_maf_text.as<char>();
    collect_text_timing(c, c->s());  // the slot's previous emission finished long ago
    HIP_OK(hipEventRecord(c->s().ev_t0, c->s().stream));
    launch_text_emit(t, c->s().b_slots_max, flags, c->s().stream);
    HIP_OK(hipEventRecord(c->s().ev_t1, c->s().stream));
    c->s().text_timed = true;
    c->s().text_in = (int64_t)regions_of(c) * bi.maf_columns;  // the MAF rows (+ the quality row) of the emitted reads
    c->s().text_out = bi.read_text_bytes + bi.maf_text_bytes;
    HIP_OK(hipGetLastError());
    HIP_OK(hipEventRecord(c->s().ev_text, c->s().stream));
    // the job pipeline's round loop does not wait for the emission (5-8 ms a round): its delivery thread does (ev_text)
    if (!c->defer_text_sync) HIP_OK(hipStreamSynchronize(c->s().stream));
  }
  c->s().b_info = bi;
  c->s().b_finalized = true;
  if (info) *info = bi;
  return PBSIM_SUCCEEDED;
}

int pbsim_batch_finalize(pbsim_ctx *c, int64_t len_total_before, pbsim_batch_info *info) {
  pbsim_batch_info bi;
  if (!finalize_cut(c, len_total_before, &bi)) return PBSIM_FAILED;
  return finalize_text(c, info);
}

int pbsim_batch_fetch(pbsim_ctx *c, char *read_text, char *maf_text) {
  if (!c || !c->s().b_finalized) return fail("pbsim_batch_fetch: no finalized batch");
  NEED_DEVICE(c);
  HIP_OK(hipSetDevice(c->device));
  if (read_text && c->s().b_info.read_text_bytes)
    HIP_OK(hipMemcpyAsync(read_text, c->s().d_read_text.p, (size_t)c->s().b_info.read_text_bytes, hipMemcpyDeviceToHost,
                          c->s().stream));
  if (maf_text && c->s().b_info.maf_text_bytes)
    HIP_OK(hipMemcpyAsync(maf_text, c->s().d_maf_text.p, (size_t)c->s().b_info.maf_text_bytes, hipMemcpyDeviceToHost,
                          c->s().stream));
  HIP_OK(hipStreamSynchronize(c->s().stream));
  return PBSIM_SUCCEEDED;
}

// pbsim.cpp:3986-4005 (errhmm) / 2293-2316 (qshmm), applied in read order so the
// order-dependent double sum `accuracy_total` matches the CPU bit for bit
// the per-task counters of the batch's final reads -> h_stats (pinned), asynchronously on the slot's stream
static int fetch_stats(pbsim_ctx *c, Slot &sl, int64_t n_tasks) {
  HIP_OK(sl.h_stats.ensure((size_t)n_tasks * 24));
  int32_t *ol = reinterpret_cast<int32_t *>(sl.h_stats.p);
  int32_t *ns = ol + n_tasks, *ni = ns + n_tasks, *nd = ni + n_tasks;
  double *qs = reinterpret_cast<double *>(nd + n_tasks);
  if (sl.sq_pending) {  // the wave-walked reads' sums (k_sample_qsum, beside the text emission)
    HIP_OK(hipStreamWaitEvent(sl.stream, sl.ev_sq_done, 0));
    sl.sq_pending = false;
  }
  HIP_OK(hipMemcpyAsync(ol, sl.d_out_len.p, n_tasks * 4, hipMemcpyDeviceToHost, sl.stream));
  HIP_OK(hipMemcpyAsync(ns, sl.d_nsub.p, n_tasks * 4, hipMemcpyDeviceToHost, sl.stream));
  HIP_OK(hipMemcpyAsync(ni, sl.d_nins.p, n_tasks * 4, hipMemcpyDeviceToHost, sl.stream));
  HIP_OK(hipMemcpyAsync(nd, sl.d_ndel.p, n_tasks * 4, hipMemcpyDeviceToHost, sl.stream));
  if (has_quality_row(c)) HIP_OK(hipMemcpyAsync(qs, sl.d_qsum.p, n_tasks * 8, hipMemcpyDeviceToHost, sl.stream));
  return PBSIM_SUCCEEDED;
}

extern "C++" int pbsim::account_of(pbsim_ctx *c, Slot &sl, StatsAcc *st) {
  if (!c || !sl.b_finalized) return fail("pbsim_batch_account: no finalized batch");
  NEED_DEVICE(c);
  HIP_OK(hipSetDevice(c->device));
  const int P = c->p.pass_num;
  const int64_t n_tasks = sl.b_info.n_final * P;
  if (n_tasks == 0) return PBSIM_SUCCEEDED;
  if (!sl.stats_fetched) {  // (deferred mode: finalize_text fetched them in front of the text emission and waited)
    if (!fetch_stats(c, sl, n_tasks)) return PBSIM_FAILED;
    HIP_OK(hipStreamSynchronize(sl.stream));
  }
  const int32_t *ol = reinterpret_cast<const int32_t *>(sl.h_stats.p);
  const int32_t *ns = ol + n_tasks, *ni = ns + n_tasks, *nd = ni + n_tasks;
  const double *qs = reinterpret_cast<const double *>(nd + n_tasks);
  const bool quality = has_quality_row(c);
  st->res_num += sl.b_info.n_final;
  std::vector<double> *values = nullptr;
  if (st->keep_values) {
    st->blocks.emplace_back();
    st->blocks.back().first_task = (sl.b_first - 1) * P;
    values = &st->blocks.back().values;
    values->reserve((size_t)n_tasks);
  }
  for (int64_t t = 0; t < n_tasks; t++) stats_add_task(st, c->p.len_max, quality, ol[t], ns[t], ni[t], nd[t], quality ? qs[t] : 0.0, values);
  return PBSIM_SUCCEEDED;
}

extern "C++" int pbsim::account_slot(pbsim_ctx *c, StatsAcc *st) {
  if (!c) return fail("pbsim_batch_account: bad argument");
  return account_of(c, c->s(), st);
}

int pbsim_batch_account(pbsim_ctx *c) { return c ? account_slot(c, &c->st) : fail("pbsim_batch_account: bad argument"); }

int pbsim_get_stats(pbsim_ctx *c, pbsim_stats *o) {  // pbsim.cpp:4082-4105, 5541-5562
  if (!c || !o) return fail("pbsim_get_stats: bad argument");
  stats_finish(c->st, c->p, c->p.strategy == PBSIM_STRATEGY_WGS ? c->ref_len : 0, o);
  return PBSIM_SUCCEEDED;
}

extern "C++" std::string pbsim::sam_header_text(const pbsim_ctx *c, int64_t unit) {  // pbsim.cpp:721-722, 784-785
  std::string h = "@HD\tVN:1.5\tSO:unknown\tpb:3.0.7\n";
  h += "@RG\tID:ffffffff\tPL:PACBIO\tDS:READTYPE=SUBREAD;Ipd:CodecV1=ip;PulseWidth:CodecV1=pw;"
       "BINDINGKIT=101-789-500;SEQUENCINGKIT=101-826-100;BASECALLERVERSION=5.0.0;FRAMERATEHZ=100.000000\tPU:";
  h += c->p.id_prefix;
  if (c->p.strategy == PBSIM_STRATEGY_WGS) h += std::to_string((long)unit);
  h += "\tPM:SEQUELII\n";
  return h;
}
int64_t pbsim_sam_header(pbsim_ctx *c, char *buf, int64_t cap) {
  if (!c) return -1;
  const std::string h = sam_header_text(c, c->unit);
  if (buf && cap > (int64_t)h.size()) memcpy(buf, h.c_str(), h.size() + 1);
  return (int64_t)h.size();
}

int pbsim_set_bam_output(pbsim_ctx *c, int on) {
  if (!c) return fail("bad argument");
  if (on && c->p.pass_num < 2) return fail("BAM output applies to --pass-num >= 2 (single pass writes FASTQ, pbsim.cpp:707)");
  c->bam_output = on != 0;
  return PBSIM_SUCCEEDED;
}

// "BAM\1" + l_text + the SAM header text + n_ref = 0 (SAMv1 section 4.2)
int64_t pbsim_bam_header(pbsim_ctx *c, char *buf, int64_t cap) {
  if (!c) return -1;
  const int64_t lt = pbsim_sam_header(c, nullptr, 0);
  const int64_t n = 4 + 4 + lt + 4;
  if (buf && cap >= n) {
    std::vector<char> text((size_t)lt + 1);
    pbsim_sam_header(c, text.data(), lt + 1);
    memcpy(buf, "BAM\1", 4);
    const uint32_t l = (uint32_t)lt, zero = 0;
    memcpy(buf + 4, &l, 4);
    memcpy(buf + 8, text.data(), (size_t)lt);
    memcpy(buf + 8 + lt, &zero, 4);
  }
  return n;
}

extern "C++" int pbsim::prepare_enqueue(pbsim_ctx *c, uint8_t *d_seq, DevBuf &hp, DevBuf &tiles, DevBuf &flags, int64_t len,
                                        hipStream_t stream) {
  return enqueue_prepare(c, d_seq, hp, tiles, flags, len, 0, stream);
}
extern "C++" int pbsim::ensure_tables(pbsim_ctx *c, bool hp11) {
  if (!ensure_header_tables(c) || !ensure_class_tables(c)) return PBSIM_FAILED;
  if (c->p.method == PBSIM_METHOD_QS && !ensure_qs_tabs(c, hp11)) return PBSIM_FAILED;
  return PBSIM_SUCCEEDED;
}


// The quota loop `while (len_total < sim.len_quota)` (pbsim.cpp:3792) as
// speculative bulk batches + prefix scan + a serial tail.  Batches
// are pipelined over the slots: batch k+1 is enqueued (assuming batch k will not
// be cut) before batch k is finalised, so the GPU never idles on a batch's
// longest read or on the host round trips; a batch enqueued past the cut is
// simply dropped.
int pbsim_simulate_wgs(pbsim_ctx *c, const pbsim_sink *sink) {
  if (!c) return fail("bad argument");
  NEED_DEVICE(c);
  if (c->p.strategy != PBSIM_STRATEGY_WGS) return fail("pbsim_simulate_wgs: strategy is not wgs");
  if (!c->d_seq) return fail("no reference set (pbsim_set_reference)");
  pbsim_reset_stats(c);
  const int64_t quota = pbsim_unit_quota(c);
  // expected pass-0 output bases per read: E[L] of the length table, a little less than that while nothing has
  // been measured (deletions outweigh insertions in most models), the measured ratio afterwards
  double mean = 0.97 * std::min<double>(c->hdr.mean_len, (double)c->ref_len);
  const int n_slots = std::max(1, std::min(kMaxSlots, c->pipeline_depth));
  if (c->scratch_auto) {
    // Nobody chose a pool size: a record wants to run as two batches (one per slot) -- a batch's walk lasts at least
    // as long as its longest read, so many small batches waste the GPU on their tails -- up to a share of the free HBM
    // (the text buffers need about as much again).
    size_t free_b = 0, total_b = 0;
    HIP_OK(hipMemGetInfo(&free_b, &total_b));
    // With a sink the text leaves the GPU batch by batch (PCIe, host writes: ~0.15 s per Gbase), which hides any tail of
    // the walks, while the first hipMalloc of a 25-GB text buffer costs 0.8 s: batches of ~2.5 Gbases there.
    double batch_bases = (double)quota * c->p.pass_num / n_slots;
    if (sink && (sink->on_read_text || sink->on_maf_text)) batch_bases = std::min(batch_bases, kSinkBatchBases);
    const double want = batch_bases * 1.07 * regions_of(c) * 1.3 + (64 << 20);
    const double share = std::min(48.0 * (1LL << 30), 0.15 * (double)(free_b + c->s().d_scratch.bytes * n_slots));
    const int64_t auto_b = (int64_t)std::max(256.0 * (1 << 20), std::min(want, share));
    if (auto_b > c->scratch_budget || c->scratch_budget > 2 * auto_b) c->scratch_budget = auto_b;
  }
  int64_t cap = batch_capacity(c);
  int64_t len_total = 0, next_read = 1;
  struct Pending {
    int slot;
    int64_t first, n;
  };
  std::vector<Pending> fifo;
  int tail_slot = -1;             // slot on which the first truncated read was started ahead of time (-1: none)
  auto drop_pending = [&]() {  // speculative batches beyond a cut or after an error
    for (const Pending &pd : fifo) {
      c->cur = pd.slot;
      (void)hipStreamSynchronize(c->s().stream);
      c->s().b_enqueued = false;
    }
    fifo.clear();
    if (tail_slot >= 0) {  // a truncated read begun ahead of time must not outlive a failed run either
      c->cur = tail_slot;
      (void)hipStreamSynchronize(c->s().stream);
      c->s().b_enqueued = false;
      tail_slot = -1;
    }
    c->cur = 0;
  };
  const bool trace = getenv("PBSIM_TRACE") != nullptr;  // per-batch host timings on stderr
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_start = now();
  int next_slot = 0;
  int64_t spec_read = 1;          // first read not yet enqueued
  double spec_total = 0;          // expected pass-0 bases once everything enqueued has finished
  bool serial = false;
  while (len_total < quota) {
    if (serial) {
      pbsim_batch_info bi;
      const double t0 = now();
      if (tail_slot >= 0) {       // begun while the last bulk batch's text was being emitted
        c->cur = tail_slot;
        tail_slot = -1;
        if (!pbsim_batch_walk_end(c, nullptr)) return PBSIM_FAILED;
      } else {
        c->cur = 0;
        if (!pbsim_batch_walk(c, next_read, 1, quota - len_total, nullptr)) return PBSIM_FAILED;
      }
      if (!pbsim_batch_finalize(c, len_total, &bi)) return PBSIM_FAILED;
      if (!deliver(c, sink)) return PBSIM_FAILED;
      if (trace) fprintf(stderr, "[pbsim trace] t=%.1f ms tail read %lld: %.1f ms\n", t0 - t_start, (long long)next_read, now() - t0);
      next_read += bi.n_final;
      len_total = bi.len_total_after;
      continue;
    }
    // keep the pipeline full with speculative batches
    while ((int)fifo.size() < n_slots) {
      const double remaining = (double)quota - spec_total;
      // split what is still expected evenly over the free slots so the batches in flight are of one size
      // Overshoot slightly (0.5 % + 64 reads: the sum of n gamma lengths has a relative spread of ~0.8/sqrt(n)):
      // a batch that ends past the quota costs its surplus reads, a batch that ends short costs a whole
      // extra round trip whose duration is set by its longest read, not by its size.
      int64_t n = (int64_t)(1.005 * remaining / mean / (double)(n_slots - (int)fifo.size())) + 64;
      if (remaining <= 0) {
        if (!fifo.empty()) break;  // enough is in flight to reach the quota
        n = 64;
      }
      n = std::min(n, cap);
      c->cur = next_slot;
      if (!pbsim_batch_walk_begin(c, spec_read, n, -1)) {
        drop_pending();
        return PBSIM_FAILED;
      }
      fifo.push_back(Pending{next_slot, spec_read, n});
      next_slot = (next_slot + 1) % n_slots;
      spec_read += n;
      spec_total += (double)n * mean;
    }
    const Pending pd = fifo.front();
    fifo.erase(fifo.begin());
    c->cur = pd.slot;
    int64_t pass0 = 0;
    const double t0 = now();
    if (!pbsim_batch_walk_end(c, &pass0)) {
      const bool budget = g_err.rfind("scratch budget exceeded", 0) == 0 && pd.n > 1;
      const std::string keep = g_err;
      drop_pending();
      if (!budget) {
        g_err = keep;
        return PBSIM_FAILED;
      }
      cap = std::max<int64_t>(1, pd.n / 2);  // skewed lengths: retry from this batch with smaller ones
      spec_read = pd.first;
      spec_total = (double)len_total;
      next_slot = 0;
      continue;
    }
    pbsim_batch_info bi;
    const double t1 = now();
    if (!finalize_cut(c, len_total, &bi)) {
      drop_pending();
      return PBSIM_FAILED;
    }
    if (fifo.empty() && n_slots > 1 && bi.n_final < pd.n && bi.need_truncated_read) {
      // The quota falls inside this batch and the next read will be a truncated one (pbsim.cpp:3795-3800).  A lone
      // read walks for up to ~20 ms: start it on the free slot now, beside this batch's text emission.
      tail_slot = (pd.slot + 1) % n_slots;
      c->cur = tail_slot;
      if (!pbsim_batch_walk_begin(c, next_read + bi.n_final, 1, quota - bi.len_total_after)) tail_slot = -1;
      c->cur = pd.slot;
    }
    if (!finalize_text(c, &bi)) {
      drop_pending();
      return PBSIM_FAILED;
    }
    const double t2 = now();
    if (!deliver(c, sink)) {
      drop_pending();
      return PBSIM_FAILED;
    }
    if (trace)
      fprintf(stderr, "[pbsim trace] t=%.1f ms batch first=%lld n=%lld final=%lld wait_walk=%.1f finalize=%.1f deliver=%.1f\n",
              t0 - t_start, (long long)pd.first, (long long)pd.n, (long long)bi.n_final, t1 - t0, t2 - t1, now() - t2);
    next_read += bi.n_final;
    len_total = bi.len_total_after;
    spec_total += (double)pass0 - (double)pd.n * mean;  // replace the estimate by what the batch produced
    if (pd.n >= 1000) {  // re-base the estimate (and what is still in flight) on the measured bases per read
      const double measured = (double)pass0 / (double)pd.n;
      double inflight = 0;
      for (const Pending &q : fifo) inflight += (double)q.n;
      spec_total += inflight * (measured - mean);
      mean = measured;
    }
    if (bi.n_final < pd.n) {  // the quota was reached inside this batch: later speculation is void
      drop_pending();
      spec_read = next_read;
      spec_total = (double)len_total;
      next_slot = 0;
      if (bi.need_truncated_read) serial = true;
    }
  }
  drop_pending();
  return PBSIM_SUCCEEDED;
}

int pbsim_prof_reset(pbsim_ctx *c) {
  if (!c) return fail("bad argument");
  c->prof_walk_ms = c->prof_total_ms = c->prof_tail_ms = 0;
  c->prof_walk_launches = c->prof_tail_launches = c->prof_wave_launches = 0;
  c->prof_intervals.clear();
  {
    std::lock_guard<std::mutex> lk(c->prof_mu);
    c->prof_text_ms = c->prof_deflate_ms = 0;
    c->prof_text_launches = c->prof_text_in = c->prof_text_out = 0;
    c->prof_deflate_launches = c->prof_deflate_in = c->prof_deflate_out = 0;
  }
  for (Slot &sl : c->slots) sl.text_timed = false;
  if (c->device >= 0 && c->stream) {
    HIP_OK(hipSetDevice(c->device));
    if (!c->ev_prof_base) HIP_OK(hipEventCreate(&c->ev_prof_base));
    HIP_OK(hipEventRecord(c->ev_prof_base, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
  }
  return PBSIM_SUCCEEDED;
}
// time during which at least one walk kernel was running (union of the launches' intervals): walks of different slots
// overlap by design, so the sum of their durations counts that time more than once
int pbsim_prof_tail(pbsim_ctx *c, double *tail_ms, int64_t *tail_launches) {
  if (!c) return fail("bad argument");
  if (tail_ms) *tail_ms = c->prof_tail_ms;
  if (tail_launches) *tail_launches = c->prof_tail_launches;
  return PBSIM_SUCCEEDED;
}
int pbsim_prof_walk_busy(pbsim_ctx *c, double *busy_ms) {
  if (!c || !busy_ms) return fail("bad argument");
  std::vector<std::pair<float, float>> v = c->prof_intervals;
  std::sort(v.begin(), v.end());
  double busy = 0, end = -1e30;
  for (const auto &iv : v) {
    if (iv.first > end) {
      busy += iv.second - iv.first;
      end = iv.second;
    } else if (iv.second > end) {
      busy += iv.second - end;
      end = iv.second;
    }
  }
  *busy_ms = busy;
  return PBSIM_SUCCEEDED;
}
int pbsim_prof_get(pbsim_ctx *c, double *walk_ms, int64_t *walk_launches, double *total_ms) {
  if (!c) return fail("bad argument");
  if (walk_ms) *walk_ms = c->prof_walk_ms;
  if (walk_launches) *walk_launches = c->prof_walk_launches;
  if (total_ms) *total_ms = c->prof_total_ms;
  return PBSIM_SUCCEEDED;
}
int64_t pbsim_prof_wave_launches(pbsim_ctx *c) { return c ? c->prof_wave_launches : -1; }
int pbsim_scratch_state(pbsim_ctx *c, double out[3]) {
  if (!c || !out) return fail("bad argument");
  out[0] = scratch_factor_of(c);
  out[1] = c->need_seen;
  out[2] = (double)c->rewalked_batches;
  return PBSIM_SUCCEEDED;
}
int pbsim_prof_secondary(pbsim_ctx *c, double out[8]) {
  if (!c || !out) return fail("bad argument");
  if (c->device >= 0) (void)hipSetDevice(c->device);
  for (Slot &sl : c->slots) collect_text_timing(c, sl);
  std::lock_guard<std::mutex> lk(c->prof_mu);
  out[0] = c->prof_text_ms;
  out[1] = (double)c->prof_text_launches;
  out[2] = (double)c->prof_text_in;
  out[3] = (double)c->prof_text_out;
  out[4] = c->prof_deflate_ms;
  out[5] = (double)c->prof_deflate_launches;
  out[6] = (double)c->prof_deflate_in;
  out[7] = (double)c->prof_deflate_out;
  return PBSIM_SUCCEEDED;
}
void *pbsim_stream(pbsim_ctx *c) { return c ? (void *)c->stream : nullptr; }
int pbsim_device_synchronize(pbsim_ctx *c) {
  if (!c) return fail("bad argument");
  NEED_DEVICE(c);
  HIP_OK(hipSetDevice(c->device));
  HIP_OK(hipDeviceSynchronize());
  return PBSIM_SUCCEEDED;
}

int64_t pbsim_dump_table(pbsim_ctx *c, int which, void *buf, int64_t cap) {
  if (!c) return -1;
  const void *src = nullptr;
  int64_t n = 0;
  std::string e;
  if (which == 0) {
    src = c->hdr.prob2len.data();
    n = (int64_t)c->hdr.prob2len.size() * 4;
  } else if (which == 1) {
    src = c->hdr.prob2acc.data();
    n = (int64_t)c->hdr.prob2acc.size();
  } else if (which == 2) {
    // host-only build of the class tables (no device needed)
    if (c->p.method == PBSIM_METHOD_ERR) {
      if (!c->err) return -1;
      if (!build_err_class_tables(*c->err, c->hdr, c->bias, c->p.strategy == PBSIM_STRATEGY_WGS, &c->ect, &e)) {
        fail(e);
        return -1;
      }
      src = c->ect.blob.data();
      n = (int64_t)c->ect.blob.size();
    } else {
      if (!c->qs) return -1;
      if (!build_qs_class_tables(*c->qs, c->hdr, c->bias, c->p, &c->qct, &e)) {
        fail(e);
        return -1;
      }
      src = c->qct.blob.data();
      n = (int64_t)c->qct.blob.size();
    }
  } else {
    return -1;
  }
  if (buf && cap >= n) memcpy(buf, src, (size_t)n);
  return n;
}

}  // extern "C"
