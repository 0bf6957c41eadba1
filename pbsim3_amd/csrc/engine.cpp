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

int upload(DevBuf &b, const void *src, size_t n, hipStream_t s) {
  HIP_OK(b.ensure(n));
  HIP_OK(hipMemcpyAsync(b.p, src, n, hipMemcpyHostToDevice, s));
  return PBSIM_SUCCEEDED;
}

int ensure_header_tables(pbsim_ctx *c) {
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
int ensure_qs_tabs(pbsim_ctx *c, bool hp11) {
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

int ensure_class_tables(pbsim_ctx *c) {
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
int read_flags(pbsim_ctx *c, DeviceFlags *f) {
  Slot &sl = c->s();
  HIP_OK(sl.h_flags.ensure(sizeof(DeviceFlags) + 64));
  HIP_OK(hipMemcpyAsync(sl.h_flags.p, sl.d_flags.p, sizeof(DeviceFlags), hipMemcpyDeviceToHost, sl.stream));
  HIP_OK(hipStreamSynchronize(sl.stream));
  memcpy(f, sl.h_flags.p, sizeof(DeviceFlags));
  return PBSIM_SUCCEEDED;
}

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
int prepare_reference(pbsim_ctx *c, uint8_t *d_seq, int64_t len, int keep_first_case, int64_t census_out[kHpSlots]) {
  if (!enqueue_prepare(c, d_seq, c->d_hp, c->d_tiles, c->d_ref_flags, len, keep_first_case, c->stream)) return PBSIM_FAILED;
  return finish_prepare(c, c->d_ref_flags, c->stream, census_out);
}

void note_hp11(pbsim_ctx *c, const int64_t census[kHpSlots]) {
  // hpfreq[11]++ in get_genome_seq (pbsim.cpp:1058) lands in hp_del_bias[0]:
  // from then on the Q15 deletion test can fire when the draw is exactly 0
  if (census[11] > 0 && !c->bias.hp11_seen) {
    c->bias.hp11_seen = true;  // batches pick the matching set_mut table variant through RefDesc::hp11 (ensure_qs_tabs)
  }
}

int64_t batch_capacity(const pbsim_ctx *c) { return batch_capacity_for(c, c->ref_len); }

}  // namespace

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
    for (DfLane &L : sl.df)
      for (DevBuf &b : L.d_df_dense) b.release();
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
    // 12.6 -> 13.3 ms).  tools/coop_dynamic_ab.sh; PBSIM_COOP_DYNAMIC=0/1 forces either.
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

extern "C++" {
namespace {

int ensure_deflate_tables(pbsim_ctx *c) {
  if (c->d_df_tables.p) return PBSIM_SUCCEEDED;
  std::vector<uint32_t> t(1024 + 256);
  deflate_host_tables(t.data(), t.data() + 1024);
  if (!upload(c->d_df_tables, t.data(), t.size() * 4, c->stream)) return PBSIM_FAILED;
  HIP_OK(hipStreamSynchronize(c->stream));
  return PBSIM_SUCCEEDED;
}

// d_text[0..n) (device; 16-byte aligned with 16 bytes of slack) -> gzip members, handed to `consume` piece by piece
// (DF_PIECE_CHUNKS chunks each) from pinned staging.  While the host consumes piece k-1 (a file write, a memcpy),
// piece k is being copied down and the GPU may already be working for another slot.
// `place` (optional): where a piece of `total` compressed bytes shall be copied to (pinned host memory of the caller's, e.g.
// an arena that keeps a whole batch) instead of the lane's double-buffered staging.
template <class F>
int deflate_stream(pbsim_ctx *c, DfLane &sl, const uint8_t *d_text, int64_t n, F &&consume,
                   const std::function<char *(int64_t)> *place = nullptr) {
  if (n <= 0) return PBSIM_SUCCEEDED;
  if (!ensure_deflate_tables(c)) return PBSIM_FAILED;
  // chunks per piece = per launch and per copy (experiment knob PBSIM_DEFLATE_PIECE_CHUNKS; a piece's members stay below 4 GiB)
  static const int64_t piece_chunks = [] {
    const char *e = exp_env("PBSIM_DEFLATE_PIECE_CHUNKS");
    const int64_t v = e ? atoll(e) : DF_PIECE_CHUNKS;
    return std::max<int64_t>(256, std::min<int64_t>(65536, v));
  }();
  const int64_t piece = piece_chunks * DF_CHUNK;
  const int64_t max_ch = std::min<int64_t>(piece_chunks, (n + DF_CHUNK - 1) / DF_CHUNK);
  const int64_t n_pieces = (n + piece - 1) / piece;
  const size_t status_bytes_was = sl.d_df_status.bytes;  // (ensure() only ever grows: a new allocation has another size)
  HIP_OK(sl.d_df_status.ensure((size_t)piece_chunks * 8));
  HIP_OK(sl.d_df_ctl.ensure(DF_CTL_BYTES));
  // (Members stored straight into page-locked host memory by the deflate workgroups -- no dense buffer, no copy -- were measured
  // in round 3 and rejected: 37 vs 47 Gbases/s, profiles/r03_deflate_fused_ab.txt; the code path is gone since round 5.)
  HIP_OK(sl.h_df_total.ensure(DF_CTL_BYTES * kDfBuffers));
  HIP_OK(sl.d_df_code.ensure(DF_TABLE_BYTES + 288 * 4));
  const uint32_t *tab = c->d_df_tables.as<uint32_t>();
  int lane_index = 0;
  for (Slot &slot : c->slots)
    if (&slot.df[1] == &sl) lane_index = 1;
  hipStream_t lane_streams[2];
  if (sl.own_streams) {  // a lane that runs BESIDE the bulk deliveries (the tail chains' worker): not behind their pieces in one stream
    // (and it never touches the context-wide streams below: their lazy creation belongs to the bulk worker's lane threads
    // alone -- lane 0 and lane 1 create different elements --, so no two threads race for one handle)
    for (int i = 0; i < 2; i++)
      if (!sl.own[i]) HIP_OK(hipStreamCreateWithFlags(&sl.own[i], hipStreamNonBlocking));
    lane_streams[0] = sl.own[0];
    lane_streams[1] = sl.own[1];
  } else {
    for (int i = 0; i < 2; i++)
      if (!c->df_streams[lane_index][i]) HIP_OK(hipStreamCreateWithFlags(&c->df_streams[lane_index][i], hipStreamNonBlocking));
    lane_streams[0] = c->df_streams[lane_index][0];
    lane_streams[1] = c->df_streams[lane_index][1];
  }
  sl.stream = lane_streams[0];
  // The look-back trusts any status word that carries the launch's epoch and a flag, and the words are never cleared between
  // launches -- so a NEW array must start from zeros (flag 0 = nothing published): hipMalloc hands back the freed array of a
  // destroyed context or lane with that lane's old words in it, and a lane's epochs restart.
  if (sl.d_df_status.bytes != status_bytes_was) HIP_OK(hipMemsetAsync(sl.d_df_status.p, 0, sl.d_df_status.bytes, sl.stream));
  // the call's code table: fitted once to the head of the text (deflate.hip), shared by all its members
  launch_deflate_table(d_text, n, reinterpret_cast<uint32_t *>(sl.d_df_code.as<uint8_t>() + DF_TABLE_BYTES), sl.d_df_code.p, sl.stream);
  unsigned long long *d_prof = nullptr;
  if (getenv("PBSIM_DEFLATE_PROF")) {
    HIP_OK(c->d_df_prof.ensure(128));
    HIP_OK(hipMemsetAsync(c->d_df_prof.p, 0, 128, sl.stream));
    d_prof = c->d_df_prof.as<unsigned long long>();
  }
  // The kernels of a piece, its copy and the host's consume() are three stages that must not wait for each other's round trips:
  // the lane's stream always holds the NEXT piece's kernels (piece k + 1 is launched before piece k's total is read back, on
  // the second set of staging buffers), the copy stream the next copy, and the host consumes piece k - 1 while piece k travels.
  // (Launching a piece only after the previous one's total had arrived left the link idle whenever the other lane was not
  // copying: 1.5 ms of kernels + a host round trip per 1.46 ms of copy.)
  sl.copy_stream = lane_streams[1];
  if (!sl.ev_df[0]) {
    for (int i = 0; i < kDfBuffers; i++) {
      HIP_OK(hipEventCreateWithFlags(&sl.ev_df[i], hipEventDisableTiming));
      HIP_OK(hipEventCreateWithFlags(&sl.ev_cp[i], hipEventDisableTiming));
      HIP_OK(hipEventCreate(&sl.ev_k0[i]));
      HIP_OK(hipEventCreate(&sl.ev_k1[i]));
    }
  }
  const bool trace = getenv("PBSIM_DEFLATE_TRACE") != nullptr;  // where a call's wall time goes: kernels | link | consumer
  const auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_kernel = 0, t_copy = 0, t_consume = 0, t_begin = now();
  int64_t out_bytes = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> tev;  // trace: begin / end of every copy on the copy stream
  bool used[kDfBuffers] = {false};
  // pieces the lane's kernels run ahead of the piece whose copy is being enqueued (ahead + 1 dense buffers in use).  Four
  // since the end of round 4 (two before): the kernels of a piece share the GPU with the next round's walk and arrive late
  // now and then; two more pieces in hand cover that -- configs[1] 1116-1120 -> 1089-1102 ms, configs[4] 3200 -> 3100-3140
  // (same box, profiles/r04_replay_late_ab.txt; six or eight with more buffers: no better) for 0.5 GB of HBM per lane.
  static const int ahead_env = exp_env("PBSIM_DEFLATE_AHEAD") ? atoi(exp_env("PBSIM_DEFLATE_AHEAD")) : 4;
  const int ahead = std::max(1, std::min(kDfBuffers - 1, ahead_env)), nbuf = ahead + 1;
  int64_t *h_total = reinterpret_cast<int64_t *>(sl.h_df_total.p);
  // piece j: kernels on staging set j & 1 into dense buffer j % 3; its total lands in h_total[j % 3]
  auto launch = [&](int64_t j) -> int {
    const int b = (int)(j % nbuf);
    const int64_t off = j * piece, len = std::min(piece, n - off);
    HIP_OK(sl.d_df_dense[b].ensure((size_t)max_ch * DF_SLOT + 64, true));
    if (!place) HIP_OK(sl.h_df_out[b].ensure((size_t)max_ch * DF_SLOT + 64));
    if (used[b]) HIP_OK(hipStreamWaitEvent(sl.stream, sl.ev_cp[b], 0));  // piece j - nbuf has left this dense buffer
    uint8_t *dense = sl.d_df_dense[b].as<uint8_t>();
    if (((sl.epoch + 1) & 0x3fffffffu) == 0) {  // the epoch wraps: start over from a cleared array; epoch 0 is never used
      HIP_OK(hipMemsetAsync(sl.d_df_status.p, 0, sl.d_df_status.bytes, sl.stream));
      sl.epoch++;
    }
    launch_deflate(d_text + off, len, sl.d_df_status.as<uint64_t>(), sl.d_df_ctl.p, ++sl.epoch, dense, tab, tab + 1024,
                   sl.d_df_code.p, sl.stream, d_prof, sl.ev_k0[b], sl.ev_k1[b]);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpyAsync(&h_total[2 * b], sl.d_df_ctl.p, DF_CTL_BYTES, hipMemcpyDeviceToHost, sl.stream));  // ticket | error, total
    HIP_OK(hipEventRecord(sl.ev_df[b], sl.stream));
    return PBSIM_SUCCEEDED;
  };
  const char *prev_ptr = nullptr;  // piece k - 1: copy possibly still in flight
  int64_t prev_bytes = 0;
  int prev_buf = 0;
  for (int64_t j = 0; j < std::min<int64_t>(ahead, n_pieces); j++)
    if (!launch(j)) return PBSIM_FAILED;
  for (int64_t k = 0; k < n_pieces; k++) {
    const int b = (int)(k % nbuf);
    const double t0 = now();
    HIP_OK(hipEventSynchronize(sl.ev_df[b]));
    t_kernel += now() - t0;
    const int64_t total = h_total[2 * b + 1];
    if ((uint64_t)h_total[2 * b] >> 32) return fail("deflate: a workgroup's look-back gave up waiting for its predecessors");
    out_bytes += total;
    {
      float ms = 0;
      if (hipEventElapsedTime(&ms, sl.ev_k0[b], sl.ev_k1[b]) == hipSuccess) {
        std::lock_guard<std::mutex> lk(c->prof_mu);
        c->prof_deflate_ms += ms;
        c->prof_deflate_launches++;
        c->prof_deflate_in += std::min(piece, n - k * piece);
        c->prof_deflate_out += total;
      }
    }
    char *dst = place ? (*place)(total) : (char *)sl.h_df_out[b].p;
    if (!dst) return fail("deflate: no room for a compressed piece");
    HIP_OK(hipStreamWaitEvent(sl.copy_stream, sl.ev_df[b], 0));
    if (trace) {
      hipEvent_t e0 = nullptr, e1 = nullptr;
      HIP_OK(hipEventCreate(&e0));
      HIP_OK(hipEventCreate(&e1));
      tev.emplace_back(e0, e1);
      HIP_OK(hipEventRecord(e0, sl.copy_stream));
    }
    HIP_OK(hipMemcpyAsync(dst, sl.d_df_dense[b].p, (size_t)total, hipMemcpyDeviceToHost, sl.copy_stream));
    if (trace) HIP_OK(hipEventRecord(tev.back().second, sl.copy_stream));
    HIP_OK(hipEventRecord(sl.ev_cp[b], sl.copy_stream));
    used[b] = true;
    if (prev_bytes) {  // (before piece k + 2 is launched: it re-uses piece k - 1's buffers)
      const double t1 = now();
      HIP_OK(hipEventSynchronize(sl.ev_cp[prev_buf]));
      const double t2 = now();
      if (!consume(prev_ptr, prev_bytes)) return PBSIM_FAILED;
      t_copy += t2 - t1;
      t_consume += now() - t2;
    }
    prev_ptr = dst;
    prev_bytes = total;
    prev_buf = b;
    // piece k's staging set is free (its total has arrived), the dense buffer of piece k - 1 once its copy is through (a
    // stream wait inside launch): keep the kernels one piece ahead
    if (k + ahead < n_pieces && !launch(k + ahead)) return PBSIM_FAILED;
  }
  if (prev_bytes) {
    const double t1 = now();
    HIP_OK(hipEventSynchronize(sl.ev_cp[prev_buf]));
    const double t2 = now();
    if (!consume(prev_ptr, prev_bytes)) return PBSIM_FAILED;
    t_copy += t2 - t1;
    t_consume += now() - t2;
  }
  if (trace) {
    double t_link = 0, t_span = 0;
    float ms = 0;
    for (auto &e : tev) {
      if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) t_link += ms;
    }
    if (!tev.empty() && hipEventElapsedTime(&ms, tev.front().first, tev.back().second) == hipSuccess) t_span = ms;
    for (auto &e : tev) {
      (void)hipEventDestroy(e.first);
      (void)hipEventDestroy(e.second);
    }
    fprintf(stderr,
            "[deflate] %.1f MB -> %.1f MB in %.1f ms: waited %.1f ms for kernels, %.1f ms for copies, %.1f ms in the consumer; copies: "
            "%.1f ms on the engine within a span of %.1f ms (%.1f GB/s while copying)\n",
            n / 1e6, out_bytes / 1e6, now() - t_begin, t_kernel, t_copy, t_consume, t_link, t_span, t_link > 0 ? out_bytes / t_link / 1e6 : 0.0);
  }
  if (d_prof) {
    unsigned long long t[16];
    HIP_OK(hipMemcpy(t, d_prof, 128, hipMemcpyDeviceToHost));
    const double nch = (double)((n + DF_CHUNK - 1) / DF_CHUNK);
    static const char *names[7] = {"stage", "crc", "sizes+scan", "header", "tokens", "trailer", "store"};
    fprintf(stderr, "[deflate prof] %.0f chunks; us per chunk (lane 0):", nch);
    for (int i = 0; i < 7; ++i) fprintf(stderr, " %s %.1f", names[i], t[i] / nch / 100);
    fprintf(stderr, "\n");
  }
  return PBSIM_SUCCEEDED;
}

int deflate_to_host(pbsim_ctx *c, DfLane &sl, const uint8_t *d_text, int64_t n, char *host_dst, int64_t cap,
                    int64_t *out_bytes) {
  int64_t written = 0;
  const int ok = deflate_stream(c, sl, d_text, n, [&](const char *z, int64_t k) {
    if (written + k > cap) return fail("deflate: output buffer too small");
    memcpy(host_dst + written, z, (size_t)k);
    written += k;
    return PBSIM_SUCCEEDED;
  });
  *out_bytes = written;
  return ok;
}

}  // namespace
}  // extern "C++"

extern "C++" int pbsim::deflate_pieces(pbsim_ctx *c, DfLane &lane, const uint8_t *d_text, int64_t n,
                                       const std::function<int(const char *, int64_t)> &consume,
                                       const std::function<char *(int64_t)> *place) {
  return deflate_stream(c, lane, d_text, n, consume, place);
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
extern "C++" int pbsim::ensure_deflate_ready(pbsim_ctx *c) { return ensure_deflate_tables(c); }

int pbsim_set_deflate(pbsim_ctx *c, int on) {
  if (!c) return fail("bad argument");
  c->deflate = on & 3;
  c->deflate_parallel = (on & 4) != 0;
  return PBSIM_SUCCEEDED;
}

int64_t pbsim_deflate_bound(int64_t n) {
  if (n <= 0) return 0;
  return n + ((n + DF_CHUNK - 1) / DF_CHUNK) * 31;
}

int pbsim_batch_fetch_deflated(pbsim_ctx *c, char *read_gz, int64_t read_cap, char *maf_gz, int64_t maf_cap,
                               int64_t *read_gz_bytes, int64_t *maf_gz_bytes) {
  if (!c || !c->s().b_finalized) return fail("pbsim_batch_fetch_deflated: no finalized batch");
  NEED_DEVICE(c);
  HIP_OK(hipSetDevice(c->device));
  int64_t nr = 0, nm = 0;
  if (read_gz && !deflate_to_host(c, c->s().df[0], c->s().d_read_text.as<uint8_t>(), c->s().b_info.read_text_bytes, read_gz,
                                  read_cap, &nr))
    return PBSIM_FAILED;
  if (maf_gz && !deflate_to_host(c, c->s().df[0], c->s().d_maf_text.as<uint8_t>(), c->s().b_info.maf_text_bytes, maf_gz,
                                 maf_cap, &nm))
    return PBSIM_FAILED;
  if (read_gz_bytes) *read_gz_bytes = nr;
  if (maf_gz_bytes) *maf_gz_bytes = nm;
  return PBSIM_SUCCEEDED;
}

// host bytes -> gzip members through the same kernels (headers, tests)
int pbsim_deflate_buffer(pbsim_ctx *c, const void *src, int64_t n, void *dst, int64_t cap, int64_t *out_bytes) {
  if (!c || !out_bytes || n < 0 || (n > 0 && (!src || !dst))) return fail("pbsim_deflate_buffer: bad argument");
  NEED_DEVICE(c);
  HIP_OK(hipSetDevice(c->device));
  *out_bytes = 0;
  if (n == 0) return PBSIM_SUCCEEDED;
  Slot &sl = c->slots[0];
  DevBuf d_in;
  HIP_OK(d_in.ensure((size_t)n + 16));
  HIP_OK(hipMemcpyAsync(d_in.p, src, (size_t)n, hipMemcpyHostToDevice, sl.stream));
  HIP_OK(hipStreamSynchronize(sl.stream));  // the lane's kernels run on its own stream
  return deflate_to_host(c, sl.df[0], d_in.as<uint8_t>(), n, (char *)dst, cap, out_bytes);
}

// A driver that does not wait for a batch's text emission in finalize_text (the batch's statistics are added on the host
// meanwhile; deliver() waits before it hands text to a sink); every emission has landed when the driver returns.
struct DeferTextSync {
  pbsim_ctx *c;
  bool prev;
  explicit DeferTextSync(pbsim_ctx *ctx) : c(ctx), prev(ctx->defer_text_sync) { c->defer_text_sync = true; }
  ~DeferTextSync() {
    for (Slot &sl : c->slots)
      if (sl.stream) (void)hipStreamSynchronize(sl.stream);
    c->defer_text_sync = prev;
  }
};

static int deliver(pbsim_ctx *c, const pbsim_sink *sink) {
  const pbsim_batch_info &bi = c->s().b_info;
  if (sink && c->defer_text_sync) {  // (drivers that do not wait for the text emission in finalize_text)
    NEED_DEVICE(c);
    HIP_OK(hipEventSynchronize(c->s().ev_text));
  }
  if (sink && c->deflate) {
    // compressed sinks stream piece by piece; a sink left as text is fetched whole as before
    const bool zr = c->deflate & 1, zm = c->deflate & 2;
    NEED_DEVICE(c);
    HIP_OK(hipSetDevice(c->device));
    if (!zr) HIP_OK(c->s().h_read_text.ensure((size_t)bi.read_text_bytes + 16));
    if (!zm) HIP_OK(c->s().h_maf_text.ensure((size_t)bi.maf_text_bytes + 16));
    if ((!zr || !zm) &&
        !pbsim_batch_fetch(c, zr ? nullptr : (char *)c->s().h_read_text.p, zm ? nullptr : (char *)c->s().h_maf_text.p))
      return PBSIM_FAILED;
    Slot &sl = c->s();
    auto send_read = [&]() -> int {
      if (!(sink->on_read_text && bi.read_text_bytes)) return PBSIM_SUCCEEDED;
      if (!zr)
        return sink->on_read_text(sink->user, (const char *)sl.h_read_text.p, bi.read_text_bytes) ? PBSIM_SUCCEEDED
                                                                                                 : fail("sink aborted (read text)");
      return deflate_stream(c, sl.df[0], sl.d_read_text.as<uint8_t>(), bi.read_text_bytes, [&](const char *z, int64_t k) {
        return sink->on_read_text(sink->user, z, k) ? PBSIM_SUCCEEDED : fail("sink aborted (read text)");
      });
    };
    auto send_maf = [&]() -> int {
      if (!(sink->on_maf_text && bi.maf_text_bytes)) return PBSIM_SUCCEEDED;
      if (!zm)
        return sink->on_maf_text(sink->user, (const char *)sl.h_maf_text.p, bi.maf_text_bytes) ? PBSIM_SUCCEEDED
                                                                                              : fail("sink aborted (MAF text)");
      return deflate_stream(c, sl.df[1], sl.d_maf_text.as<uint8_t>(), bi.maf_text_bytes, [&](const char *z, int64_t k) {
        return sink->on_maf_text(sink->user, z, k) ? PBSIM_SUCCEEDED : fail("sink aborted (MAF text)");
      });
    };
    if (c->deflate_parallel && zr && zm && bi.read_text_bytes && bi.maf_text_bytes) {
      // the two sinks are independent files: the read text goes through its lane on a second host thread while this one
      // drives the MAF lane (a file's writers would serialise on its inode, two files do not)
      if (!ensure_deflate_tables(c)) return PBSIM_FAILED;
      int ok_read = PBSIM_SUCCEEDED;
      std::string err_read;
      std::thread t([&]() {
        (void)hipSetDevice(c->device);
        ok_read = send_read();
        if (!ok_read) err_read = g_err;  // the error string is thread local
      });
      const int ok_maf = send_maf();
      t.join();
      if (!ok_read) return fail(err_read);
      if (!ok_maf) return PBSIM_FAILED;
    } else {
      if (!send_read()) return PBSIM_FAILED;
      if (!send_maf()) return PBSIM_FAILED;
    }
  } else if (sink) {
    HIP_OK(c->s().h_read_text.ensure((size_t)bi.read_text_bytes + 16));
    HIP_OK(c->s().h_maf_text.ensure((size_t)bi.maf_text_bytes + 16));
    if (!pbsim_batch_fetch(c, (char *)c->s().h_read_text.p, (char *)c->s().h_maf_text.p)) return PBSIM_FAILED;
    if (sink->on_read_text && bi.read_text_bytes &&
        !sink->on_read_text(sink->user, (const char *)c->s().h_read_text.p, bi.read_text_bytes))
      return fail("sink aborted (read text)");
    if (sink->on_maf_text && bi.maf_text_bytes &&
        !sink->on_maf_text(sink->user, (const char *)c->s().h_maf_text.p, bi.maf_text_bytes))
      return fail("sink aborted (MAF text)");
  }
  return pbsim_batch_account(c);
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
  const int n_slots = std::max(1, std::min(kMaxSlots, c->pipeline_depth));
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
  int64_t next_begin = first_read, next_read = first_read;
  int next_slot = 0;
  while (next_read <= R) {
    while ((int)fifo.size() < n_slots && next_begin <= R) {
      const int64_t part = std::max<int64_t>(65536, (n_reads + n_slots - 1) / n_slots);
      const int64_t n = std::min(std::min(cap, part), R - next_begin + 1);
      c->cur = next_slot;
      if (!pbsim_batch_walk_begin(c, next_begin, n, -1)) {
        drop_pending();
        return PBSIM_FAILED;
      }
      fifo.push_back(Pending{next_slot, next_begin, n});
      next_slot = (next_slot + 1) % n_slots;
      next_begin += n;
    }
    const Pending pd = fifo.front();
    fifo.erase(fifo.begin());
    c->cur = pd.slot;
    if (!pbsim_batch_walk_end(c, nullptr)) {
      const bool budget = g_err.rfind("scratch budget exceeded", 0) == 0 && pd.n > 1;
      const std::string keep = g_err;
      drop_pending();
      if (!budget) {
        g_err = keep;
        return PBSIM_FAILED;
      }
      cap = std::max<int64_t>(1, pd.n / 2);  // retry from this batch with smaller ones
      next_begin = pd.first;
      next_slot = 0;
      continue;
    }
    pbsim_batch_info bi;
    if (!pbsim_batch_finalize(c, 0, &bi) || !deliver(c, sink)) {
      drop_pending();
      return PBSIM_FAILED;
    }
    next_read += bi.n_final;
  }
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
