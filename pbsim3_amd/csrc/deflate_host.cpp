// deflate_host.cpp -- the host side of the output back-end: a text buffer in HBM -> gzip members in pinned host memory,
// piece by piece (deflate_stream: kernels of piece k + 4, the copy of piece k and the consumer of piece k - 1 overlap), the ABI
// entries around it (pbsim_set_deflate, pbsim_batch_fetch_deflated, pbsim_deflate_buffer) and deliver(): a finalized batch's
// text or members to a pbsim_sink.  Split out of engine.cpp in round 5; the kernels are deflate.hip's.
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

extern "C" {

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
// What a call needs before its first copy: buffers, streams, the code table and the kernels of its first pieces (df_begin).
// This part can run AHEAD of the call (deflate_prelaunch, from another thread, on another stream): an experiment of round 5 that
// was measured and not taken (job.cpp prelaunch(), profiles/r05_prelaunch_ab.txt); a product build never prelaunches.
struct DfGeom {
  int64_t piece, max_ch, n_pieces;
  int ahead, nbuf;
  // (Round 5 measured a call's first two pieces SHORT -- a quarter and a half of a piece, so that its first copy starts after a
  // quarter of a piece's kernel time: configs[1] 1 077 against 1 085 ms, ranks of eight the same, configs[4] and [2] 1-3 % slower;
  // not taken, profiles/r05_prelaunch_ab.txt.)
  int64_t off(int64_t j) const { return j * piece; }
  int64_t len(int64_t j, int64_t n) const { return std::min(piece, n - off(j)); }
};
DfGeom df_geom(int64_t n) {
  // chunks per piece = per launch and per copy (experiment knob PBSIM_DEFLATE_PIECE_CHUNKS; a piece's members stay below 4 GiB)
  static const int64_t piece_chunks = [] {
    const char *e = exp_env("PBSIM_DEFLATE_PIECE_CHUNKS");
    const int64_t v = e ? atoll(e) : DF_PIECE_CHUNKS;
    return std::max<int64_t>(256, std::min<int64_t>(65536, v));
  }();
  // pieces the lane's kernels run ahead of the piece whose copy is being enqueued (ahead + 1 dense buffers in use).  Four
  // since the end of round 4 (two before): the kernels of a piece share the GPU with the next round's walk and arrive late
  // now and then; two more pieces in hand cover that -- configs[1] 1116-1120 -> 1089-1102 ms, configs[4] 3200 -> 3100-3140
  // (same box, profiles/r04_replay_late_ab.txt; six or eight with more buffers: no better) for 0.5 GB of HBM per lane.
  static const int ahead_env = exp_env("PBSIM_DEFLATE_AHEAD") ? atoi(exp_env("PBSIM_DEFLATE_AHEAD")) : 4;
  DfGeom g;
  g.piece = piece_chunks * DF_CHUNK;
  g.max_ch = std::min<int64_t>(piece_chunks, (n + DF_CHUNK - 1) / DF_CHUNK);
  g.n_pieces = (n + g.piece - 1) / g.piece;
  g.ahead = std::max(1, std::min(kDfBuffers - 1, ahead_env));
  g.nbuf = g.ahead + 1;
  return g;
}

// piece j of the call on `sl`: its kernels into dense buffer j % nbuf, its total into h_total[j % nbuf]
int df_launch_piece(pbsim_ctx *c, DfLane &sl, const DfGeom &g, const uint8_t *d_text, int64_t n, int64_t j, bool wait_buffer,
                    bool own_staging, unsigned long long *d_prof) {
  const int b = (int)(j % g.nbuf);
  const int64_t off = g.off(j), len = g.len(j, n);
  const uint32_t *tab = c->d_df_tables.as<uint32_t>();
  int64_t *h_total = reinterpret_cast<int64_t *>(sl.h_df_total.p);
  HIP_OK(sl.d_df_dense[b].ensure((size_t)g.max_ch * DF_SLOT + 64, true));
  if (own_staging) HIP_OK(sl.h_df_out[b].ensure((size_t)g.max_ch * DF_SLOT + 64));
  if (wait_buffer) HIP_OK(hipStreamWaitEvent(sl.stream, sl.ev_cp[b], 0));  // piece j - nbuf has left this dense buffer
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
}

// buffers, streams, the call's code table and its first `ahead` pieces; `text_ready` (optional): an event the lane's stream
// waits for before it reads the text (a prelaunch runs beside the text's emission on another stream)
int df_begin(pbsim_ctx *c, DfLane &sl, const uint8_t *d_text, int64_t n, bool own_staging, hipStream_t launch_stream) {
  if (!ensure_deflate_tables(c)) return PBSIM_FAILED;
  const DfGeom g = df_geom(n);
  const size_t status_bytes_was = sl.d_df_status.bytes;  // (ensure() only ever grows: a new allocation has another size)
  HIP_OK(sl.d_df_status.ensure((size_t)(g.piece / DF_CHUNK) * 8));
  HIP_OK(sl.d_df_ctl.ensure(DF_CTL_BYTES));
  // (Members stored straight into page-locked host memory by the deflate workgroups -- no dense buffer, no copy -- were measured
  // in round 3 and rejected: 37 vs 47 Gbases/s, profiles/r03_deflate_fused_ab.txt; the code path is gone since round 5.)
  HIP_OK(sl.h_df_total.ensure(DF_CTL_BYTES * kDfBuffers));
  HIP_OK(sl.d_df_code.ensure(DF_TABLE_BYTES + 288 * 4));
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
  // A prelaunch puts the table fit and the first pieces on `launch_stream` -- the slot's own stream, right behind the text
  // emission they read -- not on the lane's: that one is shared by the slots' lanes of this index, and the kernels of the round
  // being delivered meanwhile must not queue behind a text emission that is still running.  The call proper carries on on the
  // lane's stream behind an event (ev_pre).
  sl.lane_stream = lane_streams[0];
  sl.stream = launch_stream ? launch_stream : lane_streams[0];
  sl.copy_stream = lane_streams[1];
  // The look-back trusts any status word that carries the launch's epoch and a flag, and the words are never cleared between
  // launches -- so a NEW array must start from zeros (flag 0 = nothing published): hipMalloc hands back the freed array of a
  // destroyed context or lane with that lane's old words in it, and a lane's epochs restart.
  if (sl.d_df_status.bytes != status_bytes_was) HIP_OK(hipMemsetAsync(sl.d_df_status.p, 0, sl.d_df_status.bytes, sl.stream));
  // the call's code table: fitted once to the head of the text (deflate.hip), shared by all its members
  launch_deflate_table(d_text, n, reinterpret_cast<uint32_t *>(sl.d_df_code.as<uint8_t>() + DF_TABLE_BYTES), sl.d_df_code.p, sl.stream);
  if (!sl.ev_df[0]) {
    for (int i = 0; i < kDfBuffers; i++) {
      HIP_OK(hipEventCreateWithFlags(&sl.ev_df[i], hipEventDisableTiming));
      HIP_OK(hipEventCreateWithFlags(&sl.ev_cp[i], hipEventDisableTiming));
      HIP_OK(hipEventCreate(&sl.ev_k0[i]));
      HIP_OK(hipEventCreate(&sl.ev_k1[i]));
    }
  }
  unsigned long long *d_prof = nullptr;
  if (getenv("PBSIM_DEFLATE_PROF")) {
    HIP_OK(c->d_df_prof.ensure(128));
    HIP_OK(hipMemsetAsync(c->d_df_prof.p, 0, 128, sl.stream));
    d_prof = c->d_df_prof.as<unsigned long long>();
  }
  // (the previous call's copies have all been waited for by the host: every dense buffer is free, no wait_buffer)
  // A prelaunch takes the table fit and ONE piece: enough for the call's first copy to start at once; the rest of the head
  // start follows when the call begins (more pieces launched ahead took the GPU from the round being delivered:
  // profiles/r05_prelaunch_ab.txt).
  static const int pre_pieces = exp_env("PBSIM_DEFLATE_PRE_PIECES") ? atoi(exp_env("PBSIM_DEFLATE_PRE_PIECES")) : 1;
  const int64_t first = std::min<int64_t>(launch_stream ? std::max(1, std::min(pre_pieces, g.ahead)) : g.ahead, g.n_pieces);
  for (int64_t j = 0; j < first; j++)
    if (!df_launch_piece(c, sl, g, d_text, n, j, false, own_staging, d_prof)) return PBSIM_FAILED;
  sl.pre_count = (int)first;
  sl.pre_text = d_text;
  sl.pre_n = n;
  sl.pre_own_staging = own_staging;
  sl.pre_valid = true;
  sl.pre_elsewhere = launch_stream != nullptr;
  if (launch_stream) {
    if (!sl.ev_pre) HIP_OK(hipEventCreateWithFlags(&sl.ev_pre, hipEventDisableTiming));
    HIP_OK(hipEventRecord(sl.ev_pre, launch_stream));
  }
  return PBSIM_SUCCEEDED;
}

template <class F>
int deflate_stream(pbsim_ctx *c, DfLane &sl, const uint8_t *d_text, int64_t n, F &&consume,
                   const std::function<char *(int64_t)> *place = nullptr) {
  if (n <= 0) return PBSIM_SUCCEEDED;
  // The kernels of a piece, its copy and the host's consume() are three stages that must not wait for each other's round trips:
  // the lane's stream always holds the NEXT pieces' kernels (piece k + ahead is launched before piece k's total is read back),
  // the copy stream the next copy, and the host consumes piece k - 1 while piece k travels.
  // (Launching a piece only after the previous one's total had arrived left the link idle whenever the other lane was not
  // copying: 1.5 ms of kernels + a host round trip per 1.46 ms of copy.)
  const bool prelaunched = sl.pre_valid && sl.pre_text == d_text && sl.pre_n == n && sl.pre_own_staging == (place == nullptr);
  sl.pre_valid = false;
  if (!prelaunched && !df_begin(c, sl, d_text, n, place == nullptr, nullptr)) return PBSIM_FAILED;
  sl.pre_valid = false;
  if (sl.pre_elsewhere) {  // the first pieces were launched on the slot's stream: the rest follows them on the lane's
    sl.stream = sl.lane_stream;
    HIP_OK(hipStreamWaitEvent(sl.stream, sl.ev_pre, 0));
    sl.pre_elsewhere = false;
  }
  const DfGeom g = df_geom(n);
  const int64_t n_pieces = g.n_pieces;
  const int ahead = g.ahead, nbuf = g.nbuf;
  unsigned long long *d_prof = getenv("PBSIM_DEFLATE_PROF") ? c->d_df_prof.as<unsigned long long>() : nullptr;
  const bool trace = getenv("PBSIM_DEFLATE_TRACE") != nullptr;  // where a call's wall time goes: kernels | link | consumer
  const auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_kernel = 0, t_copy = 0, t_consume = 0, t_begin = now();
  int64_t out_bytes = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> tev;  // trace: begin / end of every copy on the copy stream
  bool used[kDfBuffers] = {false};
  int64_t *h_total = reinterpret_cast<int64_t *>(sl.h_df_total.p);
  auto launch = [&](int64_t j) -> int {
    return df_launch_piece(c, sl, g, d_text, n, j, used[(int)(j % nbuf)], place == nullptr, d_prof);
  };
  for (int64_t j = sl.pre_count; j < std::min<int64_t>(ahead, n_pieces); j++)  // (what a prelaunch left of the head start)
    if (!launch(j)) return PBSIM_FAILED;
  const char *prev_ptr = nullptr;  // piece k - 1: copy possibly still in flight
  int64_t prev_bytes = 0;
  int prev_buf = 0;
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
        c->prof_deflate_in += g.len(k, n);
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
extern "C++" int pbsim::ensure_deflate_ready(pbsim_ctx *c) {
  // (called by a job before its delivery threads start: the lanes' shared streams exist from here on -- the round loop's
  // prelaunch and the delivery thread's calls would otherwise both find them missing and both create them)
  for (auto &lane : c->df_streams)
    for (hipStream_t &st : lane)
      if (!st) HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  return ensure_deflate_tables(c);
}
extern "C++" int pbsim::deflate_prelaunch(pbsim_ctx *c, Slot &sl, bool want_read, bool want_maf, bool staged) {
  const pbsim_batch_info &bi = sl.b_info;
  if (want_read && bi.read_text_bytes > 0 && !df_begin(c, sl.df[0], sl.d_read_text.as<uint8_t>(), bi.read_text_bytes, staged, sl.stream))
    return PBSIM_FAILED;
  if (want_maf && bi.maf_text_bytes > 0 && !df_begin(c, sl.df[1], sl.d_maf_text.as<uint8_t>(), bi.maf_text_bytes, staged, sl.stream))
    return PBSIM_FAILED;
  return PBSIM_SUCCEEDED;
}

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

extern "C++" int pbsim::deliver(pbsim_ctx *c, const pbsim_sink *sink) {
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
  return c->defer_account ? PBSIM_SUCCEEDED : pbsim_batch_account(c);
}
}  // extern "C"
