// job.cpp -- the whole wgs job as ONE pipeline of read batches across all records and all ranks (include/pbsim3_amd.h,
// "the whole job on one or several GPUs").
//
// The reference's main() simulates its records one after the other and each record's quota loop read by read
// (pbsim.cpp:667-759, 3792-4080).  Here every record is resident in HBM and the loop runs as rounds:
//
//   round   = W blocks of n reads of one record (W = ranks), rank r walks block r; speculative, un-truncated lengths
//   pop     = rounds finish in the order they were begun.  What the ranks exchange per round (round 5: ONE all-gather for a
//             round that stays clear of the quota, two for the round that places the cut; rounds 2-4: three for every round):
//               A  pass-0 bases + largest raw length of every block -> every rank's len_total in front of its block (the quota
//                  prefix) and whether any read of the round can touch the quota at all (pbsim.cpp:3792-3800: a read stops or
//                  is truncated only if len_total + its raw length > quota)
//               B  (n_final, need_truncated, len_total_after, status) of every block -> the cut; skipped -- every value follows
//                  from A -- when len_total + sum of the blocks' pass-0 bases + the largest raw length <= quota
//               C  (compressed bytes of both streams, status) of the PREVIOUS round -> every rank's byte range; rides in the
//                  round's A message (rounds expected to stay clear of the quota: "AC") or in its B message ("BC")
//   cut     = the first block the quota rule stops in (pbsim.cpp:3792-3800); later blocks / rounds of the record are void
//   tail    = the truncated reads behind the cut, one at a time (each depends on the one before), on the cut's rank only,
//             on a slot of their own, polled between rounds -- no other rank waits for them
//   merge   = per record, at a fixed point of the round sequence (before record n+3 begins, or at the end): statistics of
//             all ranks summed (C2), accuracy_total folded in read order; then on_record_done everywhere
//
// Record n+1's rounds begin as soon as record n has enough reads in flight, so the last text emission and the tail of
// a record run beside the next record's walks.  All decisions derive from gathered values, identical on every rank: the
// ranks stay in lockstep without any control message.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <future>
#include <mutex>
#include <thread>

#include "ctx.h"

using namespace pbsim;

namespace {

double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// kSinkBatchBases (ctx.h): expected bases per rank and batch when the text leaves the GPU (PCIe hides the walks' tails)
constexpr double kMinBatchBases = 2.0e9;    // below this a batch's walk is shorter than its longest read
constexpr int kRoundsPerRecord = 4;

struct Rec {
  RefDesc ref;
  int64_t quota = 0, len_total = 0, next_read = 1, spec_read = 1;
  double spec_total = 0;
  int64_t cap = 0;           // reads per rank and round at most
  bool bulk_done = false;    // the cut is placed (or the quota reached without one)
  bool done = false;         // bulk_done and no tail pending on this rank
  int owner = -1;            // rank that runs the tail (-1: none)
  int tail_slot = -1;        // owner: slot of the truncated read in flight
  bool tail_waiting = false; // owner: a truncated read is due but no slot was free
  int64_t read_off = 0, maf_off = 0;            // bytes of the bulk rounds, all ranks (identical everywhere)
  int64_t tail_read = 0, tail_maf = 0;          // owner: bytes of its tail reads
  StatsAcc st;
  bool merged = false;
};

struct Round {
  int rec, slot;
  int64_t first, n_per;
  double mean;               // bases per read assumed when it was begun
  bool clear;                // expected to end clear of the quota (decided when it was begun, from gathered state: the same on
                             // every rank): its text is emitted before the exchange, which then carries A and C at once
};

// One host thread that takes finished batches off the main loop's hands: GPU compression + copy into pinned memory, the sink
// callbacks (file writes in the CLI) and the per-task statistics.  Strictly FIFO, so the pieces of a record reach a
// sequential sink in order and accuracy_total is accumulated in read order.
struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv, cv_idle;
  std::deque<std::function<void()>> q;
  bool stop = false, running = false;
  int active = 0;
  // Second halves of tasks that must wait for something only ANOTHER thread brings about (a chain's hand-over to the sink waits
  // for its record's bulk rounds): parked here by the task itself (defer(), on this thread) instead of blocking the FIFO -- a
  // task queued behind it may be what the main loop is waiting for.  Entries of one key keep their order; the thread looks at
  // them after every task and every millisecond while any is parked.
  struct Deferred {
    int key;
    std::function<bool()> ready;
    std::function<void()> run;
  };
  std::deque<Deferred> parked;   // this thread only
  int n_parked = 0;              // under mu
  void defer(int key, std::function<bool()> ready, std::function<void()> run) {
    parked.push_back(Deferred{key, std::move(ready), std::move(run)});
    std::lock_guard<std::mutex> lk(mu);  // (drain()'s predicate reads it under the lock: no lost wake-up)
    n_parked++;
  }
  bool run_parked() {  // on the worker thread, no lock held
    bool any = false;
    for (size_t i = 0; i < parked.size();) {
      bool first_of_key = true;
      for (size_t j = 0; j < i; j++) first_of_key &= parked[j].key != parked[i].key;
      if (first_of_key && parked[i].ready()) {
        Deferred d = std::move(parked[i]);
        parked.erase(parked.begin() + (long)i);
        d.run();
        {
          std::lock_guard<std::mutex> lk(mu);
          n_parked--;
        }
        any = true;
        i = 0;
      } else {
        i++;
      }
    }
    return any;
  }
  void start(int device) {
    running = true;
    th = std::thread([this, device]() {
      (void)hipSetDevice(device);
      for (;;) {
        std::function<void()> f;
        {
          std::unique_lock<std::mutex> lk(mu);
          if (parked.empty()) cv.wait(lk, [&] { return stop || !q.empty(); });
          else cv.wait_for(lk, std::chrono::milliseconds(1), [&] { return stop || !q.empty(); });
          if (q.empty() && parked.empty()) return;
          if (!q.empty()) {
            f = std::move(q.front());
            q.pop_front();
            active++;
          }
        }
        if (f) {
          f();
          std::lock_guard<std::mutex> lk(mu);
          active--;
        }
        if (run_parked() || f) cv_idle.notify_all();
      }
    });
  }
  void post(std::function<void()> f) {
    {
      std::lock_guard<std::mutex> lk(mu);
      q.push_back(std::move(f));
    }
    cv.notify_one();
  }
  void drain() {
    if (!running) return;
    std::unique_lock<std::mutex> lk(mu);
    cv_idle.wait(lk, [&] { return q.empty() && active == 0 && n_parked == 0; });
  }
  void finish() {
    if (!running) return;
    drain();
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv.notify_all();
    th.join();
    running = false;
  }
  ~Worker() { finish(); }
};

// a bulk round between "text emitted" and "delivered": stage 1 (worker) brings the bytes to host memory and learns their
// sizes, the main loop exchanges the sizes (collective), stage 2 (worker) hands them to the sink and accounts the tasks
struct Delivery {
  int slot = -1, rec = -1;
  bool mine = false;
  int64_t n_per = 0;
  std::shared_ptr<std::promise<int>> stage1;
  std::future<int> stage1_done;
  int64_t sizes[2] = {0, 0};
  int ok1 = PBSIM_SUCCEEDED;          // stage 1's outcome, once waited for (wait_pending)
  double t_wait0 = 0, t_wait1 = 0;
};

struct Job {
  pbsim_ctx *c;
  const pbsim_comm *comm;
  const pbsim_record_sink *sink;
  int rank = 0, W = 1;
  bool multi = false;  // the several-rank protocol is on: W > 1, or a communicator of one under PBSIM_COMM_ALWAYS (pbsim_job_run)
  std::vector<Rec> recs;
  std::deque<Round> fifo;
  double mean = 0;
  int depth = 3;
  int interleave = 1;
  int64_t reads_walked = 0, reads_delivered = 0, rounds = 0, bases = 0, ref_bases = 0, maf_columns = 0;
  double comm_us = 0;
  // where the round loop's wall time goes (pbsim_job_breakdown), microseconds
  double bd_wait_walk = 0, bd_finalize = 0, bd_wait_bytes = 0, bd_account = 0, bd_tail_block = 0, bd_drain = 0, bd_slot_wait = 0,
         bd_merge = 0, bd_begin = 0, bd_tail_steps = 0;
  double bd_worker_busy = 0;  // written by the worker thread only, read after it has finished
  int64_t n_topup = 0, n_tail_reads = 0, n_clear_missed = 0, n_exchanges = 0;
  bool trace = false;
  double t_start = 0;
  bool peer_failed = false;   // the failure came from another rank's status word (no need to abort the communicator)
  Worker worker;
  // The chains of truncated tail reads deliver their few KB through a worker of their own: behind the bulk worker's FIFO a
  // record's tail bytes -- and with them the record's merge, which the main loop waits for -- queued behind the 30-40 ms
  // deliveries of the NEXT record's rounds (measured per rank of eight: 3 x 32 ms of a 195 ms job, profiles/r04_replay_*).
  Worker tail_worker;
  std::mutex sink_mu[2];  // a callback is never called concurrently with itself (include/pbsim3_amd.h): the two workers share the sink
  std::mutex out_mu;
  std::condition_variable out_cv;
  std::atomic<bool> delivering[kMaxSlots];
  std::atomic<bool> wfailed{false};
  std::mutex werr_mu;
  std::string werr;
  std::unique_ptr<Delivery> pending;  // the one round whose sizes have not been exchanged yet
  // deliveries of a record that are with the worker (posted, not finished): a record's merge waits for ITS bytes only -- the
  // worker is FIFO, so that is as soon as it has passed them -- not for the rounds of later records behind them
  std::unique_ptr<std::atomic<int>[]> rec_out;
  // ... and the BULK deliveries among them: a record's tail bytes follow its bulk bytes in both streams, but they travel on
  // another thread (tail_worker) -- the tail waits here until the bulk worker has handed the record's last round to the sink,
  // so a sink that takes a stream front to back (one rank: "the offsets simply run up", include/pbsim3_amd.h; the CLI's
  // --gzip host and samtools consumers check it) never sees the tail first.  Every bulk delivery of the record has
  // been POSTED by the time its tail is (tail_poll: not while the record's last sizes are pending), so zero means "all through".
  std::unique_ptr<std::atomic<int>[]> rec_bulk;
  // ... and the record's bulk rounds whose sizes have not been exchanged or whose statistics have not been accounted yet (main
  // loop: +1 when a round's delivery is created, -1 behind its accounting).  A chain of tail reads is collected, its text
  // emitted and compressed as soon as its walks have finished -- also while the record's last round is still pending --, and
  // only the hand-over to the sink and the statistics wait: for this counter (the record's byte offsets are final, its bulk
  // statistics in read order) and for rec_bulk.  (Round 4 collected the chain only behind the exchange of the record's last
  // sizes: on the rank that owns the tails the record's merge then waited 8-10 ms for the chain's text and compression --
  // profiles/r05_replay_*.)
  std::unique_ptr<std::atomic<int>[]> rec_rounds_open;
  void round_closed(int rec) {
    --rec_rounds_open[(size_t)rec];
    { std::lock_guard<std::mutex> lk(out_mu); }
    out_cv.notify_all();
  }
  std::atomic<bool> giving_up{false};  // the job failed on the main loop: nobody will close the rounds a tail waits for
  void wait_record(int rec) {
    std::unique_lock<std::mutex> lk(out_mu);
    out_cv.wait(lk, [&] { return rec_out[(size_t)rec].load() == 0; });
  }
  void delivered_one(std::atomic<int> *out, std::atomic<int> *bulk = nullptr) {  // on a worker thread, at the end of a delivery
    --*out;
    if (bulk) --*bulk;
    { std::lock_guard<std::mutex> lk(out_mu); }
    out_cv.notify_all();
  }

  Job() {
    for (auto &d : delivering) d = false;
  }

  // ---- the records as they become resident (pbsim_job_expect: possibly while the job runs, added by another thread) --------
  size_t n_collected = 0;
  bool seen11 = false;      // Q15 state in front of the next record to collect (cumulative like the reference's record loop)
  double bd_record_wait = 0;
  // record i has been added and its preparation has finished: what the preparation found (or null: the feed failed / timed out)
  const DeviceFlags *record_arrived(int i) {
    JobRecord *jr = nullptr;
    {
      std::unique_lock<std::mutex> lk(c->job_mu);
      const double t0 = now_us();
      const bool came = c->job_cv.wait_for(lk, std::chrono::seconds(600),
                                           [&] { return c->job_records.size() > (size_t)i || c->job_feed_failed; });
      bd_record_wait += now_us() - t0;
      if (c->job_records.size() <= (size_t)i) {
        fail(!came ? "pbsim_job_run: a record announced by pbsim_job_expect did not arrive within 600 s" : c->job_feed_err);
        return nullptr;
      }
      jr = c->job_records[(size_t)i].get();
    }
    const double t1 = now_us();
    if (hipEventSynchronize(jr->ready) != hipSuccess) {
      fail("hipEventSynchronize failed (a record's preparation)");
      return nullptr;
    }
    bd_record_wait += now_us() - t1;
    return jr->h_flags;
  }
  // records 0 .. upto are resident: their descriptors as the walks need them (in order: Q15's state is cumulative)
  int collect(int upto) {
    while ((int)n_collected <= upto) {
      const size_t i = n_collected;
      const DeviceFlags *f = record_arrived((int)i);
      if (!f) return PBSIM_FAILED;
      JobRecord *jr;
      {
        std::lock_guard<std::mutex> lk(c->job_mu);
        jr = c->job_records[i].get();
      }
      if (trace) fprintf(stderr, "[pbsim job r%d] t=%.1f ms record %zu collected (waited %.1f ms for records so far)\n", rank, (now_us() - t_start) / 1e3, i + 1, bd_record_wait / 1e3);
      seen11 |= f->hpfreq[11] > 0;
      jr->ref.seq = jr->seq.as<uint8_t>();
      jr->ref.hp = jr->hp.as<uint8_t>();
      jr->ref.len = jr->len;
      jr->ref.unit = c->job_first_unit + (int64_t)i;
      jr->ref.hp_flag = c->p.hp_del_bias == 1 && !f->high_bytes;
      jr->ref.hp11 = seen11;
      recs[i].ref = jr->ref;
      c->bias.hp11_seen = seen11;
      if (!ensure_tables(c, jr->ref.hp11)) return PBSIM_FAILED;
      n_collected++;
    }
    return PBSIM_SUCCEEDED;
  }

  void worker_fail() {  // on the worker thread: keep the (thread local) message for the main loop
    std::lock_guard<std::mutex> lk(werr_mu);
    if (!wfailed) werr = g_err;
    wfailed = true;
  }
  int check_worker() {
    if (!wfailed) return PBSIM_SUCCEEDED;
    std::lock_guard<std::mutex> lk(werr_mu);
    return fail(werr);
  }

  // pbsim_job_progress: which exchange of the round sequence this rank is about to enter (a communicator that replays or
  // models the other ranks -- bench.py --replay-ranks -- reads it from inside its callbacks)
  void progress(int phase, int rec, int64_t first = 0, int64_t n_per = 0) {
    int64_t *g = c->job_progress;
    g[0] = phase;
    g[1] = rec;
    g[2] = first;
    g[3] = n_per;
    g[4] = W;
    g[5] = rec >= 0 ? recs[(size_t)rec].len_total : 0;
    g[6] = rec >= 0 ? recs[(size_t)rec].quota : 0;
    g[7] = rec >= 0 ? recs[(size_t)rec].next_read : 0;
  }
  int gather(const int64_t *send, int64_t n, std::vector<int64_t> *recv) {
    recv->assign((size_t)W * n, 0);
    if (!multi) {
      memcpy(recv->data(), send, (size_t)n * 8);
      return PBSIM_SUCCEEDED;
    }
    const double t0 = now_us();
    const int ok = comm->all_gather_i64(comm->user, send, n, recv->data());
    comm_us += now_us() - t0;
    n_exchanges++;
    return ok ? PBSIM_SUCCEEDED : fail("pbsim_comm.all_gather_i64 failed");
  }

  bool slot_busy(int s) const {
    if (delivering[s]) return true;
    for (const Round &r : fifo)
      if (r.slot == s) return true;
    for (const Rec &r : recs)
      if (r.tail_slot == s) return true;
    return false;
  }
  // The last two slots belong to the chains of truncated tail reads (a handful of reads: their pools stay small), the others
  // to the rounds.  max(2, interleave) records are open at a time (run(): a record is merged before the record that many
  // places behind it begins; the CLI interleaves four): a third chain waits its turn for a slot (tail_poll, tail_waiting).
  static constexpr int kTailSlot = kMaxSlots - 2;
  int free_slot() const {
    for (int s = 0; s < kTailSlot; s++)
      if (!slot_busy(s)) return s;
    return -1;
  }
  int free_tail_slot() const {
    for (int s = kTailSlot; s < kMaxSlots; s++)
      if (!slot_busy(s)) return s;
    return -1;
  }
  int bulk_in_flight() const { return (int)fifo.size(); }
  // A slot for a bulk round.  WHETHER a round is begun must not depend on how far this rank's worker has got (the ranks would
  // fall out of step), so the main loop decides from the gathered state alone and then waits here for a slot.  One always
  // comes free: at most `depth` rounds and one pending delivery hold slots that only the main loop can release; every other
  // slot is with the worker, which needs nobody.
  int acquire_slot() {
    const double t0 = now_us();
    for (;;) {
      const int s = free_slot();
      if (s >= 0) {
        bd_slot_wait += now_us() - t0;
        return s;
      }
      std::unique_lock<std::mutex> lk(worker.mu);
      worker.cv_idle.wait_for(lk, std::chrono::milliseconds(2));
    }
  }

  void drop_round(const Round &r) {
    c->cur = r.slot;
    (void)hipStreamSynchronize(c->s().stream);
    c->s().b_enqueued = false;
  }
  void drop_rounds_of(int rec) {
    for (auto it = fifo.begin(); it != fifo.end();) {
      if (it->rec == rec) {
        drop_round(*it);
        it = fifo.erase(it);
      } else {
        ++it;
      }
    }
  }
  void drop_everything(bool failing = true) {
    if (failing) {  // nobody will close the rounds (or deliver the bytes) a chain's hand-over is waiting for
      giving_up = true;
      { std::lock_guard<std::mutex> lk(out_mu); }
      out_cv.notify_all();
    }
    worker.drain();
    tail_worker.drain();
    for (const Round &r : fifo) drop_round(r);
    fifo.clear();
    for (Rec &r : recs)
      if (r.tail_slot >= 0) {
        c->cur = r.tail_slot;
        (void)hipStreamSynchronize(c->s().stream);
        c->s().b_enqueued = false;
        r.tail_slot = -1;
      }
  }

  size_t arena_max = 0;  // bytes of pinned memory one lane's arena may hold
  bool wants_text() const { return sink && (sink->on_read_text || sink->on_maf_text); }
  bool deflated() const { return wants_text() && c->deflate == 3; }

  // ---- delivery primitives on an explicit slot (they run on the worker thread; nothing here touches c->cur) ---------------
  int fetch_plain(Slot &sl) {
    const pbsim_batch_info &bi = sl.b_info;
    const bool want_r = sink->on_read_text && bi.read_text_bytes, want_m = sink->on_maf_text && bi.maf_text_bytes;
    if (want_r) {
      HIP_OK(sl.h_read_text.ensure((size_t)bi.read_text_bytes + 16));
      HIP_OK(hipMemcpyAsync(sl.h_read_text.p, sl.d_read_text.p, (size_t)bi.read_text_bytes, hipMemcpyDeviceToHost, sl.stream));
    }
    if (want_m) {
      HIP_OK(sl.h_maf_text.ensure((size_t)bi.maf_text_bytes + 16));
      HIP_OK(hipMemcpyAsync(sl.h_maf_text.p, sl.d_maf_text.p, (size_t)bi.maf_text_bytes, hipMemcpyDeviceToHost, sl.stream));
    }
    HIP_OK(hipStreamSynchronize(sl.stream));
    return PBSIM_SUCCEEDED;
  }
  int call_sink(int which, int64_t unit, const char *text, int64_t bytes, int64_t at) {
    std::lock_guard<std::mutex> lk(sink_mu[which]);
    return (which == 0 ? sink->on_read_text : sink->on_maf_text)(sink->user, unit, text, bytes, at);
  }
  int sink_plain(Slot &sl, int64_t unit, int64_t read_at, int64_t maf_at) {
    const pbsim_batch_info &bi = sl.b_info;
    if (sink->on_read_text && bi.read_text_bytes && !call_sink(0, unit, (const char *)sl.h_read_text.p, bi.read_text_bytes, read_at))
      return fail("sink aborted (read text)");
    if (sink->on_maf_text && bi.maf_text_bytes && !call_sink(1, unit, (const char *)sl.h_maf_text.p, bi.maf_text_bytes, maf_at))
      return fail("sink aborted (MAF text)");
    return PBSIM_SUCCEEDED;
  }
  // the two text streams of a slot through `lane(which)`, side by side on two threads when the sinks are independent
  // (pbsim_set_deflate bit 2: two files are written by two threads; a file's writers would serialise on its inode)
  template <class F>
  int both_lanes(Slot &sl, F &&lane) {
    const pbsim_batch_info &bi = sl.b_info;
    if (c->deflate_parallel && bi.read_text_bytes && bi.maf_text_bytes && sink->on_read_text && sink->on_maf_text) {
      int ok_read = PBSIM_SUCCEEDED;
      std::string err_read;
      std::thread t([&]() {
        (void)hipSetDevice(c->device);
        ok_read = lane(0);
        if (!ok_read) err_read = g_err;  // the error string is thread local: carry the second thread's over
      });
      const int ok_maf = lane(1);
      t.join();
      if (!ok_read) return fail(err_read);
      return ok_maf;
    }
    return lane(0) && lane(1);
  }
  // compressed and streamed piece by piece at running offsets (sizes out): one rank, or a tail read
  int stream_deflated(Slot &sl, int64_t unit, int64_t read_at, int64_t maf_at, int64_t *read_gz, int64_t *maf_gz) {
    const pbsim_batch_info &bi = sl.b_info;
    *read_gz = *maf_gz = 0;
    return both_lanes(sl, [&](int which) -> int {
      const bool is_read = which == 0;
      const int64_t n = is_read ? bi.read_text_bytes : bi.maf_text_bytes;
      auto cb = is_read ? sink->on_read_text : sink->on_maf_text;
      int64_t *sent = is_read ? read_gz : maf_gz;
      const int64_t base = is_read ? read_at : maf_at;
      const uint8_t *d = is_read ? sl.d_read_text.as<uint8_t>() : sl.d_maf_text.as<uint8_t>();
      if (!cb || n == 0) return PBSIM_SUCCEEDED;
      return deflate_pieces(c, sl.df[which], d, n, [&](const char *z, int64_t k) {
        if (!call_sink(which, unit, z, k, base + *sent)) return fail(is_read ? "sink aborted (read text)" : "sink aborted (MAF text)");
        *sent += k;
        return PBSIM_SUCCEEDED;
      });
    });
  }
  // several ranks: a rank learns where its bytes go only after every rank has compressed its block, so the whole batch is
  // compressed into the lanes' pinned arenas first (sizes out) and flushed once the offsets are known
  int arena_fill(Slot &sl, int64_t *read_gz, int64_t *maf_gz) {
    const pbsim_batch_info &bi = sl.b_info;
    *read_gz = *maf_gz = 0;
    return both_lanes(sl, [&](int which) -> int {
      DfLane &L = sl.df[which];
      L.arena_reset();
      const int64_t n = which == 0 ? bi.read_text_bytes : bi.maf_text_bytes;
      auto cb = which == 0 ? sink->on_read_text : sink->on_maf_text;
      if (!cb || n == 0) return PBSIM_SUCCEEDED;
      const uint8_t *d = which == 0 ? sl.d_read_text.as<uint8_t>() : sl.d_maf_text.as<uint8_t>();
      bool oom = false;
      const std::function<char *(int64_t)> place = [&](int64_t k) -> char * {
        char *p = L.arena_reserve(k, arena_max);
        if (!p) oom = true;
        return p;
      };
      if (!deflate_pieces(c, L, d, n, [](const char *, int64_t) { return PBSIM_SUCCEEDED; }, &place))
        return oom ? fail("out of pinned host memory for a compressed batch (PBSIM_PINNED_ARENA_MB bounds a lane's arena)") : PBSIM_FAILED;
      int64_t tot = 0;
      for (const auto &sg : L.arena_segs) tot += sg.second;
      (which == 0 ? *read_gz : *maf_gz) = tot;
      return PBSIM_SUCCEEDED;
    });
  }
  int arena_flush(Slot &sl, int64_t unit, int64_t read_at, int64_t maf_at) {
    return both_lanes(sl, [&](int which) -> int {
      int64_t at = which == 0 ? read_at : maf_at;
      for (const auto &sg : sl.df[which].arena_segs) {
        if (!call_sink(which, unit, sg.first, sg.second, at)) return fail(which == 0 ? "sink aborted (read text)" : "sink aborted (MAF text)");
        at += sg.second;
      }
      sl.df[which].arena_segs.clear();
      return PBSIM_SUCCEEDED;
    });
  }

  // ---- a bulk round's delivery: stage 1 now, the rest once the sizes of all ranks are known -------------------------------
  // mode of the bytes: none (text stays in HBM) | plain text | members streamed (one rank) | members via the arena (ranks)
  void submit_stage1(Delivery *d) {
    d->stage1 = std::make_shared<std::promise<int>>();
    d->stage1_done = d->stage1->get_future();
    if (!d->mine) {
      d->stage1->set_value(PBSIM_SUCCEEDED);
      return;
    }
    delivering[d->slot] = true;
    Slot *sl = &c->slots[d->slot];
    Rec *R = &recs[(size_t)d->rec];
    auto prom = d->stage1;
    int64_t *sizes = d->sizes;
    const int64_t read_at = R->read_off, maf_at = R->maf_off;  // one rank: the offsets simply run up (streamed in stage 1)
    std::atomic<int> *out = &rec_out[(size_t)d->rec], *bulk = &rec_bulk[(size_t)d->rec];
    ++*out;
    ++*bulk;
    worker.post([this, sl, R, prom, sizes, read_at, maf_at, out, bulk]() {
      const double w0 = now_us();
      int ok = PBSIM_SUCCEEDED;
      if (wants_text() && hipEventSynchronize(sl->ev_text) != hipSuccess) ok = fail("hipEventSynchronize failed (text emission)");
      if (!ok) {
      } else if (!wants_text()) {
        sizes[0] = sizes[1] = 0;
      } else if (!deflated()) {
        ok = fetch_plain(*sl);
        sizes[0] = sink->on_read_text ? sl->b_info.read_text_bytes : 0;
        sizes[1] = sink->on_maf_text ? sl->b_info.maf_text_bytes : 0;
      } else if (!multi) {
        ok = stream_deflated(*sl, R->ref.unit, read_at, maf_at, &sizes[0], &sizes[1]);
      } else {
        ok = arena_fill(*sl, &sizes[0], &sizes[1]);
      }
      if (!ok) worker_fail();
      bd_worker_busy += now_us() - w0;
      if (trace)
        fprintf(stderr, "[pbsim job r%d] t=%.1f ms   worker: bytes of rec %lld on their way for %.1f ms (%lld + %lld)\n", rank,
                (w0 - t_start) / 1e3, (long long)R->ref.unit, (now_us() - w0) / 1e3, (long long)sizes[0], (long long)sizes[1]);
      delivered_one(out, bulk);
      prom->set_value(ok);
    });
  }
  // collective: exchange the sizes of the pending round, place every rank's bytes, hand the rest to the worker
  // `defer_account`: the caller hands the worker its next round first and calls account_deferred() right after (the
  // statistics of 450 k reads take the main thread ~10 ms, during which the link would idle)
  Slot *acct_slot = nullptr;
  Rec *acct_rec = nullptr;
  int account_deferred() {
    if (!acct_slot) return PBSIM_SUCCEEDED;
    Slot *sl = acct_slot;
    acct_slot = nullptr;
    const double t0 = now_us();
    const int ok = account_of(c, *sl, &acct_rec->st);
    bd_account += now_us() - t0;
    round_closed((int)(acct_rec - recs.data()));
    return ok;
  }
  // The pending round's sizes: wait_pending() = this rank's are known (stage 1 has finished), apply_pending() = all ranks'
  // have been exchanged (S: stride words per rank, the three C words at `at`) -- by an exchange of their own
  // (complete_pending) or inside a round's A or B message (process_round).
  int wait_pending(int64_t c3[3]) {
    c3[0] = c3[1] = c3[2] = 0;
    if (!pending) return PBSIM_SUCCEEDED;
    const double t0 = now_us();
    const int ok1 = pending->stage1_done.get();
    pending->ok1 = ok1;
    pending->t_wait0 = t0;
    pending->t_wait1 = now_us();
    bd_wait_bytes += pending->t_wait1 - t0;
    c3[0] = pending->mine ? pending->sizes[0] : 0;
    c3[1] = pending->mine ? pending->sizes[1] : 0;
    c3[2] = ok1 ? 0 : 1;
    return PBSIM_SUCCEEDED;
  }
  int apply_pending(const std::vector<int64_t> &S, int stride, int at, bool defer_account) {
    if (!pending) return PBSIM_SUCCEEDED;
    std::unique_ptr<Delivery> d = std::move(pending);
    Rec &R = recs[(size_t)d->rec];
    const int ok1 = d->ok1;
    int64_t read_at = R.read_off, maf_at = R.maf_off, read_all = 0, maf_all = 0, bad = 0;
    for (int q = 0; q < W; q++) {
      const int64_t *v = &S[(size_t)q * stride + at];
      if (q < rank) {
        read_at += v[0];
        maf_at += v[1];
      }
      read_all += v[0];
      maf_all += v[1];
      bad += v[2];
    }
    if (bad) {
      if (ok1) peer_failed = true;
      return ok1 ? fail("another rank of the job failed") : check_worker();
    }
    R.read_off += read_all;
    R.maf_off += maf_all;
    if (d->mine) {
      Slot *sl = &c->slots[d->slot];
      Rec *Rp = &R;
      const int slot = d->slot;
      const bool flush = deflated() && multi, plain = wants_text() && !deflated();
      // The per-task statistics are accounted HERE, on the main loop (it waits for the link most of the time), not on the
      // worker, whose time is the link's: rounds are completed in the order of the reads, so accuracy_total keeps its order.
      acct_slot = sl;
      acct_rec = &R;
      // (the round's delivery is counted as outstanding BEFORE its accounting can close the record's last round: a chain of
      // the same record parked on the tail worker must not find rounds_open == 0 and bulk == 0 in between and hand its tail
      // bytes to a front-to-back sink ahead of this round's)
      std::atomic<int> *out = &rec_out[(size_t)d->rec], *bulk = &rec_bulk[(size_t)d->rec];
      if (flush || plain) {
        ++*out;
        ++*bulk;
      }
      if (!defer_account && !account_deferred()) return PBSIM_FAILED;
      if (flush || plain) {
        worker.post([this, sl, Rp, slot, flush, read_at, maf_at, out, bulk]() {
          const int ok = flush ? arena_flush(*sl, Rp->ref.unit, read_at, maf_at) : sink_plain(*sl, Rp->ref.unit, read_at, maf_at);
          if (!ok) worker_fail();
          delivering[slot] = false;
          delivered_one(out, bulk);
        });
      } else {
        delivering[slot] = false;  // streamed in stage 1 (or nothing to deliver): the slot is free
      }
    } else {
      round_closed(d->rec);  // (a block behind the cut: nothing to place, nothing to account)
    }
    if (trace)
      fprintf(stderr, "[pbsim job r%d] t=%.1f ms rec %d delivery: waited %.1f ms for the bytes (%lld + %lld)\n", rank,
              (d->t_wait0 - t_start) / 1e3, d->rec + 1, (d->t_wait1 - d->t_wait0) / 1e3, (long long)d->sizes[0], (long long)d->sizes[1]);
    return check_worker();
  }
  // an exchange of its own for the pending round's sizes (a record's merge, the end of the job, the retry with smaller caps)
  int complete_pending(bool defer_account = false) {
    if (!account_deferred()) return PBSIM_FAILED;
    if (!pending) return PBSIM_SUCCEEDED;
    int64_t mine2[3];
    if (!wait_pending(mine2)) return PBSIM_FAILED;
    std::vector<int64_t> S;
    progress(3, pending->rec, 0, pending->n_per);
    if (!gather(mine2, 3, &S)) return PBSIM_FAILED;
    return apply_pending(S, 3, 0, defer_account);
  }

  // ---- the tail of a record (owner rank only) ------------------------------------------------------------------------
  // The truncated reads behind the cut (pbsim.cpp:3792-3800) depend on each other -- each takes what the one before left of
  // the quota -- so they run as ONE chain on the device (engine.cpp walk_begin(.., chain): kChainReads steps enqueued back
  // to back, no host round trip between two reads), followed by ONE text emission and ONE delivery for the reads it made.
  // (Round 3 stepped the chain from here, read by read: 1.1-2.4 ms and a slot's worth of delivery per read.)
  int tail_begin(Rec &R) {
    const int s = free_tail_slot();
    if (s < 0) {  // both chains' slots are still with the worker (deliveries in flight): tail_poll starts it as soon as one is free
      R.tail_waiting = true;
      return PBSIM_SUCCEEDED;
    }
    R.tail_waiting = false;
    c->cur = s;
    const int64_t remaining = R.quota - R.len_total;
    if (!walk_begin(c, R.ref, R.next_read, chain_reads_for(c, R.ref.len, remaining), remaining, true)) return PBSIM_FAILED;
    R.tail_slot = s;
    return PBSIM_SUCCEEDED;
  }
  // collects the chain if it has finished (or `block`); begins the next one if the quota is still not reached.
  // A chain's bytes follow the record's bulk bytes: nothing is delivered while the sizes of the record's last round are
  // still to be exchanged (complete_pending, a collective, is the main loop's business -- only the owner is here).
  int tail_poll(int rec, bool block) {
    Rec &R = recs[(size_t)rec];
    for (;;) {
      if (R.tail_waiting) {
        if (block) {
          // both chain slots are taken: collect the other records' chains that have come through (at most one cannot be -- the
          // record whose last round's sizes are still pending), then a slot comes back from the chains' worker
          for (size_t o = 0; o < recs.size(); o++)
            if ((int)o != rec && recs[o].tail_slot >= 0 && !tail_poll((int)o, true)) return PBSIM_FAILED;
          // (not a drain of the chains' worker: the hand-over of the chain whose record's last sizes are still pending waits for
          // an exchange that only this thread can make -- but that is one record at most, the other slot's chain comes through)
          // (bounded: two chains of ONE record whose last round is still pending would hold both slots, and their hand-over
          // needs an exchange only this thread can make -- a record that takes more than one chain of truncated reads at its
          // merge; rather an error than a silent spin)
          const double t_wait = now_us();
          while (free_tail_slot() < 0 && !wfailed && !giving_up) {
            std::unique_lock<std::mutex> lk(tail_worker.mu);
            tail_worker.cv_idle.wait_for(lk, std::chrono::milliseconds(1));
            if (now_us() - t_wait > 120e6) return fail("internal: the chains of truncated reads hold both slots and none comes back (120 s)");
          }
        }
        if (!tail_begin(R)) return PBSIM_FAILED;
        if (R.tail_waiting) return block ? fail("internal: no slot for the truncated reads") : PBSIM_SUCCEEDED;
      }
      if (R.tail_slot < 0) return PBSIM_SUCCEEDED;
      c->cur = R.tail_slot;
      if (!block && hipEventQuery(c->s().ev3) != hipSuccess) return PBSIM_SUCCEEDED;
      const double t0 = now_us();
      pbsim_batch_info bi;
      if (!chain_end_finalize(c, R.len_total, &bi)) return PBSIM_FAILED;
      n_tail_reads += bi.n_final;
      const int slot = R.tail_slot;
      Slot *sl = &c->slots[slot];
      Rec *Rp = &R;
      delivering[slot] = true;
      std::atomic<int> *out = &rec_out[(size_t)rec];
      ++*out;
      // The chain's text is compressed (or fetched) at once, into memory of the lambda's own; the hand-over waits until the
      // record's bulk rounds have been exchanged and accounted (its byte offsets are final, its statistics in read order) and
      // their bytes have reached the sink.
      tail_worker.post([this, sl, Rp, slot, out, rec]() {
        int ok = PBSIM_SUCCEEDED;
        std::string zr, zm;  // the chain's members (a few reads: KBs)
        const pbsim_batch_info &bi2 = sl->b_info;
        if (wants_text() && hipEventSynchronize(sl->ev_text) != hipSuccess) ok = fail("hipEventSynchronize failed (text emission)");
        if (!ok) {
        } else if (deflated()) {
          ok = both_lanes(*sl, [&](int which) -> int {
            const int64_t n = which == 0 ? bi2.read_text_bytes : bi2.maf_text_bytes;
            if (!(which == 0 ? sink->on_read_text : sink->on_maf_text) || n == 0) return PBSIM_SUCCEEDED;
            std::string *z = which == 0 ? &zr : &zm;
            const uint8_t *d = which == 0 ? sl->d_read_text.as<uint8_t>() : sl->d_maf_text.as<uint8_t>();
            return deflate_pieces(c, sl->df[which], d, n, [z](const char *p, int64_t k) {
              z->append(p, (size_t)k);
              return PBSIM_SUCCEEDED;
            });
          });
        } else if (wants_text()) {
          ok = fetch_plain(*sl);
        }
        if (!ok) worker_fail();
        const std::string err1 = ok ? std::string() : g_err;
        auto members = std::make_shared<std::pair<std::string, std::string>>(std::move(zr), std::move(zm));
        tail_worker.defer(
            rec,
            [this, rec]() {
              return wfailed.load() || giving_up.load() ||
                     (rec_rounds_open[(size_t)rec].load() == 0 && (!wants_text() || rec_bulk[(size_t)rec].load() == 0));
            },
            [this, sl, Rp, slot, out, ok, members]() {
              int ok2 = ok && !giving_up && !wfailed;
              const int64_t read_at = Rp->read_off + Rp->tail_read, maf_at = Rp->maf_off + Rp->tail_maf;
              int64_t nr = 0, nm = 0;
              if (!ok2) {
              } else if (deflated()) {
                nr = (int64_t)members->first.size();
                nm = (int64_t)members->second.size();
                if (nr && !call_sink(0, Rp->ref.unit, members->first.data(), nr, read_at)) ok2 = fail("sink aborted (read text)");
                if (ok2 && nm && !call_sink(1, Rp->ref.unit, members->second.data(), nm, maf_at)) ok2 = fail("sink aborted (MAF text)");
              } else if (wants_text()) {
                ok2 = sink_plain(*sl, Rp->ref.unit, read_at, maf_at);
                nr = sink->on_read_text ? sl->b_info.read_text_bytes : 0;
                nm = sink->on_maf_text ? sl->b_info.maf_text_bytes : 0;
              }
              if (ok2) ok2 = account_of(c, *sl, &Rp->st);
              Rp->tail_read += nr;
              Rp->tail_maf += nm;
              if (ok && !ok2 && !giving_up && !wfailed) worker_fail();
              delivering[slot] = false;
              delivered_one(out);
            });
      });
      reads_walked += bi.n_final;
      reads_delivered += bi.n_final;
      bases += bi.bases;
      ref_bases += bi.ref_bases;
      maf_columns += bi.maf_columns;
      if (trace)
        fprintf(stderr, "[pbsim job r%d] t=%.1f ms rec %d tail chain from read %lld: %lld reads, %.2f ms to collect\n", rank,
                (t0 - t_start) / 1e3, rec + 1, (long long)R.next_read, (long long)bi.n_final, (now_us() - t0) / 1e3);
      R.next_read += bi.n_final;
      R.len_total = bi.len_total_after;
      R.tail_slot = -1;
      if (!block) bd_tail_steps += now_us() - t0;
      if (R.len_total < R.quota) {
        if (!tail_begin(R)) return PBSIM_FAILED;
      } else {
        R.done = true;
        return PBSIM_SUCCEEDED;
      }
    }
  }

  // ---- merge + completion of a record (collective) -------------------------------------------------------------------
  int finish_record(int rec) {
    Rec &R = recs[(size_t)rec];
    // The record's byte totals must be final: if the round whose sizes are still to be exchanged is one of ITS rounds, exchange
    // them now (the same decision on every rank -- every rank is here at the same point of the round sequence).  A pending
    // round of a LATER record stays pending: waiting for its bytes here would stall the loop for a whole delivery.
    if (pending && pending->rec == rec && !complete_pending()) return PBSIM_FAILED;
    if (!account_deferred()) return PBSIM_FAILED;  // (the statistics of the record's last round may still be due)
    const double tt = now_us();
    int tail_ok = PBSIM_SUCCEEDED;
    std::string tail_err;
    if (rank == R.owner && !tail_poll(rec, true)) {  // carried to the other ranks by the merge's status word
      tail_ok = PBSIM_FAILED;
      tail_err = g_err;
    }
    const double td = now_us();
    bd_tail_block += td - tt;
    wait_record(rec);  // every delivery of the record has reached the sink and the statistics
    bd_drain += now_us() - td;
    // [2] = status: a rank whose tail or worker failed tells the others here instead of leaving them in the merge's collectives
    int64_t extra[3] = {R.tail_read, R.tail_maf, (tail_ok && !wfailed) ? 0 : 1};
    const double t0 = now_us();
    progress(4, rec);
    if (!stats_merge(&R.st, c->p, multi ? comm : nullptr, extra, 3)) return PBSIM_FAILED;
    bd_merge += now_us() - t0;
    if (!tail_ok) return fail(tail_err);
    if (!check_worker()) return PBSIM_FAILED;
    if (extra[2]) {
      peer_failed = true;
      return fail("another rank of the job failed");
    }
    R.merged = R.done = true;
    pbsim_stats st;
    stats_finish(R.st, c->p, R.ref.len, &st);
    // the context's "current unit" statistics follow the last finished record (pbsim_get_stats after a one-record job)
    c->st = R.st;
    c->st.blocks.clear();
    R.st = StatsAcc();
    if (sink && sink->on_record_done &&
        !sink->on_record_done(sink->user, recs[(size_t)rec].ref.unit, &st, R.read_off + extra[0], R.maf_off + extra[1]))
      return fail("sink aborted (record done)");
    return PBSIM_SUCCEEDED;
  }

  // ---- one round -----------------------------------------------------------------------------------------------------
  int begin_round(int rec) {
    Rec &R = recs[(size_t)rec];
    const double remaining = (double)R.quota - R.spec_total;
    // overshoot slightly (0.5 % + 64 reads: the sum of n gamma lengths has a relative spread of ~0.8/sqrt(n)): a round that ends
    // past the quota costs its surplus reads, one that ends short costs a whole extra round
    int64_t n_total = (int64_t)(1.005 * remaining / mean) + 64;
    if (remaining <= 0) n_total = 64;
    int64_t n_per = std::min<int64_t>((n_total + W - 1) / W, R.cap);
    // Ramp-up: nothing moves over the link until the job's first round has been walked and its text emitted, and a full round
    // walks for 12-30 ms.  The first two rounds of a job are a fifth and a half of a full one: bytes flow after ~4 ms, and each
    // round's walk still hides behind the delivery of the round in front of it (a round's bytes take ~3x its walk).  The same
    // on every rank (`rounds` counts the rounds begun).  PBSIM_JOB_RAMP=0 turns it off (A/B).
    static const bool ramp_on = !(exp_env("PBSIM_JOB_RAMP") && atoi(exp_env("PBSIM_JOB_RAMP")) == 0);
    if (ramp_on && rounds < 2) n_per = std::min<int64_t>(n_per, std::max<int64_t>(64, (int64_t)((double)R.cap * (rounds == 0 ? 0.2 : 0.5))));
    n_per = std::max<int64_t>(n_per, 1);
    if (!collect(rec)) return PBSIM_FAILED;  // (an announced record: resident and prepared by now, or waited for here)
    const int s = acquire_slot();
    c->cur = s;
    const double tb = now_us();
    if (trace) fprintf(stderr, "[pbsim job r%d] begin rec %d first=%lld n_per=%lld\n", rank, rec + 1, (long long)R.spec_read, (long long)n_per);
    if (!walk_begin(c, R.ref, R.spec_read + (int64_t)rank * n_per, n_per, -1)) return PBSIM_FAILED;
    bd_begin += now_us() - tb;
    // Will the round end clear of the quota?  Expected bases behind it against the quota, with room for what the estimate can
    // be off by (the mean is measured from the first round on; the sum of n gamma lengths spreads by ~0.8 / sqrt(n)) and for the
    // longest read.  A wrong guess costs time only (process_round: a clear round that does touch the quota emits its text again
    // behind the cut, an unclear one that does not pays one exchange more); PBSIM_JOB_CLEAR=0 / 1 forces it (tests).
    const double after = R.spec_total + (double)W * (double)n_per * mean;
    static const char *force_clear = getenv("PBSIM_JOB_CLEAR");
    const bool clear = force_clear ? atoi(force_clear) != 0
                                   : (double)R.quota - after > 0.04 * (double)W * (double)n_per * mean + 2.0 * (double)std::min<int64_t>(c->p.len_max, R.ref.len);
    fifo.push_back(Round{rec, s, R.spec_read, n_per, mean, clear});
    R.spec_read += (int64_t)W * n_per;
    R.spec_total += (double)W * (double)n_per * mean;
    reads_walked += n_per;
    rounds++;
    return PBSIM_SUCCEEDED;
  }

  // Measured and NOT taken (round 5, profiles/r05_prelaunch_ab.txt): the table fit and the first piece(s) of a round's two
  // compressions launched behind its text emission, while the delivery thread is still moving the round in front
  // (deflate_host.cpp df_begin on the slot's stream), so that the round's delivery starts with its first totals waiting
  // instead of 2-4 ms of kernel latency.  A rank of eight gained 1-2 % (163-165 against 166-169 ms), the one-GPU jobs LOST 1.5-2 %
  // (configs[1] 1102-1110 against 1083 ms, configs[4] 3175-3186 against 3130) with one, two or four pieces launched ahead: the
  // kernels take the GPU from the round that is being delivered -- the same outcome as round 4's "next round's compression on a
  // second worker".  Read only by an EXPERIMENTAL build (knobs.h), off in the product.
  int prelaunch(const pbsim_batch_info &bi) {
    static const bool on = exp_env("PBSIM_DEFLATE_PRELAUNCH") && atoi(exp_env("PBSIM_DEFLATE_PRELAUNCH")) == 1;
    if (!on || !deflated() || bi.n_final <= 0) return PBSIM_SUCCEEDED;
    return deflate_prelaunch(c, c->s(), sink->on_read_text != nullptr, sink->on_maf_text != nullptr, !multi);
  }

  // One round comes back.  Two shapes, chosen when the round was begun (Round::clear, the same on every rank):
  //   clear   text of all n reads emitted at once (finalize_uncut: no cut kernel, no `before` needed) -> wait for the previous
  //           round's bytes -> ONE exchange "AC": pass-0 bases, largest raw length, status | the previous round's sizes.  If the
  //           gathered values say that no read of the round can touch the quota (the usual case) every block's B values follow
  //           from A and the round is done; if not, the cut is placed after all: text again + exchange B.
  //   unclear (the round that is expected to reach the quota) exchange A -> cut + text -> wait for the previous round's bytes
  //           -> exchange "BC": the cut | the previous round's sizes.
  // Neither shape waits for the previous round's bytes before this round's text is on its way, so the link never idles for a
  // text emission.  Message layout (8 words per rank, one layout for all four kinds of exchange so a communicator can tell them
  // apart by pbsim_job_progress alone): A part [0] pass-0 bases [1] code [2] largest raw length | B part [0] n_final
  // [1] need_truncated [2] len_total_after | [3] status of the text sizes | C part [4] has sizes [5] read bytes [6] MAF bytes
  // [7] delivery status.
  static constexpr int kMsg = 8;
  int process_round() {
    const Round rd = fifo.front();
    fifo.pop_front();
    Rec &R = recs[(size_t)rd.rec];
    c->cur = rd.slot;
    const double t0 = now_us();
    int64_t pass0 = 0, code = 0;
    std::string my_err;
    if (!pbsim_batch_walk_end(c, &pass0)) {
      my_err = g_err;
      code = (my_err.rfind("scratch budget exceeded", 0) == 0 && rd.n_per > 1) ? 1 : 2;
    }
    if (wfailed) code = 2;
    const double t1 = now_us();
    bd_wait_walk += t1 - t0;
    // chains of truncated reads that have finished meanwhile: their text and compression run beside this round (the hand-over
    // waits for their record's sizes by itself)
    for (size_t r = 0; r < recs.size(); r++)
      if ((recs[r].tail_slot >= 0 || recs[r].tail_waiting) && !tail_poll((int)r, false)) return PBSIM_FAILED;
    c->cur = rd.slot;
    const int64_t max_raw = c->s().b_max_raw;
    pbsim_batch_info bi;
    memset(&bi, 0, sizeof bi);
    int fin_ok = PBSIM_SUCCEEDED;
    double t_fin = 0;
    std::vector<int64_t> A, B;
    int64_t msg[kMsg] = {pass0, code, max_raw, 0, 0, 0, 0, 0};
    bool c_applied = false;
    if (rd.clear) {
      if (code == 0) {  // every read final: the text leaves now, `before` arrives with the exchange
        const double tf = now_us();
        fin_ok = finalize_uncut(c, &bi) && finalize_text(c, &bi) && prelaunch(bi);
        if (!fin_ok) my_err = g_err;
        t_fin += now_us() - tf;
      }
      msg[3] = fin_ok ? 0 : 1;
      if (pending) {
        msg[4] = 1;
        if (!wait_pending(&msg[5])) return PBSIM_FAILED;
      }
      progress(6, rd.rec, rd.first, rd.n_per);
      if (!gather(msg, kMsg, &A)) return PBSIM_FAILED;
      c_applied = true;
    } else {
      progress(1, rd.rec, rd.first, rd.n_per);
      if (!gather(msg, kMsg, &A)) return PBSIM_FAILED;
    }
    int64_t worst = 0, pass0_sum = 0, before = R.len_total, raw_all = 0, bad_fin = 0;
    for (int q = 0; q < W; q++) {
      worst = std::max(worst, A[(size_t)q * kMsg + 1]);
      pass0_sum += A[(size_t)q * kMsg];
      raw_all = std::max(raw_all, A[(size_t)q * kMsg + 2]);
      bad_fin += A[(size_t)q * kMsg + 3];
      if (q < rank) before += A[(size_t)q * kMsg];
    }
    // (the previous round's sizes came with a clear round's message: place its bytes first, whatever this round turns out to be)
    if (c_applied && worst != 2 && !apply_pending(A, kMsg, 5, true)) return PBSIM_FAILED;
    if (worst == 2 || (worst == 0 && bad_fin)) {
      drop_everything();
      if (wfailed) return check_worker();
      if (code != 2 && fin_ok) peer_failed = true;
      return fail(code == 2 || !fin_ok ? my_err : "another rank of the job failed");
    }
    if (worst == 1) {
      // skewed lengths: some rank's block does not fit its scratch pool.  Everything in flight is void (later rounds were sized
      // with the same cap); every record falls back to what is confirmed and retries with half the cap.
      if (!complete_pending()) return PBSIM_FAILED;
      drop_everything(false);
      for (Rec &r : recs) {
        r.spec_read = r.next_read;
        r.spec_total = (double)r.len_total;
        r.cap = std::min(r.cap, std::max<int64_t>(1, rd.n_per / 2));
        if (r.owner == rank && r.bulk_done && !r.done && !tail_begin(r)) return PBSIM_FAILED;
      }
      return PBSIM_SUCCEEDED;
    }
    // pbsim.cpp:3792-3800: a read stops the loop or is truncated only if len_total + its raw length > quota; len_total in front
    // of any read of the round is at most R.len_total + pass0_sum, so under this bound every read of every block is final
    const bool untouched = R.len_total + pass0_sum + raw_all <= R.quota;
    B.assign((size_t)W * kMsg, 0);
    if (rd.clear && untouched) {
      int64_t lt = R.len_total;
      for (int q = 0; q < W; q++) {
        lt += A[(size_t)q * kMsg];
        B[(size_t)q * kMsg] = rd.n_per;
        B[(size_t)q * kMsg + 2] = lt;
      }
      bi.len_total_after = before + pass0;
      c->s().b_info.len_total_after = bi.len_total_after;
    } else {
      // a failure here travels in the exchange's status word: every rank leaves the job at the same collective
      const double tf = now_us();
      fin_ok = (untouched ? finalize_uncut(c, &bi) : finalize_cut(c, before, &bi)) && finalize_text(c, &bi) && prelaunch(bi);
      if (!fin_ok) my_err = g_err;
      if (untouched) c->s().b_info.len_total_after = bi.len_total_after = before + pass0;
      t_fin += now_us() - tf;
      if (rd.clear) n_clear_missed++;
      int64_t msgB[kMsg] = {bi.n_final, bi.need_truncated_read, bi.len_total_after, fin_ok ? 0 : 1, 0, 0, 0, 0};
      if (!rd.clear && pending) {
        msgB[4] = 1;
        if (!wait_pending(&msgB[5])) return PBSIM_FAILED;
        c_applied = true;
      }
      progress(rd.clear ? 2 : 7, rd.rec, rd.first, rd.n_per);
      if (!gather(msgB, kMsg, &B)) return PBSIM_FAILED;
      int64_t bad_b = 0;
      for (int q = 0; q < W; q++) bad_b += B[(size_t)q * kMsg + 3];
      if (!rd.clear && !bad_b && !apply_pending(B, kMsg, 5, true)) return PBSIM_FAILED;
      if (bad_b) {
        drop_everything();
        if (fin_ok) peer_failed = true;
        return fail(fin_ok ? "another rank of the job failed" : my_err);
      }
    }
    bd_finalize += t_fin;
    const double t2 = now_us();
    int cut = -1;
    for (int q = 0; q < W && cut < 0; q++)
      if (B[(size_t)q * kMsg] < rd.n_per) cut = q;
    const int last_valid = cut < 0 ? W - 1 : cut;
    const bool mine = rank <= last_valid && bi.n_final > 0;
    // ---- delivery: the previous round's sizes have been exchanged (its bytes were on their way while this round's text was
    // emitted); this round's bytes start moving
    if (!c_applied && !complete_pending(true)) return PBSIM_FAILED;  // (nothing was pending)
    pending.reset(new Delivery);
    ++rec_rounds_open[(size_t)rd.rec];
    pending->slot = rd.slot;
    pending->rec = rd.rec;
    pending->mine = mine;
    pending->n_per = rd.n_per;
    submit_stage1(pending.get());
    if (!account_deferred()) return PBSIM_FAILED;  // the previous round's statistics, while the worker moves this round's bytes
    if (mine) {
      reads_delivered += bi.n_final;
      bases += bi.bases;
      ref_bases += bi.ref_bases;
      maf_columns += bi.maf_columns;
    }
    const double t3 = now_us();
    // ---- the record's state, identical on every rank
    const double n_round = (double)W * (double)rd.n_per;
    R.spec_total += (double)pass0_sum - n_round * rd.mean;
    if (n_round >= 1000) {  // re-base the estimate (and what is still in flight) on the measured bases per read
      const double measured = (double)pass0_sum / n_round;
      for (Round &pr : fifo) {
        recs[(size_t)pr.rec].spec_total += (double)W * (double)pr.n_per * (measured - pr.mean);
        pr.mean = measured;
      }
      mean = measured;
    }
    if (cut < 0) {
      R.next_read += (int64_t)W * rd.n_per;
      R.len_total = B[(size_t)(W - 1) * kMsg + 2];
    } else {
      R.next_read += (int64_t)cut * rd.n_per + B[(size_t)cut * kMsg];
      R.len_total = B[(size_t)cut * kMsg + 2];
      R.spec_read = R.next_read;
      R.spec_total = (double)R.len_total;
      if (B[(size_t)cut * kMsg + 1] && R.len_total < R.quota) {  // pbsim.cpp:3795-3800: the next read is a truncated one
        R.owner = cut;
        R.bulk_done = true;
      }
    }
    if (R.len_total >= R.quota) R.bulk_done = true;
    if (R.bulk_done) {
      drop_rounds_of(rd.rec);  // later speculation of this record is void
      if (rank == R.owner) {
        if (!tail_begin(R)) return PBSIM_FAILED;
      } else {
        R.done = true;  // no tail, or another rank's business
      }
    }
    if (trace)
      fprintf(stderr,
              "[pbsim job r%d] t=%.1f ms rec %d round first=%lld n=%lldx%d %s final=%lld cut=%d wait_walk=%.1f finalize=%.1f handover=%.1f "
              "comm=%.1f ms\n",
              rank, (t0 - t_start) / 1e3, rd.rec + 1, (long long)rd.first, (long long)rd.n_per, W,
              rd.clear ? (untouched ? "clear" : "clear-missed") : "cutting", (long long)bi.n_final, cut, (t1 - t0) / 1e3, t_fin / 1e3,
              (t3 - t2) / 1e3, comm_us / 1e3);
    return PBSIM_SUCCEEDED;
  }

  int run() {
    const int n = (int)recs.size();
    int merged = 0;
    worker.start(c->device);
    tail_worker.start(c->device);
    for (;;) {
      if (!check_worker()) return PBSIM_FAILED;
      // ---- keep the pipeline full.  Open records = [merged, merged + open_max): their statistics are being collected and
      // their tails may be running (two chain slots: a third chain waits its turn).  interleave == 1: the earliest open record
      // that still lacks reads in flight -- a record's rounds run back to back and its tail hides behind the next record;
      // interleave == k > 1: of the first k open records the one that is furthest behind, so that k records advance side by
      // side (a caller that writes a file pair per record then has 2k files to write at a time, pbsim_job_set_interleave).
      const int open_max = std::max(2, interleave);
      while (bulk_in_flight() < depth) {
        int cand = -1;
        const int hi = std::min(n, merged + open_max);
        double best = 2.0;
        for (int r = merged; r < hi; r++) {
          const Rec &R = recs[(size_t)r];
          if (R.bulk_done || (double)R.quota - R.spec_total <= 0) continue;
          if (interleave <= 1) {
            cand = r;
            break;
          }
          const double f = R.spec_total / (double)std::max<int64_t>(1, R.quota);
          if (f < best) {
            best = f;
            cand = r;
          }
        }
        if (cand < 0) {
          // everything that is expected to be needed is in flight; a record whose rounds all came back short of the quota
          // shows up here with nothing in flight: top it up
          for (int r = merged; r < hi && cand < 0; r++) {
            bool has = false;
            for (const Round &pr : fifo) has |= pr.rec == r;
            if (!recs[(size_t)r].bulk_done && !has) cand = r;
          }
          if (cand >= 0) n_topup++;
        }
        if (cand < 0) {
          if (hi >= n) break;
          // a record behind the open ones is next: merge the oldest first (a collective at a point of the round sequence that
          // every rank reaches alike; by then its tail reads have finished beside the next record's rounds, and the merge
          // waits for that record's own deliveries only -- the worker keeps moving the later rounds' bytes)
          if (!recs[(size_t)merged].bulk_done) break;  // its rounds are still in flight: pop first
          if (!finish_record(merged)) return PBSIM_FAILED;
          merged++;
          continue;
        }
        if (!begin_round(cand)) return PBSIM_FAILED;
      }
      // ---- tails make progress between rounds
      for (int r = 0; r < n; r++)
        if ((recs[(size_t)r].tail_slot >= 0 || recs[(size_t)r].tail_waiting) && !tail_poll(r, false)) return PBSIM_FAILED;
      if (fifo.empty()) {
        for (int r = 0; r < n; r++)
          if (!recs[(size_t)r].bulk_done) return fail("internal: a record lacks reads but nothing is in flight");
        break;
      }
      if (!process_round()) return PBSIM_FAILED;
    }
    // (the last round's sizes are exchanged by the merge of its record: the records in front of it merge while its bytes move)
    for (; merged < n; merged++) {
      if (!recs[(size_t)merged].bulk_done) return fail("internal: a record was left unfinished");
      if (!finish_record(merged)) return PBSIM_FAILED;
    }
    if (!complete_pending()) return PBSIM_FAILED;
    worker.finish();
    tail_worker.finish();
    return check_worker();
  }
};

}  // namespace

extern "C" {

// the record joins the job: an event behind its preparation (pbsim_job_run waits for it record by record, not for the whole
// prefetch stream), and -- the adding thread may not be the one that runs the job -- under the lock, with a wake-up
static int job_publish(pbsim_ctx *c, std::unique_ptr<JobRecord> r) {
  if (!r->h_flags) HIP_OK(hipHostMalloc((void **)&r->h_flags, sizeof(DeviceFlags), hipHostMallocDefault));
  HIP_OK(hipMemcpyAsync(r->h_flags, r->flags.p, sizeof(DeviceFlags), hipMemcpyDeviceToHost, c->prefetch_stream));
  HIP_OK(hipEventCreateWithFlags(&r->ready, hipEventDisableTiming));
  HIP_OK(hipEventRecord(r->ready, c->prefetch_stream));
  {
    std::lock_guard<std::mutex> lk(c->job_mu);
    const size_t i = c->job_records.size();
    if (!c->job_expect_len.empty() && (i >= c->job_expect_len.size() || c->job_expect_len[i] != r->len)) {
      c->job_feed_failed = true;
      c->job_feed_err = "pbsim_job_add_record: not the record pbsim_job_expect announced (count or length)";
      c->job_cv.notify_all();
      return fail(c->job_feed_err);
    }
    c->job_records.push_back(std::move(r));
  }
  c->job_cv.notify_all();
  return PBSIM_SUCCEEDED;
}

// a record to fill: the buffers of a dropped record of the same length when there is one (pbsim_job_begin keeps them), else new
static std::unique_ptr<JobRecord> job_new_record(pbsim_ctx *c, int64_t len) {
  {
    std::lock_guard<std::mutex> lk(c->job_mu);
    for (size_t i = 0; i < c->job_spare.size(); i++)
      if (c->job_spare[i]->len == len && c->job_spare[i]->seq.bytes >= (size_t)len + 64) {
        std::unique_ptr<JobRecord> r = std::move(c->job_spare[i]);
        c->job_spare.erase(c->job_spare.begin() + (long)i);
        r->ref = RefDesc();
        return r;
      }
    // No record of the job before has this length: its buffers (2 B per base, up to 64 GB for a record group at the CLI's
    // limit) would stay resident beside the new job's uploads, on top of pools that were sized while one group was resident
    // (ADVICE r5).  Spares whose length an announced record still waits for are kept; the rest go back now.
    for (size_t i = c->job_spare.size(); i-- > 0;) {
      bool wanted = false;
      for (size_t k = c->job_records.size(); k < c->job_expect_len.size() && !wanted; k++)
        wanted = c->job_expect_len[k] == c->job_spare[i]->len;
      if (!wanted) c->job_spare.erase(c->job_spare.begin() + (long)i);
    }
  }
  std::unique_ptr<JobRecord> r(new JobRecord);
  r->len = len;
  return r;
}

static int job_add(pbsim_ctx *c, const void *seq, int64_t len, hipMemcpyKind kind) {
  if (!c || !seq) return fail("pbsim_job_add_record: bad argument");
  const bool trace = getenv("PBSIM_TRACE") != nullptr;
  const double ta = now_us();
  NEED_DEVICE(c);
  if (c->p.strategy != PBSIM_STRATEGY_WGS || c->p.method == PBSIM_METHOD_SAMPLE)
    return fail("pbsim_job_add_record: the job pipeline runs --strategy wgs with --method errhmm or qshmm");
  if (len < 1) return fail("Reference is too short.");
  if (len > 1000000000LL) return fail("Reference is too long. Acceptable length <= 1000000000.");
  HIP_OK(hipSetDevice(c->device));
  if (!c->prefetch_stream) HIP_OK(hipStreamCreateWithFlags(&c->prefetch_stream, hipStreamNonBlocking));
  std::unique_ptr<JobRecord> r = job_new_record(c, len);
  HIP_OK(r->seq.ensure((size_t)len + 64, true));
  HIP_OK(hipMemcpyAsync(r->seq.p, seq, (size_t)len, kind, c->prefetch_stream));
  const double tb = now_us();
  HIP_OK(hipStreamSynchronize(c->prefetch_stream));  // the caller may reuse (or free) its buffer; the preparation stays asynchronous
  const double tc = now_us();
  HIP_OK(hipMemsetAsync(r->seq.as<uint8_t>() + len, 0, 64, c->prefetch_stream));
  if (!prepare_enqueue(c, r->seq.as<uint8_t>(), r->hp, r->tiles, r->flags, len, c->prefetch_stream)) return PBSIM_FAILED;
  if (trace)
    fprintf(stderr, "[pbsim job_add] record of %lld: alloc + copy enqueued %.2f ms, copy (behind the record in front) waited for %.2f ms, K0 enqueued %.2f ms\n",
            (long long)len, (tb - ta) / 1e3, (tc - tb) / 1e3, (now_us() - tc) / 1e3);
  return job_publish(c, std::move(r));
}
int pbsim_job_add_record(pbsim_ctx *c, const uint8_t *seq, int64_t len) { return job_add(c, seq, len, hipMemcpyHostToDevice); }

// The record as its FASTA lines (line feeds included): uploaded as they lie in the caller's memory -- a mapped file will do --
// and squeezed on the GPU (k_lines_*), so the host never touches the bases; `len` = bytes - line feeds (the caller counted
// them to print the reference stats) is checked against what the GPU kept.
int pbsim_job_add_record_lines(pbsim_ctx *c, const uint8_t *lines, int64_t bytes, int64_t len) {
  if (!c || !lines || bytes < len) return fail("pbsim_job_add_record_lines: bad argument");
  NEED_DEVICE(c);
  if (c->p.strategy != PBSIM_STRATEGY_WGS || c->p.method == PBSIM_METHOD_SAMPLE)
    return fail("pbsim_job_add_record: the job pipeline runs --strategy wgs with --method errhmm or qshmm");
  if (len < 1) return fail("Reference is too short.");
  if (len > 1000000000LL) return fail("Reference is too long. Acceptable length <= 1000000000.");
  HIP_OK(hipSetDevice(c->device));
  if (!c->prefetch_stream) HIP_OK(hipStreamCreateWithFlags(&c->prefetch_stream, hipStreamNonBlocking));
  std::unique_ptr<JobRecord> r = job_new_record(c, len);
  HIP_OK(r->seq.ensure((size_t)len + 64, true));
  const int64_t n_tiles = (bytes + 4095) / 4096;
  HIP_OK(c->d_lines.ensure((size_t)bytes + 64));
  HIP_OK(c->d_lines_tmp.ensure((size_t)(n_tiles + 1 + n_tiles / 1024 + 16) * 8));
  HIP_OK(hipMemcpyAsync(c->d_lines.p, lines, (size_t)bytes, hipMemcpyHostToDevice, c->prefetch_stream));
  int64_t *tmp = c->d_lines_tmp.as<int64_t>();
  launch_squeeze_lines(c->d_lines.as<uint8_t>(), bytes, r->seq.as<uint8_t>(), tmp + 1, tmp + 1 + n_tiles + 1, tmp, c->prefetch_stream);
  int64_t kept = -1;
  HIP_OK(hipMemcpyAsync(&kept, tmp, 8, hipMemcpyDeviceToHost, c->prefetch_stream));
  HIP_OK(hipStreamSynchronize(c->prefetch_stream));  // the staging buffer is the next record's too; the preparation stays asynchronous
  if (kept != len) return fail("pbsim_job_add_record_lines: `len` is not the number of bytes that are not line feeds");
  HIP_OK(hipMemsetAsync(r->seq.as<uint8_t>() + len, 0, 64, c->prefetch_stream));
  if (!prepare_enqueue(c, r->seq.as<uint8_t>(), r->hp, r->tiles, r->flags, len, c->prefetch_stream)) return PBSIM_FAILED;
  return job_publish(c, std::move(r));
}

// C1: rank `root` holds the record in host memory; it uploads it and comm->broadcast carries the device bytes to every
// other rank's GPU (over xGMI when the communicator is RCCL).  Without a broadcast callback every rank passes the bytes.
int pbsim_job_add_record_comm(pbsim_ctx *c, const uint8_t *seq, int64_t len, const pbsim_comm *comm, int32_t root) {
  if (!comm || comm->world <= 1 || !comm->broadcast) {
    if (!seq) return fail("pbsim_job_add_record_comm: no broadcast callback, so every rank must pass the record");
    return job_add(c, seq, len, hipMemcpyHostToDevice);
  }
  if (!c) return fail("pbsim_job_add_record_comm: bad argument");
  NEED_DEVICE(c);
  HIP_OK(hipSetDevice(c->device));
  // every rank learns whether every rank is ready BEFORE the broadcast: a rank that returned early would leave the others in it
  DevBuf tmp;
  int ok = PBSIM_SUCCEEDED;
  if (len < 1 || len > 1000000000LL) ok = fail("pbsim_job_add_record_comm: bad length");
  if (ok && tmp.ensure((size_t)len, true) != hipSuccess) ok = fail("pbsim_job_add_record_comm: out of device memory");
  if (ok && comm->rank == root) {
    if (!seq) ok = fail("pbsim_job_add_record_comm: the root rank must pass the record");
    else if (hipMemcpy(tmp.p, seq, (size_t)len, hipMemcpyHostToDevice) != hipSuccess) ok = fail("pbsim_job_add_record_comm: upload failed");
  }
  if (ok && hipDeviceSynchronize() != hipSuccess) ok = fail("pbsim_job_add_record_comm: device error");
  if (comm->all_reduce_i64) {
    const std::string keep = g_err;
    int64_t bad = ok ? 0 : 1;
    if (!comm->all_reduce_i64(comm->user, &bad, 1, PBSIM_OP_MAX)) return fail("pbsim_comm.all_reduce_i64 failed");
    if (bad) return ok ? fail("pbsim_job_add_record_comm: another rank failed") : fail(keep);
  } else if (!ok) {
    return PBSIM_FAILED;
  }
  if (!comm->broadcast(comm->user, tmp.p, len, root, 1)) return fail("pbsim_comm.broadcast failed");
  HIP_OK(hipDeviceSynchronize());  // whatever stream the communicator used: the bytes are in `tmp` before they are copied on
  if (!job_add(c, tmp.p, len, hipMemcpyDeviceToDevice)) return PBSIM_FAILED;
  HIP_OK(hipStreamSynchronize(c->prefetch_stream));  // tmp is released on return
  return PBSIM_SUCCEEDED;
}
int pbsim_job_add_record_device(pbsim_ctx *c, const void *seq_device, int64_t len) {
  return job_add(c, seq_device, len, hipMemcpyDeviceToDevice);
}
int64_t pbsim_job_records(pbsim_ctx *c) {
  if (!c) return -1;
  std::lock_guard<std::mutex> lk(c->job_mu);
  return (int64_t)c->job_records.size();
}

int pbsim_job_clear(pbsim_ctx *c) { return pbsim_job_begin(c, 1); }

int pbsim_job_begin(pbsim_ctx *c, int64_t first_record) {
  if (!c || first_record < 1) return fail("pbsim_job_begin: bad argument");
  c->job_first_unit = first_record;
  // Q15 state in front of the job's first record: a job that starts at record 1 starts a genome (only what an explicit
  // census pass has seen counts), a later job of the same genome carries on from the job before it.  Every pbsim_job_run
  // starts from this value, so re-running a job gives the same bytes as its first run.
  c->hp11_before_job = first_record == 1 ? c->hp11_explicit : c->bias.hp11_seen;
  if (first_record == 1 && c->census_from_job) {  // the census of the previous genome's records: the next job takes its own
    c->census_done = c->census_from_job = false;
    hp_bias_default(&c->bias);
    c->class_tables_dirty = true;
  }
  c->bias.hp11_seen = c->hp11_before_job;
  if (c->prefetch_stream) (void)hipStreamSynchronize(c->prefetch_stream);
  for (Slot &sl : c->slots) {
    if (sl.stream) (void)hipStreamSynchronize(sl.stream);
    sl.b_enqueued = sl.b_walked = sl.b_finalized = false;
  }
  {
    std::lock_guard<std::mutex> lk(c->job_mu);
    c->job_spare.clear();  // (one job's worth at most: what the job before this one left is let go now)
    for (auto &r : c->job_records) {
      if (r->ready) (void)hipEventDestroy(r->ready);
      r->ready = nullptr;
      c->job_spare.push_back(std::move(r));
    }
    c->job_records.clear();
    c->job_expect_len.clear();
    c->job_feed_failed = false;
    c->job_feed_err.clear();
  }
  return PBSIM_SUCCEEDED;
}

// The job's records announced in advance: pbsim_job_run may then be called before they have all been added, and
// pbsim_job_add_record* may be called from ANOTHER thread while it runs (in order; each must have the announced length).  The
// job begins with record 1 as soon as that one is resident and prepared, and a later record's first round when that record is
// -- its upload / broadcast and preparation hide behind the rounds of the records in front of it.  (--hp-del-bias != 1 needs the
// homopolymer census of ALL records before the first read, pbsim.cpp:677-696: such a job waits for every record first.)
int pbsim_job_expect(pbsim_ctx *c, int64_t n_records, const int64_t *lens) {
  if (!c || n_records < 1 || !lens) return fail("pbsim_job_expect: bad argument");
  std::lock_guard<std::mutex> lk(c->job_mu);
  if ((int64_t)c->job_records.size() > n_records) return fail("pbsim_job_expect: more records have been added already");
  for (int64_t i = 0; i < n_records; i++) {
    if (lens[i] < 1 || lens[i] > 1000000000LL) return fail("pbsim_job_expect: bad length");
    if (i < (int64_t)c->job_records.size() && c->job_records[(size_t)i]->len != lens[i])
      return fail("pbsim_job_expect: a record added already has another length");
  }
  c->job_expect_len.assign(lens, lens + n_records);
  c->job_feed_failed = false;
  return PBSIM_SUCCEEDED;
}
// the thread that feeds an announced job gives up: a pbsim_job_run waiting for a record fails with `why`
int pbsim_job_feed_abort(pbsim_ctx *c, const char *why) {
  if (!c) return fail("bad argument");
  {
    std::lock_guard<std::mutex> lk(c->job_mu);
    c->job_feed_failed = true;
    c->job_feed_err = why && *why ? why : "the caller gave up feeding the job's records";
  }
  c->job_cv.notify_all();
  return PBSIM_SUCCEEDED;
}

int64_t pbsim_job_sam_header(pbsim_ctx *c, int64_t record, char *buf, int64_t cap) {
  if (!c) return -1;
  const std::string h = sam_header_text(c, record);
  if (buf && cap > (int64_t)h.size()) memcpy(buf, h.c_str(), h.size() + 1);
  return (int64_t)h.size();
}
int64_t pbsim_job_bam_header(pbsim_ctx *c, int64_t record, char *buf, int64_t cap) {  // "BAM\1" l_text text n_ref=0 (SAMv1 4.2)
  if (!c) return -1;
  const std::string h = sam_header_text(c, record);
  const int64_t n = 4 + 4 + (int64_t)h.size() + 4;
  if (buf && cap >= n) {
    memcpy(buf, "BAM\1", 4);
    const uint32_t l = (uint32_t)h.size(), zero = 0;
    memcpy(buf + 4, &l, 4);
    memcpy(buf + 8, h.data(), h.size());
    memcpy(buf + 8 + h.size(), &zero, 4);
  }
  return n;
}

int pbsim_job_breakdown(pbsim_ctx *c, double out[16]) {
  if (!c || !out) return fail("bad argument");
  memcpy(out, c->job_breakdown, sizeof c->job_breakdown);
  return PBSIM_SUCCEEDED;
}

int pbsim_job_set_interleave(pbsim_ctx *c, int records) {
  if (!c || records < 1) return fail("pbsim_job_set_interleave: bad argument");
  c->job_interleave = records;
  return PBSIM_SUCCEEDED;
}

int pbsim_job_progress(pbsim_ctx *c, int64_t out[8]) {
  if (!c || !out) return fail("bad argument");
  memcpy(out, c->job_progress, sizeof c->job_progress);
  return PBSIM_SUCCEEDED;
}

int pbsim_job_counters(pbsim_ctx *c, int64_t out[8]) {
  if (!c || !out) return fail("bad argument");
  memcpy(out, c->job_counters, sizeof c->job_counters);
  return PBSIM_SUCCEEDED;
}

// `*settled`: the other ranks know of the failure (it came through / went into a status word) or the communicator has been
// aborted already; every other failing return is this rank's alone, and pbsim_job_run releases the others.
static int job_run_impl(pbsim_ctx *c, const pbsim_comm *comm, const pbsim_record_sink *sink, bool *settled) {
  size_t n = 0;
  std::vector<int64_t> lens;
  bool streaming = false;
  {
    std::lock_guard<std::mutex> lk(c->job_mu);
    streaming = !c->job_expect_len.empty();
    if (streaming) lens = c->job_expect_len;
    else
      for (auto &r : c->job_records) lens.push_back(r->len);
    n = lens.size();
  }
  if (n == 0) return fail("pbsim_job_run: no records (pbsim_job_add_record)");
  if (sink && c->deflate != 0 && c->deflate != 3)
    return fail("pbsim_job_run: pbsim_set_deflate must cover both sinks or none (mask 0, 3 or 7)");
  HIP_OK(hipSetDevice(c->device));
  Job J;
  J.c = c;
  J.comm = comm;
  J.sink = (sink && (sink->on_read_text || sink->on_maf_text || sink->on_record_done)) ? sink : nullptr;
  J.W = comm ? comm->world : 1;
  J.rank = comm ? comm->rank : 0;
  // A communicator of ONE rank has nobody to exchange with and its collectives are skipped -- unless PBSIM_COMM_ALWAYS=1
  // (test hook): the job then runs the several-rank protocol, every exchange included, through the communicator of one.  That
  // is how a single-GPU box drives pbsim_job_run through real RCCL calls (tests/rccl_native_driver.py).
  J.multi = J.W > 1 || (comm && comm->all_gather_i64 && comm->all_reduce_i64 && getenv("PBSIM_COMM_ALWAYS") &&
                        atoi(getenv("PBSIM_COMM_ALWAYS")) != 0);
  J.trace = getenv("PBSIM_TRACE") != nullptr;
  J.t_start = now_us();
  const char *jd = exp_env("PBSIM_JOB_DEPTH");
  // Rounds in flight (+ one pending delivery < the slots of the rounds).  Three keep a GPU full whose text stays in HBM or whose
  // link carries a fraction of the job (several ranks).  A job that delivers all its bytes over one or two links is bound by
  // them (80 ms a round against 18 ms of walk): one round in flight is as fast (1334 vs 1336 ms), its walk does not share the
  // GPU with two others (18.5 instead of 29.3 ms a launch) and half the slots stay unallocated.
  const int W = J.W;
  const bool multi = J.multi;
  const bool delivers = J.sink && (J.sink->on_read_text || J.sink->on_maf_text);
  {
    const char *il = getenv("PBSIM_JOB_INTERLEAVE");
    J.interleave = std::max(1, std::min(64, il ? atoi(il) : c->job_interleave));
  }
  // Several ranks (round 4, measured per rank against virtual ranks -- profiles/r04_replay_depth_ab.txt): every rank delivers
  // its own blocks over its own link, so a rank of eight is in the same regime as one GPU alone: configs[4] 594 / 592 / 570 ms
  // per rank with 3 / 2 / 1 rounds in flight, configs[1] 190 / 189 / 191.
  J.depth = std::max(1, std::min(kMaxSlots - 3, jd ? atoi(jd) : (delivers ? 1 : 3)));
  c->bias.hp11_seen = c->hp11_before_job;  // every run of the job starts from the same Q15 state
  J.seen11 = c->bias.hp11_seen;
  J.recs.resize(n);
  for (int ts = Job::kTailSlot; ts < kMaxSlots; ts++)
    for (DfLane &L : c->slots[ts].df) L.own_streams = true;
  J.rec_out.reset(new std::atomic<int>[n]);
  J.rec_bulk.reset(new std::atomic<int>[n]);
  J.rec_rounds_open.reset(new std::atomic<int>[n]);
  for (size_t i = 0; i < n; i++) J.rec_out[i] = J.rec_bulk[i] = J.rec_rounds_open[i] = 0;
  int64_t max_quota = 0;
  for (size_t i = 0; i < n; i++) {  // what the plan needs of a record is its length; the rest arrives with the record (Job::collect)
    Rec &R = J.recs[i];
    R.ref.len = lens[i];
    R.ref.unit = c->job_first_unit + (int64_t)i;
    R.quota = quota_of(c, lens[i]);
    R.st.keep_values = multi;
    max_quota = std::max(max_quota, R.quota);
  }
  // ---- the records' preparation (upload + k_hp_*) has been running since pbsim_job_add_record.  A job whose records are all
  // here collects them all now (and releases the staging of pbsim_job_add_record_lines: a record's worth of HBM the rounds can
  // use); an announced job (pbsim_job_expect) collects record 1 and every later record in front of its first round.
  // pbsim.cpp:677-696: with --hp-del-bias != 1 the census of ALL records comes before the first read -- every record first.
  const bool census_needed = c->p.hp_del_bias != 1 && (!c->census_done || c->census_from_job);
  // A follow-on job of the same genome (pbsim_job_begin with first_record > 1) cannot take the census from its own records:
  // the reference's pass covers ALL records before the first read, and a bias table per record group would silently differ
  // from it.  The caller runs pbsim_add_hp_census / pbsim_finish_hp_census over the whole genome first (the CLI does).
  if (c->p.hp_del_bias != 1 && c->job_first_unit > 1 && census_needed)
    return fail("pbsim_job_run: a job that continues a genome (pbsim_job_begin first_record > 1) with --hp-del-bias != 1 needs the "
                "homopolymer census of ALL records first (pbsim_add_hp_census per record, then pbsim_finish_hp_census; pbsim.cpp:677-696)");
  if (census_needed) {
    // (the census first, with the tables still as they are: collect() builds the class tables a record's walks will use)
    int64_t census[kHpSlots] = {0};
    bool any11 = c->hp11_before_job;
    for (size_t i = 0; i < n; i++) {
      const DeviceFlags *f = J.record_arrived((int)i);
      if (!f) return PBSIM_FAILED;
      for (int k = 0; k < kHpSlots; k++) census[k] += (int64_t)f->hpfreq[k];
      any11 |= f->hpfreq[11] > 0;
    }
    HpBias nb = c->bias;
    hp_bias_from_census(c->p.hp_del_bias, census, &nb);
    nb.hp11_seen = any11;  // the pre-pass has run get_genome_seq over every record (hpfreq[11]++, pbsim.cpp:1058)
    if (!c->census_done || memcmp(nb.bias, c->bias.bias, sizeof nb.bias) != 0) c->class_tables_dirty = true;
    c->bias = nb;
    c->census_done = c->census_from_job = true;
    J.seen11 = c->bias.hp11_seen;
  } else if (c->p.hp_del_bias != 1 && c->hp11_explicit) {
    c->bias.hp11_seen = true;
    J.seen11 = true;
  }
  if (!J.collect(streaming && !census_needed ? 0 : (int)n - 1)) return PBSIM_FAILED;
  if (!streaming) {
    HIP_OK(hipStreamSynchronize(c->prefetch_stream));
    c->d_lines.release();
    c->d_lines_tmp.release();
  }
  // ---- how much of its 2 L + pad columns does a read of this model use?  A context that has not walked yet finds out on the
  // first reads of the job's first record (8192 reads, ~1 ms; every rank the same reads, so every rank the same answer):
  // pools and rounds are then sized for what the rows really take (engine.cpp: scratch factor) -- a third less than 2 L.
  if (!c->scratch_factor_fixed && c->need_seen == 0) {
    c->cur = 0;
    // (a pool of a few MB -- tests -- may not hold that many reads at the starting factor: fewer then, or none: the job
    // keeps the reference's bound until its rounds have reported)
    for (int64_t n_probe = std::min<int64_t>(8192, batch_capacity_for(c, J.recs[0].ref.len)); n_probe >= 32; n_probe /= 4) {
      if (!walk_begin(c, J.recs[0].ref, 1, n_probe, -1)) return PBSIM_FAILED;
      const int ok = pbsim_batch_walk_end(c, nullptr);
      c->s().b_walked = false;
      if (ok) break;
      if (g_err.rfind("scratch budget exceeded", 0) != 0) return PBSIM_FAILED;
    }
  }
  const double sf = scratch_factor_of(c);
  // ---- batch size: a few rounds per record and rank, not below what keeps a walk longer than its longest read
  const int P = c->p.pass_num;
  const int regions = has_quality(c) ? 3 : 2;
  const char *jr = exp_env("PBSIM_JOB_ROUNDS");  // experiment knob: rounds per record the batches are sized for
  // (several ranks that deliver their bytes: two rounds per record and rank -- every delivery call pays a start-up of a few ms,
  // and a rank of eight has a quarter of a record's bytes per round to spread it over; measured per rank against virtual
  // ranks, configs[4] on eight: 520-537 ms with four rounds per record, 496-504 with two, 525+ with one -- profiles/r04_replay_rounds_ab.txt)
  const bool delivers_text = sink && (sink->on_read_text || sink->on_maf_text);
  // (text left in HBM -- the job is bound by its walks -- : TWO rounds per record since round 5.  Same-box sweep of rounds per
  // record x rounds in flight x lane / wave split, profiles/r05_job_rounds_ab.txt: configs[1] 197-201 Gbases/s at four rounds
  // per record, 208-215 at two (one: 211; the split, the wave walker's workgroups and the ramp-up rounds move nothing there): a
  // round of 850 000 reads amortises its longest lanes better and the wave walker's share of the GPU time drops.  A job that
  // delivers its text is bound by its link either way: 54.2 / 53.6 / 54.6 Gbases/s at two / three / four, it keeps four.)
  const int rounds_per_record = std::max(1, jr ? atoi(jr) : (delivers_text ? (multi ? 2 : kRoundsPerRecord) : 2));
  double target = (double)max_quota * P / ((double)rounds_per_record * W);
  target = std::max(target, std::min(kMinBatchBases, (double)max_quota * P / W));
  if (J.sink && (J.sink->on_read_text || J.sink->on_maf_text)) target = std::min(target, kSinkBatchBases);
  if (c->scratch_auto) {
    size_t free_b = 0, total_b = 0;
    HIP_OK(hipMemGetInfo(&free_b, &total_b));
    if (streaming) {
      // An announced job (pbsim_job_expect) sizes its pools while records 2.. are still on their way: what they will take --
      // sequence + homopolymer lengths, 2 B per base, less what a spare buffer of the same length already holds, and the
      // growth of the lines staging -- is not free, and how far the feeder thread has got must not move the batch sizes from
      // run to run: charge every announced record that has no buffers yet, whatever has arrived since.
      std::lock_guard<std::mutex> lk(c->job_mu);
      std::vector<int64_t> spare;
      for (auto &sp : c->job_spare) spare.push_back(sp->len);
      size_t pending = 0, lines_max = 0;
      for (size_t i = 0; i < n; i++) {
        if (i < c->job_records.size()) {
          // (already resident: inside free_b's complement; counted as free again so that the figure does not depend on the feeder)
          free_b += (size_t)(2 * lens[i] + 192);
        }
        auto it = std::find(spare.begin(), spare.end(), lens[i]);
        if (it != spare.end() && i >= c->job_records.size()) {
          spare.erase(it);
          continue;
        }
        pending += (size_t)(2 * lens[i] + 192) + (size_t)(lens[i] / 1024 + 64) * 16;
        lines_max = std::max(lines_max, (size_t)(lens[i] + lens[i] / 40 + 64));
      }
      if (c->d_lines.bytes > 0 && lines_max > c->d_lines.bytes) pending += lines_max - c->d_lines.bytes;
      free_b = free_b > pending ? free_b - pending : 0;
    }
    size_t held = 0, held_text = 0;
    for (Slot &sl : c->slots) {
      held += sl.d_scratch.bytes;
      held_text += sl.d_read_text.bytes + sl.d_maf_text.bytes;
    }
    {
      // every slot a round passes through (in flight, pending, with the worker) keeps its scratch rows AND its text until the
      // bytes are delivered: bound the batch so that all of them fit 75 % of what the GPU has left
      // FASTQ 2.0 | SAM text ~6.1 | BAM records ~3.5 (bases 0.5, qualities 1, the ip and pw arrays 1 each); MAF 2.23; the text
      // buffers are grown with 12.5 % of slack (DevBuf::ensure)
      const double text_per_base = 1.125 * (P > 1 ? (c->bam_output ? 5.8 : 8.4) : 4.25);
      const double scratch_per_base = (double)regions * sf * 1.12 * 1.08 + 0.1;
      // (rounds in flight + one whose delivery is pending + one with the worker; nothing is held back when the text stays put)
      const bool delivering_text = J.sink && (J.sink->on_read_text || J.sink->on_maf_text);
      const double slots_used = (double)J.depth + (delivering_text ? 2.0 : 1.0);
      const char *ff = getenv("PBSIM_JOB_FIT");  // experiment knob: share of the free HBM the slots may take
      const double fit = (ff ? atof(ff) : 0.75) * (double)(free_b + held + held_text) / (slots_used * (text_per_base + scratch_per_base));
      target = std::min(target, std::max(fit, 1.0e8));
    }
    // what batch_capacity_for() charges a read: `regions` rows of 2 * length + pad columns, 12 % slack for the per-wave rounding
    const double mean_len = std::max(1.0, c->hdr.mean_len);
    // (a read yields ~0.97 of its length in bases, and a round overshoots its share of the quota by 0.5 %: 8 % headroom)
    const double want = (target / P / mean_len) * P * ((double)regions * (sf * mean_len + kScratchPad) * 1.12 + 64.0) * 1.08 + (64 << 20);
    const double share = std::min(48.0 * (1LL << 30), 0.14 * (double)(free_b + held));
    const int64_t auto_b = (int64_t)std::max(256.0 * (1 << 20), std::min(want, share));
    // (a pool changes size only when it is clearly wrong: what the slots hold moves the free-memory figures from run to run, and
    // re-allocating every pool of a context stalls the job that does it for seconds; the caps below follow the pool.  A
    // context that switches between kinds of jobs -- delivered through a sink, then text left in HBM: other slot counts,
    // other batch sizes -- keeps its larger buffers AND allocates the new slots': pbsim_release_pools() in between)
    if ((double)auto_b > 1.10 * (double)c->scratch_budget || c->scratch_budget > 2 * auto_b) c->scratch_budget = auto_b;
  }
  if (const char *tr = getenv("PBSIM_JOB_TARGET_RANKS")) {  // test hook: a batch target per rank, "a,b,c" (ranks that see
    std::vector<double> v;                                 // different free memory size their rounds from different numbers)
    for (const char *q = tr; *q;) {
      char *e = nullptr;
      v.push_back(strtod(q, &e));
      if (e == q) break;
      q = *e == ',' ? e + 1 : e;
    }
    if (!v.empty() && v[(size_t)J.rank % v.size()] > 0) target = v[(size_t)J.rank % v.size()];
  }
  // Every rank must size its rounds alike: a round gives rank r the reads first + r * n_per .. and the cut compares every
  // block with the same n_per, so EVERY input of the caps that a rank derives from its own GPU (the pool, and the batch target
  // once the free memory bounds it) goes through one MIN over the ranks.
  std::vector<int64_t> agree(n + 1);
  agree[0] = c->scratch_budget;
  J.progress(5, -1);
  if (multi) {
    if (!comm->all_reduce_i64(comm->user, agree.data(), 1, PBSIM_OP_MIN)) return fail("pbsim_comm.all_reduce_i64 failed");
    c->scratch_budget = agree[0];
  }
  J.mean = 0.97 * c->hdr.mean_len;  // bases a read yields (deletions outweigh insertions in most models); measured from round 1 on
  for (size_t i = 0; i < n; i++) {
    Rec &R = J.recs[i];
    const double m = std::min<double>(c->hdr.mean_len, (double)R.ref.len);
    R.cap = std::max<int64_t>(1, std::min<int64_t>(batch_capacity_for(c, R.ref.len), (int64_t)(1.08 * target / P / m) + 64));
    agree[i + 1] = R.cap;
  }
  if (multi) {
    if (!comm->all_reduce_i64(comm->user, agree.data(), (int64_t)agree.size(), PBSIM_OP_MIN)) return fail("pbsim_comm.all_reduce_i64 failed");
    for (size_t i = 0; i < n; i++) J.recs[i].cap = agree[i + 1];
  }
  J.mean = std::min<double>(J.mean, (double)lens[0]);
  for (Slot &sl : c->slots) sl.b_enqueued = sl.b_walked = sl.b_finalized = false;
  {
    const char *am = getenv("PBSIM_PINNED_ARENA_MB");  // pinned host memory one lane's arena may hold (several ranks: a round's
    J.arena_max = (size_t)(am && atoll(am) > 0 ? atoll(am) : 16384) << 20;  // members wait there for their offsets)
  }
  // (several ranks: a failure in front of the first collective of the round loop is agreed on first, like every later one)
  int ready = (!J.deflated() || ensure_deflate_ready(c)) ? PBSIM_SUCCEEDED : PBSIM_FAILED;  // before the worker and its lane threads use the tables
  if (multi) {
    const std::string keep = g_err;
    int64_t bad = ready ? 0 : 1;
    if (!comm->all_reduce_i64(comm->user, &bad, 1, PBSIM_OP_MAX)) return fail("pbsim_comm.all_reduce_i64 failed");
    if (bad) {
      *settled = true;
      return ready ? fail("another rank of the job failed") : fail(keep);
    }
  } else if (!ready) {
    return PBSIM_FAILED;
  }
  // (Round 2 ran the job's walks at three workgroups per CU -- 41 KB of LDS asked for -- because the round loop then waited on
  // many short kernels queued behind walk workgroups.  Since the loop no longer waits for text emission and statistics that
  // costs more than it gives: same-box A/B in round 3, whole job in HBM 180 -> 193 Gbases/s (ERRHMM) and 161 -> 172 G subread
  // bases/s (QSHMM x10) at the batch primitives' five per CU, delivered job unchanged -- profiles/r03_occupancy_ab.txt.)
  // A job that compresses its output runs ONE lane-walk workgroup per CU (81 KB of LDS asked for; round 4).  The deflate
  // workgroups need 35 KB of LDS each: beside five walk workgroups per CU they found none, beside three (41 KB, rounds 2-3) one
  // per CU, beside one walk workgroup two -- and the lanes, which waited ~20 ms of every 78 ms round for their kernels, keep
  // the link busier: configs[1] 47.6-48.0 -> 51.9-52.2 Gbases/s, configs[4] 48.8 -> 54.8, configs[2] 31.3 -> 33.3 (same-box
  // A/B, profiles/r04_walk_occupancy_ab.txt).  The walk of a 5-Gbase round takes 29 instead of 17 ms and still hides behind the
  // round's 75 ms of delivery.  Two walk workgroups per CU (54 KB) leave room for one deflate workgroup and gain nothing; 100 KB
  // leave room for one and lose; fewer wave-walker workgroups, a high-priority stream for the lanes' kernels and CU masks for
  // the walk streams (experiments) all lose or gain nothing.
  const int keep_lds = c->walk_lds_kb;
  if (J.deflated()) c->walk_lds_kb = std::max(c->walk_lds_kb, 81);
  c->defer_text_sync = true;  // the round loop does not wait for a round's text emission; the delivery thread does
  int ok = J.run();
  c->defer_text_sync = false;
  c->walk_lds_kb = keep_lds;
  for (Slot &sl : c->slots)  // text left in HBM (no sink): its emission ends with the job
    if (sl.stream && hipStreamSynchronize(sl.stream) != hipSuccess && ok) ok = fail("hipStreamSynchronize failed at the end of the job");
  if (!ok) {
    const std::string keep = g_err;
    // A failure that only this rank knows of (not one that came in through a collective's status word) would leave the other
    // ranks waiting in their next collective: the communicator's abort, when it has one, releases them.  Without it the caller
    // must tear the process group down (include/pbsim3_amd.h).
    if (multi && !J.peer_failed && comm->abort) comm->abort(comm->user);
    *settled = true;
    J.drop_everything();
    J.worker.finish();
    J.tail_worker.finish();
    g_err = keep;
  }
  for (Slot &sl : c->slots)
    for (DfLane &L : sl.df) L.arena_trim();
  {
    const double wall = now_us() - J.t_start;
    double *b = c->job_breakdown;
    b[0] = wall;
    b[1] = J.bd_wait_walk;
    b[2] = J.bd_finalize;
    b[3] = J.bd_wait_bytes;
    b[4] = J.comm_us;
    b[5] = J.bd_account;
    b[6] = J.bd_tail_block;
    b[7] = J.bd_drain;
    b[8] = J.bd_slot_wait + J.bd_record_wait;  // (an announced record that had not arrived when its first round was due)
    b[9] = J.bd_merge;
    b[10] = J.bd_begin;
    b[11] = J.bd_tail_steps;
    b[12] = J.bd_worker_busy;
    b[13] = (double)J.n_topup;
    b[14] = (double)J.n_tail_reads;
    b[15] = (double)J.depth;
  }
  c->job_counters[0] = J.reads_walked;
  c->job_counters[1] = J.reads_delivered;
  c->job_counters[2] = J.rounds;
  c->job_counters[3] = J.bases;
  c->job_counters[4] = (int64_t)(now_us() - J.t_start);
  c->job_counters[5] = (int64_t)J.comm_us;
  c->job_counters[6] = J.ref_bases;
  c->job_counters[7] = J.maf_columns;
  return ok;
}

int pbsim_job_run(pbsim_ctx *c, const pbsim_comm *comm, const pbsim_record_sink *sink) {
  if (!c) return fail("bad argument");
  NEED_DEVICE(c);
  if (comm && (comm->world < 1 || comm->rank < 0 || comm->rank >= comm->world)) return fail("pbsim_comm: bad rank / world");
  if (comm && comm->world > 1 && (!comm->all_gather_i64 || !comm->all_reduce_i64))
    return fail("pbsim_comm: all_gather_i64 and all_reduce_i64 must be set");
  bool settled = false;
  const int ok = job_run_impl(c, comm, sink, &settled);
  if (!ok && !settled && comm && comm->world > 1 && comm->abort) {
    // a failure in front of the job's first collective (no records, a HIP error while the records' preparation is collected,
    // no memory for the pools): the other ranks are on their way into it
    const std::string keep = g_err;
    comm->abort(comm->user);
    g_err = keep;
  }
  return ok;
}

int64_t pbsim_format_stats(const pbsim_params *p, const pbsim_stats *s, int64_t unit, char *buf, int64_t cap) {
  if (!p || !s) return -1;
  char t[1024];
  int k = 0;
  if (p->strategy == PBSIM_STRATEGY_WGS) {  // pbsim.cpp:5541-5564
    k += snprintf(t + k, sizeof t - k, ":::: Simulation stats (ref.%ld) ::::\n\n", (long)unit);
    k += snprintf(t + k, sizeof t - k, "read num. : %ld\n", (long)s->res_num);
    k += snprintf(t + k, sizeof t - k, "depth : %lf\n", s->res_depth);
  } else {
    k += snprintf(t + k, sizeof t - k, ":::: Simulation stats ::::\n\n");
    k += snprintf(t + k, sizeof t - k, "read num. : %ld\n", (long)s->res_num);
  }
  k += snprintf(t + k, sizeof t - k, "read length mean (SD) : %f (%f)\n", s->res_len_mean, s->res_len_sd);
  k += snprintf(t + k, sizeof t - k, "read length min : %ld\n", (long)s->res_len_min);
  k += snprintf(t + k, sizeof t - k, "read length max : %ld\n", (long)s->res_len_max);
  k += snprintf(t + k, sizeof t - k, "read accuracy mean (SD) : %f (%f)\n", s->res_accuracy_mean, s->res_accuracy_sd);
  k += snprintf(t + k, sizeof t - k, "substitution rate. : %f\n", s->res_sub_rate);
  k += snprintf(t + k, sizeof t - k, "insertion rate. : %f\n", s->res_ins_rate);
  k += snprintf(t + k, sizeof t - k, "deletion rate. : %f\n", s->res_del_rate);
  k += snprintf(t + k, sizeof t - k, "\n");
  if (buf && cap > k) memcpy(buf, t, (size_t)k + 1);
  return k;
}

}  // extern "C"
