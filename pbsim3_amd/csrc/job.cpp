// job.cpp -- the whole wgs job as ONE pipeline of read batches across all records and all ranks (include/pbsim3_amd.h,
// "the whole job on one or several GPUs").
//
// The reference's main() simulates its records one after the other and each record's quota loop read by read
// (pbsim.cpp:667-759, 3792-4080).  Here every record is resident in HBM and the loop runs as rounds:
//
//   round   = W blocks of n reads of one record (W = ranks), rank r walks block r; speculative, un-truncated lengths
//   pop     = rounds finish in the order they were begun; per round two small all-gathers:
//               A  pass-0 bases of every block          -> every rank's len_total in front of its block (quota prefix)
//               B  (n_final, need_truncated, len_total_after, text bytes) of every block -> the cut, and every rank's
//                  byte range inside the record's streams
//   cut     = the first block the quota rule stops in (pbsim.cpp:3792-3800); later blocks / rounds of the record are void
//   tail    = the truncated reads behind the cut, one at a time (each depends on the one before), on the cut's rank only,
//             on a slot of their own, polled between rounds -- no other rank waits for them
//   merge   = per record, at a fixed point of the round sequence (before record n+2 begins, or at the end): statistics of
//             all ranks summed (C2), accuracy_total folded in read order; then on_record_done everywhere
//
// Record n+1's rounds begin as soon as record n has enough reads in flight, so the last text emission and the tail of
// a record run beside the next record's walks.  All decisions derive from gathered values, identical on every rank: the
// ranks stay in lockstep without any control message.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <deque>
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
  int64_t read_off = 0, maf_off = 0;            // bytes of the bulk rounds, all ranks (identical everywhere)
  int64_t tail_read = 0, tail_maf = 0;          // owner: bytes of its tail reads
  StatsAcc st;
  bool merged = false;
};

struct Round {
  int rec, slot;
  int64_t first, n_per;
  double mean;               // bases per read assumed when it was begun
};

struct Job {
  pbsim_ctx *c;
  const pbsim_comm *comm;
  const pbsim_record_sink *sink;
  int rank = 0, W = 1;
  std::vector<Rec> recs;
  std::deque<Round> fifo;
  double mean = 0;
  int depth = 3;
  int64_t reads_walked = 0, reads_delivered = 0, rounds = 0, bases = 0, ref_bases = 0, maf_columns = 0;
  double comm_us = 0;
  bool trace = false;
  double t_start = 0;

  int gather(const int64_t *send, int64_t n, std::vector<int64_t> *recv) {
    recv->assign((size_t)W * n, 0);
    if (W == 1) {
      memcpy(recv->data(), send, (size_t)n * 8);
      return PBSIM_SUCCEEDED;
    }
    const double t0 = now_us();
    const int ok = comm->all_gather_i64(comm->user, send, n, recv->data());
    comm_us += now_us() - t0;
    return ok ? PBSIM_SUCCEEDED : fail("pbsim_comm.all_gather_i64 failed");
  }

  bool slot_busy(int s) const {
    for (const Round &r : fifo)
      if (r.slot == s) return true;
    for (const Rec &r : recs)
      if (r.tail_slot == s) return true;
    return false;
  }
  int free_slot() const {
    for (int s = 0; s < kMaxSlots; s++)
      if (!slot_busy(s)) return s;
    return -1;
  }
  int bulk_in_flight() const { return (int)fifo.size(); }

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
  void drop_everything() {
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

  // ---- delivery of the selected slot's finalized batch ---------------------------------------------------------------
  // text mode: sizes are known from the batch info; deflate mode on several ranks: the batch is compressed into the
  // lanes' pinned arenas first (sizes unknown before), the sizes are exchanged, then the pieces go out at their offsets.
  int send_plain(int rec, int64_t read_at, int64_t maf_at) {
    const pbsim_batch_info &bi = c->s().b_info;
    if (!sink) return PBSIM_SUCCEEDED;
    const bool want_r = sink->on_read_text && bi.read_text_bytes, want_m = sink->on_maf_text && bi.maf_text_bytes;
    if (want_r) HIP_OK(c->s().h_read_text.ensure((size_t)bi.read_text_bytes + 16));
    if (want_m) HIP_OK(c->s().h_maf_text.ensure((size_t)bi.maf_text_bytes + 16));
    if ((want_r || want_m) &&
        !pbsim_batch_fetch(c, want_r ? (char *)c->s().h_read_text.p : nullptr, want_m ? (char *)c->s().h_maf_text.p : nullptr))
      return PBSIM_FAILED;
    if (want_r && !sink->on_read_text(sink->user, recs[(size_t)rec].ref.unit, (const char *)c->s().h_read_text.p, bi.read_text_bytes, read_at))
      return fail("sink aborted (read text)");
    if (want_m && !sink->on_maf_text(sink->user, recs[(size_t)rec].ref.unit, (const char *)c->s().h_maf_text.p, bi.maf_text_bytes, maf_at))
      return fail("sink aborted (MAF text)");
    return PBSIM_SUCCEEDED;
  }

  // compressed, streamed piece by piece at running offsets (one rank, or the tail owner): returns the bytes sent
  int send_deflated_stream(int rec, int64_t read_at, int64_t maf_at, int64_t *read_gz, int64_t *maf_gz) {
    const pbsim_batch_info &bi = c->s().b_info;
    Slot &sl = c->s();
    *read_gz = *maf_gz = 0;
    auto lane = [&](int which, std::string *err) -> int {
      const bool is_read = which == 0;
      const int64_t n = is_read ? bi.read_text_bytes : bi.maf_text_bytes;
      auto cb = is_read ? sink->on_read_text : sink->on_maf_text;
      int64_t *sent = is_read ? read_gz : maf_gz;
      const int64_t base = is_read ? read_at : maf_at;
      const uint8_t *d = is_read ? sl.d_read_text.as<uint8_t>() : sl.d_maf_text.as<uint8_t>();
      if (!cb || n == 0) return PBSIM_SUCCEEDED;
      const int ok = deflate_pieces(c, sl.df[which], d, n, [&](const char *z, int64_t k) {
        if (!cb(sink->user, recs[(size_t)rec].ref.unit, z, k, base + *sent)) return fail(is_read ? "sink aborted (read text)" : "sink aborted (MAF text)");
        *sent += k;
        return PBSIM_SUCCEEDED;
      });
      if (!ok && err) *err = g_err;
      return ok;
    };
    if (c->deflate_parallel && bi.read_text_bytes && bi.maf_text_bytes && sink->on_read_text && sink->on_maf_text) {
      if (!ensure_deflate_ready(c)) return PBSIM_FAILED;
      int ok_read = PBSIM_SUCCEEDED;
      std::string err_read;
      std::thread t([&]() {
        (void)hipSetDevice(c->device);
        ok_read = lane(0, &err_read);
      });
      const int ok_maf = lane(1, nullptr);
      t.join();
      if (!ok_read) return fail(err_read);  // the error string is thread local: carry the second thread's over
      return ok_maf;
    }
    return lane(0, nullptr) && lane(1, nullptr);
  }

  // compress the batch into the lanes' arenas (sizes out), to be flushed by arena_flush once the offsets are known
  int arena_fill(int64_t *read_gz, int64_t *maf_gz) {
    const pbsim_batch_info &bi = c->s().b_info;
    Slot &sl = c->s();
    *read_gz = *maf_gz = 0;
    for (int which = 0; which < 2; which++) {
      DfLane &L = sl.df[which];
      L.arena_reset();
      const int64_t n = which == 0 ? bi.read_text_bytes : bi.maf_text_bytes;
      auto cb = which == 0 ? sink->on_read_text : sink->on_maf_text;
      if (!cb || n == 0) continue;
      const uint8_t *d = which == 0 ? sl.d_read_text.as<uint8_t>() : sl.d_maf_text.as<uint8_t>();
      bool oom = false;
      const std::function<char *(int64_t)> place = [&](int64_t k) -> char * {
        char *p = L.arena_reserve(k);
        if (!p) oom = true;
        return p;
      };
      if (!deflate_pieces(c, L, d, n, [](const char *, int64_t) { return PBSIM_SUCCEEDED; }, &place))
        return oom ? fail("out of pinned host memory for a compressed batch") : PBSIM_FAILED;
      int64_t tot = 0;
      for (const auto &sg : L.arena_segs) tot += sg.second;
      (which == 0 ? *read_gz : *maf_gz) = tot;
    }
    return PBSIM_SUCCEEDED;
  }
  int arena_flush(int rec, int64_t read_at, int64_t maf_at) {
    Slot &sl = c->s();
    for (int which = 0; which < 2; which++) {
      auto cb = which == 0 ? sink->on_read_text : sink->on_maf_text;
      int64_t at = which == 0 ? read_at : maf_at;
      for (const auto &sg : sl.df[which].arena_segs) {
        if (!cb(sink->user, recs[(size_t)rec].ref.unit, sg.first, sg.second, at)) return fail(which == 0 ? "sink aborted (read text)" : "sink aborted (MAF text)");
        at += sg.second;
      }
      sl.df[which].arena_segs.clear();
    }
    return PBSIM_SUCCEEDED;
  }

  // ---- the tail of a record (owner rank only) ------------------------------------------------------------------------
  int tail_begin(Rec &R) {
    if (free_slot() < 0)  // another record's tail holds the spare slot: let it finish first
      for (size_t r = 0; r < recs.size(); r++)
        if (recs[r].tail_slot >= 0 && !tail_poll((int)r, true)) return PBSIM_FAILED;
    const int s = free_slot();
    if (s < 0) return fail("internal: no free slot for a truncated read");
    c->cur = s;
    if (!walk_begin(c, R.ref, R.next_read, 1, R.quota - R.len_total)) return PBSIM_FAILED;
    R.tail_slot = s;
    return PBSIM_SUCCEEDED;
  }
  // one step of the chain if the read in flight has finished (or `block`); begins the next truncated read if one is due
  int tail_poll(int rec, bool block) {
    Rec &R = recs[(size_t)rec];
    while (R.tail_slot >= 0) {
      c->cur = R.tail_slot;
      if (!block && hipEventQuery(c->s().ev3) != hipSuccess) return PBSIM_SUCCEEDED;
      const double t0 = now_us();
      if (!pbsim_batch_walk_end(c, nullptr)) return PBSIM_FAILED;
      pbsim_batch_info bi;
      if (!finalize_cut(c, R.len_total, &bi) || !finalize_text(c, &bi)) return PBSIM_FAILED;
      int64_t nr = bi.read_text_bytes, nm = bi.maf_text_bytes;
      if (sink && c->deflate == 3) {
        if (!send_deflated_stream(rec, R.read_off + R.tail_read, R.maf_off + R.tail_maf, &nr, &nm)) return PBSIM_FAILED;
      } else if (!send_plain(rec, R.read_off + R.tail_read, R.maf_off + R.tail_maf)) {
        return PBSIM_FAILED;
      }
      if (!account_slot(c, &R.st)) return PBSIM_FAILED;
      R.tail_read += nr;
      R.tail_maf += nm;
      reads_walked += 1;
      reads_delivered += bi.n_final;
      bases += bi.bases;
      ref_bases += bi.ref_bases;
      maf_columns += bi.maf_columns;
      if (trace)
        fprintf(stderr, "[pbsim job r%d] t=%.1f ms rec %d tail read %lld: %.1f ms\n", rank, (t0 - t_start) / 1e3, rec + 1,
                (long long)R.next_read, (now_us() - t0) / 1e3);
      R.next_read += bi.n_final;
      R.len_total = bi.len_total_after;
      R.tail_slot = -1;
      if (R.len_total < R.quota) {
        if (!tail_begin(R)) return PBSIM_FAILED;
      } else {
        R.done = true;
      }
    }
    return PBSIM_SUCCEEDED;
  }

  // ---- merge + completion of a record (collective) -------------------------------------------------------------------
  int finish_record(int rec) {
    Rec &R = recs[(size_t)rec];
    if (rank == R.owner && !tail_poll(rec, true)) return PBSIM_FAILED;
    int64_t extra[2] = {R.tail_read, R.tail_maf};
    const double t0 = now_us();
    if (!stats_merge(&R.st, c->p, W > 1 ? comm : nullptr, extra, 2)) return PBSIM_FAILED;
    comm_us += now_us() - t0;
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
    n_per = std::max<int64_t>(n_per, 1);
    const int s = free_slot();
    if (s < 0) return fail("internal: no free slot");
    c->cur = s;
    if (!walk_begin(c, R.ref, R.spec_read + (int64_t)rank * n_per, n_per, -1)) return PBSIM_FAILED;
    fifo.push_back(Round{rec, s, R.spec_read, n_per, mean});
    R.spec_read += (int64_t)W * n_per;
    R.spec_total += (double)W * (double)n_per * mean;
    reads_walked += n_per;
    rounds++;
    return PBSIM_SUCCEEDED;
  }

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
    const double t1 = now_us();
    std::vector<int64_t> A, B;
    const int64_t sendA[2] = {pass0, code};
    if (!gather(sendA, 2, &A)) return PBSIM_FAILED;
    int64_t worst = 0, pass0_sum = 0, before = R.len_total;
    for (int q = 0; q < W; q++) {
      worst = std::max(worst, A[(size_t)q * 2 + 1]);
      pass0_sum += A[(size_t)q * 2];
      if (q < rank) before += A[(size_t)q * 2];
    }
    if (worst == 2) {
      drop_everything();
      return fail(code == 2 ? my_err : "another rank of the job failed");
    }
    if (worst == 1) {
      // skewed lengths: some rank's block does not fit its scratch pool.  Everything in flight is void (later rounds were sized
      // with the same cap); every record falls back to what is confirmed and this record retries with half the cap.
      drop_everything();
      for (Rec &r : recs) {
        r.spec_read = r.next_read;
        r.spec_total = (double)r.len_total;
        if (r.owner == rank && r.bulk_done && !r.done && !tail_begin(r)) return PBSIM_FAILED;
      }
      for (Rec &r : recs) r.cap = std::min(r.cap, std::max<int64_t>(1, rd.n_per / 2));
      return PBSIM_SUCCEEDED;
    }
    pbsim_batch_info bi;
    if (!finalize_cut(c, before, &bi)) return PBSIM_FAILED;
    if (!finalize_text(c, &bi)) return PBSIM_FAILED;
    const double t2 = now_us();
    const int64_t sendB[5] = {bi.n_final, bi.need_truncated_read, bi.len_total_after, bi.read_text_bytes, bi.maf_text_bytes};
    if (!gather(sendB, 5, &B)) return PBSIM_FAILED;
    int cut = -1;
    for (int q = 0; q < W && cut < 0; q++)
      if (B[(size_t)q * 5] < rd.n_per) cut = q;
    const int last_valid = cut < 0 ? W - 1 : cut;
    const bool mine = rank <= last_valid && bi.n_final > 0;
    const bool deflated = sink && c->deflate == 3;
    // ---- deliver
    int64_t sizes_r[2] = {mine ? bi.read_text_bytes : 0, mine ? bi.maf_text_bytes : 0};
    std::vector<int64_t> S;
    if (deflated && W > 1) {
      c->cur = rd.slot;
      if (mine && !arena_fill(&sizes_r[0], &sizes_r[1])) return PBSIM_FAILED;
      if (!gather(sizes_r, 2, &S)) return PBSIM_FAILED;
    } else {
      S.assign((size_t)W * 2, 0);
      for (int q = 0; q <= last_valid; q++) {
        S[(size_t)q * 2] = B[(size_t)q * 5] > 0 ? B[(size_t)q * 5 + 3] : 0;
        S[(size_t)q * 2 + 1] = B[(size_t)q * 5] > 0 ? B[(size_t)q * 5 + 4] : 0;
      }
    }
    int64_t read_at = R.read_off, maf_at = R.maf_off, read_all = 0, maf_all = 0;
    for (int q = 0; q < W; q++) {
      if (q < rank) {
        read_at += S[(size_t)q * 2];
        maf_at += S[(size_t)q * 2 + 1];
      }
      read_all += S[(size_t)q * 2];
      maf_all += S[(size_t)q * 2 + 1];
    }
    c->cur = rd.slot;
    if (mine) {
      if (deflated && W > 1) {
        if (!arena_flush(rd.rec, read_at, maf_at)) return PBSIM_FAILED;
      } else if (deflated) {
        int64_t nr = 0, nm = 0;
        if (!send_deflated_stream(rd.rec, read_at, maf_at, &nr, &nm)) return PBSIM_FAILED;
        read_all = nr;
        maf_all = nm;
      } else if (!send_plain(rd.rec, read_at, maf_at)) {
        return PBSIM_FAILED;
      }
      if (!account_slot(c, &R.st)) return PBSIM_FAILED;
      reads_delivered += bi.n_final;
      bases += bi.bases;
      ref_bases += bi.ref_bases;
      maf_columns += bi.maf_columns;
    }
    R.read_off += read_all;
    R.maf_off += maf_all;
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
      R.len_total = B[(size_t)(W - 1) * 5 + 2];
    } else {
      R.next_read += (int64_t)cut * rd.n_per + B[(size_t)cut * 5];
      R.len_total = B[(size_t)cut * 5 + 2];
      R.spec_read = R.next_read;
      R.spec_total = (double)R.len_total;
      if (B[(size_t)cut * 5 + 1] && R.len_total < R.quota) {  // pbsim.cpp:3795-3800: the next read is a truncated one
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
              "[pbsim job r%d] t=%.1f ms rec %d round first=%lld n=%lldx%d final=%lld cut=%d wait_walk=%.1f finalize=%.1f deliver=%.1f "
              "comm=%.1f ms\n",
              rank, (t0 - t_start) / 1e3, rd.rec + 1, (long long)rd.first, (long long)rd.n_per, W, (long long)bi.n_final, cut,
              (t1 - t0) / 1e3, (t2 - t1) / 1e3, (t3 - t2) / 1e3, comm_us / 1e3);
    return PBSIM_SUCCEEDED;
  }

  int run() {
    const int n = (int)recs.size();
    int merged = 0;
    for (;;) {
      // ---- keep the pipeline full: the earliest record that still lacks reads in flight
      while (bulk_in_flight() < depth) {
        int cand = -1;
        for (int r = 0; r < n && cand < 0; r++)
          if (!recs[(size_t)r].bulk_done && (double)recs[(size_t)r].quota - recs[(size_t)r].spec_total > 0) cand = r;
        if (cand < 0) {
          // everything that is expected to be needed is in flight; a record whose rounds all came back short of the quota
          // shows up here with nothing in flight: top it up
          for (int r = 0; r < n && cand < 0; r++) {
            bool has = false;
            for (const Round &pr : fifo) has |= pr.rec == r;
            if (!recs[(size_t)r].bulk_done && !has) cand = r;
          }
          if (cand < 0) break;
        }
        if (cand >= merged + 2) {
          // at most two records' statistics are open at a time: merge the oldest first (a collective at a point of the round
          // sequence that every rank reaches alike)
          if (!recs[(size_t)merged].bulk_done) break;  // its rounds are still in flight: pop first
          if (!finish_record(merged)) return PBSIM_FAILED;
          merged++;
          continue;
        }
        if (free_slot() < 0) break;
        // one slot stays free for a tail (the owner is not known in advance)
        int busy = 0;
        for (int s = 0; s < kMaxSlots; s++) busy += slot_busy(s);
        if (busy >= kMaxSlots - 1) break;
        if (!begin_round(cand)) return PBSIM_FAILED;
      }
      // ---- tails make progress between rounds
      for (int r = 0; r < n; r++)
        if (recs[(size_t)r].tail_slot >= 0 && !tail_poll(r, false)) return PBSIM_FAILED;
      if (fifo.empty()) break;
      if (!process_round()) return PBSIM_FAILED;
    }
    for (; merged < n; merged++) {
      if (!recs[(size_t)merged].bulk_done) return fail("internal: a record was left unfinished");
      if (!finish_record(merged)) return PBSIM_FAILED;
    }
    return PBSIM_SUCCEEDED;
  }
};

}  // namespace

extern "C" {

static int job_add(pbsim_ctx *c, const void *seq, int64_t len, hipMemcpyKind kind) {
  if (!c || !seq) return fail("pbsim_job_add_record: bad argument");
  NEED_DEVICE(c);
  if (c->p.strategy != PBSIM_STRATEGY_WGS || c->p.method == PBSIM_METHOD_SAMPLE)
    return fail("pbsim_job_add_record: the job pipeline runs --strategy wgs with --method errhmm or qshmm");
  if (len < 1) return fail("Reference is too short.");
  if (len > 1000000000LL) return fail("Reference is too long. Acceptable length <= 1000000000.");
  HIP_OK(hipSetDevice(c->device));
  if (!c->prefetch_stream) HIP_OK(hipStreamCreateWithFlags(&c->prefetch_stream, hipStreamNonBlocking));
  std::unique_ptr<JobRecord> r(new JobRecord);
  r->len = len;
  HIP_OK(r->seq.ensure((size_t)len + 64, true));
  HIP_OK(hipMemcpyAsync(r->seq.p, seq, (size_t)len, kind, c->prefetch_stream));
  if (kind == hipMemcpyHostToDevice) HIP_OK(hipStreamSynchronize(c->prefetch_stream));  // the caller may reuse its buffer
  HIP_OK(hipMemsetAsync(r->seq.as<uint8_t>() + len, 0, 64, c->prefetch_stream));
  if (!prepare_enqueue(c, r->seq.as<uint8_t>(), r->hp, r->tiles, r->flags, len, c->prefetch_stream)) return PBSIM_FAILED;
  c->job_records.push_back(std::move(r));
  return PBSIM_SUCCEEDED;
}
int pbsim_job_add_record(pbsim_ctx *c, const uint8_t *seq, int64_t len) { return job_add(c, seq, len, hipMemcpyHostToDevice); }

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
  if (len < 1 || len > 1000000000LL) return fail("pbsim_job_add_record_comm: bad length");
  DevBuf tmp;
  HIP_OK(tmp.ensure((size_t)len, true));
  if (comm->rank == root) {
    if (!seq) return fail("pbsim_job_add_record_comm: the root rank must pass the record");
    HIP_OK(hipMemcpy(tmp.p, seq, (size_t)len, hipMemcpyHostToDevice));
  }
  HIP_OK(hipDeviceSynchronize());
  if (!comm->broadcast(comm->user, tmp.p, len, root, 1)) return fail("pbsim_comm.broadcast failed");
  if (!job_add(c, tmp.p, len, hipMemcpyDeviceToDevice)) return PBSIM_FAILED;
  HIP_OK(hipStreamSynchronize(c->prefetch_stream));  // tmp is released on return
  return PBSIM_SUCCEEDED;
}
int pbsim_job_add_record_device(pbsim_ctx *c, const void *seq_device, int64_t len) {
  return job_add(c, seq_device, len, hipMemcpyDeviceToDevice);
}
int64_t pbsim_job_records(pbsim_ctx *c) { return c ? (int64_t)c->job_records.size() : -1; }

int pbsim_job_clear(pbsim_ctx *c) { return pbsim_job_begin(c, 1); }

int pbsim_job_begin(pbsim_ctx *c, int64_t first_record) {
  if (!c || first_record < 1) return fail("pbsim_job_begin: bad argument");
  c->job_first_unit = first_record;
  if (c->prefetch_stream) (void)hipStreamSynchronize(c->prefetch_stream);
  for (Slot &sl : c->slots) {
    if (sl.stream) (void)hipStreamSynchronize(sl.stream);
    sl.b_enqueued = sl.b_walked = sl.b_finalized = false;
  }
  c->job_records.clear();
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

int pbsim_job_counters(pbsim_ctx *c, int64_t out[8]) {
  if (!c || !out) return fail("bad argument");
  memcpy(out, c->job_counters, sizeof c->job_counters);
  return PBSIM_SUCCEEDED;
}

int pbsim_job_run(pbsim_ctx *c, const pbsim_comm *comm, const pbsim_record_sink *sink) {
  if (!c) return fail("bad argument");
  NEED_DEVICE(c);
  if (c->job_records.empty()) return fail("pbsim_job_run: no records (pbsim_job_add_record)");
  if (comm && (comm->world < 1 || comm->rank < 0 || comm->rank >= comm->world)) return fail("pbsim_comm: bad rank / world");
  if (comm && comm->world > 1 && (!comm->all_gather_i64 || !comm->all_reduce_i64))
    return fail("pbsim_comm: all_gather_i64 and all_reduce_i64 must be set");
  if (sink && c->deflate != 0 && c->deflate != 3)
    return fail("pbsim_job_run: pbsim_set_deflate must cover both sinks or none (mask 0, 3 or 7)");
  HIP_OK(hipSetDevice(c->device));
  Job J;
  J.c = c;
  J.comm = comm;
  J.sink = (sink && (sink->on_read_text || sink->on_maf_text || sink->on_record_done)) ? sink : nullptr;
  J.W = comm ? comm->world : 1;
  J.rank = comm ? comm->rank : 0;
  J.trace = getenv("PBSIM_TRACE") != nullptr;
  J.t_start = now_us();
  const char *jd = getenv("PBSIM_JOB_DEPTH");
  J.depth = std::max(1, std::min(kMaxSlots - 2, jd ? atoi(jd) : 3));
  const int W = J.W;
  // ---- the records' preparation (upload + k_hp_*) has been running since pbsim_job_add_record: collect it
  HIP_OK(hipStreamSynchronize(c->prefetch_stream));
  const size_t n = c->job_records.size();
  std::vector<DeviceFlags> fl(n);
  for (size_t i = 0; i < n; i++)
    HIP_OK(hipMemcpy(&fl[i], c->job_records[i]->flags.p, sizeof(DeviceFlags), hipMemcpyDeviceToHost));
  int64_t census[kHpSlots] = {0};
  bool any11 = c->bias.hp11_seen;
  for (size_t i = 0; i < n; i++) {
    for (int k = 0; k < kHpSlots; k++) census[k] += (int64_t)fl[i].hpfreq[k];
    any11 |= fl[i].hpfreq[11] > 0;
  }
  if (c->p.hp_del_bias != 1 && !c->census_done) {  // pbsim.cpp:677-696: the census of ALL records comes before the first read
    hp_bias_from_census(c->p.hp_del_bias, census, &c->bias);
    c->bias.hp11_seen = any11;  // the pre-pass has run get_genome_seq over every record (hpfreq[11]++, pbsim.cpp:1058)
    c->class_tables_dirty = true;
    c->census_done = true;
  }
  J.recs.resize(n);
  bool seen11 = c->bias.hp11_seen;
  int64_t max_quota = 0;
  for (size_t i = 0; i < n; i++) {
    JobRecord &jr = *c->job_records[i];
    seen11 |= fl[i].hpfreq[11] > 0;  // cumulative like the reference's record loop (Q15)
    jr.ref.seq = jr.seq.as<uint8_t>();
    jr.ref.hp = jr.hp.as<uint8_t>();
    jr.ref.len = jr.len;
    jr.ref.unit = c->job_first_unit + (int64_t)i;
    jr.ref.hp_flag = c->p.hp_del_bias == 1 && !fl[i].high_bytes;
    jr.ref.hp11 = seen11;
    Rec &R = J.recs[i];
    R.ref = jr.ref;
    R.quota = quota_of(c, jr.len);
    R.st.keep_values = W > 1;
    max_quota = std::max(max_quota, R.quota);
    if (!ensure_tables(c, jr.ref.hp11)) return PBSIM_FAILED;
  }
  c->bias.hp11_seen = seen11;
  // ---- batch size: a few rounds per record and rank, not below what keeps a walk longer than its longest read
  const int P = c->p.pass_num;
  const int regions = has_quality(c) ? 3 : 2;
  const char *jr = getenv("PBSIM_JOB_ROUNDS");  // experiment knob: rounds per record the batches are sized for
  const int rounds_per_record = std::max(1, jr ? atoi(jr) : kRoundsPerRecord);
  double target = (double)max_quota * P / ((double)rounds_per_record * W);
  target = std::max(target, std::min(kMinBatchBases, (double)max_quota * P / W));
  if (J.sink && (J.sink->on_read_text || J.sink->on_maf_text)) target = std::min(target, kSinkBatchBases);
  if (c->scratch_auto) {
    size_t free_b = 0, total_b = 0;
    HIP_OK(hipMemGetInfo(&free_b, &total_b));
    size_t held = 0;
    for (Slot &sl : c->slots) held += sl.d_scratch.bytes;
    // what batch_capacity_for() charges a read: `regions` rows of 2 * length + pad columns, 12 % slack for the per-wave rounding
    const double mean_len = std::max(1.0, c->hdr.mean_len);
    // (a read yields ~0.97 of its length in bases, and a round overshoots its share of the quota by 0.5 %: 8 % headroom)
    const double want = (target / P / mean_len) * P * ((double)regions * (2.0 * mean_len + kScratchPad) * 1.12 + 64.0) * 1.08 + (64 << 20);
    const double share = std::min(48.0 * (1LL << 30), 0.10 * (double)(free_b + held));
    const int64_t auto_b = (int64_t)std::max(256.0 * (1 << 20), std::min(want, share));
    if (auto_b > c->scratch_budget || c->scratch_budget > 2 * auto_b) c->scratch_budget = auto_b;
  }
  if (W > 1) {  // every rank must size its rounds alike: the smallest pool decides
    int64_t b = c->scratch_budget;
    if (!comm->all_reduce_i64(comm->user, &b, 1, PBSIM_OP_MIN)) return fail("pbsim_comm.all_reduce_i64 failed");
    c->scratch_budget = b;
  }
  J.mean = 0.97 * c->hdr.mean_len;  // bases a read yields (deletions outweigh insertions in most models); measured from round 1 on
  for (size_t i = 0; i < n; i++) {
    Rec &R = J.recs[i];
    const double m = std::min<double>(c->hdr.mean_len, (double)R.ref.len);
    R.cap = std::max<int64_t>(1, std::min<int64_t>(batch_capacity_for(c, R.ref.len), (int64_t)(1.08 * target / P / m) + 64));
  }
  J.mean = std::min<double>(J.mean, (double)c->job_records[0]->len);
  for (Slot &sl : c->slots) sl.b_enqueued = sl.b_walked = sl.b_finalized = false;
  const int ok = J.run();
  if (!ok) {
    const std::string keep = g_err;
    J.drop_everything();
    g_err = keep;
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
