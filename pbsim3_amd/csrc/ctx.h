// ctx.h -- the context behind the C ABI (pbsim_ctx), shared by engine.cpp (batch primitives, per-unit drivers) and
// job.cpp (the job-level pipeline over records and ranks).  Internal: nothing here is part of include/pbsim3_amd.h.
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pbsim3_amd.h"
#include "host_tables.h"
#include "knobs.h"
#include "kernels.h"

namespace pbsim {
extern thread_local std::string g_err;  // pbsim_last_error() of the calling thread
int fail(const std::string &m);         // sets g_err, returns PBSIM_FAILED
}  // namespace pbsim


#define NEED_DEVICE(c)                                                                      \
  do {                                                                                      \
    if ((c)->device < 0 || !(c)->stream)                                                    \
      return fail("this context has no HIP device: the gfx950 product path has no CPU fallback"); \
  } while (0)

#define HIP_OK(expr)                                                                        \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(std::string("HIP error: ") + hipGetErrorString(e_) + " at " #expr);       \
  } while (0)

struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  hipError_t ensure(size_t n, bool exact = false) {
    if (n <= bytes) return hipSuccess;
    release();
    // (hipFree waits for the whole device: a buffer that creeps up in small steps -- the text of one truncated tail read
    // after the other -- would stall every kernel in flight each time, so small buffers start at a size they never outgrow)
    size_t want = exact ? n : std::max<size_t>(n + n / 8 + 256, 64u << 10);
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      p = nullptr;
      return e;
    }
    bytes = want;
    return hipSuccess;
  }
  template <class T>
  T *as() const {
    return reinterpret_cast<T *>(p);
  }
};

struct HostBuf {  // pinned staging
  void *p = nullptr;
  size_t bytes = 0;
  ~HostBuf() {
    if (p) (void)hipHostFree(p);
  }
  hipError_t ensure(size_t n) {
    if (n <= bytes) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    bytes = 0;
    size_t want = std::max<size_t>(n + n / 8 + 4096, 1u << 20);
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e != hipSuccess) {
      p = nullptr;
      return e;
    }
    bytes = want;
    return hipSuccess;
  }
};

using namespace pbsim;

constexpr int kMaxSlots = 6;

// Everything one in-flight batch owns.  Several slots (each with its own stream)
// let the walk of batch k+1 fill the GPU while the longest reads of batch k
// are still draining and while batch k's text is being emitted.
// k_text_rows prefetches whole 256-column tiles without bounds checks: up to two tiles (2 x 64 dwords x 256 B) past the
// last wave's last row
constexpr size_t kScratchSlack = 64u << 10;
constexpr double kSinkBatchBases = 5.0e9;  // pbsim_simulate_wgs: expected bases per batch when the text goes to a sink

// One deflate pipeline: staging for one piece of DF_PIECE_CHUNKS chunks, double-buffered dense output + pinned copies, its
// own streams.  A slot owns two, so that its read text and its MAF text can be compressed, copied and handed to their
// sinks side by side (pbsim_set_deflate bit 2).
constexpr int kDfBuffers = 5;  // at most: the lane's kernels run `ahead` pieces in front of the copies (ahead + 1 buffers in use)
struct DfLane {
  DevBuf d_df_status, d_df_ctl;        // look-back words of the launch in progress (told apart by `epoch`), ticket + total
  uint32_t epoch = 0;                  // one per launch on this lane
  DevBuf d_df_dense[kDfBuffers];       // compressed pieces: one being written, two waiting for / on the link (not in direct mode:
                                       // the members then go straight into h_df_out, page-locked host memory)
  DevBuf d_df_code;                    // the current call's code table (DF_TABLE_BYTES) + its histogram scratch
  HostBuf h_df_total, h_df_out[kDfBuffers];
  // kernels of this lane | D2H of compressed pieces: the streams of pbsim_ctx::df_streams (shared by the slots' lanes of the
  // same index -- one delivery runs at a time, and hardware queues are few: with a pair of streams per slot and lane, which
  // lanes had their kernels and their copies in ONE queue, one behind the other, was a matter of creation order)
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;
  bool own_streams = false;            // use `own` instead of the shared pair (job.cpp: the lanes of the tail chains' slots)
  hipStream_t own[2] = {nullptr, nullptr};
  // a call's first pieces launched ahead of the call (deflate_host.cpp df_begin / pbsim::deflate_prelaunch): for this text
  bool pre_valid = false, pre_own_staging = false, pre_elsewhere = false;
  const uint8_t *pre_text = nullptr;
  int64_t pre_n = 0;
  int pre_count = 0;                   // pieces df_begin launched
  hipStream_t lane_stream = nullptr;   // the lane's own kernel stream (`stream` is where the call's kernels are being launched)
  hipEvent_t ev_pre = nullptr;         // behind a prelaunch on another stream
  hipEvent_t ev_df[kDfBuffers] = {}, ev_cp[kDfBuffers] = {};
  hipEvent_t ev_k0[kDfBuffers] = {}, ev_k1[kDfBuffers] = {};  // timing: around k_deflate_chunks of the piece in dense buffer b
  // Pinned arena that holds ALL compressed pieces of one batch (job.cpp, several ranks): a rank learns where its bytes go in
  // the record's stream only after every rank has compressed its block, so the pieces wait here, D2H-copied straight in.
  std::vector<std::unique_ptr<HostBuf>> arena_blocks;
  std::vector<size_t> arena_fill;      // bytes handed out of each block since the last reset
  std::vector<char> arena_touched;     // block was used since the last trim (pbsim_job_run trims at its end)
  std::vector<int> arena_idle;         // jobs in a row that did not use the block
  std::vector<std::pair<const char *, int64_t>> arena_segs;
  size_t arena_bytes() const {
    size_t n = 0;
    for (const auto &b : arena_blocks) n += b->bytes;
    return n;
  }
  void arena_reset() {
    std::fill(arena_fill.begin(), arena_fill.end(), 0);
    arena_segs.clear();
  }
  // First fit over the blocks (a block too small for one piece still takes the next smaller one); a new block only when no
  // block has room, and never beyond `max_bytes` of pinned memory for this lane (0: no bound).
  char *arena_reserve(int64_t n, size_t max_bytes) {
    // (PBSIM_PINNED_BLOCK_KB: test hook -- small blocks, so that a test's few hundred KB per round spread over several)
    static const size_t kBlock = getenv("PBSIM_PINNED_BLOCK_KB") && atoll(getenv("PBSIM_PINNED_BLOCK_KB")) > 0
                                     ? (size_t)atoll(getenv("PBSIM_PINNED_BLOCK_KB")) << 10 : (size_t)256u << 20;
    const size_t need = ((size_t)n + 63) & ~(size_t)63;
    for (size_t i = 0; i < arena_blocks.size(); i++) {
      HostBuf &b = *arena_blocks[i];
      if (arena_fill[i] + need <= b.bytes) {
        char *p = (char *)b.p + arena_fill[i];
        arena_fill[i] += need;
        arena_touched[i] = 1;
        arena_segs.emplace_back(p, n);
        return p;
      }
    }
    const size_t want = std::max(kBlock, need);
    if (max_bytes && arena_bytes() + want > max_bytes) return nullptr;
    if (getenv("PBSIM_TRACE")) fprintf(stderr, "[pbsim arena] new block of %zu MB (lane holds %zu MB in %zu blocks)\n", want >> 20, arena_bytes() >> 20, arena_blocks.size());
    arena_blocks.emplace_back(new HostBuf);
    arena_fill.push_back(0);
    arena_touched.push_back(1);
    arena_idle.push_back(0);
    if (arena_blocks.back()->ensure(want) != hipSuccess) {
      arena_blocks.pop_back();
      arena_fill.pop_back();
      arena_touched.pop_back();
      arena_idle.pop_back();
      return nullptr;
    }
    arena_fill.back() = need;
    arena_segs.emplace_back((char *)arena_blocks.back()->p, n);
    return (char *)arena_blocks.back()->p;
  }
  // end of a job: blocks that no round of the last kArenaIdleJobs jobs touched go back to the host (a context that once ran a
  // large job does not keep its page-locked memory for ever).  Not after ONE idle job (rounds 2-5 did that): which slots -- and
  // how many blocks of a lane -- a job uses varies from job to job (its round count against the slot rotation; a rank that
  // compresses a few MB more than the one before it), and a 256 MB block costs tens of milliseconds to give back and as much to
  // page-lock again: a context that runs job after job (bench.py, the replay of one rank after the other) paid that again and
  // again -- stalls of 25-95 ms in a job's first rounds and behind its last collective (profiles/r05z_replay_host_noise.txt).
#ifndef PBSIM_ARENA_IDLE_JOBS
#define PBSIM_ARENA_IDLE_JOBS 16   // (-DPBSIM_ARENA_IDLE_JOBS=1: the behaviour of rounds 2-5, for an A/B)
#endif
  static constexpr int kArenaIdleJobs = PBSIM_ARENA_IDLE_JOBS;
  void arena_release() {
    arena_blocks.clear();
    arena_fill.clear();
    arena_touched.clear();
    arena_idle.clear();
    arena_segs.clear();
  }
  void arena_trim() {
    size_t k = 0;
    for (size_t i = 0; i < arena_blocks.size(); i++) arena_idle[i] = arena_touched[i] ? 0 : arena_idle[i] + 1;
    if (getenv("PBSIM_TRACE")) {
      size_t drop = 0;
      for (size_t i = 0; i < arena_blocks.size(); i++) drop += arena_idle[i] >= kArenaIdleJobs ? 1 : 0;
      if (drop) fprintf(stderr, "[pbsim arena] trim: %zu of %zu blocks go back\n", drop, arena_blocks.size());
    }
    for (size_t i = 0; i < arena_blocks.size(); i++)
      if (arena_idle[i] < kArenaIdleJobs) {
        if (k != i) {
          arena_blocks[k] = std::move(arena_blocks[i]);
          arena_idle[k] = arena_idle[i];
        }
        k++;
      }
    arena_blocks.resize(k);
    arena_idle.resize(k);
    arena_fill.assign(k, 0);
    arena_touched.assign(k, 0);
    arena_segs.clear();
  }
};

// The reference a batch reads: a FASTA record (wgs) or the concatenated unit set (trans / templ), prepared in HBM.
// A slot keeps the descriptor of the batch it holds, so batches of different records can be in flight side by side.
struct RefDesc {
  const uint8_t *seq = nullptr;  // upper-cased bases (bit 7 = hp == 11 when hp_flag)
  const uint8_t *hp = nullptr;   // homopolymer length per base
  int64_t len = 0;
  int64_t unit = 0;              // genome.num (wgs, 1-based) or 0
  bool hp_flag = false;
  bool hp11 = false;             // a base with hp == 11 has been counted up to and including this record (Q15)
};

// Statistics of one unit (pbsim.cpp:63-70 `sim.res_*`, :195-196 the two histograms), accumulated per finished task in read
// order.  `accuracy_total` (pbsim.cpp:4003) is an order-dependent double sum: when the tasks of a unit are spread over several
// ranks each rank keeps the per-task values of its blocks (keep_values) and the merge folds them in read order on every
// rank, so the merged mean is bit-identical to one GPU's (stats.cpp: stats_merge).
struct StatsAcc {
  int64_t res_num = 0, res_len_total = 0, res_len_min = LONG_MAX, res_len_max = 0;
  int64_t res_sub = 0, res_ins = 0, res_del = 0;
  double accuracy_total = 0.0;
  std::vector<int64_t> freq_len, freq_acc;  // [2*len_max+2], [100001]; sized on first use
  bool keep_values = false;
  struct Block {
    int64_t first_task;  // global 0-based task index ((read-1)*pass_num + pass) of values[0]
    std::vector<double> values;
  };
  std::vector<Block> blocks;
  void reset() {
    res_num = res_len_total = res_len_max = 0;
    res_len_min = LONG_MAX;
    res_sub = res_ins = res_del = 0;
    accuracy_total = 0.0;
    std::fill(freq_len.begin(), freq_len.end(), 0);
    std::fill(freq_acc.begin(), freq_acc.end(), 0);
    blocks.clear();
  }
};

struct Slot {
  RefDesc ref;                         // of the batch in this slot
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
  DevBuf d_flags;
  DevBuf d_rawlen, d_len, d_off, d_acc;
  DevBuf d_hist, d_bin_start, d_bin_cursor, d_class_start;
  DevBuf d_task_of_slot, d_slot_of_task, d_wave_cap, d_wave_off, d_wg_tmp, d_wg_order;
  DevBuf d_out_len, d_maf_len, d_nsub, d_nins, d_ndel, d_qsum;
  DevBuf d_cum, d_scan_tmp, d_rt_len, d_mt_len, d_row_dst;
  DevBuf d_chain, d_chain_mask;        // a chain of truncated reads: ChainState; the step's view of task_of_slot
  DevBuf d_scratch, d_read_text, d_maf_text;
  HostBuf h_read_text, h_maf_text, h_stats, h_flags;
  DfLane df[2];                        // deflate staging: [0] read text (or any single stream), [1] MAF text beside it
  hipStream_t walk_stream = nullptr;   // low priority: the walk kernel only
  hipEvent_t ev_sq_walk = nullptr, ev_sq_done = nullptr;  // sampling method: k_sample_qsum of this slot's chunk (ctx.sq_stream)
  bool sq_pending = false;
  hipEvent_t ev_prep = nullptr;        // header + sort done (walk_stream waits for it)
  hipStream_t coop_stream = nullptr;   // the long reads' walk (k_walk_errhmm_coop), beside the batch's lane walk
  hipEvent_t ev_coop = nullptr;
  bool stats_fetched = false;          // the per-task counters of the final reads are already in h_stats (finalize_text, deferred mode)
  hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;  // timing: around the text emission kernels (launch_text_emit)
  bool text_timed = false;             // ev_t0 / ev_t1 hold an emission that has not been added to the profile yet
  int64_t text_in = 0, text_out = 0;   // its bytes: scratch rows read, text written
  hipEvent_t ev_text = nullptr;        // the batch's text has been emitted (finalize_text; waited for by whoever reads the text
                                       // on another stream or thread when pbsim_ctx::defer_text_sync)
  int64_t b_first = 0, b_n = 0, b_slots_max = 0;
  bool b_truncated = false, b_enqueued = false, b_walked = false, b_finalized = false;
  bool b_chain = false;                // the batch is a chain of truncated reads (walk_begin(.., chain))
  int64_t b_trunc = -1;                // truncate_remaining of the batch (a re-walk begins it again)
  double b_factor = 2.0;               // scratch factor of the batch's layout
  int64_t b_pass0 = 0;
  int64_t b_max_raw = 0;               // largest raw length a read of the batch drew (wgs; DeviceFlags::max_rawlen)
  pbsim_batch_info b_info;
};

// One record of a job (pbsim_job_add_record): resident in HBM for the whole job
struct JobRecord {
  DevBuf seq, hp, tiles, flags;
  int64_t len = 0;
  RefDesc ref;
  hipEvent_t ready = nullptr;  // behind the record's upload and preparation on the prefetch stream (pbsim_job_add_record*)
  DeviceFlags *h_flags = nullptr;  // pinned: what the preparation found (hp census, bytes >= 0x80), copied down in front of `ready`
  ~JobRecord() {
    if (ready) (void)hipEventDestroy(ready);
    if (h_flags) (void)hipHostFree(h_flags);
  }
};

struct pbsim_ctx {
  pbsim_params p;
  int device = 0;
  hipStream_t stream = nullptr;  // reference preparation and table uploads
  Slot slots[kMaxSlots];
  int cur = 0;
  Slot &s() { return slots[cur]; }

  std::unique_ptr<ErrModel> err;
  std::unique_ptr<QsModel> qs;
  HeaderTables hdr;
  HpBias bias;
  ErrClassTables ect;
  QsClassTables qct;
  bool class_tables_dirty = true;
  int coop_wg_errhmm[2] = {0, 0};  // persistent workgroups of k_walk_errhmm_coop<hp flag bits> (from its occupancy; 0: not asked yet)
  bool header_uploaded = false;

  DevBuf d_prob2len, d_prob2acc, d_cls;
  DevBuf d_qs_tabs_v[2];         // set_mut thresholds + qprob; [1]: the variant once an hp == 11 base has been counted (Q15)
  bool qs_tabs_ready[2] = {false, false};
  // reference
  DevBuf d_seq_own, d_hp, d_tiles, d_ref_flags;
  // the NEXT record, uploaded and prepared beside the current record's simulation (pbsim_prefetch_reference*)
  DevBuf d_seq_next, d_hp_next, d_tiles_next, d_ref_flags_next;
  hipStream_t prefetch_stream = nullptr;
  DevBuf d_lines, d_lines_tmp;   // pbsim_job_add_record_lines: a record's FASTA lines as uploaded; tile counts + scan scratch
  // sampling method: k_sample_qsum runs beside the chunk's text emission (own stream); its sums are due at the statistics fetch
  hipStream_t sq_stream = nullptr;
  const void *pf_src = nullptr;
  int64_t pf_len = 0;
  bool seq_hp_flag = false;  // bit 7 of the prepared sequence bytes carries hp == 11 (k_hp_final)
  const uint8_t *d_seq = nullptr;
  int64_t ref_len = 0;
  int64_t unit = 0;
  int64_t census[kHpSlots] = {0};
  bool census_done = false;
  bool census_from_job = false;  // census_done was set by pbsim_job_run from the job's own records (recomputed per run), not by pbsim_finish_hp_census
  bool hp11_explicit = false;    // an hp == 11 base was counted by pbsim_add_hp_census (the pre-pass over all records, pbsim.cpp:677-696)
  bool hp11_before_job = false;  // Q15 state in front of the job's first record (pbsim_job_begin): every pbsim_job_run starts from it
  // trans units (pbsim_set_transcripts)
  int64_t n_units = 0, trans_reads = 0;
  DevBuf d_read_unit, d_read_minus, d_read_base, d_unit_len, d_unit_rank, d_unit_names, d_off_table, d_ssp, d_ssp_rv;
  // sampling method (pbsim_set_sample_profile): filtered quality strings, padded to 8 bytes each
  DevBuf d_sq, d_sq_line_len, d_sq_line_qoff, d_sq_vbase;
  std::vector<int32_t> sq_len;
  std::vector<int64_t> sq_off;
  int64_t sq_total = 0;        // sample.len_total_filtered
  int64_t scratch_budget = 0;  // bytes of wave scratch per slot
  // Columns a task's rows are laid out for = scratch_factor x its length + pad.  The reference's own buffers take 2 x (the
  // bound that cannot overflow: every column consumes a reference base or is one of at most... in practice ~1.06-1.2 x); a
  // context starts at 2, keeps the largest need its walks have reported (need_seen, DeviceFlags::need_q10) and lays the next
  // batches out with need_seen + 0.08.  A batch in which a read runs out of row all the same is walked again at 2 by
  // pbsim_batch_walk_end, transparently.  PBSIM_SCRATCH_FACTOR fixes the factor (tests: 1.02 makes most batches walk twice).
  double scratch_factor = 2.0, need_seen = 0.0;
  bool scratch_factor_fixed = false;
  int64_t rewalked_batches = 0;  // batches walked again at the full factor (pbsim_prof: how often the estimate was too low)
  bool scratch_auto = true;    // sized per record by pbsim_simulate_wgs unless PBSIM_SCRATCH_MB / pbsim_set_scratch_bytes said otherwise
  int pipeline_depth = 2;      // slots pbsim_simulate_* keeps in flight
  hipStream_t df_streams[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // [lane][kernels | copies]
  bool defer_text_sync = false;  // finalize_text returns once the sizes are known; the text emission is still in flight (job pipeline)
  bool defer_account = false;    // deliver() leaves the batch's statistics to its caller (units.cpp accounts batch k + 1 on a thread beside the delivery of batch k)
  int walk_lds_kb = 27;        // walk workgroups per CU: 27 KB -> five (batch primitives), 41 KB -> three (the job pipeline)
  bool bam_output = false;     // pass_num > 1: BAM records instead of SAM text
  int deflate = 0;             // bit 0 / 1: read / MAF sink receives BGZF-framed gzip members (deflate.hip)
  bool deflate_parallel = false;  // pbsim_set_deflate bit 2: the two sinks are served from two host threads
  DevBuf d_df_tables;          // crc slice-by-4 tables [4][256] + x^(8*128*k) [256]
  DevBuf d_df_prof;

  // per-unit statistics (pbsim.cpp:63-70, 195-196)
  StatsAcc st;                   // of the current unit (pbsim_batch_account / pbsim_get_stats)
  std::vector<std::unique_ptr<JobRecord>> job_records;  // pbsim_job_add_record (job.cpp)
  // pbsim_job_expect: the job's records are announced (their lengths) and may arrive WHILE pbsim_job_run is running, added by
  // another thread in order; the job begins a record's first round when that record is resident and prepared
  std::mutex job_mu;                    // job_records' size, job_feed_failed
  std::condition_variable job_cv;
  std::vector<int64_t> job_expect_len;  // empty: the job is what has been added when pbsim_job_run is called
  // the buffers of the records pbsim_job_begin dropped: the next records of the same length move into them instead of new
  // allocations (hipFree waits for the whole device, and HBM handed back and taken again comes back in smaller fragments: a job
  // on re-allocated records ran 3 % slower); at most what one job held, released with the pools (pbsim_release_pools)
  std::vector<std::unique_ptr<JobRecord>> job_spare;
  bool job_feed_failed = false;
  std::string job_feed_err;
  int64_t job_counters[8] = {0};
  int64_t job_progress[8] = {0}; // pbsim_job_progress: the exchange the round loop is about to enter
  double job_breakdown[16] = {0};  // pbsim_job_breakdown: where the round loop's wall time went (job.cpp)
  int job_interleave = 1;        // pbsim_job_set_interleave: records whose rounds alternate
  int64_t job_first_unit = 1;    // genome.num of the job's first record (pbsim_job_begin)

  // profiling
  double prof_walk_ms = 0, prof_total_ms = 0, prof_tail_ms = 0;
  int64_t prof_walk_launches = 0, prof_tail_launches = 0;   // walk launches of batches / of single truncated tail reads
  int64_t prof_wave_launches = 0;                           // launches of a wave walker (k_walk_errhmm_coop / k_walk_qshmm_coop)
  hipEvent_t ev_prof_base = nullptr;                     // pbsim_prof_reset: time zero of the walk intervals
  std::vector<std::pair<float, float>> prof_intervals;   // [start, end] ms of every walk launch since the reset
  // secondary kernels (pbsim_prof_secondary): text emission and k_deflate_chunks, HIP events on their own streams
  double prof_text_ms = 0, prof_deflate_ms = 0;
  int64_t prof_text_launches = 0, prof_text_in = 0, prof_text_out = 0;
  int64_t prof_deflate_launches = 0, prof_deflate_in = 0, prof_deflate_out = 0;
  std::mutex prof_mu;            // the deflate lanes run on delivery threads
};

// ---- internals shared by engine.cpp and job.cpp ---------------------------------------------------------------------
namespace pbsim {
RefDesc current_ref(const pbsim_ctx *c);
int64_t quota_of(const pbsim_ctx *c, int64_t ref_len);  // (long long)(depth * len), pbsim.cpp:705
// chain: the n_reads reads are the truncated reads behind a quota cut, each depending on the one before (pbsim.cpp:3792-3800);
// truncate_remaining = what is left of the quota in front of the first; ended by chain_end_finalize
int walk_begin(pbsim_ctx *c, const RefDesc &ref, int64_t first_read, int64_t n_reads, int64_t truncate_remaining, bool chain = false,
               double factor = 0.0 /* 0: the context's current scratch factor */);
double scratch_factor_of(const pbsim_ctx *c);  // what the next batches are laid out with
int chain_end_finalize(pbsim_ctx *c, int64_t len_total_before, pbsim_batch_info *out);
int chain_reads_for(const pbsim_ctx *c, int64_t ref_len, int64_t remaining);
constexpr int kChainReads = 6;  // steps enqueued per chain (a chain is 2-3 reads as a rule: each leaves ~3 % of its length)
int64_t batch_capacity_for(const pbsim_ctx *c, int64_t ref_len);  // reads one batch is sized to (scratch budget)
// the two halves of pbsim_batch_finalize on the selected slot: the quota cut, then text sizes + scans + text emission
int finalize_cut(pbsim_ctx *c, int64_t len_total_before, pbsim_batch_info *out);
int finalize_text(pbsim_ctx *c, pbsim_batch_info *info);
int finalize_uncut(pbsim_ctx *c, pbsim_batch_info *out);
std::string sam_header_text(const pbsim_ctx *c, int64_t unit);
// pbsim.cpp:3986-4005 / 2293-2316 for the n_final reads of the selected slot's finalized batch, into `st`
int account_slot(pbsim_ctx *c, StatsAcc *st);
int account_of(pbsim_ctx *c, Slot &sl, StatsAcc *st);  // the same on an explicit slot (no use of the selected-slot cursor)
// one finished task (pbsim.cpp:3986-4005): lengths, error counts, the accuracy value and its histogram bin
void stats_add_task(StatsAcc *st, int64_t len_max, bool quality, long len, long nsub, long nins, long ndel, double qsum,
                    std::vector<double> *values);
// pbsim.cpp:4082-4105, 5541-5562: means, SDs, rates
void stats_finish(const StatsAcc &st, const pbsim_params &p, int64_t ref_len, pbsim_stats *o);
// sums the accumulators of all ranks into every rank's `st` (counters, min/max, histograms, accuracy_total in read order);
// extra[0..n_extra) are summed along (byte totals of a record's streams)
int stats_merge(StatsAcc *st, const pbsim_params &p, const pbsim_comm *comm, int64_t *extra, int n_extra);
// d_text[0..n) (device) -> BGZF-framed gzip members, handed to `consume` piece by piece from pinned staging (deflate.hip)
// `place` (optional): pinned host memory for a piece of the given size instead of the lane's staging (a batch-wide arena)
// the first pieces of the selected... of slot `sl`'s two deflate calls (read text, MAF text) launched NOW, behind the slot's text
// emission: the delivery that follows (deflate_pieces on the same buffers) finds them under way.  `staged`: the pieces will go
// through the lanes' own staging (one rank) rather than a caller's arena.  Harmless when the delivery never comes.
int deflate_prelaunch(pbsim_ctx *c, Slot &sl, bool want_read, bool want_maf, bool staged);
int deflate_pieces(pbsim_ctx *c, DfLane &lane, const uint8_t *d_text, int64_t n,
                   const std::function<int(const char *, int64_t)> &consume,
                   const std::function<char *(int64_t)> *place = nullptr);
int ensure_deflate_ready(pbsim_ctx *c);  // CRC / shift tables of deflate.hip resident (call before using lanes from threads)
// upper-case + homopolymer pass of a record (k_hp_*) enqueued on `stream`; flags receives the census (DeviceFlags)
int prepare_enqueue(pbsim_ctx *c, uint8_t *d_seq, DevBuf &hp, DevBuf &tiles, DevBuf &flags, int64_t len, hipStream_t stream);
int ensure_tables(pbsim_ctx *c, bool hp11);  // header + class tables (+ the set_mut variant) resident
inline bool has_quality(const pbsim_ctx *c) { return c->p.method == PBSIM_METHOD_QS || c->p.method == PBSIM_METHOD_SAMPLE; }
}  // namespace pbsim
