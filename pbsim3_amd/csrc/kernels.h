// kernels.h -- launch interface of the gfx950 kernels (kernels.hip).
// Everything here is plain pointers + sizes; all pointers are device pointers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "modes.h"

namespace pbsim {

constexpr int kWG = 256;            // threads per workgroup of the walk kernels (4 waves)
constexpr int kLenBuckets = 4096;   // length buckets per accuracy class in the task sort
#ifndef PBSIM_BIN_PAD
#define PBSIM_BIN_PAD 16
#endif
constexpr int kBinPad = PBSIM_BIN_PAD;  // int32 slots per counter of the sort's histogram / cursors: one counter per 64-byte line,
                                      // neighbouring (hot) length buckets do not share a line that 8 XCDs fight over
constexpr int kLenShift = 8;        // bucket = len >> 8  (len_max 1e6 -> 3907 buckets)
constexpr int kScratchPad = 64;     // per-task slack: a row holds factor * L + kScratchPad columns (factor <= 2: pbsim.cpp:5488 uses 2*len_max+1)
constexpr int kMaxClasses = 64;
// wave_cap[w] bit 30: the 64 tasks of this scratch block are all walked by k_walk_errhmm_coop and its rows are stored task
// by task (row r of task l at r * cap * 256 + l * cap * 4 bytes) instead of interleaved dword by dword: a wave that walks
// ONE read writes 64 consecutive bytes per row and step there, not sixteen dwords 256 bytes apart
constexpr int32_t kWaveTransposed = 1 << 30;
constexpr int kHpTile = 4096;       // bases per workgroup in the homopolymer kernels

// error flag bits written by kernels into EngineFlags::error
enum : uint32_t {
  kErrScratchOverflow = 1u,   // a task produced more MAF columns than its rows hold
  kErrScratchBudget = 2u,     // the batch needs more scratch than the pool holds
};

struct DeviceFlags {
  uint32_t error;
  uint32_t high_bytes;    // k_hp_breaks: the unit holds bytes >= 0x80 (no room for the hp == 11 flag in bit 7)
  int64_t n_final;        // quota cut (reads)
  int64_t total_slots;    // slots after class alignment
  int64_t scratch_need;   // bytes the batch needs
  int64_t sums[8];        // [0] pass-0 bases of the batch, [1] read-text bytes, [2] maf-text bytes,
                          // [3] bases all passes (final reads), [4] ref bases, [5] maf columns
  unsigned long long hpfreq[12];
  uint32_t need_q10;      // walks: max over the tasks of (MAF columns - pad) / length, in 1/1024 (note_row_need)
  uint32_t max_rawlen;    // k_header_wgs: the largest raw (un-truncated) length a read of the batch drew (job.cpp: rounds that cannot touch the quota)
};

struct RefView {
  const uint8_t *seq;  // upper-cased, padded to a multiple of 16 bytes; bit 7 = hp == 11 when the unit was prepared with flag_hp11
  const uint8_t *hp;   // homopolymer length per base, 1..11
  int64_t len;
};


struct HeaderArgs {
  uint32_t seed, unit;
  int64_t first_read;       // 1-based index of read 0 of the batch
  int64_t n_reads;
  const int32_t *prob2len;  // [0..len_rv]
  int64_t len_rv;
  const uint8_t *prob2acc;  // [0..acc_rv]
  int64_t acc_rv;
  int64_t ref_len;
  int64_t len_min;
  int64_t truncate_remaining;  // <0: none
  int32_t *rawlen, *len, *off;
  uint8_t *acc;
  uint32_t *max_rawlen;       // wgs: atomicMax of the batch's raw lengths (DeviceFlags::max_rawlen; may be null)
  // trans (pbsim.cpp:4488-4504): per-read unit, per-unit length / rank / the 21 start offsets
  int32_t is_templ;           // templ: one full-length '+' read per unit, accuracy draw only (pbsim.cpp:5092-5097)
  const int32_t *read_unit;   // [n_reads] (already offset to the batch)
  const int64_t *unit_len;    // [n_units]
  const int32_t *unit_rank;   // [n_units] ceil(len/1000)
  const int32_t *off_table;   // [n_units][21] int(len*((5k-2.5)/100)+0.5), k=0 -> 0
  const uint8_t *ssp;         // [rank_max+1][1000] start bucket k (prob2ssp/5)
  const int32_t *ssp_rv;      // [rank_max+1]
};

// The truncated reads behind a record's quota cut (pbsim.cpp:3792-3800) form a chain: read k's length is what the reads in
// front of it left of the quota.  The chain runs on the device as steps enqueued back to back (engine.cpp walk_chain_begin):
// step k re-draws read k's header with the remaining quota (k_chain_prepare), walks read k alone (the walk kernels see a copy
// of task_of_slot that holds read k's tasks only), and subtracts its pass-0 bases (k_chain_update).  No host round trip
// between the reads; the host reads this block once when the steps are through.
struct ChainState {
  int64_t remaining;   // quota - len_total in front of the next read (<= 0: the quota is reached)
  int64_t total;       // pass-0 bases of the reads made so far
  int32_t made;        // reads made (they are final)
  int32_t done;        // remaining <= 0
};

struct SortArgs {
  int64_t n_reads;
  int32_t pass_num;
  int32_t acc_lo, ncls;
  const int32_t *len;
  const uint8_t *acc;
  int32_t *hist;        // [ncls*kLenBuckets*kBinPad]
  int32_t *bin_start;   // [ncls*kLenBuckets]
  int32_t *bin_cursor;  // [ncls*kLenBuckets*kBinPad]
  int32_t *class_start; // [ncls+1], multiples of kWG
  int32_t coop_bucket;  // tasks of length bucket >= coop_bucket are walked by k_walk_errhmm_coop (kLenBuckets: none)
  int32_t *coop_end;    // [ncls] end of those tasks' slots: they are the first of their class (longest first)
  uint64_t coop_classes; // bit c: class c takes part (not a verbatim class)
  int32_t *task_of_slot;
  int32_t *slot_of_task;
  int32_t *wave_cap;    // [n_waves_max] dwords per lane per region
  int64_t *wave_off;    // [n_waves_max] byte offset into the scratch pool
  int64_t n_slots_max;
  int32_t *wg_hist, *wg_start;  // [kLenBuckets+1] each
  int32_t *wg_order;            // [n_slots_max/kWG] walk workgroups, longest reads first
  int32_t regions;      // scratch rows per task (2 errhmm, 3 qshmm)
  int32_t cap_q8;       // columns per row = cap_q8 / 256 x the block's longest read + kScratchPad (512: the reference's 2 L)
  int64_t scratch_bytes;
  DeviceFlags *flags;
};

struct WalkArgs {
  uint32_t seed, unit;
  int64_t first_read;
  int32_t pass_num;
  int32_t ncls;
  RefView ref;
  const int32_t *len, *off;
  const int64_t *read_base;  // trans: offset of the read's unit inside the concatenated reference (NULL: 0)
  const uint8_t *read_minus; // trans: strand per read (NULL: wgs parity rule)
  const uint8_t *cls_blob;   // ncls blobs of `stride` bytes
  uint32_t stride, rows_off, init_off, tran_off, emis_off, freq_off, rv_off;
  const int32_t *class_start;
  const int32_t *task_of_slot;
  const int32_t *wg_order;
  int32_t mean_len;          // E[L] of the length table (priority thresholds)
  int32_t coop_min_len;      // errhmm: reads of at least this length belong to k_walk_errhmm_coop (a multiple of 256; INT32_MAX: none)
  int32_t coop_dynamic;         // k_walk_errhmm_coop: units drawn from a counter (flags->sums[7]) instead of dealt round-robin
  int32_t cap_q8;               // SortArgs::cap_q8 of the batch's layout (the walks' priority heuristic reads lengths off the caps)
  const int32_t *coop_end;   // [ncls] their slots are class_start[c] .. coop_end[c]
  const int32_t *wave_cap;
  const int64_t *wave_off;
  uint8_t *scratch;
  int32_t *out_len, *maf_len, *nsub, *nins, *ndel;
  double *qsum;              // qshmm: sum of error probabilities per task
  // qshmm class-independent tables
  const uint32_t *sub_thre, *ins_thre, *del_thr;  // [94], [94], [94*12]
  const double *qprob;                           // [94]
  DeviceFlags *flags;
};

// sampling method: one lane per filtered quality string of the chunk, copies walked in sequence
struct SampleArgs {
  uint32_t seed, unit;
  int64_t first_read;         // read index of task 0
  int32_t n_lines;            // strings in this chunk
  int32_t n_line_waves;       // ceil(n_lines / 64)
  int32_t n_coop_waves;       // the first n_coop_waves line waves (the longest strings): one WAVE per string (hp_bits only)
  int32_t n_coop_blocks;      // workgroups 0 .. n_coop_blocks - 1 walk them, the others one lane per string
  int64_t n_coop_slots;       // 64 x their virtual waves (k_sample_qsum: one lane per slot)
  RefView ref;
  const uint8_t *quals;       // filtered quality strings, each padded to a multiple of 8 bytes
  const int64_t *line_qoff;   // [n_lines] byte offset of the string
  const int32_t *line_len;    // [n_lines]
  const int32_t *vbase;       // [n_line_waves + 1] first virtual wave (= scratch/text slot wave) of each string wave
  const int32_t *task_of_slot;
  const int32_t *wave_cap;
  const int64_t *wave_off;
  uint8_t *scratch;
  int32_t *span, *off;        // per task: reference bases consumed, start offset in the record
  int32_t *out_len, *maf_len, *nsub, *nins, *ndel;
  double *qsum;
  const uint32_t *sub_thre, *ins_thre, *del_thr;  // [94], [94], [94*12]
  const double *qprob;                           // [94]
  DeviceFlags *flags;
};

struct TextArgs {
  int64_t first_read;
  int64_t n_reads;          // reads to emit (n_final)
  int32_t pass_num;
  int32_t is_wgs;           // id format
  int32_t is_qs;            // quality row present
  uint32_t unit;            // record number (wgs)
  int64_t ref_len;
  const int32_t *len, *off;
  const int32_t *out_len, *maf_len;
  const int32_t *slot_of_task;
  const int32_t *task_of_slot;
  int64_t *row_dst;             // [n_tasks][6] byte offsets of the big rows inside the text buffers
  const int32_t *wave_cap;
  const int64_t *wave_off;
  const uint8_t *scratch;
  const int64_t *read_text_off, *maf_text_off;  // exclusive scans [n_tasks]
  int64_t *read_text_len, *maf_text_len;        // sizes [n_tasks]
  char *read_text, *maf_text;
  char id_prefix[64];
  int32_t id_prefix_len;
  char rq_text[32];         // "%f" of accuracy_mean (SAM rq:f:)
  int32_t rq_len;
  int32_t bam;              // 1: emit BAM records instead of SAM text (pass_num > 1)
  uint32_t rq_bits;         // the float32 a SAM parser reads from rq_text
  // trans: per-read unit table
  const int32_t *read_unit;     // [n_reads] transcript index of each read (NULL for wgs)
  const int64_t *unit_len;      // [n_units]
  const char *unit_names;       // [n_units][132] NUL-terminated ids
  int32_t name_pad3;            // templ: the name is the id but pads as "ref" (digit_num1[0] = 3, pbsim.cpp:5290)
  const uint8_t *read_minus;    // [n_reads] strand for trans (NULL for wgs)
};

// ---- launches (all asynchronous on `s`) ------------------------------------
void launch_prepare_reference(uint8_t *seq, uint8_t *hp, int flag_hp11, int64_t len, int64_t *tile_first, int64_t *tile_last,
                              int64_t *carry_start, int64_t *carry_next, int keep_first_case, DeviceFlags *flags,
                              hipStream_t s);
// lines[0..bytes) (a record's FASTA sequence lines, 16-byte aligned) -> dst: every byte that is not a line feed, in order
// (pbsim.cpp:1014-1033); tile_tmp: bytes/4096 + 1 int64, scan_tmp: bytes/4096/1024 + 8 int64, *total receives the bases kept
void launch_squeeze_lines(const uint8_t *lines, int64_t bytes, uint8_t *dst, int64_t *tile_tmp, int64_t *scan_tmp, int64_t *total,
                          hipStream_t s);
void launch_header_wgs(const HeaderArgs &a, hipStream_t s);
void launch_header_trans(const HeaderArgs &a, hipStream_t s);
void launch_task_sort(const SortArgs &a, hipStream_t s);
// chain step k, before its walk: header of read k with the chain's remaining quota (a.truncate_remaining is ignored; the
// other reads' entries stay as they are), masked[slot] = task_of_slot[slot] for read k's tasks while the chain is running,
// -1 everywhere else; after its walk: k_chain_update
void launch_chain_prepare(const HeaderArgs &a, int k, int32_t pass_num, const int32_t *task_of_slot, int32_t *masked,
                          int64_t n_slots_max, ChainState *chain, hipStream_t s);
void launch_chain_update(int k, int32_t pass_num, const int32_t *out_len, ChainState *chain, hipStream_t s);
void launch_chain_init(ChainState *chain, int64_t remaining, hipStream_t s);
// min_lds_kb: LDS to ask for at least, i.e. a cap on the walk's workgroups per CU (kernels.hip walk_lds)
void launch_walk_errhmm(const WalkArgs &a, int64_t n_slots_max, uint32_t lds_bytes, bool fast_rv, bool hp_bits,
                        hipStream_t s, int min_lds_kb);
// the long reads of the batch (WalkArgs::coop_min_len), one wave per read; `n_wg` persistent workgroups; `lds_bytes` = the
// lane walk's (class blob + byte tables), the waves' own areas are added here.  Classes of at most kCoopMaxStates states.
constexpr int kCoopMaxStates = 31;
// Waves (= reads in flight) per workgroup of the wave walker; they share one class blob in LDS (24-31 KB).  Eight since the
// end of round 4 (four before; eight had been tried in round 3 for the job in HBM: no gain, profiles/r03_occupancy_ab.txt):
// the same 4096 waves in half the LDS leave the delivered job's compression its workgroups (a rank of eight 167 -> 164 ms),
// and with the kernel held to 80 registers three such workgroups -- six waves per SIMD -- fit a CU where five four-wave
// ones did (wave-only batches: ERRHMM-ONT 152 -> 156, ERRHMM-ONT-HQ 127 -> 142 G columns/s).  profiles/r04_coop_split_ab.txt
constexpr int kCoopWaves = 8;
void launch_walk_errhmm_coop(const WalkArgs &a, int n_wg, uint32_t lds_bytes, bool hp_bits, hipStream_t s);
int walk_errhmm_coop_resident(uint32_t lds_bytes, bool hp_bits);
// the QSHMM wave walker (classes with a model whose moduli are all 100, hp flag in the sequence bytes, <= 63 states);
// `lds_bytes` = class blob + byte tables, the quality rows and the waves' areas are added here
constexpr int kQCoopMaxStates = 63;
void launch_walk_qshmm_coop(const WalkArgs &a, int n_wg, uint32_t lds_bytes, hipStream_t s);
void launch_qshmm_coop_qsum(const WalkArgs &a, int64_t n_slots_max, hipStream_t s);  // behind it, on the same stream
void launch_walk_qshmm(const WalkArgs &a, int64_t n_slots_max, uint32_t lds_bytes, bool fast_rv, bool hp_bits,
                       hipStream_t s, int min_lds_kb);
void launch_walk_sample(const SampleArgs &a, bool hp_bits, hipStream_t s);
void launch_sample_qsum(const SampleArgs &a, hipStream_t s);  // the ordered quality sums of the wave-walked reads
// exclusive scan of int64 (in-place allowed: out may equal in); tmp needs (n/1024+2) int64
void launch_exclusive_scan_i64(const int64_t *in, int64_t *out, int64_t n, int64_t *tmp, int64_t *total, hipStream_t s);
// cum[r] = pass-0 output bases of reads < r (exclusive scan), *total = their sum
void launch_gather_pass0_scan(const int32_t *out_len, int64_t n_reads, int32_t pass_num, int64_t *cum, int64_t *tmp,
                              int64_t *total, hipStream_t s);
// flags->n_final = first read r with  before+cum[r] >= quota  or  before+cum[r]+rawlen[r] > quota (else n_reads)
void launch_quota_cut(const int64_t *cum, const int32_t *rawlen, int64_t n_reads, int64_t len_total_before,
                      int64_t quota, int force_all, DeviceFlags *flags, hipStream_t s);
void launch_text_sizes(const TextArgs &a, DeviceFlags *flags, hipStream_t s);
void launch_text_emit(const TextArgs &a, int64_t n_slots_max, const DeviceFlags *flags, hipStream_t s);

// ---- deflate.hip: text -> BGZF-framed gzip members, one per DF_CHUNK input bytes
#define DF_CHUNK 32768
#define DF_SLOT (DF_CHUNK + 64)          // per-chunk staging stride (a stored member is DF_CHUNK + 31 bytes)
#define DF_PIECE_CHUNKS 8192             // chunks per launch (a piece: 256 MiB of text)
#define DF_SAMPLE_CHUNKS 64              // chunks of a call's text the code table is fitted to (2 MiB)
#define DF_TABLE_HDR_DW 96               // dwords of precomputed block header (prefix + 287 code lengths, run-length coded)
#define DF_TABLE_BYTES 2048              // device bytes of one code table
void deflate_host_tables(uint32_t *crc_table /*4 x 256, slice-by-4*/, uint32_t *pow128 /*256*/);
// the code table of a call: token histogram of the first DF_SAMPLE_CHUNKS chunks of text[0..n_bytes) (hist: 288 u32 of
// scratch), every symbol floored at one occurrence, length-limited Huffman, canonical codes, block header -> table
void launch_deflate_table(const uint8_t *text, int64_t n_bytes, uint32_t *hist, void *table, hipStream_t s);
// text (16-byte aligned, 16 bytes of readable slack) -> the members of its nch = ceil(n_bytes / DF_CHUNK) <= DF_PIECE_CHUNKS
// chunks, packed densely from dense[0] in chunk order (decoupled look-back: a member is written once, at its final place);
// dense: device memory or page-locked host memory mapped into the device (nch * DF_SLOT bytes worst case, 16 bytes of slack);
// status: DF_PIECE_CHUNKS u64 look-back words (never cleared: `epoch` must differ from launch to launch on the same array);
// ctl: DF_CTL_BYTES, ctl[8..16) receives the total as int64; table from launch_deflate_table
#define DF_CTL_BYTES 16
void launch_deflate(const uint8_t *text, int64_t n_bytes, uint64_t *status, void *ctl, uint32_t epoch, uint8_t *dense,
                    const uint32_t *d_crc_table, const uint32_t *d_pow128, const void *table, hipStream_t s,
                    unsigned long long *d_prof = nullptr /* PBSIM_DEFLATE_PROF: per-phase tick sums */,
                    hipEvent_t ev_begin = nullptr, hipEvent_t ev_chunks_done = nullptr /* recorded around k_deflate_chunks */);

}  // namespace pbsim
