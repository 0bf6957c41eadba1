/* pbsim3_amd.h -- C ABI of the MI355X-native pbsim3 hot path.
 *
 * The reference (yukiteruono/pbsim3 v3.0.5, one translation unit) has no
 * plugin or FFI seam: its boundary is the process (argv -> files).  The
 * internal seam this library replaces is the family
 *     int simulate_by_errhmm(void)        src/pbsim.cpp:3594
 *     int simulate_by_qshmm(void)         src/pbsim.cpp:1955
 *     int simulate_by_errhmm_trans(void)  src/pbsim.cpp:4114
 * together with the loaders/table builders they consume through globals
 * (set_errhmm :5640, set_qshmm :5570, set_mut :5471, get_genome_seq :997).
 * Every entry point below cites the reference lines it stands in for.
 *
 * Conventions (mirroring the reference): functions return PBSIM_SUCCEEDED (1)
 * or PBSIM_FAILED (0) like SUCCEEDED/FAILED (pbsim.cpp:16-17); the message the
 * reference would have printed as "ERROR: ..." is available from
 * pbsim_last_error().  A context is single-owner (one host thread, one GPU).
 * All pointers are plain host pointers unless the name says `_device`.
 *
 * The library is the HIP product: there is no CPU fallback.  Every compute
 * entry point fails (returns 0, pbsim_last_error() says why) when no gfx950
 * device is usable.
 */
#ifndef PBSIM3_AMD_H
#define PBSIM3_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PBSIM_SUCCEEDED 1
#define PBSIM_FAILED 0

#define PBSIM_STRATEGY_WGS 1   /* pbsim.cpp:34 */
#define PBSIM_STRATEGY_TRANS 2 /* pbsim.cpp:35 */
#define PBSIM_STRATEGY_TEMPL 3 /* pbsim.cpp:36 */
#define PBSIM_METHOD_QS 1      /* pbsim.cpp:37 */
#define PBSIM_METHOD_ERR 2     /* pbsim.cpp:38 */
#define PBSIM_METHOD_SAMPLE 3  /* pbsim.cpp:39 (METHOD_SAM; the store/reuse variants :40-41 are file handling of the caller) */

/* Validated simulation parameters = the fields of `struct sim_t`
 * (pbsim.cpp:51-77) that the hot path reads.  pbsim_params_default() applies
 * set_sim_param()'s defaults (pbsim.cpp:1539-1685). */
typedef struct pbsim_params {
  int32_t strategy;     /* PBSIM_STRATEGY_*                       */
  int32_t method;       /* PBSIM_METHOD_*                         */
  uint32_t seed;        /* --seed                  (pbsim.cpp:417) */
  int32_t pass_num;     /* --pass-num              (pbsim.cpp:494) */
  double depth;         /* --depth                 (pbsim.cpp:351) */
  double accuracy_mean; /* --accuracy-mean, already int(x*100)*0.01 (pbsim.cpp:1660) */
  double len_mean;      /* --length-mean           (pbsim.cpp:469) */
  double len_sd;        /* --length-sd             (pbsim.cpp:477) */
  double hp_del_bias;   /* --hp-del-bias           (pbsim.cpp:514) */
  int64_t len_min;      /* --length-min            (pbsim.cpp:363) */
  int64_t len_max;      /* --length-max            (pbsim.cpp:375) */
  int64_t sub_ratio;    /* --difference-ratio      (pbsim.cpp:405) */
  int64_t ins_ratio;
  int64_t del_ratio;
  char id_prefix[64];   /* --id-prefix             (pbsim.cpp:347) */
} pbsim_params;

/* Per-unit results = the `sim.res_*` block (pbsim.cpp:63-70) plus the two
 * histograms' derived values, computed exactly as pbsim.cpp:4082-4105. */
typedef struct pbsim_stats {
  int64_t res_num;       /* reads                                      */
  int64_t res_pass_num;  /* reads * pass_num                           */
  int64_t res_len_total; /* simulated bases over all passes            */
  int64_t res_len_min, res_len_max;
  int64_t res_sub_num, res_ins_num, res_del_num;
  double res_depth;
  double res_len_mean, res_len_sd;
  double res_accuracy_mean, res_accuracy_sd;
  double res_sub_rate, res_ins_rate, res_del_rate;
} pbsim_stats;

/* What one batch produced (device-resident until pbsim_batch_fetch). */
typedef struct pbsim_batch_info {
  int64_t first_read;    /* 1-based index of the first read of the batch          */
  int64_t n_reads;       /* reads simulated speculatively                         */
  int64_t n_final;       /* leading reads that are final under the quota rule     */
  int32_t quota_reached; /* 1: the unit's quota loop ends inside this batch       */
  int32_t need_truncated_read; /* 1: read first_read+n_final must be re-drawn with
                            the truncated length (pbsim.cpp:3795-3800)            */
  int64_t len_total_after; /* pass-0 bases accumulated after the n_final reads    */
  int64_t bases;         /* read bases over all passes of the n_final reads       */
  int64_t read_text_bytes; /* FASTQ (pass_num==1) or SAM text bytes               */
  int64_t maf_text_bytes;  /* MAF text bytes                                      */
  int64_t ref_bases;     /* reference bases consumed (roofline accounting)        */
  int64_t maf_columns;   /* MAF columns written      (roofline accounting)        */
} pbsim_batch_info;

/* Receiver of finished text, called in read order; the bytes are exactly what
 * the reference fprintf()s into its gzip/samtools pipes (pbsim.cpp:4012-4078).
 * Return 0 from a callback to abort the simulation. */
typedef struct pbsim_sink {
  void *user;
  int (*on_read_text)(void *user, const char *text, int64_t bytes); /* FASTQ or SAM records */
  int (*on_maf_text)(void *user, const char *text, int64_t bytes);  /* MAF records          */
} pbsim_sink;

typedef struct pbsim_ctx pbsim_ctx;

/* ---- lifecycle -------------------------------------------------------------- */
void pbsim_params_default(pbsim_params *p);                     /* pbsim.cpp:1539-1685 */
pbsim_ctx *pbsim_create(const pbsim_params *p, int device);     /* main() setup, pbsim.cpp:538-578, 662 */
void pbsim_destroy(pbsim_ctx *ctx);
const char *pbsim_last_error(void);
const char *pbsim_version(void);

/* ---- model tables -----------------------------------------------------------
 * All seven model files the reference ships load (QSHMM-ONT-HQ's 52- and 56-state classes through the reference's own flat
 * indexing of struct qshmm_t, pbsim.cpp:160-166, 5606-5626).  Two shapes of model FILE are refused with PBSIM_FAILED and an
 * error text -- no shipped model has either:
 *   - an ERRHMM model with a hole inside its own accuracy range (a class between acc_min and acc_max without lines): the
 *     reference walks such a read with the rate_mag left over from whichever out-of-range read came before it
 *     (pbsim.cpp:3829-3833 sets it only outside the range), i.e. its output depends on the order of the reads;
 *   - a QSHMM file whose state / column numbers index past the end of tp[] (pbsim.cpp:5606-5626 has no bounds check: the
 *     reference overwrites whatever lies behind the struct). */
int pbsim_load_errhmm(pbsim_ctx *ctx, const char *path);        /* set_errhmm :5640 + tables :3633-3789 */
int pbsim_load_qshmm(pbsim_ctx *ctx, const char *path);         /* set_qshmm  :5570 + tables :1991-2170 */

/* ---- reference sequence of the current unit ---------------------------------
 * `seq` is the raw record (no newlines) as get_genome_seq() assembles it
 * (pbsim.cpp:1014-1033); upper-casing and the per-base homopolymer length
 * (pbsim.cpp:1035-1065) are done on the GPU.  record_index is genome.num
 * (1-based).  The _device form takes a pointer already in this GPU's HBM. */
int pbsim_set_reference(pbsim_ctx *ctx, const uint8_t *seq, int64_t len, int64_t record_index);
int pbsim_set_reference_device(pbsim_ctx *ctx, const void *seq_device, int64_t len, int64_t record_index);
/* Optional: hand the NEXT record over early.  Its upload and preparation (toupper + homopolymer lengths, get_genome_seq
 * pbsim.cpp:1014-1065) run on a stream of their own beside the simulation of the current record; the next
 * pbsim_set_reference* call with the same pointer and length adopts the result instead of doing the work again (any other
 * call drops it).  The bytes must not change in between.  wgs only. */
int pbsim_prefetch_reference(pbsim_ctx *ctx, const uint8_t *seq, int64_t len);
int pbsim_prefetch_reference_device(pbsim_ctx *ctx, const void *seq_device, int64_t len);
/* --hp-del-bias != 1 needs the homopolymer census of ALL records first
 * (pbsim.cpp:677-696): call once per record, then pbsim_finish_hp_census(). */
int pbsim_add_hp_census(pbsim_ctx *ctx, const uint8_t *seq, int64_t len);
int pbsim_finish_hp_census(pbsim_ctx *ctx);

/* ---- transcriptome units (strategy trans) -----------------------------------
 * All transcripts are handed over at once; ids are NUL-terminated strings.
 * Replaces the streaming reader inside simulate_by_errhmm_trans
 * (pbsim.cpp:4428-4485) and get_transcript_inf (:1075). */
int pbsim_set_transcripts(pbsim_ctx *ctx, int64_t n, const char *const *ids, const int64_t *plus_exp,
                          const int64_t *minus_exp, const uint8_t *const *seqs, const int64_t *lens);

/* ---- templates (strategy templ): each FASTA record of --template is one unit that is
 * read once, full length, '+' strand (get_templ_inf :1366, simulate_by_*_templ :5055-5103). */
int pbsim_set_templates(pbsim_ctx *ctx, int64_t n, const char *const *ids, const uint8_t *const *seqs,
                        const int64_t *lens);

/* ---- whole-unit drivers = the reference seam --------------------------------
 * pbsim_simulate_wgs  : one FASTA record, quota loop included
 *                       (simulate_by_errhmm :3792-4080 / simulate_by_qshmm :2173-2385)
 * pbsim_simulate_trans: every transcript (simulate_by_errhmm_trans :4428-4770) */
int pbsim_simulate_wgs(pbsim_ctx *ctx, const pbsim_sink *sink);
int pbsim_simulate_trans(pbsim_ctx *ctx, const pbsim_sink *sink);
int pbsim_simulate_templ(pbsim_ctx *ctx, const pbsim_sink *sink); /* simulate_by_errhmm_templ :4807 / _qshmm_templ :3055 */
/* Read-block shard of a unit set (multi-GPU front-end, one rank per GPU): reads first_read .. first_read + n_reads - 1 of
 * the global numbering sim.res_num of simulate_by_*_trans / _templ (pbsim.cpp:4516-4522 assigns reads to transcripts in
 * file order).  Every byte of a read depends only on (seed, read, pass, event), so the shards of all ranks concatenated
 * in read order are the bytes pbsim_simulate_trans delivers; the statistics cover the shard only (the caller sums them).
 * pbsim_unit_reads: reads of the whole set (sum of the expression values, or the number of templates). */
int pbsim_simulate_units_range(pbsim_ctx *ctx, int64_t first_read, int64_t n_reads, const pbsim_sink *sink);
int64_t pbsim_unit_reads(pbsim_ctx *ctx);
/* Host-side readers of the reference's unit files for callers above the ABI that are not C++: parse `path` like
 * simulate_by_*_trans does its --transcript file (pbsim.cpp:4428-4485; get_transcript_inf :1075) resp. get_templ_inf its
 * --template FASTA (:1366) and hand the units over (pbsim_set_transcripts / pbsim_set_templates).
 * stats[0] = units, stats[1] = total expression value (trans) resp. total template length (templ). */
int pbsim_load_transcript_file(pbsim_ctx *ctx, const char *path, int64_t stats[2]);
int pbsim_load_template_file(pbsim_ctx *ctx, const char *path, int64_t stats[2]);
int pbsim_get_stats(pbsim_ctx *ctx, pbsim_stats *out);          /* pbsim.cpp:4082-4105, 5541-5562 */
/* Sampling method (--method sample, wgs only, single pass).  pbsim_set_sample_profile hands over what
 * get_sample_inf (pbsim.cpp:1155-1330) leaves in its filtered profile: the quality strings that passed the
 * length / accuracy filter, in file order (the caller parses the FASTQ, keeps the statistics and the
 * sample_profile_<ID> files).  pbsim_simulate_sample = simulate_by_sample (pbsim.cpp:1694-1949) for the
 * current reference record. */
int pbsim_set_sample_profile(pbsim_ctx *ctx, int64_t n, const uint8_t *const *quals, const int64_t *lens);
int pbsim_simulate_sample(pbsim_ctx *ctx, const pbsim_sink *sink);
/* SAM header the reference's main() writes when it opens the samtools pipe for a
 * unit (pass_num > 1; pbsim.cpp:721-722 wgs, :784-785 trans/templ).  Returns the
 * byte count (excluding the NUL), or the size needed when buf is NULL/too small. */
int64_t pbsim_sam_header(pbsim_ctx *ctx, char *buf, int64_t cap);
/* pass_num > 1 only: make the read sink receive BAM alignment records (SAMv1 4.2, uncompressed,
 * one per subread, what `samtools view -b` would build from the SAM text of pbsim.cpp:4016-4027)
 * instead of SAM text; pbsim_bam_header() gives the bytes that precede the first record.  The
 * caller BGZF-frames the stream (the CLI does, replacing the samtools child of pbsim.cpp:716). */
int pbsim_set_bam_output(pbsim_ctx *ctx, int on);
int64_t pbsim_bam_header(pbsim_ctx *ctx, char *buf, int64_t cap);
/* Compression on the GPU, replacing the `gzip -c` children of pbsim.cpp:708-730 and the BGZF layer of
 * `samtools view -b` (pbsim.cpp:716).  pbsim_set_deflate(ctx, mask): bit 0 makes the read sink, bit 1
 * the MAF sink receive gzip members (RFC 1952) instead of text: one member per 32 KiB of text, each carrying the BGZF 'BC'
 * extra field (SAMv1 4.1), so the concatenation is a valid multi-member .gz and, for BAM records, a
 * valid BAM container once the caller has written a compressed header in front and the BGZF EOF marker
 * behind.  Decompressed bytes are exactly the text the sink would have received otherwise.
 * Bit 2 (with bits 0 and 1): the two sinks are independent -- on_read_text is then called from a second host thread
 * while on_maf_text runs on the caller's (two files are written side by side; never one callback concurrently with itself).
 * pbsim_batch_fetch_deflated is the batch-level primitive (after pbsim_batch_finalize; caps from
 * pbsim_deflate_bound).  pbsim_deflate_buffer runs host bytes through the same kernels (file headers). */
int pbsim_set_deflate(pbsim_ctx *ctx, int mask);
int64_t pbsim_deflate_bound(int64_t text_bytes);
int pbsim_batch_fetch_deflated(pbsim_ctx *ctx, char *read_gz, int64_t read_cap, char *maf_gz, int64_t maf_cap,
                               int64_t *read_gz_bytes, int64_t *maf_gz_bytes);
int pbsim_deflate_buffer(pbsim_ctx *ctx, const void *src, int64_t n, void *dst, int64_t cap, int64_t *out_bytes);

/* ---- batch primitives (used by the drivers above, bench.py, multi-GPU) ------
 * pbsim_batch_walk     header draw + bucketing + HMM walk of reads
 *                      [first_read, first_read+n_reads) of the current unit,
 *                      speculatively un-truncated; *pass0_bases = their pass-0
 *                      output bases.  truncate_remaining >= 0 (n_reads must be
 *                      1) applies the quota truncation of pbsim.cpp:3795-3800
 *                      with quota-len_total = truncate_remaining.
 * pbsim_batch_finalize places the quota cut given the pass-0 bases simulated
 *                      before this batch and emits FASTQ|SAM + MAF text for the
 *                      final reads into device buffers.
 * pbsim_batch_fetch    copies the text to host memory (either may be NULL). */
int pbsim_batch_walk(pbsim_ctx *ctx, int64_t first_read, int64_t n_reads, int64_t truncate_remaining,
                     int64_t *pass0_bases);
/* Asynchronous halves of pbsim_batch_walk, and batch slots: a context owns
 * pbsim_slot_count() independent batch slots (buffers + HIP stream each);
 * pbsim_select_slot() chooses the one the pbsim_batch_* calls act on.  Beginning
 * batch k+1 on another slot before ending batch k keeps the GPU busy while the
 * longest reads of batch k drain (the drivers above do exactly that). */
int pbsim_slot_count(void);
int pbsim_select_slot(pbsim_ctx *ctx, int slot);
int pbsim_batch_walk_begin(pbsim_ctx *ctx, int64_t first_read, int64_t n_reads, int64_t truncate_remaining);
int pbsim_batch_walk_end(pbsim_ctx *ctx, int64_t *pass0_bases);
int pbsim_batch_finalize(pbsim_ctx *ctx, int64_t len_total_before, pbsim_batch_info *info);
int pbsim_batch_fetch(pbsim_ctx *ctx, char *read_text, char *maf_text);
/* after pbsim_batch_walk_end: the header draws and pass-0 results of the batch's reads, n_reads values each (any pointer may
 * be NULL): raw length drawn (before the clip to the record, pbsim.cpp:3793-3794), length walked, pass-0 bases produced.  What
 * the quota rule (pbsim.cpp:3792-3800) is a function of -- measurement / test hook. */
int pbsim_batch_fetch_lengths(pbsim_ctx *ctx, int32_t *rawlen, int32_t *len, int32_t *out_len_pass0);
/* adds the n_final reads of the finalized batch to the unit's statistics */
int pbsim_batch_account(pbsim_ctx *ctx);
/* resets the per-unit statistics (init_sim_res, pbsim.cpp:1437) */
int pbsim_reset_stats(pbsim_ctx *ctx);
/* quota of the current unit: (long long)(depth*len), pbsim.cpp:705 */
int64_t pbsim_unit_quota(pbsim_ctx *ctx);
/* reads the engine sizes one batch to (from the scratch budget) */
int64_t pbsim_batch_capacity(pbsim_ctx *ctx);
/* scratch pool per slot in bytes (default: PBSIM_SCRATCH_MB env or 8 GiB) */
int pbsim_set_scratch_bytes(pbsim_ctx *ctx, int64_t bytes);
/* A context sizes its pools for the jobs it runs and keeps them (re-allocating tens of GB stalls a job for seconds): one that
 * switches to another KIND of job -- output delivered through a sink, then left in HBM; another pass count -- should give the
 * old pools back first.  Releases the slots' scratch pools, text and compression buffers and the page-locked host blocks a
 * several-rank job compresses into (those go back by themselves only after sixteen jobs that did not use them: giving a block
 * back and page-locking it again cost tens of milliseconds each); nothing may be in flight. */
int pbsim_release_pools(pbsim_ctx *ctx);

/* ---- the whole job on one or several GPUs -------------------------------------
 * main() runs its records one after the other (pbsim.cpp:667-759).  Here ALL records of the genome are made resident in
 * HBM (288 GB holds any genome the reference can read: 3 bytes per base there, 2 here) and ONE pipeline of read batches
 * runs across them: batches of record n+1 start as soon as record n has enough reads in flight, so the tail of a record
 * (its last text emission, its truncated last reads, pbsim.cpp:3795-3800) hides behind the next record's walks.
 *
 * Several GPUs: one context per GPU ("rank"), every rank holds every record (C1: the caller broadcasts them) and runs
 * the same job; a round of the pipeline gives rank r the r-th block of the round's reads.  The ranks only exchange
 * integers through the caller's pbsim_comm: per round ONE all-gather (C3) of the blocks' pass-0 bases and largest raw
 * length (places every rank's quota prefix and tells whether any read of the round can touch the quota) together with the
 * PREVIOUS round's text sizes (every rank learns the byte range of its text inside the record's stream) -- plus a second one for
 * the cut in the one round of a record that reaches the quota; per record three collectives for the
 * statistics (C2: counters, min/max, the two histograms; the order-dependent accuracy sum is folded in read order).  The
 * concatenation of all ranks' text in offset order is byte for byte what one GPU delivers, and so are the statistics.
 *
 * pbsim_comm: blocking collectives over the `world` ranks, called by every rank in the same order (the job is
 * deterministic in the gathered values, so the ranks stay in lockstep by construction).  torch.distributed (backend nccl
 * = RCCL), RCCL itself, MPI or -- for several contexts inside one process -- a host barrier all fit. */
#define PBSIM_OP_SUM 0
#define PBSIM_OP_MIN 1
#define PBSIM_OP_MAX 2
typedef struct pbsim_comm {
  void *user;
  int32_t rank, world;
  /* every rank contributes n values; recv holds world * n, rank-major */
  int (*all_gather_i64)(void *user, const int64_t *send, int64_t n, int64_t *recv);
  /* element-wise reduction over the ranks, in place; op = PBSIM_OP_* */
  int (*all_reduce_i64)(void *user, int64_t *buf, int64_t n, int32_t op);
  /* optional (C1): `bytes` bytes at `ptr` (device memory of this rank's GPU when on_device, else host memory) of rank
   * `root` to every rank.  NULL: the caller has every rank load the records itself. */
  int (*broadcast)(void *user, void *ptr, int64_t bytes, int32_t root, int32_t on_device);
  /* optional: called by a rank whose job failed for a reason the other ranks cannot know yet.  It must make the collectives
   * the other ranks are waiting in (or will enter) return 0, or end their processes.  Failures the library can foresee travel
   * in the status words of its own exchanges and all ranks return PBSIM_FAILED together; for the rest (a HIP error between two
   * exchanges) a communicator without `abort` leaves the other ranks waiting: the caller must then tear the process group
   * down when pbsim_job_run fails on any rank.  NULL is allowed. */
  int (*abort)(void *user);
} pbsim_comm;

/* Receiver of a job's output.  Text arrives as (record, bytes, offset): `offset` is the position of the piece inside the
 * record's read stream resp. MAF stream (what the reference writes into <prefix>_NNNN.fq|sam / .maf after any header the
 * caller puts in front), so every rank can pwrite() its pieces into the final file; on one GPU the offsets simply run up.
 * on_record_done is called on EVERY rank, in record order, with the statistics of the whole record (identical on all
 * ranks) and the total size of its two streams.  Return 0 to abort. */
typedef struct pbsim_record_sink {
  void *user;
  int (*on_read_text)(void *user, int64_t record, const char *text, int64_t bytes, int64_t offset);
  int (*on_maf_text)(void *user, int64_t record, const char *text, int64_t bytes, int64_t offset);
  int (*on_record_done)(void *user, int64_t record, const pbsim_stats *stats, int64_t read_bytes, int64_t maf_bytes);
} pbsim_record_sink;

/* Records are numbered 1.. in the order they are added (genome.num).  The bytes are uploaded and prepared (toupper +
 * homopolymer lengths, pbsim.cpp:1014-1065) asynchronously; pbsim_job_run simulates every record added since the last
 * pbsim_job_clear.  --hp-del-bias != 1 needs no separate census pass here: the census of all resident records is known
 * before the first walk (pbsim.cpp:677-696).  wgs, methods errhmm / qshmm. */
int pbsim_job_add_record(pbsim_ctx *ctx, const uint8_t *seq, int64_t len);
int pbsim_job_add_record_device(pbsim_ctx *ctx, const void *seq_device, int64_t len);
/* The record as it lies in its FASTA file: `lines` = its sequence lines, line feeds included (`bytes` of them; memory of a
 * mapped file will do), `len` = bytes - line feeds.  The copy loop of get_genome_seq (pbsim.cpp:1014-1033: every byte of the
 * lines except the line feeds) runs on the GPU behind the upload; the call fails when the GPU keeps another count than `len`. */
int pbsim_job_add_record_lines(pbsim_ctx *ctx, const uint8_t *lines, int64_t bytes, int64_t len);
/* C1 in one call: rank `root` passes the record (the others may pass NULL); with comm->broadcast the bytes travel GPU to
 * GPU, without it every rank must pass them. */
int pbsim_job_add_record_comm(pbsim_ctx *ctx, const uint8_t *seq, int64_t len, const pbsim_comm *comm, int32_t root);
int64_t pbsim_job_records(pbsim_ctx *ctx);
/* The job's records announced in advance (their number and lengths): pbsim_job_run may then be called BEFORE they have all been
 * added, and pbsim_job_add_record* may be called from another thread while it runs -- in order, each with the announced length.
 * The job begins as soon as record 1 is resident and prepared and begins a later record's first round when that record is, so
 * the upload / broadcast (C1) and preparation (K0) of records 2.. hide behind the rounds in front of them: what a caller that
 * hands over FRESH records waits for is record 1's share, not the genome's (main() reads its records one at a time,
 * pbsim.cpp:666-759; bench.py `value_from_fresh_records`).  With --hp-del-bias != 1 the job waits for every record first (the
 * homopolymer census of all records precedes the first read, pbsim.cpp:677-696).  pbsim_job_feed_abort: the feeding thread gives
 * up (a read error, a failed broadcast); a pbsim_job_run that waits for a record then fails with `why` (it also gives up by
 * itself after 600 s).  pbsim_job_begin / pbsim_job_clear forget the announcement. */
int pbsim_job_expect(pbsim_ctx *ctx, int64_t n_records, const int64_t *lens);
int pbsim_job_feed_abort(pbsim_ctx *ctx, const char *why);
/* drops the records; the next one added is record `first_record` (a genome larger than HBM runs as several jobs whose
 * numbering continues; --hp-del-bias != 1 then needs pbsim_add_hp_census / pbsim_finish_hp_census over ALL records first:
 * pbsim_job_run of a job with first_record > 1 fails without it rather than taking a census per record group).
 * pbsim_job_clear = pbsim_job_begin(ctx, 1). */
int pbsim_job_begin(pbsim_ctx *ctx, int64_t first_record);
int pbsim_job_clear(pbsim_ctx *ctx);
/* By default a record's rounds run back to back (its tail reads and its statistics hide behind the next record's rounds).
 * pbsim_job_set_interleave(ctx, k): the rounds of up to k consecutive records alternate instead, the record that is furthest
 * behind first -- the bytes of k records then arrive side by side, which is what a caller wants who writes a file pair per
 * record onto a file system that takes a few GB/s per file (the CLI sets 4).  Same bytes, same offsets, same statistics. */
int pbsim_job_set_interleave(pbsim_ctx *ctx, int records);
/* comm == NULL: this GPU alone.  sink may be NULL (text stays in HBM: measurements) and so may any of its callbacks. */
int pbsim_job_run(pbsim_ctx *ctx, const pbsim_comm *comm, const pbsim_record_sink *sink);
/* SAM / BAM header of record `record` of the job (pbsim_sam_header / pbsim_bam_header are those of the current unit) */
int64_t pbsim_job_sam_header(pbsim_ctx *ctx, int64_t record, char *buf, int64_t cap);
int64_t pbsim_job_bam_header(pbsim_ctx *ctx, int64_t record, char *buf, int64_t cap);
/* what the last pbsim_job_run did, for bench.py: [0] reads walked (speculation included), [1] reads delivered,
 * [2] rounds, [3] bases delivered (all passes), [4] wall microseconds, [5] microseconds this rank waited in collectives,
 * [6] reference bases consumed and [7] MAF columns written by the delivered reads (roofline accounting) */
int pbsim_job_counters(pbsim_ctx *ctx, int64_t out[8]);
/* where this rank's round loop spent the wall time of the last pbsim_job_run, microseconds: [0] wall; the loop waited
 * [1] for walks (pbsim_batch_walk_end of the round in front), [2] for the cut and the text sizes, [3] for the previous round's
 * bytes to reach host memory (GPU compression + link), [4] in the per-round collectives, [5] accounting statistics,
 * [6] for a record's truncated tail reads at its merge (exposed tail), [7] for the delivery thread at a merge, [8] for a free
 * slot (or for an announced record to arrive, pbsim_job_expect), [9] in the statistics merge (C2), [10] enqueueing rounds, [11] stepping tail reads between rounds; [12] time the
 * delivery thread was busy (compression + copies + sink callbacks; runs beside the loop); [13] top-up rounds, [14] truncated
 * tail reads walked by this rank, [15] rounds kept in flight. */
int pbsim_job_breakdown(pbsim_ctx *ctx, double out[16]);

/* Which exchange of the round sequence this rank's pbsim_job_run is about to enter, readable from inside a pbsim_comm callback
 * (same thread): [0] the kind of exchange -- 6 = "AC", the ONE all-gather of a round that is expected to stay clear of the
 * quota (8 words per rank: [0] pass-0 bases of the block, [1] walk status, [2] largest raw length drawn in the block, [3] status
 * of the text sizes, [4] 1 if [5..7] hold the PREVIOUS round's delivery: compressed read bytes, MAF bytes, status); 1 = "A" of
 * a round that is expected to reach the quota (same 8 words, no delivery part) and 7 = its "BC" ([0] final reads of the block,
 * [1] a truncated read is due, [2] len_total behind the block, [3] status, [4..7] the previous round's delivery); 2 = "B" on its
 * own (a round begun as clear that can touch the quota after all: same words, no delivery part); 3 = the pending round's byte
 * counts on their own (3 words: at a record's merge, at the end of the job); 4 = a record's statistics merge (three
 * collectives); 5 = the agreement on pool size and caps in front of the round loop;
 * [1] record (0-based index in the job), [2] first read of the round (rank r walks [2] + r * [3] ..), [3] reads per rank,
 * [4] ranks, [5] the record's len_total and [7] its next read in front of the round, [6] its quota.  For communicators that
 * model or replay the other ranks (bench.py --replay-ranks measures an N-rank job's per-rank critical path on one GPU). */
int pbsim_job_progress(pbsim_ctx *ctx, int64_t out[8]);

/* The sampling method (pbsim_simulate_sample, pbsim.cpp:1694-1949) on several ranks, for the current record: one context per
 * GPU, every rank has set the same reference and profile.  The copies of one string are a chain, strings are independent, and
 * the test `len_total < quota` at a read's start (pbsim.cpp:1749) is the wgs quota rule's prefix dependence -- a round gives
 * rank r the r-th run of a sweep's strings, three all-gathers of a few integers place the quota prefix, the cut and every
 * rank's byte range.  The sink receives (record, bytes, offset) like a job's (record = the record_index of
 * pbsim_set_reference); on_record_done reports the merged statistics on every rank (pbsim_get_stats too).  The concatenation
 * of all ranks' bytes in offset order is what pbsim_simulate_sample delivers on one GPU. */
int pbsim_simulate_sample_comm(pbsim_ctx *ctx, const pbsim_comm *comm, const pbsim_record_sink *sink);

/* Statistics primitives for callers that shard a unit set over several contexts themselves (trans / templ:
 * pbsim_simulate_units_range per rank): keep the per-task accuracy values while accounting (before simulating), then
 * merge -- afterwards pbsim_get_stats on every rank reports the whole unit set exactly as one GPU would
 * (pbsim.cpp:4082-4105, the order-dependent sum of :4003 included).  pbsim_stats_add_tasks accounts tasks given as plain
 * arrays (device-free; tests of the merge). */
int pbsim_stats_keep_values(pbsim_ctx *ctx, int on);
int pbsim_stats_merge(pbsim_ctx *ctx, const pbsim_comm *comm);
int pbsim_stats_add_tasks(pbsim_ctx *ctx, int64_t first_task, int64_t n, const int32_t *out_len, const int32_t *nsub,
                          const int32_t *nins, const int32_t *ndel, const double *qsum);
/* the ":::: Simulation stats ::::" block of print_simulation_stats (pbsim.cpp:5541-5564); unit = genome.num (wgs).
 * Returns the byte count (excluding the NUL), or the size needed when buf is NULL / too small. */
int64_t pbsim_format_stats(const pbsim_params *p, const pbsim_stats *s, int64_t unit, char *buf, int64_t cap);

/* ---- the command line as a library call ------------------------------------------
 * Everything `pbsim` does for one rank: the reference's options (pbsim.cpp:257-282) and set_sim_param validation
 * (:1451-1688), the stderr report blocks, the <prefix>_NNNN.ref files, the output files.  comm == NULL: this GPU alone.
 * With a communicator every rank calls it with the same argv; rank 0 prints the report and creates the files, every rank
 * writes its own byte ranges of them.  The `pbsim` binary calls this from one host thread per GPU of --devices (host
 * barrier or RCCL communicator); pbsim3_amd/run_multi.py calls it from one process per GPU under torchrun
 * (torch.distributed).  Returns the process exit status (0, or 255 like the reference's exit(-1)). */
int pbsim_cli_main(int argc, char **argv, const pbsim_comm *comm, int device);

/* ---- RCCL communicator of a one-process-per-GPU launch ------------------------------
 * `pbsim --devices` runs its ranks as threads of one process (ncclCommInitAll or a host barrier).  A launcher that starts one
 * PROCESS per GPU (torchrun, mpirun, a shell loop) gets its pbsim_comm here: rank 0 makes the id (ncclGetUniqueId, 128 bytes),
 * the launcher's side channel carries it to the others, every rank enters pbsim_rccl_comm_create with it (ncclCommInitRank:
 * collective over the `world` ranks, one distinct GPU each).  The communicator's callbacks run ncclAllGather (C3),
 * ncclAllReduce (C2) and ncclBroadcast (C1) on a stream of their own, small messages through page-locked staging, every wait
 * polled with a watchdog (PBSIM_COMM_TIMEOUT_S) -- what SURVEY 8(e) names; the reference has no analogue (one process, libc
 * only, pbsim.cpp:4-14).  librccl is opened at run time.
 * pbsim_rccl_unique_id: writes the id (returns its size; with id == NULL or cap too small only the size), 0 on failure.
 * pbsim_rccl_comm_create_file: the same with a file as the side channel -- rank 0 publishes the id at `path` (written whole,
 * then renamed), the others wait for it (PBSIM_RENDEZVOUS_TIMEOUT_S, default 120); rank 0 removes the file once every rank has joined (use a
 * path of the launch's own: a stale file of a crashed launch would be read as this launch's id).
 * pbsim_rccl_comm_info: [0] the ranks RCCL itself counts in the communicator (ncclCommCount), [1] this rank's number there,
 * [2] the device, [3] collectives issued so far.  NULL / PBSIM_FAILED + pbsim_last_error() on failure. */
#define PBSIM_RCCL_ID_BYTES 128
int64_t pbsim_rccl_unique_id(void *id, int64_t cap);
pbsim_comm *pbsim_rccl_comm_create(const void *id, int64_t id_bytes, int32_t rank, int32_t world, int32_t device);
pbsim_comm *pbsim_rccl_comm_create_file(const char *path, int32_t rank, int32_t world, int32_t device);
int pbsim_rccl_comm_info(const pbsim_comm *comm, int64_t out[4]);
void pbsim_rccl_comm_destroy(pbsim_comm *comm);

/* ---- host placement of a rank ----------------------------------------------------------
 * The reference is one thread on one socket; a rank here is a thread that feeds one GPU and receives its output over PCIe.
 * Binds the calling thread -- and the threads and page-locked buffers it creates afterwards -- to the CPUs and the memory of
 * the NUMA node GPU `device` is attached to (sysfs: KFD topology -> PCI address -> numa_node / local_cpulist).  Call it
 * before the first HIP call of the thread's process (pbsim_cli_main does, per rank).  `what` (optional) receives a one-line
 * description, empty when there was nothing to bind to (one node, no topology, PBSIM_NUMA_BIND=0).  Always succeeds. */
int pbsim_bind_host_to_device(int device, char *what, int64_t cap);

/* ---- measurement hooks (bench.py) -------------------------------------------
 * Accumulated HIP-event time of the walk kernel launches since the last reset,
 * measured on the engine's own stream, and the number of launches. */
int pbsim_prof_reset(pbsim_ctx *ctx);
int pbsim_prof_get(pbsim_ctx *ctx, double *walk_ms, int64_t *walk_launches, double *total_ms);
/* pbsim_prof_get counts the walk launches of batches; a launch that carries one truncated tail read (pbsim.cpp:3795-3800:
 * the latency of a single lane, no bytes to speak of) is counted here instead */
int pbsim_prof_tail(pbsim_ctx *ctx, double *tail_ms, int64_t *tail_launches);
/* milliseconds since the reset during which at least one walk kernel ran (the launches of different slots overlap) */
int pbsim_prof_walk_busy(pbsim_ctx *ctx, double *busy_ms);
/* launches of a wave walker (one wave per task: k_walk_errhmm_coop / k_walk_qshmm_coop) since the last reset (or creation) */
int64_t pbsim_prof_wave_launches(pbsim_ctx *ctx);
/* The scratch rows of a task (two MAF rows, a quality row for the quality-score methods) are laid out for factor x length + 64
 * columns.  2 is the reference's own bound (its buffers are 2 * len_max + 1, pbsim.cpp:5488); a context starts there, keeps the
 * largest need its walks have reported and lays the next batches out with a little more -- a batch in which a read runs out of
 * row all the same is walked again at 2 inside pbsim_batch_walk_end.  [0] the factor the next batches get, [1] the largest
 * (columns - 64) / length seen, [2] batches walked twice.  PBSIM_SCRATCH_FACTOR fixes the factor. */
int pbsim_scratch_state(pbsim_ctx *ctx, double out[3]);
/* the two kernels next in line, timed with HIP events on the streams they run on, since the last reset:
 * [0] ms, [1] launches, [2] bytes read (scratch rows), [3] bytes written (text) of the text emission (k_text_*);
 * [4] ms, [5] launches, [6] text bytes in, [7] member bytes out of k_deflate_chunks */
int pbsim_prof_secondary(pbsim_ctx *ctx, double out[8]);
/* raw HIP stream handle (hipStream_t) of the engine, for external event timing */
void *pbsim_stream(pbsim_ctx *ctx);
/* hipDeviceSynchronize on the context's GPU: the bracket of a timed region for a host that holds no HIP runtime handle of its
 * own (bench.py --no-torch: a process with the system runtime only, which rocprofv3 traces without changing how copies run) */
int pbsim_device_synchronize(pbsim_ctx *ctx);

/* ---- known-answer hooks (tests) ---------------------------------------------
 * Philox4x32-10 block as the kernels compute it, on the host build of the same
 * header (no GPU needed). */
void pbsim_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* sha256-able dumps of the host-built tables (Q13 fixtures): returns bytes
 * written, or the size needed when buf is NULL. which: 0 prob2len (int32
 * [len_rand_value+1]), 1 prob2acc (uint8 [acc_rand_value+1]), 2 class tables */
int64_t pbsim_dump_table(pbsim_ctx *ctx, int which, void *buf, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* PBSIM3_AMD_H */
