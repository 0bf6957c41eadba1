// unit_io.h -- host loaders of the CLI shim: multi-FASTA splitter / record
// loader (get_genome_inf pbsim.cpp:896-991, get_genome_seq :997-1033) and the
// transcript TSV reader (get_transcript_inf :1075-1136, the in-loop reader
// :4428-4455).  Plain host C++; the per-base work (toupper, hp) is on the GPU.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace pbsim {

struct GenomeInfo {
  long num_seq = 0;
  long max_len = 0;
  std::vector<long> len;          // per record
  std::vector<std::string> id;    // per record (truncated to 128 chars)
};

// Splits <file> into <prefix>_NNNN.ref, printing the reference's ":::: Reference stats ::::" block.
bool split_genome(const char *file, const char *prefix, GenomeInfo *info, std::string *err);
// The same pass over the FASTA without stdio (round 4): the file mapped, header lines found by memchr on threads, the line
// feeds of every record counted on threads -- a record is then its lines as they lie in the file plus its length, and the
// line feeds are squeezed out on the GPU (pbsim_job_add_record_lines).  Prints the same ":::: Reference stats ::::" block and
// fails with the same messages in the same order as the fgets pass above, which stays the checker (tests) and the fallback:
// `*fallback` is set -- and nothing printed -- for what only fgets semantics define (no regular file, NUL bytes, bytes in front
// of the first header).
struct FastaRecord {
  std::string id;           // the header's first 128 characters (pbsim.cpp:939-940)
  const uint8_t *lines = nullptr;  // the record's sequence lines in the mapped file, line feeds included
  int64_t bytes = 0;        // of `lines`
  int64_t len = 0;          // bases = bytes - line feeds
  int64_t max_line = 0;     // longest line: lines of >= 10239 characters reach the .ref file in fgets chunks (pbsim.cpp:914)
};
struct FastaMap {
  void *map = nullptr;
  size_t size = 0;
  std::vector<FastaRecord> recs;
  FastaMap() = default;
  FastaMap(const FastaMap &) = delete;
  FastaMap &operator=(const FastaMap &) = delete;
  ~FastaMap();
};
bool map_genome(const char *file, FastaMap *m, GenomeInfo *info, bool print_stats, bool *fallback, std::string *err);
// <prefix>_NNNN.ref of a mapped record, byte for byte what get_genome_inf writes (pbsim.cpp:948-964)
bool write_ref_record(const char *prefix, long num, const FastaRecord &r, std::string *err);
// Re-reads <prefix>_NNNN.ref into one contiguous record (no newlines, case preserved).
bool load_ref_record(const char *prefix, long num, std::string *seq, std::string *err);

struct Transcript {
  std::string id;
  long plus = 0, minus = 0;
  std::string seq;
};
bool read_transcripts(const char *file, std::vector<Transcript> *out, long *total_exp, std::string *err);
// --template FASTA (get_templ_inf pbsim.cpp:1366-1418, the reader inside simulate_by_*_templ :5055-5362):
// one Transcript per record with a non-empty sequence (plus=1); *num counts '>' lines, *len_total bases
bool read_templates(const char *file, std::vector<Transcript> *out, long *num, long long *len_total, std::string *err);

// Sampling method: the filtered profile of get_sample_inf (pbsim.cpp:1155-1330).  `quals` are the quality strings
// that passed --length-min/max and --accuracy-min/max, in file order; the statistics are the ones
// print_sample_stats prints (:1336-1360) and the stored profile's .stats file holds (:1317-1326).
struct SampleProfile {
  long num = 0, len_min = 0, len_max = 0;
  long long len_total = 0;
  long num_filtered = 0, len_min_filtered = 0, len_max_filtered = 0;
  long long len_total_filtered = 0;
  double len_mean_filtered = 0, len_sd_filtered = 0, accuracy_mean_filtered = 0, accuracy_sd_filtered = 0;
  std::vector<std::string> quals;
};
// parse + filter a FASTQ (line-feed counting and BUF_SIZE chunking as in the reference)
bool read_sample_fastq(const char *file, long len_min, long len_max, double acc_min, double acc_max, SampleProfile *out,
                       std::string *err);
// the same through fgets, chunk by chunk as the reference reads it (files with NUL bytes, pipes; the checker of the fast path)
bool read_sample_fastq_stdio(const char *file, long len_min, long len_max, double acc_min, double acc_max, SampleProfile *out,
                             std::string *err);
// sample_profile_<ID>.fastq (one quality string per line) + .stats ("key<TAB>value" lines)
bool write_sample_profile(const std::string &fq, const std::string &stats, const SampleProfile &p, std::string *err);
bool read_sample_profile(const std::string &fq, const std::string &stats, SampleProfile *out, std::string *err);

}  // namespace pbsim
