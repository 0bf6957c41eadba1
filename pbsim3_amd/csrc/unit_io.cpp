#include "unit_io.h"

#include <fcntl.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <thread>

namespace pbsim {

namespace {
constexpr int kBuf = 10240;       // BUF_SIZE pbsim.cpp:20: fgets chunking is observable in the .ref files
constexpr size_t kIoBuf = 4u << 20;  // stdio buffers of the genome files (the default is one 4 KiB block per read()/write())
constexpr int kIdMax = 128;       // REF_ID_LEN_MAX / TRANS_ID_LEN_MAX
constexpr long kRefNumMax = 9999; // REF_SEQ_NUM_MAX
constexpr long kRefLenMax = 1000000000L;
constexpr long kRefLenMin = 100;

bool chomp(char *s) {
  size_t n = strlen(s);
  if (n && s[n - 1] == '\n') {
    s[n - 1] = '\0';
    return true;
  }
  return false;
}
}  // namespace

bool split_genome(const char *file, const char *prefix, GenomeInfo *info, std::string *err) {
  fprintf(stderr, ":::: Reference stats ::::\n\n");
  fprintf(stderr, "file name : %s\n", file);
  fprintf(stderr, "\n");
  FILE *fp = fopen(file, "r");
  if (!fp) {
    *err = std::string("Cannot open file: ") + file;
    return false;
  }
  std::unique_ptr<char[]> line(new char[kBuf]);
  std::unique_ptr<char[]> in_buf(new char[kIoBuf]), out_buf(new char[kIoBuf]);  // outlive both streams
  setvbuf(fp, in_buf.get(), _IOFBF, kIoBuf);
  FILE *out = nullptr;
  long cur_len = 0;
  std::string cur_id;
  char name[4096];
  auto close_record = [&]() -> bool {
    if (cur_len < kRefLenMin) {
      *err = "Reference is too short. Acceptable length >= 100.";
      return false;
    }
    fprintf(stderr, "ref.%ld (len:%ld) : %s\n", info->num_seq, cur_len, cur_id.c_str());
    fclose(out);
    out = nullptr;
    info->len.push_back(cur_len);
    info->id.push_back(cur_id);
    if (cur_len > info->max_len) info->max_len = cur_len;
    return true;
  };
  bool ok = true;
  while (ok && fgets(line.get(), kBuf, fp)) {
    bool nl = chomp(line.get());
    if (line[0] == '>') {
      if (info->num_seq != 0 && !(ok = close_record())) break;
      info->num_seq++;
      if (info->num_seq > kRefNumMax) {
        *err = "References are too many. Max number of reference is 9999.";
        ok = false;
        break;
      }
      cur_id.assign(line.get() + 1, strnlen(line.get() + 1, kIdMax));
      snprintf(name, sizeof name, "%s_%04ld.ref", prefix, info->num_seq);
      if (!(out = fopen(name, "w"))) {
        *err = std::string("Cannot open output file: ") + name;
        ok = false;
        break;
      }
      setvbuf(out, out_buf.get(), _IOFBF, kIoBuf);
      cur_len = 0;
      while (!nl) {  // rest of an over-long header line
        if (!fgets(line.get(), kBuf, fp)) break;
        nl = chomp(line.get());
      }
      fprintf(out, ">%s\n", cur_id.c_str());
    } else {
      if (!out) {
        *err = "sequence data before the first FASTA header";
        ok = false;
        break;
      }
      const size_t n = strlen(line.get());
      cur_len += (long)n;
      if (cur_len > kRefLenMax) {
        *err = "Reference is too long. Acceptable length <= 1000000000.";
        ok = false;
        break;
      }
      line[n] = '\n';  // fprintf(out, "%s\n", line) without the formatter: 12.5 M lines in a 1 Gbp record
      fwrite(line.get(), 1, n + 1, out);
      line[n] = '\0';
    }
  }
  fclose(fp);
  if (ok) {
    if (!out) {
      *err = "Reference is too short. Acceptable length >= 100.";
      ok = false;
    } else {
      ok = close_record();
    }
  }
  if (out) fclose(out);
  if (ok) fprintf(stderr, "\n");
  return ok;
}

// ---- the mapped FASTA (unit_io.h) ------------------------------------------------------------------------------------------
FastaMap::~FastaMap() {
  if (map) munmap(map, size);
}

namespace {
constexpr int64_t kChunk = kBuf - 1;  // characters one fgets takes (pbsim.cpp:914)

template <class F>
void parallel_blocks(size_t n_blocks, F &&f) {
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const size_t nt = std::min<size_t>(std::min<size_t>(hw, 32), n_blocks);
  if (nt <= 1) {
    for (size_t b = 0; b < n_blocks; b++) f(b);
    return;
  }
  std::atomic<size_t> next{0};
  std::vector<std::thread> th;
  for (size_t t = 0; t < nt; t++)
    th.emplace_back([&]() {
      for (size_t b; (b = next.fetch_add(1)) < n_blocks;) f(b);
    });
  for (auto &t : th) t.join();
}
}  // namespace

bool map_genome(const char *file, FastaMap *m, GenomeInfo *info, bool print_stats, bool *fallback, std::string *err) {
  *fallback = false;
  const int fd = open(file, O_RDONLY);
  if (fd < 0) {
    if (print_stats) {
      fprintf(stderr, ":::: Reference stats ::::\n\n");
      fprintf(stderr, "file name : %s\n", file);
      fprintf(stderr, "\n");
    }
    *err = std::string("Cannot open file: ") + file;
    return false;
  }
  struct stat sb;
  if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size <= 0) {
    close(fd);
    *fallback = true;
    return false;
  }
  const size_t size = (size_t)sb.st_size;
  void *map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (map == MAP_FAILED) {
    *fallback = true;
    return false;
  }
  m->map = map;
  m->size = size;
  const uint8_t *data = (const uint8_t *)map;
  if (data[0] != '>') {  // sequence in front of the first header: the reference writes through an unopened stream
    *fallback = true;
    return false;
  }
  // ---- pass 1, on threads: NUL bytes (fgets semantics), the '>' characters, and per block the line feeds + the longest gap
  const char *be = getenv("PBSIM_FASTA_BLOCK");  // test knob: bytes per thread block (seams inside small files)
  const size_t kBlock = be && atol(be) > 0 ? (size_t)atol(be) : (size_t)16 << 20;
  const size_t n_blocks = (size + kBlock - 1) / kBlock;
  struct Block {
    std::vector<size_t> gt;            // positions of '>'
    bool nul = false;
  };
  std::vector<Block> blocks(n_blocks);
  parallel_blocks(n_blocks, [&](size_t b) {
    const uint8_t *p = data + b * kBlock, *e = data + std::min(size, (b + 1) * kBlock);
    blocks[b].nul = memchr(p, 0, (size_t)(e - p)) != nullptr;
    for (const uint8_t *q = p; q < e;) {
      const uint8_t *g = (const uint8_t *)memchr(q, '>', (size_t)(e - q));
      if (!g) break;
      blocks[b].gt.push_back((size_t)(g - data));
      q = g + 1;
    }
  });
  for (const Block &b : blocks)
    if (b.nul) {
      *fallback = true;
      return false;
    }
  // ---- headers: a '>' at the start of an fgets chunk that is not part of a header line's tail (pbsim.cpp:917, 946-951)
  struct Hdr {
    size_t pos, line_end;  // '>' | one past the header line's line feed (or the end of the file)
  };
  std::vector<Hdr> hdrs;
  size_t skip_until = 0;  // inside a header line: its tail is skipped whatever it holds
  for (const Block &b : blocks)
    for (size_t pos : b.gt) {
      if (pos < skip_until) continue;
      bool is_hdr = pos == 0 || data[pos - 1] == '\n';
      if (!is_hdr) {  // inside a sequence line: only where an fgets chunk starts (lines of >= 10239 characters)
        const uint8_t *ls = (const uint8_t *)memrchr(data, '\n', pos);
        const size_t line_start = ls ? (size_t)(ls - data) + 1 : 0;
        is_hdr = (pos - line_start) % (size_t)kChunk == 0;
      }
      if (!is_hdr) continue;
      const uint8_t *nl = (const uint8_t *)memchr(data + pos, '\n', size - pos);
      const size_t line_end = nl ? (size_t)(nl - data) + 1 : size;
      hdrs.push_back(Hdr{pos, line_end});
      skip_until = line_end;
    }
  // ---- pass 2, on threads: line feeds and the longest line of every record's region
  const size_t n_rec = hdrs.size();
  struct Piece {
    size_t rec, a, e;
    int64_t nl = 0, first = -1, last = -1, max_gap = 0;  // line feeds; offsets of the first / last one; longest run between two
  };
  std::vector<Piece> pieces;
  for (size_t r = 0; r < n_rec; r++) {
    const size_t a = hdrs[r].line_end, e = r + 1 < n_rec ? hdrs[r + 1].pos : size;
    for (size_t p = a; p < e || (p == a && a == e); p += kBlock) {
      pieces.push_back(Piece{r, p, std::min(e, p + kBlock)});
      if (a == e) break;
    }
  }
  parallel_blocks(pieces.size(), [&](size_t i) {
    Piece &pc = pieces[i];
    const uint8_t *q = data + pc.a, *e = data + pc.e;
    int64_t prev = -1;
    while (q < e) {
      const uint8_t *nl = (const uint8_t *)memchr(q, '\n', (size_t)(e - q));
      if (!nl) break;
      const int64_t at = (int64_t)(nl - (data + pc.a));
      if (pc.first < 0) pc.first = at;
      else pc.max_gap = std::max(pc.max_gap, at - prev - 1);
      prev = pc.last = at;
      pc.nl++;
      q = nl + 1;
    }
  });
  if (print_stats) {
    fprintf(stderr, ":::: Reference stats ::::\n\n");
    fprintf(stderr, "file name : %s\n", file);
    fprintf(stderr, "\n");
  }
  m->recs.resize(n_rec);
  size_t pi = 0;
  for (size_t r = 0; r < n_rec; r++) {
    FastaRecord &R = m->recs[r];
    const size_t a = hdrs[r].line_end, e = r + 1 < n_rec ? hdrs[r + 1].pos : size;
    const size_t id_end = std::min(hdrs[r].pos + 1 + (size_t)kIdMax, std::min(hdrs[r].pos + (size_t)kChunk, hdrs[r].line_end));
    size_t id_n = id_end - (hdrs[r].pos + 1);
    if (id_n && data[hdrs[r].pos + id_n] == '\n') id_n--;  // (the line feed is not part of the id)
    R.id.assign((const char *)data + hdrs[r].pos + 1, id_n);
    R.lines = data + a;
    R.bytes = (int64_t)(e - a);
    int64_t nl = 0, run = 0, max_line = 0;  // `run`: characters since the last line feed, across pieces
    for (; pi < pieces.size() && pieces[pi].rec == r; pi++) {
      const Piece &pc = pieces[pi];
      const int64_t n = (int64_t)(pc.e - pc.a);
      nl += pc.nl;
      if (pc.first < 0) {
        run += n;
      } else {
        max_line = std::max(max_line, std::max(run + pc.first, pc.max_gap));
        run = n - pc.last - 1;
      }
    }
    max_line = std::max(max_line, run);
    R.len = R.bytes - nl;
    R.max_line = max_line;
    // the reference's checks, in the order its single pass meets them (pbsim.cpp:919-961, 968-971)
    if (r > 0) {
      const FastaRecord &Q = m->recs[r - 1];
      if (Q.len < kRefLenMin) {
        *err = "Reference is too short. Acceptable length >= 100.";
        return false;
      }
      if (print_stats) fprintf(stderr, "ref.%ld (len:%ld) : %s\n", (long)r, (long)Q.len, Q.id.c_str());
      info->len.push_back((long)Q.len);
      info->id.push_back(Q.id);
      info->max_len = std::max(info->max_len, (long)Q.len);
    }
    info->num_seq = (long)r + 1;
    if (info->num_seq > kRefNumMax) {
      *err = "References are too many. Max number of reference is 9999.";
      return false;
    }
    if (R.len > kRefLenMax) {
      *err = "Reference is too long. Acceptable length <= 1000000000.";
      return false;
    }
  }
  {
    const FastaRecord &Q = m->recs[n_rec - 1];
    if (Q.len < kRefLenMin) {
      *err = "Reference is too short. Acceptable length >= 100.";
      return false;
    }
    if (print_stats) fprintf(stderr, "ref.%ld (len:%ld) : %s\n", (long)n_rec, (long)Q.len, Q.id.c_str());
    info->len.push_back((long)Q.len);
    info->id.push_back(Q.id);
    info->max_len = std::max(info->max_len, (long)Q.len);
  }
  if (print_stats) fprintf(stderr, "\n");
  return true;
}

bool write_ref_record(const char *prefix, long num, const FastaRecord &r, std::string *err) {
  char name[4096];
  snprintf(name, sizeof name, "%s_%04ld.ref", prefix, num);
  const int fd = open(name, O_WRONLY | O_CREAT | O_TRUNC, 0666);
  if (fd < 0) {
    *err = std::string("Cannot open output file: ") + name;
    return false;
  }
  auto put = [&](const void *p, size_t n) -> bool {
    const char *q = (const char *)p;
    while (n) {
      const ssize_t k = write(fd, q, std::min<size_t>(n, 1u << 30));
      if (k <= 0) return false;
      q += k;
      n -= (size_t)k;
    }
    return true;
  };
  bool ok = put(">", 1) && put(r.id.data(), r.id.size()) && put("\n", 1);
  if (ok && r.max_line < kChunk) {
    // no line reaches an fgets boundary: the file's own lines are what fprintf(fp_out, "%s\n", line) writes (pbsim.cpp:963)
    ok = put(r.lines, (size_t)r.bytes);
    if (ok && r.bytes > 0 && r.lines[r.bytes - 1] != '\n') ok = put("\n", 1);
  } else if (ok) {
    // chunk by chunk as fgets hands them out: up to 10239 bytes, the line feed (when it is among them) stripped, then "\n"
    std::string buf;
    buf.reserve(1u << 20);
    const uint8_t *p = r.lines, *e = r.lines + r.bytes;
    while (p < e && ok) {
      const uint8_t *nl = (const uint8_t *)memchr(p, '\n', (size_t)(e - p));
      const uint8_t *line_end = nl ? nl + 1 : e;  // one past the line feed
      while (p < line_end) {
        const size_t take = std::min<size_t>((size_t)kChunk, (size_t)(line_end - p));
        size_t n = take;
        if (p[n - 1] == '\n') n--;
        buf.append((const char *)p, n);
        buf.push_back('\n');
        p += take;
      }
      if (buf.size() > (1u << 20) - 2 * (size_t)kBuf) {
        ok = put(buf.data(), buf.size());
        buf.clear();
      }
    }
    if (ok && !buf.empty()) ok = put(buf.data(), buf.size());
  }
  if (close(fd) != 0) ok = false;
  if (!ok) *err = std::string("Cannot write output file: ") + name;
  return ok;
}

bool load_ref_record(const char *prefix, long num, std::string *seq, std::string *err) {
  char name[4096];
  snprintf(name, sizeof name, "%s_%04ld.ref", prefix, num);
  FILE *fp = fopen(name, "r");
  if (!fp) {
    *err = std::string("Cannot open file: ") + name;
    return false;
  }
  std::unique_ptr<char[]> line(new char[kBuf]);
  std::unique_ptr<char[]> in_buf(new char[kIoBuf]);
  setvbuf(fp, in_buf.get(), _IOFBF, kIoBuf);
  seq->clear();
  if (fseek(fp, 0, SEEK_END) == 0) {  // the record is at most as long as its file: one allocation instead of doublings
    const long bytes = ftell(fp);
    if (bytes > 0) seq->reserve((size_t)bytes);
  }
  rewind(fp);
  while (fgets(line.get(), kBuf, fp)) {
    bool nl = chomp(line.get());
    if (line[0] == '>') {
      while (!nl) {
        if (!fgets(line.get(), kBuf, fp)) break;
        nl = chomp(line.get());
      }
    } else {
      seq->append(line.get());
    }
  }
  fclose(fp);
  return true;
}

bool read_transcripts(const char *file, std::vector<Transcript> *out, long *total_exp, std::string *err) {
  FILE *fp = fopen(file, "r");
  if (!fp) {
    *err = std::string("Cannot open file: ") + file;
    return false;
  }
  std::unique_ptr<char[]> line(new char[kBuf]);
  bool first_chunk = true;
  *total_exp = 0;
  Transcript cur;
  while (fgets(line.get(), kBuf, fp)) {
    const bool nl = chomp(line.get());
    if (first_chunk) {
      cur = Transcript();
      char *tok_save = nullptr;
      char *tp = strtok_r(line.get(), "\t", &tok_save);
      const char *a = strtok_r(NULL, "\t", &tok_save);
      const char *b = strtok_r(NULL, "\t", &tok_save);
      const char *s = strtok_r(NULL, "\t", &tok_save);
      if (!tp || !a || !b || !s) {
        fclose(fp);
        *err = "malformed transcript line (expected id<TAB>plus<TAB>minus<TAB>sequence)";
        return false;
      }
      cur.id.assign(tp, strnlen(tp, kIdMax));
      cur.plus = atoi(a);
      cur.minus = atoi(b);
      cur.seq = s;
      *total_exp += cur.plus + cur.minus;  // get_transcript_inf counts per first chunk (pbsim.cpp:1105-1107)
    } else {
      cur.seq.append(line.get());  // continuation chunk of a line longer than BUF_SIZE-1 (pbsim.cpp:4447-4451)
    }
    if (nl) out->push_back(cur);  // a last line without '\n' is never simulated (flg2, pbsim.cpp:4452)
    first_chunk = nl;
  }
  fclose(fp);
  return true;
}

bool read_templates(const char *file, std::vector<Transcript> *out, long *num, long long *len_total, std::string *err) {
  FILE *fp = fopen(file, "r");
  if (!fp) {
    *err = std::string("Cannot open file: ") + file;
    return false;
  }
  std::unique_ptr<char[]> line(new char[kBuf]);
  *num = 0;
  *len_total = 0;
  Transcript cur;
  bool have = false;
  auto flush = [&]() {
    if (have && !cur.seq.empty()) out->push_back(cur);  // `offset != 0` (pbsim.cpp:5057)
    cur.seq.clear();
  };
  while (fgets(line.get(), kBuf, fp)) {
    bool nl = chomp(line.get());
    if (line[0] == '>') {
      flush();
      (*num)++;
      cur.id.assign(line.get() + 1, strnlen(line.get() + 1, kIdMax));
      cur.plus = 1;
      cur.minus = 0;
      have = true;
      while (!nl) {
        if (!fgets(line.get(), kBuf, fp)) break;
        nl = chomp(line.get());
      }
    } else {
      const size_t n = strlen(line.get());
      *len_total += (long long)n;
      cur.seq.append(line.get(), n);
      if (cur.seq.size() > 1000000) {
        fclose(fp);
        *err = "template is too long. Max acceptable length is 1000000.";
        return false;
      }
      have = true;
    }
  }
  flush();
  fclose(fp);
  return true;
}

namespace {
// mean / sd of the filtered strings' lengths and accuracies from the histograms (pbsim.cpp:1285-1330)
bool finish_sample_stats(SampleProfile &s, double acc_total, const std::vector<long> &freq_len, const std::vector<long> &freq_acc,
                         long len_max, std::string *err) {
  if (s.num_filtered < 1) {
    *err = "there is no sample in the valid range of length and accuracy.";
    return false;
  }
  s.len_mean_filtered = (double)s.len_total_filtered / s.num_filtered;
  s.accuracy_mean_filtered = acc_total / s.num_filtered;
  double variance = 0.0;
  for (long i = 0; i <= len_max; i++)
    if (freq_len[(size_t)i] > 0) variance += pow((s.len_mean_filtered - i), 2) * freq_len[(size_t)i];
  s.len_sd_filtered = sqrt(variance / s.num_filtered);
  variance = 0.0;
  for (long i = 0; i <= 100000; i++)
    if (freq_acc[(size_t)i] > 0) variance += pow((s.accuracy_mean_filtered - i * 0.00001), 2) * freq_acc[(size_t)i];
  s.accuracy_sd_filtered = sqrt(variance / s.num_filtered);
  return true;
}

}  // namespace

bool read_sample_fastq_stdio(const char *file, long len_min, long len_max, double acc_min, double acc_max, SampleProfile *out,
                       std::string *err) {
  FILE *fp = fopen(file, "r");
  if (!fp) {
    *err = std::string("Cannot open file: ") + file;
    return false;
  }
  double qprob[94];
  for (int q = 0; q < 94; q++) qprob[q] = pow(10, (double)q / -10);  // pbsim.cpp:546-549
  std::vector<long> freq_len((size_t)len_max + 1, 0), freq_acc(100001, 0);
  std::unique_ptr<char[]> line(new char[kBuf]);
  SampleProfile &s = *out;
  s = SampleProfile();
  s.len_min = s.len_min_filtered = LONG_MAX;
  std::string qc;
  double acc_total = 0.0;
  int line_num = 0;
  // a record ends with its 4th line feed; a line longer than the buffer arrives in chunks without one, and only the
  // chunks of the 4th (quality) line are kept (pbsim.cpp:1216-1283)
  while (fgets(line.get(), kBuf, fp)) {
    const bool nl = chomp(line.get());
    if (!nl) {
      if (line_num == 3) {
        qc += line.get();
        if ((long)qc.size() > 1000000) {
          fclose(fp);
          *err = "fastq is too long. Max acceptable length is 1000000.";
          return false;
        }
      }
      continue;
    }
    if (++line_num < 4) continue;
    const size_t tail = strlen(line.get());
    const long len = (long)(qc.size() + tail);
    if (len > 1000000) {
      fclose(fp);
      *err = "fastq is too long. Max acceptable length is 1000000.";
      return false;
    }
    s.num++;
    s.len_total += len;
    if (s.num > 100000000L) {
      fclose(fp);
      *err = "fastq is too many. Max acceptable number is 100000000.";
      return false;
    }
    s.len_max = std::max(s.len_max, len);
    s.len_min = std::min(s.len_min, len);
    if (len >= len_min && len <= len_max) {
      qc.append(line.get(), tail);
      double prob = 0.0;
      for (long i = 0; i < len; i++) {
        const int q = (int)(unsigned char)qc[(size_t)i] - 33;
        prob += qprob[q < 0 ? 0 : q > 93 ? 93 : q];
      }
      const double accuracy = 1.0 - (prob / len);
      if (accuracy >= acc_min && accuracy <= acc_max) {
        acc_total += accuracy;
        s.num_filtered++;
        s.len_total_filtered += len;
        freq_len[(size_t)len]++;
        freq_acc[(size_t)(int)(accuracy * 100000 + 0.5)]++;
        s.quals.push_back(qc);
        s.len_max_filtered = std::max(s.len_max_filtered, len);
        s.len_min_filtered = std::min(s.len_min_filtered, len);
      }
    }
    line_num = 0;
    qc.clear();
  }
  fclose(fp);
  return finish_sample_stats(s, acc_total, freq_len, freq_acc, len_max, err);
}

// The same parse over the mapped file: line boundaries by memchr, the per-string sums of error probabilities (each an ordered
// f64 sum of its own, pbsim.cpp:1263-1268) on several threads, the statistics and the filter in file order afterwards.  fgets'
// BUF_SIZE chunking is not observable here (a chunk without a line feed never ends a line; only the 4th line's bytes are kept),
// except through NUL bytes -- strlen() ends a chunk there -- so a file that holds one goes through the stdio path, as does
// anything that cannot be mapped.  The reference spends its sample FASTQ's parse on one core (361 MB: 0.75 s here with stdio).
bool read_sample_fastq(const char *file, long len_min, long len_max, double acc_min, double acc_max, SampleProfile *out,
                       std::string *err) {
  const int fd = open(file, O_RDONLY);
  if (fd < 0) {
    *err = std::string("Cannot open file: ") + file;
    return false;
  }
  struct stat sb;
  if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size <= 0) {
    close(fd);
    return read_sample_fastq_stdio(file, len_min, len_max, acc_min, acc_max, out, err);
  }
  const size_t size = (size_t)sb.st_size;
  void *map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (map == MAP_FAILED) return read_sample_fastq_stdio(file, len_min, len_max, acc_min, acc_max, out, err);
  (void)madvise(map, size, MADV_SEQUENTIAL);
  const char *data = (const char *)map;
  struct Unmap {
    void *p;
    size_t n;
    ~Unmap() { munmap(p, n); }
  } unmap{map, size};
  if (memchr(data, 0, size)) return read_sample_fastq_stdio(file, len_min, len_max, acc_min, acc_max, out, err);

  SampleProfile &s = *out;
  s = SampleProfile();
  s.len_min = s.len_min_filtered = LONG_MAX;
  struct Rec {
    const char *p;
    long len;
    double accuracy;
  };
  std::vector<Rec> recs;
  {  // every 4th line that ends with a line feed (a last line without one ends nothing)
    const char *p = data, *end = data + size;
    int line_num = 0;
    while (p < end) {
      const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
      if (!nl) break;
      if (++line_num == 4) {
        const long len = (long)(nl - p);
        if (len > 1000000) {
          *err = "fastq is too long. Max acceptable length is 1000000.";
          return false;
        }
        s.num++;
        s.len_total += len;
        if (s.num > 100000000L) {
          *err = "fastq is too many. Max acceptable number is 100000000.";
          return false;
        }
        s.len_max = std::max(s.len_max, len);
        s.len_min = std::min(s.len_min, len);
        if (len >= len_min && len <= len_max) recs.push_back(Rec{p, len, 0.0});
        line_num = 0;
      }
      p = nl + 1;
    }
    if (line_num == 3 && end - p > 1000000) {  // (the chunks of an unterminated 4th line trip the same test in the stdio path)
      *err = "fastq is too long. Max acceptable length is 1000000.";
      return false;
    }
  }
  double qprob[94];
  for (int q = 0; q < 94; q++) qprob[q] = pow(10, (double)q / -10);  // pbsim.cpp:546-549
  {
    const size_t n = recs.size();
    unsigned T = std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (n < 4096) T = 1;
    std::atomic<size_t> next{0};
    auto work = [&]() {
      for (;;) {
        const size_t b = next.fetch_add(256);
        if (b >= n) return;
        for (size_t r = b; r < std::min(n, b + 256); r++) {
          const unsigned char *q = (const unsigned char *)recs[r].p;
          double prob = 0.0;
          for (long i = 0; i < recs[r].len; i++) {
            const int v = (int)q[i] - 33;
            prob += qprob[v < 0 ? 0 : v > 93 ? 93 : v];
          }
          recs[r].accuracy = 1.0 - (prob / recs[r].len);
        }
      }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < T; t++) pool.emplace_back(work);
    work();
    for (std::thread &t : pool) t.join();
  }
  std::vector<long> freq_len((size_t)len_max + 1, 0), freq_acc(100001, 0);
  double acc_total = 0.0;
  size_t keep = 0;
  for (const Rec &r : recs) keep += (r.accuracy >= acc_min && r.accuracy <= acc_max);
  s.quals.reserve(keep);
  for (const Rec &r : recs) {
    if (!(r.accuracy >= acc_min && r.accuracy <= acc_max)) continue;
    acc_total += r.accuracy;
    s.num_filtered++;
    s.len_total_filtered += r.len;
    freq_len[(size_t)r.len]++;
    freq_acc[(size_t)(int)(r.accuracy * 100000 + 0.5)]++;
    s.quals.emplace_back(r.p, (size_t)r.len);
    s.len_max_filtered = std::max(s.len_max_filtered, r.len);
    s.len_min_filtered = std::min(s.len_min_filtered, r.len);
  }
  return finish_sample_stats(s, acc_total, freq_len, freq_acc, len_max, err);
}

bool write_sample_profile(const std::string &fq, const std::string &stats, const SampleProfile &p, std::string *err) {
  FILE *f = fopen(fq.c_str(), "w"), *g = fopen(stats.c_str(), "w");
  if (!f || !g) {
    if (f) fclose(f);
    if (g) fclose(g);
    *err = "Cannot open sample_profile";
    return false;
  }
  for (const std::string &q : p.quals) fprintf(f, "%s\n", q.c_str());
  fprintf(g, "num\t%ld\nlen_total\t%lld\nlen_min\t%ld\nlen_max\t%ld\n", p.num_filtered, p.len_total_filtered,
          p.len_min_filtered, p.len_max_filtered);
  fprintf(g, "len_mean\t%f\nlen_sd\t%f\naccuracy_mean\t%f\naccuracy_sd\t%f\n", p.len_mean_filtered, p.len_sd_filtered,
          p.accuracy_mean_filtered, p.accuracy_sd_filtered);
  const bool ok = fclose(f) == 0;
  return (fclose(g) == 0) && ok;
}

bool read_sample_profile(const std::string &fq, const std::string &stats, SampleProfile *out, std::string *err) {
  FILE *f = fopen(fq.c_str(), "r"), *g = fopen(stats.c_str(), "r");
  if (!f || !g) {
    if (f) fclose(f);
    if (g) fclose(g);
    *err = "Cannot open sample_profile";
    return false;
  }
  SampleProfile &s = *out;
  s = SampleProfile();
  std::unique_ptr<char[]> line(new char[kBuf]);
  while (fgets(line.get(), kBuf, g)) {  // pbsim.cpp:1185-1210
    chomp(line.get());
    char *tab = strchr(line.get(), '\t');
    if (!tab) continue;
    *tab = '\0';
    const char *item = line.get(), *val = tab + 1;
    if (!strcmp(item, "num")) s.num_filtered = atol(val);
    else if (!strcmp(item, "len_total")) s.len_total_filtered = atol(val);
    else if (!strcmp(item, "len_min")) s.len_min_filtered = atol(val);
    else if (!strcmp(item, "len_max")) s.len_max_filtered = atol(val);
    else if (!strcmp(item, "len_mean")) s.len_mean_filtered = atof(val);
    else if (!strcmp(item, "len_sd")) s.len_sd_filtered = atof(val);
    else if (!strcmp(item, "accuracy_mean")) s.accuracy_mean_filtered = atof(val);
    else if (!strcmp(item, "accuracy_sd")) s.accuracy_sd_filtered = atof(val);
  }
  fclose(g);
  // simulate_by_sample reads the strings with fgets(len_max_filtered + 2) (pbsim.cpp:1733): one string per line
  std::string cur;
  while (fgets(line.get(), kBuf, f)) {
    const bool nl = chomp(line.get());
    cur += line.get();
    if (nl) {
      s.quals.push_back(cur);
      cur.clear();
    }
  }
  if (!cur.empty()) s.quals.push_back(cur);
  fclose(f);
  if (s.quals.empty() || s.len_total_filtered < 1) {
    *err = "sample_profile is empty";
    return false;
  }
  return true;
}

}  // namespace pbsim
