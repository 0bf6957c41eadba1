#include "unit_io.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <memory>

namespace pbsim {

namespace {
constexpr int kBuf = 10240;       // BUF_SIZE pbsim.cpp:20: fgets chunking is observable in the .ref files
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
      cur_len += (long)strlen(line.get());
      if (cur_len > kRefLenMax) {
        *err = "Reference is too long. Acceptable length <= 1000000000.";
        ok = false;
        break;
      }
      fprintf(out, "%s\n", line.get());
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

bool load_ref_record(const char *prefix, long num, std::string *seq, std::string *err) {
  char name[4096];
  snprintf(name, sizeof name, "%s_%04ld.ref", prefix, num);
  FILE *fp = fopen(name, "r");
  if (!fp) {
    *err = std::string("Cannot open file: ") + name;
    return false;
  }
  std::unique_ptr<char[]> line(new char[kBuf]);
  seq->clear();
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
      char *tp = strtok(line.get(), "\t");
      const char *a = strtok(NULL, "\t");
      const char *b = strtok(NULL, "\t");
      const char *s = strtok(NULL, "\t");
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

}  // namespace pbsim
