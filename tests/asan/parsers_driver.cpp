#include <stdio.h>
#include <string>
#include "unit_io.h"
using namespace pbsim;
static int compare(const char *file, long lmin, long lmax, double amin, double amax) {
  SampleProfile a, b;
  std::string ea, eb;
  const bool oa = read_sample_fastq(file, lmin, lmax, amin, amax, &a, &ea), ob = read_sample_fastq_stdio(file, lmin, lmax, amin, amax, &b, &eb);
  const bool same = oa == ob && ea == eb &&
                    (!oa || (a.num == b.num && a.len_min == b.len_min && a.len_max == b.len_max && a.len_total == b.len_total &&
                             a.num_filtered == b.num_filtered && a.len_min_filtered == b.len_min_filtered &&
                             a.len_max_filtered == b.len_max_filtered && a.len_total_filtered == b.len_total_filtered &&
                             a.len_mean_filtered == b.len_mean_filtered && a.len_sd_filtered == b.len_sd_filtered &&
                             a.accuracy_mean_filtered == b.accuracy_mean_filtered && a.accuracy_sd_filtered == b.accuracy_sd_filtered &&
                             a.quals == b.quals));
  printf("cmp %s ok=%d kept=%ld same=%d %s\n", file, (int)oa, oa ? a.num_filtered : -1, (int)same, ea.c_str());
  return same ? 0 : 1;
}
// the mapped FASTA pass (map_genome + write_ref_record) against the fgets pass (split_genome + load_ref_record): same records,
// lengths, ids, error text, .ref files byte for byte, and the record = the mapped lines without their line feeds
static std::string slurp(const std::string &path) {
  std::string out;
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return "<missing>";
  char buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
  fclose(f);
  return out;
}
static int compare_fasta(const char *file, const std::string &dir) {
  GenomeInfo ga, gb;
  std::string ea, eb;
  FastaMap fm;
  bool fallback = false;
  FILE *keep = stderr;
  stderr = fopen("/dev/null", "w");  // (the fgets pass prints its report)
  const bool ob = split_genome(file, (dir + "/b").c_str(), &gb, &eb);
  fclose(stderr);
  stderr = keep;
  const bool oa = map_genome(file, &fm, &ga, false, &fallback, &ea);
  if (fallback) {
    printf("cmpfa %s fallback (fgets: ok=%d %s)\n", file, (int)ob, eb.c_str());
    return 0;
  }
  bool same = oa == ob && ea == eb;
  if (oa && same) {
    same = ga.num_seq == gb.num_seq && ga.len == gb.len && ga.id == gb.id && ga.max_len == gb.max_len;
    for (long n = 1; same && n <= ga.num_seq; n++) {
      std::string e2, seq;
      same = write_ref_record((dir + "/a").c_str(), n, fm.recs[(size_t)n - 1], &e2);
      char nm[64];
      snprintf(nm, sizeof nm, "_%04ld.ref", n);
      same = same && slurp(dir + "/a" + nm) == slurp(dir + "/b" + nm);
      same = same && load_ref_record((dir + "/b").c_str(), n, &seq, &e2);
      std::string squeezed;
      const FastaRecord &R = fm.recs[(size_t)n - 1];
      for (int64_t i = 0; i < R.bytes; i++)
        if (R.lines[i] != '\n') squeezed.push_back((char)R.lines[i]);
      same = same && squeezed == seq && (int64_t)seq.size() == R.len;
    }
  }
  printf("cmpfa %s ok=%d recs=%ld same=%d %s\n", file, (int)oa, oa ? ga.num_seq : -1, (int)same, ea.c_str());
  return same ? 0 : 1;
}
int main(int argc, char **argv) {
  std::string e;
  if (argc > 3 && std::string(argv[1]) == "--cmpfa") {
    int bad = 0;
    for (int i = 3; i < argc; i++) bad += compare_fasta(argv[i], argv[2]);
    return bad ? 1 : 0;
  }
  if (argc > 2 && std::string(argv[1]) == "--cmp") {  // the mapped, threaded sample-FASTQ parse against the fgets one
    int bad = 0;
    for (int i = 2; i < argc; i++) bad += compare(argv[i], 100, 1000000, 0.75, 1.0) + compare(argv[i], 30, 5000, 0.5, 0.97);
    return bad ? 1 : 0;
  }
  SampleProfile p;
  if (!read_sample_fastq(argv[1], 100, 1000000, 0.75, 1.0, &p, &e)) { printf("ERR %s\n", e.c_str()); return 1; }
  printf("num %ld filt %ld tot %lld mean %f sd %f acc %f %f\n", p.num, p.num_filtered, p.len_total_filtered, p.len_mean_filtered, p.len_sd_filtered, p.accuracy_mean_filtered, p.accuracy_sd_filtered);
  if (!write_sample_profile(std::string(argv[5]) + "/p.fastq", std::string(argv[5]) + "/p.stats", p, &e)) { printf("ERR %s\n", e.c_str()); return 1; }
  SampleProfile q;
  if (!read_sample_profile(std::string(argv[5]) + "/p.fastq", std::string(argv[5]) + "/p.stats", &q, &e)) { printf("ERR %s\n", e.c_str()); return 1; }
  printf("reread %zu %lld same=%d\n", q.quals.size(), q.len_total_filtered, (int)(q.quals == p.quals));
  GenomeInfo gi;
  if (!split_genome(argv[2], (std::string(argv[5]) + "/g").c_str(), &gi, &e)) { printf("ERR %s\n", e.c_str()); return 1; }
  std::string seq;
  for (long n = 1; n <= gi.num_seq; n++) { if (!load_ref_record((std::string(argv[5]) + "/g").c_str(), n, &seq, &e)) return 1; printf("rec %ld len %zu\n", n, seq.size()); }
  std::vector<Transcript> tr; long tot = 0;
  if (!read_transcripts(argv[3], &tr, &tot, &e)) { printf("ERR %s\n", e.c_str()); return 1; }
  printf("tr %zu exp %ld\n", tr.size(), tot);
  std::vector<Transcript> tp; long num = 0; long long lt = 0;
  if (!read_templates(argv[4], &tp, &num, &lt, &e)) { printf("ERR %s\n", e.c_str()); return 1; }
  printf("templ %zu num %ld len %lld\n", tp.size(), num, lt);
  return 0;
}
