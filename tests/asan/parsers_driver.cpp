#include <stdio.h>
#include <string>
#include "unit_io.h"
using namespace pbsim;
int main(int argc, char **argv) {
  std::string e;
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
