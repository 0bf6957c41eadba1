#include <stdio.h>
#include <memory>
#include <string>
#include <string.h>
#include "host_tables.h"
using namespace pbsim;
int main(int argc, char **argv) {
  std::string e;
  pbsim_params p; memset(&p, 0, sizeof p);
  p.seed = 1; p.depth = 20; p.len_min = 100; p.len_max = 1000000; p.len_mean = 9000; p.len_sd = 7000; p.accuracy_mean = 0.85;
  p.sub_ratio = 6; p.ins_ratio = 55; p.del_ratio = 39; p.pass_num = 1; p.hp_del_bias = 1; strcpy(p.id_prefix, "S");
  p.strategy = 1; p.method = 2;
  HeaderTables h;
  if (!build_header_tables(p, &h, &e)) { printf("ERR %s\n", e.c_str()); return 1; }
  HpBias b; hp_bias_default(&b);
  for (int i = 1; i < argc; i++) {
    std::string f = argv[i];
    if (f.find("ERRHMM") != std::string::npos) {
      std::unique_ptr<ErrModel> mp(new ErrModel); ErrModel &m = *mp;
      if (!parse_errhmm(f.c_str(), &m, &e)) { printf("ERR %s: %s\n", f.c_str(), e.c_str()); continue; }
      ErrClassTables t;
      for (int wgs = 0; wgs < 2; wgs++) if (!build_err_class_tables(m, h, b, wgs, &t, &e)) { printf("ERR %s\n", e.c_str()); return 1; }
      printf("%s ok stride %u\n", f.c_str(), t.stride);
    } else {
      std::unique_ptr<QsModel> mp(new QsModel); QsModel &m = *mp;
      if (!parse_qshmm(f.c_str(), &m, &e)) { printf("ERR %s: %s\n", f.c_str(), e.c_str()); continue; }
      QsClassTables t;
      if (!build_qs_class_tables(m, h, b, p, &t, &e)) { printf("ERR %s\n", e.c_str()); return 1; }
      printf("%s ok stride %u\n", f.c_str(), t.stride);
    }
  }
  // emission_magic: x % d by multiply-high must be exact for every 31-bit x (checked at the multiples of d around
  // 2^31, at small x and along a stride)
  long bad = 0;
  for (uint32_t d = 2; d <= 1000; d++) {
    uint32_t magic, shift;
    emission_magic(d, &magic, &shift);
    auto chk = [&](uint64_t x) {
      if (x >= (1ull << 31)) return;
      const uint32_t q = (uint32_t)(((uint64_t)(uint32_t)x * magic) >> 32) >> shift;
      if ((uint32_t)x - q * d != (uint32_t)x % d) bad++;
    };
    for (uint64_t x = 0; x < 4096; x++) chk(x);
    for (uint64_t k = 0; k < 3000; k++) { chk((1ull << 31) - 1 - k); }
    const uint64_t top = ((1ull << 31) - 1) / d * d;
    for (int k = -3; k <= 3; k++) { chk(top + k); chk(top / 2 + k); chk(top - d + k); }
    for (uint64_t x = 12345; x < (1ull << 31); x += 7919u * 131u) chk(x);
  }
  printf("emission_magic bad %ld\n", bad);
  return bad != 0;
}
