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
  return 0;
}
