// cli.cpp -- the reference's command line (23 long options, pbsim.cpp:257-282) in front of the C ABI of
// include/pbsim3_amd.h, as a library call: pbsim_cli_main().  This is INTEGRATION.md written out: option parsing and
// validation (pbsim.cpp:286-530, set_sim_param :1451-1688), the stderr report blocks (:5397-5465, :5541-5564), the FASTA
// splitter, the record loop of main() (:666-759) and the gzip/samtools pipes (:708-730).  Everything per read and per
// base happens on the GPU.  Extra options: --device N, --devices a,b,.. (one rank per GPU, see main.cpp), --no-gzip (write
// the text plainly to <prefix>_NNNN.{fq,maf,sam}), --gzip gpu|host, --gzip-threads N, --samtools.  Instead of one `gzip`
// child per file, the .fq.gz/.maf.gz/.bam bytes are compressed on the GPU (deflate.hip: BGZF-framed gzip members, only
// compressed bytes cross PCIe) and written here; --gzip host uses the in-process multi-threaded zlib writer (gzout.h);
// --samtools pipes SAM text into `samtools view -b` like the reference.
//
// wgs (errhmm / qshmm) runs as ONE job over all records (pbsim_job_run): the records are resident in HBM, and on several
// ranks every rank pwrite()s its own byte ranges of the final files.  Rank 0 alone prints and creates files.
#include <fcntl.h>
#include <unistd.h>
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/syscall.h>
#include <sys/time.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pbsim3_amd.h"
#include "gzout.h"
#include "knobs.h"
#include "unit_io.h"

namespace {

struct Cli {
  int set_flg[32] = {0};
  pbsim_params p;
  std::string genome, transcript, templ, prefix = "sd", model, sample, profile_id;
  bool sam_store = false, sam_reuse = false;  // METHOD_SAM_STORE / METHOD_SAM_REUSE (pbsim.cpp:40-41, 1567-1580)
  double accuracy_min = 0.75, accuracy_max = 1.0;
  int device = 0;
  bool no_gzip = false, use_samtools = false;
  bool gzip_on_gpu = true;  // --gzip gpu|host: where the .gz / BGZF members are produced
  int gzip_threads = 0;
};

// the reference's exit(-1).  Other ranks of the process may be inside HIP calls on their own threads: leave without running
// static destructors under them (everything buffered is flushed first)
[[noreturn]] void quit(int status) {
  fflush(NULL);
  _exit(status & 255);
}

[[noreturn]] void die(const char *fmt, const char *a = "", const char *b = "") {
  fprintf(stderr, "ERROR");
  fprintf(stderr, fmt, a, b);
  fprintf(stderr, "\n");
  quit(-1);
}

// an empty BGZF block: the EOF marker of a BAM file (SAMv1 4.1.2), and a valid empty gzip member
const unsigned char kBgzfEof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};

// one output file: plain FILE*, samtools pipe, the in-process parallel gzip, or a plain FILE* that
// receives gzip members already compressed on the GPU (pbsim_set_deflate)
struct Out {
  FILE *fp = nullptr;
  bool pipe = false;
  pbsim::ParallelGz gz;
  bool use_gz = false;
  bool members = false, bam = false;
  size_t wrote = 0;
  bool write(const char *t, size_t n) {
    wrote += n;
    return use_gz ? gz.write(t, n) : fwrite(t, 1, n, fp) == n;
  }
  void close() {
    if (use_gz) {
      if (!gz.close()) die(": write error on a .gz output");
    } else if (members) {
      bool ok = true;
      if (bam || wrote == 0) ok = fwrite(kBgzfEof, 1, sizeof kBgzfEof, fp) == sizeof kBgzfEof;
      if (fclose(fp) != 0 || !ok) die(": write error on a compressed output");
    } else if (pipe) {
      pclose(fp);
    } else if (fp) {
      fclose(fp);
    }
  }
};

void open_sink(const Cli &c, Out *o, const std::string &plain_name, const std::string &target, bool bam) {
  std::string err;
  if (c.no_gzip) {
    o->fp = fopen(plain_name.c_str(), "w");
    if (!o->fp) die(": Cannot open output file: %s", plain_name.c_str());
  } else if (c.gzip_on_gpu && !(bam && c.use_samtools)) {  // members arrive compressed (deflate.hip)
    o->fp = fopen(target.c_str(), "wb");
    o->members = true;
    o->bam = bam;
    if (!o->fp) die(": Cannot open output file: %s", target.c_str());
  } else if (bam && !c.use_samtools) {  // BAM records come from the GPU; BGZF framing here
    o->use_gz = true;
    if (!o->gz.open(target, c.gzip_threads, &err, true)) die(": %s", err.c_str());
  } else if (bam) {  // --samtools: the reference's pipe (pbsim.cpp:715-719)
    const std::string cmd = "samtools view -b -o " + target + " -";
    o->fp = popen(cmd.c_str(), "w");
    o->pipe = true;
    if (!o->fp) die(": Cannot open output file: %s", target.c_str());
  } else {
    o->use_gz = true;
    if (!o->gz.open(target, c.gzip_threads, &err)) die(": %s", err.c_str());
  }
}

bool native_bam(const Cli &c) { return c.p.pass_num > 1 && !c.no_gzip && !c.use_samtools; }

// what main() writes when it opens the samtools pipe (pbsim.cpp:721-722), as SAM text or as the BAM header
// (`record`: genome.num for wgs -- the PU tag carries it; 0 for trans / templ)
void write_read_header(const Cli &c, pbsim_ctx *ctx, Out *o, int64_t record) {
  std::vector<char> h;
  if (native_bam(c)) {
    h.resize((size_t)pbsim_job_bam_header(ctx, record, NULL, 0));
    pbsim_job_bam_header(ctx, record, h.data(), (int64_t)h.size());
  } else {
    h.resize((size_t)pbsim_job_sam_header(ctx, record, NULL, 0) + 1);
    pbsim_job_sam_header(ctx, record, h.data(), (int64_t)h.size());
    h.pop_back();
  }
  if (o->members) {  // the header as a gzip member of its own, through the same encoder as the records
    std::vector<char> z((size_t)pbsim_deflate_bound((int64_t)h.size()) + 64);
    int64_t k = 0;
    if (!pbsim_deflate_buffer(ctx, h.data(), (int64_t)h.size(), z.data(), (int64_t)z.size(), &k))
      die(": %s", pbsim_last_error());
    o->write(z.data(), (size_t)k);
  } else {
    o->write(h.data(), h.size());
  }
}

struct Two {
  Out *r, *m;
};
int cb_read(void *u, const char *t, int64_t k) { return ((Two *)u)->r->write(t, (size_t)k); }
int cb_maf(void *u, const char *t, int64_t k) { return ((Two *)u)->m->write(t, (size_t)k); }

void print_sim_param(const Cli &c) {  // pbsim.cpp:5397-5465
  const pbsim_params &p = c.p;
  fprintf(stderr, ":::: Simulation parameters :::\n\n");
  fprintf(stderr, "strategy : %s\n",
          p.strategy == PBSIM_STRATEGY_WGS ? "wgs" : p.strategy == PBSIM_STRATEGY_TRANS ? "trans" : "templ");
  const bool sampling = p.method == PBSIM_METHOD_SAMPLE;
  if (p.method == PBSIM_METHOD_QS) fprintf(stderr, "method : qshmm\nqshmm : %s\n", c.model.c_str());
  else if (p.method == PBSIM_METHOD_ERR) fprintf(stderr, "method : errhmm\nerrhmm : %s\n", c.model.c_str());
  else fprintf(stderr, "method : sample\n");
  if (p.strategy == PBSIM_STRATEGY_WGS) fprintf(stderr, "genome : %s\n", c.genome.c_str());
  else if (p.strategy == PBSIM_STRATEGY_TRANS) fprintf(stderr, "transcript : %s\n", c.transcript.c_str());
  else fprintf(stderr, "template : %s\n", c.templ.c_str());
  fprintf(stderr, "prefix : %s\n", c.prefix.c_str());
  fprintf(stderr, "id-prefix : %s\n", p.id_prefix);
  if (p.strategy == PBSIM_STRATEGY_WGS) fprintf(stderr, "depth : %lf\n", p.depth);
  if (p.strategy != PBSIM_STRATEGY_TEMPL) {
    if (sampling) {
      fprintf(stderr, "length-mean : (sample FASTQ)\nlength-sd : (sample FASTQ)\n");
    } else {
      fprintf(stderr, "length-mean : %f\n", p.len_mean);
      fprintf(stderr, "length-sd : %f\n", p.len_sd);
    }
    fprintf(stderr, "length-min : %ld\n", (long)p.len_min);
    fprintf(stderr, "length-max : %ld\n", (long)p.len_max);
  }
  if (p.method != PBSIM_METHOD_ERR)
    fprintf(stderr, "difference-ratio : %ld:%ld:%ld\n", (long)p.sub_ratio, (long)p.ins_ratio, (long)p.del_ratio);
  fprintf(stderr, "seed : %d\n", p.seed);
  if (sampling) {  // pbsim.cpp:5453-5460; an option that was not given prints as glibc's "(null)"
    fprintf(stderr, "sample : %s\n", c.set_flg[11] ? c.sample.c_str() : "(null)");
    fprintf(stderr, "sample-profile-id : %s\n", c.set_flg[12] ? c.profile_id.c_str() : "(null)");
    fprintf(stderr, "accuracy-mean : (sample FASTQ)\naccuracy-sd : (sample FASTQ)\n");
    fprintf(stderr, "accuracy-min : %f\n", c.accuracy_min);
    fprintf(stderr, "accuracy-max : %f\n", c.accuracy_max);
  } else {
    fprintf(stderr, "accuracy-mean : %f\n", p.accuracy_mean);
  }
  fprintf(stderr, "pass_num : %d\n", p.pass_num);
  fprintf(stderr, "hp-del-bias : %f\n", p.hp_del_bias);
  fprintf(stderr, "\n");
}

void print_simulation_stats(const Cli &c, const pbsim_stats &s, long unit) {  // pbsim.cpp:5541-5564
  if (c.p.strategy == PBSIM_STRATEGY_WGS) {
    fprintf(stderr, ":::: Simulation stats (ref.%ld) ::::\n\n", unit);
    fprintf(stderr, "read num. : %ld\n", (long)s.res_num);
    fprintf(stderr, "depth : %lf\n", s.res_depth);
  } else {
    fprintf(stderr, ":::: Simulation stats ::::\n\n");
    fprintf(stderr, "read num. : %ld\n", (long)s.res_num);
  }
  fprintf(stderr, "read length mean (SD) : %f (%f)\n", s.res_len_mean, s.res_len_sd);
  fprintf(stderr, "read length min : %ld\n", (long)s.res_len_min);
  fprintf(stderr, "read length max : %ld\n", (long)s.res_len_max);
  fprintf(stderr, "read accuracy mean (SD) : %f (%f)\n", s.res_accuracy_mean, s.res_accuracy_sd);
  fprintf(stderr, "substitution rate. : %f\n", s.res_sub_rate);
  fprintf(stderr, "insertion rate. : %f\n", s.res_ins_rate);
  fprintf(stderr, "deletion rate. : %f\n", s.res_del_rate);
  fprintf(stderr, "\n");
}

void print_help() {
  fprintf(stderr,
          "\nUSAGE: pbsim [options]\n\n"
          "  --prefix --id-prefix --seed\n"
          "  --strategy wgs   --genome FASTA --depth (20.0) --length-min (100) --length-max (1000000)\n"
          "  --strategy trans --transcript TSV (id, plus, minus, sequence)\n"
          "  --method errhmm  --errhmm MODEL   |   --method qshmm --qshmm MODEL --difference-ratio (6:55:39)\n"
          "  --length-mean (9000.0) --length-sd (7000.0) --accuracy-mean (0.85) --pass-num (1) --hp-del-bias (1)\n"
          "  --device N (0)   --devices a,b,.. (one rank per GPU)   --no-gzip (plain .fq/.maf/.sam instead of gzip/samtools pipes)\n"
          "  --gzip gpu|host (gpu)   --gzip-threads N (host)   --samtools (pipe SAM into samtools view -b)\n\n");
}

void check(int ok) {
  if (!ok) {
    // a rank that only learnt of another rank's failure lets that rank report first (they share this process's stderr and exit)
    if (!strncmp(pbsim_last_error(), "another rank", 12)) usleep(300000);
    fprintf(stderr, "ERROR: %s\n", pbsim_last_error());
    quit(-1);
  }
}

// options + set_sim_param (pbsim.cpp:286-530, 1451-1688); exits like the reference on a bad value
void parse_args(int argc, char **argv, Cli &c) {
  pbsim_params_default(&c.p);
  c.p.strategy = 0;
  c.p.method = 0;
  c.p.seed = (unsigned int)time(NULL);  // pbsim.cpp:253
  static struct option long_options[] = {
      {"strategy", 1, NULL, 0},   {"method", 1, NULL, 0},        {"genome", 1, NULL, 0},
      {"transcript", 1, NULL, 0}, {"prefix", 1, NULL, 0},        {"id-prefix", 1, NULL, 0},
      {"depth", 1, NULL, 0},      {"length-min", 1, NULL, 0},    {"length-max", 1, NULL, 0},
      {"difference-ratio", 1, NULL, 0}, {"seed", 1, NULL, 0},    {"sample", 1, NULL, 0},
      {"sample-profile-id", 1, NULL, 0}, {"accuracy-min", 1, NULL, 0}, {"accuracy-max", 1, NULL, 0},
      {"qshmm", 1, NULL, 0},      {"errhmm", 1, NULL, 0},        {"length-mean", 1, NULL, 0},
      {"length-sd", 1, NULL, 0},  {"accuracy-mean", 1, NULL, 0}, {"pass-num", 1, NULL, 0},
      {"template", 1, NULL, 0},   {"hp-del-bias", 1, NULL, 0},   {"device", 1, NULL, 0},
      {"no-gzip", 0, NULL, 0},    {"gzip-threads", 1, NULL, 0}, {"gzip-file", 1, NULL, 0}, {"samtools", 0, NULL, 0},
      {"gzip", 1, NULL, 0},       {"devices", 1, NULL, 0},      {"comm", 1, NULL, 0},          {0, 0, 0, 0}};
  optind = 0;  // glibc: a full re-initialisation (this function runs once per rank)
  int opt, idx = 0;
  while ((opt = getopt_long(argc, argv, "", long_options, &idx)) != -1) {
    if (opt != 0) quit(-1);
    c.set_flg[idx] = 1;
    switch (idx) {
    case 0:
      if (!strncmp(optarg, "wgs", 3)) c.p.strategy = PBSIM_STRATEGY_WGS;
      else if (!strncmp(optarg, "trans", 5)) c.p.strategy = PBSIM_STRATEGY_TRANS;
      else if (!strncmp(optarg, "templ", 5)) c.p.strategy = PBSIM_STRATEGY_TEMPL;
      else die(" (strategy: %s): Acceptable value: wgs, trans, templ.", optarg);
      break;
    case 1:
      if (!strncmp(optarg, "qshmm", 5)) c.p.method = PBSIM_METHOD_QS;
      else if (!strncmp(optarg, "errhmm", 6)) c.p.method = PBSIM_METHOD_ERR;
      else if (!strncmp(optarg, "sample", 6)) c.p.method = PBSIM_METHOD_SAMPLE;
      else die(" (method: %s): Acceptable value: qshmm, errhmm, sample.", optarg);
      break;
    case 2: c.genome = optarg; break;
    case 3: c.transcript = optarg; break;
    case 4: c.prefix = optarg; break;
    case 5:
      if (strlen(optarg) >= sizeof c.p.id_prefix) die(" (id-prefix: %s): too long.", optarg);
      strcpy(c.p.id_prefix, optarg);
      break;
    case 6:
      c.p.depth = atof(optarg);
      if (c.p.depth <= 0.0) die(" (depth: %s): Acceptable range is more than 0.", optarg);
      break;
    case 7:
      c.p.len_min = atoi(optarg);
      if (strlen(optarg) >= 8 || c.p.len_min < 1 || c.p.len_min > 1000000)
        die(" (length-min: %s): Acceptable range is 1-1000000.", optarg);
      break;
    case 8:
      c.p.len_max = atoi(optarg);
      if (strlen(optarg) >= 8 || c.p.len_max < 1 || c.p.len_max > 1000000)
        die(" (length-max: %s): Acceptable range is 1-1000000.", optarg);
      break;
    case 9: {
      std::string buf = optarg;
      char *tok_save = nullptr;
      char *tp = strtok_r(&buf[0], ":", &tok_save);
      for (int num = 0; num < 3; num++) {
        if (!tp) die(" (difference-ratio: %s): Format is sub:ins:del.", optarg);
        long r = atoi(tp);
        if (strlen(tp) >= 5 || r < 0 || r > 1000) die(" (difference-ratio: %s): Acceptable range is 0-1000.", optarg);
        (num == 0 ? c.p.sub_ratio : num == 1 ? c.p.ins_ratio : c.p.del_ratio) = r;
        tp = strtok_r(NULL, ":", &tok_save);
      }
      break;
    }
    case 10: c.p.seed = (unsigned int)atoi(optarg); break;
    case 11: c.sample = optarg; break;
    case 12: c.profile_id = optarg; break;
    case 13:
      c.accuracy_min = atof(optarg);
      if (c.accuracy_min < 0.0 || c.accuracy_min > 1.0) die(" (accuracy-min: %s): Acceptable range is 0.0-1.0.", optarg);
      break;
    case 14:
      c.accuracy_max = atof(optarg);
      if (c.accuracy_max < 0.0 || c.accuracy_max > 1.0) die(" (accuracy-max: %s): Acceptable range is 0.0-1.0.", optarg);
      break;
    case 15: case 16: c.model = optarg; break;
    case 17:
      c.p.len_mean = atof(optarg);
      if (c.p.len_mean < 1 || c.p.len_mean > 1000000) die(" (length-mean: %s): Acceptable range is 1-1000000.", optarg);
      break;
    case 18:
      c.p.len_sd = atof(optarg);
      if (c.p.len_sd < 0 || c.p.len_sd > 1000000) die(" (length-sd: %s): Acceptable range is 0-1000000.", optarg);
      break;
    case 19:
      c.p.accuracy_mean = atof(optarg);
      if (c.p.accuracy_mean < 0.0 || c.p.accuracy_mean > 1.0)
        die(" (accuracy-mean: %s): Acceptable range is 0.0-1.0.", optarg);
      break;
    case 20:
      c.p.pass_num = atoi(optarg);
      if (c.p.pass_num < 1) die(" (pass_num: %s): Acceptable range is more than 1.", optarg);
      break;
    case 21: c.templ = optarg; break;
    case 22:
      c.p.hp_del_bias = atof(optarg);
      if (strlen(optarg) >= 8 || c.p.hp_del_bias < 1 || c.p.hp_del_bias > 10)
        die(" (hp-del-bias: %s): Acceptable range is 1-10.", optarg);
      break;
    case 23: c.device = atoi(optarg); break;
    case 24: c.no_gzip = true; break;
    case 25: c.gzip_threads = atoi(optarg); break;
    case 27: c.use_samtools = true; break;
    case 28:
      if (!strcmp(optarg, "gpu")) c.gzip_on_gpu = true;
      else if (!strcmp(optarg, "host")) c.gzip_on_gpu = false;
      else die(" (gzip: %s): gpu or host.", optarg);
      break;
    case 29: case 30: break;  // --devices / --comm: main.cpp (one rank per GPU); a rank itself runs on `device`
    case 26: {  // utility/self-test: gzip FILE -> FILE.gz with the parallel writer, nothing else
      pbsim::ParallelGz gz;
      std::string e;
      const int th = c.gzip_threads > 0 ? c.gzip_threads : (int)std::max(1u, std::thread::hardware_concurrency());
      FILE *in = fopen(optarg, "rb");
      if (!in || !gz.open(std::string(optarg) + ".gz", th, &e)) die(": Cannot open file: %s", optarg);
      std::vector<char> buf(1 << 16);
      size_t k;
      while ((k = fread(buf.data(), 1, buf.size(), in)) > 0) gz.write(buf.data(), k);
      fclose(in);
      quit(gz.close() ? 0 : 255);
    }
    default: break;
    }
  }
  if (argc == 1) {
    print_help();
    quit(-1);
  }
  // ---- set_sim_param (pbsim.cpp:1451-1688)
  if (!c.set_flg[0] || !c.set_flg[1]) die(": --strategy and --method must be set.");
  if (c.p.strategy == PBSIM_STRATEGY_WGS && !c.set_flg[2]) die(": for --strategy wgs, --genome must be set.");
  if (c.p.strategy == PBSIM_STRATEGY_TRANS && !c.set_flg[3]) die(": for --strategy trans, --transcript must be set.");
  if (c.p.strategy == PBSIM_STRATEGY_TEMPL && !c.set_flg[21]) die(": for --strategy templ, --template must be set.");
  const bool sampling = c.p.method == PBSIM_METHOD_SAMPLE;
  const std::string profile_fq = "sample_profile_" + c.profile_id + ".fastq",
                    profile_stats = "sample_profile_" + c.profile_id + ".stats";
  if (sampling) {  // pbsim.cpp:1461-1464, 1567-1634
    if (c.p.strategy != PBSIM_STRATEGY_WGS) die(": sampling-based simulation is possible only for wgs strategy.");
    if (c.set_flg[11]) c.sam_store = c.set_flg[12] != 0;
    else if (c.set_flg[12]) c.sam_reuse = true;
    else die(": for --method sample, --sample (and/or --sample-profile-id) must be set.");
    for (const std::string &f : {profile_fq, profile_stats}) {
      FILE *fp = (c.sam_store || c.sam_reuse) ? fopen(f.c_str(), "r") : NULL;
      if (fp) fclose(fp);
      if (c.sam_store && fp) die(": %s exists.", f.c_str());
      if (c.sam_reuse && !fp) die(": %s does not exist.", f.c_str());
    }
  }
  c.accuracy_min = c.set_flg[13] ? (int)(c.accuracy_min * 100) * 0.01 : 0.75;
  c.accuracy_max = c.set_flg[14] ? (int)(c.accuracy_max * 100) * 0.01 : 1.0;
  if (c.p.method == PBSIM_METHOD_QS && !c.set_flg[15]) die(": for --method qshmm, --qshmm must be set.");
  if (c.p.method == PBSIM_METHOD_ERR && !c.set_flg[16]) die(": for --method errhmm, --errhmm must be set.");
  if (c.set_flg[19]) c.p.accuracy_mean = (int)(c.p.accuracy_mean * 100) * 0.01;
  if (c.p.len_min > c.p.len_max) {
    fprintf(stderr, "ERROR: length min(%ld) is greater than max(%ld).\n", (long)c.p.len_min, (long)c.p.len_max);
    quit(-1);
  }
  if (c.p.pass_num > 1 && sampling) die(": sampling-based simulation supports only single-pass.");
  if (c.gzip_threads < 1) c.gzip_threads = (int)std::max(1u, std::thread::hardware_concurrency());
}

// ---- output files ------------------------------------------------------------------------------------------------------
// Writing behind the job (round 4).  The job hands its sinks ~45 GB/s of members from two delivery threads, and a file takes
// what one writer can push into it: 4-6 GB/s per file on /dev/shm however many threads write (one inode; tools/closed_ab/shm_write_test.c:
// pwrite from 16 threads into ONE file 3.6 GB/s, into 16 files 66 GB/s).  So a sink callback only copies its piece into a
// buffer of the pool below -- on helper threads, a piece of 80 MB is gone in ~2 ms -- and returns; one writer thread per file
// pwrite()s the buffers in the order they came, and the job keeps k records going side by side (pbsim_job_set_interleave), so
// that 2k files are being written at any time.  The pool bounds the memory (PBSIM_CLI_WRITE_BUFFER_MB, default 8192): a
// callback waits for a buffer when the files fall that far behind.
struct WritePool {
  static constexpr size_t kChunk = 32u << 20, kSlice = 4u << 20;
  size_t max_chunks = 256, made = 0;
  double waited_s = 0, copy_s = 0;   // callbacks waiting for a buffer (the files are behind) | copying (PBSIM_TRACE)
  int64_t copied = 0;
  std::vector<char *> idle;
  std::mutex mu;
  std::condition_variable cv;
  // copy helpers
  struct Task {
    char *dst;
    const char *src;
    size_t n;
    std::atomic<int> *left;
  };
  std::vector<std::thread> helpers;
  std::deque<Task> tasks;
  std::mutex tmu;
  std::condition_variable tcv, dcv;
  bool stop = false;
  WritePool() {
    const char *mb = pbsim::exp_env("PBSIM_CLI_WRITE_BUFFER_MB");
    const size_t bytes = (size_t)(mb && atoll(mb) > 0 ? atoll(mb) : 8192) << 20;
    max_chunks = std::max<size_t>(4, bytes / kChunk);
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    for (unsigned i = 0; i < std::min(8u, std::max(1u, hw / 4)); i++) helpers.emplace_back([this]() { run(); });
  }
  ~WritePool() {
    {
      std::lock_guard<std::mutex> lk(tmu);
      stop = true;
    }
    tcv.notify_all();
    for (auto &t : helpers) t.join();
    const char *leave = getenv("PBSIM_CLI_LEAVE_CONTEXT");  // (the process ends behind the job: its mappings go with it)
    if (!(leave && *leave == '1'))
      for (char *p : idle) munmap(p, kChunk);
  }
  bool step(std::unique_lock<std::mutex> &lk) {  // runs one queued slice; the lock is held on entry and on return
    if (tasks.empty()) return false;
    Task t = tasks.front();
    tasks.pop_front();
    lk.unlock();
    memcpy(t.dst, t.src, t.n);
    const bool last = t.left->fetch_sub(1) == 1;
    lk.lock();
    if (last) dcv.notify_all();
    return true;
  }
  void run() {
    std::unique_lock<std::mutex> lk(tmu);
    for (;;) {
      tcv.wait(lk, [&] { return stop || !tasks.empty(); });
      if (stop && tasks.empty()) return;
      step(lk);
    }
  }
  void copy(char *dst, const char *src, size_t n) {  // returns when the bytes are in `dst`; the caller copies along
    const auto t0 = std::chrono::steady_clock::now();
    std::atomic<int> left{(int)((n + kSlice - 1) / kSlice)};
    std::unique_lock<std::mutex> lk(tmu);
    for (size_t a = 0; a < n; a += kSlice) tasks.push_back(Task{dst + a, src + a, std::min(kSlice, n - a), &left});
    tcv.notify_all();
    while (left.load() > 0)
      if (!step(lk)) dcv.wait(lk, [&] { return left.load() == 0 || !tasks.empty(); });
    copy_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    copied += (int64_t)n;
  }
  char *get() {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      if (!idle.empty()) {
        char *p = idle.back();
        idle.pop_back();
        return p;
      }
      if (made < max_chunks) {
        made++;
        lk.unlock();
        // (huge pages where the kernel grants them: 8 GB of buffers are two million first-touch faults in 4 KB pages)
        void *p = mmap(nullptr, kChunk, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p == MAP_FAILED) die(": Cannot allocate memory.");
        (void)madvise(p, kChunk, MADV_HUGEPAGE);
        return (char *)p;
      }
      const auto t0 = std::chrono::steady_clock::now();
      cv.wait(lk);
      waited_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
  }
  void put(char *p) {
    {
      std::lock_guard<std::mutex> lk(mu);
      idle.push_back(p);
    }
    cv.notify_one();
  }
};

// A positional file: every rank opens the same path and pwrite()s its own byte ranges (rank 0 creates / truncates it first).
struct PosFile {
  int fd = -1;
  std::string path;
  int64_t base = 0;  // bytes rank 0 puts in front of the record stream (SAM / BAM header)
  // asynchronous mode (set_pool): write_at copies and queues; a thread of the file's own writes
  WritePool *pool = nullptr;
  struct Piece {
    char *buf;
    size_t n;
    int64_t off;
  };
  std::deque<Piece> q;
  std::mutex mu;
  std::condition_variable cv;
  std::thread th;
  bool started = false, finishing = false;
  std::atomic<bool> failed{false};  // set by the file's thread, read by the callbacks
  void writer() {
    // The rank's threads are bound to its GPU's NUMA node (pbsim_bind_host_to_device): right for the delivery threads and their
    // pinned staging, wrong for the threads that fill the page cache -- with every file's pages coming out of one node's
    // allocator eight writers moved 18 GB/s together; unbound they take pages (and memory bandwidth) from both sockets.
    // PBSIM_CLI_WRITERS_BOUND=1 keeps them bound (A/B).
    static const bool keep_bound = pbsim::exp_env("PBSIM_CLI_WRITERS_BOUND") && atoi(pbsim::exp_env("PBSIM_CLI_WRITERS_BOUND")) == 1;
    if (!keep_bound) {
      cpu_set_t all;
      CPU_ZERO(&all);
      for (int i = 0; i < CPU_SETSIZE; i++) CPU_SET(i, &all);
      (void)sched_setaffinity(0, sizeof all, &all);
#ifdef SYS_set_mempolicy
      (void)syscall(SYS_set_mempolicy, 0 /* MPOL_DEFAULT */, nullptr, 0);
#endif
    }
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv.wait(lk, [&] { return finishing || !q.empty(); });
      if (q.empty()) {  // finished: the descriptor goes back now, not at the end of the job (a genome of thousands of records)
        if (fd >= 0 && ::close(fd) != 0) failed = true;
        fd = -1;
        return;
      }
      const Piece p = q.front();
      q.pop_front();
      lk.unlock();
      if (!failed && !write_now(p.buf, (int64_t)p.n, p.off)) failed = true;  // (keeps draining: the buffers go back to the pool)
      pool->put(p.buf);
      lk.lock();
    }
  }
  bool write_now(const char *t, int64_t n, int64_t off) {
    while (n > 0) {
      const ssize_t k = ::pwrite(fd, t, (size_t)n, (off_t)off);
      if (k <= 0) return false;
      t += k;
      n -= k;
      off += k;
    }
    return true;
  }
  // Rank 0 creates / truncates the file (and writes `head`, the SAM / BAM header) before any rank writes, and lets go of the
  // descriptor: every rank opens the path when its first byte range arrives (write_at) and closes it when the record is done, so
  // a genome of thousands of records keeps a handful of files open -- the records of the interleave window -- instead of two
  // per record from the start (ulimit -n is 1024 on most boxes, REF_SEQ_NUM_MAX 9999).
  void create(const std::string &p, const char *head, int64_t head_bytes) {
    path = p;
    fd = ::open(p.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) die(": Cannot open output file: %s", p.c_str());
    if (head_bytes > 0 && !write_now(head, head_bytes, 0)) die(": write error on %s", p.c_str());
    if (::close(fd) != 0) die(": write error on %s", p.c_str());
    fd = -1;
  }
  void attach(const std::string &p) { path = p; }
  bool write_at(const char *t, int64_t n, int64_t off) {
    if (fd < 0) {
      fd = ::open(path.c_str(), O_WRONLY);
      if (fd < 0) {
        fprintf(stderr, "ERROR: Cannot open output file: %s\n", path.c_str());
        return false;
      }
    }
    if (!pool) return write_now(t, n, off);
    while (n > 0) {
      const size_t k = (size_t)std::min<int64_t>(n, (int64_t)WritePool::kChunk);
      char *buf = pool->get();
      pool->copy(buf, t, k);
      {
        std::lock_guard<std::mutex> lk(mu);
        if (failed) {
          pool->put(buf);
          return false;
        }
        q.push_back(Piece{buf, k, off});
        if (!started) {
          started = true;
          th = std::thread([this]() { writer(); });
        }
      }
      cv.notify_one();
      t += k;
      n -= (int64_t)k;
      off += (int64_t)k;
    }
    return true;
  }
  // no more writes will come (asynchronous mode: the file's thread writes what is queued and ends); close_checked joins it
  void finish() {
    {
      std::lock_guard<std::mutex> lk(mu);
      finishing = true;
    }
    cv.notify_all();
  }
  void close_checked() {
    if (started) {
      finish();
      th.join();
      started = false;
      if (failed) die(": write error on %s", path.c_str());
    }
    if (fd >= 0 && ::close(fd) != 0) die(": write error on %s", path.c_str());
    fd = -1;
  }
};

// what a rank needs per output stream of one wgs record
struct Stream {
  PosFile pos;      // --no-gzip and --gzip gpu: positional
  Out seq;          // --gzip host, --samtools: one sequential consumer (a single rank only)
  bool positional = false;
  int64_t expect = 0;  // sequential: the next offset
  bool write(const char *t, int64_t n, int64_t off) {
    if (positional) return pos.write_at(t, n, pos.base + off);
    if (off != expect) return false;  // a sequential consumer on a single rank: offsets simply run up
    expect += n;
    return seq.write(t, (size_t)n);
  }
};

struct RecFiles {
  Stream read, maf;
};

struct JobFiles {
  const Cli *cli = nullptr;
  pbsim_ctx *ctx = nullptr;
  const pbsim_comm *comm = nullptr;
  int64_t first = 1;
  std::vector<std::unique_ptr<RecFiles>> recs;
  RecFiles &of(int64_t record) { return *recs[(size_t)(record - first)]; }
  std::unique_ptr<WritePool> pool;  // positional files are written behind the job (WritePool above) unless PBSIM_CLI_SYNC_WRITES=1
  void use_async_writes() {
    const char *sw = pbsim::exp_env("PBSIM_CLI_SYNC_WRITES");
    if (!(sw && *sw == '1')) pool.reset(new WritePool);
  }
  // after pbsim_job_run: every file's queue written, every file closed (write errors end the process here)
  void close_all() {
    for (auto &r : recs)
      for (Stream *s : {&r->read, &r->maf})
        if (s->positional) s->pos.close_checked();
  }
};

bool positional_mode(const Cli &c, bool bam) { return c.no_gzip || (c.gzip_on_gpu && !(bam && c.use_samtools)); }

void barrier(const pbsim_comm *comm) {
  if (!comm || comm->world <= 1) return;
  int64_t x = 0;
  if (!comm->all_reduce_i64(comm->user, &x, 1, PBSIM_OP_SUM)) die(": communicator failed");
}

std::string read_name(const Cli &c, long n) {
  char name[4096];
  snprintf(name, sizeof name, "%s_%04ld", c.prefix.c_str(), n);
  if (c.p.pass_num == 1) return std::string(name) + (c.no_gzip ? ".fq" : ".fq.gz");
  return std::string(name) + (c.no_gzip ? ".sam" : ".bam");
}
std::string maf_name(const Cli &c, long n) {
  char name[4096];
  snprintf(name, sizeof name, "%s_%04ld.maf", c.prefix.c_str(), n);
  return std::string(name) + (c.no_gzip ? "" : ".gz");
}

// header bytes in front of the read stream (pbsim.cpp:721-722): SAM text, or the BAM header as a gzip member of its own
std::vector<char> read_header_bytes(const Cli &c, pbsim_ctx *ctx, int64_t record) {
  std::vector<char> h;
  if (c.p.pass_num == 1) return h;
  if (native_bam(c)) {
    h.resize((size_t)pbsim_job_bam_header(ctx, record, NULL, 0));
    pbsim_job_bam_header(ctx, record, h.data(), (int64_t)h.size());
  } else {
    h.resize((size_t)pbsim_job_sam_header(ctx, record, NULL, 0) + 1);
    pbsim_job_sam_header(ctx, record, h.data(), (int64_t)h.size());
    h.pop_back();
  }
  if (!c.no_gzip && c.gzip_on_gpu && !c.use_samtools) {
    std::vector<char> z((size_t)pbsim_deflate_bound((int64_t)h.size()) + 64);
    int64_t k = 0;
    if (!pbsim_deflate_buffer(ctx, h.data(), (int64_t)h.size(), z.data(), (int64_t)z.size(), &k)) die(": %s", pbsim_last_error());
    z.resize((size_t)k);
    return z;
  }
  return h;
}

// opens the two outputs of record n on this rank (rank 0 creates, the others attach after a barrier -- see open_job_files)
void open_record(JobFiles &jf, long n, bool creator) {
  const Cli &c = *jf.cli;
  RecFiles &rf = jf.of(n);
  const bool bam = c.p.pass_num > 1;
  rf.read.positional = positional_mode(c, bam);
  rf.maf.positional = positional_mode(c, false);
  rf.read.pos.pool = rf.maf.pos.pool = jf.pool.get();
  if (rf.read.positional) {
    const std::vector<char> h = read_header_bytes(c, jf.ctx, n);
    rf.read.pos.base = (int64_t)h.size();
    if (creator) rf.read.pos.create(read_name(c, n), h.data(), (int64_t)h.size());
    else rf.read.pos.attach(read_name(c, n));
  } else if (creator) {  // sequential consumers exist on a single rank only
    char name[4096];
    snprintf(name, sizeof name, "%s_%04ld", c.prefix.c_str(), n);
    if (bam) {
      open_sink(c, &rf.read.seq, std::string(name) + ".sam", std::string(name) + ".bam", true);
      write_read_header(c, jf.ctx, &rf.read.seq, n);
    } else {
      open_sink(c, &rf.read.seq, std::string(name) + ".fq", std::string(name) + ".fq.gz", false);
    }
  }
  if (rf.maf.positional) {
    if (creator) rf.maf.pos.create(maf_name(c, n), nullptr, 0);
    else rf.maf.pos.attach(maf_name(c, n));
  } else if (creator) {
    char name[4096];
    snprintf(name, sizeof name, "%s_%04ld.maf", c.prefix.c_str(), n);
    open_sink(c, &rf.maf.seq, name, std::string(name) + ".gz", false);
  }
}

int job_read(void *u, int64_t record, const char *t, int64_t k, int64_t off) { return ((JobFiles *)u)->of(record).read.write(t, k, off); }
int job_maf(void *u, int64_t record, const char *t, int64_t k, int64_t off) { return ((JobFiles *)u)->of(record).maf.write(t, k, off); }
int job_done(void *u, int64_t record, const pbsim_stats *st, int64_t read_bytes, int64_t maf_bytes) {
  JobFiles &jf = *(JobFiles *)u;
  const Cli &c = *jf.cli;
  RecFiles &rf = jf.of(record);
  const bool rank0 = !jf.comm || jf.comm->rank == 0;
  if (rank0) print_simulation_stats(c, *st, (long)record);
  for (int which = 0; which < 2; which++) {
    Stream &s = which == 0 ? rf.read : rf.maf;
    const int64_t bytes = which == 0 ? read_bytes : maf_bytes;
    if (s.positional) {
      // compressed streams end with the BGZF EOF marker when they are BAM, or when they would otherwise be empty files
      const bool members = !c.no_gzip;
      const bool bam = which == 0 && c.p.pass_num > 1;
      if (rank0 && members && (bam || s.pos.base + bytes == 0) &&
          !s.pos.write_at((const char *)kBgzfEof, sizeof kBgzfEof, s.pos.base + bytes))
        return 0;
      if (s.pos.pool) s.pos.finish();  // (its thread writes what is queued; JobFiles::close_all joins it behind the job)
      else s.pos.close_checked();
    } else if (rank0) {
      s.seq.close();
    }
  }
  return 1;
}

}  // namespace

extern "C" int pbsim_cli_main(int argc, char **argv, const pbsim_comm *comm, int device) {
  struct timeval tv0;
  gettimeofday(&tv0, NULL);
  Cli c;
  {
    // getopt_long keeps its state in globals: the ranks of one process (main.cpp: one thread per GPU) parse one after the other
    static std::mutex parse_mu;
    std::lock_guard<std::mutex> lock(parse_mu);
    parse_args(argc, argv, c);
  }
  if (device >= 0) c.device = device;
  // this rank's thread, the delivery threads it starts and their pinned staging: on the NUMA node of its GPU
  (void)pbsim_bind_host_to_device(c.device, NULL, 0);
  const int world = comm ? comm->world : 1, rank = comm ? comm->rank : 0;
  const bool rank0 = rank == 0;
  const bool sampling = c.p.method == PBSIM_METHOD_SAMPLE;
  // PBSIM_TRACE: where the process's wall time goes, phase by phase (stderr, not part of the report)
  const bool trace = getenv("PBSIM_TRACE") != nullptr;
  struct timespec ts0;
  clock_gettime(CLOCK_MONOTONIC, &ts0);
  auto phase = [&](const char *what) {
    if (!trace || !rank0) return;
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    fprintf(stderr, "[pbsim cli] %8.1f ms  %s\n", (t.tv_sec - ts0.tv_sec) * 1e3 + (t.tv_nsec - ts0.tv_nsec) * 1e-6, what);
  };
  const std::string profile_fq = "sample_profile_" + c.profile_id + ".fastq",
                    profile_stats = "sample_profile_" + c.profile_id + ".stats";
  if (world > 1) {
    if (!c.no_gzip && (!c.gzip_on_gpu || c.use_samtools))
      die(": several GPUs write their own byte ranges of the outputs: use --gzip gpu (default) or --no-gzip.");
  }
  if (rank0) print_sim_param(c);

  pbsim::SampleProfile prof;
  if (sampling) {  // pbsim.cpp:580-617: read (or re-read) the profile, print its statistics
    std::string e;
    if (c.sam_reuse) {
      if (!pbsim::read_sample_profile(profile_fq, profile_stats, &prof, &e)) die(": %s", e.c_str());
    } else {
      if (!pbsim::read_sample_fastq(c.sample.c_str(), (long)c.p.len_min, (long)c.p.len_max, c.accuracy_min,
                                    c.accuracy_max, &prof, &e))
        die(": %s", e.c_str());
      if (c.sam_store && rank0 && !pbsim::write_sample_profile(profile_fq, profile_stats, prof, &e)) die(": %s", e.c_str());
    }
    if (rank0) {
    fprintf(stderr, ":::: sample reads stats ::::\n\n");  // print_sample_stats, pbsim.cpp:1336-1360
    if (c.sam_reuse) {
      fprintf(stderr, "file name : %s\n", profile_fq.c_str());
    } else {
      fprintf(stderr, "file name : %s\n", c.sample.c_str());
      fprintf(stderr, "\n:: all reads ::\n");
      fprintf(stderr, "read num. : %ld\n", prof.num);
      fprintf(stderr, "read total length : %lld\n", prof.len_total);
      fprintf(stderr, "read min length : %ld\n", prof.len_min);
      fprintf(stderr, "read max length : %ld\n", prof.len_max);
    }
    fprintf(stderr, "\n:: filtered reads ::\n");
    fprintf(stderr, "read num. : %ld\n", prof.num_filtered);
    fprintf(stderr, "read total length : %lld\n", prof.len_total_filtered);
    fprintf(stderr, "read min length : %ld\n", prof.len_min_filtered);
    fprintf(stderr, "read max length : %ld\n", prof.len_max_filtered);
    fprintf(stderr, "read length mean (SD) : %f (%f)\n", prof.len_mean_filtered, prof.len_sd_filtered);
    fprintf(stderr, "read accuracy mean (SD) : %f (%f)\n", prof.accuracy_mean_filtered, prof.accuracy_sd_filtered);
    fprintf(stderr, "\n");
    }
  }

  phase("parameters printed, sample profile parsed");
  pbsim_ctx *ctx = pbsim_create(&c.p, c.device);
  if (!ctx) check(0);
  phase("context created (HIP initialised)");
  if (sampling) {
    std::vector<const uint8_t *> qp;
    std::vector<int64_t> ql;
    for (const std::string &q : prof.quals) {
      qp.push_back((const uint8_t *)q.data());
      ql.push_back((int64_t)q.size());
    }
    check(pbsim_set_sample_profile(ctx, (int64_t)qp.size(), qp.data(), ql.data()));
    phase("sample profile on the GPU");
  } else {
    check(c.p.method == PBSIM_METHOD_ERR ? pbsim_load_errhmm(ctx, c.model.c_str()) : pbsim_load_qshmm(ctx, c.model.c_str()));
  }
  if (native_bam(c)) check(pbsim_set_bam_output(ctx, 1));
  if (!c.no_gzip && c.gzip_on_gpu) {  // bit 0: read sink, bit 1: MAF sink; a samtools pipe still wants SAM text
    // bit 2: the read file and the MAF file are written by two host threads (each Out is touched by one of them only)
    const char *one = getenv("PBSIM_CLI_ONE_WRITER");  // measurement knob: both files from the calling thread
    check(pbsim_set_deflate(ctx, (c.p.pass_num > 1 && c.use_samtools) ? 2 : (one && *one == '1') ? 3 : 7));
  }
  std::string err;
  char name[4096];

  if (c.p.strategy == PBSIM_STRATEGY_WGS && !sampling && !(c.p.pass_num > 1 && c.use_samtools && !c.no_gzip && c.gzip_on_gpu)) {
    // ---- pbsim.cpp:667-759 as one job: rank 0 splits the FASTA into <prefix>_NNNN.ref (and prints the reference stats),
    // every rank loads the records (C1: broadcast GPU to GPU when the communicator can, else from the .ref files),
    // groups of records that fit the GPU's memory run as one pipeline each
    pbsim::GenomeInfo gi;
    // The FASTA mapped and scanned on threads, the records uploaded as their lines and squeezed on the GPU, the .ref files
    // written beside the simulation (round 4; the fgets pass of the reference stays as the fallback -- pipes, NUL bytes, a
    // genome of several jobs -- and as PBSIM_FASTA_LOADER=stdio for the tests that compare the two).
    pbsim::FastaMap fm;
    bool mapped = false;
    {
      const char *fl = getenv("PBSIM_FASTA_LOADER");
      if (!(fl && !strcmp(fl, "stdio"))) {
        // (anything but a clean genome that one job holds goes through the fgets pass below, which then prints the reference
        // stats and the reference's error messages itself)
        bool fallback = false;
        pbsim::GenomeInfo gm;
        std::string e2;
        if (pbsim::map_genome(c.genome.c_str(), &fm, &gm, false, &fallback, &e2)) {
          double sum = 0;
          for (long l : gm.len) sum += (double)l;
          const char *gbs = getenv("PBSIM_JOB_REF_GB");
          if (sum <= (gbs && atof(gbs) > 0 ? atof(gbs) : 64.0) * (double)(1LL << 30) / 2.0) {  // one job holds the genome
            mapped = true;
            gi = gm;
          }
        }
      }
      // Every rank maps the file by itself, and a rank that cannot (no shared file system, ENOMEM) would take the fgets branch
      // with its own collectives while the others sit in the job's: the ranks agree first -- all map, or none does
      if (comm && comm->world > 1) {
        int64_t all = mapped ? 1 : 0;
        if (!comm->all_reduce_i64(comm->user, &all, 1, PBSIM_OP_MIN)) die(": communicator failed");
        if (!all && mapped) {
          mapped = false;
          gi = pbsim::GenomeInfo();  // (the mapping itself goes with `fm` at the end of the scope)
        }
      }
      if (mapped && rank0) {  // get_genome_inf's report (pbsim.cpp:902-904, 925, 972, 979)
        fprintf(stderr, ":::: Reference stats ::::\n\n");
        fprintf(stderr, "file name : %s\n", c.genome.c_str());
        fprintf(stderr, "\n");
        for (long n = 1; n <= gi.num_seq; n++)
          fprintf(stderr, "ref.%ld (len:%ld) : %s\n", n, gi.len[(size_t)n - 1], gi.id[(size_t)n - 1].c_str());
        fprintf(stderr, "\n");
      }
    }
    std::vector<std::thread> ref_writers;
    std::atomic<bool> ref_failed{false};
    std::string ref_err;
    std::mutex ref_mu;
    if (mapped) {
      phase("genome mapped and scanned");
      check(pbsim_job_begin(ctx, 1));
      // main() reads its records one at a time (pbsim.cpp:666-759): here record 1 goes up now, the job is told what else to
      // expect (pbsim_job_expect) and a feeder thread uploads records 2.. from the mapped file WHILE the job runs -- the first
      // read is walked after one record's upload and preparation, not the genome's (3 Gbp: ~130 ms of the 3.3 s).
      // (--hp-del-bias != 1: the job itself waits for every record, pbsim.cpp:677-696.)
      {
        std::vector<int64_t> lens;
        for (long n = 1; n <= gi.num_seq; n++) lens.push_back((int64_t)fm.recs[(size_t)n - 1].len);
        check(pbsim_job_expect(ctx, (int64_t)lens.size(), lens.data()));
        check(pbsim_job_add_record_lines(ctx, fm.recs[0].lines, fm.recs[0].bytes, fm.recs[0].len));
      }
      std::thread feeder([&]() {
        for (long n = 2; n <= gi.num_seq; n++) {
          const pbsim::FastaRecord &R = fm.recs[(size_t)n - 1];
          if (!pbsim_job_add_record_lines(ctx, R.lines, R.bytes, R.len)) {
            pbsim_job_feed_abort(ctx, pbsim_last_error());  // the job fails with this message where it needs the record
            return;
          }
        }
      });
      struct Joiner {
        std::thread &t;
        ~Joiner() {
          if (t.joinable()) t.join();
        }
      } feeder_joined{feeder};
      // <prefix>_NNNN.ref (pbsim.cpp:948-964), written while the records are prepared and simulated: a small pool of threads
      // that draw record numbers from a counter (a thread per record was up to REF_SEQ_NUM_MAX = 9999 threads at once -- under a
      // pids / ulimit -u bound std::thread throws while earlier threads are joinable: std::terminate in a GPU process)
      std::atomic<long> ref_next{1};
      if (rank0)
        for (long t = 0; t < std::min<long>(8, gi.num_seq); t++)
          ref_writers.emplace_back([&]() {
            for (long n; (n = ref_next.fetch_add(1)) <= gi.num_seq;) {
              std::string e;
              if (!pbsim::write_ref_record(c.prefix.c_str(), n, fm.recs[(size_t)n - 1], &e)) {
                std::lock_guard<std::mutex> lk(ref_mu);
                ref_failed = true;
                ref_err = e;
              }
            }
          });
      JobFiles jf;
      jf.cli = &c;
      jf.ctx = ctx;
      jf.comm = comm;
      jf.first = 1;
      jf.use_async_writes();
      if (jf.pool) check(pbsim_job_set_interleave(ctx, (int)std::min<long>(4, gi.num_seq)));
      for (long n = 1; n <= gi.num_seq; n++) jf.recs.emplace_back(new RecFiles);
      if (rank0) for (long n = 1; n <= gi.num_seq; n++) open_record(jf, n, true);
      barrier(comm);  // the files exist
      if (!rank0) for (long n = 1; n <= gi.num_seq; n++) open_record(jf, n, false);
      pbsim_record_sink sink = {&jf, job_read, job_maf, job_done};
      phase("records uploaded, output files open");
      check(pbsim_job_run(ctx, comm, &sink));
      feeder.join();
      phase("job run, bytes handed over");
      jf.close_all();
      phase("output files written");
      if (trace && jf.pool)
        fprintf(stderr, "[pbsim cli]   write pool: %.1f GB copied in %.2f s of the sinks' time (two delivery threads), %.2f s waiting for a buffer, %zu buffers of 32 MB\n",
                jf.pool->copied / 1e9, jf.pool->copy_s, jf.pool->waited_s, jf.pool->made);
      for (auto &t : ref_writers) t.join();
      if (ref_failed) die(": %s", ref_err.c_str());
      phase(".ref files written");
      if (!(getenv("PBSIM_CLI_LEAVE_CONTEXT") && *getenv("PBSIM_CLI_LEAVE_CONTEXT") == '1' && world == 1)) check(pbsim_job_begin(ctx, 1));
      else fm.map = nullptr;  // (3 GB of mapped file: unmapped with the process)
    } else {
    if (rank0 && !pbsim::split_genome(c.genome.c_str(), c.prefix.c_str(), &gi, &err)) die(": %s", err.c_str());
    phase("genome split into .ref files");
    if (world > 1) {
      int64_t nrec = rank0 ? gi.num_seq : 0;
      if (!comm->all_reduce_i64(comm->user, &nrec, 1, PBSIM_OP_SUM)) die(": communicator failed");
      gi.num_seq = (long)nrec;
      std::vector<int64_t> lens((size_t)nrec, 0);
      if (rank0) for (long i = 0; i < gi.num_seq; i++) lens[(size_t)i] = gi.len[(size_t)i];
      if (nrec && !comm->all_reduce_i64(comm->user, lens.data(), nrec, PBSIM_OP_SUM)) die(": communicator failed");
      gi.len.assign(lens.begin(), lens.end());
    }
    // records per job: what leaves most of the HBM to the batches (2 bytes per base resident: sequence + homopolymer lengths)
    const char *gb = getenv("PBSIM_JOB_REF_GB");
    const double ref_budget = (gb && atof(gb) > 0 ? atof(gb) : 64.0) * (double)(1LL << 30) / 2.0;
    std::vector<std::pair<long, long>> groups;  // [first, last] record numbers
    for (long n = 1; n <= gi.num_seq;) {
      long m = n;
      double sum = (double)gi.len[(size_t)n - 1];
      while (m + 1 <= gi.num_seq && sum + (double)gi.len[(size_t)m] <= ref_budget) sum += (double)gi.len[(size_t)m++];
      groups.emplace_back(n, m);
      n = m + 1;
    }
    std::string seq;
    const bool bcast = world > 1 && comm->broadcast != NULL;
    auto load = [&](long n) -> const uint8_t * {
      if (bcast && !rank0) return NULL;
      if (!pbsim::load_ref_record(c.prefix.c_str(), n, &seq, &err)) die(": %s", err.c_str());
      return (const uint8_t *)seq.data();
    };
    if (!bcast) barrier(comm);  // the .ref files are complete before anybody else reads them
    if (c.p.hp_del_bias != 1 && groups.size() > 1) {  // pbsim.cpp:677-696: the census covers ALL records before the first read
      for (long n = 1; n <= gi.num_seq; n++) {
        if (!pbsim::load_ref_record(c.prefix.c_str(), n, &seq, &err)) die(": %s", err.c_str());
        check(pbsim_add_hp_census(ctx, (const uint8_t *)seq.data(), (int64_t)seq.size()));
      }
      check(pbsim_finish_hp_census(ctx));
    }
    for (const auto &g : groups) {
      check(pbsim_job_begin(ctx, g.first));
      for (long n = g.first; n <= g.second; n++) {
        const uint8_t *p = load(n);
        check(pbsim_job_add_record_comm(ctx, p, (int64_t)gi.len[(size_t)n - 1], comm, 0));
      }
      JobFiles jf;
      jf.cli = &c;
      jf.ctx = ctx;
      jf.comm = comm;
      jf.first = g.first;
      jf.use_async_writes();
      if (jf.pool) check(pbsim_job_set_interleave(ctx, (int)std::min<long>(4, g.second - g.first + 1)));
      for (long n = g.first; n <= g.second; n++) jf.recs.emplace_back(new RecFiles);
      if (rank0) for (long n = g.first; n <= g.second; n++) open_record(jf, n, true);
      barrier(comm);  // the files exist
      if (!rank0) for (long n = g.first; n <= g.second; n++) open_record(jf, n, false);
      pbsim_record_sink sink = {&jf, job_read, job_maf, job_done};
      phase("records loaded and uploaded, output files open");
      check(pbsim_job_run(ctx, comm, &sink));
      phase("job run, bytes handed over");
      jf.close_all();
      phase("output files written");
    }
    check(pbsim_job_begin(ctx, 1));
    }  // !mapped
  } else if (c.p.strategy == PBSIM_STRATEGY_WGS) {  // the sampling method, or SAM text into a samtools pipe: record by record
    if (world > 1 && !sampling) die(": this combination of options runs on one GPU.");
    pbsim::GenomeInfo gi;
    if (rank0 && !pbsim::split_genome(c.genome.c_str(), c.prefix.c_str(), &gi, &err)) die(": %s", err.c_str());
    if (world > 1) {  // the record count travels; every rank then reads the <prefix>_NNNN.ref files rank 0 wrote
      int64_t nrec = rank0 ? gi.num_seq : 0;
      if (!comm->all_reduce_i64(comm->user, &nrec, 1, PBSIM_OP_SUM)) die(": communicator failed");
      gi.num_seq = (long)nrec;
    }
    std::string seq;
    if (c.p.hp_del_bias != 1) {
      for (long n = 1; n <= gi.num_seq; n++) {
        if (!pbsim::load_ref_record(c.prefix.c_str(), n, &seq, &err)) die(": %s", err.c_str());
        check(pbsim_add_hp_census(ctx, (const uint8_t *)seq.data(), (int64_t)seq.size()));
      }
      check(pbsim_finish_hp_census(ctx));
    }
    for (long n = 1; n <= gi.num_seq; n++) {
      if (!pbsim::load_ref_record(c.prefix.c_str(), n, &seq, &err)) die(": %s", err.c_str());
      check(pbsim_set_reference(ctx, (const uint8_t *)seq.data(), (int64_t)seq.size(), n));
      if (world > 1) {
        // the sampling method on several ranks (pbsim.cpp:1694-1949): string blocks per rank, every rank writes its byte
        // ranges of the record's two files (rank 0 creates them), the merged statistics come back on every rank
        JobFiles jf;
        jf.cli = &c;
        jf.ctx = ctx;
        jf.comm = comm;
        jf.first = n;
        jf.recs.emplace_back(new RecFiles);
        if (rank0) open_record(jf, n, true);
        barrier(comm);  // the files exist
        if (!rank0) open_record(jf, n, false);
        pbsim_record_sink rsink = {&jf, job_read, job_maf, job_done};
        check(pbsim_simulate_sample_comm(ctx, comm, &rsink));
        continue;
      }
      Out o_read, o_maf;
      if (c.p.pass_num == 1) {
        snprintf(name, sizeof name, "%s_%04ld.fq", c.prefix.c_str(), n);
        open_sink(c, &o_read, name, std::string(name) + ".gz", false);
      } else {
        snprintf(name, sizeof name, "%s_%04ld", c.prefix.c_str(), n);
        open_sink(c, &o_read, std::string(name) + ".sam", std::string(name) + ".bam", true);
        write_read_header(c, ctx, &o_read, n);
      }
      snprintf(name, sizeof name, "%s_%04ld.maf", c.prefix.c_str(), n);
      open_sink(c, &o_maf, name, std::string(name) + ".gz", false);
      Two two = {&o_read, &o_maf};
      pbsim_sink sink = {&two, cb_read, cb_maf};
      phase("record loaded, sinks open");
      check(sampling ? pbsim_simulate_sample(ctx, &sink) : pbsim_simulate_wgs(ctx, &sink));
      phase("record simulated and delivered");
      pbsim_stats st;
      check(pbsim_get_stats(ctx, &st));
      print_simulation_stats(c, st, n);
      o_read.close();
      o_maf.close();
      phase("files closed");
    }
  } else {  // pbsim.cpp:761-812 (trans), 813-866 (templ)
    const bool templ = c.p.strategy == PBSIM_STRATEGY_TEMPL;
    std::vector<pbsim::Transcript> tr;
    long total_exp = 0;
    if (templ) {
      long num = 0;
      long long len_total = 0;
      if (!pbsim::read_templates(c.templ.c_str(), &tr, &num, &len_total, &err)) die(": %s", err.c_str());
      if (rank0) {
        fprintf(stderr, ":::: Template stats ::::\n\n");
        fprintf(stderr, "file name : %s\n", c.templ.c_str());
        fprintf(stderr, "template num. : %ld\n", num);
        fprintf(stderr, "template total length : %lld\n", len_total);
        fprintf(stderr, "\n");
      }
    } else {
      if (!pbsim::read_transcripts(c.transcript.c_str(), &tr, &total_exp, &err)) die(": %s", err.c_str());
      if (rank0) {
        fprintf(stderr, ":::: transcript stats ::::\n\n");
        fprintf(stderr, "file name : %s\n", c.transcript.c_str());
        fprintf(stderr, "transcript num : %ld\n", (long)tr.size());
        fprintf(stderr, "total expression value : %ld\n", total_exp);
        fprintf(stderr, "\n");
      }
    }
    std::vector<const char *> ids;
    std::vector<int64_t> plus, minus, lens;
    std::vector<const uint8_t *> seqs;
    for (auto &t : tr) {
      ids.push_back(t.id.c_str());
      plus.push_back(t.plus);
      minus.push_back(t.minus);
      seqs.push_back((const uint8_t *)t.seq.data());
      lens.push_back((int64_t)strlen(t.seq.c_str()));
    }
    if (templ) check(pbsim_set_templates(ctx, (int64_t)tr.size(), ids.data(), seqs.data(), lens.data()));
    else check(pbsim_set_transcripts(ctx, (int64_t)tr.size(), ids.data(), plus.data(), minus.data(), seqs.data(), lens.data()));
    if (world == 1) {
      Out o_read, o_maf;
      if (c.p.pass_num == 1) {
        open_sink(c, &o_read, c.prefix + ".fq", c.prefix + ".fq.gz", false);
      } else {
        open_sink(c, &o_read, c.prefix + ".sam", c.prefix + ".bam", true);
        write_read_header(c, ctx, &o_read, 0);
      }
      open_sink(c, &o_maf, c.prefix + ".maf", c.prefix + ".maf.gz", false);
      Two two = {&o_read, &o_maf};
      pbsim_sink sink = {&two, cb_read, cb_maf};
      check(pbsim_simulate_trans(ctx, &sink));
      pbsim_stats st;
      check(pbsim_get_stats(ctx, &st));
      print_simulation_stats(c, st, 0);
      o_read.close();
      o_maf.close();
    } else {
      // No quota here: rank r takes the r-th contiguous block of the unit set's read numbering (pbsim.cpp:4516-4522 assigns
      // reads to transcripts in file order), keeps its bytes, learns its offsets from the other ranks' sizes and writes them.
      const int64_t R = pbsim_unit_reads(ctx), per = (R + world - 1) / world;
      const int64_t first = 1 + (int64_t)rank * per, n = std::max<int64_t>(0, std::min(per, R - first + 1));
      std::string buf_r, buf_m;
      struct Keep {
        std::string *r, *m;
      } keep = {&buf_r, &buf_m};
      pbsim_sink sink = {&keep,
                         [](void *u, const char *t, int64_t k) { ((Keep *)u)->r->append(t, (size_t)k); return 1; },
                         [](void *u, const char *t, int64_t k) { ((Keep *)u)->m->append(t, (size_t)k); return 1; }};
      check(pbsim_stats_keep_values(ctx, 1));
      check(pbsim_simulate_units_range(ctx, n > 0 ? first : 1, n, &sink));  // more ranks than reads: the last ranks take none
      check(pbsim_stats_merge(ctx, comm));
      const bool bam = c.p.pass_num > 1;
      const std::string rname = c.prefix + (c.p.pass_num == 1 ? (c.no_gzip ? ".fq" : ".fq.gz") : (c.no_gzip ? ".sam" : ".bam"));
      const std::string mname = c.prefix + (c.no_gzip ? ".maf" : ".maf.gz");
      std::vector<char> h;
      if (bam) {
        if (native_bam(c)) {
          h.resize((size_t)pbsim_bam_header(ctx, NULL, 0));
          pbsim_bam_header(ctx, h.data(), (int64_t)h.size());
          std::vector<char> z((size_t)pbsim_deflate_bound((int64_t)h.size()) + 64);
          int64_t k = 0;
          check(pbsim_deflate_buffer(ctx, h.data(), (int64_t)h.size(), z.data(), (int64_t)z.size(), &k));
          z.resize((size_t)k);
          h.swap(z);
        } else {
          h.resize((size_t)pbsim_sam_header(ctx, NULL, 0) + 1);
          pbsim_sam_header(ctx, h.data(), (int64_t)h.size());
          h.pop_back();
        }
      }
      const int64_t mine[2] = {(int64_t)buf_r.size(), (int64_t)buf_m.size()};
      std::vector<int64_t> all((size_t)world * 2);
      if (!comm->all_gather_i64(comm->user, mine, 2, all.data())) die(": communicator failed");
      int64_t at_r = (int64_t)h.size(), at_m = 0, tot_r = (int64_t)h.size(), tot_m = 0;
      for (int q = 0; q < world; q++) {
        if (q < rank) {
          at_r += all[(size_t)q * 2];
          at_m += all[(size_t)q * 2 + 1];
        }
        tot_r += all[(size_t)q * 2];
        tot_m += all[(size_t)q * 2 + 1];
      }
      PosFile fr, fm;
      if (rank0) {
        fr.create(rname, h.data(), (int64_t)h.size());
        fm.create(mname, nullptr, 0);
        if (!c.no_gzip) {  // BAM: the BGZF end-of-file marker; .gz: an empty member keeps an empty output a valid gzip file
          if ((bam || tot_r == 0) && !fr.write_at((const char *)kBgzfEof, sizeof kBgzfEof, tot_r)) die(": write error on %s", rname.c_str());
          if (tot_m == 0 && !fm.write_at((const char *)kBgzfEof, sizeof kBgzfEof, 0)) die(": write error on %s", mname.c_str());
        }
      }
      barrier(comm);
      if (!rank0) {
        fr.attach(rname);
        fm.attach(mname);
      }
      if (!fr.write_at(buf_r.data(), (int64_t)buf_r.size(), at_r) || !fm.write_at(buf_m.data(), (int64_t)buf_m.size(), at_m))
        die(": write error on %s", rname.c_str());
      fr.close_checked();
      fm.close_checked();
      pbsim_stats st;
      check(pbsim_get_stats(ctx, &st));
      if (rank0) print_simulation_stats(c, st, 0);
    }
  }
  phase("simulation done");
  // The `pbsim` binary ends the process right behind this call (main.cpp sets PBSIM_CLI_LEAVE_CONTEXT=1 for a single rank):
  // handing 150 GB of HBM pools and the pinned staging back piece by piece takes 0.8 s that the process exit does at once.
  // A caller that lives on (pbsim3_amd.cli_main, a rank thread of --devices) gets its memory back here.
  const char *leave = getenv("PBSIM_CLI_LEAVE_CONTEXT");
  if (!(leave && *leave == '1' && world == 1)) pbsim_destroy(ctx);
  phase("context destroyed");
  barrier(comm);

  if (rank0) {
    struct rusage ru;
    getrusage(RUSAGE_SELF, &ru);
    struct timeval tv1;
    gettimeofday(&tv1, NULL);
    fprintf(stderr, ":::: System utilization ::::\n\n");
    fprintf(stderr, "CPU time(s) : %ld\n", (long)ru.ru_utime.tv_sec);
    fprintf(stderr, "Elapsed time(s) : %ld\n", (long)(tv1.tv_sec - tv0.tv_sec));
  }
  return 0;
}
