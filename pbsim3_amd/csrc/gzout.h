// gzout.h -- multi-threaded gzip writer of the CLI shim.  The reference pipes its
// text into one `gzip` child per file (popen, pbsim.cpp:708-730), which costs six
// times the simulation even on the CPU.  Here the text is cut
// into blocks, each block is deflated on a worker thread into its own gzip
// member, and the members are written in order: the result is a standard
// multi-member .gz (RFC 1952; gzip -d, zcat, zlib's gzread all accept it) whose
// decompressed bytes are exactly the text.
#pragma once
#include <stddef.h>
#include <stdio.h>

#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace pbsim {

class ParallelGz {
 public:
  ParallelGz() = default;
  ~ParallelGz();
  // bgzf: BGZF framing (SAMv1 section 4.1: <= 64 KiB blocks, 'BC' extra field, EOF marker) for BAM
  bool open(const std::string &path, int threads, std::string *err, bool bgzf = false);
  bool write(const char *data, size_t n);  // false after an I/O or zlib error
  bool close();

 private:
  struct Job {
    size_t seq;
    std::string in;
  };
  void worker();
  void submit();
  FILE *fp_ = nullptr;
  std::vector<std::thread> pool_;
  std::mutex mu_;
  std::condition_variable cv_job_, cv_done_;
  std::deque<Job> jobs_;
  std::map<size_t, std::string> done_;
  std::string cur_;
  size_t next_submit_ = 0, next_write_ = 0;
  size_t max_pending_ = 0;
  bool stop_ = false, failed_ = false, bgzf_ = false;
  size_t block_ = 1 << 20;
};

}  // namespace pbsim
