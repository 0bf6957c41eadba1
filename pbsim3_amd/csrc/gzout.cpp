#include "gzout.h"

#include <string.h>
#include <stdint.h>
#include <zlib.h>

#include <algorithm>

namespace pbsim {

namespace {
bool deflate_member(const std::string &in, std::string *out) {
  z_stream zs;
  memset(&zs, 0, sizeof zs);
  if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
  out->resize(deflateBound(&zs, (uLong)in.size()) + 32);
  zs.next_in = reinterpret_cast<Bytef *>(const_cast<char *>(in.data()));
  zs.avail_in = (uInt)in.size();
  zs.next_out = reinterpret_cast<Bytef *>(&(*out)[0]);
  zs.avail_out = (uInt)out->size();
  const int rc = deflate(&zs, Z_FINISH);
  const bool ok = rc == Z_STREAM_END;
  out->resize(ok ? zs.total_out : 0);
  deflateEnd(&zs);
  return ok;
}
// one BGZF block: gzip member with the BC subfield carrying the block size, raw deflate payload
bool deflate_bgzf(const std::string &in, std::string *out) {
  for (int level : {Z_DEFAULT_COMPRESSION, 0}) {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    out->assign(18 + deflateBound(&zs, (uLong)in.size()) + 8, '\0');
    zs.next_in = reinterpret_cast<Bytef *>(const_cast<char *>(in.data()));
    zs.avail_in = (uInt)in.size();
    zs.next_out = reinterpret_cast<Bytef *>(&(*out)[18]);
    zs.avail_out = (uInt)(out->size() - 18 - 8);
    const int rc = deflate(&zs, Z_FINISH);
    const size_t clen = zs.total_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END) return false;
    const size_t total = 18 + clen + 8;
    if (total > 65536) continue;  // incompressible: store
    static const unsigned char hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    memcpy(&(*out)[0], hdr, 16);
    (*out)[16] = (char)((total - 1) & 0xff);
    (*out)[17] = (char)((total - 1) >> 8);
    const uLong crc = crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const Bytef *>(in.data()), (uInt)in.size());
    for (int i = 0; i < 4; i++) (*out)[18 + clen + i] = (char)(crc >> (8 * i));
    for (int i = 0; i < 4; i++) (*out)[18 + clen + 4 + i] = (char)((uint32_t)in.size() >> (8 * i));
    out->resize(total);
    return true;
  }
  return false;
}
}  // namespace

ParallelGz::~ParallelGz() { close(); }

bool ParallelGz::open(const std::string &path, int threads, std::string *err, bool bgzf) {
  bgzf_ = bgzf;
  block_ = bgzf ? 0xff00 : (1 << 20);
  fp_ = fopen(path.c_str(), "wb");
  if (!fp_) {
    *err = "Cannot open output file: " + path;
    return false;
  }
  if (threads < 1) threads = 1;
  max_pending_ = (size_t)threads * 4;
  for (int i = 0; i < threads; i++) pool_.emplace_back(&ParallelGz::worker, this);
  return true;
}

void ParallelGz::worker() {
  for (;;) {
    Job job;
    {
      std::unique_lock<std::mutex> lk(mu_);
      cv_job_.wait(lk, [&] { return stop_ || !jobs_.empty(); });
      if (jobs_.empty()) return;
      job = std::move(jobs_.front());
      jobs_.pop_front();
    }
    std::string out;
    const bool ok = bgzf_ ? deflate_bgzf(job.in, &out) : deflate_member(job.in, &out);
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (!ok) failed_ = true;
      done_[job.seq] = std::move(out);
    }
    cv_done_.notify_all();
  }
}

// hands the current block to the pool and writes whatever finished in order
void ParallelGz::submit() {
  std::unique_lock<std::mutex> lk(mu_);
  if (!cur_.empty()) {
    jobs_.push_back(Job{next_submit_++, std::move(cur_)});
    cur_.clear();
    cv_job_.notify_one();
  }
  for (;;) {
    auto it = done_.find(next_write_);
    if (it != done_.end()) {
      std::string blk = std::move(it->second);
      done_.erase(it);
      next_write_++;
      lk.unlock();
      if (fwrite(blk.data(), 1, blk.size(), fp_) != blk.size()) failed_ = true;
      lk.lock();
      continue;
    }
    if (next_submit_ - next_write_ <= max_pending_) break;  // bounded memory: wait for the oldest block
    cv_done_.wait(lk);
  }
}

bool ParallelGz::write(const char *data, size_t n) {
  if (!fp_) return false;
  while (n) {
    const size_t take = std::min(n, block_ - cur_.size());
    cur_.append(data, take);
    data += take;
    n -= take;
    if (cur_.size() >= block_) submit();
  }
  return !failed_;
}

bool ParallelGz::close() {
  if (!fp_) return !failed_;
  if (next_submit_ == 0 && cur_.empty() && !bgzf_) {  // an empty file is still one (empty) gzip member
    std::lock_guard<std::mutex> lk(mu_);
    jobs_.push_back(Job{next_submit_++, std::string()});
    cv_job_.notify_one();
  }
  submit();
  {
    std::unique_lock<std::mutex> lk(mu_);
    while (next_write_ < next_submit_) {
      auto it = done_.find(next_write_);
      if (it == done_.end()) {
        cv_done_.wait(lk);
        continue;
      }
      std::string blk = std::move(it->second);
      done_.erase(it);
      next_write_++;
      lk.unlock();
      if (fwrite(blk.data(), 1, blk.size(), fp_) != blk.size()) failed_ = true;
      lk.lock();
    }
    stop_ = true;
  }
  cv_job_.notify_all();
  for (auto &t : pool_) t.join();
  pool_.clear();
  if (bgzf_) {  // the 28-byte EOF marker block
    static const unsigned char eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0,
                                          0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (fwrite(eof, 1, 28, fp_) != 28) failed_ = true;
  }
  if (fclose(fp_) != 0) failed_ = true;
  fp_ = nullptr;
  return !failed_;
}

}  // namespace pbsim
