#include "gzout.h"

#include <string.h>
#include <zlib.h>

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
}  // namespace

ParallelGz::~ParallelGz() { close(); }

bool ParallelGz::open(const std::string &path, int threads, std::string *err) {
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
    const bool ok = deflate_member(job.in, &out);
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
    const size_t take = std::min(n, kBlock - cur_.size());
    cur_.append(data, take);
    data += take;
    n -= take;
    if (cur_.size() >= kBlock) submit();
  }
  return !failed_;
}

bool ParallelGz::close() {
  if (!fp_) return !failed_;
  if (next_submit_ == 0 && cur_.empty()) cur_.assign("", 0), jobs_.push_back(Job{next_submit_++, std::string()}), cv_job_.notify_one();
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
  if (fclose(fp_) != 0) failed_ = true;
  fp_ = nullptr;
  return !failed_;
}

}  // namespace pbsim
