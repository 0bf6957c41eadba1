// comm_abort_driver.cpp -- the abortable host barrier of csrc/thread_comm.h (pbsim_comm.abort): ranks waiting in a collective
// that one rank will never enter are released with a failure instead of waiting for ever.  Test driver, built by
// tests/test_host_sanitizers.py under ThreadSanitizer-free ASan/UBSan (no GPU call is made: the device paths are not entered).
#include <stdio.h>

#include <atomic>
#include <thread>
#include <vector>

#include "thread_comm.h"

int main() {
  const int W = 4;
  std::vector<int> devs(W, 0);
  pbsim::ThreadCommShared sh(devs);
  std::vector<pbsim::ThreadCommRank> ranks;
  for (int r = 0; r < W; r++) ranks.push_back(pbsim::ThreadCommRank{&sh, r});
  std::atomic<int> ok_first{0}, failed_second{0}, failed_third{0};
  auto body = [&](int r) {
    pbsim_comm c = pbsim::thread_comm(&ranks[(size_t)r]);
    int64_t send[2] = {r, 100 + r}, recv[2 * W];
    if (c.all_gather_i64(c.user, send, 2, recv)) {  // everybody enters: succeeds
      bool good = true;
      for (int q = 0; q < W; q++) good = good && recv[q * 2] == q && recv[q * 2 + 1] == 100 + q;
      if (good) ok_first++;
    }
    int64_t v = r;
    if (c.all_reduce_i64(c.user, &v, 1, PBSIM_OP_SUM) && v == W * (W - 1) / 2) ok_first++;
    if (r == 2) {  // rank 2 "fails between two exchanges": it never enters the next collective
      c.abort(c.user);
      return;
    }
    if (!c.all_gather_i64(c.user, send, 2, recv)) failed_second++;
    if (!c.all_reduce_i64(c.user, &v, 1, PBSIM_OP_MAX)) failed_third++;  // and every later one fails at once
  };
  std::vector<std::thread> th;
  for (int r = 0; r < W; r++) th.emplace_back(body, r);
  for (auto &t : th) t.join();
  printf("first %d second_failed %d third_failed %d\n", ok_first.load(), failed_second.load(), failed_third.load());
  return (ok_first == 2 * W && failed_second == W - 1 && failed_third == W - 1) ? 0 : 1;
}
