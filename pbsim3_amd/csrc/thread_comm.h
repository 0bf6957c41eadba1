// thread_comm.h -- pbsim_comm for several ranks inside ONE process (one host thread per GPU): the collectives of the job
// are a handful of integers per round, so a host barrier over shared memory is the natural transport; the one bulk
// transfer, the broadcast of a record (C1), is a device-to-device copy from the root's GPU (xGMI peer copy when the
// devices differ).  Used by the `pbsim` binary (--devices) and by examples; header-only, needs only HIP and the C++ thread library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <condition_variable>
#include <mutex>
#include <vector>

#include "../../include/pbsim3_amd.h"

namespace pbsim {

// A barrier that can be given up: pbsim_comm.abort (a rank whose job failed between two exchanges) wakes every waiter and
// makes this and every later collective return 0, so the other ranks fail instead of waiting for ever (a pthread barrier has
// no such exit).
struct AbortableBarrier {
  std::mutex mu;
  std::condition_variable cv;
  int world = 1, count = 0;
  uint64_t gen = 0;
  bool aborted = false;
  bool wait() {
    std::unique_lock<std::mutex> lk(mu);
    if (aborted) return false;
    const uint64_t g = gen;
    if (++count == world) {
      count = 0;
      gen++;
      cv.notify_all();
      return true;
    }
    cv.wait(lk, [&] { return gen != g || aborted; });
    return gen != g;
  }
  void abort() {
    {
      std::lock_guard<std::mutex> lk(mu);
      aborted = true;
    }
    cv.notify_all();
  }
};

struct ThreadCommShared {
  int world = 1;
  AbortableBarrier bar;
  std::vector<const void *> ptr;  // what each rank published for the collective in progress
  std::vector<int> device;        // GPU of each rank
  explicit ThreadCommShared(const std::vector<int> &devices) : world((int)devices.size()), ptr(devices.size()), device(devices) {
    bar.world = world;
  }
};

struct ThreadCommRank {
  ThreadCommShared *sh;
  int rank;
};

inline int tc_all_gather(void *user, const int64_t *send, int64_t n, int64_t *recv) {
  ThreadCommRank *r = (ThreadCommRank *)user;
  ThreadCommShared *sh = r->sh;
  sh->ptr[(size_t)r->rank] = send;
  if (!sh->bar.wait()) return 0;
  for (int q = 0; q < sh->world; q++) memcpy(recv + (size_t)q * n, sh->ptr[(size_t)q], (size_t)n * 8);
  return sh->bar.wait() ? 1 : 0;  // nobody's send buffer goes away before everybody has read it
}

inline int tc_all_reduce(void *user, int64_t *buf, int64_t n, int32_t op) {
  ThreadCommRank *r = (ThreadCommRank *)user;
  ThreadCommShared *sh = r->sh;
  sh->ptr[(size_t)r->rank] = buf;
  if (!sh->bar.wait()) return 0;
  std::vector<int64_t> out((size_t)n);
  for (int64_t i = 0; i < n; i++) {
    int64_t v = ((const int64_t *)sh->ptr[0])[i];
    for (int q = 1; q < sh->world; q++) {
      const int64_t x = ((const int64_t *)sh->ptr[(size_t)q])[i];
      v = op == PBSIM_OP_SUM ? v + x : op == PBSIM_OP_MIN ? (x < v ? x : v) : (x > v ? x : v);
    }
    out[(size_t)i] = v;
  }
  if (!sh->bar.wait()) return 0;  // everybody has read the inputs: now they may be overwritten
  if (n) memcpy(buf, out.data(), (size_t)n * 8);
  return 1;
}

inline int tc_broadcast(void *user, void *p, int64_t bytes, int32_t root, int32_t on_device) {
  ThreadCommRank *r = (ThreadCommRank *)user;
  ThreadCommShared *sh = r->sh;
  if (r->rank == root) sh->ptr[(size_t)root] = p;
  if (!sh->bar.wait()) return 0;
  int ok = 1;
  if (r->rank != root) {
    const void *src = sh->ptr[(size_t)root];
    if (!on_device) {
      memcpy(p, src, (size_t)bytes);
    } else if (sh->device[(size_t)root] == sh->device[(size_t)r->rank]) {
      ok = hipMemcpy(p, src, (size_t)bytes, hipMemcpyDeviceToDevice) == hipSuccess;
    } else {  // GPU to GPU over xGMI
      ok = hipMemcpyPeer(p, sh->device[(size_t)r->rank], src, sh->device[(size_t)root], (size_t)bytes) == hipSuccess;
    }
    // a device-to-device hipMemcpy returns before the bytes have moved: the root must not free its buffer, and this rank
    // must not read its own, before they have
    if (on_device) ok = ok && hipDeviceSynchronize() == hipSuccess;
  }
  if (!sh->bar.wait()) return 0;
  return ok;
}

inline int tc_abort(void *user) {
  ((ThreadCommRank *)user)->sh->bar.abort();
  return 1;
}

inline pbsim_comm thread_comm(ThreadCommRank *r) {
  pbsim_comm c;
  c.user = r;
  c.rank = r->rank;
  c.world = r->sh->world;
  c.all_gather_i64 = tc_all_gather;
  c.all_reduce_i64 = tc_all_reduce;
  c.broadcast = tc_broadcast;
  c.abort = tc_abort;
  return c;
}

}  // namespace pbsim
