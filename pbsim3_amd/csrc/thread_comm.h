// thread_comm.h -- pbsim_comm for several ranks inside ONE process (one host thread per GPU): the collectives of the job
// are a handful of integers per round, so a host barrier over shared memory is the natural transport; the one bulk
// transfer, the broadcast of a record (C1), is a device-to-device copy from the root's GPU (xGMI peer copy when the
// devices differ).  Used by the `pbsim` binary (--devices) and by examples; header-only, needs only HIP and pthreads.
#pragma once
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../include/pbsim3_amd.h"

namespace pbsim {

struct ThreadCommShared {
  int world = 1;
  pthread_barrier_t bar;
  std::vector<const void *> ptr;  // what each rank published for the collective in progress
  std::vector<int> device;        // GPU of each rank
  explicit ThreadCommShared(const std::vector<int> &devices) : world((int)devices.size()), ptr(devices.size()), device(devices) {
    pthread_barrier_init(&bar, NULL, (unsigned)world);
  }
  ~ThreadCommShared() { pthread_barrier_destroy(&bar); }
};

struct ThreadCommRank {
  ThreadCommShared *sh;
  int rank;
};

inline int tc_all_gather(void *user, const int64_t *send, int64_t n, int64_t *recv) {
  ThreadCommRank *r = (ThreadCommRank *)user;
  ThreadCommShared *sh = r->sh;
  sh->ptr[(size_t)r->rank] = send;
  pthread_barrier_wait(&sh->bar);
  for (int q = 0; q < sh->world; q++) memcpy(recv + (size_t)q * n, sh->ptr[(size_t)q], (size_t)n * 8);
  pthread_barrier_wait(&sh->bar);  // nobody's send buffer goes away before everybody has read it
  return 1;
}

inline int tc_all_reduce(void *user, int64_t *buf, int64_t n, int32_t op) {
  ThreadCommRank *r = (ThreadCommRank *)user;
  ThreadCommShared *sh = r->sh;
  sh->ptr[(size_t)r->rank] = buf;
  pthread_barrier_wait(&sh->bar);
  std::vector<int64_t> out((size_t)n);
  for (int64_t i = 0; i < n; i++) {
    int64_t v = ((const int64_t *)sh->ptr[0])[i];
    for (int q = 1; q < sh->world; q++) {
      const int64_t x = ((const int64_t *)sh->ptr[(size_t)q])[i];
      v = op == PBSIM_OP_SUM ? v + x : op == PBSIM_OP_MIN ? (x < v ? x : v) : (x > v ? x : v);
    }
    out[(size_t)i] = v;
  }
  pthread_barrier_wait(&sh->bar);  // everybody has read the inputs: now they may be overwritten
  if (n) memcpy(buf, out.data(), (size_t)n * 8);
  return 1;
}

inline int tc_broadcast(void *user, void *p, int64_t bytes, int32_t root, int32_t on_device) {
  ThreadCommRank *r = (ThreadCommRank *)user;
  ThreadCommShared *sh = r->sh;
  if (r->rank == root) sh->ptr[(size_t)root] = p;
  pthread_barrier_wait(&sh->bar);
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
  pthread_barrier_wait(&sh->bar);
  return ok;
}

inline pbsim_comm thread_comm(ThreadCommRank *r) {
  pbsim_comm c;
  c.user = r;
  c.rank = r->rank;
  c.world = r->sh->world;
  c.all_gather_i64 = tc_all_gather;
  c.all_reduce_i64 = tc_all_reduce;
  c.broadcast = tc_broadcast;
  return c;
}

}  // namespace pbsim
