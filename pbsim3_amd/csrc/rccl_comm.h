// rccl_comm.h -- pbsim_comm over RCCL (xGMI): ncclBroadcast carries a record from the loading rank's GPU to all others
// (C1), ncclAllGather the per-round integers (C3), ncclAllReduce the statistics (C2) -- the three collectives SURVEY 8(e)
// names.  Two ways to a communicator: the ranks of ONE process, one host thread per GPU (rccl_init_all = ncclCommInitAll;
// `pbsim --devices`), and ONE PROCESS PER GPU (rccl_init_rank = ncclGetUniqueId on rank 0 + ncclCommInitRank on every rank;
// the id travels through whatever the launcher offers -- torch's store, a rendezvous file: pbsim_rccl_* in rccl_capi.cpp;
// `bench.py --gpus N`, run_multi, `pbsim --rank R --world N --rendezvous FILE`).  librccl is opened at run time (dlopen), so
// the binary has no link-time dependency on it and a box without it still runs --comm host.  RCCL refuses two ranks on one
// device: distinct GPUs only (a single-GPU box can run a communicator of one).
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <memory>
#include <string>
#include <vector>

#include "../../include/pbsim3_amd.h"

namespace pbsim {

struct RcclApi {
  void *lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;     // optional
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;  // optional
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional
  ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  bool load(std::string *err) {
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) {
      *err = "cannot open librccl.so";
      return false;
    }
#define PBSIM_RCCL_SYM(field, sym)                                \
  field = reinterpret_cast<decltype(field)>(dlsym(lib, sym));     \
  if (!field) {                                                   \
    *err = std::string("librccl lacks ") + sym;                   \
    return false;                                                 \
  }
    PBSIM_RCCL_SYM(CommInitAll, "ncclCommInitAll")
    PBSIM_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
    PBSIM_RCCL_SYM(CommInitRank, "ncclCommInitRank")
    PBSIM_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    PBSIM_RCCL_SYM(Broadcast, "ncclBroadcast")
    PBSIM_RCCL_SYM(AllReduce, "ncclAllReduce")
    PBSIM_RCCL_SYM(AllGather, "ncclAllGather")
    PBSIM_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef PBSIM_RCCL_SYM
    CommAbort = reinterpret_cast<decltype(CommAbort)>(dlsym(lib, "ncclCommAbort"));
    CommCount = reinterpret_cast<decltype(CommCount)>(dlsym(lib, "ncclCommCount"));
    CommUserRank = reinterpret_cast<decltype(CommUserRank)>(dlsym(lib, "ncclCommUserRank"));
    return true;
  }
};

struct RcclRank {
  RcclApi *api = nullptr;
  // pbsim_comm.abort of any rank of THIS group (rccl_init_all makes one per group: a later communicator of the same process
  // starts clean -- the flag used to live in the process-wide RcclApi and failed every later group)
  std::shared_ptr<std::atomic<bool>> aborted;
  bool is_aborted() const { return aborted && aborted->load(std::memory_order_relaxed); }
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t stream = nullptr;
  void *d_send = nullptr, *d_recv = nullptr;  // staging for the integer collectives
  size_t cap = 0;
  // Page-locked, persistent host staging for the small collectives (round 5): a job has 30-100 exchanges of a few dozen
  // words, and a copy to or from PAGEABLE memory is a staged, synchronous copy inside the runtime (~20-40 us each way); from
  // pinned memory both copies are plain DMA packets on the collective's stream.  Larger messages (the statistics merge's
  // histograms and accuracy values) go from the caller's buffers as before.
  static constexpr size_t kPinned = 256 << 10;
  void *h_send = nullptr, *h_recv = nullptr;
  int64_t collectives = 0;  // issued through this rank since the communicator was made (pbsim_rccl_comm_info)
  bool pinned_for(size_t bytes_send, size_t bytes_recv) {
    if (bytes_send > kPinned || bytes_recv > kPinned) return false;
    if (h_send) return true;
    if (hipHostMalloc(&h_send, kPinned, hipHostMallocDefault) != hipSuccess) {
      h_send = nullptr;
      return false;
    }
    if (hipHostMalloc(&h_recv, kPinned, hipHostMallocDefault) != hipSuccess) {
      (void)hipHostFree(h_send);
      h_send = h_recv = nullptr;
      return false;
    }
    return true;
  }
  bool ensure(size_t bytes_send, size_t bytes_recv) {
    if (hipSetDevice(device) != hipSuccess) return false;
    if (!stream && hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) return false;
    const size_t need = bytes_send > bytes_recv ? bytes_send : bytes_recv;
    if (need <= cap) return true;
    if (d_send) (void)hipFree(d_send);
    if (d_recv) (void)hipFree(d_recv);
    cap = need + need / 4 + 4096;
    return hipMalloc(&d_send, cap) == hipSuccess && hipMalloc(&d_recv, cap) == hipSuccess;
  }
};

// The watchdog: a collective is waited for by polling its stream.  One that does not come back within PBSIM_COMM_TIMEOUT_S
// seconds (default 600) means that some rank never entered it -- it failed, or the ranks fell out of step -- and nothing the
// process could do would bring it back: say so and end the process (a fresh exit; no restart in-process).  A rank whose job
// failed calls pbsim_comm.abort: the others then give their communicators up and return 0 instead of waiting for the timeout.
inline bool rccl_wait(RcclRank *r, const char *what) {
  static const double timeout_s = [] {
    const char *e = getenv("PBSIM_COMM_TIMEOUT_S");
    return e && atof(e) > 0 ? atof(e) : 600.0;
  }();
  const auto t0 = std::chrono::steady_clock::now();
  for (long spins = 0;; spins++) {
    const hipError_t e = hipStreamQuery(r->stream);
    if (e == hipSuccess) return true;
    if (e != hipErrorNotReady) return false;
    if (r->is_aborted() && r->api->CommAbort) {
      if (r->comm) (void)r->api->CommAbort(r->comm);
      r->comm = nullptr;
      return false;
    }
    // (a librccl without ncclCommAbort: the collective's kernel keeps spinning on the stream, so the communicator is NOT
    // dropped -- rccl_destroy_all would block in hipFree behind that kernel; the wait runs into the time-out below and the
    // process ends)
    if (spins < 4000) {
      sched_yield();
      continue;
    }
    usleep(50);
    if ((spins & 1023) == 0 &&
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
      fprintf(stderr,
              "ERROR: rank %d of %d: the RCCL %s did not complete within %.0f s (PBSIM_COMM_TIMEOUT_S): another rank failed or the "
              "ranks fell out of step.  Aborting the process.\n",
              r->rank, r->world, what, timeout_s);
      fflush(stderr);
      _exit(124);
    }
  }
}

inline int rccl_abort(void *user) {
  RcclRank *r = (RcclRank *)user;
  if (r->aborted) r->aborted->store(true);
  return 1;
}

inline int rccl_all_gather(void *user, const int64_t *send, int64_t n, int64_t *recv) {
  RcclRank *r = (RcclRank *)user;
  if (!r->comm || r->is_aborted()) return 0;
  const size_t bs = (size_t)n * 8, br = bs * (size_t)r->world;
  if (!r->ensure(bs, br)) return 0;
  const bool pin = r->pinned_for(bs, br);
  if (pin) memcpy(r->h_send, send, bs);
  if (hipMemcpyAsync(r->d_send, pin ? r->h_send : send, bs, hipMemcpyHostToDevice, r->stream) != hipSuccess) return 0;
  if (r->api->AllGather(r->d_send, r->d_recv, (size_t)n, ncclInt64, r->comm, r->stream) != ncclSuccess) return 0;  // C3
  if (hipMemcpyAsync(pin ? r->h_recv : recv, r->d_recv, br, hipMemcpyDeviceToHost, r->stream) != hipSuccess) return 0;
  if (!rccl_wait(r, "all-gather")) return 0;
  if (pin) memcpy(recv, r->h_recv, br);
  r->collectives++;
  return 1;
}

inline int rccl_all_reduce(void *user, int64_t *buf, int64_t n, int32_t op) {
  RcclRank *r = (RcclRank *)user;
  if (!r->comm || r->is_aborted()) return 0;
  if (n == 0) return 1;
  const size_t bs = (size_t)n * 8;
  if (!r->ensure(bs, bs)) return 0;
  const bool pin = r->pinned_for(bs, bs);
  const ncclRedOp_t rop = op == PBSIM_OP_SUM ? ncclSum : op == PBSIM_OP_MIN ? ncclMin : ncclMax;
  if (pin) memcpy(r->h_send, buf, bs);
  if (hipMemcpyAsync(r->d_send, pin ? r->h_send : buf, bs, hipMemcpyHostToDevice, r->stream) != hipSuccess) return 0;
  if (r->api->AllReduce(r->d_send, r->d_recv, (size_t)n, ncclInt64, rop, r->comm, r->stream) != ncclSuccess) return 0;  // C2
  if (hipMemcpyAsync(pin ? r->h_recv : buf, r->d_recv, bs, hipMemcpyDeviceToHost, r->stream) != hipSuccess) return 0;
  if (!rccl_wait(r, "all-reduce")) return 0;
  if (pin) memcpy(buf, r->h_recv, bs);
  r->collectives++;
  return 1;
}

inline int rccl_broadcast(void *user, void *p, int64_t bytes, int32_t root, int32_t on_device) {
  RcclRank *r = (RcclRank *)user;
  if (!r->comm || r->is_aborted()) return 0;
  if (hipSetDevice(r->device) != hipSuccess) return 0;
  if (!r->stream && hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking) != hipSuccess) return 0;
  void *d = p;
  if (!on_device) {
    if (!r->ensure((size_t)bytes, (size_t)bytes)) return 0;
    d = r->d_send;
    if (r->rank == root && hipMemcpyAsync(d, p, (size_t)bytes, hipMemcpyHostToDevice, r->stream) != hipSuccess) return 0;
  }
  if (r->api->Broadcast(d, d, (size_t)bytes, ncclUint8, root, r->comm, r->stream) != ncclSuccess) return 0;  // C1
  if (!on_device && r->rank != root && hipMemcpyAsync(p, d, (size_t)bytes, hipMemcpyDeviceToHost, r->stream) != hipSuccess) return 0;
  r->collectives++;
  return rccl_wait(r, "broadcast");
}

inline RcclApi &rccl_api() {
  static RcclApi api;
  return api;
}

inline bool rccl_init_all(const std::vector<int> &devices, std::vector<RcclRank> *ranks, std::string *err) {
  RcclApi &api = rccl_api();
  if (!api.lib && !api.load(err)) return false;
  for (size_t i = 0; i < devices.size(); i++)
    for (size_t j = i + 1; j < devices.size(); j++)
      if (devices[i] == devices[j]) {
        *err = "RCCL takes one rank per GPU; list every device once (or use --comm host)";
        return false;
      }
  std::vector<ncclComm_t> comms(devices.size());
  const ncclResult_t rc = api.CommInitAll(comms.data(), (int)devices.size(), devices.data());
  if (rc != ncclSuccess) {
    *err = std::string("ncclCommInitAll: ") + api.GetErrorString(rc);
    return false;
  }
  ranks->resize(devices.size());
  const auto flag = std::make_shared<std::atomic<bool>>(false);
  for (size_t i = 0; i < devices.size(); i++) {
    RcclRank &r = (*ranks)[i];
    r.api = &api;
    r.aborted = flag;
    r.comm = comms[i];
    r.rank = (int)i;
    r.world = (int)devices.size();
    r.device = devices[i];
  }
  return true;
}

// One process per GPU.  Rank 0 makes the id (ncclGetUniqueId: 128 bytes, opens the bootstrap socket the others connect to),
// the launcher's side channel carries it to every rank, and every rank -- rank 0 included -- enters ncclCommInitRank with it:
// that call is itself collective over the `world` ranks.
inline bool rccl_unique_id(ncclUniqueId *id, std::string *err) {
  RcclApi &api = rccl_api();
  if (!api.lib && !api.load(err)) return false;
  const ncclResult_t rc = api.GetUniqueId(id);
  if (rc != ncclSuccess) {
    *err = std::string("ncclGetUniqueId: ") + api.GetErrorString(rc);
    return false;
  }
  return true;
}

inline bool rccl_init_rank(const ncclUniqueId &id, int rank, int world, int device, RcclRank *r, std::string *err) {
  RcclApi &api = rccl_api();
  if (!api.lib && !api.load(err)) return false;
  if (rank < 0 || world < 1 || rank >= world) {
    *err = "rank / world out of range";
    return false;
  }
  if (hipSetDevice(device) != hipSuccess) {
    *err = "hipSetDevice(" + std::to_string(device) + ") failed";
    return false;
  }
  ncclComm_t comm = nullptr;
  const ncclResult_t rc = api.CommInitRank(&comm, world, id, rank);
  if (rc != ncclSuccess) {
    *err = std::string("ncclCommInitRank: ") + api.GetErrorString(rc);
    return false;
  }
  r->api = &api;
  r->aborted = std::make_shared<std::atomic<bool>>(false);  // per process: an abort ends this rank's waits; the others see
  r->comm = comm;                                            // their collectives fail or time out (rccl_wait)
  r->rank = rank;
  r->world = world;
  r->device = device;
  return true;
}

inline void rccl_destroy_all(std::vector<RcclRank> *ranks) {
  for (RcclRank &r : *ranks) {
    (void)hipSetDevice(r.device);
    if (r.stream && r.comm) (void)hipStreamSynchronize(r.stream);
    if (r.comm) (void)r.api->CommDestroy(r.comm);
    if (r.d_send) (void)hipFree(r.d_send);
    if (r.d_recv) (void)hipFree(r.d_recv);
    if (r.h_send) (void)hipHostFree(r.h_send);
    if (r.h_recv) (void)hipHostFree(r.h_recv);
    r.h_send = r.h_recv = nullptr;
    if (r.stream) (void)hipStreamDestroy(r.stream);
    r.comm = nullptr;
  }
}

inline pbsim_comm rccl_comm(RcclRank *r) {
  pbsim_comm c;
  c.user = r;
  c.rank = r->rank;
  c.world = r->world;
  c.all_gather_i64 = rccl_all_gather;
  c.all_reduce_i64 = rccl_all_reduce;
  c.broadcast = rccl_broadcast;
  c.abort = rccl_abort;
  return c;
}

}  // namespace pbsim
