// rccl_capi.cpp -- the RCCL communicator of a one-process-per-GPU launch behind the C ABI (include/pbsim3_amd.h,
// pbsim_rccl_*): ncclGetUniqueId on rank 0, the id carried by the caller (torch's store) or by a rendezvous file, and
// ncclCommInitRank on every rank.  The pbsim_comm handed back is what pbsim_job_run / pbsim_cli_main take; its callbacks are
// rccl_comm.h's (page-locked staging, stream polling with a watchdog, abort).  The reference has no analogue: it is one
// process on one socket (pbsim.cpp:4-14, libc only).
#include <errno.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <time.h>

#include <chrono>
#include <string>

#include "ctx.h"
#include "rccl_comm.h"

namespace pbsim {
namespace {

struct NativeComm {
  pbsim_comm comm;  // first member: the pointer the caller holds is also the NativeComm's
  RcclRank rank;
};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// <path>: 8 bytes magic + 128 bytes id, written to <path>.tmp.<pid> and renamed -- a reader sees all of it or nothing
const char kMagic[8] = {'P', 'B', 'R', 'C', 'C', 'L', '1', '\n'};

bool publish_id(const char *path, const ncclUniqueId &id, std::string *err) {
  const std::string tmp = std::string(path) + ".tmp." + std::to_string((long)getpid());
  const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
  if (fd < 0) {
    *err = "cannot create " + tmp + ": " + strerror(errno);
    return false;
  }
  char buf[sizeof(kMagic) + sizeof(ncclUniqueId)];
  memcpy(buf, kMagic, sizeof(kMagic));
  memcpy(buf + sizeof(kMagic), &id, sizeof(ncclUniqueId));
  const bool ok = write(fd, buf, sizeof(buf)) == (ssize_t)sizeof(buf);
  close(fd);
  if (!ok || rename(tmp.c_str(), path) != 0) {
    *err = "cannot write " + std::string(path) + ": " + strerror(errno);
    unlink(tmp.c_str());
    return false;
  }
  return true;
}

bool fetch_id(const char *path, ncclUniqueId *id, double timeout_s, std::string *err) {
  const double t0 = now_s();
  for (;;) {
    const int fd = open(path, O_RDONLY);
    if (fd >= 0) {
      char buf[sizeof(kMagic) + sizeof(ncclUniqueId)];
      const ssize_t n = read(fd, buf, sizeof(buf));
      close(fd);
      if (n == (ssize_t)sizeof(buf) && !memcmp(buf, kMagic, sizeof(kMagic))) {
        memcpy(id, buf + sizeof(kMagic), sizeof(ncclUniqueId));
        return true;
      }
      *err = std::string(path) + " is not a pbsim RCCL rendezvous file";
      return false;
    }
    if (now_s() - t0 > timeout_s) {
      *err = "rank 0 did not publish " + std::string(path) + " within " + std::to_string((int)timeout_s) + " s";
      return false;
    }
    struct timespec ts = {0, 2000000};
    nanosleep(&ts, nullptr);
  }
}

}  // namespace
}  // namespace pbsim

using namespace pbsim;

extern "C" {

int64_t pbsim_rccl_unique_id(void *id, int64_t cap) {
  if (!id || cap < (int64_t)sizeof(ncclUniqueId)) return (int64_t)sizeof(ncclUniqueId);
  std::string err;
  ncclUniqueId u;
  if (!rccl_unique_id(&u, &err)) {
    (void)fail("pbsim_rccl_unique_id: " + err);
    return 0;
  }
  memcpy(id, &u, sizeof(u));
  return (int64_t)sizeof(u);
}

pbsim_comm *pbsim_rccl_comm_create(const void *id, int64_t id_bytes, int32_t rank, int32_t world, int32_t device) {
  if (!id || id_bytes != (int64_t)sizeof(ncclUniqueId)) {
    (void)fail("pbsim_rccl_comm_create: the id is the " + std::to_string(sizeof(ncclUniqueId)) + " bytes of pbsim_rccl_unique_id");
    return nullptr;
  }
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  NativeComm *nc = new NativeComm();
  std::string err;
  if (!rccl_init_rank(u, rank, world, device, &nc->rank, &err)) {
    (void)fail("pbsim_rccl_comm_create (rank " + std::to_string(rank) + " of " + std::to_string(world) + ", device " +
              std::to_string(device) + "): " + err);
    delete nc;
    return nullptr;
  }
  nc->comm = rccl_comm(&nc->rank);
  return &nc->comm;
}

pbsim_comm *pbsim_rccl_comm_create_file(const char *path, int32_t rank, int32_t world, int32_t device) {
  if (!path || !*path) {
    (void)fail("pbsim_rccl_comm_create_file: no rendezvous path");
    return nullptr;
  }
  std::string err;
  ncclUniqueId u;
  if (rank == 0) {
    if (!rccl_unique_id(&u, &err) || !publish_id(path, u, &err)) {
      (void)fail("pbsim_rccl_comm_create_file: " + err);
      return nullptr;
    }
  } else {
    const char *e = getenv("PBSIM_RENDEZVOUS_TIMEOUT_S");
    if (!fetch_id(path, &u, e && atof(e) > 0 ? atof(e) : 120.0, &err)) {
      (void)fail("pbsim_rccl_comm_create_file: " + err);
      return nullptr;
    }
  }
  pbsim_comm *cm = pbsim_rccl_comm_create(&u, (int64_t)sizeof(u), rank, world, device);
  // ncclCommInitRank is collective: when it returns on rank 0, every rank has read the id -- the file goes, so that a later
  // launch that reuses the path cannot pick up this launch's id (a crash in between leaves it: use a path per launch)
  if (rank == 0) unlink(path);
  return cm;
}

int pbsim_rccl_comm_info(const pbsim_comm *comm, int64_t out[4]) {
  if (!comm || comm->all_gather_i64 != rccl_all_gather) {
    (void)fail("pbsim_rccl_comm_info: not a communicator of pbsim_rccl_comm_create");
    return PBSIM_FAILED;
  }
  const RcclRank *r = (const RcclRank *)comm->user;
  int count = -1, urank = -1;
  if (r->comm && r->api->CommCount) (void)r->api->CommCount(r->comm, &count);
  if (r->comm && r->api->CommUserRank) (void)r->api->CommUserRank(r->comm, &urank);
  out[0] = count;  // ranks RCCL itself counts in the communicator
  out[1] = urank;
  out[2] = r->device;
  out[3] = r->collectives;
  return PBSIM_SUCCEEDED;
}

void pbsim_rccl_comm_destroy(pbsim_comm *comm) {
  if (!comm || comm->all_gather_i64 != rccl_all_gather) return;
  NativeComm *nc = reinterpret_cast<NativeComm *>(comm);
  RcclRank &r = nc->rank;
  (void)hipSetDevice(r.device);
  if (r.stream && r.comm) (void)hipStreamSynchronize(r.stream);
  if (r.comm) (void)r.api->CommDestroy(r.comm);
  if (r.d_send) (void)hipFree(r.d_send);
  if (r.d_recv) (void)hipFree(r.d_recv);
  if (r.h_send) (void)hipHostFree(r.h_send);
  if (r.h_recv) (void)hipHostFree(r.h_recv);
  if (r.stream) (void)hipStreamDestroy(r.stream);
  delete nc;
}

}  // extern "C"
