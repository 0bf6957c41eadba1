// main.cpp -- the `pbsim` binary: the reference's command line on one or several MI355X.
//   pbsim ... [--device N]            one GPU: pbsim_cli_main on the calling thread
//   pbsim ... --devices 0,1,2,3       one rank per listed GPU, one host thread each (a GPU may be listed more than once:
//                                     several contexts on one device, the plumbing check of a single-GPU box);
//                                     --comm host (default): host barrier + GPU-to-GPU peer copies (thread_comm.h)
//                                     --comm rccl: RCCL communicator over xGMI for C1 / C2 / C3 (rccl_comm.h; distinct GPUs)
//   pbsim ... --rank R --world N --rendezvous FILE [--device D]
//                                     one PROCESS per GPU (started N times by a shell loop, mpirun, srun ..): the ranks meet in an
//                                     RCCL communicator made with ncclCommInitRank, rank 0's id published through FILE
//                                     (pbsim_rccl_comm_create_file); device D defaults to R.  Under torchrun / mpirun / srun
//                                     --rendezvous FILE alone will do: rank, world and the node-local rank (the device) are
//                                     taken from RANK / WORLD_SIZE / LOCAL_RANK, OMPI_COMM_WORLD_*, PMI_*, SLURM_*
//   pbsim ... --processes N          the same, started by this binary itself: N children of it (posix_spawn of /proc/self/exe,
//                                     before this process has made any HIP call), child i = --rank i --world N on GPU i, a
//                                     rendezvous file of the launch's own under /dev/shm (or $TMPDIR); exit status = the first
//                                     child's that is not 0
// Every rank runs the same pbsim_cli_main(argv): the job is deterministic in the values the ranks exchange, so they stay
// in lockstep; rank 0 prints the report and creates the files, every rank writes its own byte ranges.
#include <hip/hip_runtime.h>
#include <spawn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

#include <initializer_list>
#include <signal.h>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pbsim3_amd.h"
#include "rccl_comm.h"
#include "thread_comm.h"

int main(int argc, char **argv) {
  std::vector<int> devices;
  std::string comm_kind = "host";
  int proc_rank = -1, proc_world = 0, proc_device = -1;  // --rank / --world / --device: this process is ONE rank of several
  int n_processes = 0;                                    // --processes N: this process starts N such ranks and waits for them
  std::string rendezvous;
  bool selftest = false;  // --comm-selftest: run C1 / C2 / C3 once on the communicator of --devices / --comm and check the values
  for (int i = 1; i < argc; i++) {
    const char *a = argv[i];
    const char *v = NULL;
    if (!strncmp(a, "--devices=", 10)) v = a + 10;
    else if (!strcmp(a, "--devices") && i + 1 < argc) v = argv[i + 1];
    if (v) {
      devices.clear();
      for (const char *p = v; *p;) {
        char *e = NULL;
        const long d = strtol(p, &e, 10);
        if (e == p || d < 0) {
          fprintf(stderr, "ERROR (devices: %s): a comma-separated list of GPU indices.\n", v);
          return 255;
        }
        devices.push_back((int)d);
        p = (*e == ',') ? e + 1 : e;
        if (*e && *e != ',') {
          fprintf(stderr, "ERROR (devices: %s): a comma-separated list of GPU indices.\n", v);
          return 255;
        }
      }
    }
    if (!strncmp(a, "--comm=", 7)) comm_kind = a + 7;
    else if (!strcmp(a, "--comm") && i + 1 < argc) comm_kind = argv[i + 1];
    if (!strcmp(a, "--comm-selftest")) selftest = true;
    if (!strcmp(a, "--processes") && i + 1 < argc) n_processes = atoi(argv[i + 1]);
    if (!strcmp(a, "--rank") && i + 1 < argc) proc_rank = atoi(argv[i + 1]);
    if (!strcmp(a, "--world") && i + 1 < argc) proc_world = atoi(argv[i + 1]);
    if (!strcmp(a, "--rendezvous") && i + 1 < argc) rendezvous = argv[i + 1];
    if (!strcmp(a, "--device") && i + 1 < argc) proc_device = atoi(argv[i + 1]);
  }
  if (n_processes > 0) {
    if (!devices.empty() || proc_rank >= 0 || proc_world > 0 || !rendezvous.empty()) {
      fprintf(stderr, "ERROR: --processes N starts the ranks itself: no --devices / --rank / --world / --rendezvous beside it.\n");
      return 255;
    }
    const char *dir = access("/dev/shm", W_OK) == 0 ? "/dev/shm" : (getenv("TMPDIR") ? getenv("TMPDIR") : "/tmp");
    const std::string rdv = std::string(dir) + "/pbsim_rdv_" + std::to_string((long)getpid());
    unlink(rdv.c_str());
    std::vector<pid_t> kids;
    for (int r = 0; r < n_processes; r++) {
      std::vector<std::string> a;
      for (int i = 0; i < argc; i++) {
        if (!strcmp(argv[i], "--processes")) {
          i++;
          continue;
        }
        a.push_back(argv[i]);
      }
      for (const std::string &x : {std::string("--rank"), std::to_string(r), std::string("--world"), std::to_string(n_processes),
                                   std::string("--rendezvous"), rdv})
        a.push_back(x);
      std::vector<char *> av;
      for (std::string &x : a) av.push_back(&x[0]);
      av.push_back(NULL);
      pid_t pid = 0;
      extern char **environ;
      if (posix_spawn(&pid, "/proc/self/exe", NULL, NULL, av.data(), environ) != 0) {
        fprintf(stderr, "ERROR: cannot start rank %d (posix_spawn of /proc/self/exe).\n", r);
        for (pid_t k : kids) kill(k, SIGTERM);
        return 255;
      }
      kids.push_back(pid);
    }
    int rc = 0;
    for (pid_t k : kids) {
      int st = 0;
      if (waitpid(k, &st, 0) < 0 || !WIFEXITED(st)) st = 255 << 8;
      if (!rc && WEXITSTATUS(st)) rc = WEXITSTATUS(st);
    }
    unlink(rdv.c_str());
    return rc;
  }
  if (!rendezvous.empty() && proc_rank < 0 && proc_world <= 0) {
    // under a launcher (torchrun, mpirun, srun) rank, world and the node-local rank come from its environment
    auto env_int = [](std::initializer_list<const char *> names, int fallback) {
      for (const char *n : names)
        if (const char *v = getenv(n))
          if (*v) return atoi(v);
      return fallback;
    };
    proc_rank = env_int({"RANK", "OMPI_COMM_WORLD_RANK", "PMI_RANK", "SLURM_PROCID"}, -1);
    proc_world = env_int({"WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE", "SLURM_NTASKS"}, 0);
    if (proc_device < 0) proc_device = env_int({"LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID", "SLURM_LOCALID"}, -1);
  }
  if (proc_world > 0 || proc_rank >= 0 || !rendezvous.empty()) {
    if (proc_world < 1 || proc_rank < 0 || proc_rank >= proc_world || rendezvous.empty() || !devices.empty()) {
      fprintf(stderr, "ERROR: one process per GPU takes --rank R --world N --rendezvous FILE (0 <= R < N) and no --devices.\n");
      return 255;
    }
    if (proc_device < 0) proc_device = proc_rank;
    char what[256];
    pbsim_bind_host_to_device(proc_device, what, sizeof(what));  // before the first HIP call of the process
    pbsim_comm *cm = pbsim_rccl_comm_create_file(rendezvous.c_str(), proc_rank, proc_world, proc_device);
    if (!cm) {
      fprintf(stderr, "ERROR: %s\n", pbsim_last_error());
      return 255;
    }
    // the launcher's own options go no further than this file: pbsim_cli_main sees the reference's command line
    std::vector<char *> args;
    for (int i = 0; i < argc; i++) {
      if (!strcmp(argv[i], "--rank") || !strcmp(argv[i], "--world") || !strcmp(argv[i], "--rendezvous") || !strcmp(argv[i], "--device")) {
        i++;
        continue;
      }
      args.push_back(argv[i]);
    }
    args.push_back(NULL);
    int rc;
    if (selftest) {
      int64_t send[3] = {proc_rank, 10 * proc_rank, -proc_rank}, red[2] = {proc_rank, 1};
      std::vector<int64_t> recv((size_t)proc_world * 3, 77);
      rc = cm->all_gather_i64(cm->user, send, 3, recv.data()) && cm->all_reduce_i64(cm->user, red, 2, PBSIM_OP_SUM) ? 0 : 255;
      for (int q = 0; q < proc_world && !rc; q++)
        if (recv[(size_t)q * 3] != q || recv[(size_t)q * 3 + 1] != 10 * q || recv[(size_t)q * 3 + 2] != -q) rc = 255;
      if (red[0] != (int64_t)proc_world * (proc_world - 1) / 2 || red[1] != proc_world) rc = 255;
      int64_t info[4] = {0, 0, 0, 0};
      pbsim_rccl_comm_info(cm, info);
      fprintf(stderr, "comm selftest (rccl, rank %d of %d processes, RCCL counts %lld): %s\n", proc_rank, proc_world, (long long)info[0],
              rc ? "FAILED" : "ok");
    } else {
      rc = pbsim_cli_main((int)args.size() - 1, args.data(), cm, proc_device) & 255;
    }
    pbsim_rccl_comm_destroy(cm);
    return rc;
  }
  if (comm_kind != "host" && comm_kind != "rccl") {
    fprintf(stderr, "ERROR (comm: %s): host or rccl.\n", comm_kind.c_str());
    return 255;
  }
  if (devices.empty() && !selftest) {
    // one rank, and the process ends behind it: the context's pools go back to the driver with the process (cli.cpp)
    // PBSIM_CLI_LEAVE_CONTEXT=0: an ordinary exit instead -- the context is destroyed, atexit handlers and static destructors
    // run: what a profiler (rocprofv3 writes its kernel_stats.csv at exit), gcov or a sanitizer's exit report need
    if (!getenv("PBSIM_CLI_LEAVE_CONTEXT")) setenv("PBSIM_CLI_LEAVE_CONTEXT", "1", 1);
    const bool leave = atoi(getenv("PBSIM_CLI_LEAVE_CONTEXT")) != 0;
    const int rc = pbsim_cli_main(argc, argv, NULL, -1) & 255;
    fflush(stdout);
    fflush(stderr);
    if (!leave) return rc;
    _exit(rc);  // every file is closed; nothing is left but tearing the runtime's threads and mappings down one by one
  }
  if (devices.empty()) devices.push_back(0);

  const int world = (int)devices.size();
  pbsim::ThreadCommShared shared(devices);
  std::vector<pbsim::ThreadCommRank> ranks((size_t)world);
  std::vector<pbsim_comm> comms((size_t)world);
  std::vector<pbsim::RcclRank> rccl((size_t)world);
  for (int r = 0; r < world; r++) {
    ranks[(size_t)r] = pbsim::ThreadCommRank{&shared, r};
    comms[(size_t)r] = pbsim::thread_comm(&ranks[(size_t)r]);
  }
  if (comm_kind == "rccl") {
    std::string err;
    if (!pbsim::rccl_init_all(devices, &rccl, &err)) {
      fprintf(stderr, "ERROR: --comm rccl: %s\n", err.c_str());
      return 255;
    }
    for (int r = 0; r < world; r++) comms[(size_t)r] = pbsim::rccl_comm(&rccl[(size_t)r]);
  }
  if (selftest) {
    std::vector<int> bad((size_t)world, 0);
    auto one = [&](int r) {
      const pbsim_comm &cm = comms[(size_t)r];
      (void)hipSetDevice(devices[(size_t)r]);
      // C3: all-gather (rank-major)
      int64_t send[3] = {r, 10 * r, -r};
      std::vector<int64_t> recv((size_t)world * 3, 77);
      if (!cm.all_gather_i64(cm.user, send, 3, recv.data())) bad[(size_t)r] |= 1;
      for (int q = 0; q < world; q++)
        if (recv[(size_t)q * 3] != q || recv[(size_t)q * 3 + 1] != 10 * q || recv[(size_t)q * 3 + 2] != -q) bad[(size_t)r] |= 1;
      // C2: all-reduce sum / min / max
      const int ops[3] = {PBSIM_OP_SUM, PBSIM_OP_MIN, PBSIM_OP_MAX};
      const int64_t want[3] = {(int64_t)world * (world - 1) / 2, 0, world - 1};
      for (int k = 0; k < 3; k++) {
        std::vector<int64_t> buf(1000, r);
        if (!cm.all_reduce_i64(cm.user, buf.data(), (int64_t)buf.size(), ops[k])) bad[(size_t)r] |= 2;
        for (int64_t v : buf)
          if (v != want[k]) bad[(size_t)r] |= 2;
      }
      // C1: broadcast of device memory from the last rank
      const int root = world - 1;
      const size_t nbytes = 1 << 20;
      void *d = NULL;
      std::vector<unsigned char> h(nbytes);
      if (hipMalloc(&d, nbytes) != hipSuccess) bad[(size_t)r] |= 4;
      for (size_t i = 0; i < nbytes; i++) h[i] = (unsigned char)((i * 131 + (size_t)r) & 255);
      (void)hipMemcpy(d, h.data(), nbytes, hipMemcpyHostToDevice);
      (void)hipDeviceSynchronize();
      if (!cm.broadcast || !cm.broadcast(cm.user, d, (int64_t)nbytes, root, 1)) bad[(size_t)r] |= 4;
      (void)hipMemcpy(h.data(), d, nbytes, hipMemcpyDeviceToHost);
      for (size_t i = 0; i < nbytes; i++)
        if (h[i] != (unsigned char)((i * 131 + (size_t)root) & 255)) bad[(size_t)r] |= 4;
      (void)hipFree(d);
    };
    std::vector<std::thread> th;
    for (int r = 1; r < world; r++) th.emplace_back(one, r);
    one(0);
    for (std::thread &t : th) t.join();
    if (comm_kind == "rccl") pbsim::rccl_destroy_all(&rccl);
    int any = 0;
    for (int r = 0; r < world; r++) any |= bad[(size_t)r];
    fprintf(stderr, "comm selftest (%s, %d rank%s): all-gather %s, all-reduce %s, device broadcast %s\n", comm_kind.c_str(), world,
            world == 1 ? "" : "s", (any & 1) ? "FAILED" : "ok", (any & 2) ? "FAILED" : "ok", (any & 4) ? "FAILED" : "ok");
    return any ? 255 : 0;
  }
  std::vector<int> rc((size_t)world, 0);
  std::vector<std::thread> th;
  for (int r = 1; r < world; r++)
    th.emplace_back([&, r]() { rc[(size_t)r] = pbsim_cli_main(argc, argv, &comms[(size_t)r], devices[(size_t)r]); });
  rc[0] = pbsim_cli_main(argc, argv, &comms[0], devices[0]);
  for (std::thread &t : th) t.join();
  if (comm_kind == "rccl") pbsim::rccl_destroy_all(&rccl);
  for (int r = 0; r < world; r++)
    if (rc[(size_t)r]) return rc[(size_t)r] & 255;
  return 0;
}
