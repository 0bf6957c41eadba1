// numa_bind.cpp -- host placement of a rank: the thread that drives GPU `device`, the threads it starts (the delivery worker
// and its lane threads) and the pinned staging they allocate belong on the NUMA node the GPU's PCIe slot hangs off.  At eight
// ranks every GPU pushes ~45 GB/s of members into host memory; a staging buffer on the other socket crosses the inter-socket
// fabric once per byte and the link rate a single rank measured is no longer what eight get.
//
// No HIP call here (this must run BEFORE the runtime starts its own threads): the GPU's PCI address comes from the KFD topology
// in sysfs, whose GPU nodes are in the order HIP enumerates them (ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES are applied when
// they are plain index lists; anything else -- UUIDs -- leaves the process unbound), the node from the PCI device's
// `numa_node` / `local_cpulist`.
#include <dirent.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/pbsim3_amd.h"

namespace {

std::string sys_root() {
  const char *r = getenv("PBSIM_SYSFS_ROOT");  // tests: a fake tree
  return r ? r : "";
}

bool read_file(const std::string &path, std::string *out) {
  FILE *f = fopen(path.c_str(), "r");
  if (!f) return false;
  char buf[4096];
  out->clear();
  size_t k;
  while ((k = fread(buf, 1, sizeof buf, f)) > 0) out->append(buf, k);
  fclose(f);
  return true;
}

// "0-15,64-79" -> cpu numbers
std::vector<int> parse_cpulist(const std::string &s) {
  std::vector<int> v;
  const char *p = s.c_str();
  while (*p) {
    char *e = nullptr;
    const long a = strtol(p, &e, 10);
    if (e == p) break;
    long b = a;
    p = e;
    if (*p == '-') {
      b = strtol(p + 1, &e, 10);
      p = e;
    }
    for (long c = a; c <= b && c < 4096; c++) v.push_back((int)c);
    while (*p == ',' || *p == '\n' || *p == ' ') p++;
  }
  return v;
}

// plain index list "0,2,3" -> indices; false when the variable holds anything else (UUIDs)
bool parse_index_list(const char *s, std::vector<int> *out) {
  out->clear();
  for (const char *p = s; *p;) {
    char *e = nullptr;
    const long a = strtol(p, &e, 10);
    if (e == p || a < 0) return false;
    out->push_back((int)a);
    p = e;
    if (*p == ',') p++;
    else if (*p) return false;
  }
  return true;
}

struct GpuNode {
  long domain = 0, location = 0;
};

// GPU nodes of the KFD topology in node order (= the order ROCr, hence HIP, enumerates them)
bool kfd_gpus(std::vector<GpuNode> *out) {
  const std::string base = sys_root() + "/sys/class/kfd/kfd/topology/nodes";
  std::vector<int> ids;
  DIR *d = opendir(base.c_str());
  if (!d) return false;
  while (struct dirent *e = readdir(d)) {
    char *end = nullptr;
    const long id = strtol(e->d_name, &end, 10);
    if (end != e->d_name && *end == 0) ids.push_back((int)id);
  }
  closedir(d);
  std::sort(ids.begin(), ids.end());
  for (int id : ids) {
    std::string props;
    if (!read_file(base + "/" + std::to_string(id) + "/properties", &props)) continue;
    long simd = 0, domain = 0, location = 0;
    const char *p = props.c_str();
    while (*p) {
      char key[64];
      long long val;
      if (sscanf(p, "%63s %lld", key, &val) == 2) {
        if (!strcmp(key, "simd_count")) simd = (long)val;
        else if (!strcmp(key, "domain")) domain = (long)val;
        else if (!strcmp(key, "location_id")) location = (long)val;
      }
      const char *nl = strchr(p, '\n');
      if (!nl) break;
      p = nl + 1;
    }
    if (simd > 0) out->push_back(GpuNode{domain, location});
  }
  return !out->empty();
}

}  // namespace

extern "C" int pbsim_bind_host_to_device(int device, char *what, int64_t cap) {
  auto say = [&](const std::string &m) {
    if (what && cap > 0) snprintf(what, (size_t)cap, "%s", m.c_str());
  };
  say("");
  const char *off = getenv("PBSIM_NUMA_BIND");
  if (off && !strcmp(off, "0")) return PBSIM_SUCCEEDED;
  if (device < 0) return PBSIM_SUCCEEDED;
  std::vector<GpuNode> gpus;
  if (!kfd_gpus(&gpus)) return PBSIM_SUCCEEDED;  // no KFD topology (no GPU driver): nothing to bind to
  // visible-device remapping: ROCr filters first, HIP indexes into what ROCr left
  // HIP takes its list from HIP_VISIBLE_DEVICES, CUDA_VISIBLE_DEVICES (torch launchers set that one) or GPU_DEVICE_ORDINAL: all
  // the same mask to the runtime.  When several are set and disagree, which one wins is the runtime's business -- binding to
  // the wrong GPU's node is worse than not binding (every byte would cross the socket link), so stay unbound.
  int index = device;
  const char *hip_list = nullptr;
  for (const char *var : {"HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL"}) {
    const char *v = getenv(var);
    if (!v || !*v) continue;
    if (hip_list && strcmp(hip_list, v) != 0) return PBSIM_SUCCEEDED;
    hip_list = v;
  }
  for (const char *v : {hip_list, (const char *)getenv("ROCR_VISIBLE_DEVICES")}) {
    if (!v || !*v) continue;
    std::vector<int> list;
    if (!parse_index_list(v, &list)) return PBSIM_SUCCEEDED;
    if (index >= (int)list.size()) return PBSIM_SUCCEEDED;
    index = list[(size_t)index];
  }
  if (index >= (int)gpus.size()) return PBSIM_SUCCEEDED;
  const GpuNode &g = gpus[(size_t)index];
  char bdf[64];
  snprintf(bdf, sizeof bdf, "%04lx:%02lx:%02lx.%lx", g.domain & 0xffff, (g.location >> 8) & 0xff, (g.location >> 3) & 0x1f, g.location & 7);
  const std::string dev = sys_root() + "/sys/bus/pci/devices/" + bdf;
  std::string node_s, cpus_s;
  if (!read_file(dev + "/numa_node", &node_s) || !read_file(dev + "/local_cpulist", &cpus_s)) return PBSIM_SUCCEEDED;
  const int node = atoi(node_s.c_str());
  if (node < 0) return PBSIM_SUCCEEDED;  // one node, or the platform does not say
  const std::vector<int> cpus = parse_cpulist(cpus_s);
  if (cpus.empty()) return PBSIM_SUCCEEDED;
  cpu_set_t cur, want;
  CPU_ZERO(&want);
  if (sched_getaffinity(0, sizeof cur, &cur) != 0) return PBSIM_SUCCEEDED;
  int n = 0;
  for (int c : cpus)
    if (c < CPU_SETSIZE && CPU_ISSET(c, &cur)) {
      CPU_SET(c, &want);
      n++;
    }
  if (n == 0) return PBSIM_SUCCEEDED;  // the launcher pinned us elsewhere: its choice
  if (!sys_root().empty()) {           // a fake tree: report what would be done, change nothing
    say(std::string("gpu ") + std::to_string(device) + " (" + bdf + "): numa node " + std::to_string(node) + ", " + std::to_string(n) + " cpus (dry run)");
    return PBSIM_SUCCEEDED;
  }
  if (sched_setaffinity(0, sizeof want, &want) != 0) return PBSIM_SUCCEEDED;
  // memory follows: prefer the node for every allocation of this thread and the threads it starts (MPOL_PREFERRED = 1; raw
  // syscall, libnuma is not a dependency).  Failure is harmless: first touch on the bound CPUs gives the same pages.
  unsigned long mask[16] = {0};
  if (node < (int)(sizeof mask * 8)) {
    mask[node / (8 * sizeof(long))] |= 1ul << (node % (8 * sizeof(long)));
#ifdef SYS_set_mempolicy
    (void)syscall(SYS_set_mempolicy, 1 /* MPOL_PREFERRED */, mask, sizeof mask * 8);
#endif
  }
  say(std::string("gpu ") + std::to_string(device) + " (" + bdf + "): numa node " + std::to_string(node) + ", " + std::to_string(n) + " cpus");
  return PBSIM_SUCCEEDED;
}
