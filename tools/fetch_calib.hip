// fetch_calib.hip -- what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the access patterns of the walk kernels,
// against byte counts known by construction (MI355X_MICROARCH.md, "HBM": FETCH_SIZE reports half the bytes of a wide coalesced
// streaming read; "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//   k_stream16   every lane reads 16 B, consecutive lanes consecutive addresses: N bytes, each once          (the guide's case)
//   k_gather8    every lane reads 8 B at the start of its own pseudo-random 64-byte sector of a 4 GiB buffer   (the walk's
//                reference window: one sector per refill, no reuse: N loads = N x 64 B of sectors, N x 8 B used)
//   k_store4     every lane stores one dword, a wave a full 256-byte line, nontemporal                         (the walk's MAF rows)
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -o /tmp/fetch_calib tools/fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- /tmp/fetch_calib   (and WRITE_SIZE in its own pass); tools/pmc_round.sh does.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void k_stream16(const uint4 *src, size_t n16, unsigned long long *sink) {
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = src[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x1234567887654321ull) *sink = acc;
}

__global__ void k_gather8(const uint8_t *src, size_t n_sectors, size_t n_loads, unsigned long long *sink) {
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_loads; i += (size_t)gridDim.x * blockDim.x) {
    const size_t s = (size_t)((i * 0x9E3779B1ull) & (n_sectors - 1));  // n_sectors = 2^26, odd multiplier: a permutation -- every load its own sector
    acc += *reinterpret_cast<const unsigned long long *>(src + s * 64);
  }
  if (acc == 0x1234567887654321ull) *sink = acc;
}

__global__ void k_store4(uint32_t *dst, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store((uint32_t)i, dst + i);
}

int main() {
  const size_t bytes = 4ull << 30;
  uint8_t *buf = nullptr;
  unsigned long long *sink = nullptr;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) return 1;
  (void)hipMemset(buf, 1, bytes);
  (void)hipDeviceSynchronize();
  const size_t n_loads = 64ull << 20;  // 64 Mi loads x 64-byte sectors = 4 GiB of sectors, 512 MiB of bytes used
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k_stream16, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const uint4 *>(buf), bytes / 16, sink);
    hipLaunchKernelGGL(k_gather8, dim3(4096), dim3(256), 0, 0, buf, bytes / 64, n_loads, sink);
    hipLaunchKernelGGL(k_store4, dim3(4096), dim3(256), 0, 0, reinterpret_cast<uint32_t *>(buf), bytes / 4);
  }
  (void)hipDeviceSynchronize();
  printf("fetch_calib: k_stream16 reads %zu bytes; k_gather8 makes %zu loads of 8 bytes, one 64-byte sector each (%zu bytes of sectors); "
         "k_store4 writes %zu bytes\n", bytes, n_loads, n_loads * 64, bytes);
  return 0;
}
