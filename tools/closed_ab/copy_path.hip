// which engine carries a pinned D2H hipMemcpyAsync: run plain and under `rocprofv3 --kernel-trace` with AMD_LOG_LEVEL=4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void k_touch(unsigned *p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = (unsigned)i; }
int main() {
  const size_t bytes = 256u << 20;
  void *d, *h;
  hipMalloc(&d, bytes);
  hipHostMalloc(&h, bytes, hipHostMallocDefault);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  k_touch<<<(unsigned)(bytes / 4 / 256), 256, 0, s>>>((unsigned *)d, bytes / 4);
  hipStreamSynchronize(s);
  for (int rep = 0; rep < 3; rep++) {
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 8; i++) hipMemcpyAsync((char *)h + (size_t)i * (bytes / 8), (char *)d + (size_t)i * (bytes / 8), bytes / 8, hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("D2H 8 x 32 MB: %.2f ms = %.1f GB/s\n", dt * 1e3, bytes / dt / 1e9);
  }
  return 0;
}
