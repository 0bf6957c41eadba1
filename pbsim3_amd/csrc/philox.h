// philox.h -- Philox4x32-10 for host and gfx950 device code, and the keyed
// draw contract of DESIGN.md "RNG contract".
//
// The reference draws everything from one sequential glibc rand() stream
// (srand at pbsim.cpp:543, 111 `rand() % n` sites), which serialises reads.
// Here every draw is addressed by (unit, read, pass, event, sub-block, slot):
//   key = (seed, stream)        ctr = (event, pass<<4 | sub, read, unit)
//   draw = word[slot] >> 1      (31 bits, the range of glibc rand())
// and is then reduced with the same integer `% n` as the reference call site.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PB_HD __host__ __device__ __forceinline__
#else
#define PB_HD inline
#endif

namespace pbsim {

constexpr uint32_t kPhiloxM0 = 0xD2511F53u;
constexpr uint32_t kPhiloxM1 = 0xCD9E8D57u;
constexpr uint32_t kPhiloxW0 = 0x9E3779B9u;
constexpr uint32_t kPhiloxW1 = 0xBB67AE85u;

constexpr uint32_t kStreamHeader = 0x48445221u;  // "HDR!"
constexpr uint32_t kStreamWalk = 0x57414C4Bu;    // "WALK"

struct U4 {
  uint32_t x, y, z, w;
};

PB_HD U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const uint64_t p0 = (uint64_t)kPhiloxM0 * c0;
    const uint64_t p1 = (uint64_t)kPhiloxM1 * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += kPhiloxW0;
    k1 += kPhiloxW1;
  }
  return U4{c0, c1, c2, c3};
}

// walk-stream block of one HMM event; words are already shifted to 31 bits
PB_HD U4 walk_block(uint32_t seed, uint32_t unit, uint32_t read, uint32_t pass, uint32_t event, uint32_t sub) {
  U4 r = philox4x32_10(event, (pass << 4) | sub, read, unit, seed, kStreamWalk);
  r.x >>= 1;
  r.y >>= 1;
  r.z >>= 1;
  r.w >>= 1;
  return r;
}

PB_HD U4 header_block(uint32_t seed, uint32_t unit, uint32_t read) {
  U4 r = philox4x32_10(0u, 0u, read, unit, seed, kStreamHeader);
  r.x >>= 1;
  r.y >>= 1;
  r.z >>= 1;
  r.w >>= 1;
  return r;
}

}  // namespace pbsim
