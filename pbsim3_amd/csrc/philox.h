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

// a ^ b ^ c: one v_bitop3_b32 on gfx950 (truth table 0x96), two XORs on the host
PB_HD uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
  return a ^ b ^ c;
#endif
}

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

// ---- the walk kernels' form of the same block -------------------------------
// For the walk stream the counter is (event, pass<<4|sub, read, unit): only
// `event` changes from step to step and it is wave-uniform.  Round 1 multiplies
// M1*read (constant per lane) and M0*event (uniform -> scalar unit); round 2
// multiplies M0*n0 (constant per lane) and M1*n2 (uniform).  So the first two of
// the ten rounds cost two vector XORs per block instead of four 32x32->64
// multiplies: WalkLane holds the per-lane constants, walk_block_fast() finishes.
struct WalkLane {
  uint32_t n1;       // lo(M1*read)
  uint32_t b_hi;     // hi(M0*n0)
  uint32_t b_lo;     // lo(M0*n0)
};

PB_HD WalkLane walk_lane(uint32_t seed, uint32_t read, uint32_t pass, uint32_t sub) {
  const uint64_t a = (uint64_t)kPhiloxM1 * read;
  const uint32_t n0 = (uint32_t)(a >> 32) ^ ((pass << 4) | sub) ^ seed;
  const uint64_t b = (uint64_t)kPhiloxM0 * n0;
  return WalkLane{(uint32_t)a, (uint32_t)(b >> 32), (uint32_t)b};
}

// `event` and `unit` must be wave-uniform for the scalar part to stay on the scalar unit.
// walk_block_raw: the block's four 32-bit words as Philox made them (the draw is word >> 1: the lane walkers fold that shift
// into the remainder they take of a word, kernels.hip mod1000_raw); walk_block_fast: the four draws.
PB_HD U4 walk_block_raw(const WalkLane &l, uint32_t seed, uint32_t unit, uint32_t event) {
  // round 1, uniform half: (n2, n3) from M0*event
  const uint64_t p = (uint64_t)kPhiloxM0 * event;
  const uint32_t n2 = (uint32_t)(p >> 32) ^ unit ^ kStreamWalk;
  const uint32_t n3 = (uint32_t)p;
  // round 2 (keys bumped once)
  const uint32_t k0 = seed + kPhiloxW0, k1 = kStreamWalk + kPhiloxW1;
  const uint64_t q = (uint64_t)kPhiloxM1 * n2;
  uint32_t c0 = ((uint32_t)(q >> 32) ^ k0) ^ l.n1;
  uint32_t c1 = (uint32_t)q;
  uint32_t c2 = (n3 ^ k1) ^ l.b_hi;
  uint32_t c3 = l.b_lo;
  uint32_t ka = seed + 2u * kPhiloxW0, kb = kStreamWalk + 2u * kPhiloxW1;
#pragma unroll
  for (int r = 2; r < 10; r++) {
    const uint64_t p0 = (uint64_t)kPhiloxM0 * c0;
    const uint64_t p1 = (uint64_t)kPhiloxM1 * c2;
    const uint32_t m0 = xor3((uint32_t)(p1 >> 32), c1, ka);
    const uint32_t m2 = xor3((uint32_t)(p0 >> 32), c3, kb);
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = m0;
    c2 = m2;
    ka += kPhiloxW0;
    kb += kPhiloxW1;
  }
  return U4{c0, c1, c2, c3};
}
PB_HD U4 walk_block_fast(const WalkLane &l, uint32_t seed, uint32_t unit, uint32_t event) {
  const U4 r = walk_block_raw(l, seed, unit, event);
  return U4{r.x >> 1, r.y >> 1, r.z >> 1, r.w >> 1};
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
