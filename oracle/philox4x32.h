/* oracle/philox4x32.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy
 * as 1, 2, 3", SC'11), restated from the published algorithm.  This copy
 * belongs to the oracle and to the reference shim; the product has its own
 * definition in pbsim3_amd/csrc/philox.h.  Both are pinned by the Random123
 * known-answer vectors in tests/test_philox.py.
 */
#ifndef PBSIM_ORACLE_PHILOX4X32_H
#define PBSIM_ORACLE_PHILOX4X32_H
#include <stdint.h>

#define ORC_PHILOX_M0 0xD2511F53u
#define ORC_PHILOX_M1 0xCD9E8D57u
#define ORC_PHILOX_W0 0x9E3779B9u
#define ORC_PHILOX_W1 0xBB67AE85u

static inline void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2],
                                     uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)ORC_PHILOX_M0 * c0;
    uint64_t p1 = (uint64_t)ORC_PHILOX_M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += ORC_PHILOX_W0;
    k1 += ORC_PHILOX_W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* ---- keyed draw contract shared by the oracle and the reference shim ------
 * key  = (seed, stream)            stream: ORC_STREAM_HDR | ORC_STREAM_WALK
 * ctr  = (event, pass<<4 | sub, read, unit)
 * draw = word[slot] >> 1           (31 bits, like glibc rand())
 * DESIGN.md "RNG contract" is the normative text. */
#define ORC_STREAM_HDR  0x48445221u /* "HDR!" */
#define ORC_STREAM_WALK 0x57414C4Bu /* "WALK" */

static inline uint32_t orc_keyed_draw(uint32_t seed, uint32_t stream, uint32_t unit,
                                      uint32_t read, uint32_t pass, uint32_t event,
                                      uint32_t sub, uint32_t slot) {
  uint32_t ctr[4] = {event, (pass << 4) | sub, read, unit};
  uint32_t key[2] = {seed, stream};
  uint32_t out[4];
  orc_philox4x32_10(ctr, key, out);
  return out[slot] >> 1;
}
#endif
