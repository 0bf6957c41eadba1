// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the pbsim3 hot path.
//
//   K0  k_hp_*          upper-case + per-base homopolymer length   (pbsim.cpp:1035-1065)
//   K1  k_header_wgs    read header draw: length, accuracy, start  (pbsim.cpp:3793-3813)
//   Ks  k_sort_*        bucket tasks by (accuracy class, length) so a workgroup
//                       walks ONE class (its tables fit LDS) and the 64 lanes of a
//                       wave finish together; per-wave scratch extents
//   K2e k_walk_errhmm   the ERRHMM walk, one lane per (read, pass)  (pbsim.cpp:3835-3976)
//   K2q k_walk_qshmm    the QSHMM walk                              (pbsim.cpp:2209-2282)
//   K3  k_text_*        quota cut, exclusive scans, and FASTQ|SAM + MAF text
//                       written in read order                       (pbsim.cpp:3977-4078)
//
// Integer/byte work end to end: no MFMA, no floating point in the walk except
// the ordered double sum of quality error probabilities (pbsim.cpp:2309-2313).
// Scratch is wave-transposed and column-aligned: dword c (MAF columns 4c..4c+3)
// of lane l of a wave lives at region[(c*64 + l)*4], so every store of the walk
// is one contiguous 256-byte line per row per wave.  Rows: MAF read row (byte 0 =
// deleted column), MAF reference row, and for QSHMM the quality per column; the
// read sequence is the MAF read row with the deleted columns squeezed out (K3).
#include <hip/hip_runtime.h>
#include <type_traits>

#include "kernels.h"
#include "philox.h"

namespace pbsim {

namespace {

constexpr int kScanBlock = 256;  // single-workgroup scan kernels

// Wave priorities (s_setprio).  The walk of one slot shares the GPU with every other kernel of the other slot; its waves
// live for milliseconds, the others' for microseconds.  Without a raised priority the short kernels' waves rarely get to
// issue on a SIMD they share with walk waves (k_text_headers: 0.5 ms alone, 13 ms beside a QSHMM walk).
#ifndef PBSIM_WALK_PRIO
#define PBSIM_WALK_PRIO 1   // waves that carry the longest reads raise themselves (walk_priority)
#endif
#ifndef PBSIM_TEXT_PRIO
#define PBSIM_TEXT_PRIO 3   // every kernel that is not a walk
#endif
__device__ __forceinline__ void short_kernel_priority() {
  if (PBSIM_TEXT_PRIO) __builtin_amdgcn_s_setprio(PBSIM_TEXT_PRIO);
}
constexpr uint32_t kATGC = 0x43475441u;  // "ATGC" little-endian (mut.ins_nt / sub_nt_n, pbsim.cpp:5485-5486)

__device__ __forceinline__ uint32_t to_upper(uint32_t c) { return (c >= 'a' && c <= 'z') ? c - 32u : c; }

// The per-column scratch rows are written once by a walk and read once by the text emission: streaming (nontemporal)
// accesses keep them from evicting the reference lines the walk's gathers re-use from L2.
#ifndef PBSIM_NT
#define PBSIM_NT 1
#endif
__device__ __forceinline__ void scratch_store(uint32_t *p, uint32_t v) {
#if PBSIM_NT & 1
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
__device__ __forceinline__ uint32_t scratch_load(const uint32_t *p) {
#if PBSIM_NT & 2
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}

// revcomp()'s base map (pbsim.cpp:5853-5863): A<->T, G<->C, everything else unchanged
__device__ __forceinline__ uint32_t complement(uint32_t c) {
  uint32_t r = c;
  r = (c == 'A') ? 'T' : r;
  r = (c == 'T') ? 'A' : r;
  r = (c == 'G') ? 'C' : r;
  r = (c == 'C') ? 'G' : r;
  return r;
}

// the same map on four packed bytes (SWAR): 0x80 marks the bytes equal to k
__device__ __forceinline__ uint32_t eq_bytes(uint32_t w, uint32_t k) {
  const uint32_t z = w ^ (k * 0x01010101u);
  return ~((((z & 0x7f7f7f7fu) + 0x7f7f7f7fu) | z)) & 0x80808080u;
}
__device__ __forceinline__ uint32_t complement4(uint32_t w) {
  const uint32_t at = (eq_bytes(w, 'A') | eq_bytes(w, 'T')) >> 7;  // 0x01 per matching byte
  const uint32_t cg = (eq_bytes(w, 'C') | eq_bytes(w, 'G')) >> 7;
  return w ^ (at * 0x15u) ^ (cg * 0x04u);                          // 'A'^'T' = 0x15, 'C'^'G' = 0x04
}

// w % 3 of a 31-bit draw: 3 q == -q (mod 4) and the remainder is below 4, so w - 3 q == (w + q) & 3
__device__ __forceinline__ uint32_t mod3(uint32_t w) { return (w + __umulhi(w, 0x55555556u)) & 3u; }
// `draw % n` for the three table resolutions.  A 32 x 32 multiply (v_mul_hi_u32, v_mul_lo_u32, v_mad_u64_u32) issues at a
// quarter of the rate of an ordinary VALU instruction, a 24-bit one (v_mul_u32_u24) at the full rate, and the walks are
// bound by exactly that issue rate: the quotient comes from one multiply-high, the product quotient * n from the 24-bit
// multiplier (the remainder is below 2^24, so 24 bits of the difference are all of it, whatever the quotient's size).
__device__ __forceinline__ uint32_t mod100(uint32_t x) { return (x - __umul24(__umulhi(x, 0x51EB851Fu) >> 5, 100u)) & 0xffffffu; }
__device__ __forceinline__ uint32_t mod1000(uint32_t x) { return (x - __umul24(__umulhi(x, 0x10624DD3u) >> 6, 1000u)) & 0xffffffu; }
__device__ __forceinline__ uint32_t mod1e6(uint32_t x) { return (x - __umul24(__umulhi(x, 0x431BDE83u) >> 18, 1000000u)) & 0xffffffu; }
// The same remainders of a DRAW taken from the raw 32-bit Philox word w (draw = w >> 1, philox.h): (w >> 1) % n =
// (w - 2 n * floor(w / 2 n)) >> 1, and the 24-bit mask and the shift are ONE v_bfe_u32 -- a vector instruction less per
// remainder than shifting first (round 5; the kernel is bound by the number of vector instructions it issues).  The magic
// numbers are those of n with one more bit of shift; exact for every 32-bit w (error terms 112, 56 and 314 752 times 2^32 stay
// below 2^39, 2^38 and 2^51; all 2^32 words compared with `%` on the CPU when this was written).  The 24-bit multiplier sees the
// low 24 bits of the quotient, which is all the low 24 bits of the product depend on.
// (The two instructions are spelled out: left to itself the compiler proves that only 24 bits of the product are wanted, drops
// the 24-bit form for a 32 x 32 -> 64 multiply-add where the quotient may exceed 24 bits -- n = 100 -- and splits the bit-field
// extract into a shift and a mask.)
__device__ __forceinline__ uint32_t rem_raw(uint32_t w, uint32_t quo, uint32_t neg_2n) {
  uint32_t t, r;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(t) : "v"(quo), "s"(neg_2n), "v"(w));
  asm("v_bfe_u32 %0, %1, 1, 23" : "=v"(r) : "v"(t));
  return r;
}
__device__ __forceinline__ uint32_t mod100_raw(uint32_t w) { return rem_raw(w, __umulhi(w, 0x51EB851Fu) >> 6, 0x1000000u - 200u); }
__device__ __forceinline__ uint32_t mod1000_raw(uint32_t w) { return rem_raw(w, __umulhi(w, 0x10624DD3u) >> 7, 0x1000000u - 2000u); }
__device__ __forceinline__ uint32_t mod1e6_raw(uint32_t w) { return rem_raw(w, __umulhi(w, 0x431BDE83u) >> 19, 0x1000000u - 2000000u); }
template <bool B>
struct BoolTag {
  static constexpr bool value = B;
};
// x + (this lane's bit of `mask`): one v_addc_co_u32 with the wave mask as carry-in, where `x += cond ? 1 : 0` costs a
// v_cndmask_b32 and a v_add_u32 (the walks keep four such counters per column)
__device__ __forceinline__ int add_bit(int x, uint64_t mask) {
  int r;
  uint64_t carry_out;
  asm("v_addc_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(r), "=s"(carry_out) : "v"(x), "s"(mask));
  return r;
}

// inserted base (mut.ins_nt, pbsim.cpp:5485): "ATGC"[w % 8] for w % 8 < 4, else a copy of the reference base --
// one byte select from the eight bytes {nt nt nt nt | C G T A}
__device__ __forceinline__ uint32_t ins_base(uint32_t w, uint32_t nt) {
  const uint32_t nt4 = __builtin_amdgcn_perm(0u, nt, 0u);  // byte 0 in all four bytes
  return __builtin_amdgcn_perm(nt4, kATGC, (w & 7u) | 0x0c0c0c00u);
}

// mut.sub_nt_{a,t,g,c} (pbsim.cpp:5481-5484) packed little-endian
__device__ __forceinline__ uint32_t sub_table(uint32_t nt) {
  uint32_t t = 0;
  t = (nt == 'A') ? 0x00434754u : t;  // "TGC"
  t = (nt == 'T') ? 0x00434741u : t;  // "AGC"
  t = (nt == 'G') ? 0x00435441u : t;  // "ATC"
  t = (nt == 'C') ? 0x00475441u : t;  // "ATG"
  return t;
}

__device__ __forceinline__ int count_digit(int64_t v) {  // pbsim.cpp:5823-5835
  int d = 1;
  while (v >= 10) {
    v /= 10;
    d++;
  }
  return d;
}

__device__ __forceinline__ int dec_len(int64_t v) { return v < 0 ? 1 + count_digit(-v) : count_digit(v); }

// SAM record literals (pbsim.cpp:4017-4027)
#define PB_SAM_MID "\t4\t*\t0\t255\t*\t*\t0\t0\t"
#define PB_SAM_IP "\tcx:i:3\tip:B:C"
#define PB_SAM_PW "\tnp:i:1\tpw:B:C"
#define PB_SAM_T1 "\tqs:i:0\tqe:i:"
#define PB_SAM_T2 "\trq:f:"
#define PB_SAM_T3 "\tsn:B:f,10.0,10.0,10.0,10.0\tzm:i:"
#define PB_SAM_T4 "\tRG:Z:ffffffff\n"
#define PB_LEN(s) ((int)sizeof(s) - 1)

__device__ __forceinline__ int put_dec(char *dst, int64_t v) {
  int neg = 0;
  if (v < 0) {
    dst[0] = '-';
    dst++;
    v = -v;
    neg = 1;
  }
  const int d = count_digit(v);
  for (int i = d - 1; i >= 0; i--) {
    dst[i] = (char)('0' + (int)(v % 10));
    v /= 10;
  }
  return d + neg;
}

// ---------------------------------------------------------------------------
// K0: upper-case + homopolymer length.  A "break" at i starts a new run.
// ---------------------------------------------------------------------------
// `strip` = 0x7f once bit 7 of a byte may already carry another tile's hp == 11 flag (k_hp_final), else 0xff
__device__ __forceinline__ uint32_t ref_char(const uint8_t *seq, int64_t i, int keep_first_case, uint32_t strip = 0xffu) {
  const uint32_t c = seq[i] & strip;
  if (keep_first_case && (i == 0 || (seq[i - 1] & strip) == '\n')) return c;
  return to_upper(c);
}

// The 16 characters base .. base + 15 as ref_char() gives them, from ONE 16-byte load (base is a multiple of 16, the
// buffer ends with 64 bytes of padding); *prev = the character in front of the chunk (0x100 in front of the unit).
__device__ __forceinline__ void ref_chunk(const uint8_t *seq, int64_t base, int keep_first_case, uint32_t strip, uint32_t c[16],
                                          uint32_t *prev) {
  const uint4 v = *reinterpret_cast<const uint4 *>(seq + base);
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  uint32_t before = (base > 0) ? (uint32_t)(seq[base - 1] & strip) : (uint32_t)'\n';  // the start of the unit is a line start
  *prev = (base > 0) ? ref_char(seq, base - 1, keep_first_case, strip) : 0x100u;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const uint32_t raw = ((w[k >> 2] >> (8 * (k & 3))) & 0xffu) & strip;
    c[k] = (keep_first_case && before == '\n') ? raw : to_upper(raw);
    before = raw;
  }
}

__global__ __launch_bounds__(256) void k_hp_breaks(const uint8_t *seq, int64_t len, int keep_first_case,
                                                     int64_t *tile_first, int64_t *tile_last, DeviceFlags *flags) {
  __shared__ long long s_first, s_last;
  if (threadIdx.x == 0) {
    s_first = 0x7fffffffffffffffLL;
    s_last = -1;
  }
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kHpTile + (int64_t)threadIdx.x * 16;
  long long first = 0x7fffffffffffffffLL, last = -1;
  uint32_t high = 0;
  if (base < len) {
    uint32_t cc[16], prev;
    ref_chunk(seq, base, keep_first_case, 0xffu, cc, &prev);
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int64_t i = base + k;
      if (i >= len) break;
      const uint32_t c = cc[k];
      high |= c;
      if (c != prev) {
        if (first > i) first = i;
        last = i;
      }
      prev = c;
    }
  }
  if (high & 0x80u) flags->high_bytes = 1;  // non-ASCII input: bit 7 cannot carry the hp == 11 flag
  if (last >= 0) {
    atomicMin(&s_first, first);
    atomicMax(&s_last, last);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    tile_first[blockIdx.x] = s_first;
    tile_last[blockIdx.x] = s_last;
  }
}

// carry_start[t] = last break before tile t (exclusive max-scan of tile_last);
// carry_next[t] = first break after tile t (exclusive suffix min-scan of tile_first, `len` if none).
// One workgroup of 1024: the two scans run chunk by chunk with a running carry.
__global__ __launch_bounds__(1024) void k_hp_carry(const int64_t *tile_first, const int64_t *tile_last, int64_t n_tiles,
                                                     int64_t len, int64_t *carry_start, int64_t *carry_next) {
  __shared__ long long s_v[1024];
  __shared__ long long s_carry;
  const int tid = threadIdx.x;
  // forward: running maximum of tile_last (-1 = no break in the tile; position 0 is always a break)
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int64_t t0 = 0; t0 < n_tiles; t0 += 1024) {
    const int64_t t = t0 + tid;
    const long long v = (t < n_tiles) ? tile_last[t] : -1;
    s_v[tid] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      const long long x = (tid >= d) ? s_v[tid - d] : -1;
      __syncthreads();
      if (x > s_v[tid]) s_v[tid] = x;
      __syncthreads();
    }
    const long long carry = s_carry;
    const long long incl_prev = (tid > 0) ? s_v[tid - 1] : -1;  // max over the chunk's tiles before this one
    if (t < n_tiles) carry_start[t] = (incl_prev > carry) ? incl_prev : carry;
    __syncthreads();
    if (tid == 1023 && s_v[1023] > s_carry) s_carry = s_v[1023];
    __syncthreads();
  }
  // backward: running minimum of tile_first over the tiles after t
  if (tid == 0) s_carry = len;
  __syncthreads();
  const int64_t n_chunks = (n_tiles + 1023) / 1024;
  for (int64_t c = n_chunks - 1; c >= 0; c--) {
    const int64_t t = c * 1024 + tid;
    const long long big = 0x7fffffffffffffffLL;
    const long long v = (t < n_tiles && tile_last[t] >= 0) ? tile_first[t] : big;
    s_v[tid] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      const long long x = (tid + d < 1024) ? s_v[tid + d] : big;
      __syncthreads();
      if (x < s_v[tid]) s_v[tid] = x;
      __syncthreads();
    }
    const long long carry = s_carry;
    const long long incl_next = (tid < 1023) ? s_v[tid + 1] : big;  // min over the chunk's tiles after this one
    if (t < n_tiles) carry_next[t] = (incl_next < carry) ? incl_next : carry;
    __syncthreads();
    if (tid == 0 && s_v[0] < s_carry) s_carry = s_v[0];
    __syncthreads();
  }
}

// flag_hp11: --hp-del-bias 1 (default), where the deletion threshold depends on hp only through hp == 11 (Q1): the
// flag goes into bit 7 of the sequence byte itself (unless the input holds non-ASCII bytes), so that the walks need
// no second gather for it.  Other tiles rewrite their bytes concurrently: neighbours are read with the bit stripped.
__global__ __launch_bounds__(256) void k_hp_final(uint8_t *seq, uint8_t *hp, int flag_hp11, int64_t len,
                                                    int keep_first_case, const int64_t *carry_start,
                                                    const int64_t *carry_next, DeviceFlags *flags) {
  // positions inside the tile as int32 offsets from the tile's first base (-1 / kNone: no break)
  constexpr int kNone = 0x7fffffff;
  __shared__ int s_last[256], s_first[256];
  __shared__ unsigned int s_hist[12];
  const int tid = threadIdx.x;
  if (tid < 12) s_hist[tid] = 0;
  const int64_t tile0 = (int64_t)blockIdx.x * kHpTile;
  const int tb = tid * 16;  // this thread's first base, relative to the tile
  const int64_t base = tile0 + tb;
  const bool flag = flag_hp11 && !flags->high_bytes;
  const uint32_t strip = flag ? 0x7fu : 0xffu;
  uint32_t c[16];
  bool brk[16];
  int first = kNone, last = -1;
  int n = 0;
  if (base < len) {
    uint32_t prev;
    ref_chunk(seq, base, keep_first_case, strip, c, &prev);
#pragma unroll
    for (int k = 0; k < 16; k++) {
      if (base + k >= len) break;
      brk[k] = (c[k] != prev);
      if (brk[k]) {
        if (first == kNone) first = tb + k;
        last = tb + k;
      }
      prev = c[k];
      n = k + 1;
    }
  }
  s_last[tid] = last;
  s_first[tid] = first;
  __syncthreads();
  // inclusive max-scan of s_last / inclusive suffix min-scan of s_first
  for (int d = 1; d < 256; d <<= 1) {
    const int a = (tid >= d) ? s_last[tid - d] : -1;
    const int b = (tid + d < 256) ? s_first[tid + d] : kNone;
    __syncthreads();
    if (a > s_last[tid]) s_last[tid] = a;
    if (b < s_first[tid]) s_first[tid] = b;
    __syncthreads();
  }
  if (n > 0) {
    // run boundaries relative to THIS THREAD's first base: a run is at most 10^9 long, so int32 holds every difference
    const int64_t cs = (tid > 0 && s_last[tid - 1] >= 0) ? tile0 + s_last[tid - 1] : carry_start[blockIdx.x];
    const int64_t nx = (tid < 255 && s_first[tid + 1] != kNone) ? tile0 + s_first[tid + 1] : carry_next[blockIdx.x];
    int cur_start = (int)(cs - base), nxt = (int)(nx - base);
    int start[16], next_start[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      if (k < n && brk[k]) cur_start = k;
      start[k] = cur_start;
    }
#pragma unroll
    for (int k = 15; k >= 0; k--) {
      next_start[k] = nxt;
      if (k < n && brk[k]) nxt = k;
    }
    uint32_t so[4] = {0, 0, 0, 0}, ho[4] = {0, 0, 0, 0};  // the 16 sequence / hp bytes, packed
    unsigned long long census = 0;                        // 12 counters of 5 bits (<= 16 bases each)
#pragma unroll
    for (int k = 0; k < 16; k++) {
      if (k >= n) break;
      const int run = next_start[k] - start[k];
      // nnum++ ; if (nnum > 11) nnum = 10  (pbsim.cpp:1045-1048): 11,13,.. -> 11 ; 12,14,.. -> 10
      uint32_t v = (run <= 11) ? (uint32_t)run : ((run & 1) ? 11u : 10u);
      if (c[k] == 'N') v = 1;  // pbsim.cpp:1050-1054
      ho[k >> 2] |= v << (8 * (k & 3));
      so[k >> 2] |= (c[k] | ((flag && v == 11u) ? 0x80u : 0u)) << (8 * (k & 3));
      if (c[k] != '\n') census += 1ull << (5 * v);
    }
    for (int v = 0; v < 12; v++) {  // one LDS atomic per class this thread saw (a per-base atomic serialised on hp 1)
      const unsigned int cnt = (unsigned int)(census >> (5 * v)) & 31u;
      if (cnt) atomicAdd(&s_hist[v], cnt);
    }
    if (n == 16) {
      *reinterpret_cast<uint4 *>(seq + base) = make_uint4(so[0], so[1], so[2], so[3]);
      if (!flag) *reinterpret_cast<uint4 *>(hp + base) = make_uint4(ho[0], ho[1], ho[2], ho[3]);  // the flag replaces the array
    } else {
      for (int k = 0; k < n; k++) {
        seq[base + k] = (uint8_t)(so[k >> 2] >> (8 * (k & 3)));
        if (!flag) hp[base + k] = (uint8_t)(ho[k >> 2] >> (8 * (k & 3)));
      }
    }
  }
  __syncthreads();
  if (tid < 12 && s_hist[tid]) atomicAdd(&flags->hpfreq[tid], (unsigned long long)s_hist[tid]);
}

// ---------------------------------------------------------------------------
// K0l: a record as its FASTA lines -> the record (get_genome_seq's copy loop, pbsim.cpp:1014-1033: every byte of the sequence
// lines except the line feeds).  Tiles of 4096 bytes: keep counts, an exclusive scan over the tiles (launch_exclusive_scan_i64),
// then every tile squeezes itself to its place.  HBM-bound, once per record.
// ---------------------------------------------------------------------------
constexpr int kLineTile = 4096;

__device__ __forceinline__ uint32_t keep_mask16(const uint8_t *src, int64_t base, int64_t n, uint4 *v) {
  uint32_t keep = 0;
  if (base + 16 <= n) {
    *v = *reinterpret_cast<const uint4 *>(src + base);
    const uint32_t w[4] = {v->x, v->y, v->z, v->w};
#pragma unroll
    for (int k = 0; k < 16; k++) keep |= (uint32_t)(((w[k >> 2] >> ((k & 3) * 8)) & 0xffu) != 10u) << k;
  } else {
    uint32_t w[4] = {0, 0, 0, 0};
    for (int k = 0; k < 16 && base + k < n; k++) {
      const uint32_t c = src[base + k];
      w[k >> 2] |= c << ((k & 3) * 8);
      keep |= (uint32_t)(c != 10u) << k;
    }
    *v = uint4{w[0], w[1], w[2], w[3]};
  }
  return keep;
}

__global__ __launch_bounds__(256) void k_lines_count(const uint8_t *src, int64_t n, int64_t *tile_keep) {
  __shared__ int s_part[4];
  uint4 v;
  const int64_t base = (int64_t)blockIdx.x * kLineTile + threadIdx.x * 16;
  int c = base < n ? __popc(keep_mask16(src, base, n, &v)) : 0;
  for (int d = 32; d; d >>= 1) c += __shfl_down(c, d);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) tile_keep[blockIdx.x] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

__global__ __launch_bounds__(256) void k_lines_squeeze(const uint8_t *src, int64_t n, const int64_t *tile_off, uint8_t *dst) {
  __shared__ int s_scan[256];
  uint4 v = uint4{0, 0, 0, 0};
  const int64_t base = (int64_t)blockIdx.x * kLineTile + threadIdx.x * 16;
  const uint32_t keep = base < n ? keep_mask16(src, base, n, &v) : 0u;
  const int mine = __popc(keep);
  s_scan[threadIdx.x] = mine;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const int x = threadIdx.x >= d ? s_scan[threadIdx.x - d] : 0;
    __syncthreads();
    s_scan[threadIdx.x] += x;
    __syncthreads();
  }
  uint8_t *o = dst + tile_off[blockIdx.x] + (s_scan[threadIdx.x] - mine);
  if (keep == 0xffffu && ((uintptr_t)o & 3) == 0) {  // the usual thread: no line feed among its 16 bytes
    uint32_t *o4 = reinterpret_cast<uint32_t *>(o);
    o4[0] = v.x;
    o4[1] = v.y;
    o4[2] = v.z;
    o4[3] = v.w;
  } else {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    int k_out = 0;
#pragma unroll
    for (int k = 0; k < 16; k++)
      if ((keep >> k) & 1u) o[k_out++] = (uint8_t)(w[k >> 2] >> ((k & 3) * 8));
  }
}

// ---------------------------------------------------------------------------
// K1: read header (WGS).  pbsim.cpp:3793-3813 (= 2174-2194)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_header_wgs(HeaderArgs a) {
  short_kernel_priority();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < a.n_reads;
  const uint32_t read = (uint32_t)(a.first_read + i);
  const U4 w = header_block(a.seed, a.unit, read);
  int64_t L = live ? a.prob2len[(int64_t)(w.x % (uint32_t)a.len_rv) + 1] : 0;
  const int32_t raw = (int32_t)L;
  if (a.max_rawlen) {  // the batch's largest raw length: one atomic per wave (the job's rounds that cannot touch the quota, job.cpp)
    int32_t m = raw;
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(a.max_rawlen, (uint32_t)m);
  }
  if (!live) return;
  if (a.truncate_remaining >= 0 && L > a.truncate_remaining) {  // pbsim.cpp:3795-3800 (Q10)
    L = a.truncate_remaining;
    if (L < a.len_min) L = a.len_min;
  }
  const uint8_t acc = a.prob2acc[(int64_t)(w.y % (uint32_t)a.acc_rv) + 1];
  int64_t off;
  if (L >= a.ref_len) {  // pbsim.cpp:3804-3809
    off = 0;
    L = a.ref_len;
  } else {
    off = (int64_t)(w.z % (uint32_t)(a.ref_len - L + 1));
  }
  a.rawlen[i] = raw;
  a.len[i] = (int32_t)L;
  a.off[i] = (int32_t)off;
  a.acc[i] = acc;
}

// K1c: the chain of truncated reads behind a quota cut (kernels.h ChainState).  One launch per step and side of the walk.
__global__ void k_chain_init(ChainState *chain, int64_t remaining) {
  chain->remaining = remaining;
  chain->total = 0;
  chain->made = 0;
  chain->done = remaining <= 0;
}

__global__ __launch_bounds__(256) void k_chain_prepare(HeaderArgs a, int k, int32_t pass_num, const int32_t *task_of_slot,
                                                        int32_t *masked, int64_t n_slots_max, const ChainState *chain) {
  short_kernel_priority();
  const bool running = !chain->done;  // (only k_chain_update, a kernel of its own between two steps, writes the state)
  const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (slot < n_slots_max) {
    const int t = task_of_slot[slot];
    masked[slot] = (running && t >= 0 && t / pass_num == k) ? t : -1;
  }
  if (slot == 0 && running) {  // pbsim.cpp:3793-3809 for read k with len_total = quota - remaining
    const uint32_t read = (uint32_t)(a.first_read + k);
    const U4 w = header_block(a.seed, a.unit, read);
    int64_t L = a.prob2len[(int64_t)(w.x % (uint32_t)a.len_rv) + 1];
    const int32_t raw = (int32_t)L;
    if (L > chain->remaining) {
      L = chain->remaining;
      if (L < a.len_min) L = a.len_min;
    }
    int64_t off;
    if (L >= a.ref_len) {
      off = 0;
      L = a.ref_len;
    } else {
      off = (int64_t)(w.z % (uint32_t)(a.ref_len - L + 1));
    }
    a.rawlen[k] = raw;
    a.len[k] = (int32_t)L;
    a.off[k] = (int32_t)off;
    a.acc[k] = a.prob2acc[(int64_t)(w.y % (uint32_t)a.acc_rv) + 1];
  }
}

__global__ void k_chain_update(int k, int32_t pass_num, const int32_t *out_len, ChainState *chain) {
  if (chain->done) return;
  const int64_t made = out_len[(int64_t)k * pass_num];  // only pass 0 counts toward the quota (pbsim.cpp:3989-3991)
  chain->total += made;
  chain->remaining -= made;
  chain->made = k + 1;
  if (chain->remaining <= 0) chain->done = 1;
}

// K1t: read header (trans).  pbsim.cpp:4488-4504 (= 2809-2825): no quota, start
// position from the rank's bucket table, length clipped to the transcript end.
__global__ __launch_bounds__(256) void k_header_trans(HeaderArgs a) {
  short_kernel_priority();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n_reads) return;
  const uint32_t read = (uint32_t)(a.first_read + i);
  const U4 w = header_block(a.seed, 0u, read);
  const int u = a.read_unit[i];
  const int64_t G = a.unit_len[u];
  int64_t L = a.prob2len[(int64_t)(w.x % (uint32_t)a.len_rv) + 1];
  const uint8_t acc = a.prob2acc[(int64_t)(w.y % (uint32_t)a.acc_rv) + 1];
  const int rank = a.unit_rank[u];
  const uint32_t rv = (uint32_t)a.ssp_rv[rank];
  const uint32_t k = a.ssp[(size_t)rank * 1000 + w.z % (rv ? rv : 1u)];
  int64_t off = a.off_table[(size_t)u * 21 + k];
  if (off + L > G) L = G - off;
  if (L < 0) L = 0;
  if (a.is_templ) {  // the whole template, no length or start draw
    L = G;
    off = 0;
  }
  a.rawlen[i] = (int32_t)L;
  a.len[i] = (int32_t)L;
  a.off[i] = (int32_t)off;
  a.acc[i] = acc;
}

// ---------------------------------------------------------------------------
// Ks: counting sort of tasks by (class asc, length desc)
// ---------------------------------------------------------------------------
__device__ __forceinline__ int sort_bin(int acc, int acc_lo, int len) {
  int b = len >> kLenShift;
  if (b > kLenBuckets - 1) b = kLenBuckets - 1;
  return (acc - acc_lo) * kLenBuckets + (kLenBuckets - 1 - b);
}

__global__ __launch_bounds__(256) void k_sort_hist(SortArgs a) {
  short_kernel_priority();
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= a.n_reads) return;
  atomicAdd(&a.hist[(size_t)sort_bin(a.acc[r], a.acc_lo, a.len[r]) * kBinPad], a.pass_num);
}

// One workgroup of kScanBlock = 256 threads (one wave per SIMD, the footprint of a walk workgroup: a 1024-thread
// block needs 16 free wave slots on ONE CU at once and starves for tens of ms behind the other slot's walk): per class an exclusive scan over its kLenBuckets bins;
// class starts are rounded up to the walk workgroup size
__global__ __launch_bounds__(kScanBlock) void k_sort_scan(SortArgs a) {
  short_kernel_priority();
  __shared__ int s_part[kScanBlock];
  __shared__ int s_base;
  const int tid = threadIdx.x;
  if (tid == 0) s_base = 0;
  __syncthreads();
  constexpr int kPer = kLenBuckets / kScanBlock;
  for (int c = 0; c < a.ncls; c++) {
    int v[kPer], sum = 0;
    for (int k = 0; k < kPer; k++) {
      v[k] = a.hist[(size_t)(c * kLenBuckets + tid * kPer + k) * kBinPad];
      sum += v[k];
    }
    s_part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < kScanBlock; d <<= 1) {
      const int t = (tid >= d) ? s_part[tid - d] : 0;
      __syncthreads();
      s_part[tid] += t;
      __syncthreads();
    }
    const int base = s_base;
    int run = base + s_part[tid] - sum;
    // bins are in descending length: the long reads' bins come first (none in a class that takes no part)
    const int first_short = ((a.coop_classes >> c) & 1ull) ? kLenBuckets - a.coop_bucket : 0;
    for (int k = 0; k < kPer; k++) {
      a.bin_start[c * kLenBuckets + tid * kPer + k] = run;
      if (tid * kPer + k == first_short) a.coop_end[c] = run;
      run += v[k];
    }
    __syncthreads();
    if (tid == kScanBlock - 1) {
      if (first_short >= kLenBuckets) a.coop_end[c] = base + s_part[kScanBlock - 1];
      a.class_start[c] = base;
      s_base = (base + s_part[kScanBlock - 1] + kWG - 1) / kWG * kWG;
    }
    __syncthreads();
  }
  if (tid == 0) {
    a.class_start[a.ncls] = s_base;
    a.flags->total_slots = s_base;
  }
}

__global__ __launch_bounds__(256) void k_sort_scatter(SortArgs a) {
  short_kernel_priority();
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= a.n_reads) return;
  const int bin = sort_bin(a.acc[r], a.acc_lo, a.len[r]);
  const int slot = a.bin_start[bin] + atomicAdd(&a.bin_cursor[(size_t)bin * kBinPad], a.pass_num);
  for (int h = 0; h < a.pass_num; h++) {
    const int task = (int)(r * a.pass_num + h);
    a.task_of_slot[slot + h] = task;
    a.slot_of_task[task] = slot + h;
  }
}

// per wave: scratch columns needed = 2*Lmax + pad (in dwords per lane)
__global__ __launch_bounds__(256) void k_wave_cap(SortArgs a) {
  short_kernel_priority();
  const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n_waves = a.n_slots_max / 64;
  if (w >= n_waves) return;
  int lmax = -1, lmin = 0x7fffffff, cls = 0;
  if (w * 64 < a.flags->total_slots) {
    for (int l = 0; l < 64; l++) {
      const int task = a.task_of_slot[w * 64 + l];
      if (task >= 0) {
        const int L = a.len[task / a.pass_num];
        lmax = (L > lmax) ? L : lmax;
        lmin = (L < lmin) ? L : lmin;
        cls = a.acc[task / a.pass_num] - a.acc_lo;
      }
    }
  }
  // rows of whole 64-byte lines; a block whose tasks all belong to the wave walker is stored task by task
  // columns a row holds: cap_q8 / 256 x the block's longest read + pad (cap_q8 = 512: the reference's own bound, 2 L; the host
  // asks for less once it has seen what the model's reads need -- engine.cpp scratch factor -- and walks a batch again at 512
  // when a read does run out of row, kErrScratchOverflow)
  const int cap = (lmax < 0) ? 0 : ((((int)(((int64_t)lmax * a.cap_q8 + 255) >> 8) + kScratchPad + 3) / 4 + 15) & ~15);
  const bool coop = lmax >= 0 && a.coop_bucket < kLenBuckets && (lmin >> kLenShift) >= a.coop_bucket && ((a.coop_classes >> cls) & 1ull);
  a.wave_cap[w] = cap | (coop ? kWaveTransposed : 0);
}

__global__ __launch_bounds__(kScanBlock) void k_wave_scan(SortArgs a) {
  short_kernel_priority();
  __shared__ long long s_part[kScanBlock];
  __shared__ long long s_base;
  const int tid = threadIdx.x;
  const int64_t n_waves = a.n_slots_max / 64;
  if (tid == 0) s_base = 0;
  __syncthreads();
  constexpr int kPer = 8;
  for (int64_t w0 = 0; w0 < n_waves; w0 += kScanBlock * kPer) {
    long long v[kPer], sum = 0;
    for (int k = 0; k < kPer; k++) {
      const int64_t w = w0 + (int64_t)tid * kPer + k;
      v[k] = (w < n_waves) ? (long long)(a.wave_cap[w] & ~kWaveTransposed) * 256LL * a.regions : 0;
      sum += v[k];
    }
    s_part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < kScanBlock; d <<= 1) {
      const long long t = (tid >= d) ? s_part[tid - d] : 0;
      __syncthreads();
      s_part[tid] += t;
      __syncthreads();
    }
    long long run = s_base + s_part[tid] - sum;
    for (int k = 0; k < kPer; k++) {
      const int64_t w = w0 + (int64_t)tid * kPer + k;
      if (w < n_waves) a.wave_off[w] = run;
      run += v[k];
    }
    __syncthreads();
    if (tid == kScanBlock - 1) s_base += s_part[kScanBlock - 1];
    __syncthreads();
  }
  if (tid == 0) {
    a.flags->scratch_need = s_base;
    if (s_base > a.scratch_bytes) atomicOr(&a.flags->error, kErrScratchBudget);
  }
}

// Longest-processing-time-first launch order of the walk workgroups: the kernel
// ends when its longest read ends, so workgroups holding long reads go first
// (and their waves run at raised priority, see walk_priority()).
__global__ __launch_bounds__(256) void k_wg_hist(SortArgs a, int32_t *wg_hist) {
  short_kernel_priority();
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= a.n_slots_max / kWG) return;
  const int task = a.task_of_slot[g * kWG];
  int key = kLenBuckets;  // empty workgroups last
  if (task >= 0) {
    int b = a.len[task / a.pass_num] >> kLenShift;
    if (b > kLenBuckets - 1) b = kLenBuckets - 1;
    key = kLenBuckets - 1 - b;
  }
  atomicAdd(&wg_hist[key], 1);
}

__global__ __launch_bounds__(kScanBlock) void k_wg_scan(int32_t *wg_hist, int32_t *wg_start) {
  short_kernel_priority();
  __shared__ int s_part[kScanBlock];
  __shared__ int s_base;
  const int tid = threadIdx.x;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < kLenBuckets + 1; i0 += kScanBlock) {
    const int i = i0 + tid;
    const int v = (i < kLenBuckets + 1) ? wg_hist[i] : 0;
    s_part[tid] = v;
    __syncthreads();
    for (int d = 1; d < kScanBlock; d <<= 1) {
      const int t = (tid >= d) ? s_part[tid - d] : 0;
      __syncthreads();
      s_part[tid] += t;
      __syncthreads();
    }
    if (i < kLenBuckets + 1) wg_start[i] = s_base + s_part[tid] - v;
    __syncthreads();
    if (tid == kScanBlock - 1) s_base += s_part[kScanBlock - 1];
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_wg_scatter(SortArgs a, int32_t *wg_start, int32_t *wg_order) {
  short_kernel_priority();
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= a.n_slots_max / kWG) return;
  const int task = a.task_of_slot[g * kWG];
  int key = kLenBuckets;
  if (task >= 0) {
    int b = a.len[task / a.pass_num] >> kLenShift;
    if (b > kLenBuckets - 1) b = kLenBuckets - 1;
    key = kLenBuckets - 1 - b;
  }
  wg_order[atomicAdd(&wg_start[key], 1)] = (int32_t)g;
}

// waves that carry the longest reads are the kernel's critical path: raise them
__device__ __forceinline__ void walk_priority(int lmax_wave, int mean_len) {
  if (!PBSIM_WALK_PRIO) return;
  const int r = lmax_wave / (2 * (mean_len > 0 ? mean_len : 1));
  if (r >= 3) __builtin_amdgcn_s_setprio(3);
  else if (r == 2) __builtin_amdgcn_s_setprio(2);
  else if (r == 1) __builtin_amdgcn_s_setprio(1);
}

// ---------------------------------------------------------------------------
// Per-lane reference window shared by the walk kernels: the 8 bases under the
// cursor plus the next 8 in walking direction, fetched a group ahead so a lone
// long read never waits on HBM.  8-byte loads because every L2->fabric request
// moves a 64-byte sector whatever the load width (dword loads cost 16x their
// bytes and starved the kernel).  The homopolymer class comes from the hp byte
// array (a second window of the same shape) or, with kHpBits (default
// --hp-del-bias 1: the thresholds depend on hp only through hp == 11, Q1), from
// bit 7 of the sequence byte itself (k_hp_final put it there): no second gather.
// A cursor moves by at most 4 bases per 4-column group, hence crosses at most
// one window boundary per group; refill() runs once per group.
// ---------------------------------------------------------------------------
// what a finished task needed of its rows, as the factor the host sizes them with: (columns - pad) / length, in 1/1024
// (one atomic per task; the host keeps the largest it has seen and lays the next batches out with a little more)
__device__ __forceinline__ void note_row_need(DeviceFlags *flags, int columns, int L) {
  if (columns > kScratchPad && L > 0) atomicMax(&flags->need_q10, (uint32_t)(((uint64_t)(columns - kScratchPad) * 1024u + (uint32_t)L - 1u) / (uint32_t)L));
}

template <bool kHpBits>
struct RefCursor {
  const uint64_t *lane_seq, *lane_hp;
  int64_t p_first;
  uint32_t pl0, pl0s, smask, cur_wl;
  int wstep, rel;
  uint64_t wseq, whp, nseq, nhp;
  bool minus, need_next;

  __device__ __forceinline__ void init(const WalkArgs &a, int64_t off, int L, bool minus_, bool act) {
    init(a.ref, off, L, minus_, act);
  }
  __device__ __forceinline__ void init(const RefView &ref, int64_t off, int L, bool minus_, bool act) {
    minus = minus_;
    p_first = minus ? (off + L - 1) : off;
    pl0 = (uint32_t)p_first;  // low bits are all the cursor arithmetic needs
    smask = minus ? 0xffffffffu : 0u;  // pl0 -+ ro as ONE v_xad_u32: (ro ^ smask) + (pl0 - smask)
    pl0s = pl0 - smask;
    wstep = minus ? -1 : 1;
    lane_seq = reinterpret_cast<const uint64_t *>(ref.seq) + (p_first >> 3);
    lane_hp = reinterpret_cast<const uint64_t *>(ref.hp) + (p_first >> 3);
    rel = 0;
    cur_wl = pl0 >> 3;
    wseq = whp = nseq = nhp = 0;
    need_next = false;
    if (act) {
      wseq = lane_seq[0];
      const int nrel = ((p_first >> 3) + wstep < 0) ? 0 : wstep;
      nseq = lane_seq[nrel];
      if (!kHpBits) {
        whp = lane_hp[0];
        nhp = lane_hp[nrel];
      }
    }
  }

  // raw reference byte (forward strand, not complemented; with kHpBits bit 7 is the hp == 11 flag, which the s_comp
  // table drops) and homopolymer class at read offset ro
  __device__ __forceinline__ void at(int ro, bool act, uint32_t *raw, uint32_t *hp) {
    const uint32_t pl = ((uint32_t)ro ^ smask) + pl0s;
    const bool cross = act && ((pl >> 3) != cur_wl);
    wseq = cross ? nseq : wseq;
    cur_wl = cross ? (pl >> 3) : cur_wl;
    need_next = need_next || cross;
    // byte (pl & 7) of the window: one v_perm_b32 (selector 0x0c = constant zero for the upper bytes)
    const uint32_t sel = (pl & 7u) | 0x0c0c0c00u;
    *raw = __builtin_amdgcn_perm((uint32_t)(wseq >> 32), (uint32_t)wseq, sel);
    if (kHpBits) {
      *hp = (*raw & 0x80u) ? 11u : 1u;
    } else {
      whp = cross ? nhp : whp;
      *hp = __builtin_amdgcn_perm((uint32_t)(whp >> 32), (uint32_t)whp, sel);
    }
  }

  __device__ __forceinline__ void refill(bool act) {
    // windows walked so far, from the cursor's current window index (kept per step) instead of per-step counters
    rel = (int)cur_wl - (int)(pl0 >> 3);
    if (need_next && act) {
      const int nrel = ((p_first >> 3) + rel + wstep < 0) ? rel : rel + wstep;
      nseq = lane_seq[nrel];
      if (!kHpBits) nhp = lane_hp[nrel];
      need_next = false;
    }
  }
};

// ---------------------------------------------------------------------------
// RefStream (round 5): the lane walkers' reference window, re-cut so that a COLUMN costs two vector instructions instead of
// eight.  A group of four columns consumes at most four bases, known to lie at the cursor and the three places behind it in
// walking direction: begin_group() funnels exactly those four bytes -- already in the order they will be consumed, whatever
// the strand -- into ONE dword `g`; a column reads byte 0 of it and, if it consumed the base, shifts `g` by a byte.  (RefCursor
// located the byte from the read offset in every column: position, window test, two selects on a crossing, selector, permute.)
// The 8-byte words are kept in STREAM order: word k of the stream is word (first + k) of the record as it lies there on the '+'
// strand, word (first - k) with its bytes reversed on the '-' strand (two v_perm_b32 with per-lane selectors when a word
// arrives), so one piece of code serves both.  s0|s1 = the current word, rn = the next one as loaded (its first stream dword
// n2 is made per group: a group that starts in the upper half of a word reaches into it), rnn = the word after it.
// Two words ahead, because a word must never be waited for in the group it was asked in: at the end of a group advance() moves
// on if the group left the current word (current <- next <- the one after) and fetch() asks for a new "one after" -- into a
// register of its own (`pend`) that nothing reads before the NEXT group's advance(), a whole group (~2 000 cycles) later: that
// is where the compiler's s_waitcnt for the load lands.  The load is unconditional for that reason (lanes that did not move on
// read the record's first word, one sector for all of them, and drop it): as a conditional load its result was merged with the
// old value right behind the load -- a register copy at the loop's end with s_waitcnt vmcnt(0) in front --, so every group of
// every wave sat out a full memory round trip, hidden by the other waves at five per SIMD and fully exposed at one per SIMD,
// the delivered job's occupancy.
// kHpBits / the hp byte array: as RefCursor.
// ---------------------------------------------------------------------------
template <bool kHpBits>
struct RefStream {
  const uint64_t *lane_seq, *lane_hp;   // the word that holds the read's first base
  int rel_next, rel_min;         // where `rnn` lies, in words from there (the '-' strand counts down, never below the record's first word)
  uint32_t a0;                   // stream offset of the first base inside its word: p & 7 ('+'), 7 - (p & 7) ('-').  The walk counts
                                 // its STREAM POSITION sp = a0 + reference bases consumed: word sp >> 3, byte sp & 7 of the stream
  uint32_t sel_lo, sel_hi;       // permute selectors word -> stream dwords 0 and 1
  uint32_t wc;                   // stream index of the current word
  uint32_t s0, s1, n2, g;        // current word (stream order) | first stream dword of the next | the group's four bases
  uint32_t h0, h1, hn2, gh;      // the same of the hp bytes (!kHpBits)
  uint64_t rn, hn;               // next word as loaded
  uint64_t rnn, hnn;             // the word after it
  uint64_t pend, hpend;          // what fetch() asked for at the end of the group before; pcross: the lanes it asked for
  uint64_t pcross;
  const uint64_t *base_seq, *base_hp;
  int wstep;

  __device__ __forceinline__ uint32_t lo_of(uint64_t w) const { return __builtin_amdgcn_perm((uint32_t)(w >> 32), (uint32_t)w, sel_lo); }
  __device__ __forceinline__ uint32_t hi_of(uint64_t w) const { return __builtin_amdgcn_perm((uint32_t)(w >> 32), (uint32_t)w, sel_hi); }

  __device__ __forceinline__ void init(const RefView &ref, int64_t off, int L, bool minus, bool act) {
    const int64_t p_first = minus ? (off + L - 1) : off;
    const int64_t w_first = p_first >> 3;
    const uint32_t b = (uint32_t)p_first & 7u;
    a0 = minus ? 7u - b : b;
    sel_lo = minus ? 0x04050607u : 0x03020100u;  // pool of v_perm_b32(hi, lo, sel): bytes 0-3 = lo, 4-7 = hi
    sel_hi = minus ? 0x00010203u : 0x07060504u;
    wstep = minus ? -1 : 1;
    base_seq = reinterpret_cast<const uint64_t *>(ref.seq);
    base_hp = reinterpret_cast<const uint64_t *>(ref.hp);
    lane_seq = base_seq + w_first;
    lane_hp = base_hp + w_first;
    wc = 0;
    pend = hpend = pcross = 0;
    rel_min = w_first < 0x7fffffff ? -(int)w_first : (int)0x80000001;
    rel_next = 2 * wstep;
    uint64_t w0 = 0, v0 = 0;
    rn = hn = rnn = hnn = 0;
    if (act) {
      const int nrel = wstep > rel_min ? wstep : rel_min, nnrel = rel_next > rel_min ? rel_next : rel_min;
      w0 = lane_seq[0];
      rn = lane_seq[nrel];
      rnn = lane_seq[nnrel];
      if (!kHpBits) {
        v0 = lane_hp[0];
        hn = lane_hp[nrel];
        hnn = lane_hp[nnrel];
      }
    }
    s0 = lo_of(w0);
    s1 = hi_of(w0);
    h0 = lo_of(v0);
    h1 = hi_of(v0);
    n2 = hn2 = g = gh = 0;
    // the three words have arrived before the loop begins (a use the compiler must wait for): a load still pending at the loop's
    // head would put a wait there that every later group executes too -- and that one then sits out the group's own row stores
    if (kHpBits) asm volatile("" : "+v"(rn), "+v"(rnn));
    else asm volatile("" : "+v"(rn), "+v"(rnn), "+v"(hn), "+v"(hnn));
  }

  // the four bases the group can consume, from stream position sp on: byte 0 first
  __device__ __forceinline__ void begin_group(int sp) {
    const uint32_t ca = (uint32_t)sp;  // the group starts in word wc (advance() keeps it so)
    const bool upper = (ca & 4u) != 0;
    n2 = lo_of(rn);
    g = __builtin_amdgcn_alignbit(upper ? n2 : s1, upper ? s1 : s0, ca << 3);  // v_alignbit_b32 shifts by the low 5 bits: (ca & 3) * 8
    if (!kHpBits) {
      hn2 = lo_of(hn);
      gh = __builtin_amdgcn_alignbit(upper ? hn2 : h1, upper ? h1 : h0, ca << 3);
    }
  }
  // raw reference byte under the cursor (forward strand, not complemented; with kHpBits bit 7 is the hp == 11 flag, which the
  // s_comp table drops) and its homopolymer class
  __device__ __forceinline__ void peek(uint32_t *raw, uint32_t *hp) const {
    *raw = g & 0xffu;
    *hp = kHpBits ? ((g & 0x80u) ? 11u : 1u) : (gh & 0xffu);
  }
  __device__ __forceinline__ void consume(bool took) {
    g = took ? (g >> 8) : g;
    if (!kHpBits) gh = took ? (gh >> 8) : gh;
  }
  // end of a group, in front of its row stores: what the group before asked for has arrived (the first use of `pend`: the
  // compiler's wait for that load, a whole group after it); if this group left the current word, the next becomes current and
  // the one after it the next
  __device__ __forceinline__ bool advance(int sp, bool act) {
    const bool had = __builtin_amdgcn_inverse_ballot_w64(pcross);
    rnn = had ? pend : rnn;
    if (!kHpBits) hnn = had ? hpend : hnn;
    // (pinned in front of the group's stores: left alone the compiler sinks this merge -- and the wait -- behind them, where the
    // wait then sits out the stores it has just issued)
    if (kHpBits) asm volatile("" : "+v"(rnn) : : "memory");
    else asm volatile("" : "+v"(rnn), "+v"(hnn) : : "memory");
    const uint32_t wn = (uint32_t)sp >> 3;
    const bool cross = act && wn != wc;
    s0 = cross ? n2 : s0;
    s1 = cross ? hi_of(rn) : s1;
    rn = cross ? rnn : rn;
    if (!kHpBits) {
      h0 = cross ? hn2 : h0;
      h1 = cross ? hi_of(hn) : h1;
      hn = cross ? hnn : hn;
    }
    wc = wn;
    pcross = __builtin_amdgcn_ballot_w64(cross);
    return cross;
  }
  // ... and behind the stores: the lanes that moved on ask for their new "word after the next"
  __device__ __forceinline__ void fetch(bool cross) {
    rel_next += cross ? wstep : 0;
    const int nrel = rel_next > rel_min ? rel_next : rel_min;
    pend = *(cross ? lane_seq + nrel : base_seq);
    if (!kHpBits) hpend = *(cross ? lane_hp + nrel : base_hp);
  }
};

// stages the accuracy class blob of this workgroup and the two byte LUTs behind it:
// s_comp[0..255] identity, [256..511] revcomp's base map; s_sub[c*4+k] = substitution k
// of base c (pbsim.cpp:5481-5484), 0 for a non-ACGT base
__device__ __forceinline__ void stage_class(const WalkArgs &a, int cls, uint8_t *lds, uint8_t *s_comp, uint8_t *s_sub,
                                            int tid, bool hp_flag, int n_threads = kWG) {
  const uint4 *src = reinterpret_cast<const uint4 *>(a.cls_blob + (size_t)cls * a.stride);
  uint4 *dst = reinterpret_cast<uint4 *>(lds);
  for (uint32_t i = tid; i < a.stride / 16; i += n_threads) dst[i] = src[i];
  if (tid >= 256) return;  // the byte LUTs have 256 entries
  const uint32_t c = (uint32_t)tid;
  const uint32_t base = hp_flag ? (c & 0x7fu) : c;  // bit 7 of a sequence byte is the hp == 11 flag, not part of the base
  s_comp[c] = (uint8_t)base;
  s_comp[256 + c] = (uint8_t)complement(base);
  const uint32_t t = sub_table(c);
  s_sub[c * 4 + 0] = (uint8_t)(t & 0xffu);
  s_sub[c * 4 + 1] = (uint8_t)((t >> 8) & 0xffu);
  s_sub[c * 4 + 2] = (uint8_t)((t >> 16) & 0xffu);
  s_sub[c * 4 + 3] = 0;
}

// ---------------------------------------------------------------------------
// K2e: ERRHMM walk.  One lane per task (read, pass); the 4 waves of a workgroup
// share one accuracy class whose tables were staged in LDS.
//
//   per step (= one MAF column `m`, pbsim.cpp:3850-3975):
//     w = Philox(event=m, sub 0): w.x state | w.y deletion test | w.z emission | w.w nucleotide
//     state  = q==0 ? init[w.x % init_rv] : tran[state][w.x % tran_rv[state]]      (:3853-3859, Q2)
//     e      = (w.y%1000+1 <= del_thr[state][hp]) ? 3                                (:3861-3863)
//              : emis_rv==0 ? w.z%3 : cdf(w.z % emis_rv + 1)                         (:3865-3870)
//     below/above-range classes re-draw from sub-block 1                             (:3892-3925, Q3)
//     e: 0 match | 1 substitution | 2 insertion | 3 deletion                         (:3929-3973)
// ---------------------------------------------------------------------------
struct StepOut {
  uint32_t b, mr, mf;
};

// kFastRv: every init/transition modulus of the class tables is 1000 (true for
// all shipped models, checked on the host): `% 1000` folds to a multiply-high.
//
// The step body is written branch-free (selects on the mask of lanes that still walk): the
// kernel is VALU/SALU-issue bound, and every divergent `if` costs exec-mask
// bookkeeping for all 64 lanes.  Only three rare paths stay as real branches:
// states whose emission modulus is not 1000, the sub-block-1 draws (out-of-range
// accuracy classes, non-ACGT substitutions), and the look-ahead refill.
// kHpBits: --hp-del-bias 1 (default): the deletion threshold depends on hp only through
// hp == 11 (Q1), so the walk reads the 1-bit-per-base mask instead of the hp byte array.
// The loop itself, per kind of accuracy class: kVerbatim = the class of accuracy 100 (pbsim.cpp:3836-3846: the read is the
// reference as it stands -- no draws, no tables) has a loop of its own, so that the loop of every other class carries no test
// of the mode (round 5: the merged loop initialised sixteen registers per group and one per column for the path not taken).
struct LaneWalk {
  int sp, q, m, nsub;  // sp: stream position = RefStream::a0 + reference bases consumed
};

template <bool kFastRv, bool kHpBits, bool kVerbatim>
__device__ __forceinline__ LaneWalk errhmm_lanes(const WalkArgs &a, const uint8_t *lds, const uint8_t *s_comp, const uint8_t *s_sub,
                                                 uint32_t mode, uint32_t init_rv, uint32_t rate_mag, bool valid, bool walks, int L,
                                                 int64_t off, bool minus, uint32_t read_idx, uint32_t pass, uint32_t *maf_read,
                                                 uint32_t *maf_ref, int cap, RefStream<kHpBits> &cur) {
  int q = 0, m = 0, nsub = 0;
  uint32_t state = 0, tran_rv = 1;
  uint32_t acc_r = 0, acc_f = 0;
  // Which lanes still walk is kept as a WAVE MASK (round 5): a counter then takes its increment as the carry-in of one
  // v_addc_co_u32 (add_bit), a select its condition straight from the scalar pair, and "any lane left" is a scalar compare.
  // (As a per-lane bool every `x += (act && ..) ? 1 : 0` cost a v_cndmask_b32 and a v_add_u32.)  All 64 lanes run the loop
  // body; the real branches inside it only read the mask.
  uint64_t actm = __builtin_amdgcn_ballot_w64(walks);
  int group = 0;
  const WalkLane plane = walk_lane(a.seed, read_idx, pass, 0u);
  const uint32_t comp_off = minus ? 256u : 0u;
  cur.init(a.ref, off, L, minus, walks);
  int sp = (int)cur.a0;
  const int sp_end = sp + L;

  while (actm != 0) {
    cur.begin_group(sp);
    // One Philox block per MAF column; the four columns of this group are
    // independent of the walk, so their 4 x 20 multiplies interleave.
    U4 W[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
      W[j] = kVerbatim ? U4{0, 0, 0, 0} : walk_block_raw(plane, a.seed, a.unit, (uint32_t)__builtin_amdgcn_readfirstlane(group * 4 + j));
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const U4 w = W[j];  // the raw words: a draw is word >> 1 (mod1000_raw folds the shift into the remainder)
      const uint32_t wz = w.z >> 1, ww = w.w >> 1;
      // ---- reference base under the cursor
      uint32_t raw, hp;
      cur.peek(&raw, &hp);
      const uint32_t nt = s_comp[raw + comp_off];

      // ---- state, deletion test, emission class
      uint32_t e = 0, subb = 0;
      if (!kVerbatim) {
        uint32_t idx;
        if (kFastRv) {
          // the initial-state table sits right in front of the transition rows (host_tables.cpp): it is row 0
          idx = a.init_off + __umul24((q == 0) ? 0u : state, 1000u) + mod1000_raw(w.x);
        } else {
          uint32_t mod = (q == 0) ? init_rv : tran_rv;  // Q2: re-initialise while nothing has been emitted
          mod = mod ? mod : 1u;
          idx = ((q == 0) ? a.init_off : a.tran_off + (state - 1u) * 1000u) + (w.x >> 1) % mod;
        }
        state = lds[idx];  // a finished lane keeps walking harmlessly: nothing it computes is stored
        const uint16_t *row = reinterpret_cast<const uint16_t *>(lds + a.rows_off + state * 32u);
        if (!kFastRv) tran_rv = row[0];
        // emission class: the modulus differs from state to state (emis_rv = round(1000 * (1 - P(del)))), so
        // `z % emis_rv` is a multiply-high with the state's magic number (host_tables.cpp emission_magic) and a 24-bit
        // multiply-add with 2^24 - emis_rv.  The row's last dword holds the deletion thresholds of hp != 11 | hp == 11: one
        // 16-byte read serves the whole step.
        const uint4 em = *reinterpret_cast<const uint4 *>(lds + a.emis_off + state * 16u);
        const uint32_t thr = kHpBits ? __builtin_amdgcn_ubfe(em.w, (raw >> 3) & 16u, 16u) : row[4 + (hp < 12u ? hp : 11u)];
        const bool del = mod1000_raw(w.y) < thr;  // x % 1000 + 1 <= thr (pbsim.cpp:3857-3858), one compare
        const uint32_t quo = __umulhi(wz, em.x) >> (em.y & 31u);
        const uint32_t rem = (wz + __umul24(quo, em.y >> 8)) & 0xffffffu;  // rem < d <= 1000: 24 bits of z - quo * d
        e = (uint32_t)(rem >= (em.z & 0xffffu)) + (uint32_t)(rem >= (em.z >> 16));
        e = del ? 3u : e;
        subb = s_sub[nt * 4u + mod3(ww)];  // 0 for a non-ACGT reference base
        const bool need1 = __builtin_amdgcn_inverse_ballot_w64(actm) &&
                           ((mode == kModeBelow && e == 0) || (mode == kModeAbove && e != 0) || (subb == 0 && e == 1));
        if (need1) {
          const U4 v = walk_block(a.seed, a.unit, read_idx, pass, (uint32_t)(group * 4 + j), 1u);
          if (mode == kModeBelow && e == 0) {
            if (v.x % 100u + 1u <= rate_mag) e = v.y % 3u + 1u;
          } else if (mode == kModeAbove && e != 0) {
            if (v.x % 100u + 1u <= rate_mag) e = 0;
          }
          if (subb == 0) subb = (kATGC >> ((v.z & 3u) * 8u)) & 0xffu;
        }
      }
      // ---- emit
      // the read row's byte is byte e of {nt, substituted, inserted, 0}: 0 marks a deleted column (the text kernel prints
      // '-' there).  One v_perm_b32 selects it and drops it into byte j of the accumulator.
      if (kVerbatim) {
        acc_r |= nt << (8 * j);
        acc_f |= nt << (8 * j);
      } else {
        const uint32_t cand = nt | (subb << 8) | (ins_base(ww, nt) << 16);
        acc_r = __builtin_amdgcn_perm(cand, acc_r, (0x03020100u & ~(0xffu << (8 * j))) + ((4u + e) << (8 * j)));
        const uint32_t mf = (e == 2) ? (uint32_t)'-' : nt;
        acc_f |= mf << (8 * j);  // columns past the lane's last one are never read back (maf_len bounds them)
      }
      const uint64_t took = kVerbatim ? actm : (__builtin_amdgcn_ballot_w64(e != 2) & actm);  // every column consumes a reference base except insertions
      q = add_bit(q, kVerbatim ? actm : (__builtin_amdgcn_ballot_w64(e != 3) & actm));       // ... and emits a read base except deletions
      if (!kVerbatim) nsub = add_bit(nsub, __builtin_amdgcn_ballot_w64(e == 1) & actm);
      sp = add_bit(sp, took);
      m = add_bit(m, actm);
      cur.consume(__builtin_amdgcn_inverse_ballot_w64(took));
      actm &= __builtin_amdgcn_ballot_w64(sp < sp_end);
      actm = (group * 4 + j + 1 < cap) ? actm : 0;  // (wave-uniform: a walking lane's column count is the loop's own counter)
    }
    const bool moved = cur.advance(sp, __builtin_amdgcn_inverse_ballot_w64(actm));
    if (valid && m > group * 4) {
      scratch_store(&maf_read[(size_t)group * 64], acc_r);
      scratch_store(&maf_ref[(size_t)group * 64], acc_f);
    }
    cur.fetch(moved);
    acc_r = 0;
    acc_f = 0;
    group++;
  }
  return LaneWalk{sp, q, m, nsub};
}

template <bool kFastRv, bool kHpBits>
__global__ __launch_bounds__(kWG) void k_walk_errhmm(WalkArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x;
  const int64_t slot0 = (int64_t)a.wg_order[blockIdx.x] * kWG;
  if (slot0 >= a.flags->total_slots) return;
  if (a.flags->error & kErrScratchBudget) return;  // the pool cannot hold this batch: the host retries smaller
  int cls = 0;
  while (cls < a.ncls - 1 && slot0 >= a.class_start[cls + 1]) cls++;
  uint8_t *s_comp = lds + a.stride;
  uint8_t *s_sub = s_comp + 512;
  stage_class(a, cls, lds, s_comp, s_sub, tid, kHpBits);
  __syncthreads();
  const uint32_t *hdr = reinterpret_cast<const uint32_t *>(lds);
  const uint32_t init_rv = hdr[1], mode = hdr[2], rate_mag = hdr[3];

  const int64_t slot = slot0 + tid;
  const int lane = tid & 63;
  const int64_t wave = slot >> 6;
  const int task = a.task_of_slot[slot];
  const bool valid = task >= 0;
  int L = 0;
  int64_t off = 0;
  uint32_t read_idx = 0, pass = 0;
  bool minus = false;
  if (valid) {
    const int r = task / a.pass_num;
    pass = (uint32_t)(task - r * a.pass_num);
    read_idx = (uint32_t)(a.first_read + r);
    L = a.len[r];
    off = a.off[r] + (a.read_base ? a.read_base[r] : 0);
    // strand: wgs by parity of the read number (pbsim.cpp:3820-3826, Q9); trans by expression (:4516-4522)
    minus = a.read_minus ? (a.read_minus[r] != 0) : ((read_idx & 1u) == 0);
  }
  const int cap_dw = __builtin_amdgcn_readfirstlane(a.wave_cap[wave]) & ~kWaveTransposed;  // wave-uniform: keeps row offsets scalar
  walk_priority((int)(((int64_t)(cap_dw * 4 - kScratchPad) << 8) / a.cap_q8), a.mean_len);
  const int64_t woff = ((int64_t)__builtin_amdgcn_readfirstlane((int)(a.wave_off[wave] >> 32)) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)a.wave_off[wave]);
  uint32_t *maf_read = reinterpret_cast<uint32_t *>(a.scratch + woff) + lane;
  uint32_t *maf_ref = maf_read + (size_t)cap_dw * 64;
  const int cap = cap_dw * 4;  // columns the task's rows hold (k_wave_cap)

  const bool coop_lane = mode != kModeVerbatim && L >= a.coop_min_len;  // a long read: k_walk_errhmm_coop walks it
  const bool walks = valid && L > 0 && !coop_lane;
  RefStream<kHpBits> cur;
  const LaneWalk o = (mode == kModeVerbatim)
                         ? errhmm_lanes<kFastRv, kHpBits, true>(a, lds, s_comp, s_sub, mode, init_rv, rate_mag, valid, walks, L, off, minus,
                                                                read_idx, pass, maf_read, maf_ref, cap, cur)
                         : errhmm_lanes<kFastRv, kHpBits, false>(a, lds, s_comp, s_sub, mode, init_rv, rate_mag, valid, walks, L, off, minus,
                                                                 read_idx, pass, maf_read, maf_ref, cap, cur);
  if (valid && !coop_lane) {
    const int ro = o.sp - (int)cur.a0;
    if (ro < L) atomicOr(&a.flags->error, kErrScratchOverflow);
    a.out_len[task] = o.q;
    a.maf_len[task] = o.m;
    a.nsub[task] = o.nsub;
    a.nins[task] = o.m - ro;    // every column consumes a reference base except insertions
    a.ndel[task] = o.m - o.q;   // every column emits a read base except deletions
    note_row_need(a.flags, o.m, L);
  }
}

// ---------------------------------------------------------------------------
// K2c: the same walk for the LONG reads of a batch, one WAVE per read.
//
// A lane of k_walk_errhmm needs ~0.4 us per column whatever else the GPU does, so the longest of a few hundred thousand
// gamma-distributed lengths (70-76 k columns) holds a launch for ~30 ms and a single truncated tail read for milliseconds.
// Nothing in a column's draws depends on the walk (the block is keyed by the column), and the state chain does
// not depend on the reference: state[m] = tran[state[m-1]][x[m]].  A wave therefore takes 64 columns at a time, lane i
// column m0 + i:
//   1. every lane draws its column's block,
//   2. the state chain runs through the 64 draws (coop_chain: eight speculated groups of eight, the only serial part),
//   3. each lane classifies its column both ways (deleted / not deleted); which of the two holds depends on the
//      homopolymer class of the reference base under the column's cursor, and the cursor on the insertions before it:
//      a ballot + popcount gives every lane its cursor, the lanes re-decide, and the loop repeats while a decision moved a
//      cursor (by induction over the lanes it ends in the sequential walk's answer; in practice after one or two turns),
//   4. 64 columns of the two MAF rows leave as sixteen dwords each: 64 consecutive bytes per row in a scratch block whose tasks
//      all belong to this kernel (kWaveTransposed: rows stored task by task), sixteen stores 256 bytes apart in the one block
//      per class it shares with the lane walker (rows interleaved dword by dword; written that way by a whole batch the
//      kernel was bound by those partial lines: 39 instead of 100 G columns/s).
// Re-initialisation while nothing has been emitted (Q2: q == 0 selects the initial-state table) is handled the same way:
// assume it for column 0 only, walk, compare with what the columns decided, repeat if the assumption was wrong.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mbcnt64(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// mask bit of the lane ? a : b -- the inverse of a ballot, one v_cndmask (written out, the compiler tests the lane's bit)
__device__ __forceinline__ uint32_t select_by_mask(uint64_t mask, uint32_t a, uint32_t b) {
  uint32_t r;
  // (s_nop: the wait states gfx950 wants between a VALU write of the mask and its use -- the compiler cannot see into the asm)
  asm("s_nop 1\n\tv_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
  return r;
}

// writes of one lane of the wave, reads of another: LDS operations of a wave complete in order, the compiler must keep it
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LDS of one wave of the wave walker: reference window [128] | its homopolymer lengths [128] | x[64] u16 |
// the true chains' states [8 groups][4 chains][8] | two output rows
constexpr int kCoopWin = 0, kCoopHp = 128, kCoopX = 256, kCoopStates = 384, kCoopRows = 640, kCoopWaveLds = 768;
static_assert(kCoopMaxStates == 31, "four chains of eight start states per lane");  // the shipped models have <= 29 states

// States of the 64 columns: lane i receives the state after column i, given the state `s_in` in front of column 0.
// The chain state[i] = tran[state[i - 1]][x[i]] is serial, so it is cut into eight groups of eight columns:
//   A. lane (g, j) walks group g from the start states j, j + 8, j + 16, j + 24 (every state the group could start from),
//      four independent chains of eight LDS lookups, and keeps the states they pass through;
//   B. the groups' true start states follow from the end states of A: eight dependent steps on the SCALAR unit (the end
//      states stay in a register, a step is one v_readlane and a few s_ instructions -- through LDS the eight dependent
//      reads were a fifth of the step's vector instructions and most of its latency);
//   C. the lane that walked group g from its true start state hands its chains to LDS, the group's lanes pick their column.
// `qz`: columns that start from the initial-state table (row 0 of the transition table) whatever the state in front.
template <bool kInit>
__device__ __forceinline__ uint32_t coop_chain(const uint8_t *lds, uint32_t init_off, uint32_t smax, uint32_t reach, uint8_t *s_w,
                                               uint32_t x, uint64_t qz, uint32_t s_in, int lane) {
  uint16_t *s_x = reinterpret_cast<uint16_t *>(s_w + kCoopX);
  uint8_t *s_st = s_w + kCoopStates;
  const int g = lane >> 3, j = lane & 7;
  wave_sync();
  s_x[lane] = (uint16_t)x;
  wave_sync();
  const uint4 xv = *reinterpret_cast<const uint4 *>(s_x + g * 8);
  const uint32_t xt[8] = {xv.x & 0xffffu, xv.x >> 16, xv.y & 0xffffu, xv.y >> 16, xv.z & 0xffffu, xv.z >> 16, xv.w & 0xffffu, xv.w >> 16};
  const uint32_t qg = kInit ? (uint32_t)(qz >> (g * 8)) & 0xffu : 0u;
  uint32_t st[4], lo[4], hi[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    st[k] = (uint32_t)(j + 8 * k);
    st[k] = st[k] > smax ? 0u : st[k];  // no such state: any row inside the table will do
    lo[k] = hi[k] = 0;
  }
  auto walk_chains = [&](const int k0, const int k1) {  // chains side by side: their eight dependent table reads each overlap
#pragma unroll
    for (int t = 0; t < 8; t++) {
#pragma unroll
      for (int k = k0; k < k1; k++) {
        uint32_t row = st[k];
        if (kInit) row = ((qg >> t) & 1u) ? 0u : row;
        st[k] = lds[init_off + __umul24(row, 1000u) + xt[t]];
        if (t < 4) lo[k] |= st[k] << (8 * t);
        else hi[k] |= st[k] << (8 * (t - 4));
      }
    }
  };
  walk_chains(0, 2);
  // most classes have fewer than sixteen states: half the work; up to 24 (most of ERRHMM-ONT-HQ's): three chains
  const bool wide = reach > 15u;
  if (wide) {
    if (reach > 23u) walk_chains(2, 4);
    else walk_chains(2, 3);
  }
  // byte k = the state chain k ends in
  const uint32_t ends = __builtin_amdgcn_perm(hi[1], hi[0], 0x0c0c0703u) | (__builtin_amdgcn_perm(hi[3], hi[2], 0x0c0c0703u) << 16);
  uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_in);
  uint64_t starts = 0;  // byte gg = the state in front of group gg
#pragma unroll
  for (int gg = 0; gg < 8; gg++) {
    starts |= (uint64_t)s << (8 * gg);
    const uint32_t e4 = (uint32_t)__builtin_amdgcn_readlane((int)ends, (int)(gg * 8 + (s & 7u)));
    s = (e4 >> ((s >> 3) * 8u)) & 0xffu;
  }
  const uint32_t sg = (uint32_t)(starts >> (8 * g)) & 0xffu;
  if ((uint32_t)j == (sg & 7u)) {
    *reinterpret_cast<uint4 *>(s_st + g * 32) = make_uint4(lo[0], hi[0], lo[1], hi[1]);
    if (wide) *reinterpret_cast<uint4 *>(s_st + g * 32 + 16) = make_uint4(lo[2], hi[2], lo[3], hi[3]);
  }
  wave_sync();
  return s_st[g * 32 + j + (sg & 0x18u)];
}

template <bool kHpBits>
__device__ __forceinline__ void coop_walk_task(const WalkArgs &a, const uint8_t *lds, const uint8_t *s_comp, const uint8_t *s_sub,
                                               uint8_t *s_w, int slot, int lane) {
  const uint32_t *hdr = reinterpret_cast<const uint32_t *>(lds);
  const uint32_t smax = hdr[0], mode = hdr[2], rate_mag = hdr[3], reach = hdr[6];
  uint8_t *s_tr = s_w + kCoopRows;
  const int task = __builtin_amdgcn_readfirstlane(a.task_of_slot[slot]);
  if (task < 0) return;
  const int r = task / a.pass_num;
  const uint32_t pass = (uint32_t)(task - r * a.pass_num);
  const uint32_t read_idx = (uint32_t)(a.first_read + r);
  const int L = __builtin_amdgcn_readfirstlane(a.len[r]);
  const int64_t off = a.off[r] + (a.read_base ? a.read_base[r] : 0);
  const bool minus = a.read_minus ? (a.read_minus[r] != 0) : ((read_idx & 1u) == 0);
  const int64_t wave = slot >> 6;
  const int cap_raw = __builtin_amdgcn_readfirstlane(a.wave_cap[wave]);
  const int cap_dw = cap_raw & ~kWaveTransposed;
  // a block of wave-walked tasks only keeps each task's rows contiguous (kWaveTransposed); in the one block per class that
  // also holds lane-walked tasks the rows are interleaved dword by dword
  const bool transposed = (cap_raw & kWaveTransposed) != 0;
  const uint32_t row_step = transposed ? 1u : 64u;
  uint32_t *maf_read = reinterpret_cast<uint32_t *>(a.scratch + a.wave_off[wave]) + (transposed ? (size_t)(slot & 63) * cap_dw : (size_t)(slot & 63));
  uint32_t *maf_ref = maf_read + (size_t)cap_dw * 64;
  const uint32_t lane_dw = (uint32_t)lane * row_step;  // the lane's dword of a step's sixteen, from the step's first
  const int cap = cap_dw * 4;  // columns the task's rows hold (k_wave_cap)
  const uint32_t comp_off = minus ? 256u : 0u;
  const int64_t p_last = a.ref.len - 1;
  int64_t p_first = minus ? (off + L - 1) : off;
  p_first = p_first < 0 ? 0 : (p_first > p_last ? p_last : p_first);

  // the reference under the cursor: read coordinate t is record position p_first +- t, held at the record's ends.  A window
  // of 128 bases from `wb` lies in LDS (s_win, with the homopolymer lengths beside it where the record has no flag bits), the
  // 64 behind it are on their way in `nxt`: a load has a whole step to arrive
  const int64_t room = minus ? p_first : p_last - p_first;
  const uint32_t tmax = (uint32_t)(room > 0x7fffffff ? 0x7fffffff : (room < 0 ? 0 : room));
  const int64_t p0 = minus ? p_first - (int64_t)tmax : p_first;
  const uint8_t *seq0 = a.ref.seq + p0;
  const uint8_t *hp0 = kHpBits ? nullptr : a.ref.hp + p0;
  auto ref_at = [&](int t, uint32_t *hpv) -> uint32_t {
    const uint32_t tc = (uint32_t)t < tmax ? (uint32_t)t : tmax;
    const uint32_t o = minus ? tmax - tc : tc;
    if (!kHpBits) *hpv = hp0[o];
    return seq0[o];
  };
  uint8_t *s_win = s_w + kCoopWin, *s_hp = s_w + kCoopHp;
  int wb = 0;
  uint32_t nxt_hp = 0, h0 = 0, h1 = 0;
  const uint32_t b0 = ref_at(lane, &h0), b1 = ref_at(64 + lane, &h1);
  uint32_t nxt = ref_at(128 + lane, &nxt_hp);
  wave_sync();
  s_win[lane] = (uint8_t)b0;
  s_win[64 + lane] = (uint8_t)b1;
  if (!kHpBits) {
    s_hp[lane] = (uint8_t)h0;
    s_hp[64 + lane] = (uint8_t)h1;
  }
  wave_sync();
  const uint32_t win_off = (uint32_t)(s_win - lds);

  int m0 = 0, ro0 = 0, q0 = 0, nsub = 0;
  uint32_t st_in = 0;
  bool more = L > 0;
  while (more) {
    const uint32_t event = (uint32_t)(m0 + lane);
    const U4 w = walk_block(a.seed, a.unit, read_idx, pass, event, 0u);
    const uint32_t x = mod1000(w.x), y1 = mod1000(w.y) + 1u, w3 = mod3(w.w);
    U4 v = U4{0, 0, 0, 0};
    bool have_v = false;
    if (mode != kModeInRange) {  // out-of-range accuracy classes re-draw from sub-block 1 (Q3)
      v = walk_block(a.seed, a.unit, read_idx, pass, event, 1u);
      have_v = true;
    }
    uint64_t qz = (q0 == 0) ? 1ull : 0ull;
    const uint64_t in_cap = __ballot(event < (uint32_t)cap);
    // the lane's index into lds[] for the first base of the window, less the bases consumed in front of the step
    const uint32_t idx0 = win_off + (uint32_t)(ro0 - wb);
    const uint32_t idx_end = win_off + (uint32_t)(L - wb);  // roi < L
    uint32_t st, e, raw = 0;
    uint64_t valid, nonins;
    for (;;) {
      st = (q0 == 0) ? coop_chain<true>(lds, a.init_off, smax, reach, s_w, x, qz, st_in, lane)
                     : coop_chain<false>(lds, a.init_off, smax, reach, s_w, x, 0ull, st_in, lane);
      const uint4 em = *reinterpret_cast<const uint4 *>(lds + a.emis_off + st * 16u);
      const uint16_t *row = reinterpret_cast<const uint16_t *>(lds + a.rows_off + st * 32u);
      const uint32_t quo = __umulhi(w.z, em.x) >> (em.y & 31u);
      const uint32_t rem = (w.z + __umul24(quo, em.y >> 8)) & 0xffffffu;  // em.y >> 8 = 2^24 - d
      const uint32_t ce = (uint32_t)(rem >= (em.z & 0xffffu)) + (uint32_t)(rem >= (em.z >> 16));
      // the column's class if it is not deleted (e_keep) / if it is (e_del)
      uint32_t e_keep = ce, e_del = 3u;
      if (mode == kModeBelow) {
        if (ce == 0 && v.x % 100u + 1u <= rate_mag) e_keep = v.y % 3u + 1u;
      } else if (mode == kModeAbove) {
        const bool redraw = v.x % 100u + 1u <= rate_mag;
        if (ce != 0 && redraw) e_keep = 0;
        if (redraw) e_del = 0;
      }
      // the turns below only move masks: a column consumes a reference base unless it is an insertion that is not deleted
      // (e_del is never 2), and whether it is deleted depends on the base under its cursor through one of two thresholds
      const uint64_t keep_nonins = __ballot(e_keep != 2u);
      uint64_t del_lo = 0, del_hi = 0;
      if (kHpBits) {
        del_lo = __ballot(y1 <= (em.w & 0xffffu));
        del_hi = __ballot(y1 <= (em.w >> 16));
      }
      nonins = keep_nonins;
      uint64_t del;
      for (;;) {
        const uint32_t idx = __builtin_amdgcn_mbcnt_hi((uint32_t)(nonins >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nonins, idx0));
        raw = lds[idx];
        if (kHpBits) {
          const uint64_t in_hp = __ballot(raw > 127u);
          del = (in_hp & del_hi) | (~in_hp & del_lo);
        } else {
          const uint32_t hp = lds[idx + (uint32_t)(kCoopHp - kCoopWin)];
          del = __ballot(y1 <= row[4 + (hp < 12u ? hp : 11u)]);
        }
        valid = __ballot(idx < idx_end) & in_cap;
        const uint64_t now = keep_nonins | del;
        const uint64_t moved = (now ^ nonins) & valid;
        nonins = now;
        if (!moved) break;
      }
      e = select_by_mask(del, e_del, e_keep);
      if (q0 != 0) break;
      // Q2: the columns up to and including the first one that emits a base start from the initial-state table
      const uint64_t emits = __ballot(e != 3u) & valid;
      const uint64_t qa = emits ? ((2ull << __builtin_ctzll(emits)) - 1ull) : ~0ull;
      if (((qa ^ qz) & valid) == 0) break;
      qz = qa;
    }
    const int nv = __builtin_popcountll(valid);
    // ---- emit
    const uint32_t nt = s_comp[raw + comp_off];
    uint32_t subb = s_sub[nt * 4u + w3];  // 0 for a non-ACGT reference base
    if (e == 1u && subb == 0) {
      if (!have_v) v = walk_block(a.seed, a.unit, read_idx, pass, event, 1u);
      subb = (kATGC >> ((v.z & 3u) * 8u)) & 0xffu;
    }
    const uint32_t cand = nt | (subb << 8) | (ins_base(w.w, nt) << 16);
    const uint32_t rb = (cand >> (8u * e)) & 0xffu;  // byte e of {nt, substituted, inserted, 0}
    const uint32_t fb = (e == 2u) ? (uint32_t)'-' : nt;
    nsub += __builtin_popcountll(__ballot(e == 1u) & valid);
    q0 += __builtin_popcountll(__ballot(e != 3u) & valid);
    ro0 += __builtin_popcountll(nonins & valid);
    const int m_step = m0;  // the step's first column
    m0 += nv;
    if (nv > 0) st_in = (uint32_t)__builtin_amdgcn_readlane((int)st, nv - 1);
    more = nv == 64 && ro0 < L && m0 < cap;
    if (more && ro0 - wb >= 64) {
      // the window moves on by 64: its upper half becomes the lower, the bases that were on their way the upper
      const uint32_t up = s_win[64 + lane], up_hp = kHpBits ? 0u : s_hp[64 + lane];
      wave_sync();
      s_win[lane] = (uint8_t)up;
      s_win[64 + lane] = (uint8_t)nxt;
      if (!kHpBits) {
        s_hp[lane] = (uint8_t)up_hp;
        s_hp[64 + lane] = (uint8_t)nxt_hp;
      }
      wave_sync();
      wb += 64;
      nxt = ref_at(wb + 128 + lane, &nxt_hp);
    }
    // (the stores last: the wait for the bases in flight above must not also wait for them)
    wave_sync();
    s_tr[lane] = (uint8_t)rb;
    s_tr[64 + lane] = (uint8_t)fb;
    wave_sync();
#ifndef PBSIM_COOP_NOSTORE  // experiment: the walk without its scratch writes
    if (lane * 4 < nv) {
      const uint32_t dr = reinterpret_cast<const uint32_t *>(s_tr)[lane];
      const uint32_t df = reinterpret_cast<const uint32_t *>(s_tr + 64)[lane];
      const size_t step_dw = (size_t)(m_step >> 2) * row_step;  // the wave's part of the address; the lane's is lane_dw
      scratch_store(maf_read + step_dw + lane_dw, dr);
      scratch_store(maf_ref + step_dw + lane_dw, df);
    }
#endif
  }
  if (lane == 0) {
    if (ro0 < L) atomicOr(&a.flags->error, kErrScratchOverflow);
    a.out_len[task] = q0;
    a.maf_len[task] = m0;
    a.nsub[task] = nsub;
    a.nins[task] = m0 - ro0;
    a.ndel[task] = m0 - q0;
    note_row_need(a.flags, m0, L);
  }
}

template <bool kHpBits>
__global__ __launch_bounds__(kCoopWaves * 64) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_walk_errhmm_coop(WalkArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (a.flags->error & kErrScratchBudget) return;
  uint8_t *s_comp = lds + a.stride;
  uint8_t *s_sub = s_comp + 512;
  uint8_t *s_w = s_sub + 1024 + wv * kCoopWaveLds;
  // units of kCoopWaves tasks (one per wave, one class per workgroup), dealt round-robin to the persistent workgroups -- or
  // (a.coop_dynamic) drawn from a counter: the units of a class are sorted longest first, a workgroup that drew long ones draws fewer
  __shared__ int s_unit;
  unsigned long long *ticket = reinterpret_cast<unsigned long long *>(&a.flags->sums[7]);
  int staged = -1, c = 0, ubase = 0;
  for (int u = blockIdx.x;; u += gridDim.x) {
    if (a.coop_dynamic) {
      __syncthreads();  // every wave has read the previous draw
      if (tid == 0) s_unit = (int)atomicAdd(ticket, 1ull);
      __syncthreads();
      u = s_unit;
    }
    int nc = 0;
    for (; c < a.ncls; c++) {
      nc = a.coop_end[c] - a.class_start[c];
      nc = nc > 0 ? nc : 0;
      const int nu = (nc + kCoopWaves - 1) / kCoopWaves;
      if (u < ubase + nu) break;
      ubase += nu;
    }
    if (c >= a.ncls) break;
    if (staged != c) {
      __syncthreads();  // the other waves may still read the previous class
      stage_class(a, c, lds, s_comp, s_sub, tid, kHpBits, kCoopWaves * 64);
      __syncthreads();
      staged = c;
    }
    const int k = (u - ubase) * kCoopWaves + wv;
    const uint32_t mode = reinterpret_cast<const uint32_t *>(lds)[2];
    if (k < nc && mode != kModeVerbatim) coop_walk_task<kHpBits>(a, lds, s_comp, s_sub, s_w, a.class_start[c] + k, lane);
  }
}

// ---------------------------------------------------------------------------
// K2q: QSHMM walk (pbsim.cpp:2209-2282).  Column m is either an emitted base or
// a deleted reference base.  Every column m >= 1 first takes the deletion test
// of sub-block 2 (the inner while of :2268-2281; word m & 3 of the block of event m >> 2); a column that is not deleted
// emits a base from sub-block 0: w.x state | w.y quality | w.z error class |
// w.w nucleotide; sub-block 1 w.x = non-ACGT substitution.
// ---------------------------------------------------------------------------
// kFastRv: every init / transition / emission modulus of the class is 100.
template <bool kFastRv, bool kHpBits>
__global__ __launch_bounds__(kWG) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_walk_qshmm(WalkArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x;
  const int64_t slot0 = (int64_t)a.wg_order[blockIdx.x] * kWG;
  if (slot0 >= a.flags->total_slots) return;
  if (a.flags->error & kErrScratchBudget) return;  // the pool cannot hold this batch: the host retries smaller
  int cls = 0;
  while (cls < a.ncls - 1 && slot0 >= a.class_start[cls + 1]) cls++;
  // LDS: [class blob | s_comp 512 | s_sub 1024 | sub_thre 96 u32 | ins_thre 96 u32 | del_thr 94*12 u32 | qprob 94 f64]
  uint8_t *s_comp = lds + a.stride;
  uint8_t *s_sub = s_comp + 512;
  stage_class(a, cls, lds, s_comp, s_sub, tid, kHpBits);
  uint32_t *s_subt = reinterpret_cast<uint32_t *>(s_sub + 1024);
  uint32_t *s_ins = s_subt + 96;
  uint32_t *s_del = s_ins + 96;
  double *s_qprob = reinterpret_cast<double *>(s_del + 94 * 12);
  for (int i = tid; i < 94; i += kWG) {
    s_subt[i] = a.sub_thre[i];
    s_ins[i] = a.ins_thre[i];
    s_qprob[i] = a.qprob[i];
  }
  for (int i = tid; i < 94 * 12; i += kWG) s_del[i] = a.del_thr[i];
  // With the default --hp-del-bias 1 a quality's deletion threshold depends on hp only through "none yet" (slot 0, Q15),
  // hp == 11 (bias 0.0, Q1: never) and everything else (one value): everything a column needs of its quality Q then sits in
  // ONE 32-byte row {sub_thre, ins_thre, del (hp 1..10), del (slot 0) | qprob f64 | -} -- a 16-byte and an 8-byte read behind
  // the quality lookup instead of four table reads spread over two columns.
  uint32_t *s_row = reinterpret_cast<uint32_t *>(s_qprob + 94);
  if (kHpBits)
    for (int i = tid; i < 94; i += kWG) {
      s_row[i * 8 + 0] = a.sub_thre[i];
      s_row[i * 8 + 1] = a.ins_thre[i];
      s_row[i * 8 + 2] = a.del_thr[i * 12 + 1];
      s_row[i * 8 + 3] = a.del_thr[i * 12 + 0];
      *reinterpret_cast<double *>(&s_row[i * 8 + 4]) = a.qprob[i];
      s_row[i * 8 + 6] = s_row[i * 8 + 7] = 0;
    }
  __syncthreads();
  const uint32_t *hdr = reinterpret_cast<const uint32_t *>(lds);
  const uint32_t init_rv = hdr[1], has_model = hdr[2], freq_rv = hdr[3];
  const uint16_t *rvs = reinterpret_cast<const uint16_t *>(lds + a.rv_off);

  const int64_t slot = slot0 + tid;
  const int lane = tid & 63;
  const int64_t wave = slot >> 6;
  const int task = a.task_of_slot[slot];
  const bool valid = task >= 0;
  int L = 0;
  int64_t off = 0;
  uint32_t read_idx = 0, pass = 0;
  bool minus = false;
  if (valid) {
    const int r = task / a.pass_num;
    pass = (uint32_t)(task - r * a.pass_num);
    read_idx = (uint32_t)(a.first_read + r);
    L = a.len[r];
    off = a.off[r] + (a.read_base ? a.read_base[r] : 0);
    // strand: wgs by parity of the read number (pbsim.cpp:3820-3826, Q9); trans by expression (:4516-4522)
    minus = a.read_minus ? (a.read_minus[r] != 0) : ((read_idx & 1u) == 0);
  }
  const int cap_dw = __builtin_amdgcn_readfirstlane(a.wave_cap[wave]) & ~kWaveTransposed;  // wave-uniform: keeps row offsets scalar
  walk_priority((int)(((int64_t)(cap_dw * 4 - kScratchPad) << 8) / a.cap_q8), a.mean_len);
  const int64_t woff = ((int64_t)__builtin_amdgcn_readfirstlane((int)(a.wave_off[wave] >> 32)) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)a.wave_off[wave]);
  uint32_t *maf_read = reinterpret_cast<uint32_t *>(a.scratch + woff) + lane;
  uint32_t *maf_ref = maf_read + (size_t)cap_dw * 64;
  uint32_t *qual_row = maf_ref + (size_t)cap_dw * 64;  // quality per MAF column (0 in deleted columns)
  const int cap = cap_dw * 4;  // columns the task's rows hold (k_wave_cap)

  int q = 0, m = 0;
  uint32_t state = 0, last_q = 0, hp_prev = 0;  // hp of the last consumed reference base; none yet -> slot 0 (Q15)
  uint32_t del_norm = 0, del_none = 0;          // kHpBits: the last emitted quality's deletion thresholds (hp 1..10 | slot 0)
  uint32_t acc_r = 0, acc_f = 0, acc_q = 0;
  int nsub = 0;
  double qsum = 0.0;
  const bool coop_lane = L >= a.coop_min_len;  // a long task: k_walk_qshmm_coop walks it (classes with and without a model)
  // which lanes still walk: a wave mask (errhmm_lanes says why); sp: the stream position (RefStream)
  uint64_t actm = __builtin_amdgcn_ballot_w64(valid && L > 0 && !coop_lane);
  int group = 0;
  const uint32_t comp_off = minus ? 256u : 0u;
  RefStream<kHpBits> cur;
  cur.init(a.ref, off, L, minus, __builtin_amdgcn_inverse_ballot_w64(actm));
  int sp = (int)cur.a0;
  const int sp_end = sp + L;
  const WalkLane lane_e = walk_lane(a.seed, read_idx, pass, 0u);  // emission blocks
  const WalkLane lane_d = walk_lane(a.seed, read_idx, pass, 2u);  // deletion-test blocks

  while (actm != 0) {
    cur.begin_group(sp);
    U4 E[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
      E[j] = walk_block_raw(lane_e, a.seed, a.unit, (uint32_t)__builtin_amdgcn_readfirstlane(group * 4 + j));
    // the deletion tests of the group's four columns are the four words of ONE block (event = column >> 2)
    const U4 Dq = walk_block_raw(lane_d, a.seed, a.unit, (uint32_t)__builtin_amdgcn_readfirstlane(group));
    const uint32_t D[4] = {Dq.x, Dq.y, Dq.z, Dq.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const U4 w = E[j];  // raw words: a draw is word >> 1 (mod100_raw / mod1e6_raw fold the shift into the remainder)
      const uint32_t ww = w.w >> 1;
      const bool act = __builtin_amdgcn_inverse_ballot_w64(actm);
      uint32_t raw, hp;
      cur.peek(&raw, &hp);
      const uint32_t nt = s_comp[raw + comp_off];
      // every column m >= 1 first takes the deletion test of the inner while (pbsim.cpp:2268-2281)
      const uint32_t del_thr = kHpBits ? ((hp_prev == 11u) ? 0u : (hp_prev == 0u) ? del_none : del_norm)
                                       : s_del[last_q * 12u + (hp_prev < 12u ? hp_prev : 11u)];
      // (m > 0: a walking lane's column count is the loop's own counter, wave-uniform)
      const bool deleted = (group * 4 + j > 0) & (mod1e6_raw(D[j]) < del_thr);
      // ---- emission (computed for every lane, used where the column is not deleted)
      uint32_t qv, st = state;
      if (has_model) {
        uint32_t idx;
        if (kFastRv) {
          idx = a.init_off + __umul24((q == 0) ? 0u : state, 100u) + mod100_raw(w.x);  // init table = row 0 (host_tables.cpp)
        } else {
          uint32_t mod = (q == 0) ? init_rv : rvs[2 * state];
          mod = mod ? mod : 1u;
          idx = ((q == 0) ? a.init_off : a.tran_off + (state - 1u) * 100u) + (w.x >> 1) % mod;
        }
        st = lds[idx];
        uint32_t eidx;
        if (kFastRv) {
          eidx = a.emis_off + __umul24(st - 1u, 100u) + mod100_raw(w.y);
        } else {
          uint32_t emod = rvs[2 * st + 1];
          emod = emod ? emod : 1u;
          eidx = a.emis_off + (st - 1u) * 100u + (w.y >> 1) % emod;
        }
        qv = lds[eidx];
      } else {
        qv = lds[a.freq_off + (w.y >> 1) % (freq_rv ? freq_rv : 1u)];
      }
      qv = (qv < 94u) ? qv : 93u;
      const bool emit = act && !deleted;
      state = emit ? st : state;
      last_q = emit ? qv : last_q;
      uint32_t thr_sub, thr_ins;
      double qp;
      if (kHpBits) {
        const uint4 row = *reinterpret_cast<const uint4 *>(&s_row[qv * 8u]);
        qp = *reinterpret_cast<const double *>(&s_row[qv * 8u + 4u]);
        thr_sub = row.x;
        thr_ins = row.y;
        del_norm = emit ? row.z : del_norm;
        del_none = emit ? row.w : del_none;
      } else {
        thr_sub = s_subt[qv];
        thr_ins = s_ins[qv];
        qp = s_qprob[qv];
      }
      qsum += emit ? qp : 0.0;  // ordered double sum (pbsim.cpp:2309-2313); + 0.0 leaves it unchanged
      const uint32_t x = mod1e6_raw(w.z);
      const bool is_sub = x < thr_sub;             // pbsim.cpp:2233-2249
      const bool is_ins = !is_sub && x < thr_ins;   // pbsim.cpp:2250-2258
      uint32_t subb = s_sub[nt * 4u + mod3(ww)];
      if (emit && is_sub && subb == 0) {  // non-ACGT reference base: one more draw (rare)
        const U4 v = walk_block(a.seed, a.unit, read_idx, pass, (uint32_t)(group * 4 + j), 1u);
        subb = (kATGC >> ((v.x & 3u) * 8u)) & 0xffu;
      }
      const uint32_t insb = (ww & 4u) ? nt : ((kATGC >> ((ww & 3u) * 8u)) & 0xffu);
      const uint32_t b = is_sub ? subb : is_ins ? insb : nt;
      const uint32_t mr = deleted ? 0u : b;                      // 0 marks a deleted column
      const uint32_t mf = (!deleted && is_ins) ? (uint32_t)'-' : nt;
      acc_r |= mr << (8 * j);
      acc_f |= mf << (8 * j);
      acc_q |= (deleted ? 0u : (qv + 33u)) << (8 * j);
      const uint64_t delm = __builtin_amdgcn_ballot_w64(deleted);
      const uint64_t took = actm & (delm | ~__builtin_amdgcn_ballot_w64(is_ins));  // every column consumes a reference base except insertions
      const bool consumed = __builtin_amdgcn_inverse_ballot_w64(took);
      hp_prev = consumed ? hp : hp_prev;
      q = add_bit(q, actm & ~delm);
      nsub = add_bit(nsub, actm & ~delm & __builtin_amdgcn_ballot_w64(is_sub));
      sp = add_bit(sp, took);
      m = add_bit(m, actm);
      cur.consume(consumed);
      actm &= __builtin_amdgcn_ballot_w64(sp < sp_end);
      actm = (group * 4 + j + 1 < cap) ? actm : 0;
    }
    const bool moved = cur.advance(sp, __builtin_amdgcn_inverse_ballot_w64(actm));
    if (valid && m > group * 4) {
      scratch_store(&maf_read[(size_t)group * 64], acc_r);
      scratch_store(&maf_ref[(size_t)group * 64], acc_f);
      scratch_store(&qual_row[(size_t)group * 64], acc_q);
    }
    cur.fetch(moved);
    acc_r = 0;
    acc_f = 0;
    acc_q = 0;
    group++;
  }
  if (valid && !coop_lane) {
    const int ro = sp - (int)cur.a0;
    if (ro < L) atomicOr(&a.flags->error, kErrScratchOverflow);
    a.out_len[task] = q;
    a.maf_len[task] = m;
    a.nsub[task] = nsub;
    a.nins[task] = m - ro;
    a.ndel[task] = m - q;
    a.qsum[task] = qsum;
    note_row_need(a.flags, m, L);
  }
}

// ---------------------------------------------------------------------------
// K2qc: the QSHMM walk for the LONG tasks of a batch (and every task of a small one), one WAVE per task, 64 columns per step
// (round 3).  Every draw of a column is keyed by the column, as in the ERRHMM wave walker -- but here the state chain is NOT
// independent of what the columns decide: a deleted column does not advance the state (pbsim.cpp:2268-2281 loops over the
// reference, the state moves per emitted base), and whether column m is deleted depends on the quality the last emitted
// column drew, i.e. on the state chain in front of it.  A step therefore iterates to the sequential walk's fixed point:
//   guess which columns are deleted -> state chain that passes deleted columns through (qcoop_chain: eight groups of eight
//   columns from every possible start state) -> qualities -> every column's deletion test from the quality of the nearest
//   emitted column in front of it and the homopolymer flag under its cursor (ballot + popcount give the cursors) -> compare
//   with the guess, repeat with the outcome.
// Column p's outcome depends on columns < p only, so each turn extends the correct prefix by at least one column and the
// fixed point is the sequential answer; in practice two or three turns (95 % of the deletion draws lie above every quality's
// threshold and ~2 % below every one: a wrong state in front of a column rarely flips it).  The ordered f64 sum of error
// probabilities (pbsim.cpp:2309-2313; the %f report and the accuracy histogram depend on its exact value) is added lane by
// lane in column order (since round 4 behind the walk: k_qshmm_coop_qsum).  Models whose moduli are all 100 and records with the
// hp == 11 flag in their bytes (kFastRv && kHpBits); classes without a model take the same steps without the chain.
// ---------------------------------------------------------------------------
// LDS of one wave: x[64] u16 | the true chains' states [8 groups][8 chains][8] | three output rows | the deletion draws of 256 columns
constexpr int kQCoopX = 0, kQCoopStates = 128, kQCoopRows = 640, kQCoopD = 832, kQCoopT = 1856, kQCoopWaveLds = 2368;

// state after each of the 64 columns, given s_in in front of column 0; `delm`: columns that leave the state alone;
// `init0`: column 0 starts from the initial-state table (row 0 of the transition table) whatever s_in is.
//
// The chains live across the turns of a step's fixed point (QChains, in registers): a group's chains depend on the draws of its
// eight columns -- fixed for the step -- and on ITS eight bits of `delm`, and a turn changes few of those (it extends the correct
// prefix; measured 2.2 turns a step, one or two groups touched per further turn).  Round 5: the first turn of a step walks every
// group from every start state as before (lane (g, j): group g from the start states j + 8 k); a FURTHER turn walks only the
// groups whose bits of `delm` changed, each by all 64 lanes at once -- lane s walks the group from start state s: ONE chain of
// at most eight lookups, deleted columns skipped outright (the group's bits are wave-uniform there) -- and hands the chains to
// the group's own lanes through LDS.  (Until then every turn re-ran all n_chains x 8 lookups per lane: 477 of the step's 636
// vector instructions.)  More than two groups touched: the full walk again.
struct QChains {
  uint32_t lo[8], hi[8];   // chain k of this lane: states behind columns 0..3 / 4..7 of its group, a byte each
  uint64_t delm;           // the mask the chains were walked with
};

__device__ __forceinline__ uint32_t qcoop_chain(const uint8_t *lds, uint32_t init_off, uint32_t n_chains, uint32_t reach, uint8_t *s_w,
                                                uint32_t x, uint64_t delm, bool init0, uint32_t s_in, int lane, QChains &C, bool fresh) {
  uint16_t *s_x = reinterpret_cast<uint16_t *>(s_w + kQCoopX);
  uint8_t *s_st = s_w + kQCoopStates;
  const int g = lane >> 3, j = lane & 7;
  uint32_t touched = 0xffu;  // groups to walk (bit per group)
  if (fresh) {
    wave_sync();
    s_x[lane] = (uint16_t)x;
    wave_sync();
  } else {
    const uint64_t ch = delm ^ C.delm;
    touched = 0;
#pragma unroll
    for (int gg = 0; gg < 8; gg++) touched |= ((uint32_t)(ch >> (8 * gg)) & 0xffu) ? (1u << gg) : 0u;
  }
  C.delm = delm;
  if (fresh || __builtin_popcount(touched) > 2) {
    const uint4 xv = *reinterpret_cast<const uint4 *>(s_x + g * 8);
    const uint32_t xt[8] = {xv.x & 0xffffu, xv.x >> 16, xv.y & 0xffffu, xv.y >> 16, xv.z & 0xffffu, xv.z >> 16, xv.w & 0xffffu, xv.w >> 16};
    const uint32_t dg = (uint32_t)(delm >> (g * 8)) & 0xffu;
    const bool first_init = init0 && g == 0;
#pragma unroll
    for (int k = 0; k < 8; k++) C.lo[k] = C.hi[k] = 0;
    // kN chains side by side: their eight dependent table reads each overlap (walked pair after pair the step took n/2 times as long)
    auto walk = [&](auto tag) {
      constexpr int kN = decltype(tag)::value;
      uint32_t st[kN];
#pragma unroll
      for (int k = 0; k < kN; k++) st[k] = (uint32_t)(j + 8 * k);
#pragma unroll
      for (int t = 0; t < 8; t++) {
#pragma unroll
        for (int k = 0; k < kN; k++) {
          uint32_t row = st[k];
          if (t == 0) row = first_init ? 0u : row;
          const uint32_t nx = lds[init_off + __umul24(row, 100u) + xt[t]];
          st[k] = ((dg >> t) & 1u) ? st[k] : nx;
          if (t < 4) C.lo[k] |= st[k] << (8 * t);
          else C.hi[k] |= st[k] << (8 * (t - 4));
        }
      }
    };
    // (one instantiation per count: QSHMM-RSII's classes reach 37 states = five chains; walked as six they cost a fifth more)
    switch (n_chains) {
      case 1: walk(std::integral_constant<int, 1>()); break;
      case 2: walk(std::integral_constant<int, 2>()); break;
      case 3: walk(std::integral_constant<int, 3>()); break;
      case 4: walk(std::integral_constant<int, 4>()); break;
      case 5: walk(std::integral_constant<int, 5>()); break;
      case 6: walk(std::integral_constant<int, 6>()); break;
      case 7: walk(std::integral_constant<int, 7>()); break;
      default: walk(std::integral_constant<int, 8>()); break;
    }
  } else {
    // ---- repair: every touched group by the whole wave, lane s from start state s
    uint2 *s_t = reinterpret_cast<uint2 *>(s_w + kQCoopT);
    uint32_t todo = touched;
    while (todo) {
      const int gc = __builtin_ctz(todo);  // wave-uniform
      todo &= todo - 1u;
      const uint4 xv = *reinterpret_cast<const uint4 *>(s_x + gc * 8);
      const uint32_t xt[8] = {xv.x & 0xffffu, xv.x >> 16, xv.y & 0xffffu, xv.y >> 16, xv.z & 0xffffu, xv.z >> 16, xv.w & 0xffffu, xv.w >> 16};
      const uint32_t dgc = (uint32_t)(delm >> (gc * 8)) & 0xffu;
      uint32_t st = (uint32_t)lane <= reach ? (uint32_t)lane : 0u;  // (no such state: any row inside the table will do)
      uint32_t lo = 0, hi = 0;
#pragma unroll
      for (int t = 0; t < 8; t++) {
        if (!((dgc >> t) & 1u)) {  // uniform: a deleted column costs nothing
          const uint32_t row = (t == 0 && init0 && gc == 0) ? 0u : st;
          st = lds[init_off + __umul24(row, 100u) + xt[t]];
        }
        if (t < 4) lo |= st << (8 * t);
        else hi |= st << (8 * (t - 4));
      }
      wave_sync();
      s_t[lane] = make_uint2(lo, hi);
      wave_sync();
      if (g == gc) {
#pragma unroll
        for (int k = 0; k < 8; k++)
          if ((uint32_t)k < n_chains) {
            const uint2 c2 = s_t[j + 8 * k];
            C.lo[k] = c2.x;
            C.hi[k] = c2.y;
          }
      }
    }
  }
  // the groups' true start states: eight dependent steps on the scalar unit, as in coop_chain (byte k & 3 of ends[k >> 2] = the
  // state chain k ends in); then the lane that walked a group from its true start state hands its chains to LDS
  const uint32_t ends_a = __builtin_amdgcn_perm(C.hi[1], C.hi[0], 0x0c0c0703u) | (__builtin_amdgcn_perm(C.hi[3], C.hi[2], 0x0c0c0703u) << 16);
  const uint32_t ends_b = __builtin_amdgcn_perm(C.hi[5], C.hi[4], 0x0c0c0703u) | (__builtin_amdgcn_perm(C.hi[7], C.hi[6], 0x0c0c0703u) << 16);
  uint32_t st0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_in);
  uint64_t starts = 0;  // byte gg = the state in front of group gg
#pragma unroll
  for (int gg = 0; gg < 8; gg++) {
    starts |= (uint64_t)st0 << (8 * gg);
    const int src = (int)(gg * 8 + (st0 & 7u));
    const uint32_t ea = (uint32_t)__builtin_amdgcn_readlane((int)ends_a, src), eb = (uint32_t)__builtin_amdgcn_readlane((int)ends_b, src);
    st0 = (((st0 & 32u) ? eb : ea) >> (((st0 >> 3) & 3u) * 8u)) & 0xffu;
  }
  const uint32_t sg = (uint32_t)(starts >> (8 * g)) & 0xffu;
  wave_sync();
  if ((uint32_t)j == (sg & 7u)) {
    uint4 *dst = reinterpret_cast<uint4 *>(s_st + g * 64);
    dst[0] = make_uint4(C.lo[0], C.hi[0], C.lo[1], C.hi[1]);
    if (n_chains > 2) dst[1] = make_uint4(C.lo[2], C.hi[2], C.lo[3], C.hi[3]);
    if (n_chains > 4) dst[2] = make_uint4(C.lo[4], C.hi[4], C.lo[5], C.hi[5]);
    if (n_chains > 6) dst[3] = make_uint4(C.lo[6], C.hi[6], C.lo[7], C.hi[7]);
  }
  wave_sync();
  return s_st[g * 64 + j + (sg & 0x38u)];
}

__device__ __forceinline__ void qcoop_walk_task(const WalkArgs &a, const uint8_t *lds, const uint8_t *s_comp, const uint8_t *s_sub,
                                                const uint32_t *s_row, uint8_t *s_w, int slot, int lane) {
  const uint32_t *hdr = reinterpret_cast<const uint32_t *>(lds);
  const uint32_t reach = hdr[6];
  const uint32_t n_chains = (reach + 8u) >> 3;  // start states 0 .. reach in chains of eight
  QChains chains;
#pragma unroll
  for (int k = 0; k < 8; k++) chains.lo[k] = chains.hi[k] = 0;
  chains.delm = 0;
  const uint32_t has_model = hdr[2], freq_rv = hdr[3];
  uint8_t *s_tr = s_w + kQCoopRows, *s_d = s_w + kQCoopD;
  const int task = __builtin_amdgcn_readfirstlane(a.task_of_slot[slot]);
  if (task < 0) return;
  const int r = task / a.pass_num;
  const uint32_t pass = (uint32_t)(task - r * a.pass_num);
  const uint32_t read_idx = (uint32_t)(a.first_read + r);
  const int L = __builtin_amdgcn_readfirstlane(a.len[r]);
  const int64_t off = a.off[r] + (a.read_base ? a.read_base[r] : 0);
  const bool minus = a.read_minus ? (a.read_minus[r] != 0) : ((read_idx & 1u) == 0);
  const int64_t wave = slot >> 6;
  const int cap_raw = __builtin_amdgcn_readfirstlane(a.wave_cap[wave]);
  const int cap_dw = cap_raw & ~kWaveTransposed;
  const bool transposed = (cap_raw & kWaveTransposed) != 0;
  const size_t row_step = transposed ? 1 : 64;
  uint32_t *maf_read = reinterpret_cast<uint32_t *>(a.scratch + a.wave_off[wave]) + (transposed ? (size_t)(slot & 63) * cap_dw : (size_t)(slot & 63));
  uint32_t *maf_ref = maf_read + (size_t)cap_dw * 64;
  uint32_t *qual_row = maf_ref + (size_t)cap_dw * 64;
  const int cap = cap_dw * 4;  // columns the task's rows hold (k_wave_cap)
  const uint32_t comp_off = minus ? 256u : 0u;
  const int64_t p_first = minus ? (off + L - 1) : off;
  const int64_t p_last = a.ref.len - 1;
  int wb = 0;
  auto ref_at = [&](int t) -> uint32_t {
    int64_t p = minus ? (p_first - t) : (p_first + t);
    p = p < 0 ? 0 : (p > p_last ? p_last : p);
    return a.ref.seq[p];
  };
  uint32_t win_a = ref_at(lane), win_b = ref_at(64 + lane);
  auto win_at = [&](int wi) -> uint32_t {  // window byte at read offset wb + wi, wi in 0 .. 127
    const uint32_t va = (uint32_t)__builtin_amdgcn_ds_bpermute((wi & 63) << 2, (int)win_a);
    const uint32_t vb = (uint32_t)__builtin_amdgcn_ds_bpermute((wi & 63) << 2, (int)win_b);
    return (wi & 64) ? vb : va;
  };

  int m0 = 0, ro0 = 0, q0 = 0, nsub = 0;
  uint32_t st_in = 0, lastq_in = 0;
  uint32_t hpf_in = 0;  // bit 7 of the last consumed reference byte (hp == 11 flag); meaningless while ro0 == 0
  bool more = L > 0;
  while (more) {
    const uint32_t event = (uint32_t)(m0 + lane);
    const U4 w = walk_block(a.seed, a.unit, read_idx, pass, event, 0u);
    if ((m0 & 255) == 0) {  // m0 is a multiple of 64 here: the deletion blocks of columns m0 .. m0 + 255, lane i holds group m0 / 4 + i
      const U4 dq = walk_block(a.seed, a.unit, read_idx, pass, (uint32_t)(m0 >> 2) + (uint32_t)lane, 2u);
      wave_sync();
      *reinterpret_cast<uint4 *>(s_d + lane * 16) = make_uint4(dq.x, dq.y, dq.z, dq.w);
      wave_sync();
    }
    const uint32_t dw = *reinterpret_cast<const uint32_t *>(s_d + (((m0 & 255) + lane) << 2));  // group (event >> 2), word (event & 3)
    const uint32_t x = mod100(w.x), y = mod100(w.y), z = mod1e6(w.z), d = mod1e6(dw);
    const uint32_t y_free = has_model ? 0u : w.y % (freq_rv ? freq_rv : 1u);
    const uint64_t below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // first guess: every column tested against the thresholds of the quality in front of the window
    uint64_t delm = __ballot(event > 0u && d < s_row[lastq_in * 8u + 2u]);
    uint32_t st, qv, raw = 0, roi_u = 0;
    bool is_sub, is_ins, emitted;
    uint64_t valid, consm;
    bool fresh = true;  // the step's first turn: new draws, every group is walked
    for (;;) {
      if (has_model) {
        st = qcoop_chain(lds, a.init_off, n_chains, reach, s_w, x, delm, q0 == 0, st_in, lane, chains, fresh);
        fresh = false;
        qv = lds[a.emis_off + __umul24(st - 1u, 100u) + y];
      } else {  // an accuracy class outside the model's range: the quality from the class's own table, no state (pbsim.cpp:2218-2222)
        st = 0;
        qv = lds[a.freq_off + y_free];
      }
      qv = (qv < 94u) ? qv : 93u;
      const uint4 row = *reinterpret_cast<const uint4 *>(&s_row[qv * 8u]);
      emitted = ((delm >> lane) & 1ull) == 0;
      is_sub = z < row.x;               // pbsim.cpp:2233-2249
      is_ins = !is_sub && z < row.y;    // pbsim.cpp:2250-2258
      // the quality in front of this column: the nearest emitted column of the window, else the one carried in
      const uint64_t em_below = ~delm & below;
      const int src = em_below ? 63 - __builtin_clzll(em_below) : 0;
      const uint32_t lq_w = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)qv);
      const uint32_t lq = em_below ? lq_w : lastq_in;
      const uint32_t thr_norm = s_row[lq * 8u + 2u], thr_none = s_row[lq * 8u + 3u];
      // cursors: a column consumes a reference base unless it is an emitted insertion
      consm = __ballot(!emitted || !is_ins);
      const int roi = ro0 + (int)mbcnt64(consm);
      roi_u = (uint32_t)roi;
      const int wi = roi - wb;  // 0 .. 127 for the columns that count
      raw = win_at(wi);
      // hp flag of the last consumed base in front of this column: the window byte at roi - 1, or the one carried in
      const uint32_t prevb = win_at((wi - 1) & 127);
      const uint32_t hpf = (roi == ro0) ? hpf_in : (prevb & 0x80u);
      const uint32_t thr = (roi == 0) ? thr_none : (hpf ? 0u : thr_norm);
      const bool counts = roi < L && m0 + lane < cap;
      valid = __ballot(counts);
      const uint64_t newm = __ballot(event > 0u && d < thr);
      if (((newm ^ delm) & valid) == 0) break;
      delm = newm;
    }
    const int nv = __builtin_popcountll(valid);
    const bool deleted = !emitted;
    // ---- emit
    const uint32_t nt = s_comp[raw + comp_off];
    uint32_t subb = s_sub[nt * 4u + mod3(w.w)];
    if (emitted && is_sub && subb == 0) {  // non-ACGT reference base: one more draw (rare)
      const U4 v = walk_block(a.seed, a.unit, read_idx, pass, event, 1u);
      subb = (kATGC >> ((v.x & 3u) * 8u)) & 0xffu;
    }
    const uint32_t insb = (w.w & 4u) ? nt : ((kATGC >> ((w.w & 3u) * 8u)) & 0xffu);
    const uint32_t b = is_sub ? subb : is_ins ? insb : nt;
    const uint32_t rb = deleted ? 0u : b;
    const uint32_t fb = (!deleted && is_ins) ? (uint32_t)'-' : nt;
    const uint32_t qb = deleted ? 0u : (qv + 33u);
    wave_sync();
    s_tr[lane] = (uint8_t)rb;
    s_tr[64 + lane] = (uint8_t)fb;
    s_tr[128 + lane] = (uint8_t)qb;
    wave_sync();
    if (lane * 4 < nv) {
      const uint32_t dr = reinterpret_cast<const uint32_t *>(s_tr)[lane];
      const uint32_t df = reinterpret_cast<const uint32_t *>(s_tr + 64)[lane];
      const uint32_t dqv = reinterpret_cast<const uint32_t *>(s_tr + 128)[lane];
      scratch_store(&maf_read[(size_t)((m0 >> 2) + lane) * row_step], dr);
      scratch_store(&maf_ref[(size_t)((m0 >> 2) + lane) * row_step], df);
      scratch_store(&qual_row[(size_t)((m0 >> 2) + lane) * row_step], dqv);
    }
    // (the ordered sum of error probabilities: k_qshmm_coop_qsum, from the quality row stored above)
    const uint64_t em_valid = ~delm & valid;
    nsub += __builtin_popcountll(__ballot(emitted && is_sub) & valid);
    q0 += __builtin_popcountll(em_valid);
    const int ro_new = ro0 + __builtin_popcountll(consm & valid);
    // carries into the next step: state and quality of the last column that had them, flag of the last consumed base
    if (nv > 0) st_in = (uint32_t)__builtin_amdgcn_readlane((int)st, nv - 1);
    if (em_valid) lastq_in = (uint32_t)__builtin_amdgcn_readlane((int)qv, 63 - __builtin_clzll(em_valid));
    if (ro_new > ro0) {  // the last consuming column's base sits at read offset ro_new - 1
      const uint32_t lastb = win_at((ro_new - 1 - wb) & 127);
      hpf_in = (uint32_t)__builtin_amdgcn_readfirstlane((int)lastb) & 0x80u;
    }
    ro0 = ro_new;
    m0 += nv;
    more = nv == 64 && ro0 < L && m0 < cap;
    if (more && ro0 - wb >= 64) {
      wb += 64;
      win_a = win_b;
      win_b = ref_at(wb + 64 + lane);
    }
    (void)roi_u;
  }
  if (lane == 0) {
    if (ro0 < L) atomicOr(&a.flags->error, kErrScratchOverflow);
    a.out_len[task] = q0;
    a.maf_len[task] = m0;
    a.nsub[task] = nsub;
    a.nins[task] = m0 - ro0;
    a.ndel[task] = m0 - q0;
    note_row_need(a.flags, m0, L);
  }
}

__global__ __launch_bounds__(kWG) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_walk_qshmm_coop(WalkArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (a.flags->error & kErrScratchBudget) return;
  uint8_t *s_comp = lds + a.stride;
  uint8_t *s_sub = s_comp + 512;
  uint32_t *s_row = reinterpret_cast<uint32_t *>(s_sub + 1024);   // [94][8]: sub_thre, ins_thre, del (hp 1..10), del (none) | qprob f64 | -
  uint8_t *s_w = reinterpret_cast<uint8_t *>(s_row + 94 * 8) + wv * kQCoopWaveLds;
  for (int i = tid; i < 94; i += kWG) {
    s_row[i * 8 + 0] = a.sub_thre[i];
    s_row[i * 8 + 1] = a.ins_thre[i];
    s_row[i * 8 + 2] = a.del_thr[i * 12 + 1];
    s_row[i * 8 + 3] = a.del_thr[i * 12 + 0];
    *reinterpret_cast<double *>(&s_row[i * 8 + 4]) = a.qprob[i];
    s_row[i * 8 + 6] = s_row[i * 8 + 7] = 0;
  }
  __shared__ int s_unit;  // a.coop_dynamic: units drawn from a counter, as in k_walk_errhmm_coop
  unsigned long long *ticket = reinterpret_cast<unsigned long long *>(&a.flags->sums[7]);
  int staged = -1, c = 0, ubase = 0;
  for (int u = blockIdx.x;; u += gridDim.x) {
    if (a.coop_dynamic) {
      __syncthreads();
      if (tid == 0) s_unit = (int)atomicAdd(ticket, 1ull);
      __syncthreads();
      u = s_unit;
    }
    int nc = 0;
    for (; c < a.ncls; c++) {
      nc = a.coop_end[c] - a.class_start[c];
      nc = nc > 0 ? nc : 0;
      const int nu = (nc + 3) >> 2;
      if (u < ubase + nu) break;
      ubase += nu;
    }
    if (c >= a.ncls) break;
    if (staged != c) {
      __syncthreads();  // the other waves may still read the previous class
      stage_class(a, c, lds, s_comp, s_sub, tid, true);
      __syncthreads();
      staged = c;
    }
    const int k = (u - ubase) * 4 + wv;
    if (k < nc) qcoop_walk_task(a, lds, s_comp, s_sub, s_row, s_w, a.class_start[c] + k, lane);
  }
}

// The ordered sum of error probabilities of the wave-walked tasks (pbsim.cpp:2309-2313; the %f report and the accuracy histogram
// depend on its exact value): a lane per task over the quality row the wave walker stored, in column order.  Inside the wave
// walker's step it was 64 dependent f64 additions and 128 v_readlane per 64 columns -- a fifth of the step; a lane's additions
// are as dependent, but 64 tasks at a time, and a task of 9 000 columns takes 40 us.
__global__ __launch_bounds__(256) void k_qshmm_coop_qsum(WalkArgs a, int n_slots) {
  __shared__ double s_qprob[94];
  for (int i = threadIdx.x; i < 94; i += 256) s_qprob[i] = a.qprob[i];
  __syncthreads();
  const int slot = (int)(blockIdx.x * 256u + threadIdx.x);
  if (slot >= n_slots || (a.flags->error & kErrScratchBudget)) return;  // (no room for the batch's rows: nothing was walked)
  bool walked = false;  // by the wave walker: the first coop_end[c] - class_start[c] slots of class c
  for (int c = 0; c < a.ncls; c++) walked = walked || (slot >= a.class_start[c] && slot < a.coop_end[c]);
  if (!walked) return;
  const int task = a.task_of_slot[slot];
  if (task < 0) return;
  const int64_t wave = slot >> 6;
  const int cap_raw = a.wave_cap[wave];
  const int cap_dw = cap_raw & ~kWaveTransposed;
  const bool transposed = (cap_raw & kWaveTransposed) != 0;
  const size_t row_step = transposed ? 1 : 64;
  const uint32_t *maf_read = reinterpret_cast<const uint32_t *>(a.scratch + a.wave_off[wave]) + (transposed ? (size_t)(slot & 63) * cap_dw : (size_t)(slot & 63));
  const uint32_t *qual_row = maf_read + (size_t)cap_dw * 128;
  const int n = a.maf_len[task];
  double sum = 0.0;
  for (int m = 0; m < n; m += 4) {
    uint32_t v = qual_row[(size_t)(m >> 2) * row_step];
    const int k = n - m < 4 ? n - m : 4;
    for (int i = 0; i < k; i++, v >>= 8) {
      const uint32_t q = v & 0xffu;  // 0: a deleted column
      if (q) sum += s_qprob[q - 33u];
    }
  }
  a.qsum[task] = sum;
}

// ---------------------------------------------------------------------------
// K2s: the sampling method's walk (pbsim.cpp:1749-1834).  The quality string comes
// from a sample FASTQ instead of an HMM; every filtered string is used for several
// reads in a row and is cut to the length of the read it has just produced
// (`mut.qc[read_offset] = '\0'`, :1834), so the copies of one string form a chain:
// they are walked one after the other (the strings of a chunk run in parallel, the
// chain inside a string is the reference's own serial dependence).  Copy k of the 64
// strings of line wave w lands in "virtual wave" vbase[w] + k of the scratch pool, so
// the text kernels see ordinary tasks.  k_walk_sample has two paths in one launch:
// one WAVE per string (scoop_walk_string below, the default for every string) and
// one LANE per string (the kernel's second half; PBSIM_COOP_LEN and the tests).
//
//   column m:  m >= 1 -> deletion test  D.x % 1e6 < del_thr[qc[q-1]][hp(last consumed base)]   (:1816-1831)
//              else      error class    E.z % 1e6 vs sub_thre / ins_thre of qc[q]                (:1779-1810)
//   ends when the reference window OR the quality string is used up                              (:1776)
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// K2sc (round 3): the strings of the chunk's first n_coop_waves line waves (by default all) get one WAVE each, 64 columns per step -- the chain of a
// string's copies is serial, and so is a copy's walk in the lane version (0.6 us per column on a GPU that the strings of a
// profile cannot fill: a 60 000-character string held the launch for 40-80 ms).  There is no state chain here; what a column
// decides depends on the columns in front of it only through two cursors, q (characters of the string used) and ro
// (reference bases used), and a column is deleted or not by a test whose thresholds come from the character at q - 1 and the
// homopolymer flag at ro - 1.  A step iterates over the deletion mask to the sequential walk's fixed point, exactly like the
// QSHMM wave walker (column p's outcome depends on columns < p only: every turn extends the correct prefix).
// LDS of one wave: 128-byte rings of the reference window and the string window, three output rows of 64 bytes, the deletion
// draws of 256 columns (one Philox block holds four columns' draws: the 64 lanes compute the blocks of four steps at once).
// The ordered f64 sum of error probabilities is NOT taken here: 64 dependent adds per step were a third of the step's
// instructions; k_sample_qsum takes it afterwards from the quality row, one LANE per read (the same additions in the same order).
// ---------------------------------------------------------------------------
constexpr int kSCoopRef = 0, kSCoopQ = 128, kSCoopRows = 256, kSCoopD = 448, kSCoopHp = 448 + 1024, kSCoopWaveLds = 448 + 1024 + 128;

template <bool kHpBits>
__device__ __forceinline__ void scoop_walk_string(const SampleArgs &a, const uint8_t *s_comp, const uint8_t *s_sub, const uint2 *s_si,
                                                  const uint32_t *s_del, uint8_t *s_w, int lw, int l, int lane) {
  uint8_t *s_ref = s_w + kSCoopRef, *s_q = s_w + kSCoopQ, *s_tr = s_w + kSCoopRows, *s_d = s_w + kSCoopD;
  uint8_t *s_hp = s_w + kSCoopHp;  // !kHpBits (--hp-del-bias): the homopolymer classes come from the byte array, a third ring
  const int line = lw * 64 + l;
  if (line >= a.n_lines) return;
  int Lcur = __builtin_amdgcn_readfirstlane(a.line_len[line]);
  const uint8_t *qsrc = a.quals + a.line_qoff[line];
  const int v0 = __builtin_amdgcn_readfirstlane(a.vbase[lw]), K = __builtin_amdgcn_readfirstlane(a.vbase[lw + 1]) - v0;
  const int64_t G = a.ref.len;
  for (int k = 0; k < K; ++k) {
    const int64_t wave = v0 + k;
    const int task = __builtin_amdgcn_readfirstlane(a.task_of_slot[wave * 64 + l]);
    if (task < 0) break;  // a string's copies take the virtual waves v0 .. v0 + num - 1
    const uint32_t read_idx = (uint32_t)(a.first_read + task);
    int L = Lcur;
    int64_t off = 0;
    if ((int64_t)L >= G) L = (int)G;  // pbsim.cpp:1753-1759
    else  // (the same draw in every lane: one scalar, so that the window's loads take a scalar base)
      off = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(header_block(a.seed, a.unit, read_idx).z % (uint32_t)(G - L + 1)));
    const bool minus = (read_idx & 1u) == 0;  // :1767-1773
    const int cap_dw = __builtin_amdgcn_readfirstlane(a.wave_cap[wave]) & ~kWaveTransposed;
    uint32_t *maf_read = reinterpret_cast<uint32_t *>(a.scratch + a.wave_off[wave]) + (size_t)l * cap_dw;  // rows task by task
    uint32_t *maf_ref = maf_read + (size_t)cap_dw * 64;
    uint32_t *qual_row = maf_ref + (size_t)cap_dw * 64;
    const int cap = cap_dw * 4;
    const uint32_t comp_off = minus ? 256u : 0u;
    const int64_t p_last = G - 1;
    int64_t p_first = minus ? (off + L - 1) : off;
    p_first = p_first < 0 ? 0 : (p_first > p_last ? p_last : p_first);
    // read coordinate t is record position p_first +- t, held at the record's ends: a scalar base and a 32-bit lane offset
    const int64_t room = minus ? p_first : p_last - p_first;
    const uint32_t tmax = (uint32_t)(room > 0x7fffffff ? 0x7fffffff : (room < 0 ? 0 : room));
    const int64_t p0 = minus ? p_first - (int64_t)tmax : p_first;
    const uint8_t *seq0 = a.ref.seq + p0;
    const uint8_t *hp0 = kHpBits ? nullptr : a.ref.hp + p0;
    auto ref_off = [&](int t) -> uint32_t {
      const uint32_t tc = (uint32_t)t < tmax ? (uint32_t)t : tmax;
      return minus ? tmax - tc : tc;
    };
    auto ref_at = [&](int t) -> uint8_t { return seq0[ref_off(t)]; };
    auto hp_at = [&](int t) -> uint8_t { return hp0[ref_off(t)]; };
    auto hp_class = [](uint32_t b) -> uint32_t { return kHpBits ? ((b & 0x80u) ? 11u : 1u) : (b < 12u ? b : 11u); };
    const uint32_t q_last = L > 0 ? (uint32_t)(L - 1) : 0u;
    auto q_at = [&](int t) -> uint8_t { return qsrc[(uint32_t)t < q_last ? (uint32_t)t : q_last]; };
    auto conv = [](uint32_t ch) -> uint32_t { return (ch >= 33u && ch < 127u) ? ch - 33u : 0u; };
    // rings: positions wb .. wb + 127 of the read's reference window, qb .. qb + 127 of the string; the next 64 of each on their way
    int wb = 0, qb = 0;
    wave_sync();
    s_ref[lane] = ref_at(lane);
    s_ref[64 + lane] = ref_at(64 + lane);
    s_q[lane] = q_at(lane);
    s_q[64 + lane] = q_at(64 + lane);
    uint8_t pre_ref = ref_at(128 + lane), pre_q = q_at(128 + lane), pre_hp = 0;
    if (!kHpBits) {
      s_hp[lane] = hp_at(lane);
      s_hp[64 + lane] = hp_at(64 + lane);
      pre_hp = hp_at(128 + lane);
    }
    wave_sync();

    int m0 = 0, ro0 = 0, q0 = 0, nsub = 0;
    uint32_t lastq_in = 0, hp_in = 0;  // nothing emitted, no reference base consumed yet: mut.hp[-1], observed 0 (Q15)
    bool more = L > 0;
    while (more) {
      const uint32_t event = (uint32_t)(m0 + lane);
      const U4 w = walk_block(a.seed, a.unit, read_idx, 0u, event, 0u);
      if ((m0 & 255) == 0) {  // m0 is a multiple of 64 here: the blocks of columns m0 .. m0 + 255, lane i holds group m0 / 4 + i
        const U4 dq = walk_block(a.seed, a.unit, read_idx, 0u, (uint32_t)(m0 >> 2) + (uint32_t)lane, 2u);
        wave_sync();
        *reinterpret_cast<uint4 *>(s_d + lane * 16) = make_uint4(dq.x, dq.y, dq.z, dq.w);
        wave_sync();
      }
      const uint32_t dw = *reinterpret_cast<const uint32_t *>(s_d + (((m0 & 255) + lane) << 2));  // group (event >> 2), word (event & 3)
      const uint32_t z = mod1e6(w.z), d = mod1e6(dw);
      const uint64_t not_first = __ballot(event > 0u), in_cap = __ballot(event < (uint32_t)cap);
      uint64_t delm = __ballot(d < s_del[lastq_in * 12u + hp_in]) & not_first;  // first guess
      uint32_t qv = 0, raw = 0;
      bool is_sub = false, is_ins = false;
      uint64_t valid = 0, consm = 0;
      for (;;) {
        const int qi = q0 + (int)mbcnt64(~delm);  // emitted columns in front of the lane's
        // (every read is unconditional and the two of a pair are issued together: a branch around a read costs its latency)
        const uint32_t c_at = s_q[qi & 127], c_prev = s_q[(qi - 1) & 127];
        qv = conv(c_at);
        const uint32_t lq = (qi == q0) ? lastq_in : conv(c_prev);
        const uint2 si = s_si[qv];
        is_sub = z < si.x;                      // pbsim.cpp:1779-1810
        is_ins = !is_sub && z < si.y;
        consm = delm | __ballot(!is_ins);  // a column uses a reference base unless it is an emitted insertion
        const int roi = ro0 + (int)mbcnt64(consm);
        raw = s_ref[roi & 127];
        const uint32_t prevb = (kHpBits ? s_ref : s_hp)[(roi - 1) & 127];
        const uint32_t hsel = (roi == ro0) ? hp_in : hp_class(prevb);
        const uint32_t thr = s_del[lq * 12u + hsel];  // :1816-1831
        valid = __ballot(roi < L && qi < L) & in_cap;  // :1776
        const uint64_t newm = __ballot(d < thr) & not_first;
        if (((newm ^ delm) & valid) == 0) break;
        delm = newm;
      }
      const int nv = __builtin_popcountll(valid);
      const bool deleted = select_by_mask(delm, 1u, 0u) != 0, emitted = !deleted;
      // ---- emit
      const uint32_t nt = s_comp[raw + comp_off];
      uint32_t subb = s_sub[nt * 4u + mod3(w.w)];
      if (emitted && is_sub && subb == 0) {  // non-ACGT reference base (:1794-1796)
        const U4 v = walk_block(a.seed, a.unit, read_idx, 0u, event, 1u);
        subb = (kATGC >> ((v.x & 3u) * 8u)) & 0xffu;
      }
      const uint32_t insb = (w.w & 4u) ? nt : ((kATGC >> ((w.w & 3u) * 8u)) & 0xffu);
      const uint32_t b = is_sub ? subb : is_ins ? insb : nt;
      wave_sync();
      s_tr[lane] = (uint8_t)(deleted ? 0u : b);
      s_tr[64 + lane] = (uint8_t)((!deleted && is_ins) ? (uint32_t)'-' : nt);
      s_tr[128 + lane] = (uint8_t)(deleted ? 0u : (qv + 33u));
      wave_sync();
      if (lane * 4 < nv) {
        scratch_store(&maf_read[(size_t)(m0 >> 2) + lane], reinterpret_cast<const uint32_t *>(s_tr)[lane]);
        scratch_store(&maf_ref[(size_t)(m0 >> 2) + lane], reinterpret_cast<const uint32_t *>(s_tr + 64)[lane]);
        scratch_store(&qual_row[(size_t)(m0 >> 2) + lane], reinterpret_cast<const uint32_t *>(s_tr + 128)[lane]);
      }
      const uint64_t em_valid = ~delm & valid;
      nsub += __builtin_popcountll(__ballot(emitted && is_sub) & valid);
      const int ro_new = ro0 + __builtin_popcountll(consm & valid);
      if (em_valid) lastq_in = (uint32_t)__builtin_amdgcn_readlane((int)qv, 63 - __builtin_clzll(em_valid));
      if (ro_new > ro0) hp_in = hp_class((uint32_t)__builtin_amdgcn_readfirstlane((int)(kHpBits ? s_ref : s_hp)[(ro_new - 1) & 127]));
      q0 += __builtin_popcountll(em_valid);
      ro0 = ro_new;
      m0 += nv;
      more = nv == 64 && ro0 < L && q0 < L && m0 < cap;
      if (more && ro0 - wb >= 64) {  // positions wb .. wb + 63 are behind the cursor: their slots take wb + 128 .. wb + 191
        wave_sync();
        s_ref[(wb + lane) & 127] = pre_ref;
        if (!kHpBits) s_hp[(wb + lane) & 127] = pre_hp;
        wb += 64;
        pre_ref = ref_at(wb + 128 + lane);
        if (!kHpBits) pre_hp = hp_at(wb + 128 + lane);
        wave_sync();
      }
      if (more && q0 - qb >= 64) {
        wave_sync();
        s_q[(qb + lane) & 127] = pre_q;
        qb += 64;
        pre_q = q_at(qb + 128 + lane);
        wave_sync();
      }
    }
    if (lane == 0) {
      if (ro0 < L && q0 < L) atomicOr(&a.flags->error, kErrScratchOverflow);
      a.span[task] = ro0;         // seq_right - seq_left + 1 = ref_offset (:1846-1847)
      a.off[task] = (int32_t)off;
      a.out_len[task] = q0;
      a.maf_len[task] = m0;
      a.nsub[task] = nsub;
      a.nins[task] = m0 - ro0;
      a.ndel[task] = m0 - q0;   // a.qsum[task]: k_sample_qsum
    }
    Lcur = q0;                    // the string is cut to this read's length (:1834)
  }
}

template <bool kHpBits>
__global__ __launch_bounds__(kWG) void k_walk_sample(SampleArgs a) {
  __shared__ uint8_t s_comp[512];
  __shared__ uint8_t s_sub[1024];
  __shared__ uint32_t s_subt[96], s_ins[96], s_del[94 * 12];
  __shared__ uint2 s_si[96];  // {sub_thre, ins_thre}: one read for the wave walker
  __shared__ double s_qprob[94];
  __shared__ __attribute__((aligned(16))) uint8_t s_wave[(kWG / 64) * kSCoopWaveLds];
  const int tid = threadIdx.x;
  {
    const uint32_t c = (uint32_t)tid;
    const uint32_t base = kHpBits ? (c & 0x7fu) : c;  // bit 7 of a sequence byte is the hp == 11 flag (RefCursor)
    s_comp[c] = (uint8_t)base;
    s_comp[256 + c] = (uint8_t)complement(base);
    const uint32_t t = sub_table(c);
    s_sub[c * 4 + 0] = (uint8_t)(t & 0xffu);
    s_sub[c * 4 + 1] = (uint8_t)((t >> 8) & 0xffu);
    s_sub[c * 4 + 2] = (uint8_t)((t >> 16) & 0xffu);
    s_sub[c * 4 + 3] = 0;
    for (int i = tid; i < 94; i += kWG) {
      s_subt[i] = a.sub_thre[i];
      s_ins[i] = a.ins_thre[i];
      s_si[i] = make_uint2(a.sub_thre[i], a.ins_thre[i]);
      s_qprob[i] = a.qprob[i];
    }
    for (int i = tid; i < 94 * 12; i += kWG) s_del[i] = a.del_thr[i];
  }
  __syncthreads();
  const int lane = tid & 63;
  if ((int)blockIdx.x < a.n_coop_blocks) {
    // the strings of line waves 0 .. n_coop_waves - 1 (the chunk's longest), one per wave, longest first: every wave draws its
    // next string from a counter (flags->sums[7], zero at launch), so a wave that got a long string simply draws fewer
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_units = a.n_coop_waves * 64;
    unsigned long long *ticket = reinterpret_cast<unsigned long long *>(&a.flags->sums[7]);
    for (;;) {
      unsigned long long t = 0;
      if (lane == 0) t = atomicAdd(ticket, 1ull);
      const int u = __builtin_amdgcn_readfirstlane((int)t);
      if (u >= n_units) break;
      scoop_walk_string<kHpBits>(a, s_comp, s_sub, s_si, s_del, s_wave + wv * kSCoopWaveLds, u >> 6, u & 63, lane);
    }
    return;
  }
  const int lw = a.n_coop_waves + ((int)blockIdx.x - a.n_coop_blocks) * (kWG / 64) + (tid >> 6);
  if (lw >= a.n_line_waves) return;
  const int line = lw * 64 + lane;
  const bool has_line = line < a.n_lines;
  int Lcur = has_line ? a.line_len[line] : 0;
  const uint64_t *qsrc = reinterpret_cast<const uint64_t *>(a.quals + (has_line ? a.line_qoff[line] : 0));
  const int v0 = a.vbase[lw], K = a.vbase[lw + 1] - v0;
  const int64_t G = a.ref.len;

  for (int k = 0; k < K; ++k) {
    const int64_t wave = v0 + k;
    const int task = a.task_of_slot[wave * 64 + lane];
    const bool valid = task >= 0;
    const uint32_t read_idx = (uint32_t)(a.first_read + (valid ? task : 0));
    int L = Lcur;
    int64_t off = 0;
    if (valid) {  // pbsim.cpp:1753-1759
      if ((int64_t)L >= G) L = (int)G;
      else off = (int64_t)(header_block(a.seed, a.unit, read_idx).z % (uint32_t)(G - L + 1));
    }
    const bool minus = (read_idx & 1u) == 0;  // :1767-1773
    const int cap_dw = __builtin_amdgcn_readfirstlane(a.wave_cap[wave]) & ~kWaveTransposed;  // wave-uniform: keeps row offsets scalar
    const int64_t woff = ((int64_t)__builtin_amdgcn_readfirstlane((int)(a.wave_off[wave] >> 32)) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)a.wave_off[wave]);
  uint32_t *maf_read = reinterpret_cast<uint32_t *>(a.scratch + woff) + lane;
    uint32_t *maf_ref = maf_read + (size_t)cap_dw * 64;
    uint32_t *qual_row = maf_ref + (size_t)cap_dw * 64;
    const int cap = cap_dw * 4;

    int ro = 0, q = 0, m = 0, nsub = 0, group = 0;
    uint32_t last_q = 0, hp_prev = 0;  // no reference base consumed yet: mut.hp[-1], observed 0 (Q15)
    uint32_t acc_r = 0, acc_f = 0, acc_q = 0;
    double qsum = 0.0;
    bool act = valid && L > 0;
    const uint32_t comp_off = minus ? 256u : 0u;
    RefCursor<kHpBits> cur;
    cur.init(a.ref, off, L, minus, act);
    const WalkLane lane_e = walk_lane(a.seed, read_idx, 0u, 0u);
    const WalkLane lane_d = walk_lane(a.seed, read_idx, 0u, 2u);
    uint64_t qwin = act ? qsrc[0] : 0;
    int qwin_idx = 0;

    while (__any(act)) {
      U4 E[4];
#pragma unroll
      for (int j = 0; j < 4; j++)
        E[j] = walk_block_fast(lane_e, a.seed, a.unit, (uint32_t)__builtin_amdgcn_readfirstlane(group * 4 + j));
      const U4 Dq = walk_block_fast(lane_d, a.seed, a.unit, (uint32_t)__builtin_amdgcn_readfirstlane(group));
      const uint32_t D[4] = {Dq.x, Dq.y, Dq.z, Dq.w};   // deletion tests of the four columns (event = column >> 2)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const U4 w = E[j];
        uint32_t raw, hp;
        cur.at(ro, act, &raw, &hp);
        const uint32_t nt = s_comp[raw + comp_off];
        const bool deleted = (m > 0) & (mod1e6(D[j]) < s_del[last_q * 12u + (hp_prev < 12u ? hp_prev : 11u)]);
        if (act && (q >> 3) != qwin_idx) {  // next 8 quality characters of the string
          qwin_idx = q >> 3;
          qwin = qsrc[qwin_idx];
        }
        uint32_t qv = (uint32_t)(qwin >> ((q & 7) * 8)) & 0xffu;
        qv = (qv >= 33u && qv < 127u) ? qv - 33u : 0u;
        const bool emit = act && !deleted;
        last_q = emit ? qv : last_q;
        qsum += emit ? s_qprob[qv] : 0.0;  // ordered sum, pbsim.cpp:1857-1860
        const uint32_t x = mod1e6(w.z);
        const bool is_sub = x < s_subt[qv];
        const bool is_ins = !is_sub && x < s_ins[qv];
        uint32_t subb = s_sub[nt * 4u + mod3(w.w)];
        if (emit && is_sub && subb == 0) {  // non-ACGT reference base (:1794-1796)
          const U4 v = walk_block(a.seed, a.unit, read_idx, 0u, (uint32_t)(group * 4 + j), 1u);
          subb = (kATGC >> ((v.x & 3u) * 8u)) & 0xffu;
        }
        const uint32_t insb = (w.w & 4u) ? nt : ((kATGC >> ((w.w & 3u) * 8u)) & 0xffu);
        const uint32_t b = is_sub ? subb : is_ins ? insb : nt;
        const uint32_t mr = deleted ? 0u : b;
        const uint32_t mf = (!deleted && is_ins) ? (uint32_t)'-' : nt;
        acc_r |= mr << (8 * j);
        acc_f |= mf << (8 * j);
        acc_q |= (deleted ? 0u : (qv + 33u)) << (8 * j);
        const bool consumed = act && (deleted || !is_ins);
        hp_prev = consumed ? hp : hp_prev;
        q += emit ? 1 : 0;
        nsub += (emit && is_sub) ? 1 : 0;
        ro += consumed ? 1 : 0;
        m += act ? 1 : 0;
        act = act && (ro < L) && (q < L) && (m < cap);
      }
      if (valid && m > group * 4) {
        scratch_store(&maf_read[(size_t)group * 64], acc_r);
        scratch_store(&maf_ref[(size_t)group * 64], acc_f);
        scratch_store(&qual_row[(size_t)group * 64], acc_q);
      }
      cur.refill(act);
      acc_r = 0;
      acc_f = 0;
      acc_q = 0;
      group++;
    }
    if (valid) {
      if (ro < L && q < L) atomicOr(&a.flags->error, kErrScratchOverflow);
      a.span[task] = ro;          // seq_right - seq_left + 1 = ref_offset (:1846-1847)
      a.off[task] = (int32_t)off;
      a.out_len[task] = q;
      a.maf_len[task] = m;
      a.nsub[task] = nsub;
      a.nins[task] = m - ro;
      a.ndel[task] = m - q;
      a.qsum[task] = qsum;
      Lcur = q;                   // the string is cut to this read's length (:1834)
    }
  }
}

// The ordered sum of error probabilities (pbsim.cpp:1857-1860) of the reads the wave walker made: one LANE per read over its
// quality row (rows stored task by task; a deleted column holds 0, an emitted one its character), the additions the lane
// walker performs column by column, in the same order.  `n_slots` = 64 x the virtual waves of the coop line waves.
__global__ __launch_bounds__(256) void k_sample_qsum(SampleArgs a, int64_t n_slots) {
  // indexed by the row's byte itself: 0 (a deleted column) adds 0.0, '!' + q adds qprob[q] -- no test per column
  __shared__ double s_qp[256];
  {
    const int i = threadIdx.x;
    s_qp[i] = (i >= 33 && i < 127) ? a.qprob[i - 33] : 0.0;
  }
  __syncthreads();
  const int64_t slot = (int64_t)blockIdx.x * 256 + (int64_t)threadIdx.x;
  if (slot >= n_slots) return;
  const int task = a.task_of_slot[slot];
  if (task < 0) return;
  const int64_t wave = slot >> 6;
  const int cap_dw = a.wave_cap[wave] & ~kWaveTransposed;
  const uint32_t *row = reinterpret_cast<const uint32_t *>(a.scratch + a.wave_off[wave]) + (size_t)(slot & 63) * cap_dw + (size_t)cap_dw * 128;
  const int n = a.maf_len[task];
  const int n_full = n >> 2;                                             // dwords whose four columns all count
  const uint32_t last_mask = (n & 3) ? (1u << (8 * (n & 3))) - 1u : 0u;  // the columns of the dword behind them that do
  double sum = 0.0;
  // 64 columns per turn as four 16-byte loads (the planner keeps a task-by-task row's capacity a multiple of four dwords, so
  // every row starts on a 16-byte boundary), the next 64 on their way while these are added: a lane's loads are its own row's,
  // one 64-byte line per turn -- 64 different lines per load instruction of the wave, so few, wide loads.  Bytes behind the
  // read's last column are whatever the walk's last step left there: masked.
  const uint4 *row4 = reinterpret_cast<const uint4 *>(row);
  const int n_q = (n + 15) >> 4;
  uint4 cur[4], nxt[4];
#pragma unroll
  for (int i = 0; i < 4; i++) cur[i] = (i < n_q) ? row4[i] : make_uint4(0u, 0u, 0u, 0u);
  for (int g0 = 0; g0 < n_q; g0 += 4) {
#pragma unroll
    for (int i = 0; i < 4; i++) nxt[i] = (g0 + 4 + i < n_q) ? row4[g0 + 4 + i] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      uint32_t w4[4] = {cur[i].x, cur[i].y, cur[i].z, cur[i].w};
      double qp[16];
#pragma unroll
      for (int d = 0; d < 4; d++) {
        const int gi = (g0 + i) * 4 + d;
        w4[d] = gi < n_full ? w4[d] : (gi == n_full ? (w4[d] & last_mask) : 0u);
#pragma unroll
        for (int j = 0; j < 4; j++) qp[d * 4 + j] = s_qp[(w4[d] >> (8 * j)) & 0xffu];
      }
#pragma unroll
      for (int k = 0; k < 16; k++) sum += qp[k];  // in column order; + 0.0 changes nothing
    }
#pragma unroll
    for (int i = 0; i < 4; i++) cur[i] = nxt[i];
  }
  a.qsum[task] = sum;
}

// ---------------------------------------------------------------------------
// exclusive scan of int64 (3 phases)
// ---------------------------------------------------------------------------
constexpr int kScanTile = 2048;  // 256 threads x 8

__global__ __launch_bounds__(256) void k_scan_sums(const int64_t *in, int64_t n, int64_t *sums) {
  short_kernel_priority();
  __shared__ long long s[256];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * 8;
  long long t = 0;
  for (int k = 0; k < 8; k++)
    if (base + k < n) t += in[base + k];
  s[threadIdx.x] = t;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if (threadIdx.x < d) s[threadIdx.x] += s[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) sums[blockIdx.x] = s[0];
}

__global__ __launch_bounds__(kScanBlock) void k_scan_single(int64_t *sums, int64_t n, int64_t *total) {
  short_kernel_priority();
  __shared__ long long s_part[kScanBlock];
  __shared__ long long s_base;
  const int tid = threadIdx.x;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int64_t i0 = 0; i0 < n; i0 += kScanBlock) {
    const int64_t i = i0 + tid;
    const long long v = (i < n) ? sums[i] : 0;
    s_part[tid] = v;
    __syncthreads();
    for (int d = 1; d < kScanBlock; d <<= 1) {
      const long long t = (tid >= d) ? s_part[tid - d] : 0;
      __syncthreads();
      s_part[tid] += t;
      __syncthreads();
    }
    if (i < n) sums[i] = s_base + s_part[tid] - v;
    __syncthreads();
    if (tid == kScanBlock - 1) s_base += s_part[kScanBlock - 1];
    __syncthreads();
  }
  if (tid == 0 && total) *total = s_base;
}

__global__ __launch_bounds__(256) void k_scan_apply(const int64_t *in, int64_t *out, int64_t n, const int64_t *sums) {
  short_kernel_priority();
  __shared__ long long s[256];
  const int tid = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)tid * 8;
  long long v[8], t = 0;
  for (int k = 0; k < 8; k++) {
    v[k] = (base + k < n) ? in[base + k] : 0;
    t += v[k];
  }
  s[tid] = t;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const long long x = (tid >= d) ? s[tid - d] : 0;
    __syncthreads();
    s[tid] += x;
    __syncthreads();
  }
  long long run = sums[blockIdx.x] + s[tid] - t;
  for (int k = 0; k < 8; k++) {
    if (base + k < n) out[base + k] = run;
    run += v[k];
  }
}

// ---------------------------------------------------------------------------
// quota cut (pbsim.cpp:3792-3800, 3989-3991)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather_pass0(const int32_t *out_len, int64_t n_reads, int32_t pass_num,
                                                        int64_t *cum) {
  short_kernel_priority();
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r < n_reads) cum[r] = out_len[r * pass_num];
}

__global__ void k_quota_init(DeviceFlags *flags, int64_t n_reads) { flags->n_final = n_reads; }

__global__ __launch_bounds__(256) void k_quota_find(const int64_t *cum, const int32_t *rawlen, int64_t n_reads,
                                                      int64_t before, int64_t quota, DeviceFlags *flags) {
  short_kernel_priority();
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n_reads) return;
  const int64_t t = before + cum[r];
  if (!(t < quota) || (rawlen && t + rawlen[r] > quota)) atomicMin((long long *)&flags->n_final, (long long)r);
}

// ---------------------------------------------------------------------------
// K3: text.  Sizes per task, then one workgroup per task writes the bytes.
// ---------------------------------------------------------------------------
struct TaskText {
  int name_len;   // bytes of the reference name printed on the MAF reference line
  int idl;        // strlen(id)
  int q0;         // 1 + count_digit(readnum)        (digit_num2[0], pbsim.cpp:4031)
  int r0;         // 3 ("ref") or strlen(transcript id)
  int w0, w1, w2, w3, r1, r2, r3, q2;
  int64_t start0, span, reflen, readnum;
};

__device__ __forceinline__ int id_length(const TextArgs &a, int64_t readnum, int pass) {
  int n = a.id_prefix_len + 1 + count_digit(readnum);
  if (a.is_wgs) n += count_digit((int64_t)a.unit);
  if (a.pass_num > 1) n += 1 + count_digit(pass);
  return n;
}

// id = "<prefix><rec>_<n>" | "<prefix><rec>/<n>/<h>" (wgs, pbsim.cpp:4013,4016)
//      "<prefix>_<n>"      | "<prefix>/<n>/<h>"      (trans/templ, :4703,4706)
__device__ __forceinline__ int put_id(char *dst, const TextArgs &a, int64_t readnum, int pass) {
  int n = 0;
  for (int i = 0; i < a.id_prefix_len; i++) dst[n++] = a.id_prefix[i];
  if (a.is_wgs) n += put_dec(dst + n, (int64_t)a.unit);
  dst[n++] = (a.pass_num > 1) ? '/' : '_';
  n += put_dec(dst + n, readnum);
  if (a.pass_num > 1) {
    dst[n++] = '/';
    n += put_dec(dst + n, pass);
  }
  return n;
}

__device__ __forceinline__ void task_text(const TextArgs &a, int64_t r, int pass, int q, TaskText *t) {
  t->readnum = a.first_read + r;
  t->idl = id_length(a, t->readnum, pass);
  t->q0 = 1 + count_digit(t->readnum);
  t->start0 = a.off[r];
  t->span = a.len[r];
  if (a.read_unit) {
    const int u = a.read_unit[r];
    t->reflen = a.unit_len[u];
    int n = 0;
    const char *nm = a.unit_names + (size_t)u * 132;
    while (nm[n]) n++;
    t->name_len = n;
    t->r0 = a.name_pad3 ? 3 : n;
  } else {
    t->reflen = a.ref_len;
    t->name_len = 3;
    t->r0 = 3;
  }
  t->r1 = count_digit(t->start0);
  t->r2 = count_digit(t->span);
  t->r3 = count_digit(t->reflen);
  t->q2 = count_digit(q);
  t->w0 = max(t->r0, t->q0);
  t->w1 = max(t->r1, 1);
  t->w2 = max(t->r2, t->q2);
  t->w3 = max(t->r3, t->q2);
}

// ---- BAM records (SAMv1 section 4.2) of the unaligned subreads the reference pipes into
// `samtools view -b` (pbsim.cpp:4016-4027).  Integer tags take the smallest type that holds
// the value, as htslib's SAM parser chooses them.
__device__ __forceinline__ int bam_int_size(int64_t v) {
  if (v < 0) return v >= -128 ? 1 : v >= -32768 ? 2 : 4;
  return v < 256 ? 1 : v < 65536 ? 2 : 4;
}
__device__ __forceinline__ char *bam_put_int_tag(char *o, char t0, char t1, int64_t v) {
  *o++ = t0;
  *o++ = t1;
  const int n = bam_int_size(v);
  *o++ = (v < 0) ? (n == 1 ? 'c' : n == 2 ? 's' : 'i') : (n == 1 ? 'C' : n == 2 ? 'S' : 'I');
  for (int i = 0; i < n; i++) *o++ = (char)((uint64_t)v >> (8 * i));
  return o;
}
__device__ __forceinline__ char *bam_put_u32(char *o, uint32_t v) {
  for (int i = 0; i < 4; i++) *o++ = (char)(v >> (8 * i));
  return o;
}
__device__ __forceinline__ int64_t bam_record_size(int idl, int q, int64_t readnum) {
  const int64_t tags = 4                    // cx:C
                       + (8 + (int64_t)q)   // ip:B:C
                       + 4                  // np:C
                       + (8 + (int64_t)q)   // pw:B:C
                       + 4                  // qs:C
                       + 3 + bam_int_size((int64_t)q - 1)  // qe
                       + 7                  // rq:f
                       + 24                 // sn:B:f x4
                       + 3 + bam_int_size(readnum)         // zm
                       + 12;                // RG:Z:ffffffff
  return 4 + 32 + (idl + 1) + (q + 1) / 2 + q + tags;
}

__global__ __launch_bounds__(256) void k_text_sizes(TextArgs a, DeviceFlags *flags) {
  short_kernel_priority();
  __shared__ unsigned long long s_sum[3];  // one global atomic per workgroup and counter: all tasks hit the same line
  if (threadIdx.x < 3) s_sum[threadIdx.x] = 0;
  __syncthreads();
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n_tasks = a.n_reads * a.pass_num;
  if (t < n_tasks) {
    const int64_t r = t / a.pass_num;
    const int pass = (int)(t - r * a.pass_num);
    const int q = a.out_len[t], m = a.maf_len[t];
    TaskText x;
    task_text(a, r, pass, q, &x);
    int64_t rt;
    if (a.bam) {
      rt = bam_record_size(x.idl, q, x.readnum);
    } else if (a.pass_num == 1) {
      rt = 2LL * x.idl + 2LL * q + 6;  // "@id\n" seq "\n+id\n" qual "\n"
    } else {
      // pbsim.cpp:4017-4027
      rt = (int64_t)x.idl + PB_LEN(PB_SAM_MID) + q + 1 + q + PB_LEN(PB_SAM_IP) + 2LL * q + PB_LEN(PB_SAM_PW) + 2LL * q +
           PB_LEN(PB_SAM_T1) + dec_len((int64_t)q - 1) + PB_LEN(PB_SAM_T2) + a.rq_len + PB_LEN(PB_SAM_T3) +
           count_digit(x.readnum) + PB_LEN(PB_SAM_T4);
    }
    const int64_t mt = (11LL + (x.name_len - x.r0) + x.w0 + x.w1 + x.w2 + x.w3 + m) +
                       (10LL + x.idl + (x.w0 - x.q0) + x.w1 + x.w2 + x.w3 + m);
    a.read_text_len[t] = rt;
    a.maf_text_len[t] = mt;
    atomicAdd(&s_sum[0], (unsigned long long)q);
    atomicAdd(&s_sum[1], (unsigned long long)a.len[r]);
    atomicAdd(&s_sum[2], (unsigned long long)m);
  }
  __syncthreads();
  if (threadIdx.x < 3) atomicAdd((unsigned long long *)&flags->sums[3 + threadIdx.x], s_sum[threadIdx.x]);
}

// ---- small pieces: one thread per task writes every header/separator and
// records where the four (six for SAM) big rows of the task go
__device__ __forceinline__ char *g_lit(char *o, const char *lit, int n) {
  for (int i = 0; i < n; i++) o[i] = lit[i];
  return o + n;
}
__device__ __forceinline__ char *g_pad(char *o, int n) {
  for (int i = 0; i < n; i++) o[i] = ' ';
  return o + (n > 0 ? n : 0);
}

__global__ __launch_bounds__(256) void k_text_headers(TextArgs a) {
  short_kernel_priority();
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n_tasks = a.n_reads * a.pass_num;
  if (t >= n_tasks) return;
  const int64_t r = t / a.pass_num;
  const int pass = (int)(t - r * a.pass_num);
  const int q = a.out_len[t], m = a.maf_len[t];
  const int64_t readnum = a.first_read + r;
  const bool minus = a.read_minus ? (a.read_minus[r] != 0) : ((readnum & 1) == 0);
  int64_t *rd = a.row_dst + t * 6;
  char idbuf[160];
  const int idl = put_id(idbuf, a, readnum, pass);

  // ---------------- FASTQ (pbsim.cpp:4013-4014) / SAM (:4016-4027) ----------------
  char *base = a.read_text + a.read_text_off[t];
  char *o = base;
  if (a.bam) {
    const int64_t size = bam_record_size(idl, q, readnum);
    o = bam_put_u32(o, (uint32_t)(size - 4));  // block_size
    o = bam_put_u32(o, 0xffffffffu);           // refID -1
    o = bam_put_u32(o, 0xffffffffu);           // pos -1
    *o++ = (char)(idl + 1);                    // l_read_name
    *o++ = (char)255;                          // mapq
    *o++ = (char)(4680 & 0xff);                // bin of an unplaced read = reg2bin(-1, 0)
    *o++ = (char)(4680 >> 8);
    *o++ = 0;                                  // n_cigar_op
    *o++ = 0;
    *o++ = 4;                                  // flag 4
    *o++ = 0;
    o = bam_put_u32(o, (uint32_t)q);           // l_seq
    o = bam_put_u32(o, 0xffffffffu);           // next_refID
    o = bam_put_u32(o, 0xffffffffu);           // next_pos
    o = bam_put_u32(o, 0);                     // tlen
    o = g_lit(o, idbuf, idl);
    *o++ = 0;
    rd[0] = o - a.read_text;                   // packed bases, (q+1)/2 bytes
    o += (q + 1) / 2;
    rd[1] = o - a.read_text;                   // phred qualities
    o += q;
    o = bam_put_int_tag(o, 'c', 'x', 3);
    *o++ = 'i'; *o++ = 'p'; *o++ = 'B'; *o++ = 'C';
    o = bam_put_u32(o, (uint32_t)q);
    rd[4] = o - a.read_text;
    o += q;
    o = bam_put_int_tag(o, 'n', 'p', 1);
    *o++ = 'p'; *o++ = 'w'; *o++ = 'B'; *o++ = 'C';
    o = bam_put_u32(o, (uint32_t)q);
    rd[5] = o - a.read_text;
    o += q;
    o = bam_put_int_tag(o, 'q', 's', 0);
    o = bam_put_int_tag(o, 'q', 'e', (int64_t)q - 1);
    *o++ = 'r'; *o++ = 'q'; *o++ = 'f';
    o = bam_put_u32(o, a.rq_bits);
    *o++ = 's'; *o++ = 'n'; *o++ = 'B'; *o++ = 'f';
    o = bam_put_u32(o, 4);
    for (int i = 0; i < 4; i++) o = bam_put_u32(o, 0x41200000u);  // 10.0f
    o = bam_put_int_tag(o, 'z', 'm', readnum);
    *o++ = 'R'; *o++ = 'G'; *o++ = 'Z';
    o = g_lit(o, "ffffffff", 8);
    *o++ = 0;
  } else if (a.pass_num == 1) {
    *o++ = '@';
    o = g_lit(o, idbuf, idl);
    *o++ = '\n';
    rd[0] = o - a.read_text;
    o += q;
    *o++ = '\n';
    *o++ = '+';
    o = g_lit(o, idbuf, idl);
    *o++ = '\n';
    rd[1] = o - a.read_text;
    o += q;
    *o++ = '\n';
    rd[4] = rd[5] = 0;
  } else {
    o = g_lit(o, idbuf, idl);
    o = g_lit(o, PB_SAM_MID, PB_LEN(PB_SAM_MID));
    rd[0] = o - a.read_text;
    o += q;
    *o++ = '\t';
    rd[1] = o - a.read_text;
    o += q;
    o = g_lit(o, PB_SAM_IP, PB_LEN(PB_SAM_IP));
    rd[4] = o - a.read_text;
    o += 2 * q;
    o = g_lit(o, PB_SAM_PW, PB_LEN(PB_SAM_PW));
    rd[5] = o - a.read_text;
    o += 2 * q;
    o = g_lit(o, PB_SAM_T1, PB_LEN(PB_SAM_T1));
    o += put_dec(o, (int64_t)q - 1);
    o = g_lit(o, PB_SAM_T2, PB_LEN(PB_SAM_T2));
    o = g_lit(o, a.rq_text, a.rq_len);
    o = g_lit(o, PB_SAM_T3, PB_LEN(PB_SAM_T3));
    o += put_dec(o, readnum);
    o = g_lit(o, PB_SAM_T4, PB_LEN(PB_SAM_T4));
  }

  // ---------------- MAF (pbsim.cpp:4030-4078) ----------------
  TaskText x;
  task_text(a, r, pass, q, &x);
  o = a.maf_text + a.maf_text_off[t];
  *o++ = 'a';
  *o++ = '\n';
  *o++ = 's';
  *o++ = ' ';
  if (a.read_unit) {
    o = g_lit(o, a.unit_names + (size_t)a.read_unit[r] * 132, x.name_len);
  } else {
    *o++ = 'r';
    *o++ = 'e';
    *o++ = 'f';
  }
  o = g_pad(o, x.w0 - x.r0);
  o = g_pad(o, x.w1 - x.r1);
  *o++ = ' ';
  o += put_dec(o, x.start0);
  o = g_pad(o, x.w2 - x.r2);
  *o++ = ' ';
  o += put_dec(o, x.span);
  *o++ = ' ';
  *o++ = '+';
  o = g_pad(o, x.w3 - x.r3);
  *o++ = ' ';
  o += put_dec(o, x.reflen);
  *o++ = ' ';
  rd[2] = o - a.maf_text;
  o += m;
  *o++ = '\n';
  *o++ = 's';
  *o++ = ' ';
  o = g_lit(o, idbuf, idl);
  o = g_pad(o, x.w0 - x.q0);
  o = g_pad(o, x.w1 - 1);
  *o++ = ' ';
  *o++ = '0';
  o = g_pad(o, x.w2 - x.q2);
  *o++ = ' ';
  o += put_dec(o, q);
  *o++ = ' ';
  *o++ = minus ? '-' : '+';
  o = g_pad(o, x.w3 - x.q2);
  *o++ = ' ';
  o += put_dec(o, q);
  *o++ = ' ';
  rd[3] = o - a.maf_text;
  o += m;
  *o++ = '\n';
  *o++ = '\n';
}

// ---- big rows: one workgroup per (scratch wave, pass).  A 64-task x 256-column tile is read as
// coalesced 256-byte lines of the wave-transposed scratch and turned through LDS (one row per
// task).  Each of the 4 waves owns 16 tasks and serves FOUR of them per step: 16 lanes per task,
// one 16-byte chunk per lane, so a 256-column window of a task leaves as 16 aligned 16-byte
// stores.  A task's destination is not 16-byte aligned with its columns; instead of patching
// edges in every tile, the row keeps the last 16 bytes of the previous tile in front of the new
// one (carry) and every tile writes whole destination chunks; only a row's first and last
// <= 15 bytes go out as bytes.  '-' strand rows are reverse-complemented on the way
// (pbsim.cpp:3981-3984): same columns, destination chunks descending, bytes swapped + complemented.
//   pass 0  MAF read row -> MAF read line (0 prints as '-') AND, with the deleted columns (byte 0)
//           squeezed out, the read bases (FASTQ / SAM sequence; BAM: parked in the pw array)
//   pass 1  MAF reference row -> MAF reference line
//   pass 2  QSHMM quality row, squeezed the same way -> quality line
// Constant fills ('!' qualities, SAM ",9" tags, BAM ip array) are k_text_fill's.
constexpr int kTileStride = 69;  // dwords per task row in LDS: [16 carry bytes | 256 tile bytes] + 1 (5 mod 64: conflict-free turn)
constexpr int kOutStride = 76;   // squeezed row: [<= 15 pending bytes | <= 256 kept bytes] + the dword an OR may spill into; 16-byte aligned

// zero bytes (deleted columns) of a MAF read row print as '-'
__device__ __forceinline__ uint32_t dash_zero_bytes(uint32_t w) { return w | ((eq_bytes(w, 0u) >> 7) * 0x2Du); }

// LDS hand-off between the lanes of ONE wave (its tile rows and out rows are private to it): the LDS pipeline
// executes a wave's ds operations in order, so only the compiler has to be kept from reordering them.  A
// workgroup-scope fence here would also wait for the wave's outstanding global stores (vmcnt) at every step.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// inclusive prefix sum over the 16 lanes of a DPP row
__device__ __forceinline__ int row16_scan(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
  return x;
}

// One MAF line of one task for the tile of columns [256 t, 256 t + 256), served by the 16 lanes `sub` of a group.
// `row` = [carry | tile] in LDS; D = byte offset of the line in `text`, m = its length, minus = reverse-complement.
__device__ __forceinline__ void write_maf_tile(char *text, long long D, int m, int t, const uint32_t *row, int minus,
                                               bool dash_zero, int sub) {
  // h = bytes in front of the first whole destination chunk (in walking direction), written as bytes by tile 0
  int h = minus ? (int)((D + m) & 15) : (int)((16 - (D & 15)) & 15);
  h = h < m ? h : m;
  const int cons = (t == 0) ? h : 256 * t - ((16 - h) & 15);  // columns already written before this tile
  const int avail = (m < 256 * (t + 1)) ? m : 256 * (t + 1);
  const uint8_t *row8 = reinterpret_cast<const uint8_t *>(row);
  if (t == 0 && sub < h) {
    uint32_t v = row8[16 + sub];
    if (minus) v = complement(v);
    if (dash_zero && v == 0) v = '-';
    text[minus ? D + m - 1 - sub : D + sub] = (char)v;
  }
  const int n_ch = (avail - cons) >> 4;
  if (sub < n_ch) {
    const int b = 16 + cons - 256 * t + 16 * sub;  // byte index of the chunk's first column in the row buffer (>= 1)
    const uint32_t *src = row + (b >> 2);
    const uint32_t sh = (uint32_t)(b & 3);
    const uint32_t d0 = src[0], d1 = src[1], d2 = src[2], d3 = src[3], d4 = src[4];
    const uint32_t a0 = __builtin_amdgcn_alignbyte(d1, d0, sh), a1 = __builtin_amdgcn_alignbyte(d2, d1, sh),
                   a2 = __builtin_amdgcn_alignbyte(d3, d2, sh), a3 = __builtin_amdgcn_alignbyte(d4, d3, sh);
    // the line's bytes in destination order: '-' strand = descending columns
    const uint32_t rev = minus ? 0x00010203u : 0x03020100u;
    const uint32_t x0 = __builtin_amdgcn_perm(0u, minus ? a3 : a0, rev), x1 = __builtin_amdgcn_perm(0u, minus ? a2 : a1, rev),
                   x2 = __builtin_amdgcn_perm(0u, minus ? a1 : a2, rev), x3 = __builtin_amdgcn_perm(0u, minus ? a0 : a3, rev);
    const long long dst = minus ? D + m - cons - 16 * sub - 16 : D + cons + 16 * sub;
    // Byte map (complement on the '-' strand, 0 -> '-' on the read line) as ONE v_perm_b32 per dword: the low three bits
    // of the bytes a row normally holds are distinct (0 -> 0, 'A' -> 1, 'C' -> 3, 'T' -> 4, '-' -> 5, 'N' -> 6, 'G' -> 7), so
    // they select from an 8-byte table.  Mapping the result back through the inverse table returns the input exactly for
    // the bytes of that set and never for any other byte (IUPAC codes of the reference): those chunks take the SWAR path.
    const uint32_t f_lo = (minus ? 0x47025400u : 0x43024100u) | (dash_zero ? 0x2Du : 0u);  // [0] [1] [2] [3]
    const uint32_t f_hi = minus ? 0x434E2D41u : 0x474E2D54u;                              // [4] [5] [6] [7]
    const uint32_t b_lo = minus ? 0x47025400u : 0x43024100u;
    const uint32_t b_hi = (minus ? 0x434E0041u : 0x474E0054u) | (dash_zero ? 0u : 0x2D00u);
    uint32_t w0 = __builtin_amdgcn_perm(f_hi, f_lo, x0 & 0x07070707u), w1 = __builtin_amdgcn_perm(f_hi, f_lo, x1 & 0x07070707u),
             w2 = __builtin_amdgcn_perm(f_hi, f_lo, x2 & 0x07070707u), w3 = __builtin_amdgcn_perm(f_hi, f_lo, x3 & 0x07070707u);
    const uint32_t bad = (__builtin_amdgcn_perm(b_hi, b_lo, w0 & 0x07070707u) ^ x0) |
                         (__builtin_amdgcn_perm(b_hi, b_lo, w1 & 0x07070707u) ^ x1) |
                         (__builtin_amdgcn_perm(b_hi, b_lo, w2 & 0x07070707u) ^ x2) |
                         (__builtin_amdgcn_perm(b_hi, b_lo, w3 & 0x07070707u) ^ x3);
    if (__builtin_expect(bad != 0, 0)) {
      w0 = minus ? complement4(x0) : x0;
      w1 = minus ? complement4(x1) : x1;
      w2 = minus ? complement4(x2) : x2;
      w3 = minus ? complement4(x3) : x3;
      if (dash_zero) {
        w0 = dash_zero_bytes(w0);
        w1 = dash_zero_bytes(w1);
        w2 = dash_zero_bytes(w2);
        w3 = dash_zero_bytes(w3);
      }
    }
    *reinterpret_cast<uint4 *>(text + dst) = make_uint4(w0, w1, w2, w3);
  }
  if (avail == m) {  // the row ends in this tile: its last <= 15 columns
    const int r = (m - cons) & 15, c0 = m - r;
    if (sub < r) {
      uint32_t v = row8[16 + (c0 - 256 * t) + sub];
      if (minus) v = complement(v);
      if (dash_zero && v == 0) v = '-';
      text[minus ? D + m - 1 - (c0 + sub) : D + c0 + sub] = (char)v;
    }
  }
}

__global__ __launch_bounds__(256) void k_text_rows(TextArgs a, const DeviceFlags *flags) {
  __shared__ uint32_t s_tile[64 * kTileStride + 4];  // + the dwords an aligned alignbyte of the last row reads past it
  __shared__ __attribute__((aligned(16))) uint32_t s_out[16 * kOutStride];  // one squeezed row per (wave, group)
  __shared__ uint32_t s_sel[16];                     // v_perm selectors that move the kept bytes of a dword to its low end
  __shared__ uint32_t s_pend[64 * 4];                // squeezed bytes of a task that do not fill a 16-byte chunk yet
  __shared__ int s_q[64], s_m[64], s_task[64], s_done[64];
  __shared__ int s_minus[64];
  __shared__ long long s_dmaf[64], s_dsq[64];       // destination offsets of the task's MAF line / squeezed line
  const int64_t wave = blockIdx.x;
  const int pass = blockIdx.y;
  short_kernel_priority();
  if (wave * 64 >= flags->total_slots) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid < 64) {
    const int task = a.task_of_slot[wave * 64 + tid];
    int q = 0, m = 0, minus = 0, tk = -1;
    long long dmaf = 0, dsq = 0;
    if (task >= 0) {
      const int64_t r = task / a.pass_num;
      if (r < a.n_reads) {
        const int64_t readnum = a.first_read + r;
        minus = a.read_minus ? (a.read_minus[r] != 0) : ((readnum & 1) == 0);
        m = a.maf_len[task];
        q = a.out_len[task];
        tk = task;
        const int64_t *rd = a.row_dst + (int64_t)task * 6;
        dmaf = rd[(pass == 1) ? 2 : 3];
        // BAM: the bases are parked as ASCII in the pw array (k_bam_finish packs them to 4 bits)
        dsq = rd[(pass == 0) ? (a.bam ? 5 : 0) : 1];
      }
    }
    s_q[tid] = q;
    s_m[tid] = m;
    s_task[tid] = tk;
    s_done[tid] = 0;
    s_minus[tid] = minus;
    s_dmaf[tid] = dmaf;
    s_dsq[tid] = dsq;
  }
  s_pend[tid] = 0;
  if (tid < 16) {
    uint32_t sel = 0x0c0c0c0cu, n = 0;  // 0x0c selects a zero byte
    for (uint32_t b = 0; b < 4; b++)
      if ((tid >> b) & 1) {
        sel = (sel & ~(0xffu << (8 * n))) | (b << (8 * n));
        n++;
      }
    s_sel[tid] = sel;
  }
  __syncthreads();
  // extent of this wave's 16 tasks
  int mmax = 0;
  {
    const int j = wv * 16 + (lane & 15);
    int v = (s_task[j] >= 0) ? s_m[j] : 0;
#pragma unroll
    for (int d = 8; d > 0; d >>= 1) {
      const int t = __shfl_xor(v, d, 64);
      v = (t > v) ? t : v;
    }
    mmax = __shfl(v, 0, 64);
  }
  if (mmax == 0) return;
  const int cap_raw = __builtin_amdgcn_readfirstlane(a.wave_cap[wave]);  // wave-uniform: keeps row offsets scalar
  const int cap_dw = cap_raw & ~kWaveTransposed;
  const bool transposed = (cap_raw & kWaveTransposed) != 0;  // rows stored task by task (k_walk_errhmm_coop)
  const uint32_t *region =
      reinterpret_cast<const uint32_t *>(a.scratch + a.wave_off[wave] + (size_t)pass * cap_dw * 256);
  uint32_t *tile = s_tile + wv * 16 * kTileStride;
  const int lt = lane & 15, lc = lane >> 4;  // load role: task-in-wave, which quarter of the tile's dwords
  const int g = lane >> 4, sub = lane & 15;  // serve role: task group, 16-byte chunk
  uint32_t *outb = s_out + (wv * 4 + g) * kOutStride;
  uint8_t *orow = reinterpret_cast<uint8_t *>(outb);

  // lane (lt, lc) loads dword c + 16 lc of task lt (LDS banks 5 lt + c + 16 lc: all different); the loads of the NEXT
  // tile are issued before this tile is served, so their latency hides behind the four serving steps
  // (no bounds checks: columns past a row's end are never used, and the pool ends with kScratchSlack bytes of slack)
  uint32_t pre[16];
  const uint32_t *lane_src = transposed ? region + (size_t)(wv * 16 + lt) * cap_dw + 16 * lc : region + wv * 16 + lt + (size_t)(16 * lc) * 64;
  const size_t dw_step = transposed ? 1 : 64;  // dword d of the lane's task: interleaved 64 dwords apart, transposed adjacent
#pragma unroll
  for (int c = 0; c < 16; ++c) pre[c] = scratch_load(lane_src + (size_t)c * dw_step);
  for (int s0 = 0, t = 0; s0 < mmax; s0 += 256, ++t) {
    // carry: the previous tile's last 16 bytes move in front (lane = task row x dword)
    if (t > 0) tile[lt * kTileStride + lc] = tile[lt * kTileStride + 64 + lc];
    wave_lds_sync();
#pragma unroll
    for (int c = 0; c < 16; ++c) tile[lt * kTileStride + 4 + c + 16 * lc] = pre[c];
    wave_lds_sync();
    if (s0 + 256 < mmax) {
      const int c1 = (s0 + 256) >> 2;
#pragma unroll
      for (int c = 0; c < 16; ++c) pre[c] = scratch_load(lane_src + (size_t)(c1 + c) * dw_step);
    }
#pragma unroll 1
    for (int it = 0; it < 4; ++it) {
      const int i = it * 4 + g, j = wv * 16 + i;
      const int task = s_task[j];
      const int m = s_m[j], minus = s_minus[j];
      const bool act = task >= 0 && s0 < m;
      const uint32_t *trow = tile + i * kTileStride;
      if (pass == 1) {
        if (act) write_maf_tile(a.maf_text, s_dmaf[j], m, t, trow, minus, false, sub);
        continue;
      }
      if (pass == 0 && act) write_maf_tile(a.maf_text, s_dmaf[j], m, t, trow, minus, true, sub);
      // ---- squeeze the columns that carry a read base (non-zero byte): lane `sub` holds columns 16 sub .. 16 sub + 15
      // of the window; its output offset is the count of kept bytes in the lower lanes of its row of 16 (DPP scan).
      // The out row starts with the task's pending bytes, so that it is destination-chunk aligned.
      const int ncol = act ? ((m - s0 < 256) ? m - s0 : 256) : 0;
      const int dn = s_done[j];
      const long long D = s_dsq[j];
      const int p = (int)((D + dn) & 15);  // bytes of the destination chunk in front of this window's first byte
      uint32_t w[4];
      uint32_t keep = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        w[k] = trow[4 + 4 * sub + k];
        const uint32_t nz = ~eq_bytes(w[k], 0u) & 0x80808080u;          // 0x80 per non-zero byte
        keep |= ((((nz >> 7) * 0x00204081u) >> 21) & 15u) << (4 * k);    // -> 4 bits
      }
      const int nv = ncol - 16 * sub;  // valid columns of this lane
      keep &= (nv >= 16) ? 0xffffu : (nv > 0 ? (1u << nv) - 1u : 0u);
      const int cnt = __popc(keep);
      const int incl = row16_scan(cnt);
      const int total = __shfl(incl, (lane & 48) | 15, 64);
      // the out row: pending bytes in front, zeros behind; every lane ORs its kept bytes in, one dword of <= 4 at a time
      // (compacted by a v_perm_b32, shifted to its byte offset: two ds_or_b32, the second one usually of zero)
      reinterpret_cast<uint4 *>(outb)[1 + sub] = make_uint4(0u, 0u, 0u, 0u);
      if (sub < 2) reinterpret_cast<uint4 *>(outb)[17 + sub] = make_uint4(0u, 0u, 0u, 0u);
      if (sub < 4) outb[sub] = s_pend[j * 4 + sub];
      wave_lds_sync();
      {
        const int o = p + incl - cnt;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint32_t v = __builtin_amdgcn_perm(0u, w[k], s_sel[(keep >> (4 * k)) & 15u]);
          const int off = o + __popc(keep & ((1u << (4 * k)) - 1u));
          const unsigned long long sv = (unsigned long long)v << (8 * (off & 3));
          atomicOr(&outb[off >> 2], (uint32_t)sv);
          atomicOr(&outb[(off >> 2) + 1], (uint32_t)(sv >> 32));
        }
      }
      wave_lds_sync();
      const int have = p + total;              // row bytes: [0, p) belong to earlier windows (or to the text in front)
      const bool last = s0 + 256 >= m;
      const int n_ch = have >> 4;
      char *dst = a.read_text + (D + dn - p);  // 16-byte aligned
      const int foreign = (p > dn) ? p - dn : 0;  // leading bytes of chunk 0 that belong to the text in front of the row
      if (act && sub < n_ch) {
        const uint32_t *src = outb + 4 * sub;
        if (sub == 0 && foreign) {
#pragma unroll
          for (int k = 0; k < 4; k++) {
            if (4 * k >= foreign) {
              reinterpret_cast<uint32_t *>(dst)[k] = src[k];
            } else if (4 * k + 4 > foreign) {
              for (int b = foreign; b < 4 * k + 4; b++) dst[b] = (char)orow[b];
            }
          }
        } else {
          reinterpret_cast<uint4 *>(dst)[sub] = reinterpret_cast<const uint4 *>(outb)[sub];
        }
      }
      const int rem = have & 15;
      if (act && last) {
        const int lo = (n_ch == 0) ? foreign : 0;  // a row that ends inside its first chunk
        if (sub >= lo && sub < rem) dst[16 * n_ch + sub] = (char)orow[16 * n_ch + sub];
      }
      wave_lds_sync();
      if (act && sub < 4) s_pend[j * 4 + sub] = outb[4 * n_ch + sub];
      if (act && sub == 0) s_done[j] = dn + total;
      wave_lds_sync();
    }
  }
}

// Constant stretches of the read text, streamed with 16-byte stores (no turn through LDS needed):
//   ERRHMM quality line: out_len x '!' (pbsim.cpp:4007-4010);  SAM ip / pw tags: out_len x ",9" each
//   (pbsim.cpp:4019-4025);  BAM ip array: out_len bytes of value 9.  One wave per task.
__device__ __forceinline__ void fill_run(char *dst, int64_t n, uint32_t even_byte, uint32_t odd_byte, int lane) {
  // byte i of the run is even_byte for even i, odd_byte for odd i
  const int64_t head = min(n, (int64_t)((16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15));
  if (lane < head) dst[lane] = (char)((lane & 1) ? odd_byte : even_byte);
  const uint32_t lo = (head & 1) ? odd_byte : even_byte, hi = (head & 1) ? even_byte : odd_byte;
  const uint32_t w = lo | (hi << 8) | (lo << 16) | (hi << 24);
  const int64_t n16 = (n - head) >> 4;
  uint4 *d16 = reinterpret_cast<uint4 *>(dst + head);
  const uint4 v = make_uint4(w, w, w, w);
  for (int64_t i = lane; i < n16; i += 64) d16[i] = v;
  const int64_t done = head + (n16 << 4);
  if (lane < n - done) dst[done + lane] = (char)(((done + lane) & 1) ? odd_byte : even_byte);
}

__global__ __launch_bounds__(256) void k_text_fill(TextArgs a) {
  short_kernel_priority();
  const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (t >= a.n_reads * a.pass_num) return;
  const int64_t q = a.out_len[t];
  const int64_t *rd = a.row_dst + t * 6;
  if (!a.is_qs) fill_run(a.read_text + rd[1], q, '!', '!', lane);
  if (a.pass_num > 1) {
    if (a.bam) {
      fill_run(a.read_text + rd[4], q, 9, 9, lane);
    } else {
      fill_run(a.read_text + rd[4], 2 * q, ',', '9', lane);
      fill_run(a.read_text + rd[5], 2 * q, ',', '9', lane);
    }
  }
}

// BAM only, after the rows: one wave per task packs the ASCII bases parked in the pw array
// into 4-bit codes ("=ACMGRSVTWYHKDBN"), turns the qualities into phred values and fills pw with 9.
__device__ __forceinline__ uint32_t bam_base_code(uint32_t c) {
  switch (to_upper(c)) {  // htslib's seq_nt16_table is case-insensitive
    case '=': return 0;  case 'A': return 1;  case 'C': return 2;  case 'M': return 3;
    case 'G': return 4;  case 'R': return 5;  case 'S': return 6;  case 'V': return 7;
    case 'T': return 8;  case 'W': return 9;  case 'Y': return 10; case 'H': return 11;
    case 'K': return 12; case 'D': return 13; case 'B': return 14;
    default: return 15;
  }
}

// four ASCII bases (one dword, first base in the low byte) -> two BAM bytes (low 16 bits: first base in the high nibble).
// A / C / G / T in either case take the v_perm path: index (c >> 1) & 3 = 0 A, 1 C, 2 T, 3 G picks the code from one constant
// and the letter back from another (the check that the dword held nothing else); anything else (N, IUPAC codes) goes base by base.
__device__ __forceinline__ uint32_t bam_pack4(uint32_t w) {
  const uint32_t u = w & 0xDFDFDFDFu;                       // upper case (htslib's seq_nt16_table is case-insensitive)
  const uint32_t idx = (u >> 1) & 0x03030303u;
  uint32_t c = __builtin_amdgcn_perm(0u, 0x04080201u, idx);  // A 1, C 2, T 8, G 4
  if (__builtin_amdgcn_perm(0u, 0x47544341u, idx) != u)       // 'A' 'C' 'T' 'G'
    c = bam_base_code(w & 255u) | (bam_base_code((w >> 8) & 255u) << 8) | (bam_base_code((w >> 16) & 255u) << 16) |
        (bam_base_code(w >> 24) << 24);
  const uint32_t x = ((c & 0x000F000Fu) << 4) | ((c >> 8) & 0x000F000Fu);  // bytes 0 and 2 hold the two packed bytes
  return (x & 0xFFu) | ((x >> 8) & 0xFF00u);
}

// One wave per task.  Everything moves as 16-byte vectors at the DESTINATION's alignment (a record's fields sit at byte
// offsets): the ASCII bases are read as aligned dwords and shifted into place (v_alignbyte), qualities and the pw fill are in
// place.  (The first version moved single bytes through a 16-way switch per base: as much GPU time as k_text_rows itself --
// 2.7 s of the 4.7 s configs[2] job, profiles/r03_qshmm_timeline.txt.)
__global__ __launch_bounds__(256) void k_bam_finish(TextArgs a) {
  short_kernel_priority();
  const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (t >= a.n_reads * a.pass_num) return;
  const int q = a.out_len[t];
  const int64_t *rd = a.row_dst + t * 6;
  uint8_t *seq = reinterpret_cast<uint8_t *>(a.read_text) + rd[0];
  uint8_t *qual = reinterpret_cast<uint8_t *>(a.read_text) + rd[1];
  uint8_t *pw = reinterpret_cast<uint8_t *>(a.read_text) + rd[5];
  // ---- bases: q ASCII bytes at pw -> (q + 1) / 2 packed bytes at seq; whole 16-byte vectors cover full byte pairs only
  {
    const int nfull = q >> 1;  // packed bytes that take two bases
    const int head = min(nfull, (int)((16 - (reinterpret_cast<uintptr_t>(seq) & 15)) & 15));
    if (lane < head) seq[lane] = (uint8_t)((bam_base_code(pw[2 * lane]) << 4) | bam_base_code(pw[2 * lane + 1]));
    const int nvec = (nfull - head) >> 4;
    uint4 *d16 = reinterpret_cast<uint4 *>(seq + head);
    for (int v = lane; v < nvec; v += 64) {
      const uint8_t *src = pw + 2 * (head + 16 * v);  // 32 bases
      const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 3);
      const uint32_t *p = reinterpret_cast<const uint32_t *>(src - sh);
      uint32_t in[9];
#pragma unroll
      for (int k = 0; k < 9; k++) in[k] = p[k];   // (k == 8 is read for sh != 0 only in effect; the text buffer has slack)
      uint32_t o[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t w0 = __builtin_amdgcn_alignbyte(in[2 * k + 1], in[2 * k], sh);
        const uint32_t w1 = __builtin_amdgcn_alignbyte(in[2 * k + 2], in[2 * k + 1], sh);
        o[k] = bam_pack4(w0) | (bam_pack4(w1) << 16);
      }
      d16[v] = make_uint4(o[0], o[1], o[2], o[3]);
    }
    const int done = head + 16 * nvec;
    for (int i = done + lane; i < (q + 1) / 2; i += 64) {
      const uint32_t hi = bam_base_code(pw[2 * i]);
      const uint32_t lo = (2 * i + 1 < q) ? bam_base_code(pw[2 * i + 1]) : 0u;
      seq[i] = (uint8_t)((hi << 4) | lo);
    }
  }
  // ---- qualities: ASCII -> phred, in place (every byte is >= 33: the subtraction never borrows across bytes)
  {
    const int head = min(q, (int)((16 - (reinterpret_cast<uintptr_t>(qual) & 15)) & 15));
    if (lane < head) qual[lane] = (uint8_t)(qual[lane] - 33u);
    const int nvec = (q - head) >> 4;
    uint4 *d16 = reinterpret_cast<uint4 *>(qual + head);
    for (int v = lane; v < nvec; v += 64) {
      uint4 x = d16[v];
      x.x -= 0x21212121u;
      x.y -= 0x21212121u;
      x.z -= 0x21212121u;
      x.w -= 0x21212121u;
      d16[v] = x;
    }
    const int done = head + 16 * nvec;
    if (lane < q - done) qual[done + lane] = (uint8_t)(qual[done + lane] - 33u);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  fill_run(reinterpret_cast<char *>(pw), q, 9, 9, lane);  // every lane has read its bases: the pw array takes its values
}

}  // namespace

// ---------------------------------------------------------------------------
// launches
// ---------------------------------------------------------------------------
static inline unsigned blocks_for(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

void launch_prepare_reference(uint8_t *seq, uint8_t *hp, int flag_hp11, int64_t len, int64_t *tile_first, int64_t *tile_last,
                              int64_t *carry_start, int64_t *carry_next, int keep_first_case, DeviceFlags *flags,
                              hipStream_t s) {
  const int64_t n_tiles = (len + kHpTile - 1) / kHpTile;
  if (n_tiles == 0) return;
  hipLaunchKernelGGL(k_hp_breaks, dim3((unsigned)n_tiles), dim3(256), 0, s, seq, len, keep_first_case, tile_first,
                     tile_last, flags);
  hipLaunchKernelGGL(k_hp_carry, dim3(1), dim3(1024), 0, s, tile_first, tile_last, n_tiles, len, carry_start,
                     carry_next);
  hipLaunchKernelGGL(k_hp_final, dim3((unsigned)n_tiles), dim3(256), 0, s, seq, hp, flag_hp11, len, keep_first_case,
                     carry_start, carry_next, flags);
}

void launch_header_wgs(const HeaderArgs &a, hipStream_t s) {
  if (a.n_reads <= 0) return;
  hipLaunchKernelGGL(k_header_wgs, dim3(blocks_for(a.n_reads, 256)), dim3(256), 0, s, a);
}

void launch_squeeze_lines(const uint8_t *lines, int64_t bytes, uint8_t *dst, int64_t *tile_tmp, int64_t *scan_tmp, int64_t *total,
                          hipStream_t s) {
  const int64_t n_tiles = (bytes + kLineTile - 1) / kLineTile;
  if (n_tiles <= 0) return;
  hipLaunchKernelGGL(k_lines_count, dim3((unsigned)n_tiles), dim3(256), 0, s, lines, bytes, tile_tmp);
  launch_exclusive_scan_i64(tile_tmp, tile_tmp, n_tiles, scan_tmp, total, s);
  hipLaunchKernelGGL(k_lines_squeeze, dim3((unsigned)n_tiles), dim3(256), 0, s, lines, bytes, tile_tmp, dst);
}

void launch_chain_init(ChainState *chain, int64_t remaining, hipStream_t s) {
  hipLaunchKernelGGL(k_chain_init, dim3(1), dim3(1), 0, s, chain, remaining);
}
void launch_chain_prepare(const HeaderArgs &a, int k, int32_t pass_num, const int32_t *task_of_slot, int32_t *masked,
                          int64_t n_slots_max, ChainState *chain, hipStream_t s) {
  hipLaunchKernelGGL(k_chain_prepare, dim3(blocks_for(n_slots_max, 256)), dim3(256), 0, s, a, k, pass_num, task_of_slot, masked,
                     n_slots_max, chain);
}
void launch_chain_update(int k, int32_t pass_num, const int32_t *out_len, ChainState *chain, hipStream_t s) {
  hipLaunchKernelGGL(k_chain_update, dim3(1), dim3(1), 0, s, k, pass_num, out_len, chain);
}

void launch_header_trans(const HeaderArgs &a, hipStream_t s) {
  if (a.n_reads <= 0) return;
  hipLaunchKernelGGL(k_header_trans, dim3(blocks_for(a.n_reads, 256)), dim3(256), 0, s, a);
}

void launch_task_sort(const SortArgs &a, hipStream_t s) {
  const size_t nbins = (size_t)a.ncls * kLenBuckets;
  (void)hipMemsetAsync(a.hist, 0, nbins * kBinPad * sizeof(int32_t), s);
  (void)hipMemsetAsync(a.bin_cursor, 0, nbins * kBinPad * sizeof(int32_t), s);
  (void)hipMemsetAsync(a.task_of_slot, 0xff, (size_t)a.n_slots_max * sizeof(int32_t), s);
  hipLaunchKernelGGL(k_sort_hist, dim3(blocks_for(a.n_reads, 256)), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_sort_scan, dim3(1), dim3(kScanBlock), 0, s, a);
  hipLaunchKernelGGL(k_sort_scatter, dim3(blocks_for(a.n_reads, 256)), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_wave_cap, dim3(blocks_for(a.n_slots_max / 64, 256)), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_wave_scan, dim3(1), dim3(kScanBlock), 0, s, a);
  const int64_t n_wg = a.n_slots_max / kWG;
  (void)hipMemsetAsync(a.wg_hist, 0, (kLenBuckets + 1) * sizeof(int32_t), s);
  hipLaunchKernelGGL(k_wg_hist, dim3(blocks_for(n_wg, 256)), dim3(256), 0, s, a, a.wg_hist);
  hipLaunchKernelGGL(k_wg_scan, dim3(1), dim3(kScanBlock), 0, s, a.wg_hist, a.wg_start);
  hipLaunchKernelGGL(k_wg_scatter, dim3(blocks_for(n_wg, 256)), dim3(256), 0, s, a, a.wg_start, a.wg_order);
}

// A walk workgroup lives for milliseconds (until its longest read ends) and six of them fill a CU's registers and LDS: the
// short kernels and copies of the other slots -- text emission, scans, the flag reads the host waits for -- then queue until
// one of them ends (measured: a 200-byte device-to-host read took up to 39 ms beside three walks).  Asking for more LDS per
// walk workgroup than it needs caps the workgroups per CU (160 KB): `min_lds_kb` 27 -> five per CU (neutral for the walk,
// same-box A/B), 41 -> three per CU (the walk alone 6 % slower, but the job pipeline, whose round loop waits on many short
// kernels, 8 % faster; the two-giant-batches steady state 11 % slower -- so the caller chooses).  PBSIM_WALK_LDS_KB overrides.
static uint32_t walk_lds(uint32_t lds_bytes, int min_lds_kb) {
  static const int env_kb = getenv("PBSIM_WALK_LDS_KB") ? atoi(getenv("PBSIM_WALK_LDS_KB")) : -1;
  const uint32_t pad = (uint32_t)(env_kb >= 0 ? env_kb : min_lds_kb) << 10;
  return lds_bytes > pad ? lds_bytes : pad;
}

void launch_walk_errhmm(const WalkArgs &a, int64_t n_slots_max, uint32_t lds_bytes, bool fast_rv, bool hp_bits,
                        hipStream_t s, int min_lds_kb) {
  lds_bytes = walk_lds(lds_bytes, min_lds_kb);
  const dim3 grid((unsigned)(n_slots_max / kWG)), block(kWG);
  if (fast_rv && hp_bits) hipLaunchKernelGGL((k_walk_errhmm<true, true>), grid, block, lds_bytes, s, a);
  else if (fast_rv) hipLaunchKernelGGL((k_walk_errhmm<true, false>), grid, block, lds_bytes, s, a);
  else if (hp_bits) hipLaunchKernelGGL((k_walk_errhmm<false, true>), grid, block, lds_bytes, s, a);
  else hipLaunchKernelGGL((k_walk_errhmm<false, false>), grid, block, lds_bytes, s, a);
}

// workgroups of k_walk_errhmm_coop a CU holds at once (registers: five waves per SIMD; LDS: the class tables + kCoopWaveLds
// per wave); 0 if the runtime cannot say
int walk_errhmm_coop_resident(uint32_t lds_bytes, bool hp_bits) {
  int n = 0;
  const size_t lds = lds_bytes + kCoopWaves * kCoopWaveLds;
  const hipError_t e = hp_bits ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_walk_errhmm_coop<true>, kCoopWaves * 64, lds)
                               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_walk_errhmm_coop<false>, kCoopWaves * 64, lds);
  return e == hipSuccess ? n : 0;
}

void launch_walk_errhmm_coop(const WalkArgs &a, int n_wg, uint32_t lds_bytes, bool hp_bits, hipStream_t s) {
  const dim3 grid((unsigned)n_wg), block(kCoopWaves * 64);
  lds_bytes += kCoopWaves * kCoopWaveLds;
  if (hp_bits) hipLaunchKernelGGL((k_walk_errhmm_coop<true>), grid, block, lds_bytes, s, a);
  else hipLaunchKernelGGL((k_walk_errhmm_coop<false>), grid, block, lds_bytes, s, a);
}

void launch_walk_qshmm_coop(const WalkArgs &a, int n_wg, uint32_t lds_bytes, hipStream_t s) {
  const dim3 grid((unsigned)n_wg), block(kWG);
  lds_bytes += 94 * 32 + (kWG / 64) * kQCoopWaveLds;
  hipLaunchKernelGGL(k_walk_qshmm_coop, grid, block, lds_bytes, s, a);
}

void launch_qshmm_coop_qsum(const WalkArgs &a, int64_t n_slots_max, hipStream_t s) {
  hipLaunchKernelGGL(k_qshmm_coop_qsum, dim3((unsigned)((n_slots_max + 255) / 256)), dim3(256), 0, s, a, (int)n_slots_max);
}

void launch_walk_qshmm(const WalkArgs &a, int64_t n_slots_max, uint32_t lds_bytes, bool fast_rv, bool hp_bits,
                       hipStream_t s, int min_lds_kb) {
  lds_bytes = walk_lds(lds_bytes, min_lds_kb);
  const dim3 grid((unsigned)(n_slots_max / kWG)), block(kWG);
  if (fast_rv && hp_bits) hipLaunchKernelGGL((k_walk_qshmm<true, true>), grid, block, lds_bytes, s, a);
  else if (fast_rv) hipLaunchKernelGGL((k_walk_qshmm<true, false>), grid, block, lds_bytes, s, a);
  else if (hp_bits) hipLaunchKernelGGL((k_walk_qshmm<false, true>), grid, block, lds_bytes, s, a);
  else hipLaunchKernelGGL((k_walk_qshmm<false, false>), grid, block, lds_bytes, s, a);
}

void launch_walk_sample(const SampleArgs &a, bool hp_bits, hipStream_t s) {
  if (a.n_line_waves <= 0) return;
  const int lane_waves = a.n_line_waves - a.n_coop_waves;
  const dim3 grid((unsigned)(a.n_coop_blocks + (lane_waves + kWG / 64 - 1) / (kWG / 64))), block(kWG);
  if (hp_bits) hipLaunchKernelGGL((k_walk_sample<true>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((k_walk_sample<false>), grid, block, 0, s, a);
}

void launch_sample_qsum(const SampleArgs &a, hipStream_t s) {
  if (a.n_coop_slots <= 0) return;
  hipLaunchKernelGGL(k_sample_qsum, dim3((unsigned)((a.n_coop_slots + 255) / 256)), dim3(256), 0, s, a, a.n_coop_slots);
}

void launch_exclusive_scan_i64(const int64_t *in, int64_t *out, int64_t n, int64_t *tmp, int64_t *total,
                               hipStream_t s) {
  if (n <= 0) {
    if (total) (void)hipMemsetAsync(total, 0, sizeof(int64_t), s);
    return;
  }
  const unsigned nb = blocks_for(n, kScanTile);
  hipLaunchKernelGGL(k_scan_sums, dim3(nb), dim3(256), 0, s, in, n, tmp);
  hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(kScanBlock), 0, s, tmp, (int64_t)nb, total);
  hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(256), 0, s, in, out, n, (const int64_t *)tmp);
}

void launch_quota_cut(const int64_t *cum, const int32_t *rawlen, int64_t n_reads, int64_t len_total_before,
                      int64_t quota, int force_all, DeviceFlags *flags, hipStream_t s) {
  hipLaunchKernelGGL(k_quota_init, dim3(1), dim3(1), 0, s, flags, n_reads);
  if (!force_all && n_reads > 0)
    hipLaunchKernelGGL(k_quota_find, dim3(blocks_for(n_reads, 256)), dim3(256), 0, s, cum, rawlen, n_reads,
                       len_total_before, quota, flags);
}

void launch_gather_pass0_scan(const int32_t *out_len, int64_t n_reads, int32_t pass_num, int64_t *cum, int64_t *tmp,
                              int64_t *total, hipStream_t s) {
  if (n_reads > 0)
    hipLaunchKernelGGL(k_gather_pass0, dim3(blocks_for(n_reads, 256)), dim3(256), 0, s, out_len, n_reads, pass_num,
                       cum);
  launch_exclusive_scan_i64(cum, cum, n_reads, tmp, total, s);
}

void launch_text_sizes(const TextArgs &a, DeviceFlags *flags, hipStream_t s) {
  const int64_t n_tasks = a.n_reads * a.pass_num;
  if (n_tasks <= 0) return;
  hipLaunchKernelGGL(k_text_sizes, dim3(blocks_for(n_tasks, 256)), dim3(256), 0, s, a, flags);
}

void launch_text_emit(const TextArgs &a, int64_t n_slots_max, const DeviceFlags *flags, hipStream_t s) {
  const int64_t n_tasks = a.n_reads * a.pass_num;
  if (n_tasks <= 0) return;
  hipLaunchKernelGGL(k_text_headers, dim3(blocks_for(n_tasks, 256)), dim3(256), 0, s, a);
  if (!a.is_qs || a.pass_num > 1) hipLaunchKernelGGL(k_text_fill, dim3(blocks_for(n_tasks, 4)), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_text_rows, dim3((unsigned)(n_slots_max / 64), a.is_qs ? 3 : 2), dim3(256), 0, s, a, flags);
  if (a.bam) hipLaunchKernelGGL(k_bam_finish, dim3(blocks_for(n_tasks, 4)), dim3(256), 0, s, a);
}

}  // namespace pbsim
