// deflate.hip -- DEFLATE on the GPU for the text the emit kernels leave in HBM.
//
// The reference pipes every output file through a `gzip -c` child (popen,
// pbsim.cpp:708-730) and its BAM through `samtools view -b` (pbsim.cpp:715-719);
// SURVEY 8(f)1 names that the true end-to-end bottleneck.  Here the FASTQ / SAM /
// BAM / MAF bytes are compressed where they already are: one workgroup turns one
// 32 KiB chunk of text into one BGZF-framed gzip member (RFC 1952 with the 'BC'
// extra field of SAMv1 4.1, so the same stream is a valid .gz and a valid BAM
// container) holding one dynamic-Huffman block (RFC 1951 3.2.7).  Only the
// compressed bytes cross PCIe.  tests/deflate_model.py is the executable
// specification; the GPU output is compared with it byte for byte.
//
//   tokens  : each thread owns 128 consecutive bytes; a byte equal to its
//             predecessor extends a run, runs >= 3 become one distance-1 match
//             (that is what removes the constant '!' quality line and the SAM
//             ",9" tags), everything else is a literal.  The text has no other
//             redundancy an LZ window would find (random bases).
//   Huffman : LDS histogram -> rank sort -> two-queue merge -> depth limit 15 ->
//             canonical codes; the code-length alphabet uses a fixed complete code
//   layout  : the chunk sits in LDS padded by one dword per 32 so that the 64
//             lanes' sequential walks over their segments hit 64 different banks;
//             the member is assembled in LDS by ds_or on a zeroed buffer and leaves
//             as full dwords
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "kernels.h"

namespace pbsim {

namespace {

constexpr int kChunk = DF_CHUNK;          // input bytes per member
constexpr int kThreads = 256;
constexpr int kSeg = kChunk / kThreads;   // 128 bytes per thread
constexpr int kSegDw = kSeg / 4;          // 32 dwords
constexpr int kInDw = kChunk / 4 + kChunk / 128;
constexpr int kOutBytes = 33024;          // member capacity of the LDS image (it is reused as the output buffer)
constexpr int kOutDw = kOutBytes / 4;
constexpr int kSyms = 288;
constexpr uint32_t kPoly = 0xEDB88320u;
constexpr int kHeadBits = 18 * 8;         // BGZF header bytes in front of the deflate payload
constexpr int kPrefixBits = 3 + 5 + 5 + 4 + 19 * 3;

// fixed code for the code-length alphabet (tests/deflate_model.py CL_LEN): 0,17,18 -> 3 bits; 2,3,4,12 -> 4; rest 5.
// Everything derived from it is a compile-time constant: no table in memory on the kernel's critical path.
constexpr int kClLenTab[19] = {3, 5, 4, 4, 4, 5, 5, 5, 5, 5, 5, 5, 4, 5, 5, 5, 5, 3, 3};
constexpr int kClOrderTab[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
constexpr uint64_t pack_cl_len() {
  uint64_t v = 0;
  for (int s = 0; s < 19; ++s) v |= (uint64_t)kClLenTab[s] << (3 * s);
  return v;
}
constexpr uint64_t kClLenPacked = pack_cl_len();   // 3 bits per symbol
__device__ __forceinline__ uint32_t cl_len(uint32_t s) { return (uint32_t)(kClLenPacked >> (3 * s)) & 7u; }
// block prefix after the 18 BGZF bytes: BFINAL=1, BTYPE=2, HLIT-257 (5 bits, added at run time), HDIST-1 = 0,
// HCLEN-4 = 15, then the 19 code-length code lengths in the order of RFC 1951 3.2.7: 74 bits
constexpr unsigned __int128 pack_prefix() {
  unsigned __int128 v = 1u | (2u << 1) | ((unsigned __int128)15 << 13);
  for (int i = 0; i < 19; ++i) v |= (unsigned __int128)kClLenTab[kClOrderTab[i]] << (17 + 3 * i);
  return v;
}
constexpr uint64_t kPrefixLo = (uint64_t)pack_prefix();
constexpr uint32_t kPrefixHi = (uint32_t)(pack_prefix() >> 64);

__device__ __forceinline__ uint32_t rev_bits(uint32_t code, uint32_t len) { return __brev(code) >> (32 - len); }

// canonical code of code-length symbol s, bit-reversed for the LSB-first stream
__device__ __forceinline__ uint32_t cl_code(uint32_t s) {
  // per length: first canonical code and the symbols in index order
  // len 3: {0,17,18} -> 0,1,2 ; len 4: {2,3,4,12} -> 6,7,8,9 ; len 5: {1,5,6,7,8,9,10,11,13,14,15,16} -> 20..31
  uint32_t c, l = cl_len(s);
  if (l == 3) c = s == 0 ? 0 : s - 16;
  else if (l == 4) c = s == 12 ? 9 : 4 + s;
  else c = s == 1 ? 20 : s <= 11 ? 16 + s : 15 + s;
  return rev_bits(c, l);
}

__device__ __forceinline__ uint32_t gf2_mul(uint32_t a, uint32_t b) {  // a*b mod P, reflected (bit 31 = x^0)
  uint32_t p = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    p ^= b & (0u - ((a >> (31 - i)) & 1u));
    b = (b >> 1) ^ (kPoly & (0u - (b & 1u)));
  }
  return p;
}

struct BitWriter {
  uint32_t *out;
  uint32_t word, nb;
  uint64_t acc;
  __device__ __forceinline__ void init(uint32_t *o, uint32_t bitpos) {
    out = o;
    word = bitpos >> 5;
    nb = bitpos & 31;
    acc = 0;
  }
  __device__ __forceinline__ void put(uint32_t v, uint32_t k) {  // k <= 32
    acc |= (uint64_t)v << nb;
    nb += k;
    if (nb >= 32) {
      atomicOr(&out[word++], (uint32_t)acc);
      acc >>= 32;
      nb -= 32;
    }
  }
  __device__ __forceinline__ void put64(uint64_t v, uint32_t k) {
    if (k > 32) {
      put((uint32_t)v, 32);
      put((uint32_t)(v >> 32), k - 32);
    } else {
      put((uint32_t)v, k);
    }
  }
  __device__ __forceinline__ void flush() {
    if (nb) atomicOr(&out[word], (uint32_t)acc);
  }
};

// length 3..258 -> (symbol, extra bit count, extra value)
__device__ __forceinline__ void length_symbol(uint32_t L, uint32_t *sym, uint32_t *eb, uint32_t *ev) {
  const uint32_t v = L - 3;
  if (v < 8) {
    *sym = 257 + v;
    *eb = 0;
    *ev = 0;
  } else if (L == 258) {
    *sym = 285;
    *eb = 0;
    *ev = 0;
  } else {
    const uint32_t e = (31 - __clz(v)) - 2;
    *sym = 261 + 4 * e + ((v >> e) & 3);
    *eb = e;
    *ev = v & ((1u << e) - 1);
  }
}

// exclusive scan of one u32 per thread over the 256-thread block; *total = sum
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t v, uint32_t *s_wave, uint32_t *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  __syncthreads();  // s_wave may still be read from a previous scan
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kThreads / 64; ++w) {
    const uint32_t x = s_wave[w];
    if (w < wave) base += x;
    tot += x;
  }
  *total = tot;
  return base + inc - v;
}

// 128 positions of one thread's segment as bit masks
struct Mask128 {
  uint64_t lo, hi;
};
__device__ __forceinline__ Mask128 m_and(Mask128 a, Mask128 b) { return {a.lo & b.lo, a.hi & b.hi}; }
__device__ __forceinline__ Mask128 m_or(Mask128 a, Mask128 b) { return {a.lo | b.lo, a.hi | b.hi}; }
__device__ __forceinline__ Mask128 m_not(Mask128 a) { return {~a.lo, ~a.hi}; }
template <int K>
__device__ __forceinline__ Mask128 m_shl(Mask128 a) { return {a.lo << K, (a.hi << K) | (a.lo >> (64 - K))}; }
template <int K>
__device__ __forceinline__ Mask128 m_shr(Mask128 a) { return {(a.lo >> K) | (a.hi << (64 - K)), a.hi >> K}; }
__device__ __forceinline__ uint32_t m_nibble(Mask128 a, int j) {  // bits 4j..4j+3
  return (uint32_t)((j < 16 ? a.lo >> (4 * j) : a.hi >> (4 * j - 64)) & 15u);
}
__device__ __forceinline__ int m_first(Mask128 a) { return a.lo ? __builtin_ctzll(a.lo) : 64 + __builtin_ctzll(a.hi); }
__device__ __forceinline__ Mask128 m_clear(Mask128 a, int pos) {
  if (pos < 64) a.lo &= ~(1ull << pos);
  else a.hi &= ~(1ull << (pos - 64));
  return a;
}
// number of consecutive set bits of m starting at pos (m has bit pos set)
__device__ __forceinline__ int m_run(Mask128 m, int pos) {
  uint64_t lo, hi;
  if (pos == 0) {
    lo = m.lo;
    hi = m.hi;
  } else if (pos < 64) {
    lo = (m.lo >> pos) | (m.hi << (64 - pos));
    hi = m.hi >> pos;
  } else {
    lo = m.hi >> (pos - 64);
    hi = 0;
  }
  if (~lo) return __builtin_ctzll(~lo);
  return ~hi ? 64 + __builtin_ctzll(~hi) : 128;
}

// Token structure of a segment (tests/deflate_model.py tokenize): a position whose byte equals its predecessor is a
// run position; maximal groups of >= 3 run positions become one distance-1 match, every other position is a literal.
struct SegTokens {
  Mask128 lit;    // literal positions
  Mask128 start;  // first position of each match
  Mask128 cover;  // positions covered by matches
};

__device__ __forceinline__ SegTokens seg_tokens(const uint32_t (&seg)[kSegDw], int seg_n, uint32_t prev_dword,
                                                bool chunk_start) {
  uint64_t e[2] = {0, 0};
  uint32_t pw = prev_dword;
#pragma unroll
  for (int j = 0; j < kSegDw; ++j) {
    const uint32_t w = seg[j];
    const uint32_t d = w ^ ((w << 8) | (pw >> 24));
    const uint32_t z = ~(((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d | 0x7F7F7F7Fu);  // 0x80 in every byte equal to its predecessor
    const uint32_t nib = (((z >> 7) * 0x00204081u) >> 21) & 15u;
    e[j >> 4] |= (uint64_t)nib << (4 * (j & 15));
    pw = w;
  }
  Mask128 valid;
  valid.lo = seg_n >= 64 ? ~0ull : (1ull << seg_n) - 1;
  valid.hi = seg_n >= 128 ? ~0ull : seg_n > 64 ? (1ull << (seg_n - 64)) - 1 : 0ull;
  Mask128 eq = {e[0] & valid.lo, e[1] & valid.hi};
  if (chunk_start) eq.lo &= ~1ull;
  const Mask128 r3 = m_and(eq, m_and(m_shr<1>(eq), m_shr<2>(eq)));
  SegTokens t;
  t.cover = m_or(r3, m_or(m_shl<1>(r3), m_shl<2>(r3)));
  t.start = m_and(t.cover, m_not(m_shl<1>(t.cover)));
  t.lit = m_and(valid, m_not(t.cover));
  return t;
}

// ---------------------------------------------------------------------------------------------------------------------
// The code table of a stream.  The text of one deflate call (a batch's FASTQ, its MAF, its BAM records) is statistically
// the same from its first chunk to its last, so the Huffman code is fitted ONCE per call -- to the token histogram of the
// call's first kSampleChunks chunks, every symbol floored at one occurrence so that any later chunk stays encodable -- and
// every member carries the same (precomputed) block header.  Per chunk that leaves tokenising, the CRC, one table lookup
// per literal for the size, one for the bits: no histogram atomics, no sort, no serial tree construction.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kSampleChunks = DF_SAMPLE_CHUNKS;
constexpr int kHdrDw = DF_TABLE_HDR_DW;   // header image: member dwords 4.. (BSIZE field left zero, prefix, code lengths)

struct DfTable {             // device memory, DF_TABLE_BYTES
  uint32_t code[kSyms];      // reversed code | length << 16
  uint32_t hdr_end_bits;     // bit position in the member image where the tokens start (144 + prefix + header symbols)
  uint32_t pad[3];
  uint32_t hdr[kHdrDw];      // image dwords 4 .. 4 + kHdrDw - 1
};
static_assert(sizeof(DfTable) <= DF_TABLE_BYTES, "DF_TABLE_BYTES too small");

// the thread's 128 bytes of chunk `chunk` in registers + its token masks; also leaves the padded image in s_in
__device__ __forceinline__ void load_chunk(const uint8_t *__restrict__ text, int64_t n_bytes, int64_t chunk, uint32_t *s_in,
                                           uint32_t (&seg)[kSegDw], int *n_out, int *seg_n_out, uint32_t *prev_dw) {
  const int tid = threadIdx.x;
  const int64_t base = chunk * (int64_t)kChunk;
  const int n = (int)((n_bytes - base) < (int64_t)kChunk ? (n_bytes - base) : (int64_t)kChunk);
  const uint4 *src4 = reinterpret_cast<const uint4 *>(text + base);
  for (int q = tid; q < kChunk / 16; q += kThreads) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (q * 16 < n) v = src4[q];   // the text buffers carry 16 bytes of slack past their end
    const int d = q * 4;
    uint32_t *dst = &s_in[d + (d >> 5)];
    dst[0] = v.x;
    dst[1] = v.y;
    dst[2] = v.z;
    dst[3] = v.w;
  }
  __syncthreads();
  const int beg = tid * kSeg;
  *n_out = n;
  *seg_n_out = n - beg < 0 ? 0 : (n - beg < kSeg ? n - beg : kSeg);
#pragma unroll
  for (int j = 0; j < kSegDw; ++j) seg[j] = s_in[tid * (kSegDw + 1) + j];
  *prev_dw = tid ? s_in[tid * (kSegDw + 1) - 2] : 0u;
  __syncthreads();
}

// token histogram of the sample: hist[0..285] += literals / length symbols, hist[256] += 1 per chunk
__global__ __launch_bounds__(kThreads) void k_deflate_hist(const uint8_t *__restrict__ text, int64_t n_bytes,
                                                            uint32_t *__restrict__ hist) {
  __shared__ uint32_t s_in[kInDw];
  __shared__ uint32_t s_hist[3 * kSyms + 64];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 3 * kSyms + 64; i += kThreads) s_hist[i] = 0;
  uint32_t seg[kSegDw], prev_dw;
  int n, seg_n;
  load_chunk(text, n_bytes, blockIdx.x, s_in, seg, &n, &seg_n, &prev_dw);
  const SegTokens tk = seg_tokens(seg, seg_n, prev_dw, tid == 0);
  const uint32_t copy = (uint32_t)(lane % 3) * kSyms, dummy = 3 * kSyms + lane;
#pragma unroll
  for (int j = 0; j < kSegDw; ++j) {
    const uint32_t nl = m_nibble(tk.lit, j);
    if (nl == 0) continue;
    const uint32_t w = seg[j];
#pragma unroll
    for (int k = 0; k < 4; ++k) atomicAdd(&s_hist[((nl >> k) & 1u) ? copy + ((w >> (8 * k)) & 255u) : dummy], 1u);
  }
  Mask128 st = tk.start;
  while (st.lo | st.hi) {
    const int pos = m_first(st);
    st = m_clear(st, pos);
    uint32_t sy, eb, ev;
    length_symbol((uint32_t)m_run(tk.cover, pos), &sy, &eb, &ev);
    atomicAdd(&s_hist[copy + sy], 1u);
  }
  __syncthreads();
  for (int i = tid; i < 286; i += kThreads) {
    const uint32_t f = s_hist[i] + s_hist[kSyms + i] + s_hist[2 * kSyms + i] + (i == 256 ? 1u : 0u);
    if (f) atomicAdd(&hist[i], f);
  }
}

// one workgroup: histogram (+1 per symbol) -> length-limited Huffman code -> canonical codes -> the block header bits
__global__ __launch_bounds__(kThreads) void k_deflate_build(const uint32_t *__restrict__ hist, DfTable *__restrict__ tbl) {
  __shared__ uint32_t s_freq[kSyms], s_sfreq[kSyms], s_w[kSyms], s_code[kSyms];
  __shared__ uint16_t s_sorted[kSyms], s_par[kSyms], s_leafpar[kSyms], s_dep[kSyms];
  __shared__ uint8_t s_len[kSyms];
  __shared__ uint32_t s_cnt[16], s_first[16];
  __shared__ uint32_t s_img[4 + kHdrDw];
  const int tid = threadIdx.x;
  constexpr int m = 286;  // every symbol is used (floor of one occurrence)
  for (int i = tid; i < kSyms; i += kThreads) {
    s_freq[i] = i < m ? hist[i] + 1u : 0u;
    s_code[i] = 0;
    s_len[i] = 0;
  }
  for (int i = tid; i < 4 + kHdrDw; i += kThreads) s_img[i] = 0;
  if (tid < 16) s_cnt[tid] = 0;
  __syncthreads();
  for (int i = tid; i < m; i += kThreads) {  // rank sort by (freq, symbol)
    const uint32_t fi = s_freq[i];
    uint32_t rank = 0;
    for (int u = 0; u < m; ++u) {
      const uint32_t g = s_freq[u];
      rank += (g < fi) | ((g == fi) & (u < i));
    }
    s_sorted[rank] = (uint16_t)i;
    s_sfreq[rank] = fi;
  }
  __syncthreads();
  if (tid == 0) {
    // two-queue merge on the sorted leaves, depths limited to 15 (tests/deflate_model.py huffman_lengths)
    int leaf = 0, root = 0;
    for (int nxt = 0; nxt < m - 1; ++nxt) {
      uint32_t tot = 0;
      for (int k = 0; k < 2; ++k) {
        const bool take_leaf = leaf < m && (root >= nxt || s_sfreq[leaf] <= s_w[root]);
        if (take_leaf) {
          tot += s_sfreq[leaf];
          s_leafpar[leaf] = (uint16_t)nxt;
          ++leaf;
        } else {
          tot += s_w[root];
          s_par[root] = (uint16_t)nxt;
          ++root;
        }
      }
      s_w[nxt] = tot;
    }
    s_dep[m - 2] = 0;
    for (int i = m - 3; i >= 0; --i) s_dep[i] = s_dep[s_par[i]] + 1;
    for (int i = 0; i < m; ++i) {
      int d = s_dep[s_leafpar[i]] + 1;
      if (d > 15) d = 15;
      s_cnt[d] += 1;
    }
    uint32_t total = 0;
    for (int l = 1; l <= 15; ++l) total += s_cnt[l] << (15 - l);
    while (total > (1u << 15)) {
      s_cnt[15] -= 1;
      for (int l = 14; l >= 1; --l)
        if (s_cnt[l]) {
          s_cnt[l] -= 1;
          s_cnt[l + 1] += 2;
          break;
        }
      total -= 1;
    }
    int i = 0;
    for (int l = 15; l >= 1; --l)
      for (uint32_t k = 0; k < s_cnt[l]; ++k) s_len[s_sorted[i++]] = (uint8_t)l;
    uint32_t code = 0;
    s_cnt[0] = 0;
    for (int l = 1; l <= 15; ++l) {
      code = (code + s_cnt[l - 1]) << 1;
      s_first[l] = code;
    }
  }
  __syncthreads();
  for (int i = tid; i < m; i += kThreads) {  // canonical: index among the symbols of the same length, in symbol order
    const uint32_t l = s_len[i];
    uint32_t idx = 0;
    for (int u = 0; u < i; ++u) idx += s_len[u] == l;
    s_code[i] = rev_bits(s_first[l] + idx, l) | (l << 16);
  }
  __syncthreads();
  if (tid == 0) {
    // the block header behind the 18 BGZF bytes: prefix (HLIT = 286, HDIST = 1, HCLEN = 19, the fixed code-length code),
    // then the 286 + 1 code lengths; a length that repeats is run-length coded with symbol 16 (3..6 copies of the previous)
    BitWriter bw;
    bw.init(s_img, kHeadBits);
    bw.put64(kPrefixLo | ((uint64_t)(m - 257) << 3), 64);
    bw.put(kPrefixHi, kPrefixBits - 64);
    uint32_t bits = kHeadBits + kPrefixBits;
    int p = 0;
    const int npos = m + 1;
    while (p < npos) {
      const uint32_t v = p < m ? s_len[p] : 1u;   // the single distance code has length 1
      bw.put(cl_code(v), cl_len(v));
      bits += cl_len(v);
      ++p;
      int r = 0;
      while (p + r < npos && (p + r < m ? s_len[p + r] : 1u) == v) ++r;
      while (r >= 3) {
        const int t = r < 6 ? r : 6;
        bw.put(cl_code(16) | ((uint32_t)(t - 3) << cl_len(16)), cl_len(16) + 2);
        bits += cl_len(16) + 2;
        r -= t;
        p += t;
      }
    }
    bw.flush();
    tbl->hdr_end_bits = bits;
  }
  __syncthreads();
  for (int i = tid; i < kSyms; i += kThreads) tbl->code[i] = s_code[i];
  for (int i = tid; i < kHdrDw; i += kThreads) tbl->hdr[i] = s_img[4 + i];
}

// ---------------------------------------------------------------------------------------------------------------------
// Where a member goes: straight to its final place.  The members of a launch are packed densely, so member c starts at the
// sum of the sizes in front of it -- and a workgroup knows its member's size long before it has the bits (the sizes come out
// of the token scan, the emission follows).  Decoupled look-back: the workgroup publishes its size as soon as it has it
// (status word = epoch | flag | value; flag 1 = this member's size, 2 = inclusive prefix), emits its bits, and only then
// looks back over its predecessors' words -- 64 at a time, one per lane of the first wave -- until it meets an inclusive
// prefix; by then the workgroups in front of it have usually published theirs, and the wait is one read.  Chunk numbers are
// tickets drawn from a counter (not blockIdx), so a workgroup only ever waits for workgroups that started before it.  The
// status words carry the launch's epoch: no array is cleared between launches.
//   dense may be device memory (a piece that a copy carries to the host) or page-locked HOST memory mapped into the GPU's
// address space: the member then crosses the link as the workgroup's own stores, no copy exists at all.  Either way a
// member leaves LDS once, as 16-byte stores at the destination's alignment (the image is read at the byte offset that
// makes them so), byte stores on its ragged ends.
// ---------------------------------------------------------------------------------------------------------------------
struct DfCtl {               // device memory, zeroed by launch_deflate in front of every launch
  uint32_t ticket;
  uint32_t error;            // 1: a look-back gave up waiting (never seen; every spin is bounded all the same)
  int64_t total;             // bytes of the launch's members (written by the workgroup of the last chunk)
};
static_assert(sizeof(DfCtl) == DF_CTL_BYTES, "DF_CTL_BYTES");

constexpr uint64_t kStSize = 1ull << 32, kStPrefix = 2ull << 32;
__device__ __forceinline__ uint64_t st_word(uint32_t epoch, uint64_t flag, uint32_t v) { return ((uint64_t)epoch << 34) | flag | v; }

// exclusive prefix of chunk c (> 0) from the status words in front of it; called by the 64 lanes of one wave
__device__ __forceinline__ uint32_t look_back(const uint64_t *status, int64_t c, uint32_t epoch, int lane, uint32_t *error) {
  uint32_t sum = 0;
  int64_t hi = c - 1;  // the window is chunks hi, hi - 1, ..., hi - 63
  // (a workgroup only waits for workgroups that drew their tickets before it, and those need nothing from it; the bound is
  // for the case that can not happen -- a kernel that spins for ever takes the GPU with it)
  for (uint32_t spins = 0;; spins++) {
    if (spins > (1u << 24)) {  // ~2 s of s_sleep
      if (lane == 0) atomicOr(error, 1u);
      return sum;
    }
    const int64_t i = hi - lane;
    uint64_t w = 0;
    if (i >= 0) w = __hip_atomic_load(&status[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool mine = i >= 0 && (uint32_t)(w >> 34) == epoch && ((w >> 32) & 3u) != 0;
    const uint64_t ready = __ballot(mine || i < 0);
    const uint64_t pref = __ballot(mine && ((w >> 32) & 3u) == 2u);
    // lanes 0 .. k - 1 ready, where k = first lane that is not; the nearest inclusive prefix inside them ends the walk
    const int k = ~ready ? __builtin_ctzll(~ready) : 64;
    const uint64_t usable = k >= 64 ? ~0ull : ((1ull << k) - 1);
    const uint64_t stop = pref & usable;
    if (stop) {
      const int e = __builtin_ctzll(stop);  // lanes 0..e: sizes of hi .. hi - e + 1 and the inclusive prefix of hi - e
      uint32_t v = (lane <= e && i >= 0) ? (uint32_t)w : 0u;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
      return sum + v;
    }
    if (k == 0) {  // the chunk right in front has not published yet
      __builtin_amdgcn_s_sleep(2);
      continue;
    }
    uint32_t v = (lane < k && i >= 0) ? (uint32_t)w : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    sum += v;
    hi -= k;
    if (hi < 0) return sum;  // walked past chunk 0: everything in front was sizes
  }
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// s_img[0 .. bytes) -> dst (any alignment): 16-byte stores where dst is 16-byte aligned, bytes on the ragged ends
__device__ __forceinline__ void store_member(uint8_t *dst, const uint32_t *s_img, uint32_t bytes, int tid) {
  const uint32_t head = min((uint32_t)((16u - ((uintptr_t)dst & 15u)) & 15u), bytes);
  const uint8_t *s8 = reinterpret_cast<const uint8_t *>(s_img);
  if ((uint32_t)tid < head) dst[tid] = s8[tid];
  const uint32_t nvec = (bytes - head) >> 4;
  const uint32_t sh = head & 3u, dw0 = head >> 2;
  u32x4 *d4 = reinterpret_cast<u32x4 *>(dst + head);
  for (uint32_t v = tid; v < nvec; v += kThreads) {
    const uint32_t *p = s_img + dw0 + 4u * v;  // image bytes head + 16 v ..: dwords p[0..4] shifted by sh bytes
    const uint32_t a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];   // (the image is padded past the member)
    u32x4 o;
    o.x = __builtin_amdgcn_alignbyte(b, a, sh);
    o.y = __builtin_amdgcn_alignbyte(c, b, sh);
    o.z = __builtin_amdgcn_alignbyte(d, c, sh);
    o.w = __builtin_amdgcn_alignbyte(e, d, sh);
    __builtin_nontemporal_store(o, &d4[v]);
  }
  const uint32_t done = head + 16u * nvec;
  if ((uint32_t)tid < bytes - done) dst[done + tid] = s8[done + tid];
}

__global__ __launch_bounds__(kThreads) void k_deflate_chunks(const uint8_t *__restrict__ text, int64_t n_bytes,
                                                              uint8_t *__restrict__ dense, uint64_t *__restrict__ status,
                                                              DfCtl *__restrict__ ctl, uint32_t epoch,
                                                              const uint32_t *__restrict__ crc_table,
                                                              const uint32_t *__restrict__ pow128, uint32_t x8rem,
                                                              const DfTable *__restrict__ tbl,
                                                              unsigned long long *__restrict__ prof) {
  // The chunk's LDS image lives only until every thread holds its 128 bytes in registers; the same memory is then the CRC
  // tables' home and finally the (zeroed) output buffer the member is assembled in.  35 KB per workgroup.
  __shared__ uint32_t s_in[kInDw];
  static_assert(kOutDw <= kInDw, "the output image must fit the input image");
  uint32_t *const s_out = s_in;
  __shared__ uint32_t s_code[kSyms];     // reversed code | length << 16 (the stream's table; [287] = 0)
  uint32_t *const s_crc = s_in;          // [1024] slice-by-4 tables, between the register load and the emission
  __shared__ uint32_t s_wave[kThreads / 64];
  __shared__ uint32_t s_misc[4];         // 0 ticket, 1 exclusive prefix, 3 crc

  const int tid = threadIdx.x, lane = tid & 63;
  unsigned long long t_prev = prof ? wall_clock64() : 0;
  int phase = 0;
  auto mark = [&]() {   // PBSIM_DEFLATE_PROF: per-phase time of lane 0, summed over chunks (100 MHz ticks)
    if (prof && tid == 0) {
      const unsigned long long t = wall_clock64();
      atomicAdd(&prof[phase], t - t_prev);
      t_prev = t;
    }
    ++phase;
  };
  for (int i = tid; i < kSyms; i += kThreads) s_code[i] = i < 286 ? tbl->code[i] : 0u;
  if (tid < 4) s_misc[tid] = tid == 0 ? atomicAdd(&ctl->ticket, 1u) : 0u;
  __syncthreads();
  const int64_t chunk = s_misc[0];
  const int64_t n_chunks = (n_bytes + kChunk - 1) / kChunk;
  uint32_t seg[kSegDw], prev_dw;  // the thread's segment, in registers from here on (every loop over it is fully unrolled)
  int n, seg_n;
  load_chunk(text, n_bytes, chunk, s_in, seg, &n, &seg_n, &prev_dw);
  const int beg = tid * kSeg;
  mark();  // 0 stage
  for (int i = tid; i < kInDw; i += kThreads) s_in[i] = (i < 1024) ? crc_table[i] : 0u;  // CRC tables, then the output buffer
  __syncthreads();
  const SegTokens tk = seg_tokens(seg, seg_n, prev_dw, tid == 0);

  // ---- CRC of the segment, and the size of its tokens
  uint32_t tok_bits = 0;
  {
    uint32_t c = tid == 0 ? 0xFFFFFFFFu : 0u;
    const int full = seg_n >> 2;
#pragma unroll
    for (int j = 0; j < kSegDw; ++j) {
      if (j < full) {
        const uint32_t x = c ^ seg[j];
        c = s_crc[768 + (x & 255u)] ^ s_crc[512 + ((x >> 8) & 255u)] ^ s_crc[256 + ((x >> 16) & 255u)] ^ s_crc[x >> 24];
      } else if (j == full) {
        for (int k = 0; k < (seg_n & 3); ++k) c = s_crc[(c ^ (seg[j] >> (8 * k))) & 255u] ^ (c >> 8);
      }
    }
    // bytes after this segment = 128 * (q - tid - 1) + r for full segments, 0 for the last (partial) one
    const int q = n >> 7, r = n & 127;
    if (seg_n == kSeg && tid < q) {
      c = gf2_mul(c, pow128[q - tid - 1]);
      if (r) c = gf2_mul(c, x8rem);
    }
    if (seg_n == 0) c = 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c ^= __shfl_xor(c, d, 64);
    if (lane == 0) atomicXor(&s_misc[3], c);
  }
  mark();  // 1 crc
  {
#pragma unroll
    for (int j = 0; j < kSegDw; ++j) {
      const uint32_t nl = m_nibble(tk.lit, j);
      if (nl == 0) continue;
      const uint32_t w = seg[j];
#pragma unroll
      for (int k = 0; k < 4; ++k)   // s_code[287] is zero: no length for non-literal positions
        tok_bits += s_code[((nl >> k) & 1u) ? (w >> (8 * k)) & 255u : 287u] >> 16;
    }
    Mask128 st = tk.start;
    while (st.lo | st.hi) {
      const int pos = m_first(st);
      st = m_clear(st, pos);
      uint32_t sy, eb, ev;
      length_symbol((uint32_t)m_run(tk.cover, pos), &sy, &eb, &ev);
      tok_bits += (s_code[sy] >> 16) + eb + 1;
    }
  }
  uint32_t tok_total;
  const uint32_t tok_off = block_scan_excl(tok_bits, s_wave, &tok_total);
  const uint32_t hdr_end = tbl->hdr_end_bits;
  const uint32_t eob_len = s_code[256] >> 16;
  const uint32_t payload_bits = hdr_end - kHeadBits + tok_total + eob_len;
  const uint32_t payload_bytes = (payload_bits + 7) >> 3;
  const bool huff = (18 + payload_bytes + 8 <= (uint32_t)kOutBytes) && (payload_bytes < (uint32_t)n + 5);
  const uint32_t crc = s_misc[3] ^ 0xFFFFFFFFu;
  const uint32_t member = huff ? 18 + payload_bytes + 8 : 18 + 5 + (uint32_t)n + 8;
  // the size is known: publish it (chunk 0's size is its inclusive prefix), the bits follow
  if (tid == 0)
    __hip_atomic_store(&status[chunk], st_word(epoch, chunk == 0 ? kStPrefix : kStSize, member), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  mark();  // 2 sizes + scan

  if (huff) {
    __syncthreads();  // everybody is done with the CRC tables
    for (int i = tid; i < 1024; i += kThreads) s_in[i] = 0;
    __syncthreads();
    // BGZF header: 1f 8b 08 04 | mtime 0 | xfl 0 | os ff | xlen 6 | 'B' 'C' 2 0 | bsize-1, then the stream's block header
    if (tid < kHdrDw) s_out[4 + tid] = tbl->hdr[tid];
    if (tid == 0) {
      s_out[0] = 0x04088b1fu;
      s_out[1] = 0;
      s_out[2] = 0x0006ff00u;
      s_out[3] = 0x00024342u;
    }
    __syncthreads();  // the plain stores above precede everybody's ds_or
    if (tid == 0) atomicOr(&s_out[4], member - 1);
    mark();  // 3 header
    {
      BitWriter bw;
      bw.init(s_out, hdr_end + tok_off);
      // One code path for every dword.  A match covers >= 3 positions, so at most one starts in a dword; its start position is
      // not a literal (s_code[287] == 0 bits stands in for it), and the positions behind it in the dword are covered except
      // byte 3 of a 3-byte match that starts at byte 0.  In stream order the dword's tokens are therefore
      //     [match if it starts at byte 0] c0 c1 [match if at byte 1 or 2] c2 c3 [match if at byte 3]
      // with zero-length entries wherever nothing is emitted.  (A separate loop for dwords that hold a match start was
      // executed, exec-masked, for 96 % of the dwords: some lane of the 64 nearly always has one.)
#pragma unroll
      for (int j = 0; j < kSegDw; ++j) {
        const uint32_t nl = m_nibble(tk.lit, j), ns = m_nibble(tk.start, j);
        if ((nl | ns) == 0) continue;   // inside a match
        const uint32_t w = seg[j];
        uint32_t c[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] = s_code[((nl >> k) & 1u) ? (w >> (8 * k)) & 255u : 287u];   // [287] == 0
        uint32_t mv = 0, ml = 0, ks = 4;
        if (ns) {
          ks = (uint32_t)__builtin_ctz(ns);
          uint32_t sy, eb, ev;
          length_symbol((uint32_t)m_run(tk.cover, 4 * j + (int)ks), &sy, &eb, &ev);
          const uint32_t cm = s_code[sy];
          mv = (cm & 0xFFFFu) | (ev << (cm >> 16));  // code, extra bits, then the single distance code (one 0 bit)
          ml = (cm >> 16) + eb + 1;
        }
        bw.put(ks == 0 ? mv : 0u, ks == 0 ? ml : 0u);
        {
          const uint32_t l0 = c[0] >> 16;   // two codes (<= 30 bits) per accumulator step
          bw.put((c[0] & 0xFFFFu) | ((c[1] & 0xFFFFu) << l0), l0 + (c[1] >> 16));
        }
        const bool mid = ks == 1 || ks == 2;
        bw.put(mid ? mv : 0u, mid ? ml : 0u);
        {
          const uint32_t l2 = c[2] >> 16;
          bw.put((c[2] & 0xFFFFu) | ((c[3] & 0xFFFFu) << l2), l2 + (c[3] >> 16));
        }
        bw.put(ks == 3 ? mv : 0u, ks == 3 ? ml : 0u);
      }
      bw.flush();
    }
    mark();  // 4 tokens (lane 0's own)
    if (tid == 0) {
      BitWriter bw;
      bw.init(s_out, hdr_end + tok_total);
      bw.put(s_code[256] & 0xFFFFu, eob_len);
      bw.flush();
      bw.init(s_out, (18 + payload_bytes) * 8);
      bw.put(crc, 32);
      bw.put((uint32_t)n, 32);
      bw.flush();
    }
    // where the member goes: the first wave looks back while the others finish their bits
    if (tid < 64 && chunk > 0) {
      const uint32_t excl = look_back(status, chunk, epoch, tid, &ctl->error);
      if (tid == 0) {
        s_misc[1] = excl;
        __hip_atomic_store(&status[chunk], st_word(epoch, kStPrefix, excl + member), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
    mark();  // 5 trailer + look-back + barrier
    const uint32_t off = s_misc[1];
    store_member(dense + off, s_out, member, tid);
    if (tid == 0 && chunk == n_chunks - 1) ctl->total = (int64_t)off + member;
    mark();  // 6 store
  } else {
    // stored block (RFC 1951 3.2.4): incompressible input, rare for this text
    if (tid < 64 && chunk > 0) {
      const uint32_t excl = look_back(status, chunk, epoch, tid, &ctl->error);
      if (tid == 0) {
        s_misc[1] = excl;
        __hip_atomic_store(&status[chunk], st_word(epoch, kStPrefix, excl + member), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
    uint8_t *slot = dense + s_misc[1];
    if (tid == 0 && chunk == n_chunks - 1) ctl->total = (int64_t)s_misc[1] + member;
    if (tid == 0) {
      const uint8_t head[23] = {0x1f, 0x8b, 8,   4,   0,   0,
                                0,    0,    0,   0xff, 6,  0,
                                'B',  'C',  2,   0,   (uint8_t)((member - 1) & 255), (uint8_t)((member - 1) >> 8),
                                1,    (uint8_t)(n & 255), (uint8_t)(n >> 8), (uint8_t)(~n & 255), (uint8_t)((~n >> 8) & 255)};
      for (int i = 0; i < 23; ++i) slot[i] = head[i];
      uint8_t *t = slot + 23 + n;
      for (int i = 0; i < 4; ++i) t[i] = (uint8_t)(crc >> (8 * i));
      for (int i = 0; i < 4; ++i) t[4 + i] = (uint8_t)((uint32_t)n >> (8 * i));
    }
#pragma unroll
    for (int j = 0; j < kSegDw; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (4 * j + k < seg_n) slot[23 + beg + 4 * j + k] = (uint8_t)(seg[j] >> (8 * k));
  }
}

uint32_t host_gf2_mul(uint32_t a, uint32_t b) {
  uint32_t p = 0;
  for (int i = 0; i < 32; ++i) {
    if (a & (0x80000000u >> i)) p ^= b;
    b = (b >> 1) ^ ((b & 1u) ? kPoly : 0u);
  }
  return p;
}

uint32_t host_xpow8(uint64_t nbytes) {  // x^(8 nbytes) mod P
  uint32_t r = 0x80000000u, b = 0x00800000u;
  while (nbytes) {
    if (nbytes & 1) r = host_gf2_mul(r, b);
    b = host_gf2_mul(b, b);
    nbytes >>= 1;
  }
  return r;
}

}  // namespace

void deflate_host_tables(uint32_t *crc_table, uint32_t *pow128) {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? kPoly : 0u);
    crc_table[i] = c;
  }
  for (int t = 1; t < 4; ++t)   // slice-by-4: table t advances a byte t positions further
    for (uint32_t i = 0; i < 256; ++i) {
      const uint32_t c = crc_table[(t - 1) * 256 + i];
      crc_table[t * 256 + i] = (c >> 8) ^ crc_table[c & 255u];
    }
  const uint32_t step = host_xpow8(128);
  uint32_t v = 0x80000000u;
  for (int k = 0; k < 256; ++k) {
    pow128[k] = v;
    v = host_gf2_mul(v, step);
  }
}

void launch_deflate_table(const uint8_t *text, int64_t n_bytes, uint32_t *hist, void *table, hipStream_t s) {
  if (n_bytes <= 0) return;
  const int64_t nch = (n_bytes + DF_CHUNK - 1) / DF_CHUNK;
  const int64_t ns = nch < kSampleChunks ? nch : kSampleChunks;
  const int64_t sample_bytes = ns * (int64_t)DF_CHUNK < n_bytes ? ns * (int64_t)DF_CHUNK : n_bytes;
  (void)hipMemsetAsync(hist, 0, 288 * sizeof(uint32_t), s);
  hipLaunchKernelGGL(k_deflate_hist, dim3((unsigned)ns), dim3(kThreads), 0, s, text, sample_bytes, hist);
  hipLaunchKernelGGL(k_deflate_build, dim3(1), dim3(kThreads), 0, s, (const uint32_t *)hist, reinterpret_cast<DfTable *>(table));
}

void launch_deflate(const uint8_t *text, int64_t n_bytes, uint64_t *status, void *ctl, uint32_t epoch, uint8_t *dense,
                    const uint32_t *d_crc_table, const uint32_t *d_pow128, const void *table, hipStream_t s,
                    unsigned long long *d_prof, hipEvent_t ev_begin, hipEvent_t ev_chunks_done) {
  if (n_bytes <= 0) return;
  const int64_t nch = (n_bytes + DF_CHUNK - 1) / DF_CHUNK;
  const uint32_t x8rem = host_xpow8((uint64_t)(n_bytes % DF_CHUNK) & 127u);
  (void)hipMemsetAsync(ctl, 0, DF_CTL_BYTES, s);  // the ticket counter; the status words are told apart by the epoch
  if (ev_begin) (void)hipEventRecord(ev_begin, s);
  hipLaunchKernelGGL(k_deflate_chunks, dim3((unsigned)nch), dim3(kThreads), 0, s, text, n_bytes, dense, status,
                     reinterpret_cast<DfCtl *>(ctl), epoch & 0x3fffffffu, d_crc_table, d_pow128, x8rem,
                     reinterpret_cast<const DfTable *>(table), d_prof);
  if (ev_chunks_done) (void)hipEventRecord(ev_chunks_done, s);
}

}  // namespace pbsim
