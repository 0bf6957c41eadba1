// host_tables.h -- host side of the path: FIC-HMM model files -> the integer
// lookup tables the kernels consume.  All floating point of the path lives
// here (one-time, libm); the kernels see integers only.
//
// Reference: set_errhmm pbsim.cpp:5640-5714, set_qshmm :5570-5634, the table
// blocks at the top of each simulate_by_* (:3633-3789, :1991-2170, :4166-4340),
// qc[]/uni_ep[] in main (:545-578), set_mut (:5471-5479), hp-del-bias (:673-697).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/pbsim3_amd.h"
#include "modes.h"

namespace pbsim {

constexpr int kAccMax = 100;    // ACCURACY_MAX pbsim.cpp:42
constexpr int kStateMax = 50;   // STATE_MAX    pbsim.cpp:43
constexpr int kQcNum = 94;      // quality codes 0..93
constexpr int kHpSlots = 12;    // hp 0 (Q15) .. 11 (Q1)

struct ErrModel {  // struct errhmm_t, pbsim.cpp:169-178
  double ip[kAccMax + 1][kStateMax + 1];
  double ep[kAccMax + 1][kStateMax + 1][4];
  double tp[kAccMax + 1][kStateMax + 1][kStateMax + 1];
  int exist[kAccMax + 1];
  int state_max[kAccMax + 1];
  int acc_min, acc_max;
};

// struct qshmm_t, pbsim.cpp:160-166: ip[101][51], ep[101][51][94], tp[101][51][51] back to back.  set_qshmm (:5570-5634)
// indexes them with whatever state / column numbers the file holds, and QSHMM-ONT-HQ.model has classes with up to 56
// states: in the compiled reference such an entry lands STATE_MAX+1 slots further on in the same block (the
// next state's row, the next class, the next member) where a later line overwrites it or nobody reads it.  Pinned, not
// refused: the block is ONE flat array with the reference's strides, written without a per-dimension check (only a write
// past the end of tp[] is an error), and read through ip()/ep()/tp() with in-range indices like the table loops do.
struct QsModel {
  static constexpr long kIpN = (long)(kAccMax + 1) * (kStateMax + 1);
  static constexpr long kEpN = kIpN * kQcNum;
  static constexpr long kTpN = kIpN * (kStateMax + 1);
  double blk[kIpN + kEpN + kTpN];
  int exist[kAccMax + 1];
  static long ip_at(long a, long j) { return a * (kStateMax + 1) + j; }
  static long ep_at(long a, long j, long k) { return kIpN + (a * (kStateMax + 1) + j) * kQcNum + k; }
  static long tp_at(long a, long j, long k) { return kIpN + kEpN + (a * (kStateMax + 1) + j) * (kStateMax + 1) + k; }
  double ip(int a, int j) const { return blk[ip_at(a, j)]; }
  double ep(int a, int j, int k) const { return blk[ep_at(a, j, k)]; }
  double tp(int a, int j, int k) const { return blk[tp_at(a, j, k)]; }
};

bool parse_errhmm(const char *path, ErrModel *m, std::string *err);
bool parse_qshmm(const char *path, QsModel *m, std::string *err);

// ---- read-header tables (length / accuracy / transcript start position) ------
struct HeaderTables {
  std::vector<int32_t> prob2len;  // [0..len_rv], slot 0 unused
  int64_t len_rv = 0;
  std::vector<uint8_t> prob2acc;  // [0..acc_rv]
  int64_t acc_rv = 0;
  int acc_lo = 0, acc_hi = 0;     // accuracy_min / accuracy_max of pbsim.cpp:3673-3677
  double mean_len = 0;            // E[L] under prob2len (batch sizing only)
};
bool build_header_tables(const pbsim_params &p, HeaderTables *t, std::string *err);

// start-position buckets per rank (pbsim.cpp:4200-4224): ssp[rank][1..rv]
struct SspTables {
  int rank_max = 0;
  std::vector<int32_t> value;  // [rank][1001]
  std::vector<int32_t> rv;     // [rank]
};
void build_ssp_tables(int rank_max, SspTables *t);

// ---- homopolymer deletion bias (pbsim.cpp:673-697) ------------
struct HpBias {
  double bias[kHpSlots];  // [1..10] as the reference; [11] pinned 0.0
  bool hp11_seen = false; // hp_del_bias[0] aliases hpfreq[11]: non-zero denormal once any hp==11 base was counted
};
void hp_bias_default(HpBias *b);
void hp_bias_from_census(double hp_del_bias, const int64_t hpfreq[kHpSlots], HpBias *b);
// trans pre-pass (pbsim.cpp:4385-4409): hpfreq[v] += weight * (run length) per homopolymer run
void hp_census_weighted(const uint8_t *seq, int64_t len, int64_t weight, int keep_first_case, int64_t hpfreq[kHpSlots]);

// ---- device class tables -------------------------------------------------------
// ERRHMM: one blob per accuracy class acc_lo..acc_hi, uniform stride.
//   +0    u32 hdr[16]: S, init_rv, mode, rate_mag, acc, model_class
//   +64   StateRow rows[S+1] (32 B): u16 tran_rv, emis_rv, E0, E1, del_thr[12]
//   +init u8 init[1000] (padded to 1008)
//   +tran u8 tran[S][1000]   row (state-1)
// E0/E1 are the cumulative ends of emission classes 0/1 with skipped classes
// collapsed, so that e = (i > E0) + (i > E1) equals the expanded emis2err[i].
struct ErrClassTables {
  int acc_lo = 0, acc_hi = 0, smax = 0;
  uint32_t rows_off = 64, emis_off = 0, init_off = 0, tran_off = 0, stride = 0;  // emis_off: 16 B per state {magic, shift | (2^24 - d) << 8, E0' | E1' << 16, del_thr[hp 1] | del_thr[hp 11] << 16}
  bool all_rv_1000 = false;  // every reachable modulus == 1000 (fast-path eligibility)
  std::vector<uint8_t> blob; // (acc_hi-acc_lo+1) * stride
};
// emis_skip_le: WGS skips ep<=0, trans/templ skip ep==0
bool build_err_class_tables(const ErrModel &m, const HeaderTables &h, const HpBias &b, bool emis_skip_le,
                            ErrClassTables *t, std::string *err);

// QSHMM: one blob per class.
//   +0    u32 hdr[16]: S, init_rv, has_model, freq_rv, acc
//   +64   u16 rv[S+1][2]  (tran_rv, emis_rv)
//   +init u8 init[100] (pad 112) | +tran u8 tran[S][100] | +emis u8 emis[S][100] | +freq u8 freq[1000]
// plus class-independent thresholds out of 1e6 (set_mut, pbsim.cpp:5474-5479)
struct QsClassTables {
  int acc_lo = 0, acc_hi = 0, smax = 0;
  uint32_t rv_off = 64, init_off = 0, tran_off = 0, emis_off = 0, freq_off = 0, stride = 0;
  bool all_rv_100 = false;  // every init/transition/emission modulus of a reachable state is 100
  std::vector<uint8_t> blob;
  uint32_t sub_thre[kQcNum], ins_thre[kQcNum];
  uint32_t del_thr[kQcNum][kHpSlots];  // ceil(del_thre[q]*bias[hp]): x < d*b  <=>  x < ceil(d*b)
  double qprob[kQcNum];
};
bool build_qs_class_tables(const QsModel &m, const HeaderTables &h, const HpBias &b, const pbsim_params &p,
                           QsClassTables *t, std::string *err);
// fills qprob / sub_thre / ins_thre / del_thr only (the sampling method has no model)
void emission_magic(uint32_t d, uint32_t *magic, uint32_t *shift);
void build_mut_tables(const pbsim_params &p, const HpBias &b, QsClassTables *t);

}  // namespace pbsim
