#!/usr/bin/env python3
"""usage: tools/pmc_to_json.py DIR TAG -- the rocprofv3 counter passes of tools/pmc_round.sh as one JSON document
(profiles/<TAG>_walk_pmc.json; bench.py reads the newest one by name for roofline.issue_bound / roofline.traffic)."""
import collections
import csv
import glob
import json
import os
import sys

D, TAG = sys.argv[1], sys.argv[2]
CLOCK_GHZ = 2.4        # MI355X_MICROARCH.md: shader clock
SIMDS = 1024           # 256 CUs x 4


def counters(name, kernel):
    agg, dur = collections.defaultdict(list), []
    for f in glob.glob(os.path.join(D, name, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(D, name, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    return {k: sum(v) / len(v) for k, v in agg.items()}, (sum(dur) / len(dur) if dur else None), len(dur)


def solo_line(name):
    for line in open(os.path.join(D, name + ".plain.log"), errors="replace"):
        if line.startswith("WALK_SOLO_JSON "):
            return json.loads(line[len("WALK_SOLO_JSON "):])
    return None


def calib():
    out = {}
    expect = {"k_stream16": 4 << 30, "k_gather8": (64 << 20) * 64, "k_store4": 4 << 30}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        for k, n in expect.items():
            c, ms, calls = counters("calib_" + ctr, k)
            if ctr in c:
                out.setdefault(k, {})[ctr + "_KiB"] = c[ctr]
                out[k][ctr + "_x1024_over_known_bytes"] = c[ctr] * 1024 / n
                out[k]["known_bytes"] = n
    out["note"] = ("raw counter x 1024 / bytes known by construction (tools/fetch_calib.hip): k_stream16 = the guide's wide coalesced read "
                   "(expected 0.5), k_gather8 = one 8-byte load per 64-byte sector (the walk's reference window; known_bytes counts whole "
                   "sectors), k_store4 = a dword per lane, full lines, nontemporal (the walk's rows)")
    return out


def kernel_block(name, kernel, what):
    solo = solo_line(name)
    c = {}
    ms = {}
    for p in ("sq1", "sq2", "fetch", "write"):
        cc, m, calls = counters(name + "_" + p, kernel)
        c.update(cc)
        ms[p] = m
    if not c or not solo:
        return {"error": "no counters collected for " + kernel}
    cols = solo["maf_columns_per_launch"]
    steps = cols / 64.0
    b = {"what": what, "kernel": kernel, "env": solo["env"], "reads_per_launch": solo["reads_per_launch"],
         "bases_per_launch": solo["bases_per_launch"], "maf_columns_per_launch": cols,
         "launch_ms_without_counters": solo["avg_ms"], "launch_ms_under_counters": ms,
         "per_launch": c,
         "per_wave_step": {"note": "wave-step = 64 MAF columns (64 lanes x 1 column of the lane walker; one step of the wave walker)",
                           "wave_steps_per_launch": steps}}
    for k, n in (("valu", "SQ_INSTS_VALU"), ("salu", "SQ_INSTS_SALU"), ("lds", "SQ_INSTS_LDS"), ("branch", "SQ_INSTS_BRANCH"),
                 ("vmem_rd", "SQ_INSTS_VMEM_RD"), ("vmem_wr", "SQ_INSTS_VMEM_WR")):
        if n in c:
            b["per_wave_step"][k] = c[n] / steps
    pws = b["per_wave_step"]
    if "lds" in pws and "vmem_rd" in pws:
        # where the step's reads go (north_star: "LDS hit-rate on the HMM tables"): every table is staged in LDS once per
        # workgroup, so the LDS instructions ARE the table look-ups; the vector-memory reads are the reference window's refills
        # (0.25 per column in the lane walker: one 8-byte load per 8 bases in two streams) and a read's header words
        b["lds_table_reads_per_column"] = pws["lds"]
        b["vmem_reads_per_column"] = pws["vmem_rd"]
        b["lds_share_of_read_instructions"] = pws["lds"] / (pws["lds"] + pws["vmem_rd"])
    if "SQ_INSTS_VALU" in c and ms["sq1"]:
        b["valu_busy_fraction"] = c["SQ_INSTS_VALU"] * 4 / (SIMDS * CLOCK_GHZ * 1e9 * ms["sq1"] / 1e3)
        b["valu_busy_note"] = "SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x 2.4 GHz x the launch under the SQ pass)"
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        b["lds_bank_conflict_fraction"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        b["traffic"] = {"FETCH_SIZE_KiB": c["FETCH_SIZE"], "WRITE_SIZE_KiB": c["WRITE_SIZE"],
                        "bytes_per_launch_raw_x1024": (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024,
                        "bytes_per_base_raw_x1024": (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / solo["bases_per_launch"]}
    return b


doc = {"round": TAG,
       "command": "tools/pmc_round.sh " + TAG + " (rocprofv3 --pmc <set> --kernel-trace, one counter set per run: 2 x SQ, FETCH_SIZE, "
                  "WRITE_SIZE; tools/walk_solo.py = one walk launch at a time; the same launches timed without counters beside them)",
       "calibration": calib(),
       "k_walk_errhmm_job_occupancy": kernel_block("lane1", "k_walk_errhmm<", "lane walker, ONE workgroup per CU (PBSIM_WALK_LDS_KB=81: the delivered job, which leaves the rest of the CU's LDS to its deflate workgroups)"),
       "k_walk_errhmm_three_per_cu": kernel_block("lane3", "k_walk_errhmm<", "lane walker, three workgroups per CU (PBSIM_WALK_LDS_KB=41: the delivered job of rounds 2-3)"),
       "k_walk_errhmm": kernel_block("lane5", "k_walk_errhmm<", "lane walker, five workgroups per CU (batch primitives, the job with its text left in HBM)"),
       "k_walk_errhmm_coop": kernel_block("coop", "k_walk_errhmm_coop", "wave walker, every read of the batch (PBSIM_COOP_LEN=0), as many persistent workgroups as are resident (768 of eight waves: six waves per SIMD)")}
# the fields bench.py reads (same names as profiles/r02z_walk_pmc.json / r02z_walk_traffic.json)
lane = doc["k_walk_errhmm_job_occupancy"]
if "per_wave_step" in lane and "valu" in lane["per_wave_step"]:
    doc["per_wave_step"] = lane["per_wave_step"]
    doc["valu_busy_fraction"] = lane.get("valu_busy_fraction")
    cal = doc["calibration"]
    fscale = 1.0 / cal["k_gather8"]["FETCH_SIZE_x1024_over_known_bytes"] if "k_gather8" in cal and cal["k_gather8"].get("FETCH_SIZE_x1024_over_known_bytes") else 1.0
    wscale = 1.0 / cal["k_store4"]["WRITE_SIZE_x1024_over_known_bytes"] if "k_store4" in cal and cal["k_store4"].get("WRITE_SIZE_x1024_over_known_bytes") else 1.0
    if "traffic" in lane:
        t = lane["traffic"]
        doc["traffic_bytes_per_launch"] = t["FETCH_SIZE_KiB"] * 1024 * fscale + t["WRITE_SIZE_KiB"] * 1024 * wscale
        doc["bases_per_launch"] = lane["bases_per_launch"]
        doc["traffic_note"] = ("FETCH_SIZE x 1024 x %.3f + WRITE_SIZE x 1024 x %.3f: the raw counters scaled by what the calibration kernels of the same "
                               "access patterns report against their known bytes (the guide: raw x 1024; gfx950 corrections per access width)" % (fscale, wscale))
print(json.dumps(doc, indent=1))
