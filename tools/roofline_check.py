#!/usr/bin/env python3
"""usage: tools/roofline_check.py KERNEL_STATS.csv BENCH_LINE.json [BENCH_DETAIL.json] -- re-derives the bench line's roofline
figures from the rocprofv3 --kernel-trace --stats summary OF THE SAME RUN (tools/profile_round.sh keeps both side by side in
profiles/).  The line is bench.py's compact one; the secondary kernels' rows are in its sidecar (third argument).

dominant kernel: achieved = roofline.alg_bytes_per_launch / (the walk kernel's average duration in the CSV); the line's own
`avg_launch_ms` comes from HIP events on the walk streams around lane walker + wave walker of a round, so the two agree when
the wave walker (launched beside it) ends first -- the usual case; the script prints both and their ratio.  Since round 3
a batch whose reads all go to the wave walker (the one-read tail batches) does not launch the lane walker at all, so every
`k_walk_errhmm` call in the CSV is a bulk launch (round 2's CSVs mixed 40 empty 4.5 us launches into the average).
secondary kernels: `k_text_rows` (+ headers + fill = one emission) and `k_deflate_chunks` the same way."""
import csv
import json
import sys


def main():
    stats, line = sys.argv[1], sys.argv[2]
    detail = sys.argv[3] if len(sys.argv) > 3 else None
    rows = {}
    for r in csv.DictReader(open(stats)):
        name = r["Name"].replace("pbsim::(anonymous namespace)::", "").replace("void ", "")
        rows[name.split("(")[0].split("<")[0]] = rows.get(name.split("(")[0].split("<")[0], [0, 0.0])
        rows[name.split("(")[0].split("<")[0]][0] += int(r["Calls"])
        rows[name.split("(")[0].split("<")[0]][1] += float(r["TotalDurationNs"])
    d = json.loads([l for l in open(line) if l.startswith("{")][-1])
    rf = d["roofline"]
    kern = rf["kernel"]
    calls, total = rows[kern]
    avg_ms = total / calls / 1e6
    ach = rf["alg_bytes_per_launch"] / (avg_ms / 1e3) / 1e9
    print(f"{kern}: CSV {calls} calls, avg {avg_ms:.3f} ms  |  line: {rf['launches']} timed launches, avg {rf['avg_launch_ms']:.3f} ms (HIP events)")
    print(f"  achieved from the CSV  {ach:8.1f} GB/s = {ach / rf['peak']:.4f} of {rf['peak']:.0f}")
    print(f"  achieved in the line   {rf['achieved']:8.1f} GB/s = {rf['frac']:.4f}   (CSV / line = {ach / rf['achieved']:.3f})")
    if detail:
        rf = json.load(open(detail))["roofline"]
    for s in rf.get("secondary", []):
        k = s["kernel"].split(" ")[0]
        if k not in rows:
            continue
        if k == "k_text_rows":
            tot = sum(rows[x][1] for x in ("k_text_rows", "k_text_headers", "k_text_fill") if x in rows)
            avg = tot / rows["k_text_rows"][0] / 1e6
            b = s["bytes_read_per_launch"] + s["bytes_written_per_launch"]
            print(f"{k} (+ headers + fill): CSV avg {avg:.3f} ms per emission -> {b / avg / 1e6:8.1f} GB/s = {b / avg / 1e6 / 8000:.4f}"
                  f"  |  line {s['avg_ms']:.3f} ms, {s['achieved']:.1f} GB/s = {s['frac']:.4f}")
        else:
            c, t = rows[k]
            avg = t / c / 1e6
            print(f"{k}: CSV {c} calls avg {avg:.3f} ms -> {s['text_bytes_per_launch'] / avg / 1e6:8.1f} GB/s of text"
                  f"  |  line {s['avg_ms']:.3f} ms, {s['text_GBps']:.1f} GB/s of text")


if __name__ == "__main__":
    main()
