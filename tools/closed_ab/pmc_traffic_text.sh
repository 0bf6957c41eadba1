#!/bin/bash
# usage (GPU box, repo root): tools/closed_ab/pmc_traffic_text.sh OUTDIR -- HBM traffic of the text emission and of the deflate kernel
# (FETCH_SIZE and WRITE_SIZE in separate --pmc passes, each alone with --kernel-trace; raw counts x 1024 B).  Both kernels
# read with 16 bytes per lane, for which gfx950's FETCH_SIZE tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md,
# HBM / rocprofv3 section): the read figure is doubled before it is compared with the algorithmic bytes.
out=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/$out/f $R/$out/w $R/$out/df $R/$out/dw
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$out/f -- python3 $R/tools/walk_solo.py errhmm 2 > $R/$out/log_f.txt 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$out/w -- python3 $R/tools/walk_solo.py errhmm 2 > $R/$out/log_w.txt 2>&1
export NOPROF=1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$out/df -- python3 $R/tools/deflate_prof.py > $R/$out/log_df.txt 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$out/dw -- python3 $R/tools/deflate_prof.py > $R/$out/log_dw.txt 2>&1
cd $R
python3 - $out <<'PY'
import csv, glob, collections, sys, json
out = sys.argv[1]
res = {}
for kern, dirs in (("k_text_rows", ("f", "w")), ("k_text_fill", ("f", "w")), ("k_text_headers", ("f", "w")), ("k_deflate_chunks", ("df", "dw"))):
    vals = collections.defaultdict(list)
    grid = collections.Counter()
    for d in dirs:
        for f in glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if kern in r["Kernel_Name"]:
                    if kern == "k_deflate_chunks" and int(r["Grid_Size"]) // 256 < 1000:
                        continue      # full-size launches only
                    vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    grid[r["Grid_Size"]] += 1
    if not vals:
        continue
    fetch = sum(vals["FETCH_SIZE"]) / max(1, len(vals["FETCH_SIZE"])) * 1024
    write = sum(vals["WRITE_SIZE"]) / max(1, len(vals["WRITE_SIZE"])) * 1024
    res[kern] = {"launches_seen": len(vals["FETCH_SIZE"]), "fetch_bytes_raw": fetch, "fetch_bytes_corrected_x2": 2 * fetch,
                 "write_bytes": write, "grids": dict(grid.most_common(3))}
print(json.dumps(res, indent=1))
json.dump(res, open(out + "/traffic.json", "w"), indent=1)
PY
rm -rf $out/f $out/w $out/df $out/dw
