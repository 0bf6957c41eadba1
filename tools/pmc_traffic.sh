#!/bin/bash
# usage (GPU box, repo root): tools/pmc_traffic.sh OUTDIR [bench args...]
# HBM traffic of the walk kernel: FETCH_SIZE and WRITE_SIZE in SEPARATE passes
# (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots")
out=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $out/f $out/w
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/f -- python bench.py --steps 2 --warmup 0 --slots 1 --no-cpu-baseline "$@" > $out/bench_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/w -- python bench.py --steps 2 --warmup 0 --slots 1 --no-cpu-baseline "$@" > $out/bench_w.log 2>&1
python3 - $out <<'PY'
import csv,glob,collections,sys,json
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        name="walk" if "k_walk" in k else "text_rows" if "k_text_rows" in k else None
        if name: agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
d=json.loads([l for l in open(out+"/bench_f.log") if l.startswith("{")][-1])
print(json.dumps({"alg_bytes_per_launch": d["roofline"]["alg_bytes_per_launch"], "bases_per_step": d["config"]["bases_per_step"]}))
for name,c in agg.items():
    for k,v in c.items():
        # rocprofv3 reports one row per dispatch (already summed over XCDs); unit = KiB
        print(name,k,"per launch KiB:",[round(x) for x in v], "-> GB:",[round(x*1024/1e9,2) for x in v])
PY
