#!/bin/bash
# usage (GPU box, repo root): tools/closed_ab/deflate_solo.sh OUT  -- k_deflate_chunks alone on ~1 GB of MAF-like text: duration of
# its full-size launches (8192 chunks of 32 KB) with nothing else on the GPU
out=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/$out
cat > /tmp/deflate_solo.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
import pbsim3_amd as P
rng = np.random.default_rng(1)
acgt = np.frombuffer(b"ACGT-", dtype=np.uint8)
L = 10000
maf = b"".join(b"a\ns ref 12345 %d + 100000000 " % L + acgt[rng.choice(5, L, p=[.235, .235, .235, .235, .06])].tobytes() + b"\ns S1_%d 0 %d + %d " % (i, L, L)
               + acgt[rng.choice(5, L, p=[.235, .235, .235, .235, .06])].tobytes() + b"\n\n" for i in range(3000))
big = maf * 18
with P.Context(P.default_params(), 0) as ctx:
    ctx.deflate_buffer(maf[:1 << 20])
    z = ctx.deflate_buffer(big)
    print(len(big), "->", len(z))
PY
rocprofv3 --kernel-trace --output-format csv -d $R/$out/t -- python3 /tmp/deflate_solo.py > $R/$out/log.txt 2>&1
cd $R
tail -1 $out/log.txt
python3 - $out <<'PY'
import csv,glob,sys
v=[]
for f in glob.glob(sys.argv[1]+"/t/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_deflate_chunks" in r["Kernel_Name"] and int(r["Grid_Size_X"])>=8192*256:
            v.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
v.sort()
print("full-size launches alone: n=%d median %.2f ms p10 %.2f p90 %.2f  -> %.0f GB/s of text" % (len(v), v[len(v)//2], v[len(v)//10], v[len(v)*9//10], 8192*32768/v[len(v)//2]/1e6))
PY
