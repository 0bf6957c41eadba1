for n in 4 8 16 32; do
  echo "DEBUG_CLR_LIMIT_BLIT_WG=$n"
  DEBUG_CLR_LIMIT_BLIT_WG=$n tools/closed_ab/deflate_solo.sh gpurun_out/dfsolo_b$n | tail -1
  python3 - gpurun_out/dfsolo_b$n <<PY
import csv,glob,sys
for f in glob.glob(sys.argv[1]+"/t/**/*kernel_trace.csv",recursive=True):
    v=[((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, r["Grid_Size_X"]) for r in csv.DictReader(open(f)) if "copyBuffer" in r["Kernel_Name"] and int(r["End_Timestamp"])-int(r["Start_Timestamp"])>3e5]
    print("   big copy kernels (ms, grid):", [(round(a,2),g) for a,g in v])
PY
  DEBUG_CLR_LIMIT_BLIT_WG=$n python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   bench: value', round(d['value']/1e9,1), 'Gbases/s', round(d['ms_per_step'],1), 'ms  pcie', round(d['delivery']['pcie_frac'],3))"
done
