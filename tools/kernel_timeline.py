"""start / duration / gap of the long kernels of the LAST job in a `rocprofv3 --kernel-trace --output-format csv` trace of a
bench.py run with a k_walk_sample in it.   usage: python tools/kernel_timeline.py KERNEL_TRACE.csv"""
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
ws=[i for i,r in enumerate(rows) if 'k_walk_sample' in r['Kernel_Name']]
i0=ws[len(ws)//2]
t0=int(rows[i0]['Start_Timestamp'])
prev_end=int(rows[i0-1]['End_Timestamp'])
for r in rows[i0:]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    name=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0].split('::')[-1].split('<')[0][:28]
    if e-s>100000 or s-prev_end>300000:
        print(f"{(s-t0)/1e6:8.2f} ms  dur {(e-s)/1e6:7.3f}  gap_before {(s-prev_end)/1e6:7.3f}  {name}")
    prev_end=max(prev_end,e)
