cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
python3 - <<'PY'
import time, multiprocessing as mp
def spin(q):
    t=time.time(); n=0
    while time.time()-t<1.0:
        for _ in range(10000): n+=1
    q.put(n)
for k in (1,8,16,32,64,128):
    q=mp.Queue(); ps=[mp.Process(target=spin,args=(q,)) for _ in range(k)]
    [p.start() for p in ps]; tot=sum(q.get() for _ in ps); [p.join() for p in ps]
    print(k, "procs: total work", tot/1e6, "per proc", tot/1e6/k)
PY
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | grep -i thrott
