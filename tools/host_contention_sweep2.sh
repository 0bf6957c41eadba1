R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r05_host_sweep2.txt
cc -O2 -pthread -o /tmp/host_load $R/tools/host_load.c || exit 1
node=$(python3 -c "
import re, pbsim3_amd as P
m = re.search(r'numa node (\d+)', str(P.bind_host_to_device(0) or ''))
print(m.group(1) if m else 0)")
echo "node $node" > $out
job() { python3 bench.py --no-extras --no-cpu-baseline --steps 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.1f ms per step, pcie_frac %.3f' % (d['ms_per_step'], d['delivery']['pcie_frac']))"; }
for ranks in 2 4 8; do
  HOST_LOAD_READ_ONLY=1 /tmp/host_load $ranks 47 600 0 $node > /tmp/hl.log 2>&1 &
  hl=$!
  sleep 1
  echo "beside $((ranks-1)) x 3 threads READING only: $(job)" >> $out
  kill -INT $hl; wait $hl 2>/dev/null
  tail -1 /tmp/hl.log | sed 's/^/    /' >> $out
done
# the other socket only: real rank's node left alone (virtual ranks 4..7 of 8 land on the other node)
cat $out
