#!/bin/bash
# the drop-in path by the size of the team that calls estimate(): Little's law — calls in flight / latency of a call
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r06/a1_teams; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for t in ${TEAMS:-64 96 128 192 256 384 512 64}; do
timeout 300 python $R/bench.py --workload a1 --team $t --steps 6 --warmup 2 --no-cpu-baseline > $out/team_$t.json 2>/dev/null
python - $t $out/team_$t.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f'team {int(sys.argv[1]):4d}  {d["ms_per_step"]:8.1f} ms per pass  {d["value"]/1e6:7.1f} M read pairs/s  equal={d["estimates_equal_estimate_batch"]}  {d["ms_per_step_in_order"]}')
PY
done
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
