#!/bin/bash
# step time and CPU use of the S3 bench under different host-thread policies (gpurun); results under gpurun_out/r02/<tag>
tag=${1:-hostcpu}
out=/root/repo/gpurun_out/r02/$tag; mkdir -p $out
cd /root/repo
cat /sys/fs/cgroup/cpu.max > $out/cpu_max.txt
run() { # name, env...
  name=$1; shift
  t0=$(grep -E "^(usage_usec|nr_throttled|throttled_usec)" /sys/fs/cgroup/cpu.stat | awk '{print $2}' | tr '\n' ' ')
  env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_$name.json
  t1=$(grep -E "^(usage_usec|nr_throttled|throttled_usec)" /sys/fs/cgroup/cpu.stat | awk '{print $2}' | tr '\n' ' ')
  python - <<PY
import json
a=[int(x) for x in "$t0".split()]; b=[int(x) for x in "$t1".split()]
d=json.loads(open("$out/bench_$name.json").read())
print("%-28s ms/step %6.2f  with_h2d %6.2f  cpu %5.1f s  throttled %d times %.2f s" % ("$name", d["ms_per_step"], d.get("ms_per_step_with_h2d",0), (b[0]-a[0])/1e6, b[1]-a[1], (b[2]-a[2])/1e6))
PY
}
run default X=1
run passive OMP_WAIT_POLICY=passive
run threads8 RPVG_AMD_HOST_THREADS=8
run threads8_passive RPVG_AMD_HOST_THREADS=8 OMP_WAIT_POLICY=passive
run threads4_passive RPVG_AMD_HOST_THREADS=4 OMP_WAIT_POLICY=passive
run threads16_passive RPVG_AMD_HOST_THREADS=16 OMP_WAIT_POLICY=passive
run lanes3_threads8_passive RPVG_AMD_LANES=3 RPVG_AMD_HOST_THREADS=8 OMP_WAIT_POLICY=passive
run lanes3_threads4_passive RPVG_AMD_LANES=3 RPVG_AMD_HOST_THREADS=4 OMP_WAIT_POLICY=passive
run lanes4_threads4_passive RPVG_AMD_LANES=4 RPVG_AMD_HOST_THREADS=4 OMP_WAIT_POLICY=passive
run default_again X=1
