#!/bin/bash
# model tests + single-lane kernel profile + A/B bench of the pair-layout search (run through gpurun)
tag=${1:-x}
out=/root/repo/gpurun_out/r02/$tag; mkdir -p $out
cd /root/repo
python -m pytest tests/test_hip_models.py tests/test_hip_collapse.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | head -5
cd /tmp; export TMPDIR=/tmp
RPVG_AMD_SINGLE_LANE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline >/dev/null 2>$out/err.log
cp $out/prof/*/*kernel_stats.csv $out/stats_1lane.csv; rm -rf $out/prof
cd /root/repo
python tools/kernel_stats_table.py $out/stats_1lane.csv 2 | grep -i "pair\|resolve\|Build\|bounded"
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sequential ', round(d['ms_per_step'],2), round(d['kernels']['loglik_ms_per_step'],2))"
RPVG_HIP_PAIR_LAYOUT=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pair rows  ', round(d['ms_per_step'],2), round(d['kernels']['loglik_ms_per_step'],2))"
done
