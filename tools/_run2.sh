cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_hip_large_clusters.py -m gpu -x -q 2>&1 | tail -8
cd /tmp; export TMPDIR=/tmp
RPVG_BENCH_NO_SINGLE=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06/prof_c2 -- python $GRAFT_REPO_ROOT/bench.py --workload c2 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $GRAFT_REPO_ROOT/gpurun_out/r06/c2_fused_profiled.json
cd $GRAFT_REPO_ROOT
cp gpurun_out/r06/prof_c2/*/*kernel_stats.csv gpurun_out/r06/c2_fused_kernel_stats.csv; rm -rf gpurun_out/r06/prof_c2
head -4 gpurun_out/r06/c2_fused_kernel_stats.csv | cut -c1-150
timeout 300 python bench.py --workload c2 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r06/c2_fused.json
python - <<'PY'
import json
for n in ("c2_fused",):
    d=json.loads(open(f"gpurun_out/r06/{n}.json").read())
    print(n, round(d["ms_per_step"],2), d["step_breakdown_ms"]["build_and_compaction"], d["step_breakdown_ms"]["streaming_passes"], d["mass_conserved"])
PY
