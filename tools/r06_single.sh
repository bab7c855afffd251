#!/bin/bash
# single_dataset_ms of the default line for several cuts (RPVG_BENCH_PARTS): 12 steps only, the single-dataset figure is what is read
out=gpurun_out/r06/single${TAG:+_$TAG}; mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
i=0
while IFS= read -r variant; do
  [ -z "$variant" ] && continue
  i=$((i+1))
  env $variant RPVG_BENCH_NO_GIBBS_LINE=1 RPVG_BENCH_NO_HOST_BOUND=1 timeout 200 python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline > $R/$out/v$i.json 2> $R/$out/v$i.err
  python - "$variant" $R/$out/v$i.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    s=d.get("single_dataset",{})
    print(f'{sys.argv[1]:60s} single {s.get("ms")} min {s.get("ms_min")} max {s.get("ms_max")} equal {s.get("equal_whole_batch")} parts {s.get("rows_per_part")} err {s.get("error")} | serial {d["one_batch_in_flight"]["ms_per_step_with_h2d_serial"]:.2f} pipe {d["ms_per_step"]:.2f}')
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done <<< "$VARIANTS"
