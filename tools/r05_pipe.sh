#!/bin/bash
# pipeline bench variants: $@ = env assignments per run, separated by ---
run() { env "$@" timeout 300 python bench.py --steps ${STEPS:-60} --no-cpu-baseline 2>/tmp/bench.err > /tmp/bench.json; tail -2 /tmp/bench.err | grep -v amdgpu.ids; python - "$*" <<PY
import json,sys
d=json.load(open("/tmp/bench.json"))
sp=d["ms_per_step_spread"]
print(sys.argv[1] or "default", "| ms/step", round(d["ms_per_step"],2), "cadence median", round(sp["median"],2), "resident", round(d["ms_per_step_resident"],2), "cpu", round(d["host_cpu_ms_per_step"],1), "upload ms", round(d["h2d_ms_per_batch"],2), "copies", round(d.get("h2d_copies_device_ms_per_batch") or 0,2), "kernels", round(d.get("h2d_kernels_device_ms_per_batch") or 0,2), "worker", {k: round(v,2) for k,v in d["pipeline"]["worker_ms_per_batch"].items() if k!="note"}, "p10/p90/max", round(sp["p10"],2), round(sp["p90"],2), round(sp["max"],2), "active", round(d["gpu_active_frac"],2), "equal", d.get("pipeline_estimates_equal_single_engine"), "one-batch", round(d["one_batch_in_flight"]["ms_per_step_with_h2d"],2))
PY
}
args=()
for a in "$@"; do if [ "$a" = "---" ]; then run "${args[@]}"; args=(); else args+=("$a"); fi; done
run "${args[@]}"
