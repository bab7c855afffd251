mkdir -p gpurun_out/r04/exp
for L in 2 3 4; do
  RPVG_AMD_LANES=$L python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/exp/lanes$L.json 2>/dev/null
done
RPVG_HIP_NO_COLLAPSE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/exp/nocollapse.json 2>/dev/null
RPVG_HIP_NO_EM_COLLAPSE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/exp/noemcollapse.json 2>/dev/null
RPVG_AMD_LANES=3 RPVG_HIP_NO_COLLAPSE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/exp/lanes3_nocollapse.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --in-flight 2 > gpurun_out/r04/exp/inflight2.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04/exp/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d["ms_per_step"],2), round(d.get("ms_per_step_resident",0),2), round(d.get("gpu_active_frac",0),3))
    except Exception as e: print(f, "ERR", e)
PY
