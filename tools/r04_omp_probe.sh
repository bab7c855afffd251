cd /root/repo
python - <<'PY'
import os
os.environ.setdefault("OMP_WAIT_POLICY","passive")
import torch
from rpvg_amd import engine
e = engine.Engine(0)
print([l.split()[-1] for l in open("/proc/self/maps") if ("omp" in l and ".so" in l)][::4])
PY
for v in "A=1" "GOMP_SPINCOUNT=0" "KMP_BLOCKTIME=0" "OMP_WAIT_POLICY=active"; do
echo "== $v"; env $v RPVG_BENCH_THREAD_CPU=1 python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>&1 >/dev/null | grep -A5 "thread cpu" | tail -4; env $v RPVG_BENCH_THREAD_CPU=1 python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>&1 >/dev/null | grep "threads "
done
