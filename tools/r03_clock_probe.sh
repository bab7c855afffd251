rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk" | head -4
rocm-smi --showperflevel 2>&1 | grep -i perf | head -2
(python tools/em_iter_latency.py 200000 2>&1 | head -3) &
sleep 2.5
for i in 1 2 3; do rocm-smi --showclocks 2>&1 | grep -i "sclk" | head -1; sleep 0.3; done
wait
echo "--- set perf level high"
rocm-smi --setperflevel high 2>&1 | tail -2
rocm-smi --showclocks 2>&1 | grep -i "sclk" | head -1
python tools/em_iter_latency.py 20000 2>&1 | head -3
rocm-smi --setperflevel auto 2>&1 | tail -1
