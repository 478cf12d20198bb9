#!/bin/bash
# samples GPU power / clocks while bench.py runs (is the chip power-capped under this load?)
python bench.py --steps 3000 --warmup 20 --no-cpu-baseline > gpurun_out/power_bench.log 2>&1 &
BP=$!
sleep 4
for i in 1 2 3 4 5; do
  rocm-smi --showpower --showclocks --showuse --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|GPU use|Temperature \(Sensor (edge|junction)" | tr '\n' ';'; echo
  sleep 0.7
done
wait $BP
grep "^{" gpurun_out/power_bench.log | cut -c1-160
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -3
