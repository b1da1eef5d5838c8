#!/bin/bash
# Sample GPU clocks and power while the benchmark loop runs (is the kernel clock- or power-limited?).
root=${GRAFT_REPO_ROOT:-$(pwd)}
rocm-smi --showclocks --showpower --showperflevel 2>&1 | grep -iE "sclk|mclk|power|perf" | head -8
echo "--- under load"
timeout -k 5 40 python3 $root/bench.py --steps 60000 --warmup 10 --no-cpu-baseline > /dev/null 2>&1 &
pid=$!
sleep 8
for i in 1 2 3; do
  rocm-smi --showclocks --showpower 2>&1 | grep -iE "sclk|mclk|fclk|power" | tr '\n' ' '; echo
  sleep 2
done
wait $pid
