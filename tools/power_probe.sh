#!/bin/bash
# Samples rocm-smi power / clocks while bench.py runs (is the step power-limited?).  usage: tools/power_probe.sh [steps]
cd ${GRAFT_REPO_ROOT:-.}
python3 bench.py --steps ${1:-400} --warmup 5 --no-cpu-baseline --no-prof > /tmp/pp_bench.log 2>&1 &
BP=$!
sleep 25
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|memory)" | tr '\n' ';' ; echo
  sleep 0.7
done
wait $BP
tail -1 /tmp/pp_bench.log | cut -c1-200
rocm-smi --showmaxpower 2>/dev/null | grep -i power
