#!/bin/bash
# Samples rocm-smi power / clocks of GPU 0 while a command runs.  usage: tools/power_sample.sh OUT.txt -- cmd args...
OUT=$1; shift; shift
"$@" > /dev/null 2>&1 &
PID=$!
sleep 20   # (index build / warm-up)
for i in $(seq 1 12); do
  rocm-smi -d 0 --showpower --showclocks --showmaxpower 2>/dev/null | grep -E "Power|sclk|mclk|Max Graphics" >> $OUT
  echo "--" >> $OUT
  sleep 1
  kill -0 $PID 2>/dev/null || break
done
wait $PID
