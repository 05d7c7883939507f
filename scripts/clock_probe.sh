#!/bin/bash
# sclk and power while the headline loop runs on k of 4 partitions: is the chip power-limited with all of them busy?
R=$GRAFT_REPO_ROOT
for K in 1 2 4; do
  python $R/scripts/sustained_load.py $K 7 > /tmp/load_$K.log 2>&1 &
  BG=$!
  while ! grep -q "load starts" /tmp/load_$K.log 2>/dev/null; do sleep 0.3; done
  sleep 1.5
  for i in 1 2 3 4 5; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/.*: //' | tr '\n' ' '; echo
    sleep 0.6
  done
  wait $BG; tail -1 /tmp/load_$K.log
done
