#!/bin/bash
# tools/pc_ab.sh lib1 lib2 ...   -- the pair-count scan of each library on one box, two rounds (same process order)
for round in 1 2; do
  for lib in "$@"; do
    PC_TIME_TEXT=0 PC_TIME_LAUNCHES=40 MBPE_LIB=$lib python tools/pc_time.py 2>/dev/null | sed -e 's/\[.*\]//'
  done
done
