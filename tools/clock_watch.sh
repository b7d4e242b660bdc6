#!/bin/bash
# Sample the GPU's shader clock and socket power while a bench line runs:  bash tools/clock_watch.sh <label> <bench args...>
# (the bench is a child process; rocm-smi only reads).  Prints "<label> <sclk line> | <power line>" once a second, then the bench line.
LABEL=$1; shift
python bench.py "$@" > /tmp/cw_$LABEL.json 2>/dev/null &
PID=$!
sleep 18                                   # import, task batch, warmup
N=0
while kill -0 $PID 2>/dev/null; do
  S=$(rocm-smi -c -P 2>/dev/null)
  N=$((N+1)); [ $N -eq 5 ] && echo "$S" > /tmp/cw_raw_$LABEL.txt
  echo "$LABEL $(echo "$S" | grep -i 'sclk' | head -n 1 | sed 's/  */ /g') | $(echo "$S" | grep -i 'power' | head -n 1 | sed 's/  */ /g')"
  sleep 1
done
wait $PID
cat /tmp/cw_raw_$LABEL.txt 2>/dev/null | grep -v "^$" | grep -i "power\|sclk\|mclk\|fclk" | sed 's/  */ /g'
cut -c1-125 /tmp/cw_$LABEL.json
