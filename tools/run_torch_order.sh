#!/bin/bash
for mode in none import after after; do
  if [ $mode = none ]; then unset EXP_TORCH; else export EXP_TORCH=$mode; fi
  echo "== EXP_TORCH=$mode"; EXP_HOST=2 timeout 300 python tools/exp_throughput.py 384 8 2>&1 | grep -E "torch after|libamdhip|reg/s|rate|steady|Error|error" | head -5
done
