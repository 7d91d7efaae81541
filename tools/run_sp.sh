#!/bin/bash
for pts in 128 32 8; do
export PLADE_SPACING_PTS=$pts
bash tools/prof_exp.sh 64 8 > /dev/null 2>&1
echo "pts $pts: $(python tools/show_stats.py gpurun_out/prof_exp_stats.csv 80 | grep -E 'k_sp_' | tr '\n' ' ')"
done
