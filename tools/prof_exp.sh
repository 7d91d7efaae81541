# rocprofv3 kernel statistics of the steady-state pipeline (tools/exp_throughput.py); summary -> gpurun_out/prof_exp_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/prof_exp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_exp -o exp -- python $R/tools/exp_throughput.py ${1:-96} ${2:-8} > $O/prof_exp.log 2>&1
find $O/prof_exp -name "*kernel_trace.csv" -delete
cp $(find $O/prof_exp -name "*kernel_stats.csv" | head -1) $O/prof_exp_stats.csv
tail -2 $O/prof_exp.log | cut -c1-400
