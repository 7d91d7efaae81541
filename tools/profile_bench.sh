#!/bin/bash
# Runs on the GPU box (via gpurun): the bench line, the rocprofv3 kernel statistics of the same command
# and the two HBM counter passes (separate --pmc runs, as MI355X_MICROARCH.md prescribes).
# Outputs under gpurun_out/; `python tools/summarize_profiles.py rN` copies the summaries to profiles/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python bench.py > $O/bench_r1.json 2> $O/bench_r1.err
tail -c 3000 $O/bench_r1.json
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 32 --warmup 8 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- $CMD > $O/prof_bench.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_pmc_fetch -o pmc -- $CMD > $O/prof_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_pmc_write -o pmc -- $CMD > $O/prof_pmc_write.log 2>&1
ls -la $O/prof_bench $O/prof_pmc_fetch $O/prof_pmc_write
