#!/bin/bash
# Runs on the GPU box (via gpurun): the bench line, the rocprofv3 kernel statistics of the same command
# and the two HBM counter passes (separate --pmc runs, as MI355X_MICROARCH.md prescribes).
# Outputs under gpurun_out/; `python tools/summarize_profiles.py rN` copies the summaries to profiles/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r3}
mkdir -p $O
cd $R
timeout 900 python bench.py > $O/bench_$TAG.json 2> $O/bench_$TAG.err
tail -c 600 $O/bench_$TAG.json
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_${TAG}_driver.json 2> $O/bench_${TAG}_driver.err
python -c "import json; d=json.load(open('$O/bench_${TAG}_driver.json')); print('driver command:', d['value'], d['ms_per_step'], d['resident_rank0']['value'], d['default_mode_rank0'], d['cpu_baseline']['value'])"
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 96 --warmup 8 --resident-steps 0 --no-cpu-baseline --no-default-mode --profiled-steps 2"
rm -rf $O/prof_bench $O/prof_pmc_fetch $O/prof_pmc_write
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- $CMD > $O/prof_bench.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_pmc_fetch -o pmc -- $CMD > $O/prof_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_pmc_write -o pmc -- $CMD > $O/prof_pmc_write.log 2>&1
# keep only what the summaries need (the per-dispatch trace is tens of MB)
find $O/prof_bench -name "*kernel_trace.csv" -delete
ls -la $O/prof_bench $O/prof_pmc_fetch $O/prof_pmc_write
