#!/bin/bash
# Runs on the GPU box (via gpurun): the bench line, the rocprofv3 kernel statistics of the same command
# and the two HBM counter passes (separate --pmc runs, as MI355X_MICROARCH.md prescribes).
# Outputs under gpurun_out/; `python tools/summarize_profiles.py rN` copies the summaries to profiles/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r5}
mkdir -p $O
cd $R
if [ -z "${ONLY_PROF:-}" ]; then
timeout 900 python bench.py > $O/bench_$TAG.json 2> $O/bench_$TAG.err
tail -c 600 $O/bench_$TAG.json
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_${TAG}_driver.json 2> $O/bench_${TAG}_driver.err
python -c "import json; d=json.load(open('$O/bench_${TAG}_driver.json')); print('driver command:', d['value'], d['ms_per_step'], d['resident_rank0']['value'], d['default_mode_rank0'], d['cpu_baseline']['value'])"
fi
cd /tmp && export TMPDIR=/tmp
# (lead-in of one round instead of eight: with ~170 registrations in one traced process rocprofv3 7.2 segfaults inside the
#  profiled process, below hipGraphLaunch; ~110 are fine)
export BENCH_LEAD_ROUNDS=1
export BENCH_MIN_ROUNDS=${BENCH_MIN_ROUNDS:-6}   # 6 rounds of 32 in flight = 192 timed steps (+ lead-in, warm-up, tail)
CMD="python $R/bench.py --pairs 16 --steps 96 --warmup 8 --resident-steps 0 --closed-form-steps 0 --no-cpu-baseline --no-cli --no-default-mode --no-parity --profiled-steps 0"
rm -rf $O/prof_bench $O/prof_pmc_fetch $O/prof_pmc_write
# every pass is tried up to three times
prof() {   # prof <dir> <output name> <rocprofv3 options...>
    local dir=$1 name=$2; shift 2
    for attempt in 1 2 3; do
        rm -rf $dir
        timeout 600 rocprofv3 "$@" --output-format csv -d $dir -o $name -- $CMD > $dir.log 2>&1
        if ls $dir 2> /dev/null | grep -qE "kernel_stats|counter_collection" && ! grep -q SIGSEGV $dir.log; then return 0; fi
        echo "rocprofv3 pass $dir failed (attempt $attempt)"
    done
    return 1
}
prof $O/prof_bench bench --kernel-trace --stats
prof $O/prof_pmc_fetch pmc --pmc FETCH_SIZE
prof $O/prof_pmc_write pmc --pmc WRITE_SIZE
# keep only what the summaries need (the per-dispatch trace is tens of MB)
find $O/prof_bench -name "*kernel_trace.csv" -delete
ls -la $O/prof_bench $O/prof_pmc_fetch $O/prof_pmc_write
