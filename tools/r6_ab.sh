#!/bin/bash
# round 6 quick A/B on one box: selected GPU tests, the quick kernel profile, a short bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
if [ -n "${AB_TESTS:-}" ]; then timeout 1500 python -m pytest $AB_TESTS -x -q -m gpu 2>&1 | tail -8; fi
bash tools/prof_quick.sh > $O/r6_ab_prof.txt 2>&1
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 --no-cli --no-cpu-baseline --no-default-mode ${AB_BENCH:-} > $O/r6_ab_bench.json 2> $O/r6_ab_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6_ab_bench.json'))
print({k:d[k] for k in ('value','ms_per_step','steps','registrations_ok','results_bit_identical_to_the_pair_alone_rank0')}, 'latency', d['single_registration_latency_ms'])
print('closed', d['closed_form_mode_rank0'] and d['closed_form_mode_rank0']['value'], 'resident', d['resident_rank0'] and d['resident_rank0']['value'])
PY
head -${AB_LINES:-45} $O/r6_ab_prof.txt
