#!/bin/bash
# round 6 baseline: quick kernel profile + bench line with the reference's solver arithmetic as the default mode
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
bash tools/prof_quick.sh > $O/r6_base_prof.txt 2>&1
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 --no-cli > $O/r6_base_bench.json 2> $O/r6_base_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6_base_bench.json'))
print({k:d[k] for k in ('value','ms_per_step','steps','registrations_ok','results_bit_identical_to_the_pair_alone_rank0','value_at_requested_steps')})
print('closed', d['closed_form_mode_rank0'])
print('parity', json.dumps(d['parity_vs_reference_solver'])[:1500])
PY
head -40 $O/r6_base_prof.txt
