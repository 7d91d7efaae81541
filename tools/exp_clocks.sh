#!/bin/bash
# clocks and power while the batch pipeline runs (and while one group runs alone): is the part throttled under the pipeline's load?
O=gpurun_out
( for i in $(seq 1 120); do /opt/rocm/bin/rocm-smi --showclocks --showpower --showuse --json 2>/dev/null | tr -d '\n'; echo; sleep 0.25; done ) > $O/smi_pipeline.txt &
SMI=$!
python tools/exp_groups.py 2048 4 8 1 > $O/clk_groups.json 2>/dev/null
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
python - <<'PY'
import json,re
rows=[]
for l in open('gpurun_out/smi_pipeline.txt'):
    try: d=json.loads(l)
    except Exception: continue
    c=d.get('card0',{})
    rows.append({k:v for k,v in c.items() if any(t in k.lower() for t in ('sclk','mclk','power','use','fclk'))})
print(len(rows),'samples')
for r in rows[::6][:24]: print(r)
print(json.load(open('gpurun_out/clk_groups.json'))['reg_per_s'])
PY
