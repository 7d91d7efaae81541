#!/bin/bash
# driver-style runs (20 steps) with different lead-ins and one default run: value vs window_value
mkdir -p gpurun_out
for k in 1 2 4 8 8 8; do BENCH_LEAD_ROUNDS=$k python bench.py --gpus 1 --steps 20 --warmup 3 --no-default-mode > gpurun_out/b20_$k.json 2> gpurun_out/b20_$k.err
python - $k <<'PY'
import json,sys
f=f"b20_{sys.argv[1]}"
d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
print(f, round(d["value"],1), round(d["pipeline"]["window_value_rank0"],1), round(d["resident_rank0"]["value"],1), round(d["host_buffers_rank0"]["bracketed_value"],1))
PY
done
