for rep in 1 2; do
for v in "base X=1" "wgs256 PLADE_EXP_OV_WGS=256" "wgs1024 PLADE_EXP_OV_WGS=1024" "wgs2048 PLADE_EXP_OV_WGS=2048" "ch4 PLADE_EXP_OV_CH=4" "ch8 PLADE_EXP_OV_CH=8" "ch4w256 PLADE_EXP_OV_CH=4 PLADE_EXP_OV_WGS=256"; do
  set -- $v; name=$1; shift
  env "$@" timeout 600 python tools/exp_groups.py 1536 4 8 1 > gpurun_out/env_$name.json 2>/dev/null
  python -c "
import json
d=json.load(open('gpurun_out/env_$name.json'))
print('$name', round(d['reg_per_s'],1), d['identical_to_single'], round(d['cpu_ms_per_registration'],2), round(d['busy_threads'],2))"
done; done
