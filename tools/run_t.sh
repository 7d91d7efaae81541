cd $GRAFT_REPO_ROOT
for L in 6 7; do
PLADE_SPACING_LEVEL=$L python tools/exp_throughput.py 384 8 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.pop('stats'); print('level $L', round(d['reg_per_s'],1), d['ok'], d['identical'], 'busy', round(d['busy_threads'],2))"
done
