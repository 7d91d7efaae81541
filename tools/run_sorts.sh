cd $GRAFT_REPO_ROOT
PLADE_TRACE_SORT=1 python - 2>&1 <<'PY' | grep "\[sort\]" | sort | uniq -c
import sys
sys.path.insert(0,'.')
import plade_amd
from plade_amd.synth import make_pair
tg,sr,_=make_pair(1000000,seed=0)
c=plade_amd.Context(0,orient_normals=1)
c.registration(tg,sr)
print("----", file=sys.stderr)
PY
