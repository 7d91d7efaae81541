cd $GRAFT_REPO_ROOT
make cli >/dev/null 2>&1
python - <<PY
import os,sys,subprocess,tempfile,time
sys.path.insert(0,os.getcwd())
from plade_amd.plyio import write_ply
from plade_amd.synth import make_pair
d=tempfile.mkdtemp()
files=[]
for s in range(2):
    tg,sr,_=make_pair(1000000,seed=s); pt,ps=f"{d}/t{s}.ply",f"{d}/s{s}.ply"; write_ply(pt,tg); write_ply(ps,sr); files.append((pt,ps))
with open(f"{d}/pairs.txt","w") as f:
    for i in range(64): f.write(f"{files[i%2][0]}\n{files[i%2][1]}\n")
for infl in (1,4):
    env=dict(os.environ,PLADE_INFLIGHT=str(infl),PLADE_GPUS="1",PLADE_DEBUG_ALLOC="1",PLADE_ORIENT_NORMALS="1")
    for rep in range(2):
        t0=time.perf_counter(); r=subprocess.run(["plade_amd/PLADE",f"{d}/pairs.txt",f"{d}/out.txt"],capture_output=True,text=True,env=env); dt=time.perf_counter()-t0
    print(infl, round(dt,3), r.stderr.strip().splitlines()[-3:])
PY
