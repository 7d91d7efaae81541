cd $GRAFT_REPO_ROOT
PLADE_DBG_CC=1 python - <<'PY'
import sys, threading, time
sys.path.insert(0, '.')
import numpy as np, plade_amd
from plade_amd.synth import make_pair
pairs=[make_pair(1000000, seed=s) for s in range(2)]
M=8
ctxs=[plade_amd.Context(0, orient_normals=1, host_wait=1) for _ in range(M)]
cl=[[(c.upload(t), c.upload(s)) for t,s,_ in pairs] for c in ctxs]
acc={}
lock=threading.Lock()
def work(w):
    for i in range(24):
        ctxs[w].registration_dev(*cl[w][i%2])
        if i>=4:
            st=ctxs[w].stats()
            with lock:
                for k,v in st.items():
                    if k.startswith('dbg_cc_'): acc[k]=acc.get(k,0)+v
ths=[threading.Thread(target=work,args=(w,)) for w in range(M)]
[t.start() for t in ths]; [t.join() for t in ths]
n=acc['dbg_cc_7']
names=['header','bitmap load','label','select rows','collect','fold(fit)']
for q in range(6): print(f"{names[q]:14s} {acc['dbg_cc_%d'%q]/n/100:.2f} us")


print("samples",n,'avg npx',acc['dbg_cc_8']/n,'avg rows',acc['dbg_cc_9']/n)
PY
