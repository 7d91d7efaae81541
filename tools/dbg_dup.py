import sys, numpy as np
sys.path.insert(0,'.')
import plade_amd
from plade_amd.synth import make_pair, CONFIG4
n=int(sys.argv[1]); ms=int(sys.argv[2])
tg,sr,Tgt=make_pair(n,seed=0,**CONFIG4)
for hw in (0,1):
    ctx=plade_amd.Context(0,orient_normals=1,host_wait=hw)
    for rep in range(2):
        coef,off,idx=ctx.extract_planes(tg,ms,max_planes=400)
        u,c=np.unique(idx,return_counts=True)
        print("host_wait",hw,"rep",rep,"planes",len(coef),"sum",len(idx),"unique",len(u),"dups",int((c>1).sum()), "offsets monotone", bool(np.all(np.diff(off)>=0)))
        if (c>1).any():
            dup=set(u[c>1][:5].tolist())
            for p in range(len(coef)):
                ids=idx[off[p]:off[p+1]]
                if dup & set(ids[:2000000].tolist()): print("  plane",p,"size",len(ids),"coef",np.round(coef[p],4))
    ctx.close()
