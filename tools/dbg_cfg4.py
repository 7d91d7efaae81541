import sys, numpy as np, time
sys.path.insert(0,'.')
import plade_amd
from plade_amd.synth import make_pair, planes_from_labels, CONFIG4
n=int(sys.argv[1]) if len(sys.argv)>1 else 2000000
tg,sr,Tgt,tl,sl=make_pair(n,seed=0,return_labels=True,**CONFIG4)
tp,sp=planes_from_labels(tg,tl,min_points=n//1000),planes_from_labels(sr,sl,min_points=n//1000)
print("gt planes",len(tp[0]),len(sp[0]))
ctx=plade_amd.Context(0,orient_normals=1,max_planes=100,max_candidates=10000,dump=1)
if len(sys.argv)>2: t0=time.time(); ok,T=ctx.registration_planes(tg,sr,tp,sp); print("planes-given ok",ok,"err",np.linalg.norm(T-Tgt),"sec",time.time()-t0)
t0=time.time(); ok,T=ctx.registration(tg,sr); print("full ok",ok,"err",np.linalg.norm(T-Tgt),"sec",time.time()-t0)
st=ctx.stats(); print({k:st[k] for k in st if k.startswith("n_")})
d=ctx.dump(); print("pen flags", np.bincount(d["pen_flags"]) if len(d["pen_flags"]) else None, "plane_match_counts top", sorted(d["plane_match_counts"].tolist(),reverse=True)[:10])

pm=d["plane_match_counts"]; order=np.argsort(-pm,kind="stable")[:5]
print("top candidates (index in tested list?)", pm[order], "flags of first 10 tested", d["pen_flags"][:10], "tested ids", d["pen_tested"][:10])
c=d["candidates"].reshape(-1,4,4) if len(d.get("candidates",[])) else None
rt=d["initial_RT"].reshape(-1,12); seeds=d["cluster_seeds"]
# the candidates tested, in order: error vs ground truth of the first 10
import itertools
sizes=d["cluster_sizes"]; 
print("clusters",len(seeds),"largest",sorted(sizes.tolist(),reverse=True)[:5])
