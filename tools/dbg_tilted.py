import sys, numpy as np
sys.path.insert(0,'.')
import plade_amd
from plade_amd.synth import make_pair, _faces, CONFIG4
n=int(sys.argv[1]) if len(sys.argv)>1 else 2000000
tg,sr,Tgt,tl,sl=make_pair(n,seed=0,return_labels=True,**CONFIG4)
ctx=plade_amd.Context(0,orient_normals=1)
coef,off,idx=ctx.extract_planes(tg,int(sys.argv[2]) if len(sys.argv)>2 else max(200,n//1000),max_planes=400)
print("planes",len(coef),"supports",sorted(np.diff(off).tolist(),reverse=True)[:12], "...", sorted(np.diff(off).tolist())[:8])
st=ctx.stats()
print({k:v for k,v in st.items() if k.startswith("ransac")})
lab_counts=np.bincount(tl[tl>=0])
found={}
for p in range(len(coef)):
    ids=idx[off[p]:off[p+1]]
    l=tl[ids]; l=l[l>=0]
    b=np.bincount(l,minlength=len(lab_counts))
    f=int(np.argmax(b)); found.setdefault(f,[]).append((int(b[f]),len(ids)))
missing=[f for f in range(len(lab_counts)) if f not in found]
print("faces",len(lab_counts),"found",len(found),"missing",missing[:40])
print("face sizes of missing:",[int(lab_counts[f]) for f in missing[:20]])
multi={f:v for f,v in found.items() if len(v)>1}
print("faces split over several planes:",len(multi), list(multi.items())[:6])
fs=_faces(1000,32,np.array([32.,28.,12.]),True)
for f in missing[:6]:
    o,eu,ev,nrm,w=fs[f]; print(f, "normal",np.round(nrm,2),"size",round(np.linalg.norm(eu),2),round(np.linalg.norm(ev),2))
print("---- composition of planes larger than 1.5 faces")
for p in range(len(coef)):
    ids=idx[off[p]:off[p+1]]
    if len(ids) < 1.5*lab_counts[6] or len(ids) > 5*lab_counts[6]: continue
    l=tl[ids]
    b=np.bincount(l[l>=0],minlength=len(lab_counts))
    top=np.argsort(-b)[:6]
    print(len(ids), "normal", np.round(coef[p,:3],2), [(int(f), int(b[f]), np.round(fs[f][3],2).tolist()) for f in top if b[f]>200], "outliers", int((l<0).sum()))
