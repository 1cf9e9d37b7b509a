import sys, numpy as np, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import forge_ec_amd as F
from oracle import c_oracle
import vectors as V
ctx=F.Context(0)
curve=1
n=1<<12
k=V.scalars(n,curve,2004); p=V.points(n,curve,2005)
want=c_oracle.batch_mul(curve,k,p,nthreads=16)
for rep in range(6):
    t=time.time(); got=ctx.batch_mul(curve,k,p); dt=time.time()-t
    bad=np.where((got!=want).any(axis=1))[0]
    print("rep",rep,"time %.2fs"%dt,"mismatches",len(bad), sorted(set(bad//512)), flush=True)
    if len(bad):
        g=got.view(np.uint32).reshape(n,24)
        for wg in sorted(set(bad//512))[:3]:
            e0=wg*512
            print(" WG",wg,"ctl",[int(x) for x in g[e0,:8]], "marker",hex(int(g[e0,9])), "steps min/max", int(g[e0:e0+512,8].min()), int(g[e0:e0+512,8].max()), "hist of unfinished:", int((g[e0:e0+512,8]<256).sum()))
