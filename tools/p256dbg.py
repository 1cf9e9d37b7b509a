import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import forge_ec_amd as F
from oracle import c_oracle
import vectors as V
ctx=F.Context(0)
curve=1
rng=np.random.default_rng(5)
m=3000
a=rng.integers(0,2**64,size=(m,4),dtype=np.uint64); b=rng.integers(0,2**64,size=(m,4),dtype=np.uint64)
a[::7,3]=0xFFFFFFFFFFFFFFFF; b[::11,3]=0xFFFFFFFF00000000
for op,name in ((2,'mul'),(3,'sqr'),(0,'add'),(1,'sub')):
    g=ctx.field_op(curve,op,a,b if op!=3 else None)
    w=np.array([c_oracle.field_op(curve,name,a[i],b[i]) for i in range(m)])
    print(name,"mismatches",int((g!=w).any(axis=1).sum()),flush=True)
n=1<<13
k=V.scalars(n,curve,2004); p=V.points(n,curve,2005); q=V.points(n,curve,77)
g=ctx.point_op(curve,0,p[:m],q[:m]); w=np.array([c_oracle.point_add(curve,p[i],q[i]) for i in range(m)])
print("padd mismatches",int((g!=w).any(axis=1).sum()),flush=True)
g=ctx.point_op(curve,1,p[:m]); w=np.array([c_oracle.point_double(curve,p[i]) for i in range(m)])
print("pdouble mismatches",int((g!=w).any(axis=1).sum()),flush=True)
for nn in (1,63,64,65,511,512,513,1024,1536,n):
    got=ctx.batch_mul(curve,k[:nn],p[:nn]); want=c_oracle.batch_mul(curve,k[:nn],p[:nn],nthreads=16)
    bad=np.where((got!=want).any(axis=1))[0]
    print("batch_mul n=%d mismatches: %d"%(nn,len(bad)), bad[:8], bad[-3:] if len(bad) else "",flush=True)
gen=ctx.generator(curve)
got=ctx.batch_mul_fixed(curve,k[:1000],gen); want=c_oracle.batch_mul_fixed(curve,k[:1000],gen,nthreads=16)
print("fixed mismatches", int((got!=want).any(axis=1).sum()))
