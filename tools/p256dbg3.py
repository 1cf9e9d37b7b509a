import sys, numpy as np, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch
import forge_ec_amd as F
import vectors as V
ctx=F.Context(0); ctx.set_timing(True)
curve=int(sys.argv[1]) if len(sys.argv)>1 else 1
for logn in (9,12,16,18,20):
    n=1<<logn
    k=V.scalars(n,curve,1); p=V.points(n,curve,2)
    dk=torch.from_numpy(k.view(np.int64)).cuda(); dp=torch.from_numpy(p.view(np.int64)).cuda(); do=torch.empty_like(dp)
    st=torch.cuda.current_stream().cuda_stream
    for rep in range(2):
        ctx.batch_mul_dev(curve, dk.data_ptr(), dp.data_ptr(), do.data_ptr(), n, st)
        ms,name=ctx.last_kernel_ms()
        o=do.cpu().numpy().view(np.uint32).reshape(n,-1)
        err=int((o[:,9]==0xDEADBEEF).sum())
        print("n=2^%d %s %.3f ms  err-elements %d"%(logn,name,ms,err),flush=True)
