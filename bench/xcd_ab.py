import os, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests'); sys.path.insert(0,'bench')
import torch, ctdirect_jl_amd as ct
from helpers import bench_inputs, describe
from stamps import CFGS
for name in ("cfg2_4M","cfg4_4M","cfg3_8M"):
    prob, sch, N = CFGS[name]
    row=[]
    for flag in ("0","1","0","1"):
        os.environ["CTD_XCD"]=flag
        d=ct.DOCP(prob,N,sch,device=0)
        x=torch.from_numpy(bench_inputs(describe(d,prob,sch),perturb=1e-3)).cuda()
        c=torch.zeros(d.dim_NLP_constraints,dtype=torch.float64,device="cuda"); v=torch.zeros(d.nnzj,dtype=torch.float64,device="cuda")
        ms=sorted(d.time_cons_jac(x,c,v,iters=12) for _ in range(3))[1]
        row.append(f"xcd={flag}: {ms*1e3:.1f}")
        d.close(); del x,c,v; torch.cuda.empty_cache()
    print(name," | ".join(row),flush=True)
