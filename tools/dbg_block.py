import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import ref_cpu as R
from autoprog_amd.models import volo as V
def rel(a,b):
    a=torch.as_tensor(a).detach().double().cpu(); b=torch.as_tensor(b).detach().double().cpu()
    return float((a-b).norm()/(b.norm()+1e-30))
torch.manual_seed(0)
for (C,H,B,G) in [(32,1,2,8),(64,2,2,8),(192,6,2,28)]:
    blk = V.Outlooker(C,3,1,stride=2,num_heads=H,mlp_ratio=3.0)
    with torch.no_grad():
        for n,p in blk.named_parameters():
            if p.dim()>=2: p.copy_(torch.randn(p.shape)*(1.0/p.shape[-1]**0.5))
            elif n.endswith("weight"): p.copy_(1+0.2*torch.randn(p.shape))
            else: p.copy_(0.2*torch.randn(p.shape))
    blk=blk.cuda().train()
    x=torch.randn(B,G,G,C).bfloat16(); dy=torch.randn(B,G,G,C).bfloat16()
    xg=x.cuda().requires_grad_(True); y=blk(xg); y.backward(dy.cuda())
    p={k:v.detach().double().cpu().requires_grad_(True) for k,v in blk.state_dict().items()}
    xr=x.double().requires_grad_(True); yr=R.outlooker(xr,p,"",H); yr.backward(dy.double())
    errs={n:rel(q.grad,p[n].grad) for n,q in blk.named_parameters()}
    print(C,H,"y",round(rel(y,yr),4),"dx",round(rel(xg.grad,xr.grad),4),{k:round(v,4) for k,v in errs.items()})
# mix swap backward
from autoprog_amd import functional as AF
x=torch.randn(4,8,8,32).bfloat16().cuda().requires_grad_(True)
y=AF.MixSwapFn.apply(x,2,6,0,4); g=torch.randn_like(y); y.backward(g)
xr=x.detach().float().cpu().requires_grad_(True); yr=R.mix_token_swap(xr,(1,0,3,2),2); yr.backward(g.float().cpu())
print("mixswap fwd", rel(y,yr), "bwd", rel(x.grad, xr.grad))
