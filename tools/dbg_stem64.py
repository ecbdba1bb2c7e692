"""stem64 fixture: gradient error vs the reference (fp64) of (a) the HIP convolution stem, (b) the MIOpen bf16 stem, (c) the fp32 stem."""
import numpy as np, torch, sys
sys.path.insert(0, ".")
from tests._golden import load, sub
from autoprog_amd.models import volo as V

def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu(); b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))

d = load("stem64")
def run(hip, dtype):
    pe = V.PatchEmbed(stem_conv=True, stem_stride=2, patch_size=8, in_chans=3, hidden_dim=64, embed_dim=32)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in sub(d, "train.w").items()}
    for k in sd:
        if k.endswith("running_mean"): sd[k] = torch.zeros_like(sd[k])
        elif k.endswith("running_var"): sd[k] = torch.ones_like(sd[k])
        elif k.endswith("num_batches_tracked"): sd[k] = torch.zeros_like(sd[k])
    pe.load_state_dict(sd, strict=True)
    pe = pe.cuda().train(); pe.hip_conv = hip; pe.compute_dtype = dtype
    x = torch.from_numpy(d["train.x"]).cuda()
    y = pe(x)
    y.backward(torch.from_numpy(d["train.dy"]).cuda().to(y.dtype))
    errs = {n: round(rel(p.grad, d["train.g." + n]), 4) for n, p in pe.named_parameters()}
    return round(rel(y, d["train.y"]), 5), errs
for name, hip, dt in (("hip bf16", True, torch.bfloat16), ("miopen bf16", False, torch.bfloat16), ("fp32", False, torch.float32)):
    print(name, run(hip, dt))
