import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests._golden import load, sub
from tests.test_gpu_model import build, load_sd, rel
from autoprog_amd.loss import TokenLabelCrossEntropy
d = load("volo_full"); tag="h2_l3"
for trial in range(2):
    model = load_sd(build("volo_h2_l3", 16), d, tag).cuda().train()
    if trial == 1: model.patch_embed.compute_dtype = torch.float32
    x = torch.from_numpy(d[tag + ".x"]).cuda(); target = torch.from_numpy(d[tag + ".target"]).cuda()
    np.random.seed(int(d[tag + ".np_seed"]))
    out = model(x)
    loss = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=16)(out, target)
    loss.backward()
    errs = {n: rel(p.grad, d[tag + ".g." + n]) for n, p in model.named_parameters()}
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
    print(os.environ.get("AP_GEMM_SMALL_TILES"), "loss", float(loss.detach()), float(d[tag+".loss"]), "xcls", rel(out[0], d[tag+".x_cls"]), [(k, round(v,3)) for k,v in top])
