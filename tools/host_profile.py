#!/usr/bin/env python3
"""cProfile of the launching thread over a few training steps of the bench workload (where the ~10 ms of host time per step go)"""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from autoprog_amd.models import create_model
from autoprog_amd.loss import TokenLabelCrossEntropy
from autoprog_amd.dist import GradientBucketReducer
from autoprog_amd.optim import FlatAdamWEma
torch.manual_seed(42); np.random.seed(42)
dev = torch.device("cuda:0")
model = create_model("model_variant", variant="volo_h12_l18", drop_path_rate=0.1).to(dev).train()
loss_fn = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=1000)
red = GradientBucketReducer(list(model.parameters()), world_size=1, defer_mean=True); red.install_sink(model)
opt = FlatAdamWEma(model, red, lr=1.6e-3, weight_decay=0.05, ema_decays=[0.998, 0.9986, 0.999, 0.9996])
gen = torch.Generator().manual_seed(42)
images = torch.randn(128, 3, 224, 224, generator=gen).to(dev)
target = bench.make_target(128, 1000, 196, dev, gen, sparse=True)
def step():
    red.zero_grad(); loss = loss_fn(model(images), target); loss.backward(); red.finish(); opt.step(); return loss
for _ in range(5): step()
torch.cuda.synchronize()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5
pr = cProfile.Profile()
with torch.autograd.set_multithreading_enabled(False):        # the backward Functions then run on THIS thread, where cProfile sees them
    step(); torch.cuda.synchronize()
    pr.enable()
    for _ in range(N):
        step()
        torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(40)
st.sort_stats("cumtime").print_stats(40)
