#!/usr/bin/env python3
"""host time to ENQUEUE one training step of the bench workload against the time the GPU needs for it: how far the launching thread runs
ahead (a step whose enqueue time approaches its GPU time is launch-bound; RCCL launches at N > 1 come on top)"""
import os, sys, time
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
enq, tot, parts = [], [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    red.zero_grad(); out = model(images); ta = time.perf_counter()
    loss = loss_fn(out, target); tb = time.perf_counter()
    loss.backward(); tc = time.perf_counter()
    red.finish(); opt.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append(t1 - t0); tot.append(t2 - t0); parts.append((ta - t0, tb - ta, tc - tb, t1 - tc))
p = np.median(np.array(parts), 0) * 1e3
print("enqueue %.2f ms (forward %.2f, loss %.2f, backward %.2f, finish + optimizer %.2f)  |  step from an idle GPU %.2f ms" %
      (np.median(enq) * 1e3, p[0], p[1], p[2], p[3], np.median(tot) * 1e3))
