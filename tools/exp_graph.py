#!/usr/bin/env python3
"""experiment: capture the whole bench step in a HIP graph (semantics of host scalars ignored) to size the launch overhead"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from autoprog_amd.models import create_model
from autoprog_amd.loss import TokenLabelCrossEntropy
from autoprog_amd.dist import GradientBucketReducer
from autoprog_amd.optim import FlatAdamWEma
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda", 0)
torch.manual_seed(42); np.random.seed(42)
VAR = sys.argv[1] if len(sys.argv) > 1 else "volo_h12_l18"
RES = int(sys.argv[2]) if len(sys.argv) > 2 else 224
model = create_model("model_variant", variant=VAR, drop_path_rate=0.1, img_size=RES).to(dev).train()
loss_fn = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=1000)
reducer = GradientBucketReducer(list(model.parameters()), world_size=1); reducer.install_sink(model)
opt = FlatAdamWEma(model, reducer, lr=1.6e-3, weight_decay=0.05, ema_decays=[0.998, 0.9986, 0.999, 0.9996])
gen = torch.Generator().manual_seed(42)
images = torch.randn(128, 3, RES, RES, generator=gen).to(dev)
target = bench.make_target(128, 1000, (RES // 16) ** 2, dev, gen, sparse=True)
def step():
    reducer.zero_grad()
    loss = loss_fn(model(images), target)
    loss.backward()
    reducer.finish()
    opt.step()
    return loss
def timeit(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(5): step()
torch.cuda.current_stream().wait_stream(s)
print(VAR, RES, "eager ms/step: %.3f" % timeit(step))
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        static_loss = step()
    print("captured")
    print("graph ms/step: %.3f  (loss %.4f)" % (timeit(g.replay), float(static_loss)))
except Exception as e:
    import traceback; traceback.print_exc()
