#!/usr/bin/env python3
"""which torch (at::native) kernels are left in the training step, and who calls them: one profiled step of the bench workload, the
non-HIP-library kernels grouped by the innermost autoprog_amd / bench frame of their launching op"""
import os, sys, collections
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
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
        continue
    ks = [k for k in ev.kernels if not k.name.startswith(("k_", "void k_"))]
    if not ks:
        continue
    frame = next((f for f in (ev.stack or []) if ("autoprog_amd" in f or "bench.py" in f or "prof_glue" in f)), "(no repo frame)")
    key = (ev.name, str(ev.input_shapes)[:60], frame.strip()[-90:])
    agg[key][0] += len(ks); agg[key][1] += sum(k.duration for k in ks)
tot = sum(v[1] for v in agg.values())
print("torch kernels in one step: %d launches, %.1f us" % (sum(v[0] for v in agg.values()), tot))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%3d x %7.1f us  %-28s %-60s %s" % (v[0], v[1], k[0], k[1], k[2]))
