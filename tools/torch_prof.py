"""torch.profiler attribution of the small ATen kernels of one training step: which Python lines launch them.
  python tools/torch_prof.py        (GPU box)"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
import bench
from autoprog_amd.models import create_model
from autoprog_amd.loss import TokenLabelCrossEntropy
from autoprog_amd.dist import GradientBucketReducer
from autoprog_amd.optim import FlatAdamWEma
torch.backends.cudnn.benchmark = True
torch.manual_seed(0); np.random.seed(0)
model = create_model("model_variant", variant="volo_h12_l18", drop_path_rate=0.1).cuda().train()
loss_fn = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=1000)
red = GradientBucketReducer(list(model.parameters()), world_size=1); red.install_sink(model)
opt = FlatAdamWEma(model, red, lr=1.6e-3, weight_decay=0.05, ema_decays=[0.998, 0.9986, 0.999, 0.9996])
B = 128
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 3, 224, 224, generator=g).cuda(); t = bench.make_target(B, 1000, 196, "cuda", g)
def step():
    red.zero_grad(); l = loss_fn(model(x), t); l.backward(); red.finish(); opt.step()
for _ in range(4): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=6)
rows = []
for e in ka:
    if e.key.startswith("aten::") and e.device_time_total > 0 and e.self_device_time_total > 0:
        stack = [s for s in e.stack if "/root/repo/" in s or "autoprog_amd" in s or "bench.py" in s]
        rows.append((e.self_device_time_total, e.count, e.key, stack[0].split("/root/repo/")[-1] if stack else (e.stack[0] if e.stack else "?")))
rows.sort(reverse=True)
tot = collections.Counter(); cnt = collections.Counter()
for us, n, key, where in rows:
    tot[(key, where)] += us; cnt[(key, where)] += n
print("%-28s %6s %9s  %s" % ("aten op", "calls", "gpu us", "first repo frame"))
for (key, where), us in tot.most_common(45):
    print("%-28s %6d %9.1f  %s" % (key, cnt[(key, where)], us, where[:110]))
print("total self GPU time of aten ops: %.1f us in %d launches" % (sum(tot.values()), sum(cnt.values())))
