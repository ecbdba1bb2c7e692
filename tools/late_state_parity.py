"""Parity in the regime a trained network is in, not the one it is initialised in: the bench workload (VOLO-D1, one synthetic batch of
128 images, lr 1.6e-3) is trained for N steps -- residual stream up to ~1000, attention logits down to -100 -- and then the HIP model and
the CPU oracle (oracle/ref_cpu.py, fp64) evaluate the SAME weights on a small batch: outputs, loss and every parameter gradient.
   python tools/late_state_parity.py [steps]      (GPU box; the oracle leg is test infrastructure, as in tests/)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import ref_cpu as R
from autoprog_amd.models import create_model
from autoprog_amd.loss import TokenLabelCrossEntropy
from autoprog_amd.dist import GradientBucketReducer
from autoprog_amd.optim import FlatAdamWEma
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
variant = "volo_h12_l18"
torch.manual_seed(42); np.random.seed(42)
dev = torch.device("cuda:0")
model = create_model("model_variant", variant=variant, drop_path_rate=0.1).to(dev).train()
loss_fn = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=1000)
red = GradientBucketReducer(list(model.parameters()), world_size=1, defer_mean=True)
red.install_sink(model)
opt = FlatAdamWEma(model, red, lr=1.6e-3, weight_decay=0.05, ema_decays=[0.998])
gen = torch.Generator().manual_seed(42)
images = torch.randn(128, 3, 224, 224, generator=gen).to(dev)
target = bench.make_target(128, 1000, 196, dev, gen, sparse=True)
for i in range(steps):
    red.zero_grad()
    loss = loss_fn(model(images), target)
    loss.backward()
    red.finish()
    opt.step()
print("trained %d steps, loss %.4f" % (steps, float(loss)))
# ---- the same weights, a small batch, no DropPath, a fixed mix-token box: HIP vs oracle
B, r = 2, 224
model.set_drop_path_rate(0.0)
x = images[:B].contiguous()
dense = target.dense(1000)[:B].contiguous() if hasattr(target, "dense") else target[:B].contiguous()
red.zero_grad()
np.random.seed(3)
out = model(x)
loss = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0)(out, dense)
loss.backward()
red.finish()
p = {k: v.detach().double().cpu() for k, v in model.state_dict().items()}
for v in p.values():
    if v.dtype.is_floating_point:
        v.requires_grad_(True)
arch = R.variant_arch(variant)
lam, box = R.draw_mix_box((B, r // 8, r // 8, 192), 2, 1.0, np.random.RandomState(3))
assert tuple(out[2]) == tuple(box), (out[2], box)
torch.set_num_threads(min(os.cpu_count() or 1, 64))
ref = R.volo_forward(p, x.double().cpu(), train=True, mix=(lam, box), dp_masks={}, drop_path_rate=0.0, **arch)
ref_loss = R.token_label_ce(ref, dense.double().cpu(), 0.5, 1.0)
ref_loss.backward()
rel = lambda a, b: float((a.detach().double().cpu() - b.detach().double().cpu()).norm() / (b.detach().double().cpu().norm() + 1e-30))
print("logits: cls %.4f aux %.4f   |   loss HIP %.5f oracle %.5f" % (rel(out[0], ref[0]), rel(out[1], ref[1]), float(loss), float(ref_loss)))
errs = {n: rel(q.grad * red.grad_scale, p[n].grad) for n, q in model.named_parameters() if p[n].grad is not None and float(p[n].grad.norm()) > 1e-12}
worst = sorted(errs.items(), key=lambda kv: -kv[1])[:10]
print("gradient rel-L2 errors: median %.4f, max %.4f" % (float(np.median(list(errs.values()))), worst[0][1]))
for n, e in worst:
    print("   %-44s %.4f   |g| %.3e" % (n, e, float(p[n].grad.norm())))
ga = torch.cat([(q.grad * red.grad_scale).flatten().double().cpu() for n, q in model.named_parameters() if n in errs])
gb = torch.cat([p[n].grad.flatten() for n, q in model.named_parameters() if n in errs])
print("all gradients as one vector: rel-L2 %.4f, cosine %.6f" % (float((ga - gb).norm() / gb.norm()), float(torch.dot(ga, gb) / (ga.norm() * gb.norm()))))

# ---- yardstick: the oracle's own code on the GPU under bf16 autocast (the reference's AMP recipe: matmuls in bf16, softmax / LayerNorm /
# loss in fp32) against the same fp64 run: what does bf16 cost in THIS state, whatever computes it?
pa = {k: (v.detach().float().to(dev).requires_grad_(True) if v.dtype.is_floating_point else v.detach().to(dev)) for k, v in p.items()}
with torch.autocast("cuda", dtype=torch.bfloat16):
    ra = R.volo_forward(pa, x.float(), train=True, mix=(lam, box), dp_masks={}, drop_path_rate=0.0, **arch)
    la = R.token_label_ce(ra, dense.float(), 0.5, 1.0)
la.backward()
print("bf16-autocast oracle vs fp64 oracle: logits cls %.4f aux %.4f, loss %.5f" % (rel(ra[0], ref[0]), rel(ra[1], ref[1]), float(la)))
ea = {n: rel(pa[n].grad, p[n].grad) for n in errs}
wa = sorted(ea.items(), key=lambda kv: -kv[1])[:6]
print("  its gradient errors: median %.4f, max %.4f" % (float(np.median(list(ea.values()))), wa[0][1]))
for n, e in wa:
    print("   %-44s %.4f   (HIP: %.4f)" % (n, e, errs[n]))
gc = torch.cat([pa[n].grad.flatten().double().cpu() for n, q in model.named_parameters() if n in errs])
print("  all gradients as one vector: rel-L2 %.4f, cosine %.6f" % (float((gc - gb).norm() / gb.norm()), float(torch.dot(gc, gb) / (gc.norm() * gb.norm()))))
print("HIP vs bf16-autocast oracle, one vector: rel-L2 %.4f" % float((ga - gc).norm() / gc.norm()))
for n in ("network.2.12.attn.qkv.weight", "network.2.12.norm1.weight"):
    print("  %-40s HIP-vs-fp64 %.3f  autocast-vs-fp64 %.3f  HIP-vs-autocast %.3f" % (n, errs[n], ea[n], rel(dict(model.named_parameters())[n].grad * red.grad_scale, pa[n].grad)))
