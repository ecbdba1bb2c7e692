"""Many optimizer steps of the bench workload (VOLO-D1, synthetic batch, lr 1.6e-3 as bench.py) with the loss printed every 25 steps:
does training on the one synthetic batch stay finite?  Environment switches select kernel paths.  tools/soak.py [steps] [lr]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from autoprog_amd.models import create_model
from autoprog_amd.loss import TokenLabelCrossEntropy
from autoprog_amd.dist import GradientBucketReducer
from autoprog_amd.optim import FlatAdamWEma
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1.6e-3
torch.manual_seed(42); np.random.seed(42)
dev = torch.device("cuda:0")
model = create_model("model_variant", variant="volo_h12_l18", drop_path_rate=0.1).to(dev).train()
loss_fn = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=1000)
red = GradientBucketReducer(list(model.parameters()), world_size=1, defer_mean=True)
red.install_sink(model)
opt = FlatAdamWEma(model, red, lr=lr, weight_decay=0.05, ema_decays=[0.998, 0.9986, 0.999, 0.9996])
gen = torch.Generator().manual_seed(42)
images = torch.randn(128, 3, 224, 224, generator=gen).to(dev)
target = bench.make_target(128, 1000, 196, dev, gen, sparse=True)
out = []
DBG = os.environ.get("SOAK_DEBUG") == "1"
if os.environ.get("SOAK_WATCH") == "1":
    from autoprog_amd import ops
    real_bwd, real_fwd = ops.mhsa_bwd, ops.mhsa_fwd
    def watch_bwd(qkv, o, do, lse, B, N, heads, scale):
        d = real_bwd(qkv, o, do, lse, B, N, heads, scale)
        if not bool(torch.isfinite(d.float()).all()):
            fin = lambda t: bool(torch.isfinite(t.float()).all())
            print("mhsa_bwd: non-finite dqkv; inputs finite: qkv %s o %s do %s lse %s | max |qkv| %.1f |do| %.3e lse [%.1f, %.1f]" % (
                fin(qkv), fin(o), fin(do), fin(lse), float(qkv.float().abs().max()), float(do.float().abs().max()), float(lse.min()), float(lse.max())))
            C = qkv.shape[1] // 3
            bad = (~torch.isfinite(d.float())).reshape(B, N, 3, heads, C // heads)
            idx = bad.nonzero()
            print("bad entries:", int(bad.sum()), "first:", idx[:5].tolist(), "images:", sorted(set(idx[:, 0].tolist()))[:10], "heads:", sorted(set(idx[:, 3].tolist())), "parts:", sorted(set(idx[:, 2].tolist())))
            q = qkv.float().reshape(B, N, 3, heads, C // heads)
            b0, h0 = int(idx[0, 0]), int(idx[0, 3])
            S = (q[b0, :, 0, h0] @ q[b0, :, 1, h0].t()) * scale
            print("that head: max |S| %.1f, lse fwd [%.2f, %.2f], lse recomputed [%.2f, %.2f]" % (float(S.abs().max()), float(lse[b0, h0].min()), float(lse[b0, h0].max()),
                  float(torch.logsumexp(S, -1).min()), float(torch.logsumexp(S, -1).max())))
            torch.save({"qkv": qkv.cpu(), "o": o.cpu(), "do": do.cpu(), "lse": lse.cpu(), "cfg": (B, N, heads, scale)}, "gpurun_out/mhsa_bad.pt")
            raise SystemExit(0)
        return d
    ops.mhsa_bwd = watch_bwd
for i in range(steps):
    red.zero_grad()
    loss = loss_fn(model(images), target)
    loss.backward()
    red.finish()
    if DBG and (not bool(torch.isfinite(loss)) or not bool(torch.isfinite(red.flat).all())):
        print("step", i, "loss", float(loss), "finite grads:", bool(torch.isfinite(red.flat).all()))
        bad = [(n, float(p.detach().abs().max())) for n, p in model.named_parameters() if not bool(torch.isfinite(p).all())]
        print("non-finite params:", bad[:10])
        big = sorted(((float(p.detach().abs().max()), n) for n, p in model.named_parameters()), reverse=True)[:8]
        print("largest params:", big)
        badg = [n for n, p in model.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
        print("non-finite grads in %d tensors, first:" % len(badg), badg[:6], "last:", badg[-14:])
        okg = [n for n, p in model.named_parameters() if p.grad is not None and bool(torch.isfinite(p.grad).all())]
        print("finite grads in:", okg[:4], "...", okg[-14:])
        print("largest finite grad tensors:", sorted(((float(p.grad.abs().max()), n) for n, p in model.named_parameters() if p.grad is not None and bool(torch.isfinite(p.grad).all())), reverse=True)[:6])
        hooks, seen = [], []
        def mk(name):
            def hook(mod, inp, outp):
                t = outp[0] if isinstance(outp, (tuple, list)) else outp
                if torch.is_tensor(t):
                    seen.append((name, bool(torch.isfinite(t).all()), float(t.detach().float().abs().max())))
            return hook
        for n, m in model.named_modules():
            if n and n.count(".") <= 2:
                hooks.append(m.register_forward_hook(mk(n)))
        with torch.no_grad():
            model(images)
        first = next((x for x in seen if not x[1]), None)
        print("first non-finite module output:", first)
        print("max |out| along the way:", [(n, round(v, 1)) for n, ok, v in seen if v > 200][:20])
        break
    opt.step()
    if i % 25 == 0 or i == steps - 1:
        gn = float(red.flat.norm()) * red.grad_scale
        out.append("%d:%.4f(g%.2e,%.2fGB)" % (i, float(loss), gn, torch.cuda.memory_allocated() / 2 ** 30))
print(" ".join(out))
# every parameter after the last step, as one hash: two runs that print the same hash took the same steps bit for bit (tools/soak_fused_pair.sh)
import hashlib
hsh = hashlib.sha256()
for n_, p_ in sorted(model.named_parameters()):
    hsh.update(p_.detach().float().cpu().numpy().tobytes())
print("parameters sha256 %s" % hsh.hexdigest()[:32])
print("peak allocated %.2f GB, reserved %.2f GB" % (torch.cuda.max_memory_allocated() / 2 ** 30, torch.cuda.memory_reserved() / 2 ** 30))
