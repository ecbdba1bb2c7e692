#!/usr/bin/env python3
"""Where does the realistic-init loss curve (tests/golden/step_curve_init.npz) lose precision?  Runs the 10 AdamW steps of the
GPU test with one part of the network switched to fp32 at a time and prints |loss - reference(fp64)| per step and the first-step
gradient errors of the sampled tensors.  GPU box only:  python tools/curve_bisect.py [bf16|stem32] [repeat]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests._golden import load  # noqa: E402
from tests._initweights import init_state_dict  # noqa: E402


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def run(mode):
    from autoprog_amd.models import create_model
    from autoprog_amd.loss import TokenLabelCrossEntropy
    d = load("step_curve_init")
    classes = int(d["classes"])
    model = create_model("model_variant", variant="volo_h4_l6", num_classes=classes, img_size=64, stem_hidden_dim=16)
    model.load_state_dict(init_state_dict(model.state_dict(), int(d["init_seed"])), strict=True)
    model = model.cuda().train()
    if mode == "stem32":
        model.patch_embed.compute_dtype = torch.float32
    if mode in ("oldcls", "oldboth"):                     # the concatenating class-block path (pre-ClassBlockFn)
        from autoprog_amd import functional as AF
        from autoprog_amd.models import volo as V

        def block_cls(self, xx):
            cls = xx[:, :1] + self.attn(AF.layer_norm(xx, self.norm1.weight, self.norm1.bias, self.norm1.eps))
            return cls + self.mlp(AF.layer_norm(cls, self.norm2.weight, self.norm2.bias, self.norm2.eps))

        def forward_cls(self, xx):
            B = xx.shape[0]
            cls = self.cls_token.expand(B, -1, -1).to(torch.bfloat16)
            for block in self.post_network:
                cls = block_cls(block, torch.cat([cls, xx], dim=1))
            return cls, xx
        V.VOLO.forward_cls = forward_cls
    x = torch.from_numpy(d["x"]).cuda()
    target = torch.from_numpy(d["target"]).cuda()
    decay, no_decay = [], []
    for n, p in model.named_parameters():
        (no_decay if (p.dim() == 1 or n.endswith(".bias") or n in ("pos_embed", "cls_token")) else decay).append(p)
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": float(d["wd"])}, {"params": no_decay, "weight_decay": 0.0}], lr=float(d["lr"]))
    loss_fn = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=classes)
    if mode in ("oldloss", "oldboth"):                    # the unfused loss path (pre-TokenLabelCEFn)
        class Unfused(TokenLabelCrossEntropy):
            def _adjust_cls(self, *a, **k):
                return super()._adjust_cls(*a, **k)
        loss_fn = Unfused(dense_weight=0.5, cls_weight=1.0, classes=classes)
    np.random.seed(int(d["np_seed"]))
    losses, errs = [], {}
    for step in range(10):
        loss = loss_fn(model(x), target)
        opt.zero_grad()
        loss.backward()
        if step == 0:
            named = dict(model.named_parameters())
            errs = {k[3:]: rel(named[k[3:]].grad, v) for k, v in d.items() if k.startswith("g0.")}
        opt.step()
        losses.append(float(loss.detach()))
    dev = np.abs(np.array(losses) - d["losses"])
    print("%-7s max|dloss| %.5f  per step %s" % (mode, dev.max(), " ".join("%.5f" % v for v in dev)))
    print("        grad errs: " + "  ".join("%s %.4f" % (k.replace("patch_embed.", "pe.").replace("network.", "n."), v) for k, v in sorted(errs.items(), key=lambda kv: -kv[1])[:8]))


if __name__ == "__main__":
    modes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["bf16", "stem32"]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    for m in modes:
        for _ in range(reps):
            run(m)
