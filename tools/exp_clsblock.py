#!/usr/bin/env python3
"""experiment: ClassBlockFn (split) vs the concatenating composite vs an fp64 torch restatement, on the same inputs"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from autoprog_amd import functional as AF
from autoprog_amd.models.volo import ClassBlock
BF16 = torch.bfloat16
def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))
def ref64(blk, cls, tok):
    p = {k: v.detach().double().requires_grad_(True) for k, v in blk.named_parameters()}
    cls = cls.double().requires_grad_(True); tok = tok.double().requires_grad_(True)
    x = torch.cat([cls.unsqueeze(1), tok], 1)
    C = x.shape[-1]; H = blk.attn.num_heads
    n = F.layer_norm(x, (C,), p["norm1.weight"], p["norm1.bias"], blk.norm1.eps)
    kv = F.linear(n, p["attn.kv.weight"], p.get("attn.kv.bias")); B, N, _ = kv.shape
    hd = kv.shape[-1] // 2 // H
    kv = kv.reshape(B, N, 2, H, hd).permute(2, 0, 3, 1, 4); k, v = kv[0], kv[1]
    q = F.linear(n[:, :1], p["attn.q.weight"], p.get("attn.q.bias")).reshape(B, H, 1, hd)
    a = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(-1)
    o = (a @ v).transpose(1, 2).reshape(B, 1, H * hd)
    c = cls.unsqueeze(1) + F.linear(o, p["attn.proj.weight"], p["attn.proj.bias"])
    m = F.layer_norm(c, (C,), p["norm2.weight"], p["norm2.bias"], blk.norm2.eps)
    c = c + F.linear(F.gelu(F.linear(m, p["mlp.fc1.weight"], p["mlp.fc1.bias"])), p["mlp.fc2.weight"], p["mlp.fc2.bias"])
    return c[:, 0], cls, tok, p
def old(blk, cls, tok):
    x = torch.cat([cls.unsqueeze(1), tok], 1)
    c = x[:, :1] + blk.attn(AF.layer_norm(x, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps))
    return (c + blk.mlp(AF.layer_norm(c, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)))[:, 0]
torch.manual_seed(0)
for (B, N, C, H) in [(8, 16, 128, 4), (8, 16, 256, 8), (32, 196, 384, 12)]:
    blk = ClassBlock(C, H, mlp_ratio=3.0).cuda()
    for p in blk.parameters():
        if p.dim() > 1: torch.nn.init.trunc_normal_(p, std=.02)
    cls = (torch.randn(B, C, device="cuda") * 0.5).to(BF16); tok = torch.randn(B, N, C, device="cuda").to(BF16)
    g = torch.randn(B, C, device="cuda").to(BF16)
    y64, c64, t64, p64 = ref64(blk, cls, tok); y64.backward(g.double())
    for name, fn in (("split", lambda c, t: blk.forward_split(c, t)), ("concat", lambda c, t: old(blk, c, t))):
        blk.zero_grad()
        c = cls.clone().requires_grad_(True); t = tok.clone().requires_grad_(True)
        y = fn(c, t); y.backward(g)
        errs = {k: rel(v.grad, p64[k].grad) for k, v in blk.named_parameters()}
        print("%s B%d N%d C%d: y %.2e dcls %.2e dtok %.2e | " % (name, B, N, C, rel(y, y64), rel(c.grad, c64.grad), rel(t.grad, t64.grad))
              + " ".join("%s %.1e" % (k.replace("weight", "w").replace("bias", "b"), e) for k, e in errs.items()))
