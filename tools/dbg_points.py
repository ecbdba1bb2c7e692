#!/usr/bin/env python3
"""where the HIP network leaves the oracle's rounding-matched network (oracle/ref_cpu.py *_bf16_points): stage by stage, forward only"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import ref_cpu as R
from autoprog_amd.models import create_model
from autoprog_amd import functional as AF
def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))
torch.manual_seed(5)
classes, B, r = 40, 4, 64
model = create_model("model_variant", variant="volo_h4_l6", num_classes=classes, img_size=r).cuda().train()
x = torch.randn(B, 3, r, r, device="cuda")
p = {k: v.detach().double().cpu().clone() for k, v in model.state_dict().items()}
arch = R.variant_arch("volo_h4_l6")
rb, rw, rf = R._RoundBoth.apply, R._RoundOperand.apply, R._RoundFwd.apply
with torch.no_grad():
    t = model.forward_embeddings(x)                                   # [B,H,W,C]
    o = R.patch_embed(x.double().cpu(), p, True, 8, bf16_points=True)
    print("patch_embed", rel(t, o), t.shape)
    # continue each side FROM THE ORACLE'S tensor so that errors do not accumulate: per-stage kernel error
    net_idx = 0
    cur = o
    for s, depth in enumerate(arch["layers"]):
        if net_idx == 2:
            o2 = rb(cur + rf(R.interpolate_pos_encoding(p["pos_embed"], cur.shape[1], cur.shape[2])))
            h2 = AF.AddPosFn.apply(cur.to(torch.bfloat16).cuda(), model.interpolate_pos_encoding(cur.cuda()))
            print("pos add", rel(h2, o2)); cur = o2
        for i in range(depth):
            pre = "network.%d.%d." % (net_idx, i)
            blk = model.network[net_idx][i]
            hip = blk(cur.to(torch.bfloat16).cuda())
            if s == 0:
                ora = R.outlooker_bf16_points(cur, p, pre, arch["num_heads"][0])
            else:
                Bc, H, W, C = cur.shape
                ora = R.transformer_bf16_points(cur.reshape(Bc, H * W, C), p, pre, arch["num_heads"][s]).reshape(Bc, H, W, C)
            print(pre, rel(hip, ora)); cur = ora
        net_idx += 1
        if s == 0:
            pre = "network.%d." % net_idx
            Bc, H, W, C = cur.shape
            w = p[pre + "proj.weight"]
            patches = cur.reshape(Bc, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(Bc, H // 2, W // 2, 4 * C)
            ora = rb(patches @ rw(w.permute(0, 2, 3, 1).reshape(w.shape[0], 4 * C)).t() + p[pre + "proj.bias"])
            hip = model.network[net_idx](cur.to(torch.bfloat16).cuda())
            print("downsample", rel(hip, ora)); cur = ora
            net_idx += 1
    Bc, H, W, C = cur.shape
    tok = cur.reshape(Bc, H * W, C)
    cls = rf(p["cls_token"]).expand(Bc, -1, -1)
    hc, ht = model.forward_cls(tok.to(torch.bfloat16).cuda())
    for j in range(2):
        cls = R.class_block_bf16_points(torch.cat([cls, tok], 1), p, "post_network.%d." % j, arch["num_heads"][-1])[:, :1]
    print("class blocks", rel(hc, cls))
    ncls = rb(R.layernorm(cls, p["norm.weight"], p["norm.bias"])); ntok = rb(R.layernorm(tok, p["norm.weight"], p["norm.bias"]))
    xc = rb(R.linear(ncls[:, 0], rw(p["head.weight"]), p["head.bias"])); xa = rb(R.linear(ntok, rw(p["aux_head.weight"]), p["aux_head.bias"]))
    hcl = AF.layer_norm(cls.to(torch.bfloat16).cuda(), model.norm.weight, model.norm.bias, model.norm.eps)
    htk = AF.layer_norm(tok.to(torch.bfloat16).cuda(), model.norm.weight, model.norm.bias, model.norm.eps)
    print("final norm", rel(hcl, ncls), rel(htk, ntok))
    print("heads", rel(AF.linear(ncls[:, 0].to(torch.bfloat16).cuda(), model.head.weight, model.head.bias), xc),
          rel(AF.linear(ntok.to(torch.bfloat16).cuda(), model.aux_head.weight, model.aux_head.bias), xa))
