#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (imported from /root/reference).

Run in the build container only (the reference tree does not exist on the GPU box):

    python tools/gen_golden.py            # rewrites tests/golden/*.npz

Fixtures are data only: inputs, weights (flat name -> array), outputs and gradients of the
reference modules on small seeded problems.  No reference
source, bytecode or pickled module is stored.
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def npy(t):
    return t.detach().cpu().numpy().copy() if torch.is_tensor(t) else np.asarray(t)


def sd_arrays(module, prefix="w."):
    return {prefix + k: npy(v) for k, v in module.state_dict().items()}


def grads(module, prefix="g."):
    return {prefix + k: npy(v.grad) for k, v in module.named_parameters() if v.grad is not None}


def randomize_(module, gen, scale=0.5):
    """Give every parameter a non-trivial value (init leaves biases 0 and LN at 1/0)."""
    with torch.no_grad():
        for n, p in module.named_parameters():
            if p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=gen) * (scale / max(1.0, p.shape[-1] ** 0.5) * 2))
            elif n.endswith("weight"):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=gen))
            else:
                p.copy_(0.2 * torch.randn(p.shape, generator=gen))
    return module


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote %-28s %8.1f KB  (%d arrays)" % (name + ".npz", os.path.getsize(path) / 1024, len(arrs)))


# ----------------------------------------------------------------------------------------------

def gen_int_tables(ns):
    out = {}
    vs = np.round(np.concatenate([np.arange(1, 40) * 0.23, np.linspace(50, 260, 43), np.arange(1, 30)]), 6)
    divs = [1, 2, 8, 32]
    out["md_v"] = vs
    out["md_div"] = np.array(divs)
    out["md_out"] = np.array([[ns.progressive.make_divisible(float(v), d) for d in divs] for v in vs])
    pairs, idx_rows, new_rows = [], [], []
    for prev in range(1, 19):
        for new in range(prev, min(2 * prev, 18) + 1):
            pairs.append((prev, new))
            row = [ns.helpers.new_idx(i, prev, new) for i in range(new)] + [-1] * (18 - new)
            idx_rows.append(row)
            fresh = ns.helpers.get_new_layer_idx(prev, new)
            new_rows.append(fresh + [-1] * (18 - len(fresh)))
    out["ni_pairs"], out["ni_map"], out["ni_fresh"] = np.array(pairs), np.array(idx_rows), np.array(new_rows)
    # set_sample_config skip masks: build a tiny supernet per (min,max) and read the flags back
    cfgs, masks = [], []
    for lmin, lmax in [(9, 18), (9, 15), (12, 18), (15, 18), (9, 9), (12, 15), (4, 8), (6, 12), (3, 6)]:
        l0max = ns.progressive.make_divisible(lmax * 0.23, 2)
        layers = [l0max, lmax - l0max, 0, 0]
        net = ns.volo.VOLO(layers, img_size=32, num_classes=4, embed_dims=[16, 32, 32, 32], num_heads=[1, 2, 2, 2],
                           mlp_ratios=[1, 1, 1, 1], downsamples=[True, False, False, False],
                           outlook_attention=[True, False, False, False], post_layers=["ca", "ca"], stem_hidden_dim=8)
        for l in range(lmin, lmax + 1):
            try:
                net.set_sample_config(dict(layer_num=l, min_layer_num=lmin, max_layer_num=lmax))
            except Exception:
                continue
            flags = []
            for st in (0, 2):
                flags += [int(getattr(b, "is_identity_layer", False)) for b in net.network[st]]
            cfgs.append((l, lmin, lmax, layers[0], layers[1]))
            masks.append(flags + [-1] * (18 - len(flags)))
    out["ss_cfg"], out["ss_mask"] = np.array(cfgs), np.array(masks)
    out["depth_l"] = np.arange(1, 25)
    out["depth_l0"] = np.array([ns.progressive.make_divisible(l * 0.23, 2) for l in range(1, 25)])
    # rand_bbox / mix-token RNG sequences
    seeds, rows = [], []
    for seed in range(12):
        for g in (8, 14, 20, 28):
            np.random.seed(seed)
            lam = np.random.beta(1.0, 1.0)
            bb = ns.volo.rand_bbox((4, g, g, 8), lam, scale=2)
            seeds.append((seed, g))
            rows.append([lam] + [float(v) for v in bb])
    out["bb_seed"], out["bb_out"] = np.array(seeds), np.array(rows)
    # progressive_schedule: shipped script flags and argparse defaults
    def sched(**kw):
        a = types.SimpleNamespace(num_stages=4, r_scale=0.5, h_scale=1.0, l_scale=0.5, aa_scale=0.0, dp_scale=-0.5,
                                  re_scale=-0.5, resize_scale=[1.0, 1.0], aa="rand-m9-mstd0.5-inc1", drop_path=0.1,
                                  reprob=0.25, scale=[0.08, 1.0], epochs=300)
        for k, v in kw.items():
            setattr(a, k, v)
        e, r, h, l, aa, dp, re, rs = ns.progressive.progressive_schedule(a, r_max=224, h_max=12, l_max=18)
        mags = [int(s.split("-")[1].lstrip("m")) if s else 0 for s in aa]
        return np.array(e), np.array(r), np.array(h), np.array(l), np.array(mags), np.array(dp), np.array(re), np.array(rs)
    for tag, kw in [("script", dict(aa_scale=0.5, dp_scale=0.0, re_scale=0.0, epochs=100)), ("default", {}),
                    ("s3", dict(num_stages=3, epochs=90, r_scale=0.6, l_scale=0.4))]:
        for nm, arr in zip(("e", "r", "h", "l", "aa", "dp", "re", "rs"), sched(**kw)):
            out["ps_%s_%s" % (tag, nm)] = arr
    save("int_tables", **out)


def run_module(mod, x, gen, extra=None):
    x = x.clone().requires_grad_(True)
    y = mod(x)
    dy = torch.randn(y.shape, generator=gen)
    mod.zero_grad()
    y.backward(dy)
    d = dict(x=npy(x), y=npy(y), dy=npy(dy), dx=npy(x.grad))
    d.update(sd_arrays(mod))
    d.update(grads(mod))
    if extra:
        d.update(extra)
    return d


def gen_outlook(ns):
    out = {}
    for tag, (B, H, W, C, heads) in {"even8": (2, 8, 8, 32, 2), "odd7": (2, 7, 7, 32, 2), "rect6x10": (1, 6, 10, 64, 2),
                                     "odd5x9": (2, 5, 9, 32, 1), "even16": (1, 16, 16, 64, 2)}.items():
        gen = torch.Generator().manual_seed(100 + H * 7 + W)
        mod = randomize_(ns.volo.OutlookAttention(C, heads, kernel_size=3, padding=1, stride=2), gen, 1.5).train()
        x = torch.randn(B, H, W, C, generator=gen)
        d = run_module(mod, x, gen)
        # also pin the core alone (v, logits) -> fold output, via hooks on proj input
        d["heads"] = np.array(heads)
        for k, v in d.items():
            out["%s.%s" % (tag, k)] = v
    save("outlook_attn", **out)


def gen_blocks(ns):
    out = {}
    gen = torch.Generator().manual_seed(7)
    B, H, W, C, heads = 2, 6, 6, 64, 2
    cases = {
        "mlp": (ns.volo.Mlp(C, C * 3), torch.randn(B, H, W, C, generator=gen)),
        "attention": (ns.volo.Attention(C, heads), torch.randn(B, H, W, C, generator=gen)),
        "attention_n25": (ns.volo.Attention(C, heads), torch.randn(3, 5, 5, C, generator=gen)),
        "class_attention": (ns.volo.ClassAttention(C, heads), torch.randn(B, 1 + H * W, C, generator=gen)),
        "class_block": (ns.volo.ClassBlock(C, heads, mlp_ratio=3.0), torch.randn(B, 1 + H * W, C, generator=gen)),
        "outlooker": (ns.volo.Outlooker(C, 3, 1, stride=2, num_heads=heads, mlp_ratio=3.0), torch.randn(B, 8, 8, C, generator=gen)),
        "transformer": (ns.volo.Transformer(C, heads, mlp_ratio=3.0), torch.randn(B, H, W, C, generator=gen)),
        "downsample": (ns.volo.Downsample(32, 64, 2), torch.randn(B, 8, 8, 32, generator=gen)),
    }
    for tag, (mod, x) in cases.items():
        randomize_(mod, gen, 1.0).train()
        d = run_module(mod, x, gen)
        for k, v in d.items():
            out["%s.%s" % (tag, k)] = v
    # LayerNorm alone (eps 1e-5 / 1e-6)
    for eps in (1e-5, 1e-6):
        ln = randomize_(torch.nn.LayerNorm(48, eps=eps), gen)
        d = run_module(ln, torch.randn(5, 7, 48, generator=gen) * 3 + 1, gen)
        for k, v in d.items():
            out["layernorm_%g.%s" % (eps, k)] = v
    save("blocks", **out)


def gen_stem(ns):
    out = {}
    gen = torch.Generator().manual_seed(11)
    pe = ns.volo.PatchEmbed(stem_conv=True, stem_stride=2, patch_size=8, in_chans=3, hidden_dim=8, embed_dim=16)
    randomize_(pe, gen, 1.0).train()
    x = torch.randn(2, 3, 32, 32, generator=gen)
    xr = x.clone().requires_grad_(True)
    y = pe(xr)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy)
    out.update({"train.x": npy(x), "train.y": npy(y), "train.dy": npy(dy), "train.dx": npy(xr.grad)})
    out.update({"train." + k: v for k, v in sd_arrays(pe).items()})     # includes UPDATED running stats
    out.update({"train." + k: v for k, v in grads(pe).items()})
    pe.eval()
    out["eval.y"] = npy(pe(x))
    save("stem", **out)


def gen_stem64(ns):
    """the BASELINE-width stem (hidden_dim 64: the shapes csrc/conv7.hip and csrc/conv.hip are written for), B = 2, 32 x 32 input:
    train-mode output, input gradient, every parameter gradient, updated running statistics, eval-mode output"""
    out = {}
    gen = torch.Generator().manual_seed(12)
    pe = ns.volo.PatchEmbed(stem_conv=True, stem_stride=2, patch_size=8, in_chans=3, hidden_dim=64, embed_dim=32)
    randomize_(pe, gen, 1.0).train()
    with torch.no_grad():                   # keep the three 64-channel activations O(1): a 576-term convolution of N(0,1) weights is not
        for i in (0, 3, 6):
            pe.conv[i].weight.mul_(1.0 / (pe.conv[i].weight[0].numel() ** 0.5))
        pe.proj.weight.mul_(1.0 / (pe.proj.weight[0].numel() ** 0.5))
    x = torch.randn(2, 3, 32, 32, generator=gen)
    xr = x.clone().requires_grad_(True)
    y = pe(xr)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy)
    out.update({"train.x": npy(x), "train.y": npy(y), "train.dy": npy(dy), "train.dx": npy(xr.grad)})
    out.update({"train." + k: v for k, v in sd_arrays(pe).items()})     # includes UPDATED running stats
    out.update({"train." + k: v for k, v in grads(pe).items()})
    pe.eval()
    out["eval.y"] = npy(pe(x))
    save("stem64", **out)


def gen_stem128(ns):
    """the stem of VOLO-D4 / D5 (hidden_dim 128, models/volo.py:799-821: the width csrc/conv128.hip serves), B = 2, 32 x 32 input: as stem64"""
    out = {}
    gen = torch.Generator().manual_seed(13)
    pe = ns.volo.PatchEmbed(stem_conv=True, stem_stride=2, patch_size=8, in_chans=3, hidden_dim=128, embed_dim=32)
    randomize_(pe, gen, 1.0).train()
    with torch.no_grad():
        for i in (0, 3, 6):
            pe.conv[i].weight.mul_(1.0 / (pe.conv[i].weight[0].numel() ** 0.5))
        pe.proj.weight.mul_(1.0 / (pe.proj.weight[0].numel() ** 0.5))
    x = torch.randn(2, 3, 32, 32, generator=gen)
    xr = x.clone().requires_grad_(True)
    y = pe(xr)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy)
    out.update({"train.x": npy(x), "train.y": npy(y), "train.dy": npy(dy), "train.dx": npy(xr.grad)})
    out.update({"train." + k: v for k, v in sd_arrays(pe).items()})     # includes UPDATED running stats
    out.update({"train." + k: v for k, v in grads(pe).items()})
    pe.eval()
    out["eval.y"] = npy(pe(x))
    save("stem128", **out)


def gen_pos_interp(ns):
    out = {}
    gen = torch.Generator().manual_seed(3)
    net = types.SimpleNamespace(pos_embed=torch.randn(1, 14, 14, 8, generator=gen))
    out["pos"] = npy(net.pos_embed)
    for g in (8, 10, 12, 14, 16, 7):
        x = torch.zeros(1, g, g, 8)
        out["interp_%d" % g] = npy(ns.volo.VOLO.interpolate_pos_encoding(net, x))
    net2 = types.SimpleNamespace(pos_embed=torch.randn(1, 4, 4, 8, generator=gen))
    out["pos4"] = npy(net2.pos_embed)
    for g in (2, 3, 4, 6):
        out["interp4_%d" % g] = npy(ns.volo.VOLO.interpolate_pos_encoding(net2, torch.zeros(1, g, g, 8)))
    save("pos_interp", **out)


def tiny_volo(ns, variant, img, classes, dpr=0.0, stem=16):
    fam, h, l = variant.split("_")
    h, l = int(h[1:]), int(l[1:])
    l0 = ns.progressive.make_divisible(l * 0.23, 2) if l > 2 else 1
    layers = [l0, l - l0, 0, 0] if l > 2 else [1, 1, 0, 0]
    net = ns.volo.VOLO(layers, img_size=img, num_classes=classes, embed_dims=[16 * h, 32 * h, 32 * h, 32 * h],
                       num_heads=[h // 2, h, h, h], mlp_ratios=[3, 3, 3, 3], downsamples=[True, False, False, False],
                       outlook_attention=[True, False, False, False], post_layers=["ca", "ca"], drop_path_rate=dpr,
                       stem_hidden_dim=stem)
    return net


def make_target(B, C, N, gen, three_slots=True):
    """token-label style target [B,C,2+N]: sparse top-5 soft labels + smoothing."""
    t = torch.zeros(B, C, 2 + N)
    for b in range(B):
        for s in range(2 + N):
            idx = torch.randperm(C, generator=gen)[:min(5, C)]
            val = torch.rand(len(idx), generator=gen)
            t[b, idx, s] = val / val.sum() * (0.6 + 0.8 * torch.rand(1, generator=gen))   # rows need not sum to 1
    t = t * 0.9 + 0.1 / C
    return t


def gen_volo_full(ns):
    out = {}
    for tag, variant, img, classes in [("h2_l3", "volo_h2_l3", 64, 16), ("h2_l6", "volo_h2_l6", 96, 12)]:
        gen = torch.Generator().manual_seed(21 + img)
        torch.manual_seed(5)
        net = randomize_(tiny_volo(ns, variant, 64, classes), gen, 1.0)
        with torch.no_grad():
            net.pos_embed.copy_(0.3 * torch.randn(net.pos_embed.shape, generator=gen))
            net.cls_token.copy_(0.3 * torch.randn(net.cls_token.shape, generator=gen))
        x = torch.randn(2, 3, img, img, generator=gen)
        g2 = img // 16
        target = make_target(2, classes, g2 * g2, gen)
        loss_fn = ns.cross_entropy.TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=classes)
        net.train()
        np.random.seed(1234)
        x_cls, x_aux, bb = net(x)
        loss = loss_fn((x_cls, x_aux, bb), target)
        net.zero_grad()
        loss.backward()
        np.random.seed(1234)
        lam = np.random.beta(1.0, 1.0)
        d = {"x": npy(x), "target": npy(target), "x_cls": npy(x_cls), "x_aux": npy(x_aux), "bbox": np.array([int(v) for v in bb]),
             "lam": np.array(lam), "loss": npy(loss), "np_seed": np.array(1234)}
        d.update(sd_arrays(net))          # after forward: BN running stats already updated
        d.update(grads(net))
        net.eval()
        with torch.no_grad():
            d["eval_y"] = npy(net(x))
        for k, v in d.items():
            out["%s.%s" % (tag, k)] = v
    # supernet sub-configs (eval mode, h2_l6 supernet with min 3 / max 6)
    gen = torch.Generator().manual_seed(77)
    net = randomize_(tiny_volo(ns, "volo_h2_l6", 64, 10), gen, 1.0).eval()
    x = torch.randn(2, 3, 64, 64, generator=gen)
    out["super.x"] = npy(x)
    out.update({"super." + k: v for k, v in sd_arrays(net).items()})
    for l in (3, 4, 5, 6):
        net.set_sample_config(dict(layer_num=l, min_layer_num=3, max_layer_num=6))
        with torch.no_grad():
            out["super.eval_y_l%d" % l] = npy(net(x))
    save("volo_full", **out)


def gen_volo_full64(ns):
    """A whole network on the SHIPPED stem (missing 3): volo_h2_l3 with stem_hidden_dim=64 -- the width of every BASELINE
    config, the width the HIP convolution kernels serve -- 64 px, batch 8 (8 x 32 x 32 samples per BatchNorm channel: not degenerate),
    same stress weights as volo_full: train outputs, loss, every parameter gradient, eval output."""
    out = {}
    tag, variant, img, classes, B = "h2_l3_s64", "volo_h2_l3", 64, 16, 8
    gen = torch.Generator().manual_seed(2164)
    torch.manual_seed(5)
    net = randomize_(tiny_volo(ns, variant, img, classes, stem=64), gen, 1.0)
    with torch.no_grad():
        net.pos_embed.copy_(0.3 * torch.randn(net.pos_embed.shape, generator=gen))
        net.cls_token.copy_(0.3 * torch.randn(net.cls_token.shape, generator=gen))
    x = torch.randn(B, 3, img, img, generator=gen)
    g2 = img // 16
    target = make_target(B, classes, g2 * g2, gen)
    loss_fn = ns.cross_entropy.TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=classes)
    net.train()
    np.random.seed(4321)
    x_cls, x_aux, bb = net(x)
    loss = loss_fn((x_cls, x_aux, bb), target)
    net.zero_grad()
    loss.backward()
    np.random.seed(4321)
    lam = np.random.beta(1.0, 1.0)
    d = {"x": npy(x), "target": npy(target), "x_cls": npy(x_cls), "x_aux": npy(x_aux), "bbox": np.array([int(v) for v in bb]),
         "lam": np.array(lam), "loss": npy(loss), "np_seed": np.array(4321)}
    d.update(sd_arrays(net))          # after forward: BN running stats already updated
    d.update(grads(net))
    net.eval()
    with torch.no_grad():
        d["eval_y"] = npy(net(x))
    for k, v in d.items():
        out["%s.%s" % (tag, k)] = v
    save("volo_full64", **out)


def gen_loss(ns):
    out = {}
    gen = torch.Generator().manual_seed(9)
    B, N, C = 4, 9, 20
    cls = (torch.randn(B, C, generator=gen) * 2).requires_grad_(True)
    aux = (torch.randn(B, N, C, generator=gen) * 2).requires_grad_(True)
    t3 = make_target(B, C, N, gen)
    t2 = torch.softmax(torch.randn(B, C, generator=gen), -1)
    out.update(cls=npy(cls), aux=npy(aux), t3=npy(t3), t2=npy(t2))
    ce = ns.cross_entropy
    cases = {
        "tl_box": (ce.TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=C), (0, 1, 2, 3), t3),
        "tl_nobox": (ce.TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=C), (0, 0, 0, 0), t3),
        "tl_2d": (ce.TokenLabelCrossEntropy(dense_weight=1.0, cls_weight=1.0, classes=C), (1, 0, 3, 2), t2),
        "gt_box": (ce.TokenLabelGTCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=C), (0, 1, 2, 3), t3),
        "gt_2d": (ce.TokenLabelGTCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=C), (0, 0, 0, 0), t2),
    }
    for tag, (fn, bb, tgt) in cases.items():
        cls.grad = aux.grad = None
        loss = fn((cls, aux, bb), tgt)
        loss.backward()
        out[tag + ".bbox"], out[tag + ".loss"] = np.array(bb), npy(loss)
        out[tag + ".dcls"], out[tag + ".daux"] = npy(cls.grad), npy(aux.grad)
    # SoftTargetCrossEntropy incl. the target-repeat path, and TokenLabelSoftTargetCrossEntropy
    x = (torch.randn(8, C, generator=gen)).requires_grad_(True)
    loss = ce.SoftTargetCrossEntropy()(x, t2)
    loss.backward()
    out.update({"st.x": npy(x), "st.loss": npy(loss), "st.dx": npy(x.grad)})
    x2 = (torch.randn(B, C, generator=gen)).requires_grad_(True)
    t_pair = t3[:, :, :2].contiguous()
    loss = ce.TokenLabelSoftTargetCrossEntropy()(x2, t_pair)
    loss.backward()
    out.update({"tlst.x": npy(x2), "tlst.t": npy(t_pair), "tlst.loss": npy(loss), "tlst.dx": npy(x2.grad)})
    save("loss", **out)


def gen_step_curve(ns):
    """5 AdamW steps of the tiny model on a fixed batch: the loss-curve pin (north_star 1e-3)."""
    gen = torch.Generator().manual_seed(31)
    net = randomize_(tiny_volo(ns, "volo_h2_l3", 64, 16), gen, 1.0).train()
    x = torch.randn(4, 3, 64, 64, generator=gen)
    target = make_target(4, 16, 16, gen)
    loss_fn = ns.cross_entropy.TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=16)
    out = {"x": npy(x), "target": npy(target)}
    out.update(sd_arrays(net))
    decay, no_decay = [], []
    for n, p in net.named_parameters():
        (no_decay if (p.dim() == 1 or n.endswith(".bias") or n in ("pos_embed", "cls_token")) else decay).append(p)
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.05}, {"params": no_decay, "weight_decay": 0.0}], lr=1e-3)
    losses, boxes = [], []
    np.random.seed(99)
    for _ in range(5):
        outp = net(x)
        loss = loss_fn(outp, target)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        boxes.append([int(v) for v in outp[2]])
    out["losses"], out["boxes"], out["np_seed"], out["lr"], out["wd"] = np.array(losses), np.array(boxes), np.array(99), np.array(1e-3), np.array(0.05)
    save("step_curve", **out)


def gen_step_curve_init(ns):
    """10 AdamW steps of volo_h4_l6 (64 px, batch 8, 32 classes) from the reference's OWN initialisation statistics
    (trunc-normal .02 Linear weights, default conv init; tests/_initweights.py regenerates the identical state dict
    from the seed, so no weights are stored): the realistic-init loss-curve pin (north_star: 1e-3).

    The reference model is run twice: in fp64 (`losses`, `g0.*`: the exact curve of the reference's arithmetic) and in
    fp32 (`losses_fp32`).  The two differ by up to 7.7e-4 from step 4 on: torch's fp32 CPU convolution backward leaves a
    0.3 % error in the 7x7 / 3x3 stem weight gradients and AdamW's first steps (update ~ lr * sign(g)) amplify it --
    that is the reference's own noise floor on this curve, recorded so the 1e-3 budget can be read against it."""
    sys.path.insert(0, os.path.dirname(HERE))
    from tests._initweights import init_state_dict
    gen = torch.Generator().manual_seed(77)
    classes, B, r, seed = 32, 8, 64, 2024
    x = torch.randn(B, 3, r, r, generator=gen)
    target = make_target(B, classes, (r // 16) ** 2, gen)
    out = {"x": npy(x), "target": npy(target), "np_seed": np.array(123), "lr": np.array(1e-3), "wd": np.array(0.05),
           "init_seed": np.array(seed), "classes": np.array(classes)}
    for dt, tag in ((torch.float64, ""), (torch.float32, "_fp32"), (torch.float32, "_fp16_autocast"),
                    (torch.float32, "_bf16_autocast")):
        # "_fp16_autocast": the reference model under torch.autocast("cpu", float16) -- the closest thing to its apex-O1 training
        # path that runs here (16-bit GEMM/conv operands, fp32 master weights and loss): the reference's own mixed-precision
        # deviation from its exact curve, recorded as the yardstick for any 16-bit-activation implementation
        # "_bf16_autocast": the same under torch.autocast("cpu", bfloat16) -- the reference's own network in the arithmetic class
        # of the HIP path (bf16 operands, fp32 accumulation): how far bf16 rounding alone moves THIS curve
        amp = {"_fp16_autocast": torch.float16, "_bf16_autocast": torch.bfloat16}.get(tag)
        net = tiny_volo(ns, "volo_h4_l6", r, classes).train()
        net.load_state_dict(init_state_dict(net.state_dict(), seed), strict=True)
        net = net.to(dt)
        loss_fn = ns.cross_entropy.TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=classes)
        decay, no_decay = [], []
        for n, p in net.named_parameters():
            (no_decay if (p.dim() == 1 or n.endswith(".bias") or n in ("pos_embed", "cls_token")) else decay).append(p)
        opt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.05}, {"params": no_decay, "weight_decay": 0.0}], lr=1e-3)
        losses, boxes = [], []
        np.random.seed(123)
        g0 = None
        for step in range(10):
            if amp:
                with torch.autocast("cpu", dtype=amp):
                    outp = net(x)
                outp = (outp[0].float(), outp[1].float(), outp[2])
            else:
                outp = net(x.to(dt))
            loss = loss_fn(outp, target.to(dt))
            opt.zero_grad()
            loss.backward()
            if step == 0:
                g0 = grads(net, "g0.")
            opt.step()
            losses.append(float(loss.detach()))
            boxes.append([int(v) for v in outp[2]])
        out["losses" + tag] = np.array(losses)
        print("step_curve_init losses%s:" % tag, losses)
        if tag:
            continue
        out["boxes"] = np.array(boxes)
        # first-step gradients of a few tensors per kind (the full set would be 4 MB) + the norms of all of them
        keep = ("patch_embed.conv.0.weight", "patch_embed.conv.3.weight", "patch_embed.conv.6.weight", "patch_embed.conv.4.weight",
                "patch_embed.proj.weight", "network.0.0.attn.v.weight",
                "network.0.1.attn.attn.weight", "network.0.1.mlp.fc1.weight", "network.1.proj.weight", "network.2.0.attn.qkv.weight",
                "network.2.3.mlp.fc2.weight", "network.2.1.norm1.weight", "post_network.0.attn.kv.weight", "post_network.1.mlp.fc1.bias",
                "pos_embed", "cls_token", "head.weight", "aux_head.weight", "norm.bias")
        for k in keep:
            out["g0." + k] = g0["g0." + k].astype(np.float32)
        out["g0_norms_names"] = np.array(sorted(k[3:] for k in g0))
        out["g0_norms"] = np.array([float(np.linalg.norm(g0["g0." + k])) for k in out["g0_norms_names"]])
    save("step_curve_init", **out)


LATE_STEM = 64          # round 5: the trained-state fixture runs on the shipped stem width (the HIP convolution kernels), not the 16-wide one


def gen_late_state(ns):
    """Parity in the regime a TRAINED network is in (Weak 2): the realistic-init volo_h4_l6 of step_curve_init is trained by
    the reference, in fp64, for 300 AdamW steps on its batch (lr 1e-3: the residual stream grows, attention logits sharpen, the loss
    falls from 5.7 to the memorising regime); then the late state -- rounded to fp32, the precision a checkpoint holds -- is evaluated
    once more by the reference in fp64 on a fresh mix-token draw: logits, loss, every parameter gradient.  Next to it the reference's
    OWN bf16-autocast evaluation of the same weights (per-tensor gradient errors and the error of all gradients as one vector): the
    yardstick for any implementation in that arithmetic class.  Gradients are stored as fp16 mantissas with a per-tensor scale
    (relative error 5e-4, far below what is asserted against them)."""
    sys.path.insert(0, os.path.dirname(HERE))
    from tests._initweights import init_state_dict
    gen = torch.Generator().manual_seed(77)
    classes, B, r, seed, steps = 32, 8, 64, 2024, 300
    x = torch.randn(B, 3, r, r, generator=gen)
    target = make_target(B, classes, (r // 16) ** 2, gen)
    net = tiny_volo(ns, "volo_h4_l6", r, classes, stem=LATE_STEM).train()
    net.load_state_dict(init_state_dict(net.state_dict(), seed), strict=True)
    net = net.double()
    loss_fn = ns.cross_entropy.TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=classes)
    decay, no_decay = [], []
    for n, p in net.named_parameters():
        (no_decay if (p.dim() == 1 or n.endswith(".bias") or n in ("pos_embed", "cls_token")) else decay).append(p)
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.05}, {"params": no_decay, "weight_decay": 0.0}], lr=1e-3)
    np.random.seed(123)
    curve = []
    for step in range(steps):
        loss = loss_fn(net(x.double()), target.double())
        opt.zero_grad()
        loss.backward()
        opt.step()
        curve.append(float(loss.detach()))
    print("late_state: loss %.4f -> %.4f after %d steps" % (curve[0], curve[-1], steps))
    # the late state as a checkpoint would hold it
    sd32 = {k: (v.float() if v.dtype.is_floating_point else v) for k, v in net.state_dict().items()}
    out = {"x": npy(x), "target": npy(target), "np_seed": np.array(7), "classes": np.array(classes), "steps": np.array(steps),
           "train_curve": np.array(curve)}
    out.update({"w." + k: npy(v) for k, v in sd32.items()})

    def evaluate(dtype, amp=None):
        m = tiny_volo(ns, "volo_h4_l6", r, classes, stem=LATE_STEM).train()
        m.load_state_dict(sd32, strict=True)
        m = m.to(dtype)
        np.random.seed(7)
        if amp is not None:
            with torch.autocast("cpu", dtype=amp):
                o = m(x.to(dtype))
            o = (o[0].float(), o[1].float(), o[2])
        else:
            o = m(x.to(dtype))
        l = loss_fn(o, target.to(dtype))
        l.backward()
        return o, l, {k: v.grad.detach().double() for k, v in m.named_parameters() if v.grad is not None}
    o64, l64, g64 = evaluate(torch.float64)
    ob, lb, gb = evaluate(torch.float32, torch.bfloat16)
    rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    out["y_cls"], out["y_aux"], out["box"], out["loss"] = npy(o64[0]), npy(o64[1]), np.array([int(v) for v in o64[2]]), np.array(float(l64))
    names = sorted(k for k, g in g64.items() if float(g.norm()) > 1e-12)
    out["grad_names"] = np.array(names)
    for k in names:
        g = g64[k]
        sc = float(g.abs().max())
        out["gs." + k] = np.array(sc)
        out["g16." + k] = (g / sc).to(torch.float16).numpy()
    out["yard_names"] = np.array(names)
    out["yard_bf16_autocast"] = np.array([rel(gb[k], g64[k]) for k in names])
    va = torch.cat([gb[k].flatten() for k in names]); vb = torch.cat([g64[k].flatten() for k in names])
    out["yard_bf16_autocast_one_vector"] = np.array(float((va - vb).norm() / vb.norm()))
    out["yard_bf16_autocast_logits"] = np.array([rel(ob[0], o64[0]), rel(ob[1], o64[1])])
    out["yard_bf16_autocast_loss"] = np.array(float(lb))
    print("late_state: loss %.5f; reference bf16-autocast: logits %.4f / %.4f, loss %.5f, gradients median %.4f max %.4f one-vector %.4f"
          % (float(l64), rel(ob[0], o64[0]), rel(ob[1], o64[1]), float(lb), float(np.median(out["yard_bf16_autocast"])),
             float(out["yard_bf16_autocast"].max()), float(out["yard_bf16_autocast_one_vector"])))
    save("late_state", **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="", help="comma-separated generator names (e.g. step_curve_init); default all")
    args = ap.parse_args()
    only = set(filter(None, args.only.split(",")))
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)
    ns = ref_import.load_reference()
    for fn in (gen_int_tables, gen_outlook, gen_blocks, gen_stem, gen_stem64, gen_stem128, gen_pos_interp, gen_volo_full, gen_volo_full64, gen_loss, gen_step_curve,
               gen_step_curve_init, gen_late_state):
        if not only or fn.__name__[4:] in only:
            fn(ns)


if __name__ == "__main__":
    main()
