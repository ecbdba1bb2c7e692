"""CPU oracle for the AutoProg VOLO/DeiT training hot path.

TEST INFRASTRUCTURE ONLY.  This is a from-scratch restatement (closed forms, functional
style, plain torch CPU fp32/fp64 + numpy) of the arithmetic the reference performs on
its hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` may import it; the product package ``autoprog_amd`` never does.

Parity pin: every function here is checked against golden vectors emitted by the real
reference (``tools/gen_golden.py`` imports /root/reference in the build container and
writes ``tests/golden/*.npz``); see ``tests/test_oracle_golden.py``.  DeiT arithmetic
lives in un-vendored timm 0.4.5 (``timm.models.vision_transformer``) and is restated
from its published algorithm: **parity unpinned** for ``vit_forward`` -- no vector the reference
holds constrains it.  What stands in: its blocks are the pinned VOLO Transformer arithmetic; the
block equals ``torch.nn.MultiheadAttention`` + pre-LN wiring (fp64, 1e-10); and the whole network
-- patch embedding, class / distillation tokens, position embedding, blocks, final norm, both heads
and their eval average, outputs AND gradients -- equals the ``transformers`` library's DeiT (the
port of facebook/deit, the family models/deit.py registers) on the same weights to 1e-9
(``test_vit_and_distilled_deit_against_the_transformers_library``).

All ``file:line`` citations are relative to the reference tree.

``*_bf16_points`` / ``bf16_points=True`` (round 4): the same functions with every tensor that the MI355X pipeline keeps in bf16 rounded
to bf16 at the same place, forward and backward (custom autograd nodes where a kernel rounds inside an operation: attention
probabilities / dS, the outlook core's per-window products, the stored gelu').  They are NOT a second reference: with the rounding
nodes removed they are the functions above, which the golden vectors pin.  The GPU tests use them to separate what the 16-bit recipe
costs (HIP path against the plain oracle: percent level) from what the kernels add (HIP path against these: 1e-7 ... 3e-3 per block,
the fixtures' transformer and class blocks bit-identical in the forward).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

# --------------------------------------------------------------------------------------
# integer bookkeeping (bit-exact rows A11, A13)
# --------------------------------------------------------------------------------------

def make_divisible(v, divisor=8, min_value=None, round_limit=0.9):
    """prog/progressive.py:34-40."""
    floor = min_value or divisor
    out = max(floor, int(v + divisor / 2) // divisor * divisor)
    if out < round_limit * v:
        out += divisor
    return out


def new_idx(idx: int, prev_l: int, new_l: int) -> int:
    """prog/helpers.py:254-258 -- source layer feeding destination layer ``idx``."""
    reps = new_l // prev_l
    plain = prev_l - new_l % prev_l          # number of source layers repeated `reps` times
    first = idx * prev_l // (reps * prev_l)
    if first < plain:
        return first
    return (idx + plain) * prev_l // (reps * prev_l + prev_l)


def get_new_layer_idx(prev_l: int, new_l: int) -> List[int]:
    """prog/helpers.py:261-262 -- destination layers that duplicate their predecessor."""
    return [i for i in range(new_l) if new_idx(i, prev_l, new_l) == new_idx(i - 1, prev_l, new_l)]


def stage_depths(l: int) -> List[int]:
    """models/submodels.py:19-25 and models/volo.py:602-607: total depth -> per-stage."""
    if l > 2:
        l0 = make_divisible(l * 0.23, 2)
        return [l0, l - l0, 0, 0]
    return [1, 1, 0, 0]


def skip_layer_table(layer_num: int, min_layer_num: int, max_layer_num: int) -> List[List[int]]:
    """models/volo.py:598-609: per-stage indices of identity (skipped) layers."""
    cur = [make_divisible(layer_num * 0.23, 2)]
    cur = [cur[0], layer_num - cur[0], 0, 0]
    lo0 = make_divisible(min_layer_num * 0.23, 2)
    lo = [lo0, min_layer_num - lo0, 0, 0]
    hi0 = make_divisible(max_layer_num * 0.23, 2)
    hi = [hi0, max_layer_num - hi0, 0, 0]
    table = []
    for s in range(4):
        fresh = get_new_layer_idx(lo[s], hi[s]) if hi[s] > 0 else []
        extra = cur[s] - lo[s]
        table.append(fresh if extra == 0 else fresh[:-extra])
    return table


def parse_variant(variant: str) -> Tuple[str, int, int]:
    """'volo_h12_l18' -> ('volo', 12, 18) (models/submodels.py:16-17, intended behaviour)."""
    fam, h, l = variant.split("_")
    return fam, int(h.lstrip("h")), int(l.lstrip("l"))


def variant_arch(variant: str) -> dict:
    """models/submodels.py:9-40 architecture table for 'volo_h{H}_l{L}'."""
    fam, h, l = parse_variant(variant)
    assert fam == "volo" and h % 2 == 0
    return dict(layers=stage_depths(l), embed_dims=[16 * h, 32 * h, 32 * h, 32 * h],
                num_heads=[h // 2, h, h, h], mlp_ratios=[3, 3, 3, 3],
                downsamples=[True, False, False, False],
                outlook_attention=[True, False, False, False], post_layers=["ca", "ca"],
                stem_hidden_dim=64)


VOLO_PRESETS = {  # models/volo.py:697-821
    "volo_d1": dict(layers=[4, 4, 8, 2], embed_dims=[192, 384, 384, 384], num_heads=[6, 12, 12, 12],
                    mlp_ratios=[3, 3, 3, 3], stem_hidden_dim=64),
    "volo_d2": dict(layers=[6, 4, 10, 4], embed_dims=[256, 512, 512, 512], num_heads=[8, 16, 16, 16],
                    mlp_ratios=[3, 3, 3, 3], stem_hidden_dim=64),
    "volo_d3": dict(layers=[8, 8, 16, 4], embed_dims=[256, 512, 512, 512], num_heads=[8, 16, 16, 16],
                    mlp_ratios=[3, 3, 3, 3], stem_hidden_dim=64),
    "volo_d4": dict(layers=[8, 8, 16, 4], embed_dims=[384, 768, 768, 768], num_heads=[12, 16, 16, 16],
                    mlp_ratios=[3, 3, 3, 3], stem_hidden_dim=64),
    "volo_d5": dict(layers=[12, 12, 20, 4], embed_dims=[384, 768, 768, 768], num_heads=[12, 16, 16, 16],
                    mlp_ratios=[4, 4, 4, 4], stem_hidden_dim=128),
}


def rand_bbox(size: Sequence[int], lam: float, scale: int = 1, rng=np.random):
    """models/volo.py:319-339.  ``size`` is the [B,H,W,C] token shape; note the first box
    axis is the H axis although it is named W.  RNG call order: randint(W) then randint(H)."""
    gw = size[1] // scale
    gh = size[2] // scale
    cut = np.sqrt(1.0 - lam)
    cw = int(gw * cut)
    ch = int(gh * cut)
    cx = rng.randint(gw)
    cy = rng.randint(gh)
    x1 = int(np.clip(cx - cw // 2, 0, gw))
    y1 = int(np.clip(cy - ch // 2, 0, gh))
    x2 = int(np.clip(cx + cw // 2, 0, gw))
    y2 = int(np.clip(cy + ch // 2, 0, gh))
    return x1, y1, x2, y2


def draw_mix_box(token_shape: Sequence[int], pooling_scale: int = 2, beta: float = 1.0, rng=np.random):
    """models/volo.py:650-653: lam ~ Beta(beta,beta) then rand_bbox on the pooled grid."""
    lam = rng.beta(beta, beta)
    return lam, rand_bbox(token_shape, lam, scale=pooling_scale, rng=rng)


def progressive_schedule(num_stages, epochs, r_scale, h_scale, l_scale, aa_scale, dp_scale, re_scale,
                         resize_scale, aa, drop_path, reprob, scale, r_max=224, h_max=12, l_max=18):
    """prog/progressive.py:4-31 as a pure function of the flags."""
    lin = lambda lo: np.linspace(lo, 1.0, num_stages)
    e = [int(i) for i in np.linspace(0, epochs, num_stages + 1) // 1][:-1]
    r = [make_divisible(i, 32) for i in lin(r_scale) * r_max]
    h = [make_divisible(i, 2) for i in lin(h_scale) * h_max]
    l = [make_divisible(i, 1) for i in lin(l_scale) * l_max]
    m_max = float(aa.split("-")[1].lstrip("m"))
    mags = [round(max(0.0, i)) for i in lin(aa_scale) * m_max]
    aas = ["rand-m{}-mstd0.5-inc1".format(m) if m > 0 else "" for m in mags]
    dp = [max(0.0, i) for i in lin(dp_scale) * drop_path]
    re = [max(0.0, i) for i in lin(re_scale) * reprob]
    rs = [[max(0.0, a), max(0.0, b)] for a, b in zip(lin(resize_scale[0]) * scale[0], lin(resize_scale[1]) * scale[1])]
    return e, r, h, l, aas, dp, re, rs


# --------------------------------------------------------------------------------------
# floating-point building blocks
# --------------------------------------------------------------------------------------

def layernorm(x, w, b, eps=1e-5):
    """nn.LayerNorm over the last dim: biased variance, affine (row N0)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def linear(x, w, b=None):
    y = x @ w.t()
    return y if b is None else y + b


def gelu(x):
    """exact erf GELU (nn.GELU default), models/volo.py:157."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def mlp(x, p: Params, pre: str):
    """models/volo.py:161-167."""
    return linear(gelu(linear(x, p[pre + "fc1.weight"], p[pre + "fc1.bias"])),
                  p[pre + "fc2.weight"], p[pre + "fc2.bias"])


def avgpool_ceil(x, s: int):
    """AvgPool2d(s, s, ceil_mode=True) on [B,H,W,C]: edge windows are clipped and divided
    by the clipped element count (models/volo.py:75,87)."""
    B, H, W, C = x.shape
    h, w = -(-H // s), -(-W // s)
    xp = x.new_zeros(B, h * s, w * s, C)
    xp[:, :H, :W] = x
    ones = x.new_zeros(1, h * s, w * s, 1)
    ones[:, :H, :W] = 1
    tot = xp.reshape(B, h, s, w, s, C).sum((2, 4))
    cnt = ones.reshape(1, h, s, w, s, 1).sum((2, 4))
    return tot / cnt


def outlook_core(v, logits, heads: int, K: int = 3, P: int = 1, S: int = 2):
    """Closed form of unfold -> softmax(attn) @ v -> fold (models/volo.py:83-98).

    v      [B,H,W,C]            values (after the v projection)
    logits [B,h,w,heads*K^4]    attn projection of the pooled tokens, channel =
                                head*K^4 + p*K^2 + q  (models/volo.py:88-90)
    returns Y [B,H,W,C] = fold of per-window outputs (before proj).
    """
    B, H, W, C = v.shape
    hd = C // heads
    h, w = -(-H // S), -(-W // S)
    KK = K * K
    A = logits.reshape(B, h, w, heads, KK, KK) * (hd ** -0.5)
    Pm = torch.softmax(A, dim=-1)
    Hp, Wp = max(H + 2 * P, S * (h - 1) + K), max(W + 2 * P, S * (w - 1) + K)
    vp = v.new_zeros(B, Hp, Wp, heads, hd)
    vp[:, P:P + H, P:P + W] = v.reshape(B, H, W, heads, hd)
    # Vn[b,i,j,q] = vp[b, S*i + q//K, S*j + q%K]
    Vn = torch.stack([vp[:, (q // K):(q // K) + S * h:S, (q % K):(q % K) + S * w:S] for q in range(KK)], dim=3)
    # O[b,i,j,p,head,:] = sum_q Pm[b,i,j,head,p,q] * Vn[b,i,j,q,head,:]
    O = torch.einsum("bijhpq,bijqhd->bijphd", Pm, Vn)
    yp = v.new_zeros(B, Hp, Wp, heads, hd)
    for p in range(KK):
        yp[:, (p // K):(p // K) + S * h:S, (p % K):(p % K) + S * w:S] += O[:, :, :, p]
    return yp[:, P:P + H, P:P + W].reshape(B, H, W, C)


def outlook_attention(x, p: Params, pre: str, heads: int, K=3, P=1, S=2):
    """OutlookAttention.forward, models/volo.py:77-103 (v has no bias: qkv_bias=False)."""
    v = linear(x, p[pre + "v.weight"], p.get(pre + "v.bias"))
    logits = linear(avgpool_ceil(x, S), p[pre + "attn.weight"], p[pre + "attn.bias"])
    y = outlook_core(v, logits, heads, K, P, S)
    return linear(y, p[pre + "proj.weight"], p[pre + "proj.bias"])


def mhsa_core(qkv, heads: int):
    """softmax(q k^T * hd^-0.5) v on packed qkv [B,N,3C], channel = which*C+head*hd+d
    (models/volo.py:188-197).  Returns [B,N,C]."""
    B, N, C3 = qkv.shape
    C = C3 // 3
    hd = C // heads
    q, k, v = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    att = torch.softmax((q @ k.transpose(-1, -2)) * (hd ** -0.5), dim=-1)
    return (att @ v).transpose(1, 2).reshape(B, N, C)


def attention(x, p: Params, pre: str, heads: int):
    """Attention.forward, models/volo.py:185-201, on [B,N,C] tokens."""
    qkv = linear(x, p[pre + "qkv.weight"], p.get(pre + "qkv.bias"))
    return linear(mhsa_core(qkv, heads), p[pre + "proj.weight"], p[pre + "proj.bias"])


def class_attention(x, p: Params, pre: str, heads: int):
    """ClassAttention.forward, models/volo.py:261-277: one query (token 0), N keys; the
    scale multiplies q before the product."""
    B, N, C = x.shape
    hd = C // heads
    kv = linear(x, p[pre + "kv.weight"], p.get(pre + "kv.bias")).reshape(B, N, 2, heads, hd)
    k, v = kv[:, :, 0].transpose(1, 2), kv[:, :, 1].transpose(1, 2)          # [B,heads,N,hd]
    q = linear(x[:, :1], p[pre + "q.weight"], p.get(pre + "q.bias")).reshape(B, heads, 1, hd)
    att = torch.softmax((q * hd ** -0.5) @ k.transpose(-1, -2), dim=-1)
    out = (att @ v).transpose(1, 2).reshape(B, 1, C)
    return linear(out, p[pre + "proj.weight"], p[pre + "proj.bias"])


def drop_path_scale(mask: Optional[torch.Tensor], keep: float, y):
    """timm DropPath: y / keep * mask_b.  ``mask`` None => identity."""
    if mask is None:
        return y
    return y / keep * mask.reshape(-1, *([1] * (y.dim() - 1))).to(y.dtype)


def outlooker(x, p: Params, pre: str, heads: int):
    """Outlooker.forward, models/volo.py:140-144 (drop_path is always 0 here)."""
    x = x + outlook_attention(layernorm(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"]), p, pre + "attn.", heads)
    return x + mlp(layernorm(x, p[pre + "norm2.weight"], p[pre + "norm2.bias"]), p, pre + "mlp.")


def transformer(x, p: Params, pre: str, heads: int, dp_masks=None, keep: float = 1.0):
    """Transformer.forward, models/volo.py:230-234, x [B,H,W,C]; dp_masks = (mask1, mask2)."""
    B, H, W, C = x.shape
    m1, m2 = dp_masks if dp_masks is not None else (None, None)
    a = attention(layernorm(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"]).reshape(B, H * W, C), p, pre + "attn.", heads)
    x = x + drop_path_scale(m1, keep, a.reshape(B, H, W, C))
    return x + drop_path_scale(m2, keep, mlp(layernorm(x, p[pre + "norm2.weight"], p[pre + "norm2.bias"]), p, pre + "mlp."))


def class_block(x, p: Params, pre: str, heads: int):
    """ClassBlock.forward, models/volo.py:304-308: only token 0 is updated."""
    cls = x[:, :1] + class_attention(layernorm(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"]), p, pre + "attn.", heads)
    cls = cls + mlp(layernorm(cls, p[pre + "norm2.weight"], p[pre + "norm2.bias"]), p, pre + "mlp.")
    return torch.cat([cls, x[:, 1:]], dim=1)


def batchnorm_train(x, w, b, eps=1e-5):
    """BatchNorm2d in training mode (batch statistics, biased variance)."""
    mu = x.mean((0, 2, 3), keepdim=True)
    var = ((x - mu) ** 2).mean((0, 2, 3), keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w.reshape(1, -1, 1, 1) + b.reshape(1, -1, 1, 1)


def batchnorm_eval(x, w, b, rm, rv, eps=1e-5):
    sh = (1, -1, 1, 1)
    return (x - rm.reshape(sh)) * torch.rsqrt(rv.reshape(sh) + eps) * w.reshape(sh) + b.reshape(sh)


class _RoundBoth(torch.autograd.Function):
    """a tensor boundary of the bf16 pipeline: the value is rounded to bf16 on the way forward, its gradient on the way back"""

    @staticmethod
    def forward(ctx, x):
        return _r16(x)

    @staticmethod
    def backward(ctx, g):
        return _r16(g)


class _RoundOperand(torch.autograd.Function):
    """a parameter used as a bf16 MFMA operand: rounded forward, its gradient (an fp32 accumulation) is not"""

    @staticmethod
    def forward(ctx, w):
        return _r16(w)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundFwd(torch.autograd.Function):
    """an activation stored in bf16 whose gradient is consumed unrounded by the next epilogue: rounded forward only"""

    @staticmethod
    def forward(ctx, x):
        return _r16(x)

    @staticmethod
    def backward(ctx, g):
        return g


BF16_POINTS_ROUND = True      # tests/test_oracle_golden.py switches it off to show that the *_bf16_points functions ARE the pinned ones plus rounding


def _r16(t):
    return t.to(torch.bfloat16).to(t.dtype) if BF16_POINTS_ROUND else t


GELU_GRAD_BITS = 8            # the stored derivative's rounding point: 8 = fixed point, step 1/202 (gelu = 3 / mul_by8, the round-5 default of the fused
                              # blocks); 16 = bf16 (gelu = 2 / mul_by: rounds 3 - 4, AP_GELU_STORE_GRAD=1, and functional.LinearFn)


def _rgelu_grad(d, bits=None):
    if not BF16_POINTS_ROUND:
        return d
    if (GELU_GRAD_BITS if bits is None else bits) == 8:
        return ((d * 202.0 + 26.0).round().clamp(0.0, 255.0) - 26.0) / 202.0
    return _r16(d)


class _GeluBf16Points(torch.autograd.Function):
    """fc1's epilogue on the MI355X path: a = bf16(gelu(h)) of the ROUNDED pre-activation h, and the rounded gelu'(h) is what is stored
    for the backward, whose epilogue multiplies the fp32 accumulator of fc2's input gradient by it (csrc/gemm_epi.h: gelu = 3 / mul_by8
    stores 8-bit fixed-point codes, gelu = 2 / mul_by bf16 -- GELU_GRAD_BITS)"""

    @staticmethod
    def forward(ctx, h):
        ctx.save_for_backward(h)
        return _r16(gelu(h))

    @staticmethod
    def backward(ctx, g):
        (h,) = ctx.saved_tensors
        d = 0.5 * (1.0 + torch.erf(h / math.sqrt(2.0))) + h * torch.exp(-0.5 * h * h) / math.sqrt(2.0 * math.pi)
        return g * _rgelu_grad(d)


class _MhsaBf16Points(torch.autograd.Function):
    """softmax(q k^T s) v as csrc/mhsa.hip computes it: fp32 scores and softmax statistics from bf16 q, k; the probabilities rounded to
    bf16 as the operand of the P V product but summed unrounded for the normalisation; bf16 output.  Backward: delta = rowsum(O dO) from
    the bf16 tensors, P recomputed from the log-sum-exp, dV = bf16(P)^T dO, dS = P (dP - delta) rounded to bf16 as the operand of
    dQ = dS K s and dK = dS^T Q s."""

    @staticmethod
    def forward(ctx, qkv, heads):
        B, N, C3 = qkv.shape
        C = C3 // 3
        hd = C // heads
        q, k, v = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
        sc = hd ** -0.5
        s = (q @ k.transpose(-1, -2)) * sc
        m = s.max(-1, keepdim=True)[0]
        pr = torch.exp(s - m)
        l = pr.sum(-1, keepdim=True)
        o = _r16((_r16(pr) @ v) / l)
        ctx.save_for_backward(q, k, v, o, m + torch.log(l))
        ctx.cfg = (B, N, C, heads, hd, sc)
        return o.transpose(1, 2).reshape(B, N, C)

    @staticmethod
    def backward(ctx, g):
        q, k, v, o, lse = ctx.saved_tensors
        B, N, C, heads, hd, sc = ctx.cfg
        do = _r16(g).reshape(B, N, heads, hd).transpose(1, 2)
        delta = (o * do).sum(-1, keepdim=True)
        pr = torch.exp((q @ k.transpose(-1, -2)) * sc - lse)
        dv = _r16(pr).transpose(-1, -2) @ do
        ds = _r16(pr * (do @ v.transpose(-1, -2) - delta))
        dq = (ds @ k) * sc
        dk = (ds.transpose(-1, -2) @ q) * sc
        dqkv = torch.stack([dq, dk, dv], 0).permute(1, 3, 0, 2, 4).reshape(B, N, 3 * C)
        return dqkv, None


def transformer_bf16_points(x, p: Params, pre: str, heads: int, eps: float = 1e-5):
    """Transformer.forward (models/volo.py:230-234, DropPath off) in the caller's precision with every tensor the MI355X block keeps in bf16
    rounded where functional.TransformerBlockFn rounds it, forward and backward (LayerNorm outputs, qkv, attention probabilities /
    output / dS, both residual sums, the pre-activation and its GELU, the stored gelu', every gradient tensor, the weights as matrix
    operands): what is left against the HIP block is its kernels' own arithmetic.  x [B,N,C], bf16-valued."""
    rb, rw = _RoundBoth.apply, _RoundOperand.apply
    xn1 = rb(layernorm(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"], eps))
    qkv = rb(linear(xn1, rw(p[pre + "attn.qkv.weight"]), p.get(pre + "attn.qkv.bias")))
    o = _MhsaBf16Points.apply(qkv, heads)
    x1 = rb(linear(o, rw(p[pre + "attn.proj.weight"]), p[pre + "attn.proj.bias"]) + x)
    xn2 = rb(layernorm(x1, p[pre + "norm2.weight"], p[pre + "norm2.bias"], eps))
    h = rb(linear(xn2, rw(p[pre + "mlp.fc1.weight"]), p[pre + "mlp.fc1.bias"]))
    a = _GeluBf16Points.apply(h)
    return rb(linear(a, rw(p[pre + "mlp.fc2.weight"]), p[pre + "mlp.fc2.bias"]) + x1)


class _OutlookBf16Points(torch.autograd.Function):
    """unfold -> softmax -> attn @ v -> fold (models/volo.py:83-98; kernel 3, padding 1, stride 2) as csrc/outlook.hip computes it:
    fp32 softmax of the bf16 logits; the probabilities rounded to bf16 as the matrix operand; every WINDOW's product rounded to bf16
    before the fold adds the (up to four) windows that cover a pixel in fp32; bf16 output.  Backward: dV = fold(bf16(P)^T dY per window,
    rounded), dlogits = s P (dP - <P, dP>) with the UNROUNDED probabilities and dP = dY V^T over the window's 32 channels."""

    @staticmethod
    def forward(ctx, v, logits, heads):
        B, H, W, C = v.shape
        hd = C // heads
        h, w = -(-H // 2), -(-W // 2)
        L = h * w
        vu = F.unfold(v.permute(0, 3, 1, 2), kernel_size=3, padding=1, stride=2)               # [B, C*9, L], channel = c*9 + q
        vu = vu.reshape(B, heads, hd, 9, L).permute(0, 1, 4, 3, 2)                              # [B, heads, L, 9 q, hd]
        a = logits[..., :heads * 81].reshape(B, L, heads, 9, 9).permute(0, 2, 1, 3, 4)          # [B, heads, L, 9 p, 9 q]
        pr = torch.softmax(a * hd ** -0.5, dim=-1)
        z = _r16(_r16(pr) @ vu)                                                                  # [B, heads, L, 9 p, hd]
        y = F.fold(z.permute(0, 1, 4, 3, 2).reshape(B, C * 9, L), (H, W), kernel_size=3, padding=1, stride=2)
        ctx.save_for_backward(vu, pr)
        ctx.cfg = (B, H, W, C, heads, hd, L, logits.shape[-1])
        return _r16(y).permute(0, 2, 3, 1)

    @staticmethod
    def backward(ctx, g):
        vu, pr = ctx.saved_tensors
        B, H, W, C, heads, hd, L, ldl = ctx.cfg
        gu = F.unfold(_r16(g).permute(0, 3, 1, 2), kernel_size=3, padding=1, stride=2).reshape(B, heads, hd, 9, L).permute(0, 1, 4, 3, 2)
        dvu = _r16(_r16(pr).transpose(-1, -2) @ gu)                                              # [B, heads, L, 9 q, hd]
        dv = F.fold(dvu.permute(0, 1, 4, 3, 2).reshape(B, C * 9, L), (H, W), kernel_size=3, padding=1, stride=2).permute(0, 2, 3, 1)
        dp = gu @ vu.transpose(-1, -2)                                                           # [B, heads, L, 9 p, 9 q]
        da = hd ** -0.5 * pr * (dp - (pr * dp).sum(-1, keepdim=True))
        dl = g.new_zeros(B, L, ldl)
        dl[..., :heads * 81] = da.permute(0, 2, 1, 3, 4).reshape(B, L, heads * 81)
        return dv, dl, None


POOL_GRAD_ROUNDED = False


def outlooker_bf16_points(x, p: Params, pre: str, heads: int):
    """Outlooker.forward (models/volo.py:140-144) with the rounding points of functional.OutlookerBlockFn (see transformer_bf16_points);
    x [B,H,W,C], bf16-valued.  LayerNorm-1's output feeds the v projection and the 2x2 average pool: the v path's input gradient and the
    pool's output gradient are bf16 tensors (the rounding nodes on the two paths); their sum is formed in fp32 INSIDE the LayerNorm
    backward kernel (ap_layernorm_bwd_partial_pool, round 5) and is not rounded -- POOL_GRAD_ROUNDED = True restores the extra rounding of
    the separate ap_avgpool2_bwd_acc pass (AP_FUSE_POOL_BWD=0)."""
    rb, rw = _RoundBoth.apply, _RoundOperand.apply
    B, H, W, C = x.shape
    xn1 = (rb if POOL_GRAD_ROUNDED else rw)(layernorm(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"]))
    v = rb(linear(rb(xn1), rw(p[pre + "attn.v.weight"]), p.get(pre + "attn.v.bias")))
    pooled = rb(avgpool_ceil(xn1, 2))
    logits = rb(linear(pooled.reshape(B, -1, C), rw(p[pre + "attn.attn.weight"]), p[pre + "attn.attn.bias"]))
    yo = _OutlookBf16Points.apply(v, logits, heads)
    x1 = rb(linear(yo, rw(p[pre + "attn.proj.weight"]), p[pre + "attn.proj.bias"]) + x)
    xn2 = rb(layernorm(x1, p[pre + "norm2.weight"], p[pre + "norm2.bias"]))
    hh = rb(linear(xn2, rw(p[pre + "mlp.fc1.weight"]), p[pre + "mlp.fc1.bias"]))
    a = _GeluBf16Points.apply(hh)
    return rb(linear(a, rw(p[pre + "mlp.fc2.weight"]), p[pre + "mlp.fc2.bias"]) + x1)


def class_block_bf16_points(x, p: Params, pre: str, heads: int):
    """ClassBlock.forward (models/volo.py:304-308) with the rounding points of functional.ClassBlockFn; x [B,1+N,C], bf16-valued.  The
    class attention itself runs in fp32 on bf16 q / k / v with fp32 probabilities (csrc/mhsa.hip k_class_attn_*): plain autograd between
    rounding nodes.  LayerNorm-1 of the class token feeds q and kv: the kv path's input gradient is rounded before the q path's is added
    to it in the q GEMM's residual epilogue, hence the second rounding node there."""
    rb, rw = _RoundBoth.apply, _RoundOperand.apply
    B, N1, C = x.shape
    hd = C // heads
    c0, t0 = x[:, :1], x[:, 1:]
    n1w, n1b = p[pre + "norm1.weight"], p[pre + "norm1.bias"]
    nc, nt = rb(layernorm(c0, n1w, n1b)), rb(layernorm(t0, n1w, n1b))
    wkv, bkv = rw(p[pre + "attn.kv.weight"]), p.get(pre + "attn.kv.bias")
    kv = torch.cat([rb(linear(rb(nc), wkv, bkv)), rb(linear(nt, wkv, bkv))], dim=1).reshape(B, N1, 2, heads, hd)
    k, v = kv[:, :, 0].transpose(1, 2), kv[:, :, 1].transpose(1, 2)
    q = rb(linear(nc, rw(p[pre + "attn.q.weight"]), p.get(pre + "attn.q.bias"))).reshape(B, heads, 1, hd)
    att = torch.softmax((q * hd ** -0.5) @ k.transpose(-1, -2), dim=-1)
    o = rb((att @ v).transpose(1, 2).reshape(B, 1, C))
    c1 = rb(linear(o, rw(p[pre + "attn.proj.weight"]), p[pre + "attn.proj.bias"]) + c0)
    n2 = rb(layernorm(c1, p[pre + "norm2.weight"], p[pre + "norm2.bias"]))
    h = rb(linear(n2, rw(p[pre + "mlp.fc1.weight"]), p[pre + "mlp.fc1.bias"]))
    a = _GeluBf16Points.apply(h)
    c2 = rb(linear(a, rw(p[pre + "mlp.fc2.weight"]), p[pre + "mlp.fc2.bias"]) + c1)
    return torch.cat([c2, t0], dim=1)


def patch_embed(x, p: Params, train: bool, patch_size: int = 8, pre: str = "patch_embed.", bf16_points: bool = False):
    """PatchEmbed.forward, models/volo.py:376-380: conv7x7 s2 -> BN -> ReLU -> 2x(conv3x3 ->
    BN -> ReLU) -> conv(patch/2) stride patch/2 with bias.  Returns tokens [B,H,W,C].
    bf16_points: the same arithmetic in the caller's precision, but every tensor the MI355X pipeline keeps in bf16 (the image, each
    convolution output, each activation, the convolution weights as matrix operands -- and the gradients of those tensors on the way
    back) is rounded to bf16 where that pipeline rounds it.  Against THIS statement the kernels' own error shows (accumulation order,
    a rounding that flips on a tie), not the precision recipe's: tests/test_gpu_blocks.py holds them to 5e-3 with it (measured <= 2.6e-3)."""
    rb = _RoundBoth.apply if bf16_points else (lambda t: t)
    rw = _RoundOperand.apply if bf16_points else (lambda t: t)
    if bf16_points:
        x = _r16(x)
    strides = [(2, 3), (1, 1), (1, 1)]
    for i, (s, pad) in zip((0, 3, 6), strides):
        x = rb(F.conv2d(x, rw(p[pre + "conv.%d.weight" % i]), None, stride=s, padding=pad))
        bn = pre + "conv.%d." % (i + 1)
        if train:
            x = batchnorm_train(x, p[bn + "weight"], p[bn + "bias"])
        else:
            x = batchnorm_eval(x, p[bn + "weight"], p[bn + "bias"], p[bn + "running_mean"], p[bn + "running_var"])
        x = rb(torch.relu(x))
    k = patch_size // 2
    x = rb(F.conv2d(x, rw(p[pre + "proj.weight"]), p[pre + "proj.bias"], stride=k))
    return x.permute(0, 2, 3, 1)


def downsample(x, p: Params, pre: str, k: int = 2):
    """Downsample.forward, models/volo.py:392-396 == patch-gather GEMM on NHWC tokens."""
    B, H, W, C = x.shape
    w = p[pre + "proj.weight"]                       # [Cout, Cin, k, k]
    patches = x.reshape(B, H // k, k, W // k, k, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H // k, W // k, k * k * C)
    wmat = w.permute(0, 2, 3, 1).reshape(w.shape[0], k * k * C)   # (ky,kx,cin) order
    return patches @ wmat.t() + p[pre + "proj.bias"]


def interpolate_pos_encoding(pos, h0: int, w0: int):
    """VOLO.interpolate_pos_encoding, models/volo.py:580-596 (bicubic, scale_factor form)."""
    h, w = pos.shape[1], pos.shape[2]
    if h == h0 and w == w0:
        return pos
    out = F.interpolate(pos.permute(0, 3, 1, 2), scale_factor=((h0 + 0.1) / h, (w0 + 0.1) / w), mode="bicubic")
    assert out.shape[-2] == h0 and out.shape[-1] == w0
    return out.permute(0, 2, 3, 1)


def mix_token_swap(x, box, scale: int):
    """models/volo.py:654-658 / 685-689: region [x1*s:x2*s, y1*s:y2*s] (first index = H axis)
    is replaced by the batch-flipped tensor's region."""
    x1, y1, x2, y2 = [scale * int(v) for v in box]
    out = x.clone()
    out[:, x1:x2, y1:y2] = x.flip(0)[:, x1:x2, y1:y2]
    return out


def volo_forward(p: Params, img, layers, embed_dims, num_heads, train: bool, mix=None,
                 skip: Optional[List[List[int]]] = None, dp_masks: Optional[dict] = None,
                 drop_path_rate: float = 0.0, patch_size: int = 8, pooling_scale: int = 2, bf16_points: bool = False,
                 bn_train: Optional[bool] = None, **_):
    """VOLO.forward, models/volo.py:644-694, for the model_variant/volo_d* families
    (outlook stage -> downsample -> transformer stages -> 2 class blocks -> heads).

    mix      None or (lam, (bbx1,bby1,bbx2,bby2)) used when ``train`` (mix-token).
    skip     per-stage identity-layer indices (set_sample_config), default none.
    dp_masks {(stage, idx): (mask1, mask2)} per-sample keep masks for DropPath; the keep
             probability follows models/volo.py:428-437.
    bn_train None: the stem's BatchNorm follows ``train``; False with train=True: a training forward (mix-token, DropPath, the two
             heads) on the RUNNING statistics -- the samples of a batch are then independent, which the full-size slice tests use
             (model.train(); model.patch_embed.eval() on the module side).
    Returns (x_cls, x_aux, box) in train mode, fused logits in eval mode.
    bf16_points (train mode, DropPath off): the whole network with the rounding points of the MI355X pipeline -- patch_embed(bf16_points),
    outlooker_ / transformer_ / class_block_bf16_points, the downsample and head GEMMs, the position embedding added as a bf16 tensor, the
    class token cast once, and one rounding of the token gradient per hop of the chain class block 1 <- class block 2 <- final norm
    (functional.ClassBlockFn hands the tokens on; each hop's LayerNorm backward adds and rounds).
    """
    if bf16_points:
        assert train and not dp_masks and drop_path_rate == 0.0, "bf16_points: training forward without DropPath"
        rb, rw, rf = _RoundBoth.apply, _RoundOperand.apply, _RoundFwd.apply
        x = patch_embed(img, p, True, patch_size, bf16_points=True)
        box = (0, 0, 0, 0)
        if mix is not None:
            box = tuple(int(v) for v in mix[1])
            x = mix_token_swap(x, box, pooling_scale)
        skip = skip or [[], [], [], []]
        net_idx = 0
        for s, depth in enumerate(layers):
            if net_idx == 2:
                x = rb(x + rf(interpolate_pos_encoding(p["pos_embed"], x.shape[1], x.shape[2])))
            for i in range(depth):
                if i in skip[s]:
                    continue
                pre = "network.%d.%d." % (net_idx, i)
                if s == 0:
                    x = outlooker_bf16_points(x, p, pre, num_heads[0])
                else:
                    B, H, W, C = x.shape
                    x = transformer_bf16_points(x.reshape(B, H * W, C), p, pre, num_heads[s]).reshape(B, H, W, C)
            net_idx += 1
            if s == 0:
                pre = "network.%d." % net_idx
                B, H, W, C = x.shape
                w = p[pre + "proj.weight"]
                patches = x.reshape(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H // 2, W // 2, 4 * C)
                x = rb(patches @ rw(w.permute(0, 2, 3, 1).reshape(w.shape[0], 4 * C)).t() + p[pre + "proj.bias"])
                net_idx += 1
        B, H, W, C = x.shape
        tok = x.reshape(B, H * W, C)
        cls = rf(p["cls_token"]).expand(B, -1, -1)
        for j in range(2):
            cls = class_block_bf16_points(torch.cat([cls, tok], dim=1), p, "post_network.%d." % j, num_heads[-1])[:, :1]
            tok = rb(tok)                           # the next consumer's token gradient is a bf16 tensor when it comes back through this hop
        ncls = rb(layernorm(cls, p["norm.weight"], p["norm.bias"]))
        ntok = rb(layernorm(tok, p["norm.weight"], p["norm.bias"]))
        x_cls = rb(linear(ncls[:, 0], rw(p["head.weight"]), p["head.bias"]))
        x_aux = rb(linear(ntok, rw(p["aux_head.weight"]), p["aux_head.bias"]))
        if mix is not None:
            nc = x_aux.shape[-1]
            x_aux = mix_token_swap(x_aux.reshape(B, H, W, nc), box, 1).reshape(B, H * W, nc)
        return x_cls, x_aux, box
    x = patch_embed(img, p, train if bn_train is None else bn_train, patch_size)
    box = (0, 0, 0, 0)
    if train and mix is not None:
        box = tuple(int(v) for v in mix[1])
        x = mix_token_swap(x, box, pooling_scale)
    skip = skip or [[], [], [], []]
    total = sum(layers)
    net_idx = 0
    for s, depth in enumerate(layers):
        if net_idx == 2:    # pos embed is added before the first transformer stage (models/volo.py:627)
            x = x + interpolate_pos_encoding(p["pos_embed"], x.shape[1], x.shape[2])
        for i in range(depth):
            if i in skip[s]:
                continue
            pre = "network.%d.%d." % (net_idx, i)
            if s == 0:
                x = outlooker(x, p, pre, num_heads[0])
            else:
                rate = drop_path_rate * (i + sum(layers[:s])) / (total - 1) if total > 1 else 0.0
                masks = dp_masks.get((s, i)) if (dp_masks and train and rate > 0) else None
                x = transformer(x, p, pre, num_heads[s], masks, 1.0 - rate)
        net_idx += 1
        if s == 0:
            x = downsample(x, p, "network.%d." % net_idx)
            net_idx += 1
    B, H, W, C = x.shape
    x = x.reshape(B, H * W, C)
    x = torch.cat([p["cls_token"].expand(B, -1, -1), x], dim=1)
    for j in range(2):
        x = class_block(x, p, "post_network.%d." % j, num_heads[-1])
    x = layernorm(x, p["norm.weight"], p["norm.bias"])
    x_cls = linear(x[:, 0], p["head.weight"], p["head.bias"])
    x_aux = linear(x[:, 1:], p["aux_head.weight"], p["aux_head.bias"])
    if not train:
        return x_cls + 0.5 * x_aux.max(1)[0]
    if mix is not None:
        nc = x_aux.shape[-1]
        x_aux = mix_token_swap(x_aux.reshape(B, H, W, nc), box, 1).reshape(B, H * W, nc)
    return x_cls, x_aux, box


# --------------------------------------------------------------------------------------
# losses (rows L1-L4), loss/cross_entropy.py
# --------------------------------------------------------------------------------------

def soft_target_ce(x, t):
    """loss/cross_entropy.py:30-36: mean_i( -sum_c t_ic * log_softmax(x_i)_c ); the target is
    tiled when x has more rows."""
    if x.shape[0] != t.shape[0]:
        t = t.repeat(x.shape[0] // t.shape[0], 1)
    lse = torch.logsumexp(x, dim=-1, keepdim=True)
    return (-(t * (x - lse)).sum(-1)).mean()


def _token_label_targets(aux_shape, target):
    B, N, C = aux_shape
    if target.dim() == 2:
        return target, target.repeat(1, N).reshape(B * N, C), None
    t_aux = target[:, :, 2:].transpose(1, 2).reshape(-1, C)
    return target[:, :, 1], t_aux, target[:, :, 0]


def token_label_ce(outputs, target, dense_weight=1.0, cls_weight=1.0):
    """TokenLabelCrossEntropy.forward, loss/cross_entropy.py:136-156."""
    out, aux, (x1, y1, x2, y2) = outputs
    B, N, C = aux.shape
    t_cls, t_aux, _ = _token_label_targets(aux.shape, target)
    lam = 1 - ((x2 - x1) * (y2 - y1) / N)
    if lam < 1:
        t_cls = lam * t_cls + (1 - lam) * t_cls.flip(0)
    return cls_weight * soft_target_ce(out, t_cls) + dense_weight * soft_target_ce(aux.reshape(-1, C), t_aux)


def token_label_gt_ce(outputs, target, dense_weight=1.0, cls_weight=1.0):
    """TokenLabelGTCrossEntropy.forward, loss/cross_entropy.py:62-89."""
    out, aux, (x1, y1, x2, y2) = outputs
    B, N, C = aux.shape
    t_cls, t_aux, gt = _token_label_targets(aux.shape, target)
    if gt is not None:
        same = (gt.max(-1)[1] == t_cls.max(-1)[1])
        ratio = (0.9 - 0.4 * same.to(t_cls.dtype)).unsqueeze(-1)
        t_cls = t_cls * ratio + gt * (1 - ratio)
    lam = 1 - ((x2 - x1) * (y2 - y1) / N)
    if lam < 1:
        t_cls = lam * t_cls + (1 - lam) * t_cls.flip(0)
    return cls_weight * soft_target_ce(out, t_cls) + dense_weight * soft_target_ce(aux.reshape(-1, C), t_aux)


def token_label_soft_target_ce(x, target):
    """TokenLabelSoftTargetCrossEntropy.forward, loss/cross_entropy.py:101-109."""
    if x.shape[0] != target.shape[0]:
        target = target.repeat(x.shape[0] // target.shape[0], 1)
    if target.dim() == 3 and target.shape[-1] == 2:
        target = target[:, :, 1]
    return soft_target_ce(x, target)


# --------------------------------------------------------------------------------------
# DeiT / timm VisionTransformer restatement (parity unpinned, see module docstring)
# --------------------------------------------------------------------------------------

def vit_block(x, p: Params, pre: str, heads: int, eps=1e-6, dp_masks=None, keep=1.0):
    m1, m2 = dp_masks if dp_masks is not None else (None, None)
    x = x + drop_path_scale(m1, keep, attention(layernorm(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"], eps), p, pre + "attn.", heads))
    return x + drop_path_scale(m2, keep, mlp(layernorm(x, p[pre + "norm2.weight"], p[pre + "norm2.bias"], eps), p, pre + "mlp."))


def vit_forward(p: Params, img, depth: int, heads: int, patch: int = 16, distilled: bool = False,
                train: bool = True, skip: Sequence[int] = (), bf16_points: bool = False):
    """timm 0.4.5 VisionTransformer.forward (+ DistilledVisionTransformer, models/deit.py:32-59).
    bf16_points: the same network with the MI355X pipeline's rounding points (see transformer_bf16_points): bf16 patches and patch
    projection, class / distillation tokens cast once, the position embedding added as a bf16 tensor, bf16 final norm and heads."""
    if bf16_points:
        rb, rw, rf = _RoundBoth.apply, _RoundOperand.apply, _RoundFwd.apply
        w = p["patch_embed.proj.weight"]
        B, Cin, H, W = img.shape
        hh, ww = H // patch, W // patch
        patches = img[:, :, :hh * patch, :ww * patch].reshape(B, Cin, hh, patch, ww, patch).permute(0, 2, 4, 1, 3, 5).reshape(B, hh * ww, Cin * patch * patch)
        x = rb(linear(_r16(patches), rw(w.reshape(w.shape[0], -1)), p["patch_embed.proj.bias"]))
        toks = [rf(p["cls_token"]).expand(B, -1, -1)] + ([rf(p["dist_token"]).expand(B, -1, -1)] if distilled else [])
        pos, n_extra = p["pos_embed"], len(toks)
        g0, g = int(round((pos.shape[1] - n_extra) ** 0.5)), int(round(x.shape[1] ** 0.5))
        if g != g0:
            grid = interpolate_pos_encoding(pos[:, n_extra:].reshape(1, g0, g0, -1), g, g)
            pos = torch.cat([pos[:, :n_extra], grid.reshape(1, g * g, -1)], dim=1)
        x = rb(torch.cat(toks + [x], dim=1) + rf(pos))
        for i in range(depth):
            if i not in skip:
                x = transformer_bf16_points(x, p, "blocks.%d." % i, heads, eps=1e-6)
        x = rb(layernorm(x, p["norm.weight"], p["norm.bias"], 1e-6))
        y = rb(linear(x[:, 0], rw(p["head.weight"]), p["head.bias"]))
        if not distilled:
            return y
        yd = rb(linear(x[:, 1], rw(p["head_dist.weight"]), p["head_dist.bias"]))
        return (y, yd) if train else (y + yd) / 2
    w = p["patch_embed.proj.weight"]
    x = F.conv2d(img, w, p["patch_embed.proj.bias"], stride=patch).flatten(2).transpose(1, 2)
    B = x.shape[0]
    toks = [p["cls_token"].expand(B, -1, -1)]
    if distilled:
        toks.append(p["dist_token"].expand(B, -1, -1))
    pos, n_extra = p["pos_embed"], len(toks)
    g0, g = int(round((pos.shape[1] - n_extra) ** 0.5)), int(round(x.shape[1] ** 0.5))
    if g != g0:     # elastic resolution (build-defined for AutoProg-DeiT): patch-grid part resized as VOLO does, models/volo.py:580-596
        grid = interpolate_pos_encoding(pos[:, n_extra:].reshape(1, g0, g0, -1), g, g)
        pos = torch.cat([pos[:, :n_extra], grid.reshape(1, g * g, -1)], dim=1)
    x = torch.cat(toks + [x], dim=1) + pos
    for i in range(depth):
        if i in skip:
            continue
        x = vit_block(x, p, "blocks.%d." % i, heads)
    x = layernorm(x, p["norm.weight"], p["norm.bias"], 1e-6)
    y = linear(x[:, 0], p["head.weight"], p["head.bias"])
    if not distilled:
        return y
    yd = linear(x[:, 1], p["head_dist.weight"], p["head_dist.bias"])
    return (y, yd) if train else (y + yd) / 2
