"""VOLO on MI355X: host-side mirror of the reference model API (models/volo.py) whose
arithmetic runs in the gfx950 HIP kernels of libautoprog_hip.so.

Same constructor signatures, parameter names (state-dict keys), forward contracts and
`set_sample_config` as the reference, so `main_prog.py` / `prog/helpers.py`-style drivers can
use it as a drop-in; elastic depth / resolution / token count are launch-time arguments of the
kernels (no weight copies): a supernet keeps all layers and a per-step `ActiveLayerMask`
selects the active ones.

Activations are bf16 token-major ([B,H,W,C] / [B,N,C]); parameters stay fp32.  The conv stem
(reference PatchEmbed) runs on the HIP convolution kernels of csrc/conv7.hip / conv.hip and the
patch-addressed GEMMs at the 64-wide stem of the BASELINE configs;
other stem widths and the fp32 debugging path use torch.nn.functional.conv2d.
"""
import math

import numpy as np
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as AF
from ..prog.helpers import ActiveLayerMask
from .registry import register_model

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)
BF16 = torch.bfloat16


def _cfg(url="", **kwargs):
    cfg = dict(url=url, num_classes=1000, input_size=(3, 224, 224), pool_size=None, crop_pct=0.96,
               interpolation="bicubic", mean=IMAGENET_DEFAULT_MEAN, std=IMAGENET_DEFAULT_STD,
               first_conv="patch_embed.proj", classifier="head")
    cfg.update(kwargs)
    return cfg


default_cfgs = {"volo": _cfg(crop_pct=0.96), "volo_large": _cfg(crop_pct=1.15)}


def trunc_normal_(t, std=0.02):
    return nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0)


def _bf16(x):
    return x if x.dtype == BF16 else x.to(BF16)


class DropPathRng:
    """source of the per-sample DropPath factors mask/keep (timm DropPath: mask = floor(keep+U)) and of their per-token keep
    masks (consumed by the bias gradients of the fused blocks).  Masks can be injected for parity runs via `queue`."""

    def __init__(self):
        self.queue = []
        self._pool = []
        self._keep_cache = {}          # (keeps, device) -> device tensor: no host->device copy per step (HIP-graph capturable)

    def prefetch(self, keeps, batch, device, tokens=0):
        """draw the factors of a whole forward pass (one entry of `keeps` per DropPath site, in call order) and, with
        `tokens` > 0, their per-token bf16 masks, with a handful of small kernels instead of several per site"""
        if self.queue or not keeps:
            self._pool = []
            return
        key = (tuple(keeps), str(device))
        k = self._keep_cache.get(key)
        if k is None:
            if len(self._keep_cache) > 64:
                self._keep_cache.clear()
            k = self._keep_cache[key] = torch.tensor(keeps, dtype=torch.float32, device=device).unsqueeze(1)
        from .. import ops
        # one uniform draw + ONE kernel for the factors, the 0/1 masks and the per-token masks of every site
        f, m, tm = ops.droppath_masks(torch.rand(len(keeps), batch, device=device), k.reshape(-1), tokens)
        self._pool = [(keeps[i], f[i], m[i], tm[i] if tm is not None else None, tokens) for i in range(len(keeps))]

    def draw(self, batch, keep, device, tokens=0):
        """-> (factors mask/keep fp32 [batch], 0/1 mask fp32 [batch] or None, per-token bf16 mask or None)"""
        if self.queue:
            m = self.queue.pop(0)
            return (m.to(device=device, dtype=torch.float32) / keep).contiguous(), None, None
        if self._pool:
            k, f, m, tm, tk = self._pool.pop(0)
            if k == keep and f.shape[0] == batch:
                return f, m, (tm if tk == tokens else None)
            self._pool = []
        return ((keep + torch.rand(batch, device=device)).floor_() / keep).contiguous(), None, None


class Mlp(nn.Module):
    """fc2(gelu(fc1(x))) -- reference Mlp (models/volo.py:147-167); dropout p is 0 on this path."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        if drop:
            raise NotImplementedError("the HIP path implements drop=0 (all shipped configs)")
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, out_features)

    def forward(self, x):
        h = AF.linear(_bf16(x), self.fc1.weight, self.fc1.bias, gelu=True)
        return AF.linear(h, self.fc2.weight, self.fc2.bias)


class OutlookAttention(nn.Module):
    """reference OutlookAttention (models/volo.py:48-103); kernel 3 / padding 1 / stride 2."""

    def __init__(self, dim, num_heads, kernel_size=3, padding=1, stride=1, qkv_bias=False, qk_scale=None,
                 attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        if (kernel_size, padding, stride) != (3, 1, 2):
            raise NotImplementedError("HIP outlook attention implements kernel 3, padding 1, stride 2 (every VOLO config)")
        if qk_scale is not None or attn_drop or proj_drop:
            raise NotImplementedError("qk_scale/attn_drop/proj_drop are not used by any shipped config")
        self.num_heads = num_heads
        self.kernel_size, self.padding, self.stride = kernel_size, padding, stride
        self.scale = (dim // num_heads) ** -0.5
        self.v = nn.Linear(dim, dim, bias=qkv_bias)
        self.attn = nn.Linear(dim, kernel_size ** 4 * num_heads)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        x = _bf16(x)
        B, H, W, C = x.shape
        v = AF.linear(x, self.v.weight, self.v.bias)
        pooled = AF.AvgPool2Fn.apply(x)
        n_out = self.attn.weight.shape[0]
        logits = AF.linear(pooled.reshape(-1, C), self.attn.weight, self.attn.bias)
        if logits.shape[-1] % 8:                      # the kernel wants 16-byte rows
            pad = torch.zeros(logits.shape[0], (-n_out) % 8, dtype=BF16, device=x.device)
            logits = torch.cat([logits, pad], dim=1)
        y = AF.OutlookCoreFn.apply(v, logits, self.num_heads)
        return AF.linear(y, self.proj.weight, self.proj.bias)


class Outlooker(nn.Module):
    """reference Outlooker (models/volo.py:106-144): fused into one forward/backward pair."""

    def __init__(self, dim, kernel_size, padding, stride=1, num_heads=1, mlp_ratio=3.0, attn_drop=0.0, drop_path=0.0,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm, qkv_bias=False, qk_scale=None):
        super().__init__()
        if drop_path:
            raise NotImplementedError("outlooker blocks never receive drop_path in the reference (SURVEY.md 0.1-8)")
        self.norm1 = nn.LayerNorm(dim)
        self.attn = OutlookAttention(dim, num_heads, kernel_size=kernel_size, padding=padding, stride=stride,
                                     qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio))
        self.is_identity_layer = False

    def set_sample_config(self, is_identity_layer=False):
        self.is_identity_layer = is_identity_layer

    def forward(self, x):
        if self.is_identity_layer:
            return x
        a, m = self.attn, self.mlp
        return AF.OutlookerBlockFn.apply(_bf16(x), self.norm1.weight, self.norm1.bias, a.v.weight, a.v.bias, a.attn.weight,
                                         a.attn.bias, a.proj.weight, a.proj.bias, self.norm2.weight, self.norm2.bias,
                                         m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias, a.num_heads, self.norm1.eps)


class Attention(nn.Module):
    """reference Attention (models/volo.py:170-201), standalone form."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        if qk_scale is not None or attn_drop or proj_drop:
            raise NotImplementedError("qk_scale/attn_drop/proj_drop are not used by any shipped config")
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        x = _bf16(x)
        shape = x.shape
        B, C = shape[0], shape[-1]
        N = x.numel() // (B * C)
        qkv = AF.linear(x.reshape(B * N, C), self.qkv.weight, self.qkv.bias)
        o = AF.MhsaFn.apply(qkv, B, N, self.num_heads)
        return AF.linear(o, self.proj.weight, self.proj.bias).reshape(shape)


class Transformer(nn.Module):
    """reference Transformer (models/volo.py:204-234): pre-LN block with per-sample DropPath,
    executed as one fused forward/backward pair."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, attn_drop=0.0, drop_path=0.0,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm, eps=1e-5):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop)
        self.drop_prob = float(drop_path)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio))
        self.is_identity_layer = False
        self.rng = None                 # DropPathRng shared by the owning model

    def set_sample_config(self, is_identity_layer=False):
        self.is_identity_layer = is_identity_layer

    def forward(self, x):
        if self.is_identity_layer:
            return x
        x = _bf16(x)
        B, C = x.shape[0], x.shape[-1]
        N = x.numel() // (B * C)
        rs1 = rs2 = k1 = k2 = tm1 = tm2 = None
        keep = 1.0 - self.drop_prob
        if self.training and self.drop_prob > 0.0:
            rng = self.rng or _default_rng
            rs1, k1, tm1 = rng.draw(B, keep, x.device, N)
            rs2, k2, tm2 = rng.draw(B, keep, x.device, N)
        a, m = self.attn, self.mlp
        return AF.TransformerBlockFn.apply(x, rs1, rs2, self.norm1.weight, self.norm1.bias, a.qkv.weight, a.qkv.bias,
                                           a.proj.weight, a.proj.bias, self.norm2.weight, self.norm2.bias, m.fc1.weight,
                                           m.fc1.bias, m.fc2.weight, m.fc2.bias, B, N, a.num_heads, self.norm1.eps,
                                           k1, k2, tm1, tm2, 1.0 / keep)


_default_rng = DropPathRng()


class ClassAttention(nn.Module):
    """reference ClassAttention (models/volo.py:237-277): token 0 queries all tokens."""

    def __init__(self, dim, num_heads=8, head_dim=None, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = head_dim if head_dim is not None else dim // num_heads
        if qk_scale is not None or attn_drop or proj_drop:
            raise NotImplementedError("qk_scale/attn_drop/proj_drop are not used by any shipped config")
        self.scale = self.head_dim ** -0.5
        inner = self.head_dim * self.num_heads
        self.kv = nn.Linear(dim, inner * 2, bias=qkv_bias)
        self.q = nn.Linear(dim, inner, bias=qkv_bias)
        self.proj = nn.Linear(inner, dim)

    def forward(self, x):
        x = _bf16(x)
        B, N, C = x.shape
        kv = AF.linear(x.reshape(B * N, C), self.kv.weight, self.kv.bias).reshape(B, N, -1)
        q = AF.linear(x[:, 0], self.q.weight, self.q.bias)
        o = AF.ClassAttnFn.apply(q, kv, self.num_heads)
        return AF.linear(o, self.proj.weight, self.proj.bias).unsqueeze(1)


class ClassBlock(nn.Module):
    """reference ClassBlock (models/volo.py:280-308): only the class token is updated."""

    def __init__(self, dim, num_heads, head_dim=None, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0.0, attn_drop=0.0,
                 drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        if drop_path or drop:
            raise NotImplementedError("class blocks use drop_path=0, drop=0 in the reference (models/volo.py:529)")
        self.norm1 = nn.LayerNorm(dim)
        self.attn = ClassAttention(dim, num_heads=num_heads, head_dim=head_dim, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                   attn_drop=attn_drop, proj_drop=drop)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio))

    def forward_split(self, cls, tokens, pass_tokens=False):
        """class token [B,C] and tokens [B,N,C] kept apart -> updated class token [B,C] (one fused forward / backward pair).
        pass_tokens: -> (class token, tokens); a caller that goes on with THAT tokens tensor instead of its own chains the token
        gradients through the blocks' LayerNorm backward kernels (functional.ClassBlockFn) instead of fanning them out for autograd to add"""
        cls, tokens = _bf16(cls), _bf16(tokens)
        B, N, C = tokens.shape
        a, m = self.attn, self.mlp
        out = AF.ClassBlockFn.apply(cls, tokens, self.norm1.weight, self.norm1.bias, a.kv.weight, a.kv.bias, a.q.weight, a.q.bias,
                                    a.proj.weight, a.proj.bias, self.norm2.weight, self.norm2.bias, m.fc1.weight, m.fc1.bias,
                                    m.fc2.weight, m.fc2.bias, B, N, a.num_heads, self.norm1.eps)
        return out if pass_tokens else out[0]

    def forward_cls(self, x):
        """x [B,1+N,C] -> updated class token [B,1,C]"""
        x = _bf16(x)
        return self.forward_split(x[:, 0], x[:, 1:]).unsqueeze(1)

    def forward(self, x):
        return torch.cat([self.forward_cls(x), x[:, 1:]], dim=1)


def get_block(block_type, **kargs):
    if block_type == "ca":
        return ClassBlock(**kargs)
    raise ValueError("unknown post block %r" % (block_type,))


def rand_bbox(size, lam, scale=1):
    """reference rand_bbox (models/volo.py:319-339).  `size` is the [B,H,W,C] token shape; the
    first box axis runs over H although it is called W there.  Uses the global numpy RNG in the
    reference's call order (randint(W) then randint(H)) so seeded runs reproduce its boxes."""
    gw = size[1] // scale
    gh = size[2] // scale
    ratio = np.sqrt(1.0 - lam)
    cw, ch = int(gw * ratio), int(gh * ratio)
    cx = np.random.randint(gw)
    cy = np.random.randint(gh)
    return (int(np.clip(cx - cw // 2, 0, gw)), int(np.clip(cy - ch // 2, 0, gh)),
            int(np.clip(cx + cw // 2, 0, gw)), int(np.clip(cy + ch // 2, 0, gh)))


class PatchEmbed(nn.Module):
    """reference PatchEmbed (models/volo.py:342-380): conv7x7/s -> BN -> ReLU -> 2x(conv3x3 -> BN ->
    ReLU) -> conv(patch/s).  bf16 NHWC on the HIP stem kernels (64-wide stem); MIOpen otherwise."""

    def __init__(self, img_size=224, stem_conv=False, stem_stride=1, patch_size=8, in_chans=3, hidden_dim=64, embed_dim=384):
        super().__init__()
        assert patch_size in [4, 8, 16]
        self.stem_conv = stem_conv
        if stem_conv:
            self.conv = nn.Sequential(
                nn.Conv2d(in_chans, hidden_dim, kernel_size=7, stride=stem_stride, padding=3, bias=False),
                nn.BatchNorm2d(hidden_dim), nn.ReLU(inplace=True),
                nn.Conv2d(hidden_dim, hidden_dim, kernel_size=3, stride=1, padding=1, bias=False),
                nn.BatchNorm2d(hidden_dim), nn.ReLU(inplace=True),
                nn.Conv2d(hidden_dim, hidden_dim, kernel_size=3, stride=1, padding=1, bias=False),
                nn.BatchNorm2d(hidden_dim), nn.ReLU(inplace=True))
        self.proj = nn.Conv2d(hidden_dim, embed_dim, kernel_size=patch_size // stem_stride, stride=patch_size // stem_stride)
        self.num_patches = (img_size // patch_size) * (img_size // patch_size)
        self.compute_dtype = BF16          # torch.float32 runs the MIOpen stem un-autocast (parity debugging)
        self.hip_conv = os.environ.get("AP_STEM_HIP_CONV", "1") == "1"    # 3x3 / 64-channel stem convolutions on csrc/conv.hip (0: MIOpen)
        self.resize_to = None              # elastic input size: a fp32 batch of another size is resized on the way in (main_prog.py:973)
        self.resize_in_eval = False        # the reference resizes in its TRAINING loop only; evaluation runs at the loader's resolution

    def forward(self, x):
        """[B,3,r,r] -> NCHW feature map (channels_last memory, bf16).  64-wide stem: resize + space-to-depth kernel, the
        7x7 / stride 2 convolution of csrc/conv7.hip, the two 3x3 convolutions of csrc/conv.hip (BatchNorm statistics in
        their epilogues, BN + ReLU by csrc/bnrelu.hip) and the patch projection as a patch-addressed GEMM.  Other widths:
        torch.nn.functional.conv2d (MIOpen, bf16 channels_last) with the same HIP BatchNorm + ReLU kernels."""
        if not x.is_cuda:
            raise RuntimeError("autoprog_amd models run on the GPU only (no CPU fallback)")
        fused = self.compute_dtype == BF16
        # the stage's resolution applies to training-mode forwards (training steps and the train-mode EMA probes of a search);
        # an eval() forward keeps the resolution it is given unless resize_in_eval is set
        size = self.resize_to if (self.resize_to and (self.training or self.resize_in_eval)) else x.shape[-1]
        first = None
        if (fused and self.stem_conv and self.hip_conv and x.dtype == torch.float32 and x.shape[-1] == x.shape[-2] and not x.requires_grad
                and size % 2 == 0 and tuple(self.conv[0].weight.shape) in ((64, 3, 7, 7), (128, 3, 7, 7)) and self.conv[0].stride[0] == 2):
            # resize + space-to-depth in one kernel, then the 7x7 / stride 2 convolution of csrc/conv7.hip with its BatchNorm + ReLU
            from .. import ops
            bn = self.conv[1]
            xs = ops.resize_bilinear_s2d16(x.contiguous(), size)
            if (AF.STEM_FUSE_BN and all(tuple(self.conv[i].weight.shape) == (64, 64, 3, 3) for i in (3, 6))
                    and all(self.conv[i].running_mean is not None for i in (1, 4, 7))):
                # the three convolutions as one node: the activations between them are applied inside the next convolution's staging
                c = self.conv
                k = self.proj.kernel_size[0]
                # the last BatchNorm + ReLU inside the projection's staging (functional.PatchConvFn bn_*): its activation is never written
                fuse_last = (AF.STEM_FUSE_BN_PROJ and self.proj.stride[0] == k and self.proj.in_channels == 64 and (xs.shape[1] % k == 0) and (xs.shape[2] % k == 0)
                             and (k * 64) % 64 == 0)
                whole = AF.Stem64Fn.apply(xs, c[0].weight, c[1].weight, c[1].bias, c[1].running_mean, c[1].running_var,
                                          c[3].weight, c[4].weight, c[4].bias, c[4].running_mean, c[4].running_var,
                                          c[6].weight, c[7].weight, c[7].bias, c[7].running_mean, c[7].running_var,
                                          self.training, (c[1].momentum, c[4].momentum, c[7].momentum), (c[1].eps, c[4].eps, c[7].eps), not fuse_last)
                if self.training:
                    torch._foreach_add_([c[i].num_batches_tracked for i in (1, 4, 7) if c[i].num_batches_tracked is not None], 1)
                if fuse_last:
                    z3, mean3, rstd3 = whole
                    return AF.PatchConvFn.apply(z3, self.proj.weight, self.proj.bias, k, mean3, rstd3, c[7].weight, c[7].bias).permute(0, 3, 1, 2)
                return self._project(whole.permute(0, 3, 1, 2), fused)
            first = AF.Conv7BNReLUFn.apply(xs, self.conv[0].weight, bn.weight, bn.bias,
                                           bn.running_mean, bn.running_var, self.training, bn.momentum, bn.eps)
            x = first.permute(0, 3, 1, 2)
        elif fused and x.dtype == torch.float32 and x.shape[-1] == x.shape[-2] and not x.requires_grad:
            # one kernel: bilinear resize to the step's resolution (identity when the sizes agree) + NCHW fp32 -> NHWC bf16
            from .. import ops
            x = ops.resize_bilinear_nhwc(x.contiguous(), size).permute(0, 3, 1, 2)      # NCHW view of channels_last memory
        else:
            if size != x.shape[-1]:
                x = F.interpolate(x.float(), size=(size, size), mode="bilinear", align_corners=False)
            # one pass: NCHW fp32 -> NHWC(channels_last) in the compute dtype (autocast would otherwise cast a second time)
            x = x.to(dtype=self.compute_dtype, memory_format=torch.channels_last)
        with torch.autocast("cuda", dtype=BF16, enabled=fused):
            if self.stem_conv:
                if fused:
                    counters = []
                    for i in (0, 3, 6):
                        conv, bn = self.conv[i], self.conv[i + 1]
                        if i == 0 and first is not None:
                            if self.training and bn.num_batches_tracked is not None:
                                counters.append(bn.num_batches_tracked)
                            continue
                        if i and tuple(conv.weight.shape) in ((64, 64, 3, 3), (128, 128, 3, 3)) and self.hip_conv:
                            # the two 3x3 convolutions at 64 channels (csrc/conv.hip) or 128 (VOLO-D4 / D5: csrc/conv128.hip): HIP implicit
                            # GEMM with the BatchNorm statistics in its epilogue
                            nhwc = AF.Conv3x3BNReLUFn.apply(x.permute(0, 2, 3, 1), conv.weight, bn.weight, bn.bias, bn.running_mean,
                                                            bn.running_var, self.training, bn.momentum, bn.eps)
                        else:
                            x = F.conv2d(x, conv.weight, None, conv.stride, conv.padding)
                            x = x.contiguous(memory_format=torch.channels_last)          # no-op when MIOpen kept NHWC
                            nhwc = AF.BNReLUFn.apply(x.permute(0, 2, 3, 1), bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                                     self.training, bn.momentum, bn.eps)
                        if self.training and bn.num_batches_tracked is not None:
                            counters.append(bn.num_batches_tracked)
                        x = nhwc.permute(0, 3, 1, 2)                                    # NCHW view, channels_last strides
                    if counters:
                        torch._foreach_add_(counters, 1)                                # one launch for the three counters
                else:
                    x = self.conv(x)
        return self._project(x, fused)

    def _project(self, x, fused):
        with torch.autocast("cuda", dtype=BF16, enabled=fused):
            k = self.proj.kernel_size[0]
            nhwc = x.permute(0, 2, 3, 1)
            if fused and self.hip_conv and self.proj.stride[0] == k and nhwc.is_contiguous() and AF.patch_conv_ok(nhwc, self.proj.weight, k):
                # the patch projection: one patch-addressed GEMM per direction (csrc/gemm.hip PATCH instantiations), no MIOpen
                x = AF.PatchConvFn.apply(nhwc, self.proj.weight, self.proj.bias, k).permute(0, 3, 1, 2)
            else:
                x = self.proj(x)
        return x


class Downsample(nn.Module):
    """reference Downsample (models/volo.py:383-396): conv k=s=patch on NHWC tokens, executed as a
    patch-gather + MFMA GEMM."""

    def __init__(self, in_embed_dim, out_embed_dim, patch_size):
        super().__init__()
        self.proj = nn.Conv2d(in_embed_dim, out_embed_dim, kernel_size=patch_size, stride=patch_size)
        self.k = patch_size

    def forward(self, x):
        x = _bf16(x)
        B, H, W, C = x.shape
        k = self.k
        h, w = H // k, W // k
        if AF.patch_conv_ok(x, self.proj.weight, k):
            return AF.PatchConvFn.apply(x, self.proj.weight, self.proj.bias, k)               # patches addressed in place
        patches = x[:, :h * k, :w * k].reshape(B, h, k, w, k, C).permute(0, 1, 3, 2, 4, 5).reshape(B, h, w, k * k * C)
        wmat = self.proj.weight.permute(0, 2, 3, 1).reshape(self.proj.weight.shape[0], k * k * C)     # (ky,kx,cin)
        return AF.linear(patches, wmat, self.proj.bias)


def _stage_dpr(drop_path_rate, block_idx, index, layers):
    return drop_path_rate * (block_idx + sum(layers[:index])) / (sum(layers) - 1)


def outlooker_blocks(block_fn, index, dim, layers, num_heads=1, kernel_size=3, padding=1, stride=1, mlp_ratio=3.0,
                     qkv_bias=False, qk_scale=None, attn_drop=0, drop_path_rate=0.0, **kwargs):
    """stage-1 builder (models/volo.py:399-417).  Like the reference, the VOLO constructor does not
    forward drop_path_rate here, so outlookers run without DropPath."""
    blocks = [block_fn(dim, kernel_size=kernel_size, padding=padding, stride=stride, num_heads=num_heads, mlp_ratio=mlp_ratio,
                       qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                       drop_path=_stage_dpr(drop_path_rate, i, index, layers) if drop_path_rate else 0.0)
              for i in range(layers[index])]
    return nn.Sequential(*blocks)


def transformer_blocks(block_fn, index, dim, layers, num_heads, mlp_ratio=3.0, qkv_bias=False, qk_scale=None, attn_drop=0,
                       drop_path_rate=0.0, **kwargs):
    """stage-2 builder (models/volo.py:420-441): linearly increasing DropPath rate per block."""
    blocks = [block_fn(dim, num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                       drop_path=_stage_dpr(drop_path_rate, i, index, layers) if drop_path_rate else 0.0)
              for i in range(layers[index])]
    return nn.Sequential(*blocks)


class VOLO(nn.Module):
    """reference VOLO (models/volo.py:444-694) with the same constructor and outputs:
    train -> (x_cls [B,classes], x_aux [B,N,classes], (bbx1,bby1,bbx2,bby2)); eval -> x_cls + 0.5*max_n x_aux."""

    def __init__(self, layers, img_size=224, in_chans=3, num_classes=1000, patch_size=8, stem_hidden_dim=64, embed_dims=None,
                 num_heads=None, downsamples=None, outlook_attention=None, mlp_ratios=None, qkv_bias=False, qk_scale=None,
                 drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0, norm_layer=nn.LayerNorm, post_layers=None,
                 return_mean=False, return_dense=True, mix_token=True, pooling_scale=2, out_kernel=3, out_stride=2, out_padding=1):
        super().__init__()
        if drop_rate or attn_drop_rate:
            raise NotImplementedError("drop_rate/attn_drop_rate are 0 in every shipped config; not implemented on the HIP path")
        self.num_classes = num_classes
        self.embed_dim = embed_dims[-1]
        self.layers = list(layers)
        self.patch_embed = PatchEmbed(stem_conv=True, stem_stride=2, patch_size=patch_size, in_chans=in_chans,
                                      hidden_dim=stem_hidden_dim, embed_dim=embed_dims[0])
        grid = img_size // patch_size // pooling_scale
        self.pos_embed = nn.Parameter(torch.zeros(1, grid, grid, embed_dims[-1]))
        self.drop_path_rng = DropPathRng()
        network = []
        for i in range(len(layers)):
            if outlook_attention[i]:
                stage = outlooker_blocks(Outlooker, i, embed_dims[i], layers, num_heads=num_heads[i], kernel_size=out_kernel,
                                         stride=out_stride, padding=out_padding, mlp_ratio=mlp_ratios[i], qkv_bias=qkv_bias,
                                         qk_scale=qk_scale, attn_drop=attn_drop_rate)
            else:
                stage = transformer_blocks(Transformer, i, embed_dims[i], layers, num_heads[i], mlp_ratio=mlp_ratios[i],
                                           qkv_bias=qkv_bias, qk_scale=qk_scale, drop_path_rate=drop_path_rate,
                                           attn_drop=attn_drop_rate)
                for blk in stage:
                    blk.rng = self.drop_path_rng
            network.append(stage)
            if downsamples[i]:
                network.append(Downsample(embed_dims[i], embed_dims[i + 1], 2))
        self.network = nn.ModuleList(network)
        self.post_network = None
        if post_layers is not None:
            self.post_network = nn.ModuleList([
                get_block(post_layers[i], dim=embed_dims[-1], num_heads=num_heads[-1], mlp_ratio=mlp_ratios[-1],
                          qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop_rate, drop_path=0.0)
                for i in range(len(post_layers))])
            self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dims[-1]))
            trunc_normal_(self.cls_token, std=0.02)
        self.return_mean = return_mean
        self.return_dense = return_dense
        if return_dense:
            assert not return_mean, "cannot return both mean and dense"
        self.mix_token = mix_token
        self.pooling_scale = pooling_scale
        if mix_token:
            self.beta = 1.0
            assert return_dense, "return all tokens if mix_token is enabled"
        if return_dense:
            self.aux_head = nn.Linear(embed_dims[-1], num_classes) if num_classes > 0 else nn.Identity()
        self.norm = nn.LayerNorm(embed_dims[-1])
        self.head = nn.Linear(embed_dims[-1], num_classes) if num_classes > 0 else nn.Identity()
        trunc_normal_(self.pos_embed, std=0.02)
        self.apply(self._init_weights)
        self.active_layers = None          # ActiveLayerMask of the current elastic config (None = all)
        self.step_scalars = None           # graph.StepScalars: the mix-token box comes from device memory (a step replayed from a HIP graph)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_embed", "cls_token"}

    def multi_use_parameters(self):
        """parameters applied more than once per forward (the final LayerNorm runs on the class token and on the tokens,
        models/volo.py:676-680): dist.GradientBucketReducer keeps them on the autograd path"""
        return [self.norm.weight, self.norm.bias] if self.post_network is not None else []

    def get_classifier(self):
        return self.head

    def reset_classifier(self, num_classes):
        self.num_classes = num_classes
        self.head = nn.Linear(self.embed_dim, num_classes) if num_classes > 0 else nn.Identity()

    # ---- elastic depth: one explicit mask object shared with extraction code
    def set_sample_config(self, config: dict):
        mask = ActiveLayerMask(config["layer_num"], config["min_layer_num"], config["max_layer_num"])
        # the reference's search configs also carry the step's resolution (main_prog.py:1824-1828); its callers resize the batch
        # themselves -- a TRAINING batch that still has another size is resized by the stem's first kernel (eval() forwards are
        # left at their own resolution: PatchEmbed.resize_in_eval)
        self.patch_embed.resize_to = config.get("input_size")
        real_stage = 0
        for stage in self.network:
            if isinstance(stage, (nn.Sequential, nn.ModuleList)):
                for idx, blk in enumerate(stage):
                    blk.set_sample_config(is_identity_layer=mask.is_identity(real_stage, idx))
                real_stage += 1
        self.active_layers = mask
        return mask

    def set_drop_path_rate(self, rate):
        """per-block DropPath rates of the CURRENT elastic config as the reference's constructor assigns them to the extracted
        network (models/volo.py:428-437: rate * (index among the active blocks) / (active blocks - 1); outlookers get none):
        lets one supernet run the progressive schedule's per-stage strengths (prog/progressive.py:4-31)"""
        stages = [s for s in self.network if isinstance(s, nn.Sequential)]
        active = [[b for b in s if not b.is_identity_layer] for s in stages]
        counts = [len(a) for a in active]
        total = sum(counts)
        for si, blocks in enumerate(active):
            for bi, blk in enumerate(blocks):
                if isinstance(blk, Transformer):
                    blk.drop_prob = float(rate) * (bi + sum(counts[:si])) / max(total - 1, 1) if rate else 0.0

    def interpolate_pos_encoding(self, x):
        """reference VOLO.interpolate_pos_encoding (models/volo.py:580-596), fp32 on the host grid"""
        h0, w0 = x.shape[1], x.shape[2]
        h, w = self.pos_embed.shape[1], self.pos_embed.shape[2]
        if h == h0 and w == w0:
            return self.pos_embed
        if self.pos_embed.is_cuda and self.pos_embed.dtype == torch.float32 and max(h, w, h0, w0) <= 64:
            # the same bicubic taps as F.interpolate(scale_factor=((h0 + 0.1) / h, (w0 + 0.1) / w)), one small HIP kernel per direction
            return AF.PosEmbedInterpFn.apply(self.pos_embed, h0, w0)
        pos = F.interpolate(self.pos_embed.permute(0, 3, 1, 2), scale_factor=((h0 + 0.1) / h, (w0 + 0.1) / w), mode="bicubic")
        assert int(w0 + 0.1) == pos.shape[-1] and int(h0 + 0.1) == pos.shape[-2]
        return pos.permute(0, 2, 3, 1)

    def forward_embeddings(self, x):
        x = self.patch_embed(x)                 # NCHW logical, channels_last memory
        return x.permute(0, 2, 3, 1).to(BF16).contiguous()     # [B,H,W,C] token-major (no copy if already NHWC)

    def forward_tokens(self, x):
        if self.training:
            keeps = []
            for stage in self.network:
                if isinstance(stage, nn.Sequential):
                    for blk in stage:
                        if isinstance(blk, Transformer) and blk.drop_prob > 0.0 and not blk.is_identity_layer:
                            keeps += [1.0 - blk.drop_prob, 1.0 - blk.drop_prob]
            # the transformer stages run behind the 2x2 Downsample: (H/2)*(W/2) tokens per image
            self.drop_path_rng.prefetch(keeps, x.shape[0], x.device, tokens=(x.shape[1] // 2) * (x.shape[2] // 2))
        for idx, block in enumerate(self.network):
            if idx == 2:
                x = AF.AddPosFn.apply(x, self.interpolate_pos_encoding(x))
            x = block(x)
        B, H, W, C = x.shape
        return x.reshape(B, H * W, C)

    def forward_cls(self, x):
        """class blocks on the (class token, tokens) pair without ever concatenating them (models/volo.py:636-642 does)"""
        B = x.shape[0]
        cls = AF.ClsExpandFn.apply(self.cls_token, B)
        for block in self.post_network:
            cls, x = block.forward_split(cls, x, pass_tokens=True)      # the tokens come back unchanged: one gradient chain, no fan-out
        return cls.unsqueeze(1), x

    def forward(self, x):
        x = self.forward_embeddings(x)
        patch_h = patch_w = 0
        dev = self.step_scalars if (self.mix_token and self.training) else None
        if dev is not None:
            # graph mode (autoprog_amd/graph.py): the host drew the box (same numpy calls, same order) and put it in device memory
            patch_h, patch_w = x.shape[1] // self.pooling_scale, x.shape[2] // self.pooling_scale
            bbx1, bby1, bbx2, bby2 = 0, 0, 0, 0
            x = AF.MixSwapDevFn.apply(x, dev, self.pooling_scale)
        elif self.mix_token and self.training:
            lam = np.random.beta(self.beta, self.beta)
            patch_h, patch_w = x.shape[1] // self.pooling_scale, x.shape[2] // self.pooling_scale
            bbx1, bby1, bbx2, bby2 = rand_bbox(x.size(), lam, scale=self.pooling_scale)
            s = self.pooling_scale
            x = AF.MixSwapFn.apply(x, s * bbx1, s * bbx2, s * bby1, s * bby2)
        else:
            bbx1, bby1, bbx2, bby2 = 0, 0, 0, 0
        x = self.forward_tokens(x)                                      # [B,N,C]
        if self.post_network is not None:
            cls, x = self.forward_cls(x)
            cls = AF.layer_norm(cls, self.norm.weight, self.norm.bias, self.norm.eps)
        else:
            cls = None
        x = AF.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        if self.return_mean:
            return AF.linear(x.float().mean(1).to(BF16), self.head.weight, self.head.bias)
        x_cls = AF.linear(cls[:, 0] if cls is not None else x[:, 0], self.head.weight, self.head.bias)
        if not self.return_dense:
            return x_cls
        tokens = x if cls is not None else x[:, 1:]
        if self.mix_token and self.training:
            # models/volo.py:684-691 swaps the AUX LOGITS of the box back between image b and B-1-b.  aux_head acts on every token by
            # itself, so swapping its INPUT tokens is the same numbers: 384 channels per token to move instead of 1000 classes
            # (19 MB against 50 MB per direction at B = 128)
            Cn = tokens.shape[-1]
            grid = tokens.reshape(tokens.shape[0], patch_h, patch_w, Cn)
            if dev is not None:
                from ..graph import DeviceBox
                grid = AF.MixSwapDevFn.apply(grid, dev, 1)
            else:
                grid = AF.MixSwapFn.apply(grid, bbx1, bbx2, bby1, bby2)
            tokens = grid.reshape(grid.shape[0], patch_h * patch_w, Cn)
            x_aux = AF.linear(tokens, self.aux_head.weight, self.aux_head.bias)
            return x_cls, x_aux, (DeviceBox(dev) if dev is not None else (bbx1, bby1, bbx2, bby2))
        x_aux = AF.linear(tokens, self.aux_head.weight, self.aux_head.bias)
        if not self.training:
            return x_cls + 0.5 * x_aux.max(1)[0]
        return x_cls, x_aux, (bbx1, bby1, bbx2, bby2)


def _build(layers, embed_dims, num_heads, mlp_ratios, cfg_key, **kwargs):
    model = VOLO(layers, embed_dims=embed_dims, num_heads=num_heads, mlp_ratios=mlp_ratios,
                 downsamples=[True, False, False, False], outlook_attention=[True, False, False, False],
                 post_layers=["ca", "ca"], **kwargs)
    model.default_cfg = default_cfgs[cfg_key]
    return model


@register_model
def volo_d1(pretrained=False, **kwargs):
    """VOLO-D1, 27M params (models/volo.py:697-727)"""
    return _build([4, 4, 8, 2], [192, 384, 384, 384], [6, 12, 12, 12], [3, 3, 3, 3], "volo", **kwargs)


@register_model
def volo_d2(pretrained=False, **kwargs):
    """VOLO-D2, 59M params (models/volo.py:730-750)"""
    return _build([6, 4, 10, 4], [256, 512, 512, 512], [8, 16, 16, 16], [3, 3, 3, 3], "volo", **kwargs)


@register_model
def volo_d3(pretrained=False, **kwargs):
    """VOLO-D3, 86M params (models/volo.py:753-773)"""
    return _build([8, 8, 16, 4], [256, 512, 512, 512], [8, 16, 16, 16], [3, 3, 3, 3], "volo", **kwargs)


@register_model
def volo_d4(pretrained=False, **kwargs):
    """VOLO-D4, 193M params (models/volo.py:776-796)"""
    return _build([8, 8, 16, 4], [384, 768, 768, 768], [12, 16, 16, 16], [3, 3, 3, 3], "volo_large", **kwargs)


@register_model
def volo_d5(pretrained=False, **kwargs):
    """VOLO-D5, 296M params, stem width 128 (models/volo.py:799-821)"""
    return _build([12, 12, 20, 4], [384, 768, 768, 768], [12, 16, 16, 16], [4, 4, 4, 4], "volo_large", stem_hidden_dim=128, **kwargs)
