"""DeiT constructors (reference models/deit.py:62-179) over a from-scratch VisionTransformer with
timm 0.4.5 semantics (timm itself is not vendored by the reference: parity unpinned,
cross-checked against the pinned VOLO Transformer math).  Blocks run as the same fused HIP
forward/backward pair as VOLO's Transformer (qkv bias on, LayerNorm eps 1e-6, mlp ratio 4).

AutoProg-DeiT naming `deit_h{H}_l{L}`: heads H, embed 64*H, depth L, with
elastic depth through `set_sample_config` (single stage, by analogy with VOLO)."""
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as AF
from ..prog.helpers import get_new_layer_idx
from .registry import register_model
from .volo import Transformer, DropPathRng, trunc_normal_, IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD

BF16 = torch.bfloat16

__all__ = ["deit_tiny_patch16_224", "deit_small_patch16_224", "deit_base_patch16_224", "deit_tiny_distilled_patch16_224",
           "deit_small_distilled_patch16_224", "deit_base_distilled_patch16_224", "deit_base_patch16_384",
           "deit_base_distilled_patch16_384", "VisionTransformer", "DistilledVisionTransformer"]


def _cfg(url="", **kwargs):
    cfg = dict(url=url, num_classes=1000, input_size=(3, 224, 224), pool_size=None, crop_pct=0.9, interpolation="bicubic",
               mean=IMAGENET_DEFAULT_MEAN, std=IMAGENET_DEFAULT_STD, first_conv="patch_embed.proj", classifier="head")
    cfg.update(kwargs)
    return cfg


class PatchEmbed16(nn.Module):
    """Conv2d(k = s = patch) -> flatten -> [B, N, D], executed as patch gather + MFMA GEMM"""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size, self.patch_size = (img_size, img_size), (patch_size, patch_size)
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("autoprog_amd models run on the GPU only (no CPU fallback)")
        B, Cin, H, W = x.shape
        p = self.patch_size[0]
        h, w = H // p, W // p
        patches = x[:, :, :h * p, :w * p].reshape(B, Cin, h, p, w, p).permute(0, 2, 4, 1, 3, 5).reshape(B, h * w, Cin * p * p)
        return AF.linear(patches.to(BF16), self.proj.weight, self.proj.bias)      # weight flattens as (cin,ky,kx)


class VisionTransformer(nn.Module):
    """timm 0.4.5 VisionTransformer: patch embed, cls token, learned pos embed, `depth` pre-LN blocks with
    linearly increasing DropPath, final LayerNorm, linear head on the class token."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4.0, qkv_bias=False, qk_scale=None, representation_size=None, drop_rate=0.0, attn_drop_rate=0.0,
                 drop_path_rate=0.0, hybrid_backbone=None, norm_layer=None):
        super().__init__()
        if drop_rate or attn_drop_rate or hybrid_backbone is not None or representation_size is not None:
            raise NotImplementedError("drop/attn_drop/hybrid/representation_size are unused by the DeiT constructors")
        eps = 1e-6
        if norm_layer is not None:
            eps = norm_layer(8).eps
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.patch_embed = PatchEmbed16(img_size, patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        self.drop_path_rng = DropPathRng()
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.ModuleList([Transformer(embed_dim, num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                                 drop_path=dpr[i], eps=eps) for i in range(depth)])
        for blk in self.blocks:
            blk.rng = self.drop_path_rng
        self.norm = nn.LayerNorm(embed_dim, eps=eps)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        trunc_normal_(self.pos_embed, std=0.02)
        trunc_normal_(self.cls_token, std=0.02)
        self.apply(self._init_weights)

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

    def get_classifier(self):
        return self.head

    def reset_classifier(self, num_classes, global_pool=""):
        self.num_classes = num_classes
        self.head = nn.Linear(self.embed_dim, num_classes) if num_classes > 0 else nn.Identity()

    def set_sample_config(self, config: dict):
        """elastic depth for AutoProg-DeiT (build-defined): single stage"""
        l, lo, hi = config["layer_num"], config["min_layer_num"], config["max_layer_num"]
        fresh = get_new_layer_idx(lo, hi)
        skip = fresh if l == lo else fresh[:-(l - lo)]
        for i, blk in enumerate(self.blocks):
            blk.set_sample_config(is_identity_layer=i in skip)
        return skip

    def interpolate_pos_encoding(self, n_patches, n_extra):
        """elastic resolution for AutoProg-DeiT (BASELINE configs[3]: r in {128..224}; build-defined): the
        class / distillation embeddings are kept, the patch-grid part is resized exactly like VOLO.interpolate_pos_encoding
        (models/volo.py:580-596: bicubic, scale_factor (g+0.1)/g0)."""
        g0 = int(round((self.pos_embed.shape[1] - n_extra) ** 0.5))
        g = int(round(n_patches ** 0.5))
        if g == g0:
            return self.pos_embed
        D = self.pos_embed.shape[-1]
        if self.pos_embed.is_cuda and self.pos_embed.dtype == torch.float32 and max(g, g0) <= 64:
            # the same bicubic taps on the HIP resampling kernel (functional.PosEmbedInterpFn), forward and backward
            grid = AF.PosEmbedInterpFn.apply(self.pos_embed[:, n_extra:].reshape(1, g0, g0, D), g, g)
            return torch.cat([self.pos_embed[:, :n_extra], grid.reshape(1, g * g, D)], dim=1)
        grid = self.pos_embed[:, n_extra:].reshape(1, g0, g0, D).permute(0, 3, 1, 2)
        grid = F.interpolate(grid, scale_factor=((g + 0.1) / g0, (g + 0.1) / g0), mode="bicubic")
        assert grid.shape[-1] == g and grid.shape[-2] == g
        return torch.cat([self.pos_embed[:, :n_extra], grid.permute(0, 2, 3, 1).reshape(1, g * g, D)], dim=1)

    def _tokens(self, x, extra):
        B = x.shape[0]
        x = self.patch_embed(x)
        n_patches = x.shape[1]
        toks = [self.cls_token.expand(B, -1, -1).to(BF16)] + [t.expand(B, -1, -1).to(BF16) for t in extra]
        x = torch.cat(toks + [x], dim=1)
        pos = self.interpolate_pos_encoding(n_patches, 1 + len(extra))
        return AF.AddPosFn.apply(x.unsqueeze(1), pos.unsqueeze(1)).squeeze(1)

    def _prefetch_drop_path(self, x):
        """one uniform draw + ONE kernel for the DropPath factors and per-token masks of every site of this forward pass (as VOLO.forward_tokens
        does): without it each of the 2 x depth sites draws its own through ~7 torch kernels -- 170 launches, 7 % of a DeiT-Base step"""
        if self.training:
            keeps = []
            for blk in self.blocks:
                if blk.drop_prob > 0.0 and not blk.is_identity_layer:
                    keeps += [1.0 - blk.drop_prob, 1.0 - blk.drop_prob]
            self.drop_path_rng.prefetch(keeps, x.shape[0], x.device, tokens=x.shape[1])

    def forward_features(self, x):
        x = self._tokens(x, [])
        self._prefetch_drop_path(x)
        for blk in self.blocks:
            x = blk(x)
        x = AF.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return x[:, 0]

    def forward(self, x):
        return AF.linear(self.forward_features(x), self.head.weight, self.head.bias)


class DistilledVisionTransformer(VisionTransformer):
    """reference DistilledVisionTransformer (models/deit.py:20-59): extra dist token and head"""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.dist_token = nn.Parameter(torch.zeros(1, 1, self.embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 2, self.embed_dim))
        self.head_dist = nn.Linear(self.embed_dim, self.num_classes) if self.num_classes > 0 else nn.Identity()
        trunc_normal_(self.dist_token, std=0.02)
        trunc_normal_(self.pos_embed, std=0.02)
        self.head_dist.apply(self._init_weights)

    def forward_features(self, x):
        x = self._tokens(x, [self.dist_token])
        self._prefetch_drop_path(x)
        for blk in self.blocks:
            x = blk(x)
        x = AF.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return x[:, 0], x[:, 1]

    def forward(self, x):
        x, x_dist = self.forward_features(x)
        x = AF.linear(x, self.head.weight, self.head.bias)
        x_dist = AF.linear(x_dist, self.head_dist.weight, self.head_dist.bias)
        if self.training:
            return x, x_dist
        return (x + x_dist) / 2


def _deit(cls, embed_dim, heads, pretrained, img_size=224, **kwargs):
    if pretrained:
        raise RuntimeError("pretrained DeiT weights are not downloadable in this environment")
    model = cls(img_size=img_size, patch_size=16, embed_dim=embed_dim, depth=12, num_heads=heads, mlp_ratio=4, qkv_bias=True,
                norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
    model.default_cfg = _cfg()
    return model


@register_model
def deit_tiny_patch16_224(pretrained=False, **kwargs):
    return _deit(VisionTransformer, 192, 3, pretrained, **kwargs)


@register_model
def deit_small_patch16_224(pretrained=False, **kwargs):
    return _deit(VisionTransformer, 384, 6, pretrained, **kwargs)


@register_model
def deit_base_patch16_224(pretrained=False, **kwargs):
    return _deit(VisionTransformer, 768, 12, pretrained, **kwargs)


@register_model
def deit_tiny_distilled_patch16_224(pretrained=False, **kwargs):
    return _deit(DistilledVisionTransformer, 192, 3, pretrained, **kwargs)


@register_model
def deit_small_distilled_patch16_224(pretrained=False, **kwargs):
    return _deit(DistilledVisionTransformer, 384, 6, pretrained, **kwargs)


@register_model
def deit_base_distilled_patch16_224(pretrained=False, **kwargs):
    return _deit(DistilledVisionTransformer, 768, 12, pretrained, **kwargs)


@register_model
def deit_base_patch16_384(pretrained=False, **kwargs):
    return _deit(VisionTransformer, 768, 12, pretrained, img_size=384, **kwargs)


@register_model
def deit_base_distilled_patch16_384(pretrained=False, **kwargs):
    return _deit(DistilledVisionTransformer, 768, 12, pretrained, img_size=384, **kwargs)


def deit_variant(h, l, **kwargs):
    """`deit_h{H}_l{L}`: heads H, embed 64*H, depth L (DeiT-T h3, S h6, B h12), SURVEY.md row D3"""
    kwargs.pop("pretrained", None)
    model = VisionTransformer(patch_size=16, embed_dim=64 * h, depth=l, num_heads=h, mlp_ratio=4, qkv_bias=True,
                              norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
    model.default_cfg = _cfg()
    return model
