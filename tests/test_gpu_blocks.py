"""Block-level parity on the GPU against golden vectors of the REAL reference modules
(tests/golden/blocks.npz, outlook_attn.npz): forward output, input gradient and every parameter
gradient of the HIP-backed modules.  Tolerance: 2e-2 rel-L2 per tensor (bf16 activations inside a
single block; the fixtures use O(1)-scale random weights)."""
import numpy as np
import pytest
import torch

from tests._golden import load, sub

pytestmark = pytest.mark.gpu
TOL = 2.5e-2


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def run(mod, d, tag, tol=TOL, reshape_in=None):
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in sub(d, tag + ".w").items()}
    mod.load_state_dict(sd, strict=True)
    mod = mod.cuda().train()
    x = torch.from_numpy(d[tag + ".x"]).cuda().to(torch.bfloat16).requires_grad_(True)
    y = mod(x)
    assert rel(y, d[tag + ".y"]) < tol, ("y", rel(y, d[tag + ".y"]))
    y.backward(torch.from_numpy(d[tag + ".dy"]).cuda().to(torch.bfloat16).reshape(y.shape))
    assert rel(x.grad, d[tag + ".dx"]) < tol, ("dx", rel(x.grad, d[tag + ".dx"]))
    errs = {n: rel(p.grad, d[tag + ".g." + n]) for n, p in mod.named_parameters()}
    bad = {k: v for k, v in errs.items() if v > tol}
    assert not bad, bad


def test_blocks_vs_reference_golden():
    from autoprog_amd.models import volo as V
    d = load("blocks")
    C, H = 64, 2
    run(V.Mlp(C, C * 3), d, "mlp")
    run(V.Attention(C, H), d, "attention")
    run(V.Attention(C, H), d, "attention_n25")
    run(V.ClassAttention(C, H), d, "class_attention")
    run(V.ClassBlock(C, H, mlp_ratio=3.0), d, "class_block")
    run(V.Outlooker(C, 3, 1, stride=2, num_heads=H, mlp_ratio=3.0), d, "outlooker")
    run(V.Transformer(C, H, mlp_ratio=3.0), d, "transformer")
    run(V.Downsample(32, 64, 2), d, "downsample")


@pytest.mark.parametrize("tag,C", [("rect6x10", 64), ("odd5x9", 32), ("even16", 64)])   # head_dim 32 fixtures
def test_outlook_attention_module_vs_reference_golden(tag, C):
    from autoprog_amd.models import volo as V
    d = load("outlook_attn")
    heads = int(d[tag + ".heads"])
    run(V.OutlookAttention(C, heads, kernel_size=3, padding=1, stride=2), d, tag)
