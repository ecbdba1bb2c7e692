"""GPU parity of every loss class of loss/cross_entropy.py against the REFERENCE golden vectors (tests/golden/loss.npz:
TokenLabelCrossEntropy with / without a mix box and with a 2-D target, TokenLabelGTCrossEntropy 3-D and 2-D,
SoftTargetCrossEntropy incl. the target-repeat path, TokenLabelSoftTargetCrossEntropy) and against the oracle on the
bf16-rounded logits.  Tolerances: vs the fp32 reference goldens the bf16 rounding of the logits (2^-9 relative on |x|~2)
bounds the loss error: <= 5e-3 abs, gradients <= 1.5e-2 rel-L2; vs the oracle evaluated on the SAME rounded logits the
kernel's fp32 statistics give <= 2e-4 abs on the loss and <= 6e-3 rel-L2 on the gradients (they are stored as bf16)."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from tests._golden import load

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


CASES = [("tl_box", "TokenLabelCrossEntropy", dict(dense_weight=0.5, cls_weight=1.0), "t3"),
         ("tl_nobox", "TokenLabelCrossEntropy", dict(dense_weight=0.5, cls_weight=1.0), "t3"),
         ("tl_2d", "TokenLabelCrossEntropy", dict(dense_weight=1.0, cls_weight=1.0), "t2"),
         ("gt_box", "TokenLabelGTCrossEntropy", dict(dense_weight=0.5, cls_weight=1.0), "t3"),
         ("gt_2d", "TokenLabelGTCrossEntropy", dict(dense_weight=0.5, cls_weight=1.0), "t2")]


@pytest.mark.parametrize("tag,cls_name,kw,tkey", CASES)
def test_token_label_losses_vs_reference_golden(tag, cls_name, kw, tkey):
    import autoprog_amd.loss as L
    d = load("loss")
    C = d["cls"].shape[1]
    fn = getattr(L, cls_name)(classes=C, **kw)
    cls = torch.from_numpy(d["cls"]).cuda().requires_grad_(True)
    aux = torch.from_numpy(d["aux"]).cuda().requires_grad_(True)
    target = torch.from_numpy(d[tkey]).cuda()
    bb = tuple(int(v) for v in d[tag + ".bbox"])
    loss = fn((cls, aux, bb), target)
    loss.backward()
    assert abs(float(loss.detach()) - float(d[tag + ".loss"])) < 5e-3, (float(loss.detach()), float(d[tag + ".loss"]))
    assert rel(cls.grad, d[tag + ".dcls"]) < 1.5e-2, rel(cls.grad, d[tag + ".dcls"])
    assert rel(aux.grad, d[tag + ".daux"]) < 1.5e-2, rel(aux.grad, d[tag + ".daux"])
    # same logits rounded to bf16 through the oracle: isolates the kernel from the input rounding
    cr = torch.from_numpy(d["cls"]).bfloat16().double().requires_grad_(True)
    ar = torch.from_numpy(d["aux"]).bfloat16().double().requires_grad_(True)
    ofn = R.token_label_gt_ce if "GT" in cls_name else R.token_label_ce
    lo = ofn((cr, ar, bb), torch.from_numpy(d[tkey]).double(), kw["dense_weight"], kw["cls_weight"])
    lo.backward()
    assert abs(float(loss.detach()) - float(lo.detach())) < 2e-4
    assert rel(cls.grad, cr.grad) < 6e-3 and rel(aux.grad, ar.grad) < 6e-3


def test_soft_target_and_token_label_soft_target_vs_reference_golden():
    from autoprog_amd.loss import SoftTargetCrossEntropy, TokenLabelSoftTargetCrossEntropy
    d = load("loss")
    x = torch.from_numpy(d["st.x"]).cuda().requires_grad_(True)          # 8 rows, target has 4: the repeat path
    loss = SoftTargetCrossEntropy()(x, torch.from_numpy(d["t2"]).cuda())
    loss.backward()
    assert abs(float(loss.detach()) - float(d["st.loss"])) < 5e-3
    assert rel(x.grad, d["st.dx"]) < 1.5e-2
    x2 = torch.from_numpy(d["tlst.x"]).cuda().requires_grad_(True)
    loss2 = TokenLabelSoftTargetCrossEntropy()(x2, torch.from_numpy(d["tlst.t"]).cuda())
    loss2.backward()
    assert abs(float(loss2.detach()) - float(d["tlst.loss"])) < 5e-3
    assert rel(x2.grad, d["tlst.dx"]) < 1.5e-2
    xr = torch.from_numpy(d["tlst.x"]).bfloat16().double().requires_grad_(True)
    lo = R.token_label_soft_target_ce(xr, torch.from_numpy(d["tlst.t"]).double())
    lo.backward()
    assert abs(float(loss2.detach()) - float(lo.detach())) < 2e-4 and rel(x2.grad, xr.grad) < 6e-3


def test_token_label_gt_full_class_count():
    """1000 classes, 196 tokens, batch 4 (the kernel's production row length) for the GT variant vs the oracle"""
    from autoprog_amd.loss import TokenLabelGTCrossEntropy
    g = torch.Generator().manual_seed(3)
    B, N, C = 4, 196, 1000
    cls = (torch.randn(B, C, generator=g) * 2).bfloat16()
    aux = (torch.randn(B, N, C, generator=g) * 2).bfloat16()
    target = torch.softmax(torch.randn(B, C, 2 + N, generator=g) * 3, dim=1)
    target[:, :, 0] = torch.nn.functional.one_hot(torch.tensor([1, 5, 7, 1]), C).float() * 0.9 + 0.1 / C
    cg = cls.cuda().requires_grad_(True)
    ag = aux.cuda().requires_grad_(True)
    loss = TokenLabelGTCrossEntropy(dense_weight=0.5, cls_weight=1.0)((cg, ag, (2, 3, 9, 11)), target.cuda())
    loss.backward()
    cr, ar = cls.double().requires_grad_(True), aux.double().requires_grad_(True)
    lo = R.token_label_gt_ce((cr, ar, (2, 3, 9, 11)), target.double(), 0.5, 1.0)
    lo.backward()
    assert abs(float(loss.detach()) - float(lo.detach())) < 5e-4
    assert rel(cg.grad, cr.grad) < 6e-3 and rel(ag.grad, ar.grad) < 6e-3


def test_sparse_token_label_ce_equals_dense_on_the_densified_target():
    """ap_soft_ce_sparse_fwd_bwd / SparseTokenLabelCEFn (the token-label target construction of
    main_prog.py:994-1004 folded into the CE kernel): the loss on top-5 (class, score) pairs + label smoothing equals the dense
    kernel on the densified [B,C,2+N] target -- loss to 1e-6 relative, both logit gradients to bf16 rounding -- with and without
    the mix-token box (lam < 1), with repeated indices in a slot, at 1000 classes; and equals the oracle's token_label_ce."""
    from autoprog_amd.loss import TokenLabelCrossEntropy, SparseTokenLabelTarget
    B, N, C, K = 6, 49, 1000, 5
    g = torch.Generator().manual_seed(7)
    idx = torch.randint(0, C, (B, 2 + N, K), generator=g)
    idx[0, 3, 1] = idx[0, 3, 0]                       # a repeated class inside one slot accumulates
    val = torch.rand(B, 2 + N, K, generator=g)
    val = val / val.sum(-1, keepdim=True)
    sp = SparseTokenLabelTarget(idx.cuda(), val.cuda(), smoothing=0.1)
    dense = sp.dense(C)
    assert dense.shape == (B, C, 2 + N) and abs(float(dense.sum(1).mean()) - 1.0) < 1e-5
    x_cls = (torch.randn(B, C, generator=g) * 2).cuda().to(torch.bfloat16)
    x_aux = (torch.randn(B, N, C, generator=g) * 2).cuda().to(torch.bfloat16)
    loss_fn = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=C)
    for bb in ((0, 0, 0, 0), (1, 2, 5, 6)):
        outs = []
        for tgt in (dense, sp):
            xc, xa = x_cls.clone().requires_grad_(True), x_aux.clone().requires_grad_(True)
            loss = loss_fn((xc, xa, bb), tgt)
            loss.backward()
            outs.append((float(loss.detach()), xc.grad.float().cpu(), xa.grad.float().cpu()))
        (l0, gc0, ga0), (l1, gc1, ga1) = outs
        assert abs(l0 - l1) <= 1e-6 * abs(l0), (l0, l1)
        assert float((gc0 - gc1).norm() / gc0.norm()) < 2e-3 and float((ga0 - ga1).norm() / ga0.norm()) < 2e-3
        ref = R.token_label_ce((x_cls.double().cpu(), x_aux.double().cpu(), bb), dense.double().cpu(), 0.5, 1.0)
        assert abs(l1 - float(ref)) < 2e-5 * abs(float(ref)), (l1, float(ref))
