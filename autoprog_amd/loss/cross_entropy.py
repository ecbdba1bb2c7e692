"""Token-labeling losses (reference loss/cross_entropy.py) on the HIP dense soft-target CE
kernel: loss and d(loss)/d(logits) come out of ONE pass over (logits, target); the class-major
token-label tensor [B,C,2+N] is consumed in place through strides (no transpose copy).
Same constructor arguments and forward contracts as the reference classes."""
import torch
import torch.nn as nn

from .. import functional as AF


def _ce_rows(x2d, target, sb, sc, sn, rows_per_batch):
    return AF.SoftTargetCEFn.apply(x2d, target, sb, sc, sn, rows_per_batch)


def _dense_ce(x, target2d):
    """mean_i(-sum_c t_ic log_softmax(x_i)_c) for a [M,C] target (rows repeated when x has more rows,
    loss/cross_entropy.py:30-36)"""
    M, C = x.shape
    t = target2d.float()
    if not t.is_contiguous():
        t = t.contiguous()
    reps = M // t.shape[0]
    # row r of x uses target row r % t.shape[0]  (target.repeat(reps, 1))
    if reps == 1:
        return _ce_rows(x, t, C, 1, 0, 1)
    losses = [_ce_rows(x[i * t.shape[0]:(i + 1) * t.shape[0]], t, C, 1, 0, 1) for i in range(reps)]
    return torch.stack(losses).mean()


class SoftTargetCrossEntropy(nn.Module):
    """reference SoftTargetCrossEntropy (loss/cross_entropy.py:21-36)"""

    def forward(self, x, target):
        return _dense_ce(x.to(torch.bfloat16), target)


class TokenLabelSoftTargetCrossEntropy(nn.Module):
    """reference TokenLabelSoftTargetCrossEntropy (loss/cross_entropy.py:92-109)"""

    def forward(self, x, target):
        if target.dim() == 3 and target.shape[-1] == 2:
            target = target[:, :, 1]
        return _dense_ce(x.to(torch.bfloat16), target)


SPARSE_CE_MAX_PAIRS, SPARSE_CE_MAX_CLASSES = 16, 1024        # CE_MAXK and 64 * 2 * CE_MAXV of csrc/softce.hip


class _TokenLabelBase(nn.Module):
    def __init__(self, dense_weight=1.0, cls_weight=1.0, mixup_active=True, classes=1000):
        super().__init__()
        self.CE = SoftTargetCrossEntropy()
        self.dense_weight = dense_weight
        self.mixup_active = mixup_active
        self.classes = classes
        self.cls_weight = cls_weight
        assert dense_weight + cls_weight > 0

    def _adjust_cls(self, target_cls, target):
        return target_cls

    def forward(self, x, target):
        output, aux_output, bb = x
        dev_box = getattr(bb, "scalars", None)                  # graph.DeviceBox: the step's box / lam live in device memory (graph replay)
        bbx1, bby1, bbx2, bby2 = (0, 0, 0, 0) if dev_box is not None else bb
        B, N, C = aux_output.shape
        if isinstance(target, SparseTokenLabelTarget):
            # the sparse kernel takes up to 16 (class, score) pairs per row -- the mix-token class row carries 2K -- and rows of up to
            # 1024 (padded) classes; anything beyond is densified and takes the dense kernels
            K = target.idx.shape[-1]
            if (type(self)._adjust_cls is _TokenLabelBase._adjust_cls and target.idx.is_cuda and target.idx.shape[1] == 2 + N
                    and 2 * K <= SPARSE_CE_MAX_PAIRS and -(-C // 8) * 8 <= SPARSE_CE_MAX_CLASSES):
                lam = dev_box if dev_box is not None else float(1 - ((bbx2 - bbx1) * (bby2 - bby1) / N))
                return AF.SparseTokenLabelCEFn.apply(output.to(torch.bfloat16), aux_output.to(torch.bfloat16), target.idx, target.val,
                                                     target.smoothing, lam, float(self.cls_weight), float(self.dense_weight))
            target = target.dense(C)
        target = target.float()
        if target.dim() == 3 and type(self)._adjust_cls is _TokenLabelBase._adjust_cls and target.is_cuda:
            # the production case (TokenLabelCrossEntropy with token labels): three launches, see functional.TokenLabelCEFn
            lam = dev_box if dev_box is not None else float(1 - ((bbx2 - bbx1) * (bby2 - bby1) / N))
            return AF.TokenLabelCEFn.apply(output.to(torch.bfloat16), aux_output.to(torch.bfloat16), target, lam,
                                           float(self.cls_weight), float(self.dense_weight))
        if dev_box is not None:
            raise NotImplementedError("a graph-mode forward (device-resident mix box) needs the fused token-label loss paths")
        aux2d = aux_output.reshape(B * N, C).to(torch.bfloat16)
        if target.dim() == 2:
            target_cls = target
            t = target.contiguous()
            loss_aux = _ce_rows(aux2d, t, C, 1, 0, N)                 # target.repeat(1,N): every token sees row b
        else:
            target_cls = self._adjust_cls(target[:, :, 1], target)
            taux = target[:, :, 2:]                                   # [B,C,N] class-major view, consumed in place
            loss_aux = _ce_rows(aux2d, taux, target.stride(0), target.stride(1), target.stride(2), N)
        lam = 1 - ((bbx2 - bbx1) * (bby2 - bby1) / N)
        if lam < 1:
            target_cls = lam * target_cls + (1 - lam) * target_cls.flip(0)
        loss_cls = _dense_ce(output.to(torch.bfloat16), target_cls)
        return self.cls_weight * loss_cls + self.dense_weight * loss_aux


class TokenLabelCrossEntropy(_TokenLabelBase):
    """reference TokenLabelCrossEntropy (loss/cross_entropy.py:112-156)"""


class SparseTokenLabelTarget:
    """The token-label target before it is densified: `idx` int32 and `val` fp32, both [B, 2 + N, K] (slot 0 ground truth, slot 1
    image level, slots 2.. tokens: SURVEY.md appendix A.2), and the label-smoothing strength.  dense() is what the reference's
    create_token_label_target hands to its loss: [B, C, 2 + N] with t = (1 - s) * scatter(val) + s / C."""

    def __init__(self, idx, val, smoothing=0.1):
        if idx.shape != val.shape or idx.dim() != 3:
            raise ValueError("SparseTokenLabelTarget: idx and val must both be [B, 2 + N, K]")
        self.idx, self.val, self.smoothing = idx.to(torch.int32), val.float(), float(smoothing)

    def dense(self, classes):
        B, S, K = self.idx.shape
        t = torch.zeros(B, classes, S, dtype=torch.float32, device=self.val.device)
        t.scatter_add_(1, self.idx.long().permute(0, 2, 1), self.val.permute(0, 2, 1))
        return t * (1.0 - self.smoothing) + self.smoothing / classes


class TokenLabelGTCrossEntropy(_TokenLabelBase):
    """reference TokenLabelGTCrossEntropy (loss/cross_entropy.py:39-89): mixes the ground truth
    slot [:,:,0] into the image-level soft label"""

    def __init__(self, dense_weight=1.0, cls_weight=1.0, mixup_active=True, smoothing=0.1, classes=1000):
        super().__init__(dense_weight, cls_weight, mixup_active, classes)
        self.smoothing = smoothing

    def _adjust_cls(self, target_cls, target):
        gt = target[:, :, 0]
        same = (gt.max(-1)[1] == target_cls.max(-1)[1])
        ratio = (0.9 - 0.4 * same.to(target_cls.dtype)).unsqueeze(-1)
        return target_cls * ratio + gt * (1 - ratio)
