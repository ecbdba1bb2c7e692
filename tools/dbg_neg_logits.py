import sys, torch
sys.path.insert(0, "/root/repo")
from autoprog_amd import ops
torch.manual_seed(0)
def case(B, N, heads, hd):
    C = heads * hd
    w = torch.nn.functional.normalize(torch.randn(1, 1, heads, hd, device="cuda"), dim=-1) * hd ** 0.5
    c = (110.0 / hd ** 0.5) ** 0.5                                                # q . k * scale ~ -110 for every pair: lse ~ -105
    q = c * w + 0.2 * torch.randn(B, N, heads, hd, device="cuda")
    k = -c * w + 0.2 * torch.randn(B, N, heads, hd, device="cuda")
    v = torch.randn(B, N, heads, hd, device="cuda")
    qkv = torch.stack([q, k, v], dim=2).reshape(B * N, 3 * C).to(torch.bfloat16)
    do = torch.randn(B * N, C, device="cuda").to(torch.bfloat16) * 0.01
    scale = hd ** -0.5
    o, lse = ops.mhsa_fwd(qkv, B, N, heads, scale)
    d = ops.mhsa_bwd(qkv, o, do, lse, B, N, heads, scale)
    qf = qkv.float().reshape(B, N, 3, heads, hd).requires_grad_(True)
    S = torch.einsum("bnhd,bmhd->bhnm", qf[:, :, 0], qf[:, :, 1]) * scale
    oref = torch.einsum("bhnm,bmhd->bnhd", torch.softmax(S, -1), qf[:, :, 2]).reshape(B * N, C)
    oref.backward(do.float())
    dref = qf.grad.reshape(B * N, 3 * C)
    fin = bool(torch.isfinite(d.float()).all())
    err = float((d.float() - dref).norm() / dref.norm()) if fin else float("nan")
    print("B %d N %d heads %d hd %d: lse in [%.1f, %.1f]  o err %.3g  dqkv finite %s  err %.3g" % (B, N, heads, hd, float(lse.min()), float(lse.max()),
          float((o.float() - oref).norm() / oref.norm()), fin, err))
for cfg in [(4, 196, 3, 32), (2, 100, 2, 32), (2, 196, 2, 64), (2, 300, 2, 32), (2, 300, 2, 48), (2, 49, 2, 32), (2, 256, 2, 32)]:
    case(*cfg)
