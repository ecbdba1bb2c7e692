import sys, torch
sys.path.insert(0, "/root/repo")
from autoprog_amd import ops
torch.manual_seed(0)
def check(name, fn):
    ref = fn()
    bad = 0
    for pat in (0x7FC07FC0, 0xFFFFFFFF, 0x7F807F80):
        for rep in range(3):
            ops.poison_lds(pat)
            out = fn()
            for a, b in zip(out, ref):
                if not bool(torch.isfinite(a.float()).all()) or not torch.allclose(a.float(), b.float(), rtol=1e-2, atol=1e-2):
                    bad += 1
    print("%-28s %s" % (name, "OK" if bad == 0 else "DIFFERS / NON-FINITE x%d" % bad))
B, N, heads, hd = 8, 196, 12, 32
qkv = (torch.randn(B * N, 3 * heads * hd, device="cuda")).to(torch.bfloat16)
do = torch.randn(B * N, heads * hd, device="cuda").to(torch.bfloat16)
o, lse = ops.mhsa_fwd(qkv, B, N, heads, hd ** -0.5)
check("mhsa_fwd N=196", lambda: ops.mhsa_fwd(qkv, B, N, heads, hd ** -0.5))
check("mhsa_bwd N=196", lambda: (ops.mhsa_bwd(qkv, o, do, lse, B, N, heads, hd ** -0.5),))
B2, N2, h2, d2 = 2, 784, 16, 48
qkv2 = torch.randn(B2 * N2, 3 * h2 * d2, device="cuda").to(torch.bfloat16); do2 = torch.randn(B2 * N2, h2 * d2, device="cuda").to(torch.bfloat16)
o2, lse2 = ops.mhsa_fwd(qkv2, B2, N2, h2, d2 ** -0.5)
check("mhsa_fwd flash 784/48", lambda: ops.mhsa_fwd(qkv2, B2, N2, h2, d2 ** -0.5))
check("mhsa_bwd flash 784/48", lambda: (ops.mhsa_bwd(qkv2, o2, do2, lse2, B2, N2, h2, d2 ** -0.5),))
for N3 in (100, 49, 144, 256):
    q3 = torch.randn(4 * N3, 3 * 6 * 32, device="cuda").to(torch.bfloat16); d3 = torch.randn(4 * N3, 6 * 32, device="cuda").to(torch.bfloat16)
    o3, l3 = ops.mhsa_fwd(q3, 4, N3, 6, 32 ** -0.5)
    check("mhsa_fwd N=%d" % N3, lambda: ops.mhsa_fwd(q3, 4, N3, 6, 32 ** -0.5))
    check("mhsa_bwd N=%d" % N3, lambda: (ops.mhsa_bwd(q3, o3, d3, l3, 4, N3, 6, 32 ** -0.5),))
# GEMMs, LN, outlook, conv
a = torch.randn(4100, 384, device="cuda").to(torch.bfloat16); w = torch.randn(1152, 384, device="cuda").to(torch.bfloat16) * 0.05
check("gemm_nt 8p ragged", lambda: (ops.gemm_nt(a, w),))
a2 = torch.randn(300, 384, device="cuda").to(torch.bfloat16)
check("gemm_nt small", lambda: (ops.gemm_nt(a2, w),))
g = torch.randn(4160, 384, device="cuda").to(torch.bfloat16); x = torch.randn(4160, 576, device="cuda").to(torch.bfloat16)
def tn():
    c = torch.zeros(384, 576, device="cuda"); ops.gemm_tn_acc(g, x, c); return (c,)
check("gemm_tn 8p", tn)
g2 = torch.randn(1000, 384, device="cuda").to(torch.bfloat16); x2 = torch.randn(1000, 576, device="cuda").to(torch.bfloat16)
def tn2():
    c = torch.zeros(384, 576, device="cuda"); ops.gemm_tn_acc(g2, x2, c); return (c,)
check("gemm_tn 128 ragged", tn2)
v = torch.randn(2, 28, 28, 192, device="cuda").to(torch.bfloat16); lg = torch.randn(2 * 196, 488, device="cuda").to(torch.bfloat16); dyo = torch.randn(2, 28, 28, 192, device="cuda").to(torch.bfloat16)
check("outlook_fwd", lambda: (ops.outlook_fwd(v, lg, 6, 32 ** -0.5),))
check("outlook_bwd", lambda: ops.outlook_bwd(v, lg, dyo, 6, 32 ** -0.5))
v2 = torch.randn(2, 27, 25, 64, device="cuda").to(torch.bfloat16); lg2 = torch.randn(2 * 14 * 13, 168, device="cuda").to(torch.bfloat16)
check("outlook_fwd odd", lambda: (ops.outlook_fwd(v2, lg2, 2, 32 ** -0.5),))
xc = torch.randn(2, 37, 21, 64, device="cuda").to(torch.bfloat16); wc = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
wf, wb = ops.conv3x3_pack(wc)
check("conv3x3", lambda: (ops.conv3x3_c64(xc, wf),))
def cw():
    dw = torch.zeros(64, 64, 3, 3, device="cuda"); ops.conv3x3_c64_wgrad(xc, xc, dw); return (dw,)
check("conv3x3 wgrad", cw)
