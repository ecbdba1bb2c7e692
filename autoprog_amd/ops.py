"""Tensor-level wrappers over the C ABI (include/autoprog_hip.h).

Every function takes/returns torch CUDA tensors, allocates outputs with torch (device memory
and streams are PyTorch plumbing), and enqueues the gfx950 kernel on the current stream.
There is no fallback path: a CPU tensor or a missing library raises.
"""
import ctypes
import math
import os

import torch

from ._lib import GemmEpilogue, AutoProgHipError, check, lib

BF16 = torch.bfloat16


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """the current stream's handle.  torch.cuda.current_stream() builds a Stream object through five Python layers (9 us a call,
    ~4 ms of host time per training step at ~440 launches -- the launching thread was at 82 % of the GPU's step time); the raw
    accessor the torch compiler stack uses costs 0.3 us and follows torch.cuda.stream() contexts the same way."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _req_fail(t, dtype, name):
    if not (torch.is_tensor(t) and t.is_cuda):
        raise AutoProgHipError("%s must be a CUDA tensor (the HIP path has no CPU fallback)" % name)
    if t.dtype != dtype:
        raise AutoProgHipError("%s must be %s, got %s" % (name, dtype, t.dtype))
    raise AutoProgHipError("%s must be contiguous" % name)


def _req(t, dtype, name):
    """every tensor that goes to the library: CUDA, the kernel's dtype, contiguous (one expression on the hot path: ~1900 calls per step)"""
    try:
        ok = t.dtype is dtype and t.is_cuda and t.is_contiguous()
    except AttributeError:
        ok = False
    if not ok:
        _req_fail(t, dtype, name)
    return t


def round_up(v, m):
    return (v + m - 1) // m * m


# ------------------------------------------------------------------------------- calibration probes (bench.py `calibration`)
def calib_copy(src, dst):
    """dst <- src, 16 bytes per lane (the float4 copy the HBM figure of MI355X_MICROARCH.md is quoted on)"""
    if not (src.is_cuda and dst.is_cuda and src.is_contiguous() and dst.is_contiguous()) or src.numel() * src.element_size() != dst.numel() * dst.element_size():
        raise AutoProgHipError("calib_copy: two contiguous CUDA tensors of equal size")
    check(lib.ap_calib_copy(src.data_ptr(), dst.data_ptr(), src.numel() * src.element_size(), _stream()), "ap_calib_copy")


def calib_mfma(seed, sink, iters):
    """register-only bf16 MFMA loop on every CU; -> FLOP of the launch"""
    _req(seed, BF16, "seed")
    _req(sink, torch.float32, "sink")
    if seed.numel() < 2048:
        raise AutoProgHipError("calib_mfma: seed holds at least 2048 bf16")
    check(lib.ap_calib_mfma(seed.data_ptr(), sink.data_ptr(), int(iters), _stream()), "ap_calib_mfma")
    return 256.0 * 4 * iters * 16 * 16384


# ------------------------------------------------------------------------------------ casts
def cast_bf16(src):
    _req(src, torch.float32, "src")
    dst = torch.empty(src.shape, dtype=BF16, device=src.device)
    check(lib.ap_cast_f32_bf16(src.data_ptr(), dst.data_ptr(), src.numel(), _stream()), "ap_cast_f32_bf16")
    return dst


def cast_f32(src):
    _req(src, BF16, "src")
    dst = torch.empty(src.shape, dtype=torch.float32, device=src.device)
    check(lib.ap_cast_bf16_f32(src.data_ptr(), dst.data_ptr(), src.numel(), _stream()), "ap_cast_bf16_f32")
    return dst


def cast_transpose_bf16(w):
    """fp32 [rows, cols] -> bf16 [cols, round_up(rows, 8)] (pad columns zero)."""
    _req(w, torch.float32, "w")
    rows, cols = w.shape
    ld = round_up(rows, 8)
    dst = torch.empty((cols, ld), dtype=BF16, device=w.device)
    check(lib.ap_cast_transpose_f32_bf16(w.data_ptr(), dst.data_ptr(), rows, cols, ld, _stream()), "ap_cast_transpose_f32_bf16")
    return dst


def resize_bilinear_nhwc(x, size):
    """fp32 NCHW batch -> bf16 NHWC batch bilinearly resized to (size, size) (F.interpolate(..., mode='bilinear', align_corners=False),
    main_prog.py:973); size equal to the input size is the plain layout change + cast"""
    _req(x, torch.float32, "x")
    B, C, Hi, Wi = x.shape
    y = torch.empty((B, size, size, C), dtype=BF16, device=x.device)
    check(lib.ap_resize_bilinear_nhwc(x.data_ptr(), y.data_ptr(), B, C, Hi, Wi, size, size, _stream()), "ap_resize_bilinear_nhwc")
    return y


def droppath_masks(uniform, keep, tokens=0):
    """uniform fp32 [sites,B], keep fp32 [sites] -> (factor [sites,B], mask [sites,B], token_mask bf16 [sites,row] or None): one launch"""
    _req(uniform, torch.float32, "uniform"); _req(keep, torch.float32, "keep")
    sites, B = uniform.shape
    factor = torch.empty_like(uniform)
    mask = torch.empty_like(uniform)
    row = round_up(B * tokens, 8) if tokens else 0
    tm = torch.empty((sites, row), dtype=BF16, device=uniform.device) if tokens else None
    check(lib.ap_droppath_masks(uniform.data_ptr(), keep.data_ptr(), factor.data_ptr(), mask.data_ptr(), tm.data_ptr() if tm is not None else None,
                                sites, B, int(tokens), row, _stream()), "ap_droppath_masks")
    return factor, mask, tm


# -------------------------------------------------------------------------------- layernorm
def layernorm_fwd(x, gamma, beta, eps, fp8=None):
    """-> (y, mean, rstd); fp8 = (scale, amax) device scalars: -> (y, mean, rstd, y8) with y8 = e4m3 bytes of y * scale[0] and
    amax[0] raised to max |y| (what quantize_fp8(y, scale, amax) returns, without its pass over y)"""
    _req(x, BF16, "x"); _req(gamma, torch.float32, "gamma"); _req(beta, torch.float32, "beta")
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    if fp8 is not None:
        scale, amax = fp8
        y8 = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        check(lib.ap_layernorm_fwd_fp8(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), y8.data_ptr(), scale.data_ptr(),
                                       amax.data_ptr() if amax is not None else None, mean.data_ptr(), rstd.data_ptr(),
                                       rows, C, float(eps), _stream()), "ap_layernorm_fwd_fp8")
        return y, mean, rstd, y8
    check(lib.ap_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                               rows, C, float(eps), _stream()), "ap_layernorm_fwd")
    return y, mean, rstd


_ln_ws = {}          # (rows, C) -> ap_layernorm_bwd_workspace bytes (a pure function of the shape: one foreign call per shape, not per launch)


def layernorm_bwd(dy, x, gamma, mean, rstd, dres, dgamma, dbeta, defer=None, pool=None):
    """dx = dres + dLN/dx ; dgamma/dbeta (fp32) are accumulated in place.  defer: a list -- the dgamma/dbeta reduction is not
    launched but appended to it (layernorm_bwd_reduce_batched reduces the LayerNorms of a block in one launch).
    pool = (dpooled [B,h,w,C], (B, H, W)) with defer: the LayerNorm output also fed a 2 x 2 ceil-mode average pool whose output gradient is
    dpooled -- its backward is applied to dy inside the kernel; -> None where that kernel does not apply (the caller falls back)"""
    _req(dy, BF16, "dy"); _req(x, BF16, "x")
    C = x.shape[-1]
    rows = x.numel() // C
    dx = torch.empty_like(x)
    ws_bytes = _ln_ws.get((rows, C))
    if ws_bytes is None:
        ws_bytes = _ln_ws[(rows, C)] = lib.ap_layernorm_bwd_workspace(rows, C)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
    if pool is not None:
        dp, (B_, H_, W_) = pool
        _req(dp, BF16, "dpooled")
        if defer is None or B_ * H_ * W_ != rows:
            raise AutoProgHipError("layernorm_bwd(pool=...) needs defer and a [B,H,W,C] token grid")
        n = ctypes.c_int(0)
        rc = lib.ap_layernorm_bwd_partial_pool(dy.data_ptr(), dp.data_ptr(), B_, H_, W_, x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                               dres.data_ptr() if dres is not None else None, dx.data_ptr(), C, ws.data_ptr(), ws_bytes, ctypes.byref(n), _stream())
        if rc == -2:                   # AP_ERR_UNSUPPORTED
            return None
        check(rc, "ap_layernorm_bwd_partial_pool")
        defer.append((ws, n.value, C, dgamma, dbeta))
        return dx
    if defer is not None and rows > 0:
        n = ctypes.c_int(0)
        check(lib.ap_layernorm_bwd_partial(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                           dres.data_ptr() if dres is not None else None, dx.data_ptr(), rows, C, ws.data_ptr(), ws_bytes,
                                           ctypes.byref(n), _stream()), "ap_layernorm_bwd_partial")
        defer.append((ws, n.value, C, dgamma, dbeta))
        return dx
    check(lib.ap_layernorm_bwd(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                               dres.data_ptr() if dres is not None else None, dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                               rows, C, ws.data_ptr(), ws_bytes, _stream()), "ap_layernorm_bwd")
    return dx


def layernorm_bwd_reduce_batched(items):
    """items: the entries layernorm_bwd(..., defer=items) appended; one launch per LN_MAX_BATCH LayerNorms"""
    from ._lib import LnReduce, LN_MAX_BATCH
    for i0 in range(0, len(items), LN_MAX_BATCH):
        chunk = items[i0:i0 + LN_MAX_BATCH]
        arr = (LnReduce * len(chunk))()
        for q, (ws, n, C, dg, db) in zip(arr, chunk):
            q.partial, q.n_partial, q.C, q.dgamma, q.dbeta = ws.data_ptr(), n, C, dg.data_ptr(), db.data_ptr()
        check(lib.ap_layernorm_bwd_reduce_batched(ctypes.cast(arr, ctypes.c_void_p), len(chunk), _stream()), "ap_layernorm_bwd_reduce_batched")


# ------------------------------------------------------------------------------------- gemm
GELU_CODE_SCALE, GELU_CODE_ZERO = 202.0, 26.0       # the 8-bit derivative codes of gelu = 3 (include/autoprog_hip.h): gelu' = (code - 26) / 202


def _gelu_mode(gelu, preact_grad, preact_out):
    """ap_gemm_epilogue.gelu: 1 stores h, 2 gelu'(h) as bf16, 3 gelu'(h) as 8-bit codes (preact_out then is a uint8 tensor)"""
    if not gelu:
        return 0
    mode = 1 + int(preact_grad)
    if mode == 3 and (preact_out is None or preact_out.dtype != torch.uint8):
        raise AutoProgHipError("gemm_nt: preact_grad = 2 stores 8-bit codes: preact_out must be a uint8 tensor")
    if mode != 3 and preact_out is not None and preact_out.dtype != BF16:
        raise AutoProgHipError("gemm_nt: preact_out must be bf16 (uint8 only with preact_grad = 2)")
    return mode


def mlp_fused_ok(M, C, hidden):
    """does ap_mlp_fused take this MLP?  (csrc/mlp_fused.hip: C = 384, hidden = 3 C, whole 128-row blocks)"""
    return C == 384 and hidden == 3 * C and M % 128 == 0 and M > 0


def mlp_fused(x, wa, wb, backward=False, bias1=None, bias2=None, row_scale_hidden=None, row_scale_out=None, rows_per_scale=1,
              residual=None, codes=None, ln=None):
    """the MLP of a block in one launch (include/autoprog_hip.h ap_mlp_fused).
    forward : x [M, C], wa = fc1 weight [H, C], wb = fc2 weight [C, H] -> (out [M, C], a [M, H] = gelu(.) * row_scale_hidden, codes [M, H] uint8)
    backward: x = dL/dout, wa = fc2 weight^T copy [H, ld(C)], wb = fc1 weight^T copy [C, ld(H)], codes = the forward's -> (dL/dx [M, C], dL/dh [M, H], codes)
    ln = (rows [M, C], gamma, beta, eps) (forward; x = None): the LayerNorm in front of fc1 runs inside the launch, bit-identical to
    layernorm_fwd -> (out, a, codes, LN(rows), mean, rstd)
    -> None when the library does not take the launch (the caller issues the two ap_gemm_nt launches)"""
    from ._lib import MlpFusedArgs
    if ln is not None:
        x = _req(ln[0], BF16, "ln rows")
    _req(x, BF16, "x"); _req(wa, BF16, "wa"); _req(wb, BF16, "wb")
    M, C = x.shape
    H = wa.shape[0]
    if not mlp_fused_ok(M, C, H):
        return None
    out = torch.empty((M, C), dtype=BF16, device=x.device)
    hid = torch.empty((M, H), dtype=BF16, device=x.device)
    if backward:
        _req(codes, torch.uint8, "codes")
    else:
        codes = torch.empty((M, H), dtype=torch.uint8, device=x.device)
    a = MlpFusedArgs()
    a.x, a.ldx = x.data_ptr(), x.shape[1]
    a.wa, a.ldwa = wa.data_ptr(), wa.shape[1]
    a.wb, a.ldwb = wb.data_ptr(), wb.shape[1]
    a.out, a.ldo = out.data_ptr(), C
    a.hidden_out, a.ldh = hid.data_ptr(), H
    a.codes = codes.data_ptr()
    a.bias1 = _req(bias1, torch.float32, "bias1").data_ptr() if bias1 is not None else None
    a.bias2 = _req(bias2, torch.float32, "bias2").data_ptr() if bias2 is not None else None
    a.row_scale_hidden = _req(row_scale_hidden, torch.float32, "row_scale_hidden").data_ptr() if row_scale_hidden is not None else None
    a.row_scale_out = _req(row_scale_out, torch.float32, "row_scale_out").data_ptr() if row_scale_out is not None else None
    a.rows_per_scale = int(rows_per_scale)
    if residual is not None:
        _req(residual, BF16, "residual")
        a.residual, a.ldr = residual.data_ptr(), residual.shape[1]
    a.m, a.c, a.hidden, a.backward = M, C, H, 1 if backward else 0
    if ln is not None:
        xn = torch.empty((M, C), dtype=BF16, device=x.device)
        mean = torch.empty(M, dtype=torch.float32, device=x.device)
        rstd = torch.empty(M, dtype=torch.float32, device=x.device)
        a.x = None
        a.ln_in, a.ld_ln, a.ln_out, a.ld_lno = x.data_ptr(), C, xn.data_ptr(), C
        a.ln_gamma, a.ln_beta = _req(ln[1], torch.float32, "ln gamma").data_ptr(), _req(ln[2], torch.float32, "ln beta").data_ptr()
        a.ln_eps, a.ln_mean, a.ln_rstd = float(ln[3]), mean.data_ptr(), rstd.data_ptr()
    code = lib.ap_mlp_fused(ctypes.byref(a), _stream())
    if code == -2:                    # AP_ERR_UNSUPPORTED (e.g. the GELU table cannot be built inside a stream capture)
        return None
    check(code, "ap_mlp_fused")
    if ln is not None:
        return out, hid, codes, xn, mean, rstd
    return out, hid, codes


def gemm_nt_emits_q8(M, N, K, mul_by):
    """can a bf16 gemm_nt launch of these dimensions write its output a second time as e4m3 (q8=)?  The mul_by8 flavours of the 8-phase
    kernel (the input gradient of fc2): see ap_gemm_nt / use_8p() in csrc/gemm.hip"""
    return (mul_by is not None and mul_by.dtype == torch.uint8 and K % 64 == 0 and K >= 128 and M >= 4096 and N % 16 == 0
            and (N >= 1024 or N % 192 == 0 or N % 256 == 0) and os.environ.get("AP_GEMM_8P", "1") != "0")


def gemm_nt(a, b, n=None, k=None, bias=None, gelu=False, preact_out=None, dgelu_of=None, row_scale=None,
            rows_per_scale=1, residual=None, out=None, ldc=None, preact_grad=False, mul_by=None, q8=None):
    """out[M, :n] = epilogue(a[M, :k] @ b[:n, :k]^T); a/b bf16 2-D (row stride = shape[1]).
    preact_grad: with gelu, preact_out receives gelu'(h) instead of h (True / 1: bf16; 2: 8-bit codes, preact_out uint8); its backward
    passes that tensor as mul_by (a uint8 tensor is taken as the codes).
    q8 = (scale, amax) (gemm_nt_emits_q8 launches): -> (out, out8), out8 = the e4m3 bytes of out * scale[0], amax[0] raised to max |out|"""
    _req(a, BF16, "a"); _req(b, BF16, "b")
    M = a.shape[0]
    n = b.shape[0] if n is None else n
    k = a.shape[1] if k is None else k
    if out is None:
        ldc = round_up(n, 8) if ldc is None else ldc
        out = torch.empty((M, ldc), dtype=BF16, device=a.device)
    else:
        ldc = out.shape[1]
    if preact_out is not None and not gelu:
        raise AutoProgHipError("gemm_nt: preact_out is the GELU launches' side output (pass gelu=True)")
    if bias is None and not gelu and dgelu_of is None and mul_by is None and row_scale is None and residual is None:
        epi_ref = None                                            # plain product: the library takes a null epilogue
    else:
        epi = GemmEpilogue()
        epi.bias = bias.data_ptr() if bias is not None else None
        epi.gelu = _gelu_mode(gelu, preact_grad, preact_out)
        epi.preact_out = preact_out.data_ptr() if preact_out is not None else None
        epi.dgelu_of = dgelu_of.data_ptr() if dgelu_of is not None else None
        if mul_by is not None and mul_by.dtype == torch.uint8:
            if mul_by.shape[-1] != ldc or not mul_by.is_contiguous():
                raise AutoProgHipError("gemm_nt: the derivative codes must be contiguous [M, ldc] bytes")
            epi.mul_by, epi.mul_by8 = None, mul_by.data_ptr()
        else:
            epi.mul_by = mul_by.data_ptr() if mul_by is not None else None
        epi.row_scale = row_scale.data_ptr() if row_scale is not None else None
        epi.rows_per_scale = int(rows_per_scale)
        epi.residual = residual.data_ptr() if residual is not None else None
        epi.ldr = residual.shape[1] if residual is not None else 0
        epi_ref = ctypes.byref(epi)
    out8 = None
    if q8 is not None:
        if epi_ref is None:
            raise AutoProgHipError("gemm_nt: q8 exists for the mul_by (8-bit codes) launches only")
        out8 = torch.empty((M, ldc), dtype=torch.uint8, device=a.device)
        epi.q8_out, epi.q8_scale, epi.q8_amax = out8.data_ptr(), q8[0].data_ptr(), (q8[1].data_ptr() if q8[1] is not None else None)
    check(lib.ap_gemm_nt(a.data_ptr(), a.shape[1], b.data_ptr(), b.shape[1], out.data_ptr(), ldc, M, n, k,
                         epi_ref, _stream()), "ap_gemm_nt")
    return out if q8 is None else (out, out8)


FP8_MAX = 448.0          # OCP e4m3


def quantize_fp8(x, scale, amax=None):
    """bf16 tensor (numel % 16 == 0) -> uint8 tensor of e4m3 bytes, y = sat(x * scale[0]); scale / amax: fp32 device scalars
    (amax[0] = max(amax[0], max|x|): the statistic the NEXT step's scale is derived from -- delayed scaling)"""
    _req(x, BF16, "x"); _req(scale, torch.float32, "scale")
    y = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    check(lib.ap_quantize_fp8(x.data_ptr(), y.data_ptr(), x.numel(), scale.data_ptr(), amax.data_ptr() if amax is not None else None,
                              _stream()), "ap_quantize_fp8")
    return y


def quantize_fp8_multi(table, njobs, scales, amax):
    """table: int64 [njobs, 4] device tensor of ap_fp8_job records (x pointer, y pointer, n, slot); one launch for all of them"""
    _req(table, torch.int64, "table"); _req(scales, torch.float32, "scales")
    check(lib.ap_quantize_fp8_multi(table.data_ptr(), int(njobs), scales.data_ptr(), amax.data_ptr() if amax is not None else None, _stream()),
          "ap_quantize_fp8_multi")


def poison_lds(pattern=0x7FC07FC0):
    """test aid: every CU's LDS filled with `pattern` (default: bf16 NaN pairs)"""
    scratch = torch.zeros(2, dtype=torch.int32, device="cuda")
    check(lib.ap_debug_poison_lds(int(pattern) & 0xFFFFFFFF, scratch.data_ptr(), _stream()), "ap_debug_poison_lds")
    return scratch


def quantize_fp8_now(x):
    """current scaling (one extra pass for the amax): -> (bytes, dequantisation factor as a device scalar)"""
    amax = x.abs().amax().float().clamp_min(1e-12).reshape(1)
    return quantize_fp8(x, (FP8_MAX / amax).contiguous()), (amax / FP8_MAX).contiguous()


def gemm_nt_fp8_emits(M, N, K):
    """can this launch also emit its GELU output as e4m3 (q8)?  -- the 8-phase kernel's launches: see use_8p() in csrc/gemm.hip"""
    return K % 128 == 0 and K >= 256 and M >= 4096 and N % 8 == 0 and N >= 192 and (N >= 1024 or N % 192 == 0 or (N % 256 == 0 and N >= 192)) and os.environ.get("AP_GEMM_8P", "1") != "0"


def gemm_nt_fp8(a8, b8, dq_a, dq_b, n=None, bias=None, gelu=False, preact_out=None, residual=None, row_scale=None, rows_per_scale=1,
                preact_grad=False, q8=None):
    """out[M, :n] = epilogue(dq_a * dq_b * a8 @ b8[:n]^T): a8 [M,K], b8 [>=n,K] uint8 e4m3 bytes (K % 16 == 0), bf16 output;
    the epilogue arguments as for gemm_nt.  q8 = (scale, amax) with gelu: -> (out, out8), out8 the e4m3 bytes of out * scale[0]"""
    _req(a8, torch.uint8, "a8"); _req(b8, torch.uint8, "b8")
    M, K = a8.shape
    n = b8.shape[0] if n is None else n
    ldc = round_up(n, 8)
    out = torch.empty((M, ldc), dtype=BF16, device=a8.device)
    epi = GemmEpilogue()
    epi.bias = bias.data_ptr() if bias is not None else None
    epi.gelu = _gelu_mode(gelu, preact_grad, preact_out)
    epi.preact_out = preact_out.data_ptr() if preact_out is not None else None
    epi.dgelu_of = None
    epi.mul_by = None
    epi.row_scale = row_scale.data_ptr() if row_scale is not None else None
    epi.rows_per_scale = int(rows_per_scale)
    epi.residual = residual.data_ptr() if residual is not None else None
    epi.ldr = residual.shape[1] if residual is not None else 0
    out8 = None
    if q8 is not None:
        out8 = torch.empty((M, ldc), dtype=torch.uint8, device=a8.device)
        epi.q8_out, epi.q8_scale, epi.q8_amax = out8.data_ptr(), q8[0].data_ptr(), (q8[1].data_ptr() if q8[1] is not None else None)
    check(lib.ap_gemm_nt_fp8(a8.data_ptr(), K, b8.data_ptr(), b8.shape[1], out.data_ptr(), ldc, M, n, K, dq_a.data_ptr(), dq_b.data_ptr(),
                             ctypes.byref(epi), _stream()), "ap_gemm_nt_fp8")
    return out if q8 is None else (out, out8)


def gemm_tn_acc(a, b, c, n1=None, n2=None, colsum=None):
    """c[:n1, :n2] += a[:, :n1]^T @ b[:, :n2]   (c fp32, accumulated); optionally colsum[:n1] += a.sum(0)."""
    _req(a, BF16, "a"); _req(b, BF16, "b"); _req(c, torch.float32, "c")
    n1 = c.shape[0] if n1 is None else n1
    n2 = c.shape[1] if n2 is None else n2
    check(lib.ap_gemm_tn_acc(a.data_ptr(), a.shape[1], b.data_ptr(), b.shape[1], c.data_ptr(), c.shape[1], a.shape[0], n1, n2,
                             colsum.data_ptr() if colsum is not None else None, _stream()), "ap_gemm_tn_acc")
    return c


# AP_DETERMINISTIC=1 (or ops.deterministic = True): weight gradients through stored partial tiles + an ordered reduce instead of
# fp32 atomics -- bitwise reproducible steps (loss-curve pins, debugging) for a few percent of the backward time.
deterministic = os.environ.get("AP_DETERMINISTIC", "0") == "1"


def patch_map(H, W, C, k):
    """ap_patch_map of a k x k / stride k convolution on a contiguous NHWC [B,H,W,C] feature map (H, W multiples of k)"""
    from ._lib import PatchMap
    if H % k or W % k:
        raise AutoProgHipError("patch addressing needs H, W divisible by the patch size (got %dx%d / %d)" % (H, W, k))
    return PatchMap(W // k, k * W * C, k * C, k * C, W * C)


def gemm_nt_patch_fwd(x, wmat, bias, k, bn_in=None):
    """x [B,H,W,C] bf16 NHWC, wmat [N, k*k*C] bf16 (columns ordered (dy, dx, c)) -> [B*(H/k)*(W/k), round_up(N, 8)] bf16:
    the forward of a k x k / stride k convolution, its input read in place (no gathered patch matrix).
    bn_in = (mean, rstd, gamma, beta) (C = 64): x is a PRE-BatchNorm tensor and the kernel convolves relu(bn(x)), applied while it stages x"""
    _req(x, BF16, "x"); _req(wmat, BF16, "wmat")
    B, H, W, C = x.shape
    pm = patch_map(H, W, C, k)
    M, N, K = B * (H // k) * (W // k), wmat.shape[0], k * k * C
    ld = round_up(N, 8)
    out = torch.empty((M, ld), dtype=BF16, device=x.device)
    if bn_in is not None:
        if C != 64:
            raise AutoProgHipError("gemm_nt_patch_fwd(bn_in=...): 64-channel feature maps")
        b = _bn_input(bn_in)
        check(lib.ap_gemm_nt_patch_bn(x.data_ptr(), ctypes.byref(b), wmat.data_ptr(), wmat.shape[1], out.data_ptr(), ld, M, N, K,
                                      bias.data_ptr() if bias is not None else None, ctypes.byref(pm), 1, _stream()), "ap_gemm_nt_patch_bn")
        return out
    check(lib.ap_gemm_nt_patch(x.data_ptr(), wmat.data_ptr(), wmat.shape[1], out.data_ptr(), ld, M, N, K,
                               bias.data_ptr() if bias is not None else None, ctypes.byref(pm), 1, _stream()), "ap_gemm_nt_patch")
    return out


def gemm_nt_patch_dgrad(dy, wmat_t, shape, k):
    """dy [M, ld >= N] bf16, wmat_t [k*k*C, >= N] bf16 (the transposed weight matrix) -> dx [B,H,W,C] bf16 written in feature-map layout"""
    _req(dy, BF16, "dy"); _req(wmat_t, BF16, "wmat_t")
    B, H, W, C = shape
    pm = patch_map(H, W, C, k)
    M, Kc = B * (H // k) * (W // k), k * k * C
    dx = torch.empty(shape, dtype=BF16, device=dy.device)
    check(lib.ap_gemm_nt_patch(dy.data_ptr(), wmat_t.data_ptr(), wmat_t.shape[1], dx.data_ptr(), dy.shape[1], M, Kc, wmat_t.shape[1],
                               None, ctypes.byref(pm), 2, _stream()), "ap_gemm_nt_patch")
    return dx


def gemm_tn_acc_grouped(problems, ln=None):
    """problems: list of (a, b, c, n1, n2, colsum[, colsum_weight, colsum_scale[, alpha[, b_patch[, b_bn]]]]) as for gemm_tn_acc (b_bn: the
    (mean, rstd, gamma, beta) of a BatchNorm whose relu(bn(.)) is applied to the patch-addressed b while it is staged); ONE launch for
    the whole list (chunks of 8).  colsum_weight: bf16 per-token weights of the column sum (DropPath keep mask), colsum_scale its factor;
    alpha: factor of the product (c += alpha * a^T b); b_patch: PatchMap -- b is then an NHWC feature map whose patches are the rows.
    ln: deferred LayerNorm reductions (the entries layernorm_bwd(..., defer=...) appended): LN_MAX_BATCH of them ride in the first launch."""
    from ._lib import TnProblem, TN_MAX_GROUP, TN_MAX_GROUP_DET, LnReduce, LN_MAX_BATCH
    ln = list(ln) if ln else []
    if len(ln) > LN_MAX_BATCH:
        layernorm_bwd_reduce_batched(ln[LN_MAX_BATCH:])
        ln = ln[:LN_MAX_BATCH]
    per = TN_MAX_GROUP_DET if deterministic else TN_MAX_GROUP
    for i0 in range(0, len(problems), per):
        chunk = problems[i0:i0 + per]
        arr = (TnProblem * len(chunk))()
        keep = []
        for q, prob in zip(arr, chunk):
            a, b, c, n1, n2, colsum = prob[:6]
            csw, css = (prob[6], prob[7]) if len(prob) > 6 else (None, 1.0)
            q.alpha = float(prob[8]) if len(prob) > 8 else 1.0
            bp = prob[9] if len(prob) > 9 else None
            _req(a, BF16, "a"); _req(b, BF16, "b"); _req(c, torch.float32, "c")
            if bp is not None:
                keep.append(bp)
                q.b_patch = ctypes.addressof(bp)
                if len(prob) > 10 and prob[10] is not None:
                    bb = _bn_input(prob[10])
                    keep.append(bb)
                    q.b_bn = ctypes.addressof(bb)
                q.A, q.lda, q.B, q.ldb, q.C, q.ldc = a.data_ptr(), a.shape[1], b.data_ptr(), 0, c.data_ptr(), c.shape[1]
                q.M, q.N1, q.N2 = a.shape[0], (c.shape[0] if n1 is None else n1), (c.shape[1] if n2 is None else n2)
                q.colsum_A = colsum.data_ptr() if colsum is not None else None
                q.colsum_weight, q.colsum_scale = None, 1.0
                continue
            if a.shape[0] != b.shape[0]:
                raise ValueError("gemm_tn_acc_grouped: token counts differ")
            q.A, q.lda, q.B, q.ldb, q.C, q.ldc = a.data_ptr(), a.shape[1], b.data_ptr(), b.shape[1], c.data_ptr(), c.shape[1]
            q.M, q.N1, q.N2 = a.shape[0], (c.shape[0] if n1 is None else n1), (c.shape[1] if n2 is None else n2)
            q.colsum_A = colsum.data_ptr() if colsum is not None else None
            if csw is not None:
                _req(csw, BF16, "colsum_weight")
                if csw.numel() < round_up(a.shape[0], 8) or csw.data_ptr() % 16:
                    raise AutoProgHipError("colsum_weight needs ceil(M/8)*8 elements and 16-byte alignment")
            q.colsum_weight = csw.data_ptr() if csw is not None else None
            q.colsum_scale = float(css)
        ptr = ctypes.cast(arr, ctypes.c_void_p)
        ws, ws_bytes = None, 0
        if deterministic:
            ws_bytes = lib.ap_gemm_tn_grouped_workspace(ptr, len(chunk))
            ws = torch.empty(max(ws_bytes // 4, 1), dtype=torch.float32, device=chunk[0][0].device)
        if ln:
            larr = (LnReduce * len(ln))()
            for q, (lws, n, C, dg, db) in zip(larr, ln):
                q.partial, q.n_partial, q.C, q.dgamma, q.dbeta = lws.data_ptr(), n, C, dg.data_ptr(), db.data_ptr()
            check(lib.ap_gemm_tn_acc_grouped_ln(ptr, len(chunk), ctypes.cast(larr, ctypes.c_void_p), len(ln), ws.data_ptr() if ws is not None else None,
                                                ws_bytes, _stream()), "ap_gemm_tn_acc_grouped_ln")
            ln = []
        else:
            check(lib.ap_gemm_tn_acc_grouped(ptr, len(chunk), ws.data_ptr() if ws is not None else None, ws_bytes, _stream()), "ap_gemm_tn_acc_grouped")


def colsum_acc(a, out, n=None):
    _req(a, BF16, "a"); _req(out, torch.float32, "out")
    n = out.numel() if n is None else n
    check(lib.ap_colsum_acc(a.data_ptr(), a.shape[1], out.data_ptr(), a.shape[0], n, _stream()), "ap_colsum_acc")
    return out


# ---------------------------------------------------------------------------------- outlook
def outlook_fwd(v, logits, heads, scale):
    _req(v, BF16, "v"); _req(logits, BF16, "logits")
    B, H, W, C = v.shape
    y = torch.empty_like(v)
    check(lib.ap_outlook_fwd(v.data_ptr(), logits.data_ptr(), logits.shape[-1], y.data_ptr(), B, H, W, heads, C // heads,
                             float(scale), _stream()), "ap_outlook_fwd")
    return y


def outlook_bwd(v, logits, dy, heads, scale):
    _req(v, BF16, "v"); _req(logits, BF16, "logits"); _req(dy, BF16, "dy")
    B, H, W, C = v.shape
    dv = torch.empty_like(v)
    dlogits = torch.empty_like(logits)
    check(lib.ap_outlook_bwd(v.data_ptr(), logits.data_ptr(), logits.shape[-1], dy.data_ptr(), dv.data_ptr(), dlogits.data_ptr(),
                             B, H, W, heads, C // heads, float(scale), _stream()), "ap_outlook_bwd")
    return dv, dlogits


def avgpool2_fwd(x):
    _req(x, BF16, "x")
    B, H, W, C = x.shape
    y = torch.empty((B, (H + 1) // 2, (W + 1) // 2, C), dtype=BF16, device=x.device)
    check(lib.ap_avgpool2_fwd(x.data_ptr(), y.data_ptr(), B, H, W, C, _stream()), "ap_avgpool2_fwd")
    return y


def avgpool2_bwd_acc(dpooled, dx):
    _req(dpooled, BF16, "dpooled"); _req(dx, BF16, "dx")
    B, H, W, C = dx.shape
    check(lib.ap_avgpool2_bwd_acc(dpooled.data_ptr(), dx.data_ptr(), B, H, W, C, _stream()), "ap_avgpool2_bwd_acc")
    return dx


# ------------------------------------------------------------------------------------- mhsa
def mhsa_emits_fp8(N, hd):
    """does mhsa_fwd(fp8=...) have the e4m3 side output for this shape?  (the key/query-blocked kernel: mhsa_use_flash in csrc/mhsa.hip)"""
    e = os.environ.get("AP_MHSA_FLASH")
    return (e == "1") if e is not None else (N > 256 or hd == 48)


def mhsa_fwd(qkv, B, N, heads, scale, out_row_scale=None, fp8=None):
    """out_row_scale: fp32 [B] factors (0/1 DropPath keep mask) folded into the stored output (see include/autoprog_hip.h).
    fp8 = (scale, amax) device scalars: -> (out, lse, out8), out8 the e4m3 bytes of out * scale[0] (mhsa_emits_fp8 shapes only)"""
    _req(qkv, BF16, "qkv")
    C = qkv.shape[-1] // 3
    out = torch.empty((B * N, C), dtype=BF16, device=qkv.device)
    lse = torch.empty((B, heads, N), dtype=torch.float32, device=qkv.device)
    if fp8 is not None:
        out8 = torch.empty((B * N, C), dtype=torch.uint8, device=qkv.device)
        check(lib.ap_mhsa_fwd_fp8(qkv.data_ptr(), out.data_ptr(), out8.data_ptr(), fp8[0].data_ptr(), fp8[1].data_ptr() if fp8[1] is not None else None,
                                  lse.data_ptr(), B, N, heads, C // heads, float(scale),
                                  out_row_scale.data_ptr() if out_row_scale is not None else None, _stream()), "ap_mhsa_fwd_fp8")
        return out, lse, out8
    check(lib.ap_mhsa_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, N, heads, C // heads, float(scale),
                          out_row_scale.data_ptr() if out_row_scale is not None else None, _stream()), "ap_mhsa_fwd")
    return out, lse


def mhsa_bwd(qkv, out, dout, lse, B, N, heads, scale):
    _req(qkv, BF16, "qkv"); _req(out, BF16, "out"); _req(dout, BF16, "dout")
    C = qkv.shape[-1] // 3
    dqkv = torch.empty_like(qkv)
    ws_bytes = lib.ap_mhsa_bwd_workspace(B, N, heads, C // heads)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=qkv.device) if ws_bytes else None
    check(lib.ap_mhsa_bwd(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), B, N, heads, C // heads,
                          float(scale), ws.data_ptr() if ws is not None else None, ws_bytes, _stream()), "ap_mhsa_bwd")
    return dqkv


def class_attn_fwd(q, kv, B, N, heads, scale, kv_cls=None):
    """kv_cls given: split layout -- key 0 is kv_cls[b], keys 1..N-1 the N-1 rows of kv (see include/autoprog_hip.h)"""
    _req(q, BF16, "q"); _req(kv, BF16, "kv")
    C = q.shape[-1]
    out = torch.empty((B, C), dtype=BF16, device=q.device)
    probs = torch.empty((B, heads, N), dtype=torch.float32, device=q.device)
    check(lib.ap_class_attn_fwd(q.data_ptr(), kv.data_ptr(), kv_cls.data_ptr() if kv_cls is not None else None, out.data_ptr(), probs.data_ptr(),
                                B, N, heads, C // heads, float(scale), _stream()), "ap_class_attn_fwd")
    return out, probs


def class_attn_bwd(q, kv, probs, dout, B, N, heads, scale, kv_cls=None):
    """-> (dq, dkv) or, in the split layout, (dq, dkv_tokens, dkv_cls)"""
    _req(dout, BF16, "dout")
    dq = torch.empty_like(q)
    dkv = torch.empty_like(kv)
    dkv_cls = torch.empty_like(kv_cls) if kv_cls is not None else None
    C = q.shape[-1]
    check(lib.ap_class_attn_bwd(q.data_ptr(), kv.data_ptr(), kv_cls.data_ptr() if kv_cls is not None else None, probs.data_ptr(), dout.data_ptr(),
                                dq.data_ptr(), dkv.data_ptr(), dkv_cls.data_ptr() if dkv_cls is not None else None,
                                B, N, heads, C // heads, float(scale), _stream()), "ap_class_attn_bwd")
    return (dq, dkv) if kv_cls is None else (dq, dkv, dkv_cls)


# ------------------------------------------------------------------------------ misc fused
def mix_token_swap(x, r0, r1, c0, c1):
    _req(x, BF16, "x")
    B, H, W, C = x.shape
    y = torch.empty_like(x)
    check(lib.ap_mix_token_swap(x.data_ptr(), y.data_ptr(), B, H, W, C, int(r0), int(r1), int(c0), int(c1), _stream()), "ap_mix_token_swap")
    return y


def mix_token_swap_dev(x, box_ptr, scale):
    """the same with the box {r0, r1, c0, c1} read from device memory (int32[4] at box_ptr, times `scale`): graph.StepScalars"""
    _req(x, BF16, "x")
    B, H, W, C = x.shape
    y = torch.empty_like(x)
    check(lib.ap_mix_token_swap_dev(x.data_ptr(), y.data_ptr(), B, H, W, C, int(box_ptr), int(scale), _stream()), "ap_mix_token_swap_dev")
    return y


def loss_combine(a, wa, b=None, wb=0.0):
    """fp32 scalar tensor wa * sum(a) + wb * sum(b) (one launch)"""
    _req(a, torch.float32, "a")
    out = torch.empty((), dtype=torch.float32, device=a.device)
    check(lib.ap_loss_combine(a.data_ptr(), a.numel(), float(wa), b.data_ptr() if b is not None else None, b.numel() if b is not None else 0,
                              float(wb), out.data_ptr(), _stream()), "ap_loss_combine")
    return out


def soft_ce_fwd_bwd(logits, C, target, t_sb, t_sc, t_sn, rows_per_batch, grad_scale, mix_lam=1.0, mix_batches=0, mix_lam_ptr=None):
    """returns (row_loss fp32 [M], dlogits bf16 like logits); mix_batches = B: target of batch b is lam*t[b] + (1-lam)*t[B-1-b].
    mix_lam_ptr: device address of lam (overrides mix_lam: graph.StepScalars)"""
    _req(logits, BF16, "logits")
    if not (target.is_cuda and target.dtype == torch.float32):
        raise AutoProgHipError("target must be a CUDA fp32 tensor (any strides; pass them explicitly)")
    M, ldx = logits.shape
    row_loss = torch.empty(M, dtype=torch.float32, device=logits.device)
    dlogits = torch.empty_like(logits)
    check(lib.ap_soft_ce_fwd_bwd_dev(logits.data_ptr(), ldx, target.data_ptr(), int(t_sb), int(t_sc), int(t_sn), int(rows_per_batch),
                                     row_loss.data_ptr(), dlogits.data_ptr(), float(grad_scale), M, C, float(mix_lam), int(mix_batches),
                                     int(mix_lam_ptr) if mix_lam_ptr else None, _stream()),
          "ap_soft_ce_fwd_bwd_dev")
    return row_loss, dlogits


def soft_ce_sparse_fwd_bwd(logits, C, idx, val, p_sb, p_sn, rows_per_batch, smoothing, grad_scale, mix_lam=1.0, mix_batches=0, mix_lam_ptr=None):
    """soft-target CE against top-K (class, score) pairs + label smoothing (the token-label target before it is densified);
    idx int32 / val fp32 with K = idx.shape[-1] pairs per row at b * p_sb + n * p_sn.  Returns (row_loss fp32 [M], dlogits bf16).
    mix_batches = B: the target of row (b, n) is mix_lam * t[b, n] + (1 - mix_lam) * t[B-1-b, n] (the mix-token class target)."""
    _req(logits, BF16, "logits")
    if not (idx.is_cuda and idx.dtype == torch.int32 and val.is_cuda and val.dtype == torch.float32):
        raise AutoProgHipError("sparse targets: idx must be CUDA int32 and val CUDA fp32")
    M, ldx = logits.shape
    row_loss = torch.empty(M, dtype=torch.float32, device=logits.device)
    dlogits = torch.empty_like(logits)
    check(lib.ap_soft_ce_sparse_fwd_bwd_dev(logits.data_ptr(), ldx, idx.data_ptr(), val.data_ptr(), int(idx.shape[-1]), int(p_sb), int(p_sn),
                                            int(rows_per_batch), float(smoothing), row_loss.data_ptr(), dlogits.data_ptr(), float(grad_scale), M, C,
                                            float(mix_lam), int(mix_batches), int(mix_lam_ptr) if mix_lam_ptr else None, _stream()),
          "ap_soft_ce_sparse_fwd_bwd_dev")
    return row_loss, dlogits


def row_scale(x, scale, rows_per_scale):
    _req(x, BF16, "x"); _req(scale, torch.float32, "scale")
    C = x.shape[-1]
    y = torch.empty_like(x)
    check(lib.ap_row_scale(x.data_ptr(), scale.data_ptr(), y.data_ptr(), x.numel() // C, C, int(rows_per_scale), _stream()), "ap_row_scale")
    return y


def add_bcast(a, b):
    _req(a, BF16, "a"); _req(b, BF16, "b")
    y = torch.empty_like(a)
    check(lib.ap_add_bcast(a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), b.numel(), _stream()), "ap_add_bcast")
    return y


def resample_grid(x, wy, wx, out=None, accumulate=False):
    """x fp32 [hi, wi, C], wy fp32 [ho, hi], wx fp32 [wo, wi] -> out fp32 [ho, wo, C] (+)= the separable resampling (ap_resample_grid)"""
    _req(x, torch.float32, "x"); _req(wy, torch.float32, "wy"); _req(wx, torch.float32, "wx")
    hi, wi, C = x.shape
    ho, wo = wy.shape[0], wx.shape[0]
    if wy.shape[1] != hi or wx.shape[1] != wi:
        raise AutoProgHipError("resample_grid: tap matrices do not match the grid")
    if out is None:
        out = torch.empty((ho, wo, C), dtype=torch.float32, device=x.device)
    else:
        _req(out, torch.float32, "out")
        if out.numel() != ho * wo * C:
            raise AutoProgHipError("resample_grid: output shape")
    check(lib.ap_resample_grid(x.data_ptr(), hi, wi, wy.data_ptr(), wx.data_ptr(), out.data_ptr(), ho, wo, C, 1 if accumulate else 0, _stream()),
          "ap_resample_grid")
    return out


def sum_reps_acc(x, out, reps):
    _req(x, BF16, "x"); _req(out, torch.float32, "out")
    check(lib.ap_sum_reps_acc(x.data_ptr(), out.data_ptr(), out.numel(), int(reps), _stream()), "ap_sum_reps_acc")
    return out


# ---------------------------------------------------------------------------- stem BN + ReLU
def bn_relu_fwd(x, gamma, beta, running_mean, running_var, training, momentum, eps, partials=None, apply=True):
    """x: bf16 [..., C] NHWC rows.  returns (y, mean, rstd).  partials (training only): fp32 [rows,2,C] partial sums / sums of squares
    of x from the kernel that produced it (conv3x3_c64 want_stats) -- the statistics pass over x is skipped.  apply=False (with
    partials, or in eval mode): statistics only, y is None -- the consumer normalises while it loads (conv3x3_c64(bn_in=...))"""
    _req(x, BF16, "x")
    C = x.shape[-1]
    T = x.numel() // C
    if not apply and not training:
        return None, running_mean.float().contiguous(), torch.rsqrt(running_var.float() + eps).contiguous()
    if not apply and partials is None:
        raise AutoProgHipError("bn_relu_fwd(apply=False) needs the producer's partial statistics")
    y = torch.empty_like(x) if apply else None
    if training and partials is not None:
        _req(partials, torch.float32, "partials")
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        rstd = torch.empty(C, dtype=torch.float32, device=x.device)
        check(lib.ap_bn_relu_fwd_partials(x.data_ptr(), partials.data_ptr(), partials.shape[0], gamma.data_ptr(), beta.data_ptr(),
                                          running_mean.data_ptr() if running_mean is not None else None,
                                          running_var.data_ptr() if running_var is not None else None,
                                          float(momentum), float(eps), y.data_ptr() if y is not None else None, mean.data_ptr(), rstd.data_ptr(), T, C, _stream()),
              "ap_bn_relu_fwd_partials")
        return y, mean, rstd
    if training:
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        rstd = torch.empty(C, dtype=torch.float32, device=x.device)
    else:
        mean = running_mean.float().contiguous()
        rstd = torch.rsqrt(running_var.float() + eps).contiguous()
    ws_bytes = lib.ap_bn_relu_workspace(T, C)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
    check(lib.ap_bn_relu_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                             running_mean.data_ptr() if (training and running_mean is not None) else None,
                             running_var.data_ptr() if (training and running_var is not None) else None,
                             1 if training else 0, float(momentum), float(eps), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                             T, C, ws.data_ptr(), ws_bytes, _stream()), "ap_bn_relu_fwd")
    return y, mean, rstd


def bn_relu_bwd(dy, x, gamma, beta, mean, rstd, dgamma, dbeta, act_out=False):
    """act_out: -> (dx, relu(bn(x))) -- the activation a fused forward never stored, written by the pass that holds x anyway"""
    _req(dy, BF16, "dy"); _req(x, BF16, "x")
    C = x.shape[-1]
    T = x.numel() // C
    dx = torch.empty_like(x)
    ws_bytes = lib.ap_bn_relu_workspace(T, C)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
    if act_out:
        act = torch.empty_like(x)
        check(lib.ap_bn_relu_bwd_act(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                     dx.data_ptr(), act.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), T, C, ws.data_ptr(), ws_bytes, _stream()),
              "ap_bn_relu_bwd_act")
        return dx, act
    check(lib.ap_bn_relu_bwd(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                             dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), T, C, ws.data_ptr(), ws_bytes, _stream()), "ap_bn_relu_bwd")
    return dx


def bn_relu_bwd_partials(dy, x, gamma, beta, mean, rstd, partial, dgamma, dbeta):
    """bn_relu_bwd whose first pass ran inside the convolution that produced dy (conv3x3_c64_bwd_stats): finalize + dx only"""
    _req(dy, BF16, "dy"); _req(x, BF16, "x"); _req(partial, torch.float32, "partial")
    C = x.shape[-1]
    T = x.numel() // C
    dx = torch.empty_like(x)
    ws_bytes = lib.ap_bn_relu_workspace(T, C)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
    check(lib.ap_bn_relu_bwd_partials(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                      partial.data_ptr(), partial.shape[0], dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), T, C,
                                      ws.data_ptr(), ws_bytes, _stream()), "ap_bn_relu_bwd_partials")
    return dx


# ------------------------------------------------------------------------------------- stem 3x3 convolution (C = 64)
def conv3x3_pack(weight):
    """fp32 [64,64,3,3] (or [128,128,3,3]: the VOLO-D4 / D5 stem) -> (w_fwd, w_bwd) bf16 operand layouts of conv3x3_c64 / conv3x3_c128"""
    _req(weight, torch.float32, "weight")
    if tuple(weight.shape) == (128, 128, 3, 3):
        wf = torch.empty(9 * 128 * 128, dtype=BF16, device=weight.device)
        wb = torch.empty_like(wf)
        check(lib.ap_conv3x3_c128_pack(weight.contiguous().data_ptr(), wf.data_ptr(), wb.data_ptr(), _stream()), "ap_conv3x3_c128_pack")
        return wf, wb
    if tuple(weight.shape) != (64, 64, 3, 3):
        raise AutoProgHipError("the HIP 3x3 convolutions handle 64 -> 64 and 128 -> 128 channels (got %s)" % (tuple(weight.shape),))
    wf = torch.empty(9 * 64 * 64, dtype=BF16, device=weight.device)
    wb = torch.empty_like(wf)
    check(lib.ap_conv3x3_c64_pack(weight.contiguous().data_ptr(), wf.data_ptr(), wb.data_ptr(), _stream()), "ap_conv3x3_c64_pack")
    return wf, wb


def _bn_input(bn_in):
    from ._lib import BnInput
    mean, rstd, gamma, beta = bn_in
    for t, n in ((mean, "mean"), (rstd, "rstd"), (gamma, "gamma"), (beta, "beta")):
        _req(t, torch.float32, n)
    b = BnInput()
    b.mean, b.rstd, b.gamma, b.beta = mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr()
    return b


def conv3x3_c64(x, w_packed, want_stats=False, bn_in=None):
    """x [B,H,W,64] bf16 (NHWC, contiguous) -> conv3x3 / stride 1 / pad 1.  want_stats: also return the partial BatchNorm
    statistics of the output (fp32 [rows,2,64], for bn_relu_fwd(..., partials=)).  bn_in = (mean, rstd, gamma, beta): x is the
    PRE-BatchNorm output of the previous convolution and the kernel applies relu(bn(x)) while it stages its input"""
    _req(x, BF16, "x"); _req(w_packed, BF16, "w_packed")
    B, H, W, C = x.shape
    if C == 128:                       # csrc/conv128.hip: patch in LDS, weights streamed per (tap, 64-channel) slab
        if bn_in is not None:
            raise AutoProgHipError("conv3x3 at 128 channels: the BatchNorm input transform exists at 64 channels only")
        if w_packed.numel() != 9 * 128 * 128:
            raise AutoProgHipError("conv3x3 at 128 channels needs the operand of conv3x3_pack on a [128,128,3,3] weight")
        y = torch.empty_like(x)
        stats = torch.empty((lib.ap_conv3x3_c128_stat_rows(B, H, W), 2, 128), dtype=torch.float32, device=x.device) if want_stats else None
        check(lib.ap_conv3x3_c128(x.data_ptr(), w_packed.data_ptr(), y.data_ptr(), B, H, W, stats.data_ptr() if want_stats else None, _stream()),
              "ap_conv3x3_c128")
        return (y, stats) if want_stats else y
    if C != 64:
        raise AutoProgHipError("conv3x3_c64: 64 (or 128) channels (got %d)" % C)
    y = torch.empty_like(x)
    stats = torch.empty((lib.ap_conv3x3_c64_stat_rows(B, H, W), 2, 64), dtype=torch.float32, device=x.device) if want_stats else None
    if bn_in is not None:
        b = _bn_input(bn_in)
        check(lib.ap_conv3x3_c64_bn(x.data_ptr(), ctypes.byref(b), w_packed.data_ptr(), y.data_ptr(), B, H, W,
                                    stats.data_ptr() if want_stats else None, _stream()), "ap_conv3x3_c64_bn")
    else:
        check(lib.ap_conv3x3_c64(x.data_ptr(), w_packed.data_ptr(), y.data_ptr(), B, H, W,
                                 stats.data_ptr() if want_stats else None, _stream()), "ap_conv3x3_c64")
    return (y, stats) if want_stats else y


def conv3x3_c64_bwd_stats(dz, w_packed_bwd, z_below, bn_below):
    """the input-gradient convolution of a 64-wide stem layer, with the first pass of the BatchNorm + ReLU backward of the layer BELOW in
    its epilogue: -> (da, partial [rows,2,64]) for bn_relu_bwd_partials(da, z_below, ..., partial, ...).  bn_below = (mean, rstd, gamma, beta)"""
    _req(dz, BF16, "dz"); _req(w_packed_bwd, BF16, "w_packed_bwd"); _req(z_below, BF16, "z_below")
    B, H, W, C = dz.shape
    if C != 64 or tuple(z_below.shape) != tuple(dz.shape):
        raise AutoProgHipError("conv3x3_c64_bwd_stats: [B,H,W,64] maps of one shape (got %s, %s)" % (tuple(dz.shape), tuple(z_below.shape)))
    da = torch.empty_like(dz)
    stats = torch.empty((lib.ap_conv3x3_c64_stat_rows(B, H, W), 2, 64), dtype=torch.float32, device=dz.device)
    b = _bn_input(bn_below)
    check(lib.ap_conv3x3_c64_bwd_stats(dz.data_ptr(), w_packed_bwd.data_ptr(), da.data_ptr(), B, H, W, z_below.data_ptr(), ctypes.byref(b),
                                       stats.data_ptr(), _stream()), "ap_conv3x3_c64_bwd_stats")
    return da, stats


def conv3x3_c64_wgrad(x, dy, dw, bn_in=None):
    """dw (fp32 [64,64,3,3], contiguous) += weight gradient of conv3x3_c64 for input x and output gradient dy; bn_in as for conv3x3_c64
    (the layer's input was relu(bn(x)))"""
    _req(x, BF16, "x"); _req(dy, BF16, "dy"); _req(dw, torch.float32, "dw")
    B, H, W, C = x.shape
    if C == 128 and tuple(dy.shape) == tuple(x.shape) and tuple(dw.shape) == (128, 128, 3, 3) and bn_in is None:
        ws_bytes = lib.ap_conv3x3_c128_wgrad_workspace(B, H, W)
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
        check(lib.ap_conv3x3_c128_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, H, W, ws.data_ptr(), ws_bytes, _stream()), "ap_conv3x3_c128_wgrad")
        return dw
    if C != 64 or tuple(dy.shape) != tuple(x.shape) or tuple(dw.shape) != (64, 64, 3, 3):
        raise AutoProgHipError("conv3x3_c64_wgrad: shapes %s %s %s" % (tuple(x.shape), tuple(dy.shape), tuple(dw.shape)))
    ws_bytes = lib.ap_conv3x3_c64_wgrad_workspace(B, H, W)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
    if bn_in is not None:
        b = _bn_input(bn_in)
        check(lib.ap_conv3x3_c64_wgrad_bn(x.data_ptr(), ctypes.byref(b), dy.data_ptr(), dw.data_ptr(), B, H, W, ws.data_ptr(), ws_bytes, _stream()),
              "ap_conv3x3_c64_wgrad_bn")
        return dw
    check(lib.ap_conv3x3_c64_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, H, W, ws.data_ptr(), ws_bytes, _stream()), "ap_conv3x3_c64_wgrad")
    return dw


# ------------------------------------------------------------------------------------- stem 7x7 / stride 2 convolution (3 -> 64)
def resize_bilinear_s2d16(x, size):
    """fp32 [B,3,Hi,Wi] -> bf16 [B,size/2,size/2,16]: bilinear resize (F.interpolate align_corners=False) in the space-to-depth
    layout conv7_s2d reads (size even)"""
    _req(x, torch.float32, "x")
    B, C, Hi, Wi = x.shape
    if C != 3 or size % 2:
        raise AutoProgHipError("resize_bilinear_s2d16: 3 channels and an even output size (got C=%d, size=%d)" % (C, size))
    y = torch.empty((B, size // 2, size // 2, 16), dtype=BF16, device=x.device)
    check(lib.ap_resize_bilinear_s2d16(x.data_ptr(), y.data_ptr(), B, Hi, Wi, size, size, _stream()), "ap_resize_bilinear_s2d16")
    return y


def conv7_pack(weight):
    """fp32 [64,3,7,7] -> packed operand; [128,3,7,7] (VOLO-D4 / D5 stem): the two 64-channel halves' operands, one behind the other"""
    _req(weight, torch.float32, "weight")
    if tuple(weight.shape) not in ((64, 3, 7, 7), (128, 3, 7, 7)):
        raise AutoProgHipError("conv7_s2d handles 3 -> 64 / 128 channels, 7x7 (got %s)" % (tuple(weight.shape),))
    halves = weight.shape[0] // 64
    w = weight.contiguous()
    wp = torch.empty(halves * 16 * 64 * 16, dtype=BF16, device=weight.device)
    for h in range(halves):
        check(lib.ap_conv7_pack(w.data_ptr() + h * 64 * 147 * 4, wp.data_ptr() + h * 16 * 64 * 16 * 2, _stream()), "ap_conv7_pack")
    return wp


def conv7_s2d(xs, w_packed, want_stats=False):
    """xs [B,H,W,16] bf16 (space-to-depth image) -> [B,H,W,64] bf16 = conv7x7 / stride 2 / pad 3 of the image"""
    _req(xs, BF16, "xs"); _req(w_packed, BF16, "w_packed")
    B, H, W, C = xs.shape
    if C != 16:
        raise AutoProgHipError("conv7_s2d: 16 space-to-depth channels (got %d)" % C)
    if w_packed.numel() == 2 * 16 * 64 * 16:          # 128 output channels: the kernel once per 64-channel half, pixel stride 128
        y = torch.empty((B, H, W, 128), dtype=BF16, device=xs.device)
        rows = lib.ap_conv7_s2d_stat_rows(B, H, W)
        st = [torch.empty((rows, 2, 64), dtype=torch.float32, device=xs.device) if want_stats else None for _ in range(2)]
        for h in range(2):
            check(lib.ap_conv7_s2d_ld(xs.data_ptr(), w_packed.data_ptr() + h * 16 * 64 * 16 * 2, y.data_ptr() + h * 64 * 2, 128, B, H, W,
                                      st[h].data_ptr() if want_stats else None, _stream()), "ap_conv7_s2d_ld")
        return (y, torch.cat(st, dim=2)) if want_stats else y
    y = torch.empty((B, H, W, 64), dtype=BF16, device=xs.device)
    stats = torch.empty((lib.ap_conv7_s2d_stat_rows(B, H, W), 2, 64), dtype=torch.float32, device=xs.device) if want_stats else None
    check(lib.ap_conv7_s2d(xs.data_ptr(), w_packed.data_ptr(), y.data_ptr(), B, H, W, stats.data_ptr() if want_stats else None, _stream()),
          "ap_conv7_s2d")
    return (y, stats) if want_stats else y


def conv7_s2d_wgrad(xs, dz, dw):
    """dw (fp32 [64,3,7,7]) += weight gradient of conv7_s2d"""
    _req(xs, BF16, "xs"); _req(dz, BF16, "dz"); _req(dw, torch.float32, "dw")
    B, H, W, _ = xs.shape
    if tuple(dz.shape) == (B, H, W, 128) and tuple(dw.shape) == (128, 3, 7, 7):
        ws_bytes = lib.ap_conv7_s2d_wgrad_workspace(B, H, W)
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=xs.device)
        for h in range(2):
            check(lib.ap_conv7_s2d_wgrad_ld(xs.data_ptr(), dz.data_ptr() + h * 64 * 2, 128, dw.data_ptr() + h * 64 * 147 * 4, B, H, W, ws.data_ptr(), ws_bytes,
                                            _stream()), "ap_conv7_s2d_wgrad_ld")
        return dw
    if tuple(dz.shape) != (B, H, W, 64) or tuple(dw.shape) != (64, 3, 7, 7):
        raise AutoProgHipError("conv7_s2d_wgrad: shapes %s %s %s" % (tuple(xs.shape), tuple(dz.shape), tuple(dw.shape)))
    ws_bytes = lib.ap_conv7_s2d_wgrad_workspace(B, H, W)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=xs.device)
    check(lib.ap_conv7_s2d_wgrad(xs.data_ptr(), dz.data_ptr(), dw.data_ptr(), B, H, W, ws.data_ptr(), ws_bytes, _stream()), "ap_conv7_s2d_wgrad")
    return dw
