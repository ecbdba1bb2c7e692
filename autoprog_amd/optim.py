"""Fused AdamW + EMA update on flat slabs.

`FlatAdamWEma` re-points every parameter at a view of ONE fp32 slab (same order as the gradient slab of
`dist.GradientBucketReducer`), keeps Adam moments and up to four EMA copies as flat slabs, and performs
the whole update in one kernel (`ap_adamw_ema_step`).  Semantics = `torch.optim.AdamW` with timm's
no-weight-decay rule (1-D params, biases and `model.no_weight_decay()` names get wd 0)
followed by `ModelEmaV2.update` for each decay.  Floating-point BUFFERS (BatchNorm running stats) are
EMA-averaged with torch foreach ops, as ModelEmaV2 averages the whole state dict."""
import ctypes

import torch

from ._lib import check, lib


class FlatAdamWEma:
    def __init__(self, model, reducer, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05, ema_decays=()):
        assert len(ema_decays) <= 4
        self.model, self.reducer = model, reducer
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.ema_decays = list(ema_decays)
        self.step_count = 0
        flat_g = reducer.flat
        n = flat_g.numel()
        self.n_pad = (n + 3) // 4 * 4
        dev = flat_g.device
        self.p = torch.zeros(self.n_pad, dtype=torch.float32, device=dev)
        self.g = flat_g if self.n_pad == n else None
        self.wd_mask = torch.zeros(self.n_pad, dtype=torch.uint8, device=dev)
        skip = set(model.no_weight_decay()) if hasattr(model, "no_weight_decay") else set()
        names = {id(p): n_ for n_, p in model.named_parameters()}
        base = flat_g.data_ptr()
        self._views = []
        for p in reducer.params:
            off = (p.grad.data_ptr() - base) // 4
            view = self.p[off:off + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view                                   # parameters now live in the slab
            name = names.get(id(p), "")
            decay = not (p.dim() == 1 or name.endswith(".bias") or name in skip)
            if decay and weight_decay > 0:
                self.wd_mask[off:off + p.numel()] = 1
            self._views.append((name, off, p.shape))
        self.m = torch.zeros_like(self.p)
        self.v = torch.zeros_like(self.p)
        self.ema = [self.p.clone() for _ in self.ema_decays]
        self._ema_ptrs = (ctypes.c_void_p * max(1, len(self.ema)))(*[e.data_ptr() for e in self.ema])
        self._ema_decay = (ctypes.c_float * max(1, len(self.ema)))(*self.ema_decays)
        # bf16 copies for the GEMMs: a bf16 slab with the parameters' offsets (written by the update kernel)
        # and a slab of transposed [K, ld(N)] copies of every >=2-D parameter (one batched launch per step)
        import struct
        self.p16 = torch.empty(self.n_pad, dtype=torch.bfloat16, device=dev)
        descs, t_off, tiles = [], 0, 0
        mats = []
        for p, (name, off, shape) in zip(reducer.params, self._views):
            if p.dim() >= 2:
                rows, cols = shape[0], p.numel() // shape[0]
                ld = (rows + 7) // 8 * 8
                mats.append((p, off, rows, cols, ld, t_off))
                descs.append(struct.pack("<qqiiii", off, t_off, rows, cols, ld, tiles))
                tiles += ((rows + 31) // 32) * ((cols + 31) // 32)
                t_off += cols * ld
        self.p16_t = torch.zeros(max(t_off, 8), dtype=torch.bfloat16, device=dev)
        self._tr_count, self._tr_tiles = len(descs), tiles
        self._tr_desc = torch.frombuffer(bytearray(b"".join(descs)), dtype=torch.uint8).to(dev) if descs else None
        self.p16.copy_(self.p)                                   # initial fill (torch cast, once)
        for p, (name, off, shape) in zip(reducer.params, self._views):
            rows = shape[0] if len(shape) else 1
            p._ap_flat16 = (p.data_ptr(), self.p16[off:off + p.numel()].view(rows, -1) if p.dim() >= 2 else None, None)
        for p, off, rows, cols, ld, toff in mats:
            p._ap_flat16 = (p.data_ptr(), self.p16[off:off + rows * cols].view(rows, cols), self.p16_t[toff:toff + cols * ld].view(cols, ld))
        self._refresh_transposes()
        self._buffers = [b for b in model.buffers() if b.dtype.is_floating_point]
        self.ema_buffers = [[b.detach().clone() for b in self._buffers] for _ in self.ema_decays]

    def _refresh_transposes(self):
        if self._tr_desc is not None:
            check(lib.ap_batched_transpose_bf16(self.p16.data_ptr(), self.p16_t.data_ptr(), self._tr_desc.data_ptr(), self._tr_count,
                                                self._tr_tiles, torch.cuda.current_stream().cuda_stream), "ap_batched_transpose_bf16")

    def step(self):
        self.step_count += 1
        g = self.reducer.flat
        if self.g is None:                                   # slab length not a multiple of 4: padded copy
            gp = torch.zeros(self.n_pad, dtype=torch.float32, device=g.device)
            gp[:g.numel()] = g
            g = gp
        check(lib.ap_adamw_ema_step(self.p.data_ptr(), g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.wd_mask.data_ptr(),
                                    self.n_pad, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.step_count,
                                    self._ema_ptrs, self._ema_decay, len(self.ema), self.p16.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream), "ap_adamw_ema_step")
        self._refresh_transposes()
        from . import functional
        functional._WeightBank.generation += 1             # the kernel wrote the parameters behind autograd's back
        if self._buffers:
            with torch.no_grad():
                for d, bufs in zip(self.ema_decays, self.ema_buffers):
                    torch._foreach_lerp_(bufs, self._buffers, 1.0 - d)

    def ema_state_dict(self, i):
        """name -> tensor views of EMA copy i (checkpoint format `state_dict_ema_{i}`, prog/checkpoint_saver.py:110-130)"""
        out = {name: self.ema[i][off:off + int(torch.tensor(shape).prod())].view(shape) for name, off, shape in self._views}
        for (bname, _), t in zip(self.model.named_buffers(), self.ema_buffers[i] if self.ema_buffers else []):
            out[bname] = t
        return out
