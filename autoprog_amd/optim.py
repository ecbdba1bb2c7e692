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
from . import ops


class FlatAdamWEma:
    def __init__(self, model, reducer, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05, ema_decays=()):
        assert len(ema_decays) <= 4
        self.model, self.reducer = model, reducer
        self.betas, self.eps = betas, eps
        # torch.optim-style groups (timm's create_optimizer layout: decayed, then un-decayed parameters).  step() reads
        # `lr` / `weight_decay` from here every call, so timm LR schedulers that write param_groups[i]["lr"] work unchanged.
        self.param_groups = [{"lr": lr, "weight_decay": weight_decay, "betas": betas, "eps": eps, "params": []},
                             {"lr": lr, "weight_decay": 0.0, "betas": betas, "eps": eps, "params": []}]
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
            if decay:
                self.wd_mask[off:off + p.numel()] = 1
            self.param_groups[0 if decay else 1]["params"].append(p)
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
            p._ap_flat16 = [p.data_ptr(), self.p16[off:off + p.numel()].view(rows, -1) if p.dim() >= 2 else None, None, p._version]
        for p, off, rows, cols, ld, toff in mats:
            p._ap_flat16 = [p.data_ptr(), self.p16[off:off + rows * cols].view(rows, cols), self.p16_t[toff:toff + cols * ld].view(cols, ld), p._version]
        self._refresh_transposes()
        self._float_buffers = [(n_, b) for n_, b in model.named_buffers() if b.dtype.is_floating_point]
        self._buffers = [b for _, b in self._float_buffers]
        self.ema_buffers = [[b.detach().clone() for b in self._buffers] for _ in self.ema_decays]
        # model.load_state_dict() copies INTO the slab views (pointers unchanged): re-derive the bf16 copies afterwards
        self._load_hook = model.register_load_state_dict_post_hook(lambda module, incompatible: self.resync())

    # ------------------------------------------------------------------ properties kept for callers of the round-1 API
    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    @lr.setter
    def lr(self, v):
        for g in self.param_groups:
            g["lr"] = v

    @property
    def weight_decay(self):
        return self.param_groups[0]["weight_decay"]

    def _stamp_versions(self):
        for p in self.reducer.params:
            p._ap_flat16[3] = p._version

    def resync(self, reset_ema=False, reset_moments=False):
        """re-derive the bf16 / transposed weight copies from the fp32 slab after the parameters were written from outside
        (load_state_dict, grow_clone_ema / extract_subnet, manual edits); optionally restart the EMA copies at the new weights
        (the reference rebuilds its ModelEma list after a stage transition, main_prog.py:1406) and zero the Adam moments."""
        with torch.no_grad():
            self.p16.copy_(self.p)
            self._refresh_transposes()
            if reset_ema:
                for e in self.ema:
                    e.copy_(self.p)
                for bufs in self.ema_buffers:
                    for dst, src in zip(bufs, self._buffers):
                        dst.copy_(src)
            if reset_moments:
                self.m.zero_()
                self.v.zero_()
                self.step_count = 0
        self._stamp_versions()
        from . import functional
        functional._WeightBank.generation += 1

    def _refresh_transposes(self):
        if self._tr_desc is not None:
            check(lib.ap_batched_transpose_bf16(self.p16.data_ptr(), self.p16_t.data_ptr(), self._tr_desc.data_ptr(), self._tr_count,
                                                self._tr_tiles, ops._stream()), "ap_batched_transpose_bf16")

    def _clip_workspace(self, device):
        """the norm's workspace, allocated OUTSIDE any stream capture: a tensor first allocated while a step is being captured
        (GraphedStep(clip_grad=...).capture(warmup=0)) would come from that graph's private pool and then be shared with eager steps
        and other graphs through this cache -- GraphedStep.capture() calls this before it starts capturing"""
        if getattr(self, "_sumsq_ws", None) is None:
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("FlatAdamWEma: the clip-norm workspace must exist before a step is captured (call _clip_workspace first)")
            self._sumsq_ws = torch.empty(lib.ap_sumsq_workspace() // 8, dtype=torch.float64, device=device)
            self._gnorm_sq = torch.zeros(1, dtype=torch.float32, device=device)

    def step(self, clip_grad=None, clip_mode="norm", scalars=None):
        """one AdamW + EMA update from the gradient slab.  clip_grad / clip_mode: the reference's `--clip-grad` / `--clip-mode`
        (main_prog.py:129-132; prog/scaler.py:60-68 calls timm's dispatch_clip_grad between backward and optimizer.step()):
          'norm'  -- torch.nn.utils.clip_grad_norm_(parameters, clip_grad): ONE pass over the flat slab for the global norm
                     (ap_sumsq_f32, a device scalar), the factor min(1, clip / (norm + 1e-6)) is applied inside the update kernel;
          'value' -- clip_grad_value_: every gradient element clamped to [-clip, clip] inside the update kernel.
        Both act on the MEAN gradient: with GradientBucketReducer(defer_mean=True) the slab still holds the all-reduced SUM and the
        1/world factor is applied first -- clipping the slab from outside before step() would be off by `world`.
        `last_grad_norm` (device scalar, 'norm' mode) is the norm of the mean gradient before clipping.
        scalars: graph.StepScalars -- the learning rate and Adam's bias corrections are read from device memory (the caller refreshed
        them for this step): nothing in the launch changes from step to step, the step can be replayed from a HIP graph."""
        self.step_count += 1
        g = self.reducer.flat
        if self.g is None:                                   # slab length not a multiple of 4: padded copy
            gp = torch.zeros(self.n_pad, dtype=torch.float32, device=g.device)
            gp[:g.numel()] = g
            g = gp
        gscale = float(self.reducer.take_pending_scale())
        gnorm_ptr, max_norm, clip_value = None, 0.0, 0.0
        if clip_grad is not None and float(clip_grad) > 0:
            if clip_mode == "norm":
                self._clip_workspace(g.device)
                check(lib.ap_sumsq_f32(g.data_ptr(), g.numel(), self._gnorm_sq.data_ptr(), self._sumsq_ws.data_ptr(), self._sumsq_ws.numel() * 8,
                                       ops._stream()), "ap_sumsq_f32")
                gnorm_ptr, max_norm = self._gnorm_sq.data_ptr(), float(clip_grad)
                self.last_grad_norm = self._gnorm_sq.sqrt() * gscale
            elif clip_mode == "value":
                clip_value = float(clip_grad)
            else:
                raise NotImplementedError("FlatAdamWEma.step: clip_mode %r (the reference's default is 'norm'; 'agc' is not on the flat-slab path)" % (clip_mode,))
        lr, wd = float(self.param_groups[0]["lr"]), float(self.param_groups[0]["weight_decay"])
        if float(self.param_groups[1]["lr"]) != lr:
            raise ValueError("FlatAdamWEma: both parameter groups must share one learning rate (timm schedulers do)")
        check(lib.ap_adamw_ema_step(self.p.data_ptr(), g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.wd_mask.data_ptr(),
                                    self.n_pad, lr, self.betas[0], self.betas[1], self.eps, wd, self.step_count, gscale,
                                    gnorm_ptr, max_norm, clip_value, scalars.adam_ptr if scalars is not None else None, self._ema_ptrs, self._ema_decay, len(self.ema), self.p16.data_ptr(),
                                    ops._stream()), "ap_adamw_ema_step")
        self._refresh_transposes()
        from . import functional
        functional._WeightBank.generation += 1             # the kernel wrote the parameters behind autograd's back
        if self._buffers:
            with torch.no_grad():
                for d, bufs in zip(self.ema_decays, self.ema_buffers):
                    torch._foreach_lerp_(bufs, self._buffers, 1.0 - d)

    def grow(self, old_mask, new_mask, model_source="ema_last"):
        """stage transition of a progressive run ON the live slabs (prog/elastic.py; reference: create_stage_model_and_optimizer,
        main_prog.py:1300-1430).  EMA copy i always comes from EMA copy i, the optimizer restarts, BatchNorm statistics go back to
        their defaults.  The MODEL weights come from
          model_source = "ema_last": the LAST EMA copy (load='slice' with load_with_clone_ema, main_prog.py:1378-1382: growth, and
                                     the same depth under a new DropPath strength, main_prog.py:828-831), or
          model_source = "model":    the trained model itself (load='super', main_prog.py:1389: the sub-network a search picked
                                     out of its supernet keeps the weights the search epochs trained).
        The caller switches the model to the new stage with set_sample_config afterwards."""
        from .prog.elastic import growth_sources
        if model_source not in ("ema_last", "model"):
            raise ValueError("grow: model_source must be 'ema_last' or 'model'")
        where = {name: (off, shape) for name, off, shape in self._views}
        src_of = growth_sources(list(where), old_mask, new_mask)
        with torch.no_grad():
            old = [e.clone() for e in self.ema] if self.ema else [self.p.clone()]
            old_model = self.p.clone() if (model_source == "model" or not self.ema) else old[-1]
            for name, (off, shape) in where.items():
                sname = src_of.get(name)
                if sname is None:
                    continue                           # slot inactive in the new stage: untouched
                soff, sshape = where[sname]
                n = 1
                for d in shape:
                    n *= d
                if tuple(sshape) != tuple(shape):
                    raise ValueError("%s <- %s: shapes differ (width growth is a state-dict operation, prog/growth.py)" % (name, sname))
                self.p[off:off + n].copy_(old_model[soff:soff + n])
                for e, eo in zip(self.ema, old):
                    e[off:off + n].copy_(eo[soff:soff + n])
            for m in self.model.modules():            # the reference's new network starts with default BatchNorm statistics
                if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                    m.reset_running_stats()
            for bufs in self.ema_buffers:
                for dst, src in zip(bufs, self._buffers):
                    dst.copy_(src)
        self.resync(reset_ema=False, reset_moments=True)

    def ema_weights(self, i):
        """context manager: the model computes with EMA copy i (parameters and BatchNorm statistics) -- the reference evaluates /
        probes its ModelEma modules (main_prog.py:1698-1760); here the slabs trade places for the duration of the block"""
        opt = self

        class _Swap:
            def _swap(self_inner):
                with torch.no_grad():
                    tmp = opt.p.clone()
                    opt.p.copy_(opt.ema[i])
                    opt.ema[i].copy_(tmp)
                    for b, e in zip(opt._buffers, opt.ema_buffers[i]):
                        t = b.detach().clone()
                        b.copy_(e)
                        e.copy_(t)
                opt.resync()

            def __enter__(self_inner):
                self_inner._swap()
                return opt

            def __exit__(self_inner, *exc):
                self_inner._swap()
                return False
        return _Swap()

    def zero_grad(self, set_to_none=False):
        """gradients are views of the reducer's slab and stay attached (prog/scaler.py:60-68 step contract)"""
        self.reducer.zero_grad()

    def ema_state_dict(self, i):
        """name -> tensor of EMA copy i in the checkpoint format `state_dict_ema_{i}` (prog/checkpoint_saver.py:110-130): parameters
        are views of the EMA slab, floating-point buffers (BatchNorm running stats) the lerp-averaged copies, integer buffers
        (num_batches_tracked) the model's current value -- as ModelEmaV2 keeps them."""
        out = {}
        for name, off, shape in self._views:
            n = 1
            for d in shape:
                n *= d
            out[name] = self.ema[i][off:off + n].view(shape)
        ema_b = dict(zip((n for n, _ in self._float_buffers), self.ema_buffers[i])) if self.ema_buffers else {}
        for bname, b in self.model.named_buffers():
            out[bname] = ema_b[bname] if bname in ema_b else b.detach().clone()
        return out

    # ------------------------------------------------------------------ checkpointing (prog/checkpoint_saver.py:115 calls optimizer.state_dict())
    def state_dict(self):
        """torch.optim.AdamW layout: {"state": {i: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [...]} with parameter indices
        in group order, plus the EMA slabs under "ema" (the reference keeps those in separate ModelEma objects)."""
        state, groups, idx = {}, [], 0
        offs = {id(p): off for p, (_, off, _) in zip(self.reducer.params, self._views)}
        for g in self.param_groups:
            ids = []
            for p in g["params"]:
                off, n = offs[id(p)], p.numel()
                state[idx] = {"step": torch.tensor(float(self.step_count)), "exp_avg": self.m[off:off + n].view_as(p).clone(),
                              "exp_avg_sq": self.v[off:off + n].view_as(p).clone()}
                ids.append(idx)
                idx += 1
            groups.append({k: v for k, v in g.items() if k != "params"} | {"params": ids})
        return {"state": state, "param_groups": groups, "ema": [e.clone() for e in self.ema],
                "ema_buffers": [[b.clone() for b in bufs] for bufs in self.ema_buffers]}

    def load_state_dict(self, sd):
        offs = {id(p): off for p, (_, off, _) in zip(self.reducer.params, self._views)}
        idx = 0
        with torch.no_grad():
            for g, saved in zip(self.param_groups, sd["param_groups"]):
                for k, v in saved.items():
                    if k != "params":
                        g[k] = v
                if len(saved["params"]) != len(g["params"]):
                    raise ValueError("FlatAdamWEma.load_state_dict: parameter group sizes differ")
                for p in g["params"]:
                    st = sd["state"].get(idx)
                    if st is not None:
                        off, n = offs[id(p)], p.numel()
                        self.m[off:off + n].copy_(st["exp_avg"].reshape(-1))
                        self.v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                        self.step_count = int(st["step"])
                    idx += 1
            for dst, src in zip(self.ema, sd.get("ema", [])):
                dst.copy_(src)
            for bufs, saved in zip(self.ema_buffers, sd.get("ema_buffers", [])):
                for dst, src in zip(bufs, saved):
                    dst.copy_(src)
