"""A training step as ONE HIP graph (missing 4; SURVEY.md section 8(f) row N2 / BASELINE configs[2]).

The early AutoProg stages -- (l, r) = (9, 128), (12, 160): main_prog.py:973-974 resizes every batch to the stage's resolution,
:1824-1836 picks a sub-network per step in a search -- are launch-gap bound on an MI355X: ~450 kernels of ~10 us each, and the GPU-side
gap between two dependent launches (~1.5 us) is a tenth of the kernel it precedes.  A graph replay removes those gaps (stage 1: 5.36 ->
4.25 ms per step, profiles/r04_exp_graph.txt).  What stood in the way was that a step reads HOST scalars that change every step:

    the mix-token box and its lam            models/volo.py:319-339, 649-658, 684-691 ; loss/cross_entropy.py:149-152
    the learning rate and Adam's bias corrections   main_prog.py:1019-1027 (scheduler writes param_groups[i]["lr"]; torch.optim.AdamW's step count)

DropPath's draws are device-side already (torch's generator is graph-safe: a replay advances its philox offset).  Here those scalars
live in a small DEVICE buffer (`StepScalars`: ABI 6 entry points ap_mix_token_swap_dev, ap_soft_ce_*_dev, the `step_scalars` argument of
ap_adamw_ema_step) that is refreshed by one 64-byte host-to-device copy in front of every replay; the host draws the box with the
same numpy calls in the same order as the eager forward, so a graphed run and an eager run with the same seeds see the same boxes.

    gs = GraphedStep(model, loss_fn, reducer, opt, images, target)      # static input buffers (copied from the given tensors)
    gs.capture()                                                        # warm-up steps on a side stream, then one captured step
    loss = gs.step(new_images, new_target)                              # copies the batch in, refreshes the scalars, replays

One graph per elastic configuration: `set_sample_config` changes which kernels run, so the driver keeps a GraphedStep per (l, r).
Single-rank only (RCCL collectives are not captured here): with world > 1 the eager step runs."""
import numpy as np
import torch

from . import ops


class StepScalars:
    """the per-step host scalars of a training step in device memory.  Layout (16 x 4 bytes; include/autoprog_hip.h ap_step_scalars):
         int32 [0..3]  mix-token box on the token-label grid: bbx1, bbx2, bby1, bby2 (rows [bbx1, bbx2), columns [bby1, bby2))
         fp32  [4]     lam of the loss = 1 - box area / N
         fp32  [5..7]  learning rate, 1 - beta1^t, sqrt(1 - beta2^t)
       The host side is a RING of pinned 64-byte slots: push() is one asynchronous copy of the slot being prepared, and a pinned
       copy reads its source when it EXECUTES on the GPU, not when it is enqueued -- with one slot the host, which runs many replays
       ahead of the GPU (a replay costs it tens of microseconds, the step milliseconds), would overwrite step t's scalars with
       step t + k's before the copy of step t has run.  begin() takes the next slot and waits for the event recorded
       behind that slot's last copy, so the host runs at most SLOTS - 1 steps ahead of the copies."""
    SLOTS = 8

    def __init__(self, device):
        self.ring = torch.zeros(self.SLOTS, 16, dtype=torch.int32).pin_memory()
        self.events = [None] * self.SLOTS
        self.slot = 0
        self.dev = torch.zeros(16, dtype=torch.int32, device=device)
        self._bind()
        self._hf[4] = 1.0
        self.push()

    def _bind(self):
        self.host = self.ring[self.slot]
        self._hf = self.host.view(torch.float32)

    def begin(self):
        """start preparing a step: the next slot of the ring, once the copy that last read it has executed; the values of the
        previous step carry over (a caller may set only the box or only the optimizer scalars)"""
        prev = self.host
        self.slot = (self.slot + 1) % self.SLOTS
        ev = self.events[self.slot]
        if ev is not None:
            ev.synchronize()
        self._bind()
        self.host.copy_(prev)

    @property
    def box_ptr(self):
        return self.dev.data_ptr()

    @property
    def lam_ptr(self):
        return self.dev.data_ptr() + 4 * 4

    @property
    def adam_ptr(self):
        return self.dev.data_ptr() + 5 * 4

    def set_box(self, box, n_tokens):
        bbx1, bby1, bbx2, bby2 = (int(v) for v in box)
        self.host[0], self.host[1], self.host[2], self.host[3] = bbx1, bbx2, bby1, bby2
        self._hf[4] = 1.0 - ((bbx2 - bbx1) * (bby2 - bby1) / float(n_tokens))
        self.box = (bbx1, bby1, bbx2, bby2)

    def set_adam(self, lr, beta1, beta2, step):
        # exactly what ap_adamw_ema_step derives from its float arguments (csrc/optim.hip): double arithmetic on the fp32-rounded betas
        b1, b2 = float(np.float32(beta1)), float(np.float32(beta2))
        self._hf[5] = float(lr)
        self._hf[6] = 1.0 - b1 ** int(step)
        self._hf[7] = float(np.sqrt(1.0 - b2 ** int(step)))

    def push(self):
        self.dev.copy_(self.host, non_blocking=True)
        if self.dev.is_cuda:
            ev = self.events[self.slot]
            if ev is None:
                ev = self.events[self.slot] = torch.cuda.Event()
            ev.record()


class DeviceBox:
    """what a VOLO forward in graph mode returns in place of the (bbx1, bby1, bbx2, bby2) tuple: the loss reads lam from the device"""

    def __init__(self, scalars):
        self.scalars = scalars

    def __iter__(self):                      # callers that unpack the box get the host copy of the step being prepared
        return iter(self.scalars.box)


class GraphedStep:
    def __init__(self, model, loss_fn, reducer, opt, images, target, clip_grad=None, clip_mode="norm"):
        if getattr(reducer, "world", 1) > 1:
            raise ValueError("GraphedStep: single-rank only (the gradient exchange is not captured)")
        self.model, self.loss_fn, self.reducer, self.opt = model, loss_fn, reducer, opt
        self.clip_grad, self.clip_mode = clip_grad, clip_mode
        self.images = images.clone()
        self.target = _clone_target(target)
        self.scalars = StepScalars(self.images.device)
        self.graph = None
        self.loss = None
        # the resolution of the elastic configuration this graph is built on (the model's at construction): a replay's mix-token draw is for THIS
        # configuration's token grid whatever configuration the model object was switched to since (a search replays a different graph every step)
        pe = getattr(model, "patch_embed", None)
        self._res = (getattr(pe, "resize_to", None) or self.images.shape[-1]) if pe is not None else self.images.shape[-1]

    # ---- host side of a step: the draws VOLO.forward makes (models/volo.py:649-653: beta, then rand_bbox's two randint calls)
    def _draw(self):
        m = self.model
        if getattr(m, "mix_token", False) and m.training:
            from .models.volo import rand_bbox
            r = self._res
            pe = m.patch_embed
            patch = pe.proj.kernel_size[0] * (pe.conv[0].stride[0] if pe.stem_conv else 1)
            g1 = r // patch                                       # token grid in front of the first stage (patch size 8 in every VOLO)
            lam = np.random.beta(m.beta, m.beta)
            box = rand_bbox((self.images.shape[0], g1, g1, 0), lam, scale=m.pooling_scale)
            n = (g1 // m.pooling_scale) ** 2
        else:
            box, n = (0, 0, 0, 0), 1
        self.scalars.set_box(box, n)

    def _prepare(self):
        self.scalars.begin()
        self._draw()
        o = self.opt
        self.scalars.set_adam(o.param_groups[0]["lr"], o.betas[0], o.betas[1], o.step_count + 1)
        self.scalars.push()

    def _step_body(self):
        self.reducer.zero_grad()
        loss = self.loss_fn(self.model(self.images), self.target)
        loss.backward()
        self.reducer.finish()
        self.opt.step(clip_grad=self.clip_grad, clip_mode=self.clip_mode, scalars=self.scalars)
        return loss

    def capture(self, warmup=3):
        self.model.step_scalars = self.scalars
        if self.clip_grad is not None and float(self.clip_grad) > 0 and self.clip_mode == "norm":
            self.opt._clip_workspace(self.images.device)          # (never first allocated inside the capture: optim.FlatAdamWEma._clip_workspace)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._prepare()
                self._step_body()
        torch.cuda.current_stream().wait_stream(s)
        rng = np.random.get_state()            # the capture prepares a step it does not run: its mix-token draw is handed back
        self._prepare()
        count = self.opt.step_count
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._step_body()
        self.opt.step_count = count            # the capture recorded a step, it did not run one
        np.random.set_state(rng)
        return self

    def step(self, images=None, target=None):
        if self.graph is None:
            raise RuntimeError("GraphedStep.step() before capture()")
        if images is not None:
            self.images.copy_(images, non_blocking=True)
        if target is not None:
            _copy_target(self.target, target)
        self.model.step_scalars = self.scalars
        self._prepare()
        self.graph.replay()
        self.opt.step_count += 1
        return self.loss

    def release(self):
        if getattr(self.model, "step_scalars", None) is self.scalars:
            self.model.step_scalars = None


def _clone_target(t):
    from .loss.cross_entropy import SparseTokenLabelTarget
    if isinstance(t, SparseTokenLabelTarget):
        return SparseTokenLabelTarget(t.idx.clone(), t.val.clone(), t.smoothing)
    return t.clone()


def _copy_target(dst, src):
    from .loss.cross_entropy import SparseTokenLabelTarget
    if isinstance(dst, SparseTokenLabelTarget):
        dst.idx.copy_(src.idx, non_blocking=True)
        dst.val.copy_(src.val, non_blocking=True)
    else:
        dst.copy_(src, non_blocking=True)
