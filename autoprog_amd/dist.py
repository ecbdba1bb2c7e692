"""Data-parallel gradient exchange for the AutoProg step (reference: ApexDDP(delay_allreduce=True)
/ torch DDP wrappers at main_prog.py:538-550, 1412-1425 -- mean of gradients across ranks).

MI355X design: one process per GPU, RCCL through
torch.distributed (backend "nccl" on ROCm).  All parameter gradients live in ONE flat fp32 slab;
parameters are grouped into buckets in reverse registration order (~ backward order) and a
bucket's all-reduce is launched asynchronously from a post-accumulate-grad hook as soon as its
last gradient is produced, so the exchange overlaps the rest of backward.  xGMI is point-to-point
(7 links/GPU): buckets are sized large (default 32 MiB) so each collective can use every link.

Elastic depth leaves skipped layers without gradients; every rank runs the same (r, l) config per
step (reference: random.seed(epoch), main_prog.py:1861), so `finish()` reduces the remaining
buckets in the same order on every rank (their untouched gradients are zero).
"""
import torch
import torch.distributed as dist


class GradientBucketReducer:
    def __init__(self, params, bucket_bytes=32 << 20, process_group=None, world_size=None, defer_mean=False, comm_dtype=None,
                 accumulate_steps=1):
        """accumulate_steps: micro-batches per optimizer update (the reference's --batch-splits: main_prog.py:567-574 picks the split
        count per stage, :971 `update = (batch_idx + 1) % batch_splits == 0`, :1019-1027 / prog/scaler.py:60-68 run backward on
        loss / batch_splits every micro-batch and step the optimizer only on the update one).  Gradients of the first k - 1
        micro-batches only accumulate in the slab (every kernel adds in place): no parameter is counted ready, no bucket leaves;
        the k-th backward pass drives the bucketed exchange as usual.  finish() closes a micro-batch; `is_update_step` says whether
        the one just closed was the k-th.  zero_grad() starts a new update.
        defer_mean: finish() leaves the all-reduced SUM in the slab and the consumer applies 1/world itself
        (optim.FlatAdamWEma folds it into the fused update kernel via take_pending_scale(): no extra pass over the slab).
        comm_dtype: torch.bfloat16 (or AP_GRAD_COMM_DTYPE=bf16) sends every bucket as bf16 -- half the bytes per xGMI link; the slab
        and the sum stay fp32 on each rank (the bucket is rounded once before the exchange, the exchanged sum is written back as fp32).
        Default: fp32 buckets, the reference's arithmetic (ApexDDP / DDP all-reduce fp32 gradients, main_prog.py:538-550)."""
        import os
        if comm_dtype is None and os.environ.get("AP_GRAD_COMM_DTYPE", "").lower() in ("bf16", "bfloat16"):
            comm_dtype = torch.bfloat16
        self.comm_dtype = comm_dtype if comm_dtype not in (None, torch.float32) else None
        self._staged = []                  # (bucket range, bf16 copy) of the buckets in flight
        self.accumulate_steps = max(1, int(accumulate_steps))
        self._micro = 0                    # micro-batches of the current update already closed by finish()
        self.is_update_step = True
        self.launch_log = []               # (bucket, parameters already delivered when it left) -- tests read the overlap from it
        self.defer_mean = defer_mean
        self._pending_scale = 1.0
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = world_size if world_size is not None else (dist.get_world_size(process_group) if dist.is_initialized() else 1)
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, dtype=torch.float32, device=ref.device)
        # gradients become views of the slab; buckets are contiguous slab ranges
        order = list(reversed(self.params))
        self.buckets = []          # (start, end, [param indices in `order`])
        off, cur_start, cur_members = 0, 0, []
        self._bucket_of = {}
        for i, p in enumerate(order):
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            cur_members.append(i)
            off += n
            if (off - cur_start) * 4 >= bucket_bytes:
                self.buckets.append((cur_start, off, cur_members))
                cur_start, cur_members = off, []
        if cur_members:
            self.buckets.append((cur_start, off, cur_members))
        for b, (_, _, members) in enumerate(self.buckets):
            for i in members:
                self._bucket_of[id(order[i])] = b
        self._pending = [len(m) for (_, _, m) in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._handles = []
        self._seen = set()
        self._held = set()
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params] if self.world > 1 else []
        self._owned = {id(p) for p in self.params}
        self._no_sink = set()

    # ------------------------------------------------------------------ gradient sink (functional.set_grad_sink)
    def owns(self, p):
        return id(p) in self._owned and id(p) not in self._no_sink and p.grad is not None

    def exclude_from_sink(self, params):
        """parameters used MORE THAN ONCE per forward (VOLO applies `norm` to the class token and to the tokens) must not be
        delivered through the sink: the first backward use would mark them ready and a small bucket's all-reduce could start
        before the second use has accumulated.  Their gradients go through autograd, which sums every use and then fires
        the post-accumulate hook exactly once."""
        self._no_sink.update(id(p) for p in params)

    def needs_stream_join(self):
        """with more than one rank a ready parameter may trigger a bucket all-reduce right away"""
        return self.world > 1

    def param_ready(self, p):
        self._held.discard(id(p))
        if self.world > 1:
            self._on_grad(p)

    def set_accumulate_steps(self, k):
        """between updates only (the driver changes the split count with the stage, main_prog.py:842): after the finish() that
        closed an update, or after zero_grad() -- which abandons whatever update was open"""
        if not (self._micro == 0 or self._micro >= self.accumulate_steps):
            raise RuntimeError("set_accumulate_steps() inside an update: %d of %d micro-batches closed (zero_grad() starts over)"
                               % (self._micro, self.accumulate_steps))
        self.accumulate_steps = max(1, int(k))
        self._micro = 0

    def _accumulating(self):
        """True while the running micro-batch is not the last of its update: its gradients only add up in the slab"""
        return self._micro < self.accumulate_steps - 1

    def completes_a_bucket(self, held_params):
        """would releasing these held parameters complete a bucket that has not left yet?  (functional's weight-gradient window asks
        before it decides to launch early)"""
        if self.world <= 1 or self._accumulating():
            return False
        waiting = {}
        for p in held_params:
            if id(p) in self._held and id(p) not in self._seen:
                b = self._bucket_of.get(id(p))
                if b is not None:
                    waiting[b] = waiting.get(b, 0) + 1
        return any(not self._launched[b] and self._pending[b] == n for b, n in waiting.items())

    def hold(self, params):
        """the gradients of these parameters are still to be written (functional's weight-gradient window holds their problems): autograd
        runs their AccumulateGrad nodes -- and the post-accumulate hook -- when the block's backward returns, with nothing to accumulate;
        that must not count as ready.  param_ready() releases them."""
        self._held.update(id(p) for p in params if p is not None)

    def install_sink(self, model=None):
        """let the fused block backward passes accumulate straight into the slab (no per-parameter
        temporaries, no autograd add kernels).  `model.multi_use_parameters()` (when present) names the
        parameters that stay on the autograd path (see exclude_from_sink)."""
        from . import functional
        if model is not None and hasattr(model, "multi_use_parameters"):
            self.exclude_from_sink(model.multi_use_parameters())
        functional.set_grad_sink(self)

    def uninstall_sink(self):
        from . import functional
        functional.set_grad_sink(None)

    # ------------------------------------------------------------------ hooks
    def _on_grad(self, p):
        # idempotent per step: autograd may still run the AccumulateGrad node (and this hook) of a parameter
        # whose gradient the fused backward already delivered through param_ready()
        if id(p) in self._seen or id(p) in self._held or self._accumulating():
            return
        self._seen.add(id(p))
        b = self._bucket_of[id(p)]
        self._pending[b] -= 1
        if self._pending[b] == 0 and not self._launched[b]:
            self._launch(b)

    def _launch(self, b):
        s, e, _ = self.buckets[b]
        self._launched[b] = True
        self.launch_log.append((b, len(self._seen)))
        buf = self.flat[s:e]
        if self.comm_dtype is not None:
            buf = buf.to(self.comm_dtype)
            self._staged.append((s, e, buf))
        self._handles.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    # ------------------------------------------------------------------ step API
    def zero_grad(self):
        """gradients stay attached to the slab (never set to None)"""
        from . import functional
        functional.reset_wgrad_window()         # (a backward pass that raised may have left problems behind)
        self.flat.zero_()
        self._pending_scale = 1.0               # a deferred 1/world that nobody consumed dies with the gradients it belonged to
        self._pending = [len(m) for (_, _, m) in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._handles = []
        self._staged = []
        self._seen = set()
        self._held = set()
        self._micro = 0
        self.launch_log = []

    def finish(self):
        """call after backward(): reduce buckets whose hooks did not all fire (skipped layers), wait
        for every collective and turn sums into means."""
        from . import functional
        functional.flush_wgrad_window()         # (empty after a backward pass: the autograd engine's final callback has flushed it)
        functional.join_wgrad_stream()          # side-stream weight gradients land before anyone reads the slab
        if self._accumulating():                # a micro-batch that only accumulates: nothing is exchanged, nothing was marked ready
            self._micro += 1
            self._held = set()
            self.is_update_step = False
            return
        self._micro = 0 if self.accumulate_steps == 1 else self.accumulate_steps     # (closed: zero_grad() opens the next update)
        self.is_update_step = True
        if self.world <= 1:
            return
        for b in range(len(self.buckets)):
            if not self._launched[b]:
                self._launch(b)
        for h in self._handles:
            h.wait()
        for s, e, buf in self._staged:
            self.flat[s:e].copy_(buf)
        self._staged = []
        if self.defer_mean:
            self._pending_scale = 1.0 / self.world
        else:
            self.flat.mul_(1.0 / self.world)

    @property
    def grad_scale(self):
        """what a reader of p.grad between finish() and the optimizer step must multiply by (gradient clipping, norm logging,
        checkpoints): 1/world while a deferred mean is pending (defer_mean=True leaves the all-reduced SUM in the slab for
        ap_adamw_ema_step to scale), else 1.  Reading does not consume it."""
        return self._pending_scale

    def take_pending_scale(self):
        """factor that still has to be applied to the slab (1/world after a deferred-mean finish(), else 1); reading resets it"""
        s, self._pending_scale = self._pending_scale, 1.0
        return s

    def remove(self):
        self.uninstall_sink()
        for h in self._hooks:
            h.remove()
        self._hooks = []


def reduce_scalar_mean(t, world_size, group=None):
    """timm reduce_tensor (main_prog.py:1043): all-reduce(SUM)/n on a clone"""
    rt = t.detach().clone()
    if world_size > 1:
        dist.all_reduce(rt, op=dist.ReduceOp.SUM, group=group)
        rt /= world_size
    return rt


def distribute_bn(model, world_size, reduce=False, group=None, named_buffers=None):
    """timm.utils.distribute_bn as main_prog.py:883-887 calls it after every epoch (`--dist-bn reduce | broadcast`): every rank gets the
    same BatchNorm running means / variances -- averaged over the ranks (reduce) or rank 0's (broadcast).  timm issues one collective
    per buffer; here the buffers (six vectors of the stem's width for VOLO) travel as ONE flat message and are scattered back.
    named_buffers: (name, tensor) pairs to use instead of model.named_buffers() -- the EMA copies of the buffers live in the flat
    optimizer (FlatAdamWEma.ema_buffers), not in modules of their own (main_prog.py:899, 1650-1654: distribute_bn(model_ema_list[idx]))."""
    src = model.named_buffers() if named_buffers is None else named_buffers
    bufs = [b for n, b in src if ("running_mean" in n or "running_var" in n)]
    if world_size <= 1 or not bufs:
        return
    flat = torch.cat([b.detach().reshape(-1).float() for b in bufs])
    if reduce:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat /= float(world_size)
    else:
        dist.broadcast(flat, 0, group=group)
    off = 0
    with torch.no_grad():
        for b in bufs:
            n = b.numel()
            b.copy_(flat[off:off + n].view_as(b).to(b.dtype))
            off += n
