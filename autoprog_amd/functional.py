"""Autograd layer over the HIP kernels: hand-written forward/backward pairs for the heavy
blocks (Transformer, Outlooker) plus fine-grained Functions (LayerNorm, Linear, class
attention, dense CE, mix-token, positional add) used by the thin tail of the network.

Precision contract (the bf16 counterpart of the reference's apex-O1 recipe):
fp32 master parameters, bf16 activations and bf16 copies of the Linear weights (recast once
per optimizer step), fp32 MFMA accumulation, fp32 LayerNorm / softmax / loss statistics and
fp32 parameter gradients.
"""
import math

import os

import torch

from . import ops
from ._lib import AutoProgHipError

BF16 = torch.bfloat16


# ------------------------------------------------------------------------------ weight bank
class _WeightBank:
    """bf16 (and transposed bf16) copies of fp32 Linear weights, refreshed when the parameter's
    storage or version counter changes (i.e. after optimizer.step() / load_state_dict()); the
    copies maintained by optim.FlatAdamWEma are trusted only while the parameter's version
    counter equals the one stamped at their last refresh (an in-place write from outside --
    load_state_dict, manual edits -- falls back to the keyed path until FlatAdamWEma.resync()).  Replaces
    apex's per-call casts.  The copies are stored ON the Parameter object, so they can never
    outlive it or be confused with another parameter that reuses its address."""

    generation = 0          # bumped by optimizers that update parameters through raw pointers (optim.FlatAdamWEma)

    @classmethod
    def _key(cls, p):
        return (p.data_ptr(), p._version, cls.generation, tuple(p.shape))

    @staticmethod
    def _flat(p):
        w = p.detach()
        return w.reshape(w.shape[0], -1).contiguous()

    def get(self, p):
        if not isinstance(p, torch.nn.Parameter):      # derived tensor (e.g. permuted conv weight): no caching
            return ops.cast_bf16(self._flat(p))
        flat = getattr(p, "_ap_flat16", None)          # kept current by optim.FlatAdamWEma (no cast kernels at all)
        if flat is not None and flat[0] == p.data_ptr() and flat[3] == p._version:
            return flat[1]
        ent = getattr(p, "_ap_bf16", None)
        key = self._key(p)
        if ent is None or ent[0] != key:
            ent = (key, ops.cast_bf16(self._flat(p)))
            p._ap_bf16 = ent
        return ent[1]

    def get_t(self, p):
        if not isinstance(p, torch.nn.Parameter):
            return ops.cast_transpose_bf16(self._flat(p))
        flat = getattr(p, "_ap_flat16", None)
        if flat is not None and flat[0] == p.data_ptr() and flat[3] == p._version and flat[2] is not None:
            return flat[2]
        ent = getattr(p, "_ap_bf16_t", None)
        key = self._key(p)
        if ent is None or ent[0] != key:
            ent = (key, ops.cast_transpose_bf16(self._flat(p)))
            p._ap_bf16_t = ent
        return ent[1]

    @staticmethod
    def clear(module):
        for p in module.parameters():
            p.__dict__.pop("_ap_bf16", None)
            p.__dict__.pop("_ap_bf16_t", None)


bank = _WeightBank()


def _zeros_like_params(params):
    """one zeroed fp32 slab + per-parameter views (gradient accumulators for one block)"""
    sizes = [p.numel() if p is not None else 0 for p in params]
    total = sum(sizes)
    ref = next(p for p in params if p is not None)
    slab = torch.zeros(total, dtype=torch.float32, device=ref.device)
    out, off = [], 0
    for p, n in zip(params, sizes):
        out.append(slab[off:off + n].view(p.shape) if p is not None else None)
        off += n
    return out


# Optional gradient sink (installed by autoprog_amd.dist.GradientBucketReducer): the block-level
# backward passes then accumulate parameter gradients IN PLACE into param.grad (views of one flat fp32
# slab) and signal readiness themselves, instead of returning fresh tensors for autograd to add.
_grad_sink = None


def set_grad_sink(sink):
    global _grad_sink
    _grad_sink = sink


def _param_grad_buffers(params):
    """-> (buffers, sink_used).  With a sink: param.grad itself; otherwise one zeroed slab."""
    sink = _grad_sink
    if sink is not None and all(p is None or sink.owns(p) for p in params):
        return [p.grad if p is not None else None for p in params], True
    return _zeros_like_params(params), False


def _finish_param_grads(params, bufs, sink_used, deferred=False):
    if not sink_used:
        join_wgrad_stream()              # autograd consumes these buffers on the current stream
        return bufs
    if deferred:                         # the weight-gradient window delivers them (flush_wgrad_window)
        return [None] * len(params)
    if _grad_sink.needs_stream_join():
        join_wgrad_stream()              # a bucket all-reduce may be launched from param_ready
    for p in params:
        if p is not None:
            _grad_sink.param_ready(p)
    return [None] * len(params)


def _g2(w):
    """2-D view of a (possibly conv-shaped) weight gradient buffer"""
    return w.view(w.shape[0], -1)


# Weight gradients do not feed the backward chain, so they CAN be issued on a side stream and overlap the
# input-gradient GEMMs / LayerNorm / attention kernels of the main stream.  That paid (23.1 -> 22.0 ms/step) while
# every layer's weight gradient was its own under-filled launch; with one grouped launch per block (wgrad_batch,
# 450-512 workgroups = every CU twice) the side stream only adds contention: 19.55 ms/step vs 19.22 on one stream
# (same box, back to back).  Default: one stream; AP_ASYNC_WGRAD=1 switches the side stream on.
async_wgrad = os.environ.get("AP_ASYNC_WGRAD", "0") == "1"
_side_streams = {}


def wgrad_stream(device=None):
    dev = torch.cuda.current_device() if device is None else device
    st = _side_streams.get(dev)
    if st is None:
        st = torch.cuda.Stream(device=dev)
        _side_streams[dev] = st
    return st


def join_wgrad_stream():
    """make the current stream wait for every weight-gradient kernel issued so far"""
    if _side_streams:
        st = _side_streams.get(torch.cuda.current_device())
        if st is not None:
            torch.cuda.current_stream().wait_stream(st)


# Weight gradients of one block are collected and issued as ONE grouped launch (ops.gemm_tn_acc_grouped): the
# launch's workgroups are shared by the block's 4-5 Linear layers, so every layer is split over fewer token
# ranges -> longer reduction loops and several times fewer fp32 atomics than one launch per layer.
_wgrad_batch = None


class wgrad_batch:
    """with wgrad_batch(): ... _linear_bwd calls ... ; the collected weight gradients launch on exit -- or, with `sunk` (the gradients
    accumulate in place into param.grad, nothing is returned to autograd), join the weight-gradient window of the backward pass"""

    def __init__(self, sunk=False, params=()):
        self.sunk, self.params, self.deferred = sunk, params, False

    def __enter__(self):
        global _wgrad_batch
        self.prev = _wgrad_batch
        _wgrad_batch = []
        self.ln = []                  # deferred LayerNorm dgamma/dbeta reductions of the block: ops.layernorm_bwd(..., defer=batch.ln)
        return self

    def __exit__(self, *exc):
        global _wgrad_batch
        pending, _wgrad_batch = _wgrad_batch, self.prev
        if exc[0] is None:
            if self.sunk and pending and _window_add(pending, self.ln, self.params):
                self.deferred = True
            elif pending and not async_wgrad and fuse_ln_reduce:
                _launch_wgrads(pending, self.ln)                 # the block's LayerNorm dgamma / dbeta reductions ride in the weight-gradient launch
            else:
                if self.ln:
                    ops.layernorm_bwd_reduce_batched(self.ln)    # one launch for the block's LayerNorms (was one per LayerNorm)
                if pending:
                    _launch_wgrads(pending)
        return False


# The weight-gradient window.  A block's own launch has 40 output tiles (192 x 192) for 256 CUs, so every problem is cut into ~6 token
# ranges whose partial tiles meet in fp32 atomics: 21-27 us of a ~100 us launch, bound by the chip's atomic rate, not by anything the
# kernel does.  Nothing in the backward chain reads a weight gradient, so the blocks' problems are collected -- operands kept alive by
# the references held here -- and launched once about one tile per CU has come together (six transformer blocks of VOLO-D1): no token
# axis is cut, the tiles leave as plain read-add-stores (ops.gemm_tn_acc_grouped / k_gemm_tn_8p), results become reproducible.
# The end of the backward pass flushes what is left (an autograd engine callback).  Only with a gradient sink: the gradients land in
# param.grad in place, and the sink hears param_ready() at the flush instead of at the end of the block.
# AP_WGRAD_WINDOW: tiles per launch (0 = one launch per block, the behaviour before).
WGRAD_WINDOW = int(os.environ.get("AP_WGRAD_WINDOW", "256"))
# The window holds UNITS -- one weight-gradient problem or one LayerNorm rider each, with the parameters whose gradient that unit
# completes -- and launches the longest prefix that fits the tile kernel's table (WGRAD_WINDOW tiles = one per CU): a launch ends in
# the middle of a block when that fills it.  VOLO-D5's blocks are 64 + 64 + 16 + 48 = 192 tiles: a block per launch left a quarter of
# the chip idle, block-and-a-third launches fill it (three launches of 256 for four blocks); D1's 40-tile blocks pack 252 instead of 240.
_window = {"units": [], "tiles": 0, "armed": None, "outs": set()}

# Two private hooks of the autograd engine make the window self-flushing: queue_callback (run at the end of the backward pass the
# caller is inside) and _current_graph_task_id (which backward pass that is).  Both are probed once; without them the window still
# works and is flushed by GradientBucketReducer.finish() -- the sink's contract is "call finish() after backward()" either way.
_ENGINE = getattr(getattr(torch.autograd, "Variable", None), "_execution_engine", None)
_graph_task_id = getattr(torch._C, "_current_graph_task_id", None)
_HAS_ENGINE_CALLBACK = hasattr(_ENGINE, "queue_callback") and _graph_task_id is not None     # (a callback is queued once per pass: both or neither)


def _tiles_192(prob):
    a, c, n1, n2 = prob[0], prob[2], prob[3], prob[4]
    n1 = c.shape[0] if n1 is None else n1
    n2 = c.shape[1] if n2 is None else n2
    if n1 % 192 or n2 % 192 or a.shape[0] % 64 or a.shape[0] < 4096 or (len(prob) > 9 and prob[9] is not None):
        return 0                      # not a problem of the 192 x 192-tile kernel: rides along, costs no slot
    return (n1 // 192) * (n2 // 192)


def _window_units(problems, ln, params):
    """the block's problems and LayerNorm riders as units [kind, item, tiles, parameters completed, addresses written]"""
    owner = {}
    for p in params:
        if p is not None and p.grad is not None:
            owner.setdefault(p.grad.data_ptr(), []).append(p)
    units, claimed = [], set()

    def take(ptrs):
        ps = []
        for a in ptrs:
            for p in owner.get(a, ()):
                if id(p) not in claimed:
                    claimed.add(id(p))
                    ps.append(p)
        return ps
    for q in problems:
        ptrs = [q[2].data_ptr()] + ([q[5].data_ptr()] if q[5] is not None else [])
        units.append(["p", q, _tiles_192(q), take(ptrs), set(ptrs)])
    for item in ln:
        ptrs = [item[3].data_ptr(), item[4].data_ptr()]
        units.append(["l", item, 0, take(ptrs), set(ptrs)])
    if units:                         # a parameter no unit writes (there is none in the shipped blocks) leaves with the block's last unit
        units[-1][3] += [p for p in params if p is not None and id(p) not in claimed]
    return units


def _window_add(problems, ln, params):
    """-> True when the window took the block's weight gradients"""
    from ._lib import TN_MAX_GROUP, LN_MAX_BATCH
    if WGRAD_WINDOW <= 0 or async_wgrad or not fuse_ln_reduce or ops.deterministic:
        return False
    if len(problems) > TN_MAX_GROUP or len(ln) > LN_MAX_BATCH:
        return False
    w = _window
    gid = _graph_task_id() if _graph_task_id is not None else 0
    if w["armed"] != gid:
        # first block of THIS backward pass.  Whatever the window still holds belongs to a pass that raised (the engine runs no
        # callbacks then): those gradients are void, and the operands they pin are released here.
        if w["armed"] is not None:
            reset_wgrad_window()
        if _HAS_ENGINE_CALLBACK:
            if gid == -1:
                return False          # a block backward called outside an engine pass: nothing would flush the window -- the block launches its own
            try:                      # inside a backward pass: the engine calls back when it is over
                _ENGINE.queue_callback(flush_wgrad_window)
            except RuntimeError:
                return False
        w["armed"] = gid
    units = _window_units(problems, ln, params)
    outs = set()
    for u in units:
        outs |= u[4]
    # a parameter that is ALREADY in the window (a block applied twice before one backward, shared weights): its LayerNorm riders
    # add with plain read-modify-writes and tn8_plan only sees duplicates inside one call -- launch what is held first, so the
    # two uses are ordered by the stream like the one-launch-per-block path orders them.
    if w["units"] and (outs & w["outs"]):
        _window_launch()
    if hasattr(_grad_sink, "hold"):
        _grad_sink.hold(params)       # (autograd fires their post-accumulate hooks when the block's backward returns)
    w["units"] += units
    w["tiles"] += sum(u[2] for u in units)
    w["outs"] |= outs
    while w["tiles"] >= WGRAD_WINDOW or _window_counts_full(TN_MAX_GROUP, LN_MAX_BATCH):
        _window_launch_prefix()       # a full table's worth is there: it leaves, the rest of the block waits for the next one
    # data parallel: when everything a gradient bucket still waits for sits in this window, launching now lets the bucket's
    # all-reduce start under the rest of the backward pass (only once the launch is at least 60 % of a full window: a short
    # launch cuts its problems along the token axis again)
    if (w["tiles"] * 10 >= WGRAD_WINDOW * 6 and hasattr(_grad_sink, "completes_a_bucket") and _grad_sink.needs_stream_join()
            and _grad_sink.completes_a_bucket([p for u in w["units"] for p in u[3]])):
        _window_launch()
    return True


def _window_counts_full(max_problems, max_ln):
    """more problems / riders held than ONE launch takes: a prefix has to go whatever its tile count"""
    u = _window["units"]
    return sum(1 for x in u if x[0] == "p") > max_problems or sum(1 for x in u if x[0] == "l") > max_ln


def _window_launch_prefix():
    """launch the longest prefix of the held units that one launch of the tile kernel takes: at most WGRAD_WINDOW tiles (at least one
    unit), TN_MAX_GROUP problems, LN_MAX_BATCH riders; the parameters those units complete are handed to the gradient sink"""
    from ._lib import TN_MAX_GROUP, LN_MAX_BATCH
    w = _window
    units = w["units"]
    n = tiles = nprob = nln = 0
    while n < len(units):
        kind, _item, t = units[n][0], units[n][1], units[n][2]
        if n and (tiles + t > WGRAD_WINDOW or (kind == "p" and nprob == TN_MAX_GROUP) or (kind == "l" and nln == LN_MAX_BATCH)):
            break
        tiles += t
        nprob += kind == "p"
        nln += kind == "l"
        n += 1
    taken, w["units"] = units[:n], units[n:]
    w["tiles"] -= tiles
    w["outs"] = set()
    for u in w["units"]:
        w["outs"] |= u[4]
    problems = [u[1] for u in taken if u[0] == "p"]
    ln = [u[1] for u in taken if u[0] == "l"]
    if problems:
        ops.gemm_tn_acc_grouped(problems, ln=ln)
    elif ln:
        ops.layernorm_bwd_reduce_batched(ln)
    if _grad_sink is not None:
        for u in taken:
            for p in u[3]:
                _grad_sink.param_ready(p)


def _window_launch():
    """everything the window holds, in as many launches as it takes"""
    while _window["units"]:
        _window_launch_prefix()


def flush_wgrad_window():
    """launch what the window holds (the end of every backward pass does; harmless when it is empty)"""
    _window["armed"] = None
    _window_launch()


def reset_wgrad_window():
    """drop what a backward pass that raised left behind"""
    _window.update(units=[], tiles=0, armed=None, outs=set())


fuse_ln_reduce = os.environ.get("AP_FUSE_LN_REDUCE", "1") != "0"


def _launch_wgrads(problems, ln=None):
    if async_wgrad:
        side = wgrad_stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ops.gemm_tn_acc_grouped(problems)
        for prob in problems:
            prob[0].record_stream(side)
            prob[1].record_stream(side)
    else:
        ops.gemm_tn_acc_grouped(problems, ln=ln)


# The MLPs store gelu'(h) instead of h in the forward (the backward needs nothing else of h) and multiply by it in the backward:
# ~18 VALU instructions per element less in a store-bound epilogue.  Round 5: the derivative is stored in 8 BITS -- it is a multiplier
# in [-0.129, 1.129], fixed point with step 1/202 (0, 1/2 and 1 exact; error <= 1/404, the spacing of bf16 numbers just below 1):
# half the bytes written by every fc1 launch and read by every fc2 input-gradient launch (29 MB each at 25088 x 1152).
# AP_GELU_STORE_GRAD = 2 (default): 8-bit codes; 1: bf16 derivative (rounds 3 - 4); 0: store h and evaluate gelu' in the backward.
STORE_GELU_GRAD = int(os.environ.get("AP_GELU_STORE_GRAD", "2"))


# Round 6: the MLP of a transformer block as ONE launch per direction (csrc/mlp_fused.hip; C = 384, whole 128-row blocks).  Bit-identical to the two
# launches it replaces.  AP_FUSED_MLP: 1 (default) both directions, 2 forward only, 0 off (the two ap_gemm_nt launches).  In the step the forward launch takes
# 77 us against 51 + 35, the backward 70 against 43 + 27 (DESIGN.md section 3 "Round 6": it needed its weights prefetched into L2 to get there).
FUSED_MLP = int(os.environ.get("AP_FUSED_MLP", "1") or 0)
FUSED_MLP_BWD = FUSED_MLP == 1
# AP_FUSED_MLP_LN=1: the LayerNorm in front of fc1 inside the fused forward launch (bit-identical to ap_layernorm_fwd; one launch less per block)
FUSED_MLP_LN = os.environ.get("AP_FUSED_MLP_LN", "1") == "1"
# The fused launch runs ONE 128-row block per CU, so its time does not shrink with the row count while the launches it replaces do: measured from
# graphs on one box, 8192 rows (stage (9, 128 px), 64 workgroups) 4.08 -> 4.23 ms per step with it, 12800 rows 6.13 -> 6.24, 18432 rows 8.95 -> 8.87,
# 25088 rows 12.20 -> 11.96 (profiles/r06_mlp_fused.txt).  Below this many rows the block keeps the two launches.
FUSED_MLP_MIN_ROWS = int(os.environ.get("AP_FUSED_MLP_MIN_ROWS", "18432"))


def _gelu_bwd_kw(h):
    return {"mul_by": h} if STORE_GELU_GRAD else {"dgelu_of": h}


def _gelu_side_buffer(rows, cols, device):
    """what the fc1 launch stores next to its output for the backward: [rows, cols] codes (uint8) or bf16"""
    return torch.empty((rows, cols), dtype=torch.uint8 if STORE_GELU_GRAD == 2 else BF16, device=device)


def _linear_bwd(g, x_in, w, dw, db, n=None, dgelu_of=None, need_dx=True, row_scale=None, rows_per_scale=1, cs_weight=None, inv_keep=1.0, mul_by=None,
                q8=None, g8=None):
    """shared backward of y = x W^T + b given g = dL/dy (bf16 [M, ld]); accumulates dw/db.
    DropPath (the branch output is scaled per sample by s = mask/keep and x_in has the rows of dropped samples zeroed): pass
    row_scale = s (the input gradient gets it in the GEMM epilogue), cs_weight = per-token 0/1 mask and inv_keep = 1/keep:
    dW = inv_keep * g^T x_in and db = inv_keep * sum_m mask[m] g[m] -- g itself is never scaled."""
    n = w.shape[0] if n is None else n
    prob = (g, x_in, _g2(dw), n, _g2(dw).shape[1], db)
    if cs_weight is not None:
        prob = prob + (cs_weight, inv_keep, inv_keep)
    if _wgrad_batch is not None:
        _wgrad_batch.append(prob)
    else:
        _launch_wgrads([prob])
    if not need_dx:
        return None
    wt = bank.get_t(w)                       # [K, ld(N)]
    if g8 is not None:                       # (FP8_DGRAD) g8 = (e4m3 bytes of g, dq): the input-gradient product on the transposed e4m3 weight
        w8t, dq_w = fp8_weights.get_t(w)
        return ops.gemm_nt_fp8(g8[0], w8t, g8[1], dq_w, n=wt.shape[0], row_scale=row_scale, rows_per_scale=rows_per_scale)
    # q8 = (scale, amax): -> (dx, dx8), the input gradient a second time as e4m3 (the operand of the next fp8 input-gradient product)
    return ops.gemm_nt(g, wt, n=wt.shape[0], k=wt.shape[1], dgelu_of=dgelu_of, mul_by=mul_by, row_scale=row_scale, rows_per_scale=rows_per_scale, q8=q8)


def token_mask(keep01, n_tokens):
    """per-token bf16 keep mask (padded to whole 16-byte chunks) of a per-sample 0/1 mask (slow path; the models prefetch the
    masks of a whole forward pass in one go, models/volo.py DropPathRng.prefetch)"""
    m = keep01.to(BF16).repeat_interleave(n_tokens)
    pad = (-m.numel()) % 8
    return torch.cat([m, m.new_zeros(pad)]) if pad else m.contiguous()


# ---------------------------------------------------------------- fp8 forward GEMMs (BASELINE configs[4]: "mixed MFMA fp8 GEMM")
# With FP8_LINEAR (AP_FP8=1, bench.py --fp8) the four Linear layers of a transformer block run their FORWARD product on OCP e4m3
# operands (ap_gemm_nt_fp8: the fp8 instantiation of the 8-phase kernel, fp32 accumulation, the usual fused epilogue); the backward
# stays bf16 on the bf16 activations the block saves anyway.  Scaling is per tensor and DELAYED: an activation is quantised with
# the scale derived from the amax its site saw in the previous step (ap_quantize_fp8 records the new amax while it quantises: one
# pass, no reduction in front of it); a site's first use scales by its current amax.  Weights are quantised once per optimizer
# step from their bf16 copies.  All scales live in three device vectors; nothing is read back to the host.
FP8_LINEAR = os.environ.get("AP_FP8", "0") == "1"
# Round 5: the INPUT-GRADIENT product of fc1 on e4m3 operands as well (K = the MLP's hidden width, the long reduction of the backward): dL/dh
# leaves the fc2 input-gradient launch a second time as e4m3 bytes (its epilogue, delayed scale of its own site), the transposed weight is
# quantised with the forward copy's scale in the per-step weight launch.  AP_FP8_DGRAD=0: forward products only (round 4).
FP8_DGRAD = os.environ.get("AP_FP8_DGRAD", "1") != "0"


class _Fp8Scales:
    CAP = 4096

    def __init__(self):
        self.amax = None
        self.slots = {}
        self.generation = None

    def _init(self, device):
        self.amax = torch.zeros(self.CAP, dtype=torch.float32, device=device)
        self.scale = torch.ones(self.CAP, dtype=torch.float32, device=device)
        self.dq = torch.ones(self.CAP, dtype=torch.float32, device=device)
        self.slots = {}

    def roll(self):
        """scales of the coming step from the amax recorded since the last roll (sites that saw nothing keep theirs)"""
        if self.amax is None:
            return
        seen = self.amax > 0
        self.scale.copy_(torch.where(seen, ops.FP8_MAX / self.amax.clamp_min(1e-30), self.scale))
        self.dq.copy_(1.0 / self.scale)
        self.amax.zero_()

    def site(self, key, x):
        """-> slot index of the (per-tensor) quantisation site; a new site is scaled by the tensor in front of it"""
        if self.amax is None or self.amax.device != x.device:
            self._init(x.device)
        if self.generation != _WeightBank.generation:          # an optimizer step has passed
            self.generation = _WeightBank.generation
            self.roll()
        i = self.slots.get(key)
        if i is None:
            i = len(self.slots)
            if i >= self.CAP:
                raise AutoProgHipError("fp8: more than %d quantisation sites" % self.CAP)
            self.slots[key] = i
            amax = x.detach().abs().amax().float().clamp_min(1e-12)
            self.scale[i:i + 1].copy_((ops.FP8_MAX / amax).reshape(1))
            self.dq[i:i + 1].copy_((amax / ops.FP8_MAX).reshape(1))
        return i

    def quantize(self, key, x):
        i = self.site(key, x)
        return ops.quantize_fp8(x, self.scale[i:i + 1], self.amax[i:i + 1]), self.dq[i:i + 1]

    def producer(self, key, device):
        """(scale, amax, dq) of a site whose PRODUCER quantises (ops.layernorm_fwd(fp8=...)); None on the site's first use -- the
        tensor to scale by does not exist yet, the caller quantises it afterwards through quantize()"""
        if self.amax is None or self.amax.device != device or key not in self.slots:
            return None
        if self.generation != _WeightBank.generation:
            self.generation = _WeightBank.generation
            self.roll()
        i = self.slots[key]
        return self.scale[i:i + 1], self.amax[i:i + 1], self.dq[i:i + 1]


fp8_scales = _Fp8Scales()


class _Fp8Weights:
    """e4m3 copies of the Linear weights.  A weight is registered (and scaled by its current amax) at its first use; after that all
    registered weights are re-quantised together, once per optimizer step, in ONE launch (ops.quantize_fp8_multi) from the bf16
    copies the weight bank holds, with delayed scales like the activations."""

    def __init__(self):
        self.items = []               # (parameter, slot, persistent e4m3 buffer)
        self.table = None             # device job table + the source pointers it was built from
        self.table_src = None

    def get(self, w):
        key = (w.data_ptr(), w._version, _WeightBank.generation)
        ent = w.__dict__.get("_ap_fp8")
        if ent is not None and ent[0] == key:
            return ent[1], ent[2]
        if ent is None or ent[3] is not self:
            wb = bank.get(w)
            w8, dq = fp8_scales.quantize(("w", id(w)), wb)
            self.items.append((w, fp8_scales.slots[("w", id(w))], w8, False))
            w._ap_fp8 = (key, w8, dq, self)
            return w8, dq
        self.requantize()
        ent = w._ap_fp8
        return ent[1], ent[2]

    def get_t(self, w):
        """e4m3 copy of the TRANSPOSED weight [in, ld(out)] (the B operand of the input-gradient product), scaled like the forward copy"""
        key = (w.data_ptr(), w._version, _WeightBank.generation)
        ent = w.__dict__.get("_ap_fp8_t")
        if ent is not None and ent[0] == key:
            return ent[1], ent[2]
        if ent is None or ent[3] is not self:
            self.get(w)                                                            # the forward copy owns the site
            slot = fp8_scales.slots[("w", id(w))]
            sc = fp8_scales
            w8t = ops.quantize_fp8(bank.get_t(w), sc.scale[slot:slot + 1])
            self.items.append((w, slot, w8t, True))
            w._ap_fp8_t = (key, w8t, sc.dq[slot:slot + 1], self)
            return w8t, sc.dq[slot:slot + 1]
        self.requantize()
        ent = w._ap_fp8_t
        return ent[1], ent[2]

    def requantize(self):
        import numpy as np
        if not self.items:
            return
        sc = fp8_scales
        sc.site(("w", id(self.items[0][0])), bank.get(self.items[0][0]))          # rolls the scales when an optimizer step has passed
        srcs = [bank.get_t(w) if tr else bank.get(w) for (w, _, _, tr) in self.items]
        ptrs = tuple(t.data_ptr() for t in srcs)
        if self.table is None or self.table_src != ptrs:
            tab = np.zeros((len(self.items), 4), dtype=np.int64)                   # ap_fp8_job: x, y, n, (slot, pad)
            for r, ((w, slot, w8, tr), t) in enumerate(zip(self.items, srcs)):
                tab[r] = (t.data_ptr(), w8.data_ptr(), t.numel(), slot)
            self.table = torch.from_numpy(tab).to(srcs[0].device)
            self.table_src = ptrs
        ops.quantize_fp8_multi(self.table, len(self.items), sc.scale, sc.amax)
        for (w, slot, w8, tr) in self.items:
            ent = ((w.data_ptr(), w._version, _WeightBank.generation), w8, sc.dq[slot:slot + 1], self)
            if tr:
                w._ap_fp8_t = ent
            else:
                w._ap_fp8 = ent


fp8_weights = _Fp8Weights()


def _fp8_weight(w):
    return fp8_weights.get(w)


def reset_fp8_state():
    """forget every quantisation site and weight copy (another model in the same process)"""
    global fp8_scales, fp8_weights
    fp8_scales = _Fp8Scales()
    fp8_weights = _Fp8Weights()


def _fp8_ok(x, w):
    return FP8_LINEAR and x.shape[1] % 16 == 0 and w.shape[1] == x.shape[1]


def _linear_fwd(x, w, x8=None, emit_for=None, **kw):
    """y = epilogue(x W^T): bf16, or e4m3 operands under FP8_LINEAR (the K of every VOLO / DeiT Linear is a multiple of 16);
    x8 = (bytes, dq): the producer of x has quantised it already.  emit_for = the weight of the Linear that consumes y (GELU launches):
    -> (y, (y8, dq) or None), the e4m3 operand of that Linear from this launch's epilogue where the kernel can"""
    if _fp8_ok(x, w):
        x8, dq_x = x8 if x8 is not None else fp8_scales.quantize(("x", id(w)), x)
        w8, dq_w = _fp8_weight(w)
        if emit_for is not None:
            site = fp8_scales.producer(("x", id(emit_for)), x.device)
            if site is not None and kw.get("gelu") and ops.gemm_nt_fp8_emits(x.shape[0], w.shape[0], x.shape[1]):
                y, y8 = ops.gemm_nt_fp8(x8, w8, dq_x, dq_w, q8=(site[0], site[1]), **kw)
                return y, (y8, site[2])
            return ops.gemm_nt_fp8(x8, w8, dq_x, dq_w, **kw), None
        return ops.gemm_nt_fp8(x8, w8, dq_x, dq_w, **kw)
    y = ops.gemm_nt(x, bank.get(w), **kw)
    return (y, None) if emit_for is not None else y


def _ln_fwd_for(x, gw, gb, eps, w):
    """LayerNorm whose output feeds the Linear with weight w: under FP8_LINEAR the kernel also emits the e4m3 operand"""
    if _fp8_ok(x, w):
        site = fp8_scales.producer(("x", id(w)), x.device)
        if site is not None:
            y, m, r, y8 = ops.layernorm_fwd(x, gw, gb, eps, fp8=(site[0], site[1]))
            return y, m, r, (y8, site[2])
    y, m, r = ops.layernorm_fwd(x, gw, gb, eps)
    return y, m, r, None


# ----------------------------------------------------------------------- transformer block
class TransformerBlockFn(torch.autograd.Function):
    """x -> x + rs1*(proj(mhsa(qkv(LN1 x)))) -> + rs2*(fc2(gelu(fc1(LN2 .))))
    (Transformer.forward, models/volo.py:230-234; timm Block for DeiT).  rs1/rs2 are the per-sample DropPath factors
    (mask/keep) or None; k1/k2 the 0/1 masks as fp32 [B], tm1/tm2 as per-token bf16, inv_keep = 1/keep.

    DropPath costs no extra pass over the gradients.  The INPUT of each scaled Linear has the rows of dropped samples zeroed
    where it is produced (attention kernel / fc1 epilogue: free, and the forward result is unchanged because those rows are
    multiplied by 0 anyway), so the weight gradient sum_m s[m] dy[m]^T a[m] = (1/keep) dy^T a_masked needs no scaled copy of dy;
    the input gradient takes s in its GEMM epilogue and the bias gradient is a mask-weighted column sum."""

    @staticmethod
    def forward(ctx, x, rs1, rs2, n1w, n1b, qkv_w, qkv_b, proj_w, proj_b, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b,
                B, N, heads, eps, k1=None, k2=None, tm1=None, tm2=None, inv_keep=None):
        C = x.shape[-1]
        if rs1 is not None or rs2 is not None:
            # dW / db of proj and fc2 are formed as inv_keep * dy^T (masked input): the factor cannot be defaulted, and the
            # masks must be binary (rs = mask / keep) -- a caller on the old signature would get gradients scaled by `keep`
            if inv_keep is None:
                raise ValueError("TransformerBlockFn: DropPath factors rs1 / rs2 need inv_keep = 1 / keep_prob")
        elif inv_keep is None:
            inv_keep = 1.0
        x2 = x.reshape(B * N, C).contiguous()
        scale = (C // heads) ** -0.5
        if rs1 is not None and k1 is None:
            k1 = (rs1 != 0).float()
        if rs2 is not None and k2 is None:
            k2 = (rs2 != 0).float()
        xn1, m1, r1, xq = _ln_fwd_for(x2, n1w, n1b, eps, qkv_w)
        qkv = _linear_fwd(xn1, qkv_w, x8=xq, bias=qkv_b)
        oq = None
        site = fp8_scales.producer(("x", id(proj_w)), x.device) if (FP8_LINEAR and ops.mhsa_emits_fp8(N, C // heads)) else None
        if site is not None:          # the attention kernel emits the e4m3 operand of the output projection
            o, lse, o8 = ops.mhsa_fwd(qkv, B, N, heads, scale, out_row_scale=k1, fp8=(site[0], site[1]))
            oq = (o8, site[2])
        else:
            o, lse = ops.mhsa_fwd(qkv, B, N, heads, scale, out_row_scale=k1)                   # rows of dropped samples: zeros
        x1 = _linear_fwd(o, proj_w, x8=oq, bias=proj_b, row_scale=rs1, rows_per_scale=N, residual=x2)
        fused = None
        use_fused = FUSED_MLP and STORE_GELU_GRAD == 2 and not FP8_LINEAR and B * N >= FUSED_MLP_MIN_ROWS and ops.mlp_fused_ok(B * N, C, fc1_w.shape[0])
        if use_fused and FUSED_MLP_LN:
            # LN2 -> fc1 -> GELU -> fc2 (+ DropPath scale + residual) in ONE launch (csrc/mlp_fused.hip): bit-identical to the three launches
            fused = ops.mlp_fused(None, bank.get(fc1_w), bank.get(fc2_w), bias1=fc1_b, bias2=fc2_b, row_scale_hidden=k2, row_scale_out=rs2,
                                  rows_per_scale=N, residual=x1, ln=(x1, n2w, n2b, eps))
            if fused is not None:
                y, a, h, xn2, m2, r2 = fused
        if fused is None:
            xn2, m2, r2, xq = _ln_fwd_for(x1, n2w, n2b, eps, fc1_w)
            if use_fused:
                fused = ops.mlp_fused(xn2, bank.get(fc1_w), bank.get(fc2_w), bias1=fc1_b, bias2=fc2_b, row_scale_hidden=k2, row_scale_out=rs2,
                                      rows_per_scale=N, residual=x1)
                if fused is not None:
                    y, a, h = fused
        if fused is None:
            h = _gelu_side_buffer(B * N, fc1_w.shape[0], x.device)
            a, aq = _linear_fwd(xn2, fc1_w, x8=xq, emit_for=fc2_w, bias=fc1_b, gelu=True, preact_out=h, preact_grad=STORE_GELU_GRAD, row_scale=k2, rows_per_scale=N)
            y = _linear_fwd(a, fc2_w, x8=aq, bias=fc2_b, row_scale=rs2, rows_per_scale=N, residual=x1)
        if rs1 is not None and tm1 is None:
            tm1 = token_mask(k1, N)
        if rs2 is not None and tm2 is None:
            tm2 = token_mask(k2, N)
        ctx.save_for_backward(x2, m1, r1, xn1, qkv, o, lse, x1, m2, r2, xn2, h, a, rs1, rs2, tm1, tm2,
                              n1w, n1b, qkv_w, qkv_b, proj_w, proj_b, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b)
        ctx.cfg = (B, N, heads, scale, float(inv_keep))
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        (x2, m1, r1, xn1, qkv, o, lse, x1, m2, r2, xn2, h, a, rs1, rs2, tm1, tm2,
         n1w, n1b, qkv_w, qkv_b, proj_w, proj_b, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b) = ctx.saved_tensors
        B, N, heads, scale, inv_keep = ctx.cfg
        params = (n1w, n1b, qkv_w, qkv_b, proj_w, proj_b, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b)
        bufs, sunk = _param_grad_buffers(params)
        (dn1w, dn1b, dqkv_w, dqkv_b, dproj_w, dproj_b, dn2w, dn2b, dfc1_w, dfc1_b, dfc2_w, dfc2_b) = bufs
        dy2 = dy.reshape(x2.shape).contiguous()
        with wgrad_batch(sunk, params) as batch:         # the four weight gradients (and the two LayerNorm parameter gradients) launch together: on exit, or with the window's
            # MLP branch
            dh8 = None
            dxn2 = None
            if FP8_LINEAR and FP8_DGRAD and h.shape[1] % 16 == 0 and h.shape[1] == fc1_w.shape[0]:
                # dL/dh as e4m3 next to its bf16 form (the weight gradient keeps reading that): from the launch's epilogue once the site has a
                # scale (its second step on), by a pass of its own before
                site = fp8_scales.producer(("g", id(fc1_w)), dy2.device)
                if site is not None and ops.gemm_nt_emits_q8(dy2.shape[0], h.shape[1], dy2.shape[1], h):
                    dh, d8 = _linear_bwd(dy2, a, fc2_w, dfc2_w, dfc2_b, row_scale=rs2, rows_per_scale=N, cs_weight=tm2, inv_keep=inv_keep,
                                         q8=(site[0], site[1]), **_gelu_bwd_kw(h))
                    dh8 = (d8, site[2])
                else:
                    dh = _linear_bwd(dy2, a, fc2_w, dfc2_w, dfc2_b, row_scale=rs2, rows_per_scale=N, cs_weight=tm2, inv_keep=inv_keep, **_gelu_bwd_kw(h))
                    dh8 = fp8_scales.quantize(("g", id(fc1_w)), dh)
            else:
                fused = None
                if (FUSED_MLP and FUSED_MLP_BWD and h.dtype == torch.uint8 and dy2.shape[0] >= FUSED_MLP_MIN_ROWS
                        and ops.mlp_fused_ok(dy2.shape[0], dy2.shape[1], h.shape[1])):
                    # both input-gradient products of the MLP in one launch; the weight gradients read dL/dh and dL/dy as before
                    fused = ops.mlp_fused(dy2, bank.get_t(fc2_w), bank.get_t(fc1_w), backward=True, codes=h, row_scale_hidden=rs2, rows_per_scale=N)
                if fused is not None:
                    dxn2, dh, _ = fused
                    _linear_bwd(dy2, a, fc2_w, dfc2_w, dfc2_b, need_dx=False, cs_weight=tm2, inv_keep=inv_keep)
                    _linear_bwd(dh, xn2, fc1_w, dfc1_w, dfc1_b, need_dx=False)
                else:
                    dh = _linear_bwd(dy2, a, fc2_w, dfc2_w, dfc2_b, row_scale=rs2, rows_per_scale=N, cs_weight=tm2, inv_keep=inv_keep, **_gelu_bwd_kw(h))
            if dxn2 is None:
                dxn2 = _linear_bwd(dh, xn2, fc1_w, dfc1_w, dfc1_b, g8=dh8)
            dx1 = ops.layernorm_bwd(dxn2, x1, n2w, m2, r2, dy2, dn2w, dn2b, defer=batch.ln)
            # attention branch
            do = _linear_bwd(dx1, o, proj_w, dproj_w, dproj_b, row_scale=rs1, rows_per_scale=N, cs_weight=tm1, inv_keep=inv_keep)
            dqkv = ops.mhsa_bwd(qkv, o, do, lse, B, N, heads, scale)      # dropped samples: do = 0, so the masked rows of o do not matter
            dxn1 = _linear_bwd(dqkv, xn1, qkv_w, dqkv_w, dqkv_b)
            dx = ops.layernorm_bwd(dxn1, x2, n1w, m1, r1, dx1, dn1w, dn1b, defer=batch.ln)
        return (dx.view(dy.shape), None, None, *_finish_param_grads(params, bufs, sunk, batch.deferred), None, None, None, None, None, None, None, None, None)


# ----------------------------------------------------------------------------- class block
class ClassBlockFn(torch.autograd.Function):
    """ClassBlock.forward (models/volo.py:304-308) on SEPARATE class token [B,C] and tokens [B*N,C]:
        cls += proj(class_attn(LN1([cls; tokens])));  cls += fc2(gelu(fc1(LN2(cls))))
    The reference concatenates [cls; tokens] before every class block and slices the result apart again; LayerNorm and the kv
    projection act per row, so both run on the two pieces separately and the attention kernel reads key 0 from the class-token
    piece (split layout of ap_class_attn_*): no concatenation, no slicing, and the residual adds sit in the GEMM epilogues.
    Returns (cls, tokens): the tokens pass through unchanged (models/volo.py:308 returns them inside the concatenation) so that the
    NEXT consumer's gradient arrives here as the second output's and is added inside this block's LayerNorm backward kernel -- with
    the tokens fanned out to two class blocks and the final norm, autograd would add three [B,N,C] gradients in two extra passes."""

    @staticmethod
    def forward(ctx, cls, tok, n1w, n1b, kv_w, kv_b, q_w, q_b, proj_w, proj_b, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b, B, N, heads, eps):
        C = cls.shape[-1]
        c0 = cls.reshape(B, C).contiguous()
        t0 = tok.reshape(B * N, C).contiguous()
        inner = q_w.shape[0]
        scale = (inner // heads) ** -0.5
        nc, mc, rc = ops.layernorm_fwd(c0, n1w, n1b, eps)
        nt, mt, rt = ops.layernorm_fwd(t0, n1w, n1b, eps)
        kv_t = ops.gemm_nt(nt, bank.get(kv_w), bias=kv_b)
        kv_c = ops.gemm_nt(nc, bank.get(kv_w), bias=kv_b)
        q = ops.gemm_nt(nc, bank.get(q_w), bias=q_b)
        o, probs = ops.class_attn_fwd(q, kv_t, B, N + 1, heads, scale, kv_cls=kv_c)
        c1 = ops.gemm_nt(o, bank.get(proj_w), bias=proj_b, residual=c0)
        n2, m2, r2 = ops.layernorm_fwd(c1, n2w, n2b, eps)
        h = _gelu_side_buffer(B, fc1_w.shape[0], cls.device)
        a = ops.gemm_nt(n2, bank.get(fc1_w), bias=fc1_b, gelu=True, preact_out=h, preact_grad=STORE_GELU_GRAD)
        c2 = ops.gemm_nt(a, bank.get(fc2_w), bias=fc2_b, residual=c1)
        ctx.save_for_backward(c0, t0, mc, rc, mt, rt, nc, nt, kv_t, kv_c, q, o, probs, c1, m2, r2, n2, h, a,
                              n1w, n1b, kv_w, kv_b, q_w, q_b, proj_w, proj_b, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b)
        ctx.cfg = (B, N, heads, scale)
        ctx.shapes = (cls.shape, tok.shape)
        ctx.set_materialize_grads(False)          # an unused output hands None to backward, not a zero tensor
        return c2, tok

    @staticmethod
    def backward(ctx, dc2, dtok_pass):
        (c0, t0, mc, rc, mt, rt, nc, nt, kv_t, kv_c, q, o, probs, c1, m2, r2, n2, h, a,
         n1w, n1b, kv_w, kv_b, q_w, q_b, proj_w, proj_b, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b) = ctx.saved_tensors
        B, N, heads, scale = ctx.cfg
        params = (n1w, n1b, kv_w, kv_b, q_w, q_b, proj_w, proj_b, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b)
        bufs, sunk = _param_grad_buffers(params)
        (dn1w, dn1b, dkv_w, dkv_b, dq_w, dq_b, dproj_w, dproj_b, dn2w, dn2b, dfc1_w, dfc1_b, dfc2_w, dfc2_b) = bufs
        g = dc2.contiguous() if dc2 is not None else torch.zeros(c0.shape, dtype=BF16, device=c0.device)
        dpass = dtok_pass.reshape(t0.shape).contiguous() if dtok_pass is not None else None
        with wgrad_batch() as batch:          # (not in the weight-gradient window: 12 small problems for one 8-tile one)
            dh = _linear_bwd(g, a, fc2_w, dfc2_w, dfc2_b, **_gelu_bwd_kw(h))
            dn2 = _linear_bwd(dh, n2, fc1_w, dfc1_w, dfc1_b)
            dc1 = ops.layernorm_bwd(dn2, c1, n2w, m2, r2, g, dn2w, dn2b, defer=batch.ln)
            do = _linear_bwd(dc1, o, proj_w, dproj_w, dproj_b)
            dq, dkv_t, dkv_c = ops.class_attn_bwd(q, kv_t, probs, do, B, N + 1, heads, scale, kv_cls=kv_c)
            dnt = _linear_bwd(dkv_t, nt, kv_w, dkv_w, dkv_b)
            dnc_kv = _linear_bwd(dkv_c, nc, kv_w, dkv_w, dkv_b)                 # second contribution to the same weight gradient
            _linear_bwd(dq, nc, q_w, dq_w, dq_b, need_dx=False)
            wq_t = bank.get_t(q_w)
            dnc = ops.gemm_nt(dq, wq_t, n=wq_t.shape[0], k=wq_t.shape[1], residual=dnc_kv)      # dq Wq + dkv_c Wkv in one epilogue
            dcls = ops.layernorm_bwd(dnc, c0, n1w, mc, rc, dc1, dn1w, dn1b)       # reduced at once: it shares dn1w / dn1b with the deferred
            dtok = ops.layernorm_bwd(dnt, t0, n1w, mt, rt, dpass, dn1w, dn1b, defer=batch.ln)     # token piece (two deferred sums into one vector would race)
        return (dcls.view(ctx.shapes[0]), dtok.view(ctx.shapes[1]), *_finish_param_grads(params, bufs, sunk, batch.deferred), None, None, None, None)


FUSE_POOL_BWD = os.environ.get("AP_FUSE_POOL_BWD", "1") != "0"      # 0: the average pool's backward as a pass of its own (rounds 1 - 4)


# ------------------------------------------------------------------------- outlooker block
class OutlookerBlockFn(torch.autograd.Function):
    """Outlooker.forward (models/volo.py:140-144) with OutlookAttention (models/volo.py:77-103):
    x += proj(outlook(v(LN1 x), attn(pool(LN1 x)))) ; x += fc2(gelu(fc1(LN2 x)))."""

    @staticmethod
    def forward(ctx, x, n1w, n1b, v_w, v_b, attn_w, attn_b, proj_w, proj_b, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b,
                heads, eps):
        B, H, W, C = x.shape
        T = B * H * W
        x2 = x.reshape(T, C).contiguous()
        scale = (C // heads) ** -0.5
        xn1, m1, r1 = ops.layernorm_fwd(x2, n1w, n1b, eps)
        v = ops.gemm_nt(xn1, bank.get(v_w), bias=v_b)
        pooled = ops.avgpool2_fwd(xn1.view(B, H, W, C))
        pooled2 = pooled.view(-1, C)
        logits = ops.gemm_nt(pooled2, bank.get(attn_w), bias=attn_b)                 # [B*h*w, ld(heads*81)]
        yo = ops.outlook_fwd(v.view(B, H, W, C), logits, heads, scale)
        x1 = ops.gemm_nt(yo.view(T, C), bank.get(proj_w), bias=proj_b, residual=x2)
        xn2, m2, r2 = ops.layernorm_fwd(x1, n2w, n2b, eps)
        h = _gelu_side_buffer(T, fc1_w.shape[0], x.device)
        a = ops.gemm_nt(xn2, bank.get(fc1_w), bias=fc1_b, gelu=True, preact_out=h, preact_grad=STORE_GELU_GRAD)
        y = ops.gemm_nt(a, bank.get(fc2_w), bias=fc2_b, residual=x1)
        ctx.save_for_backward(x2, m1, r1, xn1, v, pooled2, logits, yo, x1, m2, r2, xn2, h, a,
                              n1w, n1b, v_w, v_b, attn_w, attn_b, proj_w, proj_b, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b)
        ctx.cfg = (B, H, W, C, heads, scale)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        (x2, m1, r1, xn1, v, pooled2, logits, yo, x1, m2, r2, xn2, h, a,
         n1w, n1b, v_w, v_b, attn_w, attn_b, proj_w, proj_b, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b) = ctx.saved_tensors
        B, H, W, C, heads, scale = ctx.cfg
        T = B * H * W
        params = (n1w, n1b, v_w, v_b, attn_w, attn_b, proj_w, proj_b, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b)
        bufs, sunk = _param_grad_buffers(params)
        (dn1w, dn1b, dv_w, dv_b, dattn_w, dattn_b, dproj_w, dproj_b, dn2w, dn2b, dfc1_w, dfc1_b, dfc2_w, dfc2_b) = bufs
        dy2 = dy.reshape(T, C).contiguous()
        with wgrad_batch(sunk, params) as batch:         # the five weight gradients launch together: on exit, or with the window's
            dh = _linear_bwd(dy2, a, fc2_w, dfc2_w, dfc2_b, **_gelu_bwd_kw(h))
            dxn2 = _linear_bwd(dh, xn2, fc1_w, dfc1_w, dfc1_b)
            dx1 = ops.layernorm_bwd(dxn2, x1, n2w, m2, r2, dy2, dn2w, dn2b, defer=batch.ln)
            dyo = _linear_bwd(dx1, yo.view(T, C), proj_w, dproj_w, dproj_b)
            dv, dlogits = ops.outlook_bwd(v.view(B, H, W, C), logits, dyo.view(B, H, W, C), heads, scale)
            dpooled = _linear_bwd(dlogits, pooled2, attn_w, dattn_w, dattn_b, n=attn_w.shape[0])
            dxn1 = _linear_bwd(dv.view(T, C), xn1, v_w, dv_w, dv_b)
            # the average pool's backward (dxn1 += dpooled / count) rides in the LayerNorm backward kernel: one pass over the gradient less
            dx = ops.layernorm_bwd(dxn1, x2, n1w, m1, r1, dx1, dn1w, dn1b, defer=batch.ln, pool=(dpooled.view(B, (H + 1) // 2, (W + 1) // 2, C), (B, H, W))) if FUSE_POOL_BWD else None
            if dx is None:
                ops.avgpool2_bwd_acc(dpooled.view(B, (H + 1) // 2, (W + 1) // 2, C), dxn1.view(B, H, W, C))
                dx = ops.layernorm_bwd(dxn1, x2, n1w, m1, r1, dx1, dn1w, dn1b, defer=batch.ln)
        return (dx.view(dy.shape), *_finish_param_grads(params, bufs, sunk, batch.deferred), None, None)


# --------------------------------------------------------------------- fine-grained pieces
class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        xc = x.contiguous()
        y, m, r = ops.layernorm_fwd(xc, w, b, eps)
        ctx.save_for_backward(xc, w, b, m, r)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, w, b, m, r = ctx.saved_tensors
        bufs, sunk = _param_grad_buffers((w, b))      # with a gradient sink: accumulate straight into param.grad
        dx = ops.layernorm_bwd(dy.contiguous(), xc, w, m, r, None, bufs[0], bufs[1])
        dw, db = _finish_param_grads((w, b), bufs, sunk)
        return dx, dw, db, None


class LinearFn(torch.autograd.Function):
    """y = [gelu](x W^T + b) on bf16 activations; W may be conv-shaped ([N, ...] flattened)."""

    @staticmethod
    def forward(ctx, x, w, b, gelu):
        K = x.shape[-1]
        x2 = x.reshape(-1, K).contiguous()
        N = w.shape[0]
        h = torch.empty((x2.shape[0], ops.round_up(N, 8)), dtype=BF16, device=x.device) if gelu else None
        y = ops.gemm_nt(x2, bank.get(w), n=N, k=K, bias=b, gelu=gelu, preact_out=h, preact_grad=bool(gelu and STORE_GELU_GRAD))     # (bf16 derivative: multiplied in torch below)
        ctx.save_for_backward(x2, w, b, h)
        ctx.lead = x.shape[:-1]
        ctx.gelu = gelu
        out = y.view(*x.shape[:-1], y.shape[-1])
        return out if y.shape[-1] == N else out[..., :N]

    @staticmethod
    def backward(ctx, dy):
        x2, w, b, h = ctx.saved_tensors
        N = w.shape[0]
        ld = ops.round_up(N, 8)
        g = dy.reshape(-1, N)
        if ld != N:                                   # test-size heads: re-pad the gradient rows
            gp = torch.zeros((g.shape[0], ld), dtype=BF16, device=g.device)
            gp[:, :N] = g
            g = gp
        g = g.contiguous()
        if ctx.gelu:                                  # dL/dh = dL/da * gelu'(h); the forward stored gelu'(h) (preact_grad)
            if STORE_GELU_GRAD:
                g = (g.float() * h.float()).to(BF16)
            else:
                hf = h.float()
                g = (g.float() * (0.5 * (1.0 + torch.erf(hf * 0.7071067811865476)) + hf * torch.exp(-0.5 * hf * hf) * 0.3989422804014327)).to(BF16)
        bufs, sunk = _param_grad_buffers((w, b))      # with a gradient sink (and w, b real parameters): param.grad itself
        dx = _linear_bwd(g, x2, w, bufs[0], bufs[1], n=N, need_dx=ctx.needs_input_grad[0])
        dw, db = _finish_param_grads((w, b), bufs, sunk)
        if dx is not None:
            dx = dx[:, :x2.shape[1]] if dx.shape[1] != x2.shape[1] else dx
            dx = dx.reshape(*ctx.lead, x2.shape[1])
        return dx, dw, db, None


class ClassAttnFn(torch.autograd.Function):
    """ClassAttention core (models/volo.py:264-274): q [B,C] (un-scaled), kv [B,N,2C]."""

    @staticmethod
    def forward(ctx, q, kv, heads):
        B, N, C2 = kv.shape
        qc, kvc = q.contiguous(), kv.contiguous()
        scale = (C2 // 2 // heads) ** -0.5
        o, probs = ops.class_attn_fwd(qc, kvc.view(B * N, C2), B, N, heads, scale)
        ctx.save_for_backward(qc, kvc, probs)
        ctx.cfg = (B, N, heads, scale)
        return o

    @staticmethod
    def backward(ctx, do):
        qc, kvc, probs = ctx.saved_tensors
        B, N, heads, scale = ctx.cfg
        dq, dkv = ops.class_attn_bwd(qc, kvc.view(B * N, -1), probs, do.contiguous(), B, N, heads, scale)
        return dq, dkv.view(kvc.shape), None


class ClsExpandFn(torch.autograd.Function):
    """cls_token.expand(B, -1, -1) (models/volo.py:637) as a bf16 [B, C] view of one cast row; the gradient is the column sum over
    the batch, accumulated in fp32 by ap_colsum_acc straight into the parameter's gradient (torch: cast, expand, a bf16 reduction
    kernel that takes 14 us for 128 x 384, cast back, add)."""

    @staticmethod
    def forward(ctx, cls_token, B):
        ctx.save_for_backward(cls_token)
        return ops.cast_bf16(cls_token.detach().reshape(1, -1)).expand(B, -1)

    @staticmethod
    def backward(ctx, g):
        (cls_token,) = ctx.saved_tensors
        bufs, sunk = _param_grad_buffers((cls_token,))
        ops.colsum_acc(g.contiguous(), bufs[0].view(-1))
        (dc,) = _finish_param_grads((cls_token,), bufs, sunk)
        return dc, None


class MixSwapFn(torch.autograd.Function):
    """y[b, r0:r1, c0:c1] = x[B-1-b, r0:r1, c0:c1] (models/volo.py:654-658, 685-689); the
    backward is the same permutation."""

    @staticmethod
    def forward(ctx, x, r0, r1, c0, c1):
        ctx.box = (r0, r1, c0, c1)
        return ops.mix_token_swap(x.contiguous(), r0, r1, c0, c1)

    @staticmethod
    def backward(ctx, dy):
        return ops.mix_token_swap(dy.contiguous(), *ctx.box), None, None, None, None


class MixSwapDevFn(torch.autograd.Function):
    """MixSwapFn with the step's box in device memory (graph.StepScalars: int32 {bbx1, bbx2, bby1, bby2} on the token-label grid, times
    `scale`): the launch arguments do not change from step to step, so the step can be replayed from a HIP graph"""

    @staticmethod
    def forward(ctx, x, scalars, scale):
        ctx.scalars, ctx.scale = scalars, scale
        return ops.mix_token_swap_dev(x.contiguous(), scalars.box_ptr, scale)

    @staticmethod
    def backward(ctx, dy):
        return ops.mix_token_swap_dev(dy.contiguous(), ctx.scalars.box_ptr, ctx.scale), None, None


_BICUBIC_TAPS = {}


def bicubic_tap_matrix(n_in, n_out, scale_factor):
    """dense [n_out, n_in] fp32 matrix of torch.nn.functional.interpolate(..., scale_factor=s, mode="bicubic", align_corners=False) along one
    axis, in the arithmetic of aten's upsample_bicubic2d: source coordinate (o + 0.5) / s - 0.5 (the GIVEN scale factor, not n_out / n_in:
    the reference passes (h0 + 0.1) / h, models/volo.py:590-594), cubic convolution with A = -0.75, source indices clamped to the grid."""
    import numpy as np
    f32 = np.float32
    A = f32(-0.75)
    r = f32(1.0 / float(scale_factor))
    W = np.zeros((n_out, n_in), dtype=np.float32)

    def conv1(x):            # |x| <= 1
        return ((A + f32(2)) * x - (A + f32(3))) * x * x + f32(1)

    def conv2(x):            # 1 < |x| < 2
        return ((A * x - f32(5) * A) * x + f32(8) * A) * x - f32(4) * A
    for o in range(n_out):
        real = r * (f32(o) + f32(0.5)) - f32(0.5)
        i0 = int(np.floor(real))
        t = f32(real - f32(i0))
        coeff = (conv2(t + f32(1)), conv1(t), conv1(f32(1) - t), conv2(f32(2) - t))
        for k in range(4):
            W[o, min(max(i0 - 1 + k, 0), n_in - 1)] += coeff[k]
    return W


def _pos_taps(h, w, h0, w0, device):
    key = (h, w, h0, w0, str(device))
    ent = _BICUBIC_TAPS.get(key)
    if ent is None:
        if len(_BICUBIC_TAPS) > 64:                  # (a search over many resolutions: four small matrices per entry)
            _BICUBIC_TAPS.clear()
        wy = torch.from_numpy(bicubic_tap_matrix(h, h0, (h0 + 0.1) / h))
        wx = torch.from_numpy(bicubic_tap_matrix(w, w0, (w0 + 0.1) / w))
        ent = tuple(t.to(device).contiguous() for t in (wy, wx, wy.t().contiguous(), wx.t().contiguous()))
        _BICUBIC_TAPS[key] = ent
    return ent


class PosEmbedInterpFn(torch.autograd.Function):
    """VOLO.interpolate_pos_encoding (models/volo.py:580-596): pos_embed [1, h, w, C] -> [1, h0, w0, C] by bicubic interpolation, as one
    small HIP kernel per direction on the embedding's own NHWC layout (ops.resample_grid with the tap matrices of the torch call; the
    backward multiplies by their transposes and adds straight into the parameter's gradient)."""

    @staticmethod
    def forward(ctx, pos, h0, w0):
        _, h, w, C = pos.shape
        wy, wx, wyt, wxt = _pos_taps(h, w, h0, w0, pos.device)
        ctx.save_for_backward(pos, wyt, wxt)
        return ops.resample_grid(pos.detach().reshape(h, w, C).contiguous(), wy, wx).view(1, h0, w0, C)

    @staticmethod
    def backward(ctx, g):
        pos, wyt, wxt = ctx.saved_tensors
        _, h0, w0, C = g.shape
        bufs, sunk = _param_grad_buffers((pos,))
        ops.resample_grid(g.reshape(h0, w0, C).contiguous().float(), wyt, wxt, out=bufs[0].view(pos.shape[1], pos.shape[2], C), accumulate=True)
        (dp,) = _finish_param_grads((pos,), bufs, sunk)
        return dp, None, None


class AddPosFn(torch.autograd.Function):
    """x [B,h,w,C] bf16 + pos [1,h,w,C] fp32 (models/volo.py:627-629)."""

    @staticmethod
    def forward(ctx, x, pos):
        ctx.pos_shape = pos.shape
        ctx.B = x.shape[0]
        return ops.add_bcast(x.contiguous(), ops.cast_bf16(pos.contiguous()))

    @staticmethod
    def backward(ctx, dy):
        dpos = torch.zeros(ctx.pos_shape, dtype=torch.float32, device=dy.device)
        dyc = dy.contiguous()
        ops.sum_reps_acc(dyc, dpos, ctx.B)
        return dyc, dpos


class SoftTargetCEFn(torch.autograd.Function):
    """mean over rows of -sum_c t*log_softmax(x) (loss/cross_entropy.py:35-36) with the
    gradient produced in the same pass.  target is addressed through explicit strides so the
    class-major token-label tensor [B,C,2+N] is consumed in place (loss/cross_entropy.py:147-148)."""

    @staticmethod
    def forward(ctx, logits, target, t_sb, t_sc, t_sn, rows_per_batch):
        M, C = logits.shape
        ld = ops.round_up(C, 8)
        if ld != C or not logits.is_contiguous():
            xp = torch.zeros((M, ld), dtype=BF16, device=logits.device)
            xp[:, :C] = logits
        else:
            xp = logits
        row_loss, dl = ops.soft_ce_fwd_bwd(xp, C, target, t_sb, t_sc, t_sn, rows_per_batch, 1.0 / M)
        ctx.save_for_backward(dl)
        ctx.C = C
        return row_loss.mean()

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        M = dl.shape[0]
        out = ops.row_scale(dl, g.reshape(1).float().contiguous(), M)
        return (out if out.shape[1] == ctx.C else out[:, :ctx.C]), None, None, None, None, None


class TokenLabelCEFn(torch.autograd.Function):
    """TokenLabelCrossEntropy.forward with a token-label target [B,C,2+N] (loss/cross_entropy.py:136-156) in three launches:
    dense CE of the aux logits against target[:,:,2:], CE of the class logits against the (mix-token blended) target[:,:,1],
    and cls_weight * mean + dense_weight * mean.  The reference's version of this is ~20 small tensor ops on the loss path; both
    gradients come out of the forward kernels and only take the incoming scalar in backward."""

    @staticmethod
    def forward(ctx, x_cls, x_aux, target, lam, cls_weight, dense_weight):
        B, N, C = x_aux.shape
        ld = ops.round_up(C, 8)

        def padded(x2d):
            if ld == C and x2d.is_contiguous():
                return x2d
            xp = torch.zeros((x2d.shape[0], ld), dtype=BF16, device=x2d.device)
            xp[:, :C] = x2d
            return xp
        aux2d = padded(x_aux.reshape(B * N, C))
        cls2d = padded(x_cls.reshape(B, C))
        sb, sc, sn = target.stride()
        rl_aux, d_aux = ops.soft_ce_fwd_bwd(aux2d, C, target[:, :, 2:], sb, sc, sn, N, dense_weight / (B * N))
        if hasattr(lam, "lam_ptr"):           # graph.StepScalars: lam lives in device memory (lam = 1 blends nothing: lam t + 0 t' = t)
            rl_cls, d_cls = ops.soft_ce_fwd_bwd(cls2d, C, target[:, :, 1], sb, sc, 0, 1, cls_weight / B, mix_lam=1.0, mix_batches=B, mix_lam_ptr=lam.lam_ptr)
        else:
            mixed = lam < 1
            rl_cls, d_cls = ops.soft_ce_fwd_bwd(cls2d, C, target[:, :, 1], sb, sc, 0, 1, cls_weight / B,
                                                mix_lam=lam if mixed else 1.0, mix_batches=B if mixed else 0)
        ctx.save_for_backward(d_cls, d_aux)
        ctx.dims = (B, N, C)
        return ops.loss_combine(rl_cls, cls_weight / B, rl_aux, dense_weight / (B * N))

    @staticmethod
    def backward(ctx, g):
        d_cls, d_aux = ctx.saved_tensors
        B, N, C = ctx.dims
        gs = g.reshape(1).float().contiguous()
        dc = ops.row_scale(d_cls, gs, d_cls.shape[0])
        da = ops.row_scale(d_aux, gs, d_aux.shape[0])
        dc = dc if dc.shape[1] == C else dc[:, :C]
        da = da if da.shape[1] == C else da[:, :C]
        return dc, da.reshape(B, N, C), None, None, None, None


class SparseTokenLabelCEFn(torch.autograd.Function):
    """TokenLabelCrossEntropy on the token-label target in its SOURCE form -- top-K (class, score) pairs per slot [B, 2+N, K] and a
    label-smoothing strength -- instead of the dense [B,C,2+N] tensor the reference builds from them on the GPU every step
    (main_prog.py:994-1004; loss/cross_entropy.py:136-156 for the loss itself).  Same three launches as TokenLabelCEFn; the
    mix-token class target lam * t[b] + (1 - lam) * t[B-1-b] is 2K pairs per image, gathered by the kernel from the two label maps."""

    @staticmethod
    def forward(ctx, x_cls, x_aux, idx, val, smoothing, lam, cls_weight, dense_weight):
        B, N, C = x_aux.shape
        K = idx.shape[-1]
        ld = ops.round_up(C, 8)

        def padded(x2d):
            if ld == C and x2d.is_contiguous():
                return x2d
            xp = torch.zeros((x2d.shape[0], ld), dtype=BF16, device=x2d.device)
            xp[:, :C] = x2d
            return xp
        idx, val = idx.contiguous(), val.contiguous()
        rl_aux, d_aux = ops.soft_ce_sparse_fwd_bwd(padded(x_aux.reshape(B * N, C)), C, idx[:, 2:], val[:, 2:], (2 + N) * K, K, N, smoothing,
                                                   dense_weight / (B * N))
        # the class row: slot 1 of every image, in place (stride (2 + N) K); the mix-token target lam * t[b] + (1 - lam) * t[B-1-b] is
        # formed by the kernel from the two images' pairs
        if hasattr(lam, "lam_ptr"):           # graph.StepScalars
            rl_cls, d_cls = ops.soft_ce_sparse_fwd_bwd(padded(x_cls.reshape(B, C)), C, idx[:, 1], val[:, 1], (2 + N) * K, 0, 1, smoothing, cls_weight / B,
                                                       mix_lam=1.0, mix_batches=B, mix_lam_ptr=lam.lam_ptr)
        else:
            rl_cls, d_cls = ops.soft_ce_sparse_fwd_bwd(padded(x_cls.reshape(B, C)), C, idx[:, 1], val[:, 1], (2 + N) * K, 0, 1, smoothing, cls_weight / B,
                                                       mix_lam=lam, mix_batches=B if lam < 1 else 0)
        ctx.save_for_backward(d_cls, d_aux)
        ctx.dims = (B, N, C)
        return ops.loss_combine(rl_cls, cls_weight / B, rl_aux, dense_weight / (B * N))

    @staticmethod
    def backward(ctx, g):
        d_cls, d_aux = ctx.saved_tensors
        B, N, C = ctx.dims
        gs = g.reshape(1).float().contiguous()
        dc = ops.row_scale(d_cls, gs, d_cls.shape[0])
        da = ops.row_scale(d_aux, gs, d_aux.shape[0])
        dc = dc if dc.shape[1] == C else dc[:, :C]
        da = da if da.shape[1] == C else da[:, :C]
        return dc, da.reshape(B, N, C), None, None, None, None, None, None


class OutlookCoreFn(torch.autograd.Function):
    """unfold -> softmax -> attn@v -> fold (models/volo.py:83-98) on v [B,H,W,C], logits [B*h*w, ld]."""

    @staticmethod
    def forward(ctx, v, logits, heads):
        vc, lc = v.contiguous(), logits.contiguous()
        scale = (v.shape[-1] // heads) ** -0.5
        ctx.save_for_backward(vc, lc)
        ctx.cfg = (heads, scale)
        return ops.outlook_fwd(vc, lc, heads, scale)

    @staticmethod
    def backward(ctx, dy):
        vc, lc = ctx.saved_tensors
        heads, scale = ctx.cfg
        dv, dl = ops.outlook_bwd(vc, lc, dy.contiguous(), heads, scale)
        return dv, dl, None


class AvgPool2Fn(torch.autograd.Function):
    """AvgPool2d(2,2,ceil_mode=True) on NHWC tokens (models/volo.py:75,87)."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        return ops.avgpool2_fwd(x.contiguous())

    @staticmethod
    def backward(ctx, dp):
        dx = torch.zeros(ctx.shape, dtype=BF16, device=dp.device)
        return ops.avgpool2_bwd_acc(dp.contiguous(), dx)


class MhsaFn(torch.autograd.Function):
    """softmax(q k^T scale) v on packed qkv [B*N, 3C] (models/volo.py:188-197)."""

    @staticmethod
    def forward(ctx, qkv, B, N, heads):
        qc = qkv.contiguous()
        scale = (qc.shape[-1] // 3 // heads) ** -0.5
        o, lse = ops.mhsa_fwd(qc, B, N, heads, scale)
        ctx.save_for_backward(qc, o, lse)
        ctx.cfg = (B, N, heads, scale)
        return o

    @staticmethod
    def backward(ctx, do):
        qc, o, lse = ctx.saved_tensors
        B, N, heads, scale = ctx.cfg
        return ops.mhsa_bwd(qc, o, do.contiguous(), lse, B, N, heads, scale), None, None, None


class BNReLUFn(torch.autograd.Function):
    """BatchNorm2d (batch statistics in training, running statistics in eval) + ReLU on an NHWC bf16 tensor
    (the stem's conv -> BN -> ReLU triples, models/volo.py:355-367)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps):
        xc = x.contiguous()
        y, mean, rstd = ops.bn_relu_fwd(xc, weight, bias, running_mean, running_var, training, momentum, eps)
        ctx.save_for_backward(xc, weight, bias, mean, rstd)
        ctx.training = training
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, weight, bias, mean, rstd = ctx.saved_tensors
        if not ctx.training:
            raise AutoProgHipError("BNReLUFn backward is implemented for training mode (batch statistics) only")
        bufs, sunk = _param_grad_buffers((weight, bias))      # with a gradient sink: accumulate straight into param.grad
        dx = ops.bn_relu_bwd(dy.contiguous(), xc, weight, bias, mean, rstd, bufs[0], bufs[1])
        dg, db = _finish_param_grads((weight, bias), bufs, sunk)
        return dx, dg, db, None, None, None, None, None


class PatchConvFn(torch.autograd.Function):
    """k x k / stride k convolution on an NHWC bf16 feature map as ONE GEMM per direction with patch addressing (PatchEmbed.proj and
    Downsample, models/volo.py:368-372,383-396): forward, input gradient and weight gradient read / write the feature map in place --
    no gathered [tokens, k*k*C] matrix and no scatter of its gradient.  x [B,H,W,C] (H, W multiples of k), weight [N,C,k,k] fp32,
    bias [N] or None -> [B, H/k, W/k, N] bf16."""

    @staticmethod
    def forward(ctx, x, weight, bias, k, bn_mean=None, bn_rstd=None, bn_gamma=None, bn_beta=None):
        """bn_* given (round 5): x is the PRE-BatchNorm output z of the layer in front (64 channels) and this convolution acts on
        relu(bn(z)) -- batch statistics bn_mean / bn_rstd, affine parameters bn_gamma / bn_beta -- applied while the kernels stage z (forward and
        weight gradient); the backward runs the BatchNorm + ReLU backward on its input gradient and hands out dL/dz and the gradients of
        gamma / beta: the activation between the stem's last BatchNorm and PatchEmbed.proj (models/volo.py:364-372) never exists."""
        xc = x.contiguous()
        B, H, W, C = xc.shape
        N = weight.shape[0]
        wmat = torch.empty((N, k, k, C), dtype=BF16, device=weight.device)
        wmat.copy_(weight.detach().permute(0, 2, 3, 1))                                      # re-layout + cast in ONE copy kernel
        wmat = wmat.view(N, k * k * C)                                                       # columns ordered (dy, dx, c)
        bn = (bn_mean, bn_rstd, bn_gamma, bn_beta) if bn_mean is not None else None
        y = ops.gemm_nt_patch_fwd(xc, wmat, bias, k, bn_in=bn)
        ctx.save_for_backward(xc, weight, bias, wmat, bn_mean, bn_rstd, bn_gamma, bn_beta)
        ctx.k = k
        out = y.view(B, H // k, W // k, y.shape[-1])
        return out if y.shape[-1] == N else out[..., :N]

    @staticmethod
    def backward(ctx, dy):
        xc, weight, bias, wmat, bn_mean, bn_rstd, bn_gamma, bn_beta = ctx.saved_tensors
        bn = (bn_mean, bn_rstd, bn_gamma, bn_beta) if bn_mean is not None else None
        k = ctx.k
        B, H, W, C = xc.shape
        N, K = wmat.shape
        ld = ops.round_up(N, 8)
        g = dy.reshape(-1, N)
        if ld != N:
            gp = torch.zeros((g.shape[0], ld), dtype=BF16, device=g.device)
            gp[:, :N] = g
            g = gp
        g = g.contiguous()
        params = (weight, bias) if bn is None else (weight, bias, bn_gamma, bn_beta)
        bufs, sunk = _param_grad_buffers(params)
        dx = None
        b_rows, b_bn = xc, bn             # the weight gradient's B rows: the map the forward read (+ its transform)
        if ctx.needs_input_grad[0]:
            wt = torch.zeros((K, ld), dtype=BF16, device=g.device) if ld != N else torch.empty((K, N), dtype=BF16, device=g.device)
            wt[:, :N] = wmat.t()
            dx = ops.gemm_nt_patch_dgrad(g, wt, (B, H, W, C), k)
            if bn is not None:            # dL/d(relu(bn(z))) -> dL/dz and the BatchNorm's parameter gradients
                if BN_PROJ_ACT_IN_BWD:
                    # ... and relu(bn(z)) itself, written by the dx pass (it holds z anyway) for the weight gradient below: transforming every
                    # chunk that launch stages -- twice, there are two row tiles -- doubled its time (70 -> 141 us at B = 128, 224 px)
                    dx, b_rows = ops.bn_relu_bwd(dx, xc, bn_gamma, bn_beta, bn_mean, bn_rstd, bufs[2], bufs[3], act_out=True)
                    b_bn = None
                else:
                    dx = ops.bn_relu_bwd(dx, xc, bn_gamma, bn_beta, bn_mean, bn_rstd, bufs[2], bufs[3])
        elif bn is not None:
            raise AutoProgHipError("PatchConvFn with a BatchNorm input: the BatchNorm's parameter gradients need the input gradient")
        dwmat = torch.zeros((N, K), dtype=torch.float32, device=g.device)
        _launch_wgrads([(g, b_rows, dwmat, N, K, bufs[1], None, 1.0, 1.0, ops.patch_map(H, W, C, k), b_bn)])
        join_wgrad_stream()
        bufs[0].add_(dwmat.view(N, k, k, C).permute(0, 3, 1, 2))                              # back to OIHW
        gr = _finish_param_grads(params, bufs, sunk)
        return (dx, gr[0], gr[1], None, None, None) + ((gr[2], gr[3]) if bn is not None else (None, None))


def patch_conv_ok(x, weight, k):
    """the shapes PatchConvFn's kernels take: NHWC bf16, H and W multiples of k, k*C a multiple of 64"""
    return (x.dim() == 4 and x.dtype == BF16 and x.is_cuda and x.shape[1] % k == 0 and x.shape[2] % k == 0 and (k * x.shape[3]) % 64 == 0
            and weight.shape[2] == k and weight.shape[3] == k)


class Conv7BNReLUFn(torch.autograd.Function):
    """conv7x7 / stride 2 (3 -> 64, no bias) -> BatchNorm2d -> ReLU: the first triple of the VOLO stem (models/volo.py:355-358) on the
    space-to-depth image xs [B,H,W,16] (ops.resize_bilinear_s2d16) -- csrc/conv7.hip; the image needs no gradient."""

    @staticmethod
    def forward(ctx, xs, conv_w, weight, bias, running_mean, running_var, training, momentum, eps):
        wp = ops.conv7_pack(conv_w.detach().float().contiguous())
        if training:
            z, partials = ops.conv7_s2d(xs, wp, True)
            y, mean, rstd = ops.bn_relu_fwd(z, weight, bias, running_mean, running_var, True, momentum, eps, partials=partials)
        else:
            z = ops.conv7_s2d(xs, wp)
            y, mean, rstd = ops.bn_relu_fwd(z, weight, bias, running_mean, running_var, False, momentum, eps)
        ctx.save_for_backward(xs, z, conv_w, weight, bias, mean, rstd)
        ctx.training = training
        return y

    @staticmethod
    def backward(ctx, dy):
        xs, z, conv_w, weight, bias, mean, rstd = ctx.saved_tensors
        if not ctx.training:
            raise AutoProgHipError("Conv7BNReLUFn backward is implemented for training mode (batch statistics) only")
        params = (conv_w, weight, bias)
        bufs, sunk = _param_grad_buffers(params)
        dz = ops.bn_relu_bwd(dy.contiguous(), z, weight, bias, mean, rstd, bufs[1], bufs[2])
        ops.conv7_s2d_wgrad(xs, dz, bufs[0])
        dw, dg, db = _finish_param_grads(params, bufs, sunk)
        return None, dw, dg, db, None, None, None, None, None


class Conv3x3BNReLUFn(torch.autograd.Function):
    """conv3x3(64 -> 64, stride 1, pad 1, no bias) -> BatchNorm2d -> ReLU on an NHWC bf16 tensor: the second and third triple of the
    VOLO stem (models/volo.py:359-366) on the HIP convolution kernels of csrc/conv.hip.  The convolution's epilogue also produces the
    partial batch statistics, so BatchNorm reads the convolution output once (apply) instead of twice (statistics + apply)."""

    @staticmethod
    def forward(ctx, x, conv_w, weight, bias, running_mean, running_var, training, momentum, eps):
        xc = x.contiguous()
        wf, wb = ops.conv3x3_pack(conv_w.detach().float().contiguous())
        if training:
            z, partials = ops.conv3x3_c64(xc, wf, True)
            y, mean, rstd = ops.bn_relu_fwd(z, weight, bias, running_mean, running_var, True, momentum, eps, partials=partials)
        else:
            z = ops.conv3x3_c64(xc, wf)
            y, mean, rstd = ops.bn_relu_fwd(z, weight, bias, running_mean, running_var, False, momentum, eps)
        ctx.save_for_backward(xc, z, wb, conv_w, weight, bias, mean, rstd)
        ctx.training = training
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, z, wb, conv_w, weight, bias, mean, rstd = ctx.saved_tensors
        if not ctx.training:
            raise AutoProgHipError("Conv3x3BNReLUFn backward is implemented for training mode (batch statistics) only")
        params = (conv_w, weight, bias)
        bufs, sunk = _param_grad_buffers(params)
        dz = ops.bn_relu_bwd(dy.contiguous(), z, weight, bias, mean, rstd, bufs[1], bufs[2])
        dx = ops.conv3x3_c64(dz, wb) if ctx.needs_input_grad[0] else None
        ops.conv3x3_c64_wgrad(xc, dz, bufs[0])
        dw, dg, db = _finish_param_grads(params, bufs, sunk)
        return dx, dw, dg, db, None, None, None, None, None


class Stem64Fn(torch.autograd.Function):
    """The whole 64-wide stem in front of the patch projection (models/volo.py:355-366): conv7x7/s2 -> BN -> ReLU -> conv3x3 -> BN -> ReLU
    -> conv3x3 -> BN -> ReLU on the space-to-depth image, as ONE autograd node so that the two activations between the convolutions
    never exist: a 3x3 convolution (forward and weight gradient) reads the PRE-BatchNorm output of the layer before it and applies
    relu(bn(.)) while it stages its input (csrc/conv.hip PRE_BN) -- the arithmetic of the stand-alone BatchNorm + ReLU pass, rounded
    to bf16 at the same place, so the result is bit-identical to the chain Conv7BNReLUFn -> Conv3x3BNReLUFn -> Conv3x3BNReLUFn.
    Saves two passes of 411 MB per step at B = 128, 224 px.  AP_STEM_FUSE_BN=0 keeps the chain."""

    @staticmethod
    def forward(ctx, xs, w7, g1, b1, rm1, rv1, w2, g2, b2, rm2, rv2, w3, g3, b3, rm3, rv3, training, momentum, eps, apply_last=True):
        """apply_last = False (round 5): -> (z3, mean3, rstd3), the PRE-BatchNorm output of the last convolution and its batch (or running)
        statistics; the consumer -- PatchConvFn with bn_* -- applies relu(bn3(.)) while it stages z3 and returns dL/dz3 with the BatchNorm's
        backward already applied; g3 / b3 then get their gradients there, not here"""
        wp7 = ops.conv7_pack(w7.detach().float().contiguous())
        wf2, wb2 = ops.conv3x3_pack(w2.detach().float().contiguous())
        wf3, wb3 = ops.conv3x3_pack(w3.detach().float().contiguous())
        mom1, mom2, mom3 = momentum
        eps1, eps2, eps3 = eps
        if training:
            z1, p1 = ops.conv7_s2d(xs, wp7, True)
            _, mean1, rstd1 = ops.bn_relu_fwd(z1, g1, b1, rm1, rv1, True, mom1, eps1, partials=p1, apply=False)
            z2, p2 = ops.conv3x3_c64(z1, wf2, True, bn_in=(mean1, rstd1, g1, b1))
            _, mean2, rstd2 = ops.bn_relu_fwd(z2, g2, b2, rm2, rv2, True, mom2, eps2, partials=p2, apply=False)
            z3, p3 = ops.conv3x3_c64(z2, wf3, True, bn_in=(mean2, rstd2, g2, b2))
            y, mean3, rstd3 = ops.bn_relu_fwd(z3, g3, b3, rm3, rv3, True, mom3, eps3, partials=p3, apply=apply_last)
        else:
            z1 = ops.conv7_s2d(xs, wp7)
            _, mean1, rstd1 = ops.bn_relu_fwd(z1, g1, b1, rm1, rv1, False, mom1, eps1, apply=False)
            z2 = ops.conv3x3_c64(z1, wf2, bn_in=(mean1, rstd1, g1, b1))
            _, mean2, rstd2 = ops.bn_relu_fwd(z2, g2, b2, rm2, rv2, False, mom2, eps2, apply=False)
            z3 = ops.conv3x3_c64(z2, wf3, bn_in=(mean2, rstd2, g2, b2))
            y, mean3, rstd3 = ops.bn_relu_fwd(z3, g3, b3, rm3, rv3, False, mom3, eps3, apply=apply_last)
        ctx.save_for_backward(xs, z1, z2, z3, wb2, wb3, w7, g1, b1, w2, g2, b2, w3, g3, b3, mean1, rstd1, mean2, rstd2, mean3, rstd3)
        ctx.training = training
        ctx.apply_last = apply_last
        if not apply_last:
            ctx.mark_non_differentiable(mean3, rstd3)
            return z3, mean3, rstd3
        return y

    @staticmethod
    def backward(ctx, dy, *_unused):
        (xs, z1, z2, z3, wb2, wb3, w7, g1, b1, w2, g2, b2, w3, g3, b3, mean1, rstd1, mean2, rstd2, mean3, rstd3) = ctx.saved_tensors
        if not ctx.training:
            raise AutoProgHipError("Stem64Fn backward is implemented for training mode (batch statistics) only")
        last = ctx.apply_last
        params = (w7, g1, b1, w2, g2, b2, w3, g3, b3) if last else (w7, g1, b1, w2, g2, b2, w3)
        bufs, sunk = _param_grad_buffers(params)
        if last:
            dw7, dg1, db1, dw2, dg2, db2, dw3, dg3, db3 = bufs
            dz3 = ops.bn_relu_bwd(dy.contiguous(), z3, g3, b3, mean3, rstd3, dg3, db3)
        else:                              # dy IS dL/dz3: the consumer ran the last BatchNorm's backward (PatchConvFn with bn_*)
            dw7, dg1, db1, dw2, dg2, db2, dw3 = bufs
            dz3 = dy.contiguous()
        if STEM_FUSE_BN_BWD_STATS:         # the first pass of each BatchNorm's backward rides in the convolution that produces its dy
            da2, part2 = ops.conv3x3_c64_bwd_stats(dz3, wb3, z2, (mean2, rstd2, g2, b2))
            ops.conv3x3_c64_wgrad(z2, dz3, dw3, bn_in=(mean2, rstd2, g2, b2))
            dz2 = ops.bn_relu_bwd_partials(da2, z2, g2, b2, mean2, rstd2, part2, dg2, db2)
            da1, part1 = ops.conv3x3_c64_bwd_stats(dz2, wb2, z1, (mean1, rstd1, g1, b1))
            ops.conv3x3_c64_wgrad(z1, dz2, dw2, bn_in=(mean1, rstd1, g1, b1))
            dz1 = ops.bn_relu_bwd_partials(da1, z1, g1, b1, mean1, rstd1, part1, dg1, db1)
        else:
            da2 = ops.conv3x3_c64(dz3, wb3)
            ops.conv3x3_c64_wgrad(z2, dz3, dw3, bn_in=(mean2, rstd2, g2, b2))
            dz2 = ops.bn_relu_bwd(da2, z2, g2, b2, mean2, rstd2, dg2, db2)
            da1 = ops.conv3x3_c64(dz2, wb2)
            ops.conv3x3_c64_wgrad(z1, dz2, dw2, bn_in=(mean1, rstd1, g1, b1))
            dz1 = ops.bn_relu_bwd(da1, z1, g1, b1, mean1, rstd1, dg1, db1)
        ops.conv7_s2d_wgrad(xs, dz1, dw7)
        gr = list(_finish_param_grads(params, bufs, sunk))
        if not last:
            gr += [None, None]
        return (None, gr[0], gr[1], gr[2], None, None, gr[3], gr[4], gr[5], None, None, gr[6], gr[7], gr[8], None, None, None, None, None, None)


STEM_FUSE_BN = os.environ.get("AP_STEM_FUSE_BN", "1") != "0"
STEM_FUSE_BN_BWD_STATS = os.environ.get("AP_STEM_FUSE_BN_BWD_STATS", "1") != "0"     # BatchNorm-backward sums in the input-gradient convolutions' epilogue
BN_PROJ_ACT_IN_BWD = os.environ.get("AP_BN_PROJ_ACT_IN_BWD", "1") != "0"     # 0: PatchEmbed.proj's weight gradient applies relu(bn(.)) to the chunks it stages
STEM_FUSE_BN_PROJ = os.environ.get("AP_STEM_FUSE_BN_PROJ", "1") != "0"     # the LAST BatchNorm + ReLU of the stem inside PatchEmbed.proj's patch GEMMs


def to_bf16(x):
    """fp32/bf16 torch tensor -> contiguous bf16 (autograd-aware torch cast: stem boundary)"""
    return x.to(BF16).contiguous()


def layer_norm(x, w, b, eps):
    return LayerNormFn.apply(x, w, b, eps)


def linear(x, w, b=None, gelu=False):
    return LinearFn.apply(x, w, b, gelu)
