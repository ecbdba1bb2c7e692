"""The AutoProg grow / search loop over ONE supernet.

Reference: main_prog.py:786-857 (the epoch loop: at a grow epoch either take the scheduled (r, l) or search for it),
main_prog.py:1558-1821 `auto_grow` (train the search supernet with a random sub-network per step, probe every candidate with the
first EMA copy, rank by loss * time^w) and main_prog.py:1300-1430 `create_stage_model_and_optimizer`.

What is different here, by design (north_star: "elastic depth / num_tokens ... as kernel launch-time shape parameters rather than
weight copies"): the reference builds a new network + optimizer + EMA list at every grow epoch and again for every search; this
driver keeps ONE network sized for the deepest stage with its flat parameter / moment / EMA slabs (optim.FlatAdamWEma) and moves
between stages with `FlatAdamWEma.grow` (prog/elastic.py) and `set_sample_config` -- a candidate of a search is a mask and a
resolution, nothing is copied to evaluate it.  The per-step input resize runs on the device (ap_resize_bilinear_nhwc).

Out of scope: data loading, augmentation strength schedules (only DropPath lives in the model), LR scheduler,
checkpoint files, validation -- the caller supplies `get_batch(r) -> (images, target)` and reads `history`.
"""
import random
import time

import torch

from . import search as S
from .helpers import ActiveLayerMask


def _sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()


class AutoProgDriver:
    def __init__(self, model, loss_fn, optimizer, reducer, get_batch, r_list, l_list, dp_list, grow_epochs, steps_per_epoch,
                 search_epochs=2, auto_grow=True, probe_batches=4, time_steps=4, seed=0, log=None, original_batch_splits=1,
                 r_max=None, dist_bn="", use_graphs=False, graph_after=2, clip_grad=None, clip_mode="norm"):
        """model: supernet sized for l_list[-1] (e.g. volo_h12_l18); optimizer: FlatAdamWEma over it; reducer: its
        GradientBucketReducer; r_list / l_list / dp_list / grow_epochs: the stage schedule (prog/progressive.py:4-31);
        get_batch(r): a training batch (images at ANY size -- the stem resizes to r -- and a token-label target for r // 16).
        original_batch_splits: the reference's --batch-splits-list[-1] (main_prog.py:567): micro-batches per update at the LARGEST
        (l, r); a stage runs `get_divisor(original_batch_splits, l r^2 / (l_max r_max^2))` of them (main_prog.py:570, 842) -- the
        driver then calls get_batch(r, splits) and expects a micro-batch of batch_size // splits images (get_batch(r) when 1).
        A search runs every candidate at `original_batch_splits` (main_prog.py:807-810: batch_size = original_batch_size //
        original_batch_splits for the whole search), timing and probes included.
        dist_bn: "reduce" | "broadcast" | "" -- the reference's --dist-bn: after every epoch the BatchNorm running statistics are
        averaged over the ranks or taken from rank 0 (main_prog.py:883-887).
        use_graphs (single rank, one micro-batch per update): the training steps of an epoch replay ONE HIP graph per stage
        configuration (autoprog_amd/graph.py: the per-step host scalars live in device memory) -- the early AutoProg stages
        (main_prog.py:973-974: every batch resized to the stage's r) are launch-gap bound on an MI355X, stage (9, 128) runs 5.9 -> 4.3 ms
        per step.  The first `graph_after` steps at a configuration run eagerly (lazy initialisations, the allocator's pools), the next
        one is captured; a stage transition drops the graphs.  The steps of a search (a different sub-network every step, main_prog.py:1824-1837)
        replay one graph per candidate drawn (<= 9 per search), captured on the search supernet's slabs the same way.
        With DropPath off a graphed run is bit-identical to the eager one; with it the masks come from the same generator at replay time.
        clip_grad / clip_mode: the reference's --clip-grad / --clip-mode (main_prog.py:129-132, prog/scaler.py:60-68), applied inside the fused
        optimizer step (FlatAdamWEma.step) of every update, eager or replayed."""
        self.use_graphs, self.graph_after = bool(use_graphs), int(graph_after)
        self.clip_grad, self.clip_mode = clip_grad, clip_mode
        self._graphs, self._eager_seen = {}, {}
        self.original_batch_splits = int(original_batch_splits)
        self.dist_bn = dist_bn
        self.r_max = r_max if r_max is not None else max(r_list)
        self.batch_splits = 1
        self.model, self.loss_fn, self.opt, self.reducer, self.get_batch = model, loss_fn, optimizer, reducer, get_batch
        self.r_list, self.l_list, self.dp_list, self.grow_epochs = list(r_list), list(l_list), list(dp_list), list(grow_epochs)
        self.steps_per_epoch, self.search_epochs, self.auto_grow = steps_per_epoch, search_epochs, auto_grow
        self.probe_batches, self.time_steps = probe_batches, time_steps
        self.l_min, self.l_max = min(l_list), max(l_list)
        self.rng = random.Random(seed)
        self.log = log or (lambda *a: None)
        self.history = []                     # one dict per epoch: stage, (r, l), mean loss, search decisions
        self.current_r, self.current_l, self.current_dp = None, None, 0.0
        self.mask = None

    # ------------------------------------------------------------------ elastic plumbing
    def _config(self, l, r):
        return dict(layer_num=l, min_layer_num=self.l_min, max_layer_num=self.l_max, input_size=r, token_label_size=r // 16)

    def _activate(self, l, r, dp):
        mask = self.model.set_sample_config(self._config(l, r))
        self.model.set_drop_path_rate(dp)
        return mask

    def _transition(self, l, r, dp):
        """move the live slabs to depth l (prog/elastic.py), then select (l, r, dp).  As the reference (main_prog.py:828-837): a
        new network state is made when the depth OR the DropPath strength changes; deeper or equal depth -> load='slice' (model
        from the last EMA copy), shallower (a search picked a sub-network of its supernet) -> load='super' (model from the
        trained supernet); the optimizer restarts either way."""
        new_mask = ActiveLayerMask(l, self.l_min, self.l_max)
        self._drop_graphs()                                           # (a graph holds the kernels of ONE configuration and optimizer layout)
        if self.mask is not None and (l != self.current_l or dp != self.current_dp):
            self.opt.grow(self.mask, new_mask, model_source="ema_last" if l >= self.current_l else "model")
        self.mask = self._activate(l, r, dp)
        self.current_l, self.current_r, self.current_dp = l, r, dp
        self.batch_splits = self.splits_for(l, r)

    def splits_for(self, l, r):
        """micro-batches per update at (l, r): main_prog.py:568-570, 839-842"""
        if self.original_batch_splits <= 1:
            return 1
        return S.get_divisor(self.original_batch_splits, (l * r * r) / (self.l_max * self.r_max * self.r_max))

    def _rank_mean(self, values):
        """mean over the ranks of the reducer's process group of a list of host floats (the reference reduces its probe losses
        across ranks, main_prog.py:1213,1267 reduce_tensor): every rank ranks the candidates on the SAME numbers, so every rank
        picks the same (r, l).  One all-reduce of one small tensor."""
        world = getattr(self.reducer, "world", 1)
        if world <= 1:
            return list(values)
        import torch.distributed as dist
        dev = self.reducer.flat.device
        t = torch.tensor(values, dtype=torch.float64, device=dev if dist.get_backend(self.reducer.group) != "gloo" else "cpu")
        dist.all_reduce(t, group=self.reducer.group)
        return (t / world).tolist()

    def _drop_graphs(self):
        for gs in self._graphs.values():
            gs.release()
        self._graphs.clear()
        self._eager_seen.clear()

    def _graph_step(self, l, r, dp):
        """the epoch loop's step from a HIP graph; None: not (yet) -- the caller runs the eager step"""
        key = (l, r, dp)
        gs = self._graphs.get(key)
        if gs is not None:
            images, target = self.get_batch(r)
            return gs.step(images, target).detach().clone()       # (the graph's loss tensor is overwritten by the next replay)
        seen = self._eager_seen.get(key, 0)
        if seen < self.graph_after:
            self._eager_seen[key] = seen + 1
            return None
        from ..graph import GraphedStep
        images, target = self.get_batch(r)
        self.reducer.zero_grad()
        self._set_splits(1)
        gs = GraphedStep(self.model, self.loss_fn, self.reducer, self.opt, images, target, clip_grad=self.clip_grad, clip_mode=self.clip_mode).capture(warmup=0)
        self._graphs[key] = gs
        return gs.step().detach().clone()                          # the batch the graph was built on is this step's batch

    def _train_step(self, l, r, dp, splits=None):
        """one optimizer update = `splits` micro-batches, backward on loss / splits each (main_prog.py:1019 `loss / args.batch_splits`),
        the gradient exchange and the optimizer on the last one (`update`, main_prog.py:971,1026).  -> mean loss (device scalar)"""
        self._activate(l, r, dp)
        k = self.splits_for(l, r) if splits is None else splits
        # one micro-batch per update, one rank: the step replays from the HIP graph of its configuration -- the epoch loop's (l, r), and (round 6)
        # every candidate of a search, which draws a different sub-network per step (main_prog.py:1824-1837): one graph per (l, r) drawn, at most
        # len(rs) * len(ls) <= 9 per search, all on the search supernet's slabs; the first `graph_after` steps at a configuration stay eager
        if (self.use_graphs and k == 1 and getattr(self.reducer, "world", 1) == 1 and torch.cuda.is_available()):
            loss = self._graph_step(l, r, dp)
            if loss is not None:
                return loss
        self._eager()
        self.reducer.zero_grad()                  # (closes whatever update was open: the split count may change now)
        self._set_splits(k)
        total = None
        for _ in range(k):
            images, target = self.get_batch(r) if k == 1 else self.get_batch(r, k)
            loss = self.loss_fn(self.model(images), target)
            (loss if k == 1 else loss / k).backward()
            self.reducer.finish()
            total = loss.detach() if total is None else total + loss.detach()
        if self.clip_grad is not None:
            self.opt.step(clip_grad=self.clip_grad, clip_mode=self.clip_mode)
        else:
            self.opt.step()                       # (any optimizer with the plain step() of the reference's loop works when nothing is clipped)
        return total if k == 1 else total / k

    def _eager(self):
        """an eager forward between replays (a step at a configuration without a graph yet, a probe, a timing pass): the mix-token box and
        lam come from the host again, not from the device scalars of the last replayed graph"""
        if self._graphs and getattr(self.model, "step_scalars", None) is not None:
            self.model.step_scalars = None

    def _set_splits(self, k):
        if hasattr(self.reducer, "set_accumulate_steps"):
            self.reducer.set_accumulate_steps(k)
        elif k != 1:
            raise ValueError("batch splits need a reducer with accumulate_steps")

    def _search_batch(self, r):
        """a batch of the size the reference's search runs on: one micro-batch of `original_batch_splits` (main_prog.py:807-810)"""
        k = self.original_batch_splits
        return self.get_batch(r) if k <= 1 else self.get_batch(r, k)

    # ------------------------------------------------------------------ search (main_prog.py:1558-1821)
    def _probe(self, cands, ema_index=0):
        """train-mode, no-grad loss of EMA copy `ema_index` on `probe_batches` batches per candidate (the reference's taylor0)"""
        out = {}
        self._eager()
        with self.opt.ema_weights(ema_index), torch.no_grad():
            for (r, l) in cands:
                self._activate(l, r, 0.0)
                tot = None
                for _ in range(self.probe_batches):
                    images, target = self._search_batch(r)
                    loss = self.loss_fn(self.model(images), target).float()
                    tot = loss if tot is None else tot + loss
                out[(r, l)] = tot / self.probe_batches
        vals = torch.stack([out[c] for c in cands]).tolist()          # ONE read-back per probe
        return dict(zip(cands, self._rank_mean(vals)))

    def _time(self, cands):
        """mean forward+backward seconds per candidate, measured once at search start (main_prog.py:1886-1902), on the search's
        micro-batch.  Every timed pass is an update of ONE micro-batch whatever split count the last training step left in the
        reducer: each candidate's time then includes the gradient exchange, and the reducer is left between updates."""
        out = {}
        self._eager()
        self.reducer.zero_grad()
        self._set_splits(1)
        for (r, l) in cands:
            self._activate(l, r, self.current_dp)
            images, target = self._search_batch(r)
            for i in range(self.time_steps + 1):
                if i == 1:
                    _sync()
                    t0 = time.perf_counter()
                self.reducer.zero_grad()
                self.loss_fn(self.model(images), target).backward()
                self.reducer.finish()
                self.reducer.take_pending_scale()
            _sync()
            out[(r, l)] = (time.perf_counter() - t0) / self.time_steps
        return dict(zip(cands, self._rank_mean([out[c] for c in cands])))

    def search(self, stage, epoch):
        """-> chosen (r, l).  The search supernet is the largest candidate; its sub-networks are trained with one random candidate
        per step for `search_epochs` epochs under the FINAL DropPath strength (main_prog.py:814-815), probed, and ranked by
        loss * time^w (prog/search.py converge_speed)."""
        rs, _, ls = S.search_space(stage, self.r_list, [0] * len(self.r_list), self.l_list, self.current_r or self.r_list[0], 0,
                                   self.current_l or self.l_list[0])
        if ls[-1] > 2 * ls[0]:
            raise ValueError("a search over more than 2x depth is not defined (main_prog.py:1562)")
        cands = [(r, l) for r in rs for l in ls]
        dp_final = self.dp_list[-1]
        self._transition(ls[-1], rs[-1], dp_final)                    # grow the live slabs to the search supernet
        times = self._time(cands)
        losses = []
        for e in range(self.search_epochs):
            probes = 1 if e == 0 else 4                               # main_prog.py:1617: probe points per search epoch
            per = max(1, self.steps_per_epoch // probes)
            for p in range(probes):
                losses.append(self._probe(cands))
                for _ in range(per):
                    r, l = self.rng.choice(rs), self.rng.choice(ls)   # one config per step, identical on every rank (seeded)
                    self._train_step(l, r, dp_final, splits=self.original_batch_splits)     # main_prog.py:807: the search's split count
            self._distribute_bn()                                     # main_prog.py:1634-1637: after every epoch of the search as well
        losses.append(self._probe(cands))
        mean_loss = {"r%d_l%d" % c: sum(p[c] for p in losses) / len(losses) for c in cands}
        step_time = {"r%d_l%d" % c: times[c] for c in cands}
        if len(cands) >= 2 and len({round(t, 9) for t in step_time.values()}) >= 2:
            w, scores, order = S.converge_speed(mean_loss, step_time)
        else:
            w, scores, order = 0.0, dict(mean_loss), sorted(mean_loss, key=mean_loss.get)
        best = order[0]
        r, l = (int(v[1:]) for v in best.split("_"))
        self.log("search @%d: w=%.3f scores=%s -> %s" % (epoch, w, {k: round(v, 4) for k, v in scores.items()}, best))
        self.history.append(dict(epoch=epoch, kind="search", candidates=["r%d_l%d" % c for c in cands], w=w, scores=scores, chosen=(r, l),
                                 step_time=step_time, mean_loss=mean_loss))
        return r, l

    # ------------------------------------------------------------------ epoch loop (main_prog.py:786-857)
    def run(self, num_epochs):
        skip = set()
        for epoch in range(num_epochs):
            if epoch in self.grow_epochs:
                stage = self.grow_epochs.index(epoch)
                if self.auto_grow and stage < len(self.grow_epochs) - 1:
                    r, l = self.search(stage, epoch)
                    skip.update(range(epoch, epoch + self.search_epochs))      # the search consumed these epochs (main_prog.py:856-857)
                else:
                    r, l = self.r_list[stage], self.l_list[stage]
                self._transition(l, r, self.dp_list[stage])
            if epoch in skip:
                continue
            tot = None                    # summed on the device: one read-back per epoch, not one per step
            for _ in range(self.steps_per_epoch):
                loss = self._train_step(self.current_l, self.current_r, self.current_dp)
                tot = loss.float() if tot is None else tot + loss
            tot = float(tot)
            self.history.append(dict(epoch=epoch, kind="train", r=self.current_r, l=self.current_l, dp=self.current_dp,
                                     loss=tot / self.steps_per_epoch))
            self.log("epoch %d: r=%d l=%d loss %.4f" % (epoch, self.current_r, self.current_l, tot / self.steps_per_epoch))
            self._distribute_bn()
        return self.history

    def _distribute_bn(self):
        """main_prog.py:883-887: `if args.distributed and args.dist_bn in ('broadcast', 'reduce')` after every training epoch, :1634-1637
        after every epoch of a search; and the same for each EMA copy (:895-899, :1650-1654 `distribute_bn(model_ema_list[idx], ...)`):
        here the EMA copies of the BatchNorm buffers are tensors of the flat optimizer, one flat message per copy."""
        world = getattr(self.reducer, "world", 1)
        if world > 1 and self.dist_bn in ("broadcast", "reduce"):
            from ..dist import distribute_bn
            red = self.dist_bn == "reduce"
            distribute_bn(self.model, world, reduce=red, group=self.reducer.group)
            names = [n for n, _ in getattr(self.opt, "_float_buffers", [])]
            for bufs in getattr(self.opt, "ema_buffers", []):
                distribute_bn(None, world, reduce=red, group=self.reducer.group, named_buffers=list(zip(names, bufs)))
