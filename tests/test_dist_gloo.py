"""world_size-2 CPU (gloo) test of the data-parallel gradient exchange: bucketed asynchronous
all-reduce launched from gradient hooks, finish() for layers skipped by elastic depth, and the
scalar-mean helper (timm reduce_tensor semantics)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from autoprog_amd.dist import GradientBucketReducer, reduce_scalar_mean
        torch.manual_seed(0)                       # identical weights on both ranks
        net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 32), torch.nn.ReLU(),
                                  torch.nn.Linear(32, 4))
        unused = torch.nn.Linear(8, 8)             # a layer skipped by the elastic config: never receives a gradient
        params = list(net.parameters()) + list(unused.parameters())
        red = GradientBucketReducer(params, bucket_bytes=512, world_size=world)      # several small buckets
        assert len(red.buckets) >= 3, len(red.buckets)
        results = []
        for step in range(2):
            torch.manual_seed(100 + 10 * step + rank)        # different data per rank
            x = torch.randn(8, 16)
            red.zero_grad()
            loss = net(x).pow(2).mean()
            loss.backward()
            red.finish()
            results.append([p.grad.clone() for p in params])
            # reference: plain autograd on every rank's data
            ref = [torch.zeros_like(p) for p in params]
            for r in range(world):
                torch.manual_seed(100 + 10 * step + r)
                xr = torch.randn(8, 16)
                gs = torch.autograd.grad(net(xr).pow(2).mean(), list(net.parameters()))
                for acc, g in zip(ref, gs):
                    acc += g / world
            for got, want in zip(results[-1], ref):
                assert torch.allclose(got, want, atol=1e-6), (rank, step)
        # gradient-sink protocol (functional.set_grad_sink): a fused backward writes param.grad in place, calls
        # param_ready() and returns None to autograd -- whose AccumulateGrad hook may still fire for that
        # parameter; a bucket must not be counted ready twice (it would launch before its other members exist)
        red.remove()
        red = GradientBucketReducer(list(net.parameters()), bucket_bytes=1 << 20, world_size=world)    # ONE bucket
        assert len(red.buckets) == 1

        class SinkLinear(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, w):
                ctx.save_for_backward(x, w)
                return x @ w.t()

            @staticmethod
            def backward(ctx, dy):
                x, w = ctx.saved_tensors
                for hnd in red._handles:          # make a prematurely launched all-reduce deterministic: let it complete
                    hnd.wait()
                w.grad += dy.t() @ x
                red.param_ready(w)
                return dy @ w, None

        lins = [net[0], net[2], net[4]]
        torch.manual_seed(300 + rank)
        x = torch.randn(8, 16)
        red.zero_grad()
        h = x
        for i, lin in enumerate(lins):
            h = SinkLinear.apply(h, lin.weight) + lin.bias
            h = torch.relu(h) if i < 2 else h
        h.pow(2).mean().backward()
        red.finish()
        ref = [torch.zeros_like(p) for p in net.parameters()]
        for r in range(world):
            torch.manual_seed(300 + r)
            gs = torch.autograd.grad(net(torch.randn(8, 16)).pow(2).mean(), list(net.parameters()))
            for acc, g in zip(ref, gs):
                acc += g / world
        for got, want in zip([p.grad for p in net.parameters()], ref):
            assert torch.allclose(got, want, atol=1e-6), (rank, "sink")
        # deferred delivery (functional's weight-gradient window): the fused backward only NOTES its weight gradient, tells the sink to
        # hold() the parameter -- autograd's post-accumulate hook fires when the backward returns, with nothing written yet -- and the
        # end of the backward pass writes all of them and calls param_ready().  No bucket may leave before that.
        pending = []

        class DeferredLinear(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, w):
                ctx.save_for_backward(x, w)
                return x @ w.t()

            @staticmethod
            def backward(ctx, dy):
                x, w = ctx.saved_tensors
                if not pending:
                    torch.autograd.Variable._execution_engine.queue_callback(deliver)
                pending.append((w, dy, x))
                red.hold([w])
                return dy @ w, None

        def deliver():
            assert not any(red._launched), "a bucket left before its held members were written"
            for w, dy, x in pending:
                w.grad += dy.t() @ x
                red.param_ready(w)
            pending.clear()

        red.zero_grad()
        h = x
        for i, lin in enumerate(lins):
            h = DeferredLinear.apply(h, lin.weight) + lin.bias
            h = torch.relu(h) if i < 2 else h
        h.pow(2).mean().backward()
        assert not pending and all(red._launched)          # the last param_ready completed the bucket: launched before finish()
        red.finish()
        for got, want in zip([p.grad for p in net.parameters()], ref):
            assert torch.allclose(got, want, atol=1e-6), (rank, "deferred sink")
        # bf16 buckets (comm_dtype): half the bytes on the wire, fp32 slab; equal to the fp32 exchange within bf16 rounding of the buckets
        red.remove()
        red = GradientBucketReducer(list(net.parameters()), bucket_bytes=512, world_size=world, comm_dtype=torch.bfloat16)
        torch.manual_seed(300 + rank)
        red.zero_grad()
        net(torch.randn(8, 16)).pow(2).mean().backward()
        red.finish()
        for got, want in zip([p.grad for p in net.parameters()], ref):
            assert got.dtype == torch.float32 and torch.allclose(got, want, rtol=2e-2, atol=1e-3), (rank, "bf16 buckets")
        assert any(not torch.equal(got, want) for got, want in zip([p.grad for p in net.parameters()], ref))
        m = reduce_scalar_mean(torch.tensor(float(rank)), world)
        assert float(m) == pytest.approx((world - 1) / 2)
        # distribute_bn (main_prog.py:883-887): running statistics averaged over the ranks, or rank 0's
        from autoprog_amd.dist import distribute_bn
        bn = torch.nn.Sequential(torch.nn.BatchNorm2d(4), torch.nn.Conv2d(4, 4, 1), torch.nn.BatchNorm2d(4))
        for k, mod in enumerate((bn[0], bn[2])):
            mod.running_mean.fill_(float(rank + k)); mod.running_var.fill_(float(10 * rank + k + 1))
        distribute_bn(bn, world, reduce=True)
        assert torch.allclose(bn[0].running_mean, torch.full((4,), 0.5)) and torch.allclose(bn[2].running_var, torch.full((4,), 7.0))
        assert int(bn[0].num_batches_tracked) == 0
        for k, mod in enumerate((bn[0], bn[2])):
            mod.running_mean.fill_(float(rank + k))
        distribute_bn(bn, world, reduce=False)
        assert torch.allclose(bn[0].running_mean, torch.zeros(4)) and torch.allclose(bn[2].running_mean, torch.ones(4))
        q.put((rank, "ok"))
    except Exception as e:                          # pragma: no cover
        q.put((rank, "fail: %r" % (e,)))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_world_size_2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(msg == "ok" for _, msg in out), out


def _driver_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from autoprog_amd.prog.driver import AutoProgDriver
        from autoprog_amd.prog import search as S

        class _Reducer:                      # what AutoProgDriver._rank_mean reads of a GradientBucketReducer
            def __init__(self):
                self.world, self.group, self.flat = world, None, torch.zeros(4)
        drv = AutoProgDriver(model=None, loss_fn=None, optimizer=None, reducer=_Reducer(), get_batch=None, r_list=[64, 96], l_list=[3, 6],
                             dp_list=[0.0, 0.1], grow_epochs=[0, 2], steps_per_epoch=1)
        cands = [(64, 3), (64, 6), (96, 3), (96, 6)]
        # rank-local probe losses and step times that would rank the candidates DIFFERENTLY on the two ranks
        local_loss = [[2.0, 1.0, 1.9, 0.6], [1.0, 2.4, 1.7, 1.2]][rank]
        local_time = [[1.0, 2.0, 2.0, 4.5], [1.2, 1.8, 2.2, 3.5]][rank]
        mean_loss = dict(zip(("r%d_l%d" % c for c in cands), drv._rank_mean(local_loss)))
        step_time = dict(zip(("r%d_l%d" % c for c in cands), drv._rank_mean(local_time)))
        _, _, order = S.converge_speed(mean_loss, step_time)
        _, _, local_order = S.converge_speed(dict(zip(mean_loss, local_loss)), dict(zip(step_time, local_time)))
        q.put((rank, order[0], local_order[0], [round(v, 9) for v in mean_loss.values()]))
    finally:
        dist.destroy_process_group()


def test_search_decision_is_identical_on_every_rank():
    """AutoProgDriver ranks the candidates of a search on probe losses and step times averaged over the ranks (reference:
    validate_trainset -> reduce_tensor, main_prog.py:1213,1267): with rank-local numbers that disagree, both ranks still pick the
    same (r, l) -- ranks training different sub-networks would diverge silently (same-sized gradient slab)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_driver_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, pick0, local0, means0), (_, pick1, local1, means1) = res
    assert means0 == means1 == [1.5, 1.7, 1.8, 0.9]
    assert pick0 == pick1
    assert local0 != local1            # the rank-local rankings really disagreed


# ----------------------------------------------------------------------------------------------------------------------------------
# gradient accumulation (the reference's --batch-splits / `update` flag: main_prog.py:567-574,971,1019-1027, prog/scaler.py:60-68)
def _accum_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from autoprog_amd.dist import GradientBucketReducer
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 32), torch.nn.ReLU(), torch.nn.Linear(32, 4))
        params = list(net.parameters())
        k = 3
        red = GradientBucketReducer(params, bucket_bytes=512, world_size=world, accumulate_steps=k)
        assert len(red.buckets) >= 3
        torch.manual_seed(500 + rank)
        big = torch.randn(8 * k, 16)                       # this rank's share of the update's batch
        red.zero_grad()
        for i in range(k):
            loss = net(big[8 * i:8 * (i + 1)]).pow(2).mean()
            (loss / k).backward()
            if i < k - 1:
                assert not any(red._launched), "a bucket left on a micro-batch that only accumulates"
            red.finish()
            assert red.is_update_step == (i == k - 1)
        got = [p.grad.clone() for p in params]
        # == ONE large batch per rank (mean over its 8k samples), averaged over the ranks
        ref = [torch.zeros_like(p) for p in params]
        for r in range(world):
            torch.manual_seed(500 + r)
            xr = torch.randn(8 * k, 16)
            for acc, g in zip(ref, torch.autograd.grad(net(xr).pow(2).mean(), params)):
                acc += g / world
        for a, b in zip(got, ref):
            assert torch.allclose(a, b, atol=1e-6), (rank, "accumulation")
        # a second update after zero_grad(): the micro-batch counter starts over; and the split count may change between updates
        red.set_accumulate_steps(2)
        red.zero_grad()
        for i in range(2):
            (net(big[8 * i:8 * (i + 1)]).pow(2).mean() / 2).backward()
            red.finish()
        ref2 = [torch.zeros_like(p) for p in params]
        for r in range(world):
            torch.manual_seed(500 + r)
            xr = torch.randn(8 * k, 16)[:16]
            for acc, g in zip(ref2, torch.autograd.grad(net(xr).pow(2).mean(), params)):
                acc += g / world
        for a, b in zip([p.grad for p in params], ref2):
            assert torch.allclose(a, b, atol=1e-6), (rank, "second update")
        q.put((rank, "ok"))
    except Exception as e:                          # pragma: no cover
        import traceback
        q.put((rank, "fail: %r %s" % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


def _spawn(fn, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=fn, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    return out


def test_gradient_accumulation_equals_one_large_batch_world_size_2():
    out = _spawn(_accum_worker)
    assert all(msg == "ok" for _, msg in out), out


# ----------------------------------------------------------------------------------------------------------------------------------
# the weight-gradient window of functional.py under data parallelism: the REAL window code (collect the blocks' problems, launch when
# a window is full or when it completes a gradient bucket, deliver through param_ready) over the REAL reducer; only the HIP launch
# itself (ops.gemm_tn_acc_grouped) is replaced by the same arithmetic in torch, so that this runs without a GPU.
def _window_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from autoprog_amd import functional as AF, ops
        from autoprog_amd.dist import GradientBucketReducer

        launches = []                                   # (blocks whose backward had run, problems in the launch)
        done_blocks = [0]

        def cpu_grouped(problems, ln=None):
            launches.append((done_blocks[0], len(problems)))
            for prob in problems:
                a, b, c, n1, n2, colsum = prob[:6]
                c += a[:, :n1].float().t() @ b[:, :n2].float()
                if colsum is not None:
                    colsum += a[:, :n1].float().sum(0)
        ops.gemm_tn_acc_grouped = cpu_grouped
        AF.WGRAD_WINDOW = 4                             # tiles per launch: every problem below is ONE 192 x 192 tile

        T, C, NB = 4096, 192, 12

        class Block(torch.autograd.Function):          # y = x W^T + b with a sunk, windowed weight gradient (the shape of functional.*BlockFn)
            @staticmethod
            def forward(ctx, x, w, b):
                ctx.save_for_backward(x, w)
                ctx.params = (w, b)
                return (x.float() @ w.t() + b).to(torch.bfloat16)

            @staticmethod
            def backward(ctx, dy):
                x, w = ctx.saved_tensors
                bufs, sunk = AF._param_grad_buffers(ctx.params)
                assert sunk
                with AF.wgrad_batch(sunk=True, params=ctx.params) as batch:
                    AF._wgrad_batch.append((dy.contiguous(), x, bufs[0], C, C, bufs[1]))
                assert batch.deferred
                AF._finish_param_grads(ctx.params, bufs, sunk, deferred=True)
                done_blocks[0] += 1
                return (dy.float() @ w).to(torch.bfloat16), None, None

        torch.manual_seed(0)
        ws = [torch.nn.Parameter(torch.randn(C, C) * 0.05) for _ in range(NB)]
        bs = [torch.nn.Parameter(torch.zeros(C)) for _ in range(NB)]
        params = [p for pair in zip(ws, bs) for p in pair]
        red = GradientBucketReducer(params, bucket_bytes=3 * (C * C + C) * 4, world_size=world)      # a bucket = three blocks
        assert len(red.buckets) == 4
        red.install_sink()
        torch.manual_seed(700 + rank)
        x = (torch.randn(T, C) * 0.5).to(torch.bfloat16)

        def run(twice=None):
            red.zero_grad()
            launches.clear()
            done_blocks[0] = 0
            h = x.clone().requires_grad_(True)
            for i in range(NB):
                h = Block.apply(h, ws[i], bs[i])
                if twice == i:
                    h = Block.apply(h, ws[i], bs[i])     # the same parameters a second time in one backward pass
            h.float().pow(2).mean().backward()
            log = list(red.launch_log)
            red.finish()
            return log

        log = run()
        # the window launched DURING the backward pass, and bucket all-reduces left before it was over
        assert len(launches) >= 3 and launches[0][0] < NB, launches
        assert any(n_seen < len(params) for _, n_seen in log), ("no bucket left before the end of backward", log)
        assert len(log) == 4 and [b for b, _ in log] == sorted(b for b, _ in log), log          # reverse registration = backward order
        # the second bucket (blocks 8..6) completes inside a window of 4 tiles: the window launched early for it (3 problems, not 4)
        assert any(n == 3 for _, n in launches), launches
        got = [p.grad.clone() for p in params]

        def reference(twice=None):
            acc = [torch.zeros_like(p) for p in params]
            for r in range(world):
                torch.manual_seed(700 + r)
                h = (torch.randn(T, C) * 0.5).to(torch.bfloat16)
                hs = h.clone().requires_grad_(True)
                h = hs
                for i in range(NB):
                    for _ in range(2 if twice == i else 1):
                        h = (h.float() @ ws[i].t() + bs[i]).to(torch.bfloat16)
                for a, g in zip(acc, torch.autograd.grad(h.float().pow(2).mean(), params)):
                    a += g / world
            return acc
        for a, b in zip(got, reference()):
            assert torch.allclose(a, b, rtol=2e-2, atol=2e-4), (rank, "window")
        # a block applied twice before one backward: the second use must not share a launch with the first -- the window
        # launches what it holds when a parameter comes in again -- and the two uses add up
        run(twice=5)
        ref2 = reference(twice=5)
        for a, b in zip([p.grad for p in params], ref2):
            assert torch.allclose(a, b, rtol=2e-2, atol=2e-4), (rank, "twice")
        # a backward pass that raises leaves problems behind; the next pass drops them instead of adding them to its gradients
        red.zero_grad()
        h = x.clone().requires_grad_(True)
        for i in range(3):
            h = Block.apply(h, ws[i], bs[i])

        class Boom(torch.autograd.Function):
            @staticmethod
            def forward(ctx, t):
                return t.clone()

            @staticmethod
            def backward(ctx, g):
                raise RuntimeError("boom")
        y = Boom.apply(h)
        for i in range(3, 5):
            y = Block.apply(y, ws[i], bs[i])
        try:
            y.float().pow(2).mean().backward()
            raise AssertionError("backward did not raise")
        except RuntimeError as e:
            assert "boom" in str(e)
        assert AF._window["units"], "the failed pass should have left its problems in the window"
        log = run()
        for a, b in zip([p.grad for p in params], reference()):
            assert torch.allclose(a, b, rtol=2e-2, atol=2e-4), (rank, "after a failed pass")
        red.remove()
        q.put((rank, "ok"))
    except Exception as e:                          # pragma: no cover
        import traceback
        q.put((rank, "fail: %r %s" % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


def test_weight_gradient_window_lets_buckets_leave_before_backward_ends_world_size_2():
    out = _spawn(_window_worker)
    assert all(msg == "ok" for _, msg in out), out


# ----------------------------------------------------------------------------------------------------------------------------------
# world size 8 (one node's worth of ranks): elastic skip + 3-way accumulation + the weight-gradient window in ONE
# update, bucket order identical on every rank; the driver's distribute_bn over the model AND the EMA copies of the BatchNorm buffers.
def _ws8_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from autoprog_amd import functional as AF, ops
        from autoprog_amd.dist import GradientBucketReducer

        def cpu_grouped(problems, ln=None):
            for prob in problems:
                a, b, c, n1, n2, colsum = prob[:6]
                c += a[:, :n1].float().t() @ b[:, :n2].float()
                if colsum is not None:
                    colsum += a[:, :n1].float().sum(0)
        ops.gemm_tn_acc_grouped = cpu_grouped
        AF.WGRAD_WINDOW = 4
        T, C, NB, K = 512, 192, 9, 3                    # 9 blocks, the elastic configuration skips blocks 3 and 6; 3 micro-batches per update

        class Block(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, w, b):
                ctx.save_for_backward(x, w)
                ctx.params = (w, b)
                return (x.float() @ w.t() + b).to(torch.bfloat16)

            @staticmethod
            def backward(ctx, dy):
                x, w = ctx.saved_tensors
                bufs, sunk = AF._param_grad_buffers(ctx.params)
                with AF.wgrad_batch(sunk=True, params=ctx.params) as batch:
                    AF._wgrad_batch.append((dy.contiguous(), x, bufs[0], C, C, bufs[1]))
                AF._finish_param_grads(ctx.params, bufs, sunk, deferred=True)
                return (dy.float() @ w).to(torch.bfloat16), None, None

        torch.manual_seed(0)
        ws = [torch.nn.Parameter(torch.randn(C, C) * 0.05) for _ in range(NB)]
        bs = [torch.nn.Parameter(torch.zeros(C)) for _ in range(NB)]
        params = [p for pair in zip(ws, bs) for p in pair]
        active = [i for i in range(NB) if i not in (3, 6)]
        red = GradientBucketReducer(params, bucket_bytes=2 * (C * C + C) * 4, world_size=world, accumulate_steps=K)
        red.install_sink()
        torch.manual_seed(900 + rank)
        xs = [(torch.randn(T, C) * 0.5).to(torch.bfloat16) for _ in range(K)]
        red.zero_grad()
        for i, x in enumerate(xs):
            h = x.clone().requires_grad_(True)
            for j in active:
                h = Block.apply(h, ws[j], bs[j])
            (h.float().pow(2).mean() / K).backward()
            if i < K - 1:
                assert not any(red._launched), "a bucket left on a micro-batch that only accumulates"
            log = list(red.launch_log)
            red.finish()
        order = ([b for b, _ in log], [b for b, _ in red.launch_log])      # buckets that left during the last backward pass; all of them
        acc = [torch.zeros_like(p) for p in params]
        for r in range(world):
            torch.manual_seed(900 + r)
            for _ in range(K):
                h = (torch.randn(T, C) * 0.5).to(torch.bfloat16)
                for j in active:
                    h = (h.float() @ ws[j].t() + bs[j]).to(torch.bfloat16)
                gs = torch.autograd.grad(h.float().pow(2).mean() / K, params, allow_unused=True)
                for a, g in zip(acc, gs):
                    if g is not None:
                        a += g / world
        for i, (p, want) in enumerate(zip(params, acc)):
            assert torch.allclose(p.grad, want, rtol=2e-2, atol=2e-4), (rank, "ws8 gradient", i)
        for j in (3, 6):
            assert float(ws[j].grad.abs().max()) == 0.0, "a skipped block received a gradient"
        red.remove()

        # the driver's distribute_bn: the model's BatchNorm statistics and every EMA copy of them, after a search epoch as after a training epoch
        from autoprog_amd.prog.driver import AutoProgDriver
        model = torch.nn.Sequential(torch.nn.BatchNorm2d(4), torch.nn.Conv2d(4, 4, 1), torch.nn.BatchNorm2d(4))

        class _Opt:
            def __init__(self, m):
                self._float_buffers = [(n, b) for n, b in m.named_buffers() if b.dtype.is_floating_point]
                self.ema_buffers = [[b.detach().clone() for _, b in self._float_buffers] for _ in range(2)]

        class _Red:
            def __init__(self):
                self.world, self.group, self.flat = world, None, torch.zeros(4)
        opt = _Opt(model)
        for k, mod in enumerate((model[0], model[2])):
            mod.running_mean.fill_(float(rank + k)); mod.running_var.fill_(float(2 * rank + k + 1))
        for e, bufs in enumerate(opt.ema_buffers):
            for b in bufs:
                b.fill_(float(10 * e + rank))
        drv = AutoProgDriver(model=model, loss_fn=None, optimizer=opt, reducer=_Red(), get_batch=None, r_list=[64, 96], l_list=[3, 6],
                             dp_list=[0.0, 0.1], grow_epochs=[0, 2], steps_per_epoch=1, dist_bn="reduce")
        drv._distribute_bn()
        mean_rank = (world - 1) / 2
        assert torch.allclose(model[0].running_mean, torch.full((4,), mean_rank)) and torch.allclose(model[2].running_var, torch.full((4,), 2 * mean_rank + 2))
        for e, bufs in enumerate(opt.ema_buffers):
            for b in bufs:
                assert torch.allclose(b, torch.full_like(b, 10 * e + mean_rank)), (rank, "EMA BatchNorm buffers", e)
        q.put((rank, "ok", order))
    except Exception as e:                          # pragma: no cover
        import traceback
        q.put((rank, "fail: %r %s" % (e, traceback.format_exc()), None))
    finally:
        dist.destroy_process_group()


def test_elastic_skip_accumulation_and_window_world_size_8():
    """eight ranks (gloo): blocks skipped by the elastic configuration, three micro-batches per update and the weight-gradient window in
    one update -- gradients equal the mean over 8 x 3 micro-batches, the skipped blocks stay zero, every rank launched its buckets in the
    SAME order (a rank that reduced bucket i against another rank's bucket j would hang or mix gradients), and the driver's
    distribute_bn averages the model's and the EMA copies' BatchNorm statistics (main_prog.py:883-899, 1634-1654)."""
    out = _spawn(_ws8_worker, world=8)
    assert all(o[1] == "ok" for o in out), out
    orders = [o[2] for o in out]
    # buckets without a skipped member left during the backward pass of the last micro-batch, the other two in finish(): same order everywhere
    assert all(o == orders[0] for o in orders), orders
    assert orders[0][0] == [0, 3, 4] and sorted(orders[0][1]) == [0, 1, 2, 3, 4] and orders[0][1][:3] == [0, 3, 4], orders[0]


# ----------------------------------------------------------------------------------------------------------------------------------
# bench.py's timed pre-warm on more than one rank
def _prewarm_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        import time
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        buf = torch.zeros(4)
        steps = [0]

        def step():                                # a "training step": rank-dependent host time, then the gradient exchange
            time.sleep(0.004 if rank == 0 else 0.009)
            dist.all_reduce(buf)
            steps[0] += 1
        n, secs = bench.prewarm(step, 0.25, lambda: None, lambda t: bench.all_ranks_max(t, "cpu"))
        dist.barrier()                             # every collective above has found its partner: nobody is left inside an all-reduce
        q.put((rank, n, steps[0], secs))
    finally:
        dist.destroy_process_group()


def test_bench_prewarm_stops_every_rank_after_the_same_step_world_size_2():
    """bench.py runs >= --prewarm-s seconds of the step in front of the counted warm-up.  The steps of an N-rank run contain collectives, so the
    ranks must leave that loop after the SAME number of steps although their clocks differ: the deciding clock is the maximum over the ranks
    (bench.prewarm / all_ranks_max).  Two gloo ranks whose steps take 4 and 9 ms of host time: equal step counts, no rank left waiting."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_prewarm_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, n0, s0, t0), (_, n1, s1, t1) = res
    assert n0 == n1 == s0 == s1 and n0 % 8 == 0 and n0 >= 8
    assert t0 == t1 and t0 >= 0.25               # the same (maximum) clock on both ranks
