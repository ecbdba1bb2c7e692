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
