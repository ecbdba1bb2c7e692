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
