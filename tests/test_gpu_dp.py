"""Data-parallel step on the GPU with two ranks sharing ONE device (gloo backend over CUDA tensors, since
RCCL refuses two ranks on the same GPU): exercises the production DP path end to end -- gradient sink,
side-stream weight gradients, bucketed asynchronous all-reduce from the fused backward, finish(), fused
optimizer -- and checks the averaged gradients against a single-process computation."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(seed=0):
    from autoprog_amd.models import create_model
    torch.manual_seed(seed)
    return create_model("model_variant", variant="volo_h2_l6", num_classes=16, img_size=64, stem_hidden_dim=16, drop_path_rate=0.0).cuda().train()


def _data(rank):
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(4, 3, 64, 64, generator=g).cuda()
    t = torch.softmax(torch.randn(4, 16, 2 + 16, generator=g), dim=1).cuda()
    return x, t


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from autoprog_amd.loss import TokenLabelCrossEntropy
        from autoprog_amd.dist import GradientBucketReducer
        model = _make()
        model.set_sample_config(dict(layer_num=4, min_layer_num=3, max_layer_num=6))     # some layers skipped: no gradients there
        red = GradientBucketReducer(list(model.parameters()), bucket_bytes=64 << 10, world_size=world)
        red.install_sink(model)
        x, t = _data(rank)
        np.random.seed(5)                                   # same mix-token box on both ranks
        red.zero_grad()
        loss = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=16)(model(x), t)
        loss.backward()
        red.finish()
        torch.cuda.synchronize()
        q.put((rank, {n: p.grad.detach().float().cpu().numpy() for n, p in model.named_parameters()}))
    except Exception as e:                                  # pragma: no cover
        import traceback
        q.put((rank, "fail: %r\n%s" % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_mean_on_one_gpu():
    from autoprog_amd.loss import TokenLabelCrossEntropy
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
    assert all(isinstance(v, dict) for v in out.values()), out
    # both ranks hold the same averaged gradients
    out = {r: {n: torch.from_numpy(v) for n, v in d.items()} for r, d in out.items()}
    bad = [n for n in out[0] if not torch.allclose(out[0][n], out[1][n], atol=1e-6)]
    assert not bad, "ranks disagree on %d/%d gradients: %s" % (len(bad), len(out[0]), bad[:40])
    # single-process reference: mean of the two per-rank gradients
    ref = None
    for r in range(world):
        model = _make()
        model.set_sample_config(dict(layer_num=4, min_layer_num=3, max_layer_num=6))
        x, t = _data(r)
        np.random.seed(5)
        TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=16)(model(x), t).backward()
        g = {n: (p.grad.detach().float().cpu() if p.grad is not None else torch.zeros_like(p).float().cpu()) for n, p in model.named_parameters()}
        ref = g if ref is None else {n: ref[n] + g[n] for n in g}
    # fp32 atomics order + bf16 activations: not bitwise.  Tensors whose gradient is tiny (norm far below the typical one)
    # are compared on the typical scale, not on their own (their relative error is rounding noise of the bf16 chain)
    norms = sorted(float((ref[n] / world).norm()) for n in ref if float(ref[n].abs().sum()) > 0)
    typical = norms[len(norms) // 2]
    for n in ref:
        want = ref[n] / world
        err = float((out[0][n] - want).norm()) / max(float(want.norm()), 1e-2 * typical)
        assert err < 2e-2, (n, err, float(want.norm()), typical)
    skipped = [n for n in ref if float(ref[n].abs().sum()) == 0.0 and n.startswith("network.")]
    assert skipped, "the elastic config should leave some layers without gradients"


def test_bench_two_ranks_on_one_gpu():
    """bench.py's multi-rank path (torch.distributed.run launch, barriers, max-over-ranks timing, one JSON line from rank 0)
    with two gloo ranks sharing the GPU (RCCL refuses that; the driver's real runs use nccl)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "16", "--no-roofline"]
    r = subprocess.run(cmd, cwd=root, env=dict(os.environ, AP_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 32 and out["cpu_baseline"] is None
    assert 0 < out["config"]["final_loss"] < 20
