#!/usr/bin/env python3
"""Benchmark of the AutoProg VOLO-D1 224 px training step on MI355X (BASELINE.json metric
"images/sec/GPU (fwd+bwd) VOLO-D1 224px AutoProg step").

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A "step" is one pass of the hot path over one synthetic batch already resident in HBM:
forward (mix-token, DropPath 0.1) + token-label loss + backward (+ bucketed RCCL gradient
all-reduce overlapped with backward when N>1) + AdamW update + the 4 EMA updates of
scripts/train_autoprog.sh.  Workload = BASELINE.json configs[1]: volo_h12_l18 (== VOLO-D1),
224x224, per-GPU batch 128, bf16 activations / fp32 master weights, token-label target [B,1000,198].

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     dominant kernel (k_gemm_nt, bound "mfma"): algorithmic FLOPs of its launches in one
               step / their summed duration, measured with HIP events on the launch stream
  cpu_baseline the CPU oracle (oracle/ref_cpu.py, kind "port") timed on this host on a bounded
               sample of the same workload (smaller batch), rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0       # MI355X dense bf16 MFMA peak (/opt/skills/guides/MI355X_MICROARCH.md)


def make_target(B, C, N, device, gen):
    """token-label style soft targets [B,C,2+N]: top-5 sparse labels mixed with smoothing 0.1
    (SURVEY.md section 8(d) row M2)."""
    t = torch.zeros(B, C, 2 + N)
    idx = torch.randint(0, C, (B, 5, 2 + N), generator=gen)
    val = torch.rand(B, 5, 2 + N, generator=gen)
    val = val / val.sum(1, keepdim=True)
    t.scatter_(1, idx, val)
    t = t * 0.9 + 0.1 / C
    return t.to(device)


def cpu_baseline(variant, res, seconds_budget, threads):
    """time the CPU oracle (pure torch fp32 restatement of the reference) on a bounded sample"""
    from oracle import ref_cpu as R
    torch.set_num_threads(threads)
    arch = R.variant_arch(variant)
    from autoprog_amd.models import create_model
    torch.manual_seed(42)
    model = create_model("model_variant", variant=variant, drop_path_rate=0.1)
    p = {k: v.detach().clone().float().requires_grad_(v.dtype.is_floating_point and "running_" not in k)
         for k, v in model.state_dict().items()}
    B = 8
    g = torch.Generator().manual_seed(42)
    x = torch.randn(B, 3, res, res, generator=g)
    n = (res // 16) ** 2
    target = make_target(B, 1000, n, "cpu", g)
    rng = np.random.RandomState(42)

    def step():
        lam, box = R.draw_mix_box((B, res // 8, res // 8, arch["embed_dims"][0]), 2, 1.0, rng)
        out = R.volo_forward(p, x, train=True, mix=(lam, box), drop_path_rate=0.1, **arch)
        loss = R.token_label_ce(out, target, 0.5, 1.0)
        grads = torch.autograd.grad(loss, [v for v in p.values() if v.requires_grad], allow_unused=True)
        return float(loss.detach()), grads

    step()                                   # warm-up
    t0 = time.time()
    n_steps = 0
    while True:
        step()
        n_steps += 1
        if time.time() - t0 > seconds_budget or n_steps >= 10:
            break
    dt = time.time() - t0
    return {"value": round(B * n_steps / dt, 3), "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": "oracle/ref_cpu.py fp32 fwd+loss+bwd, %s %dpx, batch %d, %d steps in %.1fs (no optimizer)" % (variant, res, B, n_steps, dt)}


def self_launch(n, argv):
    """start `torch.distributed.run --nproc-per-node n bench.py <argv>` as a child and return its exit code"""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC (RCCL needs it on this driver)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


class GemmProbe:
    """HIP-event timing of every ap_gemm_nt launch (events recorded on the launch stream)"""

    def __init__(self):
        self.records = []
        self.keys = []
        self.bytes = 0.0

    def install(self):
        from autoprog_amd import ops
        self._orig = ops.gemm_nt
        probe = self

        def timed(a, b, n=None, k=None, **kw):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            out = probe._orig(a, b, n=n, k=k, **kw)
            e1.record()
            nn = b.shape[0] if n is None else n
            kk = a.shape[1] if k is None else k
            probe.records.append((e0, e1, 2.0 * a.shape[0] * nn * kk))
            probe.keys.append((a.shape[0], nn, kk, "+".join(k2 for k2 in ("bias", "gelu", "dgelu_of", "row_scale", "residual") if kw.get(k2) is not None and kw.get(k2) is not False) or "plain"))
            # algorithmic bytes: A + B + C once, plus the epilogue operands this call reads / writes
            extra = sum(1 for k2 in ("residual", "dgelu_of", "preact_out") if kw.get(k2) is not None)
            probe.bytes += 2.0 * (a.shape[0] * kk + nn * kk + a.shape[0] * nn * (1 + extra))
            return out
        ops.gemm_nt = timed

    def remove(self):
        from autoprog_amd import ops
        ops.gemm_nt = self._orig

    def table(self):
        """per-shape HIP-event times INSIDE the training step (AP_GEMM_TABLE=1): the ground truth for tile-variant choices --
        back-to-back microbenchmarks on one buffer set run cache-warm and rank the variants differently"""
        torch.cuda.synchronize()
        agg = {}
        for (e0, e1, fl), key in zip(self.records, self.keys):
            t = agg.setdefault(key, [0, 0.0, 0.0])
            t[0] += 1; t[1] += e0.elapsed_time(e1) * 1e3; t[2] += fl
        lines = ["%7d %5d %5d %-28s x%-4d %8.1f us  %7.1f TFLOP/s" % (k[0], k[1], k[2], k[3], v[0], v[1] / v[0], v[2] / v[1] / 1e6)
                 for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])]
        return "\n".join(lines)

    def summary(self):
        torch.cuda.synchronize()
        ms = sum(e0.elapsed_time(e1) for e0, e1, _ in self.records)
        flops = sum(f for _, _, f in self.records)
        return len(self.records), ms, flops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch (scripts/train_autoprog.sh: -b 128)")
    ap.add_argument("--res", type=int, default=224)
    ap.add_argument("--variant", default="volo_h12_l18")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-optimizer", action="store_true", help="time forward+loss+backward only")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher (reference: distributed_train_prog.sh:4 starts its 8 ranks the same
        # way).  Nothing in this process has touched the GPU yet; the ranks are CHILD processes (never an exec of this one).
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node N)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # AP_DIST_BACKEND=gloo (testing only): lets several ranks share one GPU to exercise the multi-rank code path where
    # RCCL cannot run (it refuses two ranks on one device); the driver's runs use the default, nccl (= RCCL on ROCm)
    backend = os.environ.get("AP_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # reference main_prog.py:65 sets cudnn.benchmark = True: let MIOpen MEASURE its conv solvers for the stem
    # (its immediate-mode heuristics can pick solvers that are 100x slower on a fresh machine)
    torch.backends.cudnn.benchmark = True
    import torch.distributed as dist
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from autoprog_amd.models import create_model
    from autoprog_amd.loss import TokenLabelCrossEntropy
    from autoprog_amd.dist import GradientBucketReducer

    torch.manual_seed(42 + rank)
    np.random.seed(42 + rank)
    model = create_model("model_variant", variant=args.variant, drop_path_rate=0.1).to(dev).train()
    if world > 1:                      # identical initial weights on every rank (DDP broadcast)
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=0)
    loss_fn = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=1000)
    reducer = GradientBucketReducer(list(model.parameters()), world_size=world)
    reducer.install_sink(model)
    from autoprog_amd.optim import FlatAdamWEma
    ema_decays = [0.998, 0.9986, 0.999, 0.9996]            # scripts/train_autoprog.sh:5
    opt = FlatAdamWEma(model, reducer, lr=1.6e-3, weight_decay=0.05, ema_decays=ema_decays)   # one fused kernel per step

    B, res = args.batch, args.res
    gen = torch.Generator().manual_seed(42 + rank)
    images = torch.randn(B, 3, res, res, generator=gen).to(dev)
    n_tok = (res // 16) ** 2
    target = make_target(B, 1000, n_tok, dev, gen)

    def step():
        reducer.zero_grad()
        out = model(images)
        loss = loss_fn(out, target)
        loss.backward()
        reducer.finish()
        if not args.no_optimizer:
            opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    final_loss = float(loss.detach())

    # SURVEY.md section 8(d) M1 asks for the optimizer + EMA share separately: time a few fwd+loss+bwd-only steps as well
    # (reported under config, never part of `value`)
    fwd_bwd_ms = None
    if not args.no_optimizer:
        def step_nb():
            reducer.zero_grad()
            loss_fn(model(images), target).backward()
            reducer.finish()
        n_extra = max(1, min(5, args.steps))
        step_nb()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_extra):
            step_nb()
        torch.cuda.synchronize()
        fwd_bwd_ms = (time.perf_counter() - t1) / n_extra * 1e3

    roofline = None
    if not args.no_roofline:
        probe = GemmProbe()
        probe.install()
        nprobe = 3
        for _ in range(nprobe):
            step()
        launches, ms, flops = probe.summary()
        if rank == 0 and os.environ.get("AP_GEMM_TABLE") == "1":
            print(probe.table(), file=sys.stderr, flush=True)
        probe.remove()
        achieved = flops / (ms * 1e-3) / 1e12
        # HBM bytes per launch come from a separate rocprofv3 --pmc pass (tools/collect_profiles.sh writes the file below
        # from FETCH_SIZE / WRITE_SIZE with the gfx950 correction of MI355X_MICROARCH.md); null when that file is absent
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "gemm_nt_traffic.json")) as fh:
                traffic = json.load(fh).get("hbm_bytes_per_launch")
        except (OSError, ValueError):
            pass
        roofline = {"bound": "mfma", "kernel": "k_gemm_nt", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                    "algorithmic_bytes_per_launch": round(probe.bytes / max(launches, 1)),
                    "launches_per_step": launches // nprobe, "avg_launch_us": round(ms * 1e3 / launches, 2),
                    "gemm_ms_per_step": round(ms / nprobe, 3), "gemm_gflop_per_step": round(flops / nprobe / 1e9, 1)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        threads = min(os.cpu_count() or 1, 64)
        cpu = cpu_baseline(args.variant, res, args.cpu_seconds, threads)

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = B * world * args.steps / elapsed
        line = {"metric": "images/sec/GPU (fwd+bwd) VOLO-D1 224px AutoProg step", "value": round(value, 2), "unit": "images/sec",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                "images_per_sec_per_gpu": round(value / world, 2),
                "config": {"workload": "BASELINE.json configs[1]: %s (VOLO-D1) %dpx token-label training step" % (args.variant, res),
                           "model": args.variant, "global_batch": B * world, "per_gpu_batch": B, "res": res, "parallelism": "dp%d" % world,
                           "step": "fwd+loss+bwd" + ("" if args.no_optimizer else "+AdamW+4xEMA") + ("+RCCL grad all-reduce" if world > 1 else ""),
                           "final_loss": round(final_loss, 4),
                           "fwd_loss_bwd_only_ms_per_step": None if fwd_bwd_ms is None else round(fwd_bwd_ms, 3)},
                "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
