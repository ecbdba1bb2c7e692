#!/usr/bin/env python3
"""Benchmark of the AutoProg VOLO-D1 training step on MI355X (BASELINE.json metric
"images/sec/GPU (fwd+bwd) VOLO-D1 224px AutoProg step").

  python bench.py --gpus N --steps K --warmup W
      N > 1 without a launcher: bench.py starts `torch.distributed.run --nproc-per-node N` itself (child process; the
      reference starts its ranks the same way, distributed_train_prog.sh:4).  Under a launcher (WORLD_SIZE set) it is a rank.

A "step" is one pass of the hot path over one synthetic batch already resident in HBM: forward (mix-token, DropPath) +
token-label loss + backward (+ bucketed RCCL gradient all-reduce overlapped with backward when N > 1) + fused AdamW +
the 4 EMA updates of scripts/train_autoprog.sh.

Workloads (--workload):
  d1      (default, the driver's line) BASELINE.json configs[1]: volo_h12_l18 (== VOLO-D1), 224 px, per-GPU batch 128, bf16
          activations / fp32 master weights, token-label target [B,1000,198], DropPath 0.1
  stages  BASELINE.json configs[2]: the reference schedule's four stages (l, r) = (9,128) (12,160) (15,192) (18,224) with DropPath
          0 / .033 / .067 / .1 on ONE supernet (elastic depth mask + on-device bilinear resize, main_prog.py:973), a quarter
          of the steps each; --search-mix draws (l, r) uniformly per step instead (supernet search, main_prog.py:1824-1828)
  d5      BASELINE.json configs[4] in bf16: volo_d5 at 448 px (flash MHSA, head_dim 48), default per-GPU batch 16
  deit_base  BASELINE.json configs[3]: the DeiT-Base supernet (deit_h12_l12) under the AutoProg search mix -- one (l, r) per step
          drawn from l in {6, 9, 12} x r in {128, 160, 192, 224} (main_prog.py:1824-1836), soft-target CE, per-GPU batch 128

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     dominant kernel (k_gemm_nt): algorithmic FLOPs and bytes of its launches in one step / their summed duration,
               measured with HIP events on the launch stream; `bound` follows from the launches' arithmetic intensity against
               the ridge (2.5 PFLOP/s / 8 TB/s = 312 FLOP/B), the other roof's fraction is reported beside it
  cpu_baseline the CPU oracle (oracle/ref_cpu.py, kind "port") timed on this host on bounded samples: the VOLO-D1 step at batch 8
               and BASELINE.json configs[0] (DeiT-Tiny, depth-4 sub-net, batch 32, soft-target CE); rank 0 at N=1 only.
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0       # MI355X dense bf16 MFMA peak (/opt/skills/guides/MI355X_MICROARCH.md)
PEAK_FP8_TFLOPS = 5000.0        # ... dense fp8 (e4m3 / e5m2) MFMA peak, same table
PEAK_HBM_TBS = 8.0              # HBM3E peak
STAGES = [(9, 128, 0.0), (12, 160, 0.0333), (15, 192, 0.0667), (18, 224, 0.1)]      # prog/progressive.py:4-31 with scripts/train_autoprog.sh


def default_batch(workload):
    """per-GPU batch of a workload when --batch is not given (128: the reference script's, scripts/train_autoprog.sh:3)"""
    return 64 if workload == "d5" else 128


def kernel_source_hash():
    """sha256 over the kernel sources: profiles/gemm_nt_traffic.json records the hash of the tree its counters were collected on,
    and the bench line carries `roofline.traffic` only while the sources are still those (a stale file reads as null)"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "autoprog_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()


def make_target(B, C, N, device, gen, sparse=False):
    """token-label style soft targets: top-5 (class, score) label maps per slot with label smoothing 0.1 (SURVEY.md section 8(d) row
    M2).  sparse=False: the dense fp32 [B,C,2+N] tensor the reference builds from them on the GPU every step (main_prog.py:994-1004);
    sparse=True: the label maps themselves ([B,2+N,5] indices and scores) -- the loss kernel forms the target rows in registers."""
    idx = torch.randint(0, C, (B, 5, 2 + N), generator=gen)
    val = torch.rand(B, 5, 2 + N, generator=gen)
    val = val / val.sum(1, keepdim=True)
    if sparse:
        from autoprog_amd.loss import SparseTokenLabelTarget
        return SparseTokenLabelTarget(idx.permute(0, 2, 1).contiguous().to(device), val.permute(0, 2, 1).contiguous().to(device), smoothing=0.1)
    t = torch.zeros(B, C, 2 + N)
    t.scatter_(1, idx, val)
    t = t * 0.9 + 0.1 / C
    return t.to(device)


def _time_steps(step, seconds_budget, max_steps=10):
    step()                                   # warm-up
    t0 = time.time()
    n = 0
    while True:
        step()
        n += 1
        if time.time() - t0 > seconds_budget or n >= max_steps:
            break
    return n, time.time() - t0


def cpu_baseline(variant, res, seconds_budget, threads):
    """time the CPU oracle (pure torch fp32 restatement of the reference) on bounded samples"""
    from oracle import ref_cpu as R
    from autoprog_amd.models import create_model
    torch.set_num_threads(threads)
    # ---- the bench workload's twin: VOLO-D1 token-label step at batch 8
    arch = R.variant_arch(variant)
    torch.manual_seed(42)
    model = create_model("model_variant", variant=variant, drop_path_rate=0.1)
    p = {k: v.detach().clone().float().requires_grad_(v.dtype.is_floating_point and "running_" not in k)
         for k, v in model.state_dict().items()}
    B = 8
    g = torch.Generator().manual_seed(42)
    x = torch.randn(B, 3, res, res, generator=g)
    target = make_target(B, 1000, (res // 16) ** 2, "cpu", g)
    rng = np.random.RandomState(42)

    def step():
        lam, box = R.draw_mix_box((B, res // 8, res // 8, arch["embed_dims"][0]), 2, 1.0, rng)
        out = R.volo_forward(p, x, train=True, mix=(lam, box), drop_path_rate=0.1, **arch)
        loss = R.token_label_ce(out, target, 0.5, 1.0)
        torch.autograd.grad(loss, [v for v in p.values() if v.requires_grad], allow_unused=True)

    n_steps, dt = _time_steps(step, seconds_budget * 0.6)
    # ---- BASELINE.json configs[0]: DeiT-Tiny 224, depth-4 sub-net, batch 32, soft-target CE, one process
    torch.manual_seed(42)
    deit = create_model("model_variant", variant="deit_h3_l4")
    pd = {k: v.detach().clone().float().requires_grad_(True) for k, v in deit.state_dict().items()}
    xb = torch.randn(32, 3, 224, 224, generator=g)
    tb = torch.softmax(torch.randn(32, 1000, generator=g) * 3, dim=-1)

    def step_deit():
        loss = R.soft_target_ce(R.vit_forward(pd, xb, depth=4, heads=3), tb)
        torch.autograd.grad(loss, list(pd.values()), allow_unused=True)

    n2, dt2 = _time_steps(step_deit, seconds_budget * 0.4)
    return {"value": round(B * n_steps / dt, 3), "unit": "images/sec", "cores": threads, "host_cpus": os.cpu_count(), "kind": "port",
            "sample": "oracle/ref_cpu.py fp32 fwd+loss+bwd, %s %dpx, batch %d, %d steps in %.1fs (no optimizer)" % (variant, res, B, n_steps, dt),
            "configs0_deit_tiny_l4_b32": {"value": round(32 * n2 / dt2, 2), "unit": "images/sec",
                                          "sample": "oracle vit_forward(deit_h3_l4) fp32 fwd+CE+bwd, 224px, batch 32, %d steps in %.1fs" % (n2, dt2)}}


def cpu_baseline_other(workload, seconds_budget, threads):
    """the CPU oracle on a bounded sample of the d5 / deit_base workloads (reported beside those lines; never part of `value`)"""
    from oracle import ref_cpu as R
    from autoprog_amd.models import create_model
    torch.set_num_threads(threads)
    torch.manual_seed(42)
    g = torch.Generator().manual_seed(42)
    if workload == "d5":
        arch = dict(R.VOLO_PRESETS["volo_d5"])
        model = create_model("volo_d5", img_size=448, drop_path_rate=0.1)
        p = {k: v.detach().clone().float().requires_grad_(v.dtype.is_floating_point and "running_" not in k) for k, v in model.state_dict().items()}
        del model
        B, res = 1, 448
        x = torch.randn(B, 3, res, res, generator=g)
        target = make_target(B, 1000, (res // 16) ** 2, "cpu", g)
        rng = np.random.RandomState(42)

        def step():
            lam, box = R.draw_mix_box((B, res // 8, res // 8, arch["embed_dims"][0]), 2, 1.0, rng)
            out = R.volo_forward(p, x, train=True, mix=(lam, box), drop_path_rate=0.1, **arch)
            torch.autograd.grad(R.token_label_ce(out, target, 0.5, 1.0), [v for v in p.values() if v.requires_grad], allow_unused=True)
        what = "oracle/ref_cpu.py fp32 fwd+loss+bwd, volo_d5 448px, batch 1"
    else:
        model = create_model("model_variant", variant="deit_h12_l12")
        p = {k: v.detach().clone().float().requires_grad_(True) for k, v in model.state_dict().items()}
        del model
        B = 8
        x = torch.randn(B, 3, 224, 224, generator=g)
        t = torch.softmax(torch.randn(B, 1000, generator=g) * 3, dim=-1)

        def step():
            torch.autograd.grad(R.soft_target_ce(R.vit_forward(p, x, depth=12, heads=12), t), list(p.values()), allow_unused=True)
        what = "oracle vit_forward(deit_h12_l12) fp32 fwd+CE+bwd, 224px, batch 8"
    n, dt = _time_steps(step, seconds_budget, max_steps=4)
    return {"value": round(B * n / dt, 3), "unit": "images/sec", "cores": threads, "host_cpus": os.cpu_count(), "kind": "port",
            "sample": "%s, %d steps in %.1fs (no optimizer)" % (what, n, dt)}


class GemmProbe:
    """HIP-event timing of every ap_gemm_nt launch (events recorded on the launch stream)"""

    def __init__(self):
        self.records = []
        self.keys = []
        self.bytes = 0.0
        self.fp8_launches = 0
        self.fp8_flops = 0.0
        self.fused_launches = 0
        self.fused_hidden_bytes = 0.0

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
            probe.keys.append((a.shape[0], nn, kk, "+".join(k2 for k2 in ("bias", "gelu", "dgelu_of", "mul_by", "row_scale", "residual") if kw.get(k2) is not None and kw.get(k2) is not False) or "plain"))
            # algorithmic bytes: A + B + C once, plus the epilogue operands this call reads / writes
            # (an epilogue tensor counts its own element size: the 8-bit gelu' codes of round 5 are one byte per element, written by the
            # GELU launch as preact_out and read by the backward launch as mul_by)
            extra = sum(float(kw[k2].element_size()) for k2 in ("residual", "dgelu_of", "preact_out", "mul_by") if kw.get(k2) is not None)
            probe.bytes += 2.0 * (a.shape[0] * kk + nn * kk + a.shape[0] * nn) + extra * a.shape[0] * nn
            return out
        ops.gemm_nt = timed
        # the e4m3 launches of --fp8 (ops.gemm_nt_fp8: one byte per operand element, bf16 outputs) belong to the same kernel family
        self._orig8 = ops.gemm_nt_fp8

        def timed8(a8, b8, dq_a, dq_b, n=None, **kw):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            out = probe._orig8(a8, b8, dq_a, dq_b, n=n, **kw)
            e1.record()
            nn = b8.shape[0] if n is None else n
            kk = a8.shape[1]
            probe.records.append((e0, e1, 2.0 * a8.shape[0] * nn * kk))
            probe.keys.append((a8.shape[0], nn, kk, "fp8:" + ("+".join(k2 for k2 in ("bias", "gelu", "row_scale", "residual", "q8") if kw.get(k2) is not None and kw.get(k2) is not False) or "plain")))
            extra = sum(1 for k2 in ("residual", "preact_out") if kw.get(k2) is not None)
            probe.bytes += 1.0 * (a8.shape[0] * kk + nn * kk) + 2.0 * a8.shape[0] * nn * (1 + extra) + (1.0 * a8.shape[0] * nn if kw.get("q8") is not None else 0.0)
            probe.fp8_launches += 1
            probe.fp8_flops += 2.0 * a8.shape[0] * nn * kk
            return out
        ops.gemm_nt_fp8 = timed8
        # the fused MLP launches (ops.mlp_fused: fc1 -> GELU -> fc2 and its backward mirror, csrc/mlp_fused.hip) replace two ap_gemm_nt
        # launches each and stay in the family: the FLOPs of both products, the bytes of every operand once
        self._origm = ops.mlp_fused

        def timedm(x, wa, wb, backward=False, **kw):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            out = probe._origm(x, wa, wb, backward=backward, **kw)
            e1.record()
            if out is None:                # refused (shape / capture): the caller falls back to the two launches, which are counted there
                return out
            rows = x if x is not None else kw["ln"][0]            # (ln=: the LayerNorm in front of fc1 runs inside the launch, on these rows)
            M, C, Hd = rows.shape[0], rows.shape[1], wa.shape[0]
            probe.records.append((e0, e1, 4.0 * M * C * Hd))
            probe.keys.append((M, Hd, C, "fused-mlp-bwd" if backward else "fused-mlp-fwd"))
            # x + both weights + out (+ residual forward), the hidden tensor in bf16 (forward: fc2's operand for its weight gradient;
            # backward: fc1's output gradient) and the one-byte gelu' codes (written forward, read backward)
            probe.bytes += 2.0 * (M * C * (2 if backward or kw.get("residual") is None else 3) + 2 * C * Hd + M * Hd) + 1.0 * M * Hd
            if kw.get("ln") is not None:       # the residual rows ARE the LayerNorm's input (read once); the normalised rows leave for the weight gradient
                probe.bytes += 2.0 * M * C - (2.0 * M * C if kw.get("residual") is not None and kw["residual"].data_ptr() == rows.data_ptr() else 0.0) + 2.0 * M * C
            probe.fused_launches += 1
            probe.fused_hidden_bytes += 2.0 * M * Hd          # what the second of the two replaced launches would have read back
            return out
        ops.mlp_fused = timedm

    def remove(self):
        from autoprog_amd import ops
        ops.gemm_nt = self._orig
        ops.gemm_nt_fp8 = self._orig8
        ops.mlp_fused = self._origm

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


def calibrate(dev):
    """what THIS device delivers on three fixed probes, measured in-process in front of the timed region (HIP events on the launch stream,
    medians): the boxes of the pool differ by +-2-4 % in step time with identical code, so round-over-round changes are quoted against
    these.  (a) a float4 copy of 512 MiB (HBM; MI355X_MICROARCH.md quotes 6.29 TB/s for it), (b) a register-only
    v_mfma_f32_16x16x32_bf16 loop on random operands, one wave per SIMD on all 256 CUs (the matrix pipe at the clock the chip holds),
    (c) the plain 25088 x 384 x 1152 product of the step (qkv of a transformer block) over a rotating set of operands (> 256 MiB)."""
    from autoprog_amd import ops

    def timed(fn, reps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        ev[0].record()
        for i in range(reps):
            fn(i)
            ev[i + 1].record()
        torch.cuda.synchronize()
        ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
        return ts[len(ts) // 2], ts[0]

    t0 = time.perf_counter()
    g = torch.Generator().manual_seed(7)
    n = 512 << 20
    src = torch.empty(n // 4, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    ops.calib_copy(src, dst)
    copy_med, copy_min = timed(lambda i: ops.calib_copy(src, dst), 7)
    del src, dst
    seed = torch.randn(2048, generator=g).to(dev).to(torch.bfloat16)
    sink = torch.zeros(4, dtype=torch.float32, device=dev)
    iters = 4096
    for _ in range(3):
        flop = ops.calib_mfma(seed, sink, iters)
    mfma_med, mfma_min = timed(lambda i: ops.calib_mfma(seed, sink, iters), 7)
    M, N, K, R = 25088, 1152, 384, 6
    a = [(torch.randn(M, K, generator=g) * 0.5).to(dev).to(torch.bfloat16) for _ in range(2)]
    a = [a[i % 2].clone() for i in range(R)]
    w = (torch.randn(N, K, generator=g) * 0.05).to(dev).to(torch.bfloat16)
    outs = [torch.empty(M, N, dtype=torch.bfloat16, device=dev) for _ in range(R)]      # 6 x (19 + 58) MB: every launch starts cold
    for i in range(R):
        ops.gemm_nt(a[i], w, out=outs[i])
    gemm_med, gemm_min = timed(lambda i: ops.gemm_nt(a[i % R], w, out=outs[i % R]), 18)
    torch.cuda.synchronize()
    return {"float4_copy_tbs": round(2.0 * n / (copy_med * 1e-3) / 1e12, 3), "float4_copy_tbs_best": round(2.0 * n / (copy_min * 1e-3) / 1e12, 3),
            "mfma_loop_tflops": round(flop / (mfma_med * 1e-3) / 1e12, 1), "mfma_loop_tflops_best": round(flop / (mfma_min * 1e-3) / 1e12, 1),
            "gemm_25088x384x1152_plain_us": round(gemm_med * 1e3, 2), "gemm_25088x384x1152_plain_us_best": round(gemm_min * 1e3, 2),
            "seconds": round(time.perf_counter() - t0, 2),
            "how": "HIP events on the launch stream, medians (best beside them): 512 MiB float4 copy x7; 256 CUs x 4 waves x 4096 trips of 16 "
                   "v_mfma_f32_16x16x32_bf16 on random operands x7; ap_gemm_nt 25088x1152x384 plain over 6 rotating operand sets x18"}


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


def prewarm(step, seconds, sync, all_max=None):
    """>= `seconds` of step() in groups of eight -> (steps, seconds).  With more than one rank the steps contain collectives, so EVERY rank has
    to stop after the same group: the clock that decides is the maximum over the ranks (all_max), not the rank's own -- a rank that left the
    loop a group early would leave the others waiting in an all-reduce nobody answers (tests/test_dist_gloo.py runs two ranks of unequal speed)."""
    t0, n = time.perf_counter(), 0
    while True:
        for _ in range(8):
            step()
        n += 8
        sync()
        el = time.perf_counter() - t0
        if all_max is not None:
            el = all_max(el)
        if el >= seconds:
            return n, el


def all_ranks_max(value, device):
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def launch_check(world, rank):
    """--launch-check: the rendezvous + gradient-exchange plumbing of an N-rank run without the HIP path (CPU, gloo): every rank
    reduces a slab through GradientBucketReducer and rank 0 prints what it saw.  The product path itself has no CPU mode."""
    import torch.distributed as dist
    from autoprog_amd.dist import GradientBucketReducer
    dist.init_process_group("gloo")
    params = [torch.nn.Parameter(torch.zeros(1000)), torch.nn.Parameter(torch.zeros(37, 5))]
    red = GradientBucketReducer(params, bucket_bytes=1024, world_size=world, defer_mean=True)
    red.zero_grad()
    for p in params:
        p.grad.add_(float(rank + 1))
        red._on_grad(p)
    red.finish()
    mean = float(params[0].grad[0]) * red.take_pending_scale()
    ok = abs(mean - (world + 1) / 2.0) < 1e-6
    if rank == 0:
        print(json.dumps({"launch_check": "ok" if ok else "FAILED", "rccl_ranks": dist.get_world_size(), "backend": "gloo", "grad_mean": mean}), flush=True)
    dist.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch (scripts/train_autoprog.sh: -b 128)")
    ap.add_argument("--res", type=int, default=224)
    ap.add_argument("--variant", default="volo_h12_l18")
    ap.add_argument("--workload", default="d1", choices=["d1", "stages", "d5", "deit_base"],
                    help="d5: BASELINE configs[4] in bf16 -- VOLO-D5 at 448 px (use --batch 8..16); not the default line")
    ap.add_argument("--fp8", action="store_true",
                    help="d5 workload: the forward Linear GEMMs of the transformer blocks on e4m3 operands (configs[4] 'mixed MFMA fp8 GEMM'); "
                         "delayed per-tensor scaling, bf16 backward.  Never the default line: the headline metric is bf16")
    ap.add_argument("--search-mix", action="store_true", help="stages workload: uniform random (l, r) per step (supernet search)")
    ap.add_argument("--dense-target", action="store_true",
                    help="feed the token-label target as the dense fp32 [B,1000,2+N] tensor (default: the top-5 label maps it is built from; the CE "
                         "kernel forms the same target rows in registers, main_prog.py:994-1004 folded into the loss)")
    ap.add_argument("--graph", action="store_true",
                    help="d1 / stages workloads, one rank: replay each step from a HIP graph (autoprog_amd/graph.py: the mix-token box, lam, lr and "
                         "Adam's bias corrections live in device memory and are refreshed in front of every replay).  The early AutoProg stages are "
                         "launch-gap bound; never the default line")
    ap.add_argument("--stage-blocks", action="store_true",
                    help="stages workload: run the four stages in consecutive blocks of steps/4 (the schedule's order: a stage lasts 25 epochs) instead "
                         "of one stage per step in turn -- in turn, the host enqueues the short stage-1 step while the GPU still runs the long "
                         "stage-4 step before it, which hides the launch gaps a real stage-1 epoch has")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=24.0)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-optimizer", action="store_true", help="time forward+loss+backward only")
    ap.add_argument("--launch-check", action="store_true", help="rendezvous + gradient exchange only (CPU/gloo), no GPU work")
    ap.add_argument("--prewarm-s", type=float, default=2.0,
                    help="seconds of the SAME step run in front of the counted --warmup steps (reported as prewarm_s): a fresh lease starts "
                         "with cold clocks and allocator pools, and 5 warm-up steps are 60 ms; never part of the timed region")
    ap.add_argument("--no-calibration", action="store_true", help="skip the in-process box calibration (copy / MFMA loop / one GEMM)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing in this process has touched the GPU yet; the ranks are
        # CHILD processes (never an exec of this one).
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node N)" % (args.gpus, world))
    if args.launch_check:
        sys.exit(launch_check(world, rank))
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
    DEIT_L, DEIT_R = [6, 9, 12], [128, 160, 192, 224]
    if args.workload == "deit_base":
        args.variant, args.res = "deit_h12_l12", 224
        model = create_model("model_variant", variant="deit_h12_l12", drop_path_rate=0.1).to(dev).train()
    elif args.workload == "d5":
        if args.fp8:
            from autoprog_amd import functional as AF
            AF.FP8_LINEAR = True
        args.variant, args.res = "volo_d5", 448
        if args.batch == 128:
            args.batch = default_batch("d5")          # sized for 288 GB of HBM: 16 images (rounds 2 - 3) leave the D5 GEMMs at 147 - 588 tiles for 256 CUs;
                                     # measured 268 / 297 / 320 / 333 images/s at batch 16 / 32 / 64 / 96 (profiles/r04_d5_batch_scaling.txt)
        model = create_model("volo_d5", img_size=448, drop_path_rate=0.1).to(dev).train()
    else:
        model = create_model("model_variant", variant=args.variant, drop_path_rate=0.1).to(dev).train()
    if world > 1:                      # identical initial weights on every rank (DDP broadcast)
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=0)
    loss_fn = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=1000)
    if args.workload == "deit_base":
        from autoprog_amd.loss import SoftTargetCrossEntropy
        loss_fn = SoftTargetCrossEntropy()
    # the 1/world of the gradient mean is folded into the fused optimizer kernel (no extra pass over the 106 MB slab)
    reducer = GradientBucketReducer(list(model.parameters()), world_size=world, defer_mean=not args.no_optimizer)
    reducer.install_sink(model)
    from autoprog_amd.optim import FlatAdamWEma
    ema_decays = [0.998, 0.9986, 0.999, 0.9996]            # scripts/train_autoprog.sh:5
    opt = FlatAdamWEma(model, reducer, lr=1.6e-3, weight_decay=0.05, ema_decays=ema_decays)   # one fused kernel per step

    B, res = args.batch, args.res
    gen = torch.Generator().manual_seed(42 + rank)
    images = torch.randn(B, 3, res, res, generator=gen).to(dev)
    if args.workload == "deit_base":
        target = torch.softmax(torch.randn(B, 1000, generator=gen) * 3, dim=-1).to(dev)
        random.seed(0)                 # identical (l, r) sequence on every rank

        def step():
            l, r = random.choice(DEIT_L), random.choice(DEIT_R)
            model.set_sample_config(dict(layer_num=l, min_layer_num=DEIT_L[0], max_layer_num=DEIT_L[-1]))
            reducer.zero_grad()
            # the reference resizes the loader's batch to the step's resolution inside the step (main_prog.py:973-974)
            xr = images if r == res else torch.nn.functional.interpolate(images, size=(r, r), mode="bilinear", align_corners=False)
            loss = loss_fn(model(xr), target)
            loss.backward()
            reducer.finish()
            if not args.no_optimizer:
                opt.step()
            return loss
    elif args.workload == "stages":
        targets = {r: make_target(B, 1000, (r // 16) ** 2, dev, gen, sparse=not args.dense_target) for _, r, _ in STAGES}
        l_list, r_list = [s[0] for s in STAGES], [s[1] for s in STAGES]
        random.seed(0)                 # identical (l, r) sequence on every rank (reference: random.seed(epoch), main_prog.py:1861)
        counter = [0]

        def stage_of(i):
            if not args.stage_blocks:
                return i % len(STAGES)
            i -= args.warmup                                   # the timed steps: steps / 4 consecutive steps per stage (warm-up: stage 0)
            per = max(1, args.steps // len(STAGES))
            return min(max(i, 0) // per, len(STAGES) - 1) if i < args.steps else (i - args.steps) % len(STAGES)

        def step():
            i = counter[0]
            counter[0] += 1
            if args.search_mix:
                l, r, dp = random.choice(l_list), random.choice(r_list), 0.1        # search epochs use the final strengths (main_prog.py:814-815)
            else:
                l, r, dp = STAGES[stage_of(i)]
            model.set_sample_config(dict(layer_num=l, min_layer_num=l_list[0], max_layer_num=l_list[-1], input_size=r))
            model.set_drop_path_rate(dp)
            reducer.zero_grad()
            loss = loss_fn(model(images), targets[r])       # the 224-px batch is resized to r by the stem's first kernel
            loss.backward()
            reducer.finish()
            if not args.no_optimizer:
                opt.step()
            return loss
        images_per_step = B
    else:
        n_tok = (res // 16) ** 2
        target = make_target(B, 1000, n_tok, dev, gen, sparse=not args.dense_target)

        def step():
            reducer.zero_grad()
            out = model(images)
            loss = loss_fn(out, target)
            loss.backward()
            reducer.finish()
            if not args.no_optimizer:
                opt.step()
            return loss

    eager_step = step
    if args.graph:
        if world > 1 or args.no_optimizer or args.workload not in ("d1", "stages"):
            raise SystemExit("--graph: one rank, d1 or stages, with the optimizer")
        from autoprog_amd.graph import GraphedStep
        if args.workload == "stages" and args.search_mix:
            # the supernet search (main_prog.py:1824-1837): a different (l, r) every step -- one graph per candidate, 16 here (the driver's
            # searches have at most 9), captured up front; the step draws its candidate as the eager search step does
            graphs = {}
            for l_ in l_list:
                for r_ in r_list:
                    model.set_sample_config(dict(layer_num=l_, min_layer_num=l_list[0], max_layer_num=l_list[-1], input_size=r_))
                    model.set_drop_path_rate(0.1)
                    graphs[(l_, r_)] = GraphedStep(model, loss_fn, reducer, opt, images, targets[r_]).capture()
            gcount = [0]

            def step():
                gcount[0] += 1
                return graphs[(random.choice(l_list), random.choice(r_list))].step()
        elif args.workload == "stages":
            graphs = []
            for l_, r_, dp_ in STAGES:       # one graph per elastic configuration: set_sample_config decides which kernels a step launches
                model.set_sample_config(dict(layer_num=l_, min_layer_num=STAGES[0][0], max_layer_num=STAGES[-1][0], input_size=r_))
                model.set_drop_path_rate(dp_)
                graphs.append(GraphedStep(model, loss_fn, reducer, opt, images, targets[r_]).capture())
            gcount = [0]

            def step():
                g = graphs[stage_of(gcount[0])]
                gcount[0] += 1
                return g.step()
        else:
            graph1 = GraphedStep(model, loss_fn, reducer, opt, images, target).capture()

            def step():
                return graph1.step()
    elif args.workload == "stages":
        # MIOpen measures its conv solvers the first time it sees a shape (cudnn.benchmark): touch every resolution once, untimed
        for l_, r_, dp_ in STAGES:
            model.set_sample_config(dict(layer_num=l_, min_layer_num=STAGES[0][0], max_layer_num=STAGES[-1][0], input_size=r_))
            model.set_drop_path_rate(dp_)
            reducer.zero_grad()
            loss_fn(model(images), targets[r_]).backward()
            reducer.finish()
            reducer.take_pending_scale()
    # ---- disclosed pre-warm: >= --prewarm-s seconds of the same step (untimed, reported), then the box calibration, then the counted warm-up
    prewarm_steps, prewarm_s = 0, 0.0
    if args.prewarm_s > 0:
        prewarm_steps, prewarm_s = prewarm(step, args.prewarm_s, torch.cuda.synchronize,
                                           (lambda t: all_ranks_max(t, dev)) if world > 1 else None)
        if args.workload == "stages":
            counter[0] = 0                       # (the stage walk of the timed region starts where it always did)
            if args.graph:
                gcount[0] = 0
    calibration = None
    if rank == 0 and not args.no_calibration:
        calibration = calibrate(dev)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # per-step HIP-event times beside the wall clock (events on the stream the step launches on: one record per step)
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    step_ev[0].record()
    for i_ in range(args.steps):
        loss = step()
        step_ev[i_ + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    step_ms = sorted(step_ev[i_].elapsed_time(step_ev[i_ + 1]) for i_ in range(args.steps))
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    final_loss = float(loss.detach())

    # SURVEY.md section 8(d) M1 asks for the optimizer + EMA share separately: time a few fwd+loss+bwd-only steps as well
    # (reported under config, never part of `value`)
    fwd_bwd_ms = None
    allreduce_ms = None
    if not args.no_optimizer and args.workload in ("d1", "d5"):        # (the other workloads change (l, r) per step)
        def step_nb():
            reducer.zero_grad()
            loss_fn(model(images), target).backward()
            reducer.finish()
            reducer.take_pending_scale()
        n_extra = max(1, min(5, args.steps))
        step_nb()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_extra):
            step_nb()
        torch.cuda.synchronize()
        fwd_bwd_ms = (time.perf_counter() - t1) / n_extra * 1e3
    if world > 1:
        # the gradient exchange on its own (not overlapped with anything): what backward has to hide
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        for _ in range(5):
            for s_, e_, _ in reducer.buckets:
                dist.all_reduce(reducer.flat[s_:e_])
        torch.cuda.synchronize()
        allreduce_ms = (time.perf_counter() - t1) / 5 * 1e3

    roofline = None
    if not args.no_roofline:
        probe = GemmProbe()
        probe.install()
        nprobe = 3 if args.workload == "d1" else 4
        if args.graph:                 # (the probe wraps the Python launch functions: the eager step of the same configuration sequence)
            for g_ in ((graphs.values() if isinstance(graphs, dict) else graphs) if args.workload == "stages" else [graph1]):
                g_.release()
            if args.workload == "stages":
                counter[0] = args.warmup + args.steps          # the probe steps take the stages in turn
        for _ in range(nprobe):
            eager_step()
        launches, ms, flops = probe.summary()
        if rank == 0 and os.environ.get("AP_GEMM_TABLE") == "1":
            print(probe.table(), file=sys.stderr, flush=True)
        probe.remove()
        tflops = flops / (ms * 1e-3) / 1e12
        tbs = probe.bytes / (ms * 1e-3) / 1e12
        # the MFMA roof of this launch set: an e4m3 launch (v_mfma_scale_f32_16x16x128_f8f6f4) is priced against the 5 PFLOP/s dense fp8
        # peak, a bf16 launch against 2.5 -- the time the set would take at its peaks, turned back into a rate
        peak_mfma = flops / ((flops - probe.fp8_flops) / PEAK_BF16_TFLOPS + probe.fp8_flops / PEAK_FP8_TFLOPS) if flops else PEAK_BF16_TFLOPS
        ai = flops / probe.bytes                      # FLOP per algorithmic byte, averaged over the launches of a step
        ridge = peak_mfma / PEAK_HBM_TBS              # 312.5 FLOP/B for a bf16 launch set
        # HBM bytes per launch come from a separate rocprofv3 --pmc pass (tools/collect_profiles.sh writes the file below
        # from FETCH_SIZE / WRITE_SIZE with the gfx950 correction of MI355X_MICROARCH.md); null when that file is absent
        traffic = None
        try:
            tname = "gemm_nt_traffic.json" if args.workload == "d1" else "gemm_nt_traffic_%s.json" % args.workload      # tools/collect_traffic.sh
            with open(os.path.join(ROOT, "profiles", tname)) as fh:
                tj = json.load(fh)
            if tj.get("src_sha256") == kernel_source_hash() and not args.fp8 and tj.get("per_gpu_batch", default_batch(args.workload)) == B:
                traffic = tj.get("hbm_bytes_per_launch")
        except (OSError, ValueError):
            pass
        hbm_bound = ai < ridge
        roofline = {"bound": "hbm" if hbm_bound else "mfma", "kernel": "ap_gemm_nt + ap_mlp_fused launches (k_gemm_nt_8p<...> + k_gemm_nt_ws<...> + k_gemm_nt<...> + k_mlp_fused2<...>)",
                    "achieved": round(tbs * 1e3 if hbm_bound else tflops, 2), "peak": PEAK_HBM_TBS * 1e3 if hbm_bound else round(peak_mfma, 1),
                    "unit": "GB/s" if hbm_bound else "TFLOP/s",
                    "frac": round(tbs / PEAK_HBM_TBS if hbm_bound else tflops / peak_mfma, 4), "traffic": traffic,
                    "arithmetic_intensity_flop_per_byte": round(ai, 1), "ridge_flop_per_byte": round(ridge, 1),
                    "mfma_tflops": round(tflops, 2), "mfma_frac": round(tflops / peak_mfma, 4), "mfma_peak_tflops": round(peak_mfma, 1),
                    "fp8_flop_share": round(probe.fp8_flops / flops, 4) if flops else 0.0,
                    "hbm_tbs_algorithmic": round(tbs, 3), "hbm_frac": round(tbs / PEAK_HBM_TBS, 4),
                    # for comparison with rounds 1 - 5, whose family had no fused launches: the same time priced against the bytes of the launches
                    # the fused ones replace (+ the hidden tensor's read-back per fused launch)
                    "hbm_frac_two_launch_accounting": round((probe.bytes + probe.fused_hidden_bytes) / (ms * 1e-3) / 1e12 / PEAK_HBM_TBS, 4),
                    "algorithmic_bytes_per_launch": round(probe.bytes / max(launches, 1)),
                    "launches_per_step": launches // nprobe, "fp8_launches_per_step": probe.fp8_launches // nprobe,
                    "fused_mlp_launches_per_step": probe.fused_launches // nprobe, "avg_launch_us": round(ms * 1e3 / launches, 2),
                    "gemm_ms_per_step": round(ms / nprobe, 3), "gemm_gflop_per_step": round(flops / nprobe / 1e9, 1)}

    cpu = None
    # (the CPU twin is the VOLO-D1 token-label step of the default workload; the other workloads report no CPU baseline)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload in ("d1", "stages") and args.variant.startswith("volo_h"):
        threads = min(os.cpu_count() or 1, 64)
        cpu = cpu_baseline(args.variant, res, args.cpu_seconds, threads)
    elif rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload in ("d5", "deit_base"):
        cpu = cpu_baseline_other(args.workload, args.cpu_seconds, min(os.cpu_count() or 1, 64))

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = B * world * args.steps / elapsed
        if args.workload == "deit_base":
            wl = ("BASELINE.json configs[3]: deit_h12_l12 (DeiT-Base) supernet, AutoProg search mix -- uniform random l in %s x r in %s per step, "
                  "soft-target CE, batch %d, on-device resize from %d px" % (DEIT_L, DEIT_R, B, res))
        elif args.workload == "stages":
            wl = ("BASELINE.json configs[2]: %s supernet over the AutoProg stages (l,r) = %s, %s, batch %d, on-device resize from %d px"
                  % (args.variant, [(s[0], s[1]) for s in STAGES], "uniform random (l,r) per step" if args.search_mix else
                     ("a quarter of the steps each, in consecutive blocks" if args.stage_blocks else "a quarter of the steps each, one stage per step in turn"), B, res))
        else:
            wl = (("BASELINE.json configs[4]: volo_d5 448px token-label training step, batch %d; " % B) +
                  ("forward Linear GEMMs of the transformer blocks on e4m3 operands (fp32 accumulate, delayed per-tensor scaling), everything else bf16"
                   if args.fp8 else "in bf16 (no fp8)") if args.workload == "d5"
                  else "BASELINE.json configs[1]: %s (VOLO-D1) %dpx token-label training step" % (args.variant, res))
        metric = ("images/sec/GPU (fwd+bwd) VOLO-D5 448px token-label step" if args.workload == "d5"
                  else "images/sec/GPU (fwd+bwd) DeiT-Base AutoProg search-mix step" if args.workload == "deit_base"
                  else "images/sec/GPU (fwd+bwd) VOLO-D1 224px AutoProg step")
        line = {"metric": metric, "value": round(value, 2), "unit": "images/sec",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
                "ms_per_step_min": round(step_ms[0], 3), "ms_per_step_median": round(step_ms[len(step_ms) // 2], 3), "ms_per_step_max": round(step_ms[-1], 3),
                "ms_per_step_how": "value / ms_per_step: wall clock over the timed steps; min / median / max: HIP events recorded behind every step on the launch stream (rank 0)",
                "prewarm_s": round(prewarm_s, 2), "prewarm_steps": prewarm_steps, "calibration": calibration,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "fp8 (e4m3 forward GEMMs) / bf16" if (args.fp8 and args.workload == "d5") else "bf16", "data": "synthetic",
                "value_semantics": "whole-job aggregate over n_gpus (per-GPU rate in images_per_sec_per_gpu)",
                "images_per_sec_per_gpu": round(value / world, 2),
                "rccl_ranks": (dist.get_world_size() if world > 1 else 1), "allreduce_ms_standalone": None if allreduce_ms is None else round(allreduce_ms, 3),
                "config": {"workload": wl,
                           "model": args.variant, "global_batch": B * world, "per_gpu_batch": B, "res": res, "parallelism": "dp%d" % world,
                           "step": "fwd+loss+bwd" + ("" if args.no_optimizer else "+AdamW+4xEMA") + ("+RCCL grad all-reduce" if world > 1 else "") +
                                   (" (each step replayed from a HIP graph)" if args.graph else ""),
                           "target": ("soft-target [B,1000]" if args.workload == "deit_base" else
                                      "dense fp32 [B,1000,2+N]" if args.dense_target else
                                      "top-5 label maps [B,2+N,5] + smoothing 0.1, densified inside the CE kernel (== the dense tensor of main_prog.py:994-1004)"),
                           "final_loss": round(final_loss, 4),
                           "fwd_loss_bwd_only_ms_per_step": None if fwd_bwd_ms is None else round(fwd_bwd_ms, 3)},
                "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
