"""FlatAdamWEma as the drop-in for create_optimizer + the ModelEma list (main_prog.py:484,507-514; prog/checkpoint_saver.py:110-130):
EMA checkpoints incl. BatchNorm buffers, refresh of the bf16 weight copies after load_state_dict, LR-scheduler-visible
param_groups, optimizer state_dict round trip, and the deferred data-parallel mean folded into the update kernel."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tiny():
    from autoprog_amd.models import create_model
    torch.manual_seed(0)
    return create_model("model_variant", variant="volo_h2_l3", num_classes=16, img_size=64, stem_hidden_dim=16).cuda().train()


def _step(model, red, opt, loss_fn, x, target):
    red.zero_grad()
    loss = loss_fn(model(x), target)
    loss.backward()
    red.finish()
    opt.step()
    return float(loss.detach())


def _setup(decays=(0.9, 0.99), **kw):
    from autoprog_amd.dist import GradientBucketReducer
    from autoprog_amd.loss import TokenLabelCrossEntropy
    from autoprog_amd.optim import FlatAdamWEma
    model = _tiny()
    red = GradientBucketReducer(list(model.parameters()), world_size=1, **kw)
    red.install_sink(model)
    opt = FlatAdamWEma(model, red, lr=1e-3, weight_decay=0.05, ema_decays=list(decays))
    g = torch.Generator().manual_seed(1)
    x = torch.randn(4, 3, 64, 64, generator=g).cuda()
    target = torch.softmax(torch.randn(4, 16, 18, generator=g) * 2, dim=1).cuda()
    return model, red, opt, TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=16), x, target


def test_ema_state_dict_loads_strict_and_tracks_batchnorm_buffers():
    model, red, opt, loss_fn, x, target = _setup()
    try:
        bn0 = {n: b.detach().clone() for n, b in model.named_buffers()}
        ema_ref = {n: b.detach().clone() for n, b in model.named_buffers() if b.dtype.is_floating_point}
        np.random.seed(0)
        for _ in range(3):
            _step(model, red, opt, loss_fn, x, target)
            for n, b in model.named_buffers():
                if b.dtype.is_floating_point:
                    ema_ref[n].lerp_(b.detach(), 1.0 - 0.9)
        sd = opt.ema_state_dict(0)
        assert set(sd) == set(model.state_dict())
        for n, b in model.named_buffers():
            if b.dtype.is_floating_point:
                assert torch.allclose(sd[n], ema_ref[n], atol=1e-6), n            # the right statistic under the right name
                assert not torch.equal(sd[n], bn0[n]) or "num_batches" in n
            else:
                assert sd[n].dtype == b.dtype and int(sd[n]) == int(b), n           # num_batches_tracked: the model's value
        twin = copy.deepcopy(model)
        twin.load_state_dict(sd, strict=True)
    finally:
        red.remove()


def test_load_state_dict_refreshes_weight_copies_and_resync():
    """after load_state_dict the parameters live at the same addresses: the forward must NOT keep using the bf16 copies of
    the old weights; FlatAdamWEma.resync() (load post-hook) re-derives them and can restart the EMAs."""
    model, red, opt, loss_fn, x, target = _setup()
    try:
        np.random.seed(0)
        _step(model, red, opt, loss_fn, x, target)
        model.eval()
        with torch.no_grad():
            y0 = model(x).float()
            sd = {k: (v * 0.5 if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in model.state_dict().items()}
            ptr = model.head.weight.data_ptr()
            model.load_state_dict(sd, strict=True)
            assert model.head.weight.data_ptr() == ptr                             # in-place copy into the slab view
            y1 = model(x).float()
            fresh = _tiny().eval()
            fresh.load_state_dict(sd, strict=True)
            y2 = fresh(x).float()
        assert float((y1 - y2).norm() / y2.norm()) < 1e-3                          # the loaded weights are the ones used
        assert float((y1 - y0).norm() / y0.norm()) > 1e-2
        opt.resync(reset_ema=True, reset_moments=True)
        assert torch.equal(opt.ema[0], opt.p) and float(opt.m.abs().sum()) == 0.0 and opt.step_count == 0
    finally:
        red.remove()


def test_param_groups_drive_the_learning_rate_and_state_dict_round_trip():
    model, red, opt, loss_fn, x, target = _setup()
    try:
        assert len(opt.param_groups) == 2 and opt.param_groups[1]["weight_decay"] == 0.0
        names = {id(p): n for n, p in model.named_parameters()}
        assert all(names[id(p)] in ("pos_embed", "cls_token") or p.dim() == 1 or names[id(p)].endswith(".bias") for p in opt.param_groups[1]["params"])
        np.random.seed(0)
        _step(model, red, opt, loss_fn, x, target)
        for g in opt.param_groups:                      # what a timm scheduler does every epoch
            g["lr"] = 0.0
        before = opt.p.clone()
        _step(model, red, opt, loss_fn, x, target)
        wd_only = (before - opt.p).abs().max()
        assert float(wd_only) == 0.0                    # lr 0: neither the Adam step nor the decoupled decay (lr * wd) moves a weight
        for g in opt.param_groups:
            g["lr"] = 1e-3
        saved = copy.deepcopy(opt.state_dict())
        weights = {k: v.clone() for k, v in model.state_dict().items()}
        rng_state = np.random.get_state()
        losses_a = [_step(model, red, opt, loss_fn, x, target) for _ in range(2)]
        # resume: same weights, same optimizer state -> same continuation (fp32 atomics in the weight gradients: to 1e-4)
        model.load_state_dict(weights, strict=True)
        opt.load_state_dict(saved)
        assert opt.step_count == 2
        np.random.set_state(rng_state)                  # same mix-token boxes as the first continuation
        losses_b = [_step(model, red, opt, loss_fn, x, target) for _ in range(2)]
        assert len(saved["state"]) == len(list(model.parameters())) and "exp_avg_sq" in saved["state"][0]
        assert np.allclose(losses_a, losses_b, atol=2e-3), (losses_a, losses_b)
    finally:
        red.remove()


def test_deferred_mean_is_folded_into_the_update():
    """defer_mean: finish() leaves the all-reduced SUM in the slab and the fused kernel applies 1/world -- equal to scaling first"""
    from autoprog_amd.dist import GradientBucketReducer
    from autoprog_amd.optim import FlatAdamWEma
    torch.manual_seed(0)
    nets = [torch.nn.Linear(24, 16).cuda() for _ in range(2)]
    nets[1].load_state_dict(nets[0].state_dict())
    outs = []
    for net, defer in zip(nets, (False, True)):
        red = GradientBucketReducer(list(net.parameters()), world_size=1, defer_mean=defer)
        opt = FlatAdamWEma(net, red, lr=1e-2, weight_decay=0.0)
        red.zero_grad()
        torch.manual_seed(3)
        net(torch.randn(8, 24, device="cuda")).pow(2).mean().backward()
        if defer:
            red._pending_scale = 0.25                  # as finish() sets it for world = 4
        else:
            red.flat.mul_(0.25)
        opt.step()
        outs.append(net.weight.detach().clone())
    assert torch.allclose(outs[0], outs[1], atol=1e-7)


@pytest.mark.parametrize("mode,clip", [("norm", 0.05), ("norm", 1e3), ("value", 0.01)])
def test_gradient_clipping_inside_the_flat_slab_step(mode, clip):
    """FlatAdamWEma.step(clip_grad=, clip_mode=) (missing 5; reference: prog/scaler.py:60-68 -> timm dispatch_clip_grad,
    main_prog.py:129-132,1019-1027) against torch.nn.utils.clip_grad_norm_ / clip_grad_value_ + torch.optim.AdamW on the same gradients,
    three steps; with a DEFERRED data-parallel mean pending (the slab holds the all-reduced SUM of world = 4 ranks) the clipping acts on
    the mean: equal to scaling the slab first.  'norm' with a bound far above the norm is the unclipped update."""
    from autoprog_amd.dist import GradientBucketReducer
    from autoprog_amd.optim import FlatAdamWEma
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(24, 32), torch.nn.GELU(), torch.nn.Linear(32, 16)).cuda()
    ref = copy.deepcopy(net)
    red = GradientBucketReducer(list(net.parameters()), world_size=1, defer_mean=True)
    opt = FlatAdamWEma(net, red, lr=1e-2, weight_decay=0.05)
    dec = [p for n, p in ref.named_parameters() if p.dim() > 1]
    nodec = [p for n, p in ref.named_parameters() if p.dim() <= 1]
    ropt = torch.optim.AdamW([{"params": dec, "weight_decay": 0.05}, {"params": nodec, "weight_decay": 0.0}], lr=1e-2)
    world = 4
    for step in range(3):
        torch.manual_seed(10 + step)
        x = torch.randn(8, 24, device="cuda")
        red.zero_grad()
        (net(x).pow(2).mean() * world).backward()            # the slab as an all-reduce over 4 ranks would leave it: the SUM
        red._pending_scale = 1.0 / world                      # ... with the mean deferred to the update kernel (finish() sets this)
        ropt.zero_grad()
        ref(x).pow(2).mean().backward()
        if mode == "norm":
            total = torch.nn.utils.clip_grad_norm_(ref.parameters(), clip)
        else:
            torch.nn.utils.clip_grad_value_(ref.parameters(), clip)
        ropt.step()
        opt.step(clip_grad=clip, clip_mode=mode)
        if mode == "norm":
            assert abs(float(opt.last_grad_norm) - float(total)) < 1e-5 * max(1.0, float(total)), (float(opt.last_grad_norm), float(total))
        for (n, p), q in zip(net.named_parameters(), ref.parameters()):
            assert torch.allclose(p.detach(), q.detach(), atol=2e-6, rtol=1e-5), (step, n, float((p - q).abs().max()))
    with pytest.raises(NotImplementedError):
        opt.step(clip_grad=0.1, clip_mode="agc")


def test_stage_transition_on_live_slabs():
    """FlatAdamWEma.grow: depth grows 3 -> 5 on ONE volo_h2_l6 supernet.  The active sub-network
    afterwards equals what the reference's route builds (extract the previous stage's EMA state dicts, grow_clone_ema them into a
    5-layer network -- prog/growth.py, pinned bit-exact against the reference's load_slice_clone_ema), the EMA copies follow
    load_slice_clone, the optimizer restarts, BatchNorm statistics are back at their defaults, and the next steps train."""
    from autoprog_amd.dist import GradientBucketReducer
    from autoprog_amd.loss import TokenLabelCrossEntropy
    from autoprog_amd.models import create_model
    from autoprog_amd.optim import FlatAdamWEma
    from autoprog_amd.prog import elastic, growth
    torch.manual_seed(0)
    model = create_model("model_variant", variant="volo_h2_l6", num_classes=16, img_size=64, stem_hidden_dim=16).cuda().train()
    red = GradientBucketReducer(list(model.parameters()), world_size=1)
    red.install_sink(model)
    opt = FlatAdamWEma(model, red, lr=1e-3, weight_decay=0.05, ema_decays=[0.5, 0.6, 0.7, 0.8])
    loss_fn = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=16)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(4, 3, 64, 64, generator=g).cuda()
    target = torch.softmax(torch.randn(4, 16, 18, generator=g) * 2, dim=1).cuda()
    try:
        old_mask = model.set_sample_config(dict(layer_num=3, min_layer_num=3, max_layer_num=6))
        np.random.seed(0)
        for _ in range(3):
            _step(model, red, opt, loss_fn, x, target)
        ema_before = [{k: v.detach().clone() for k, v in opt.ema_state_dict(i).items()} for i in range(4)]
        new_mask = elastic.make_mask(5, 3, 6)
        opt.grow(old_mask, new_mask)
        model.set_sample_config(dict(layer_num=5, min_layer_num=3, max_layer_num=6))
        # reference route on state dicts
        prev = [elastic.export_state_dict(e, old_mask) for e in ema_before]
        shape5 = create_model("model_variant", variant="volo_h2_l5", num_classes=16, img_size=64, stem_hidden_dim=16).state_dict()
        want = growth.grow_clone_ema(shape5, prev[3], prev[:3])
        got = elastic.export_state_dict({k: v.detach() for k, v in model.state_dict().items()}, new_mask)
        assert set(got) == set(want)
        for k, v in want.items():
            if k.rsplit(".", 1)[-1] in ("running_mean", "running_var", "num_batches_tracked"):
                continue
            assert torch.equal(got[k].cpu(), v.cpu()), k
        want_e1 = growth.grow_clone_ema(shape5, prev[1], [prev[1]] * 3)
        got_e1 = elastic.export_state_dict(opt.ema_state_dict(1), new_mask)
        for k, v in want_e1.items():
            if k.rsplit(".", 1)[-1] not in ("running_mean", "running_var", "num_batches_tracked"):
                assert torch.equal(got_e1[k].cpu(), v.cpu()), k
        assert float(opt.m.abs().sum()) == 0.0 and float(opt.v.abs().sum()) == 0.0 and opt.step_count == 0
        bn = model.patch_embed.conv[1]
        assert float(bn.running_mean.abs().sum()) == 0.0 and torch.equal(bn.running_var, torch.ones_like(bn.running_var))
        # the bf16 copies follow the slabs: the forward uses the grown weights, and training goes on
        losses = [_step(model, red, opt, loss_fn, x, target) for _ in range(3)]
        assert all(np.isfinite(losses)) and losses[-1] < losses[0] + 0.5
    finally:
        red.remove()


def test_stage_transition_shrink_takes_the_trained_model():
    """FlatAdamWEma.grow(model_source="model"): the sub-network a search picks out of its supernet (6 -> 4 layers) keeps the weights
    the search epochs TRAINED (reference: load='super', main_prog.py:830-837,1389 load_super(model, prev_model)); EMA copy i comes
    from EMA copy i; the optimizer restarts.  Checked against prog/growth.extract_subnet (the state-dict restatement of load_super).
    The default source ("ema_last") on the same transition gives the last EMA copy instead -- the two must differ after training."""
    from autoprog_amd.dist import GradientBucketReducer
    from autoprog_amd.loss import TokenLabelCrossEntropy
    from autoprog_amd.models import create_model
    from autoprog_amd.optim import FlatAdamWEma
    from autoprog_amd.prog import elastic, growth
    torch.manual_seed(0)
    model = create_model("model_variant", variant="volo_h2_l6", num_classes=16, img_size=64, stem_hidden_dim=16).cuda().train()
    red = GradientBucketReducer(list(model.parameters()), world_size=1)
    red.install_sink(model)
    opt = FlatAdamWEma(model, red, lr=1e-3, weight_decay=0.05, ema_decays=[0.5, 0.6, 0.7, 0.8])
    loss_fn = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=16)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(4, 3, 64, 64, generator=g).cuda()
    target = torch.softmax(torch.randn(4, 16, 18, generator=g) * 2, dim=1).cuda()
    try:
        old_mask = model.set_sample_config(dict(layer_num=6, min_layer_num=3, max_layer_num=6))
        np.random.seed(0)
        for _ in range(3):
            _step(model, red, opt, loss_fn, x, target)
        model_before = {k: v.detach().clone() for k, v in model.state_dict().items()}
        ema_before = [{k: v.detach().clone() for k, v in opt.ema_state_dict(i).items()} for i in range(4)]
        new_mask = elastic.make_mask(4, 3, 6)
        opt.grow(old_mask, new_mask, model_source="model")
        model.set_sample_config(dict(layer_num=4, min_layer_num=3, max_layer_num=6))
        shape4 = create_model("model_variant", variant="volo_h2_l4", num_classes=16, img_size=64, stem_hidden_dim=16).state_dict()
        skip = ("running_mean", "running_var", "num_batches_tracked")
        want = growth.extract_subnet(shape4, elastic.export_state_dict(model_before, old_mask), base_layer=3)
        got = elastic.export_state_dict({k: v.detach() for k, v in model.state_dict().items()}, new_mask)
        assert set(got) == set(want)
        differs_from_ema = False
        want_ema3 = growth.extract_subnet(shape4, elastic.export_state_dict(ema_before[3], old_mask), base_layer=3)
        for k, v in want.items():
            if k.rsplit(".", 1)[-1] in skip:
                continue
            assert torch.equal(got[k].cpu(), v.cpu()), k
            differs_from_ema |= not torch.equal(v.cpu(), want_ema3[k].cpu())
        assert differs_from_ema                      # i.e. the test can tell the two sources apart
        for i in (0, 2):
            want_e = growth.extract_subnet(shape4, elastic.export_state_dict(ema_before[i], old_mask), base_layer=3)
            got_e = elastic.export_state_dict(opt.ema_state_dict(i), new_mask)
            for k, v in want_e.items():
                if k.rsplit(".", 1)[-1] not in skip:
                    assert torch.equal(got_e[k].cpu(), v.cpu()), (i, k)
        assert float(opt.m.abs().sum()) == 0.0 and float(opt.v.abs().sum()) == 0.0 and opt.step_count == 0
        losses = [_step(model, red, opt, loss_fn, x, target) for _ in range(3)]
        assert all(np.isfinite(losses))
    finally:
        red.remove()


def test_autoprog_driver_two_stage_search():
    """prog/driver.py: a miniature AutoProg run -- stage 0 opens with a search over r in {64, 96} x
    l in {3, 6} on one volo_h2_l6 supernet (random sub-network per step, EMA probes, loss * time^w ranking), the run continues at
    the chosen point, the last stage takes the scheduled (l, r); EMA probing leaves the live weights untouched."""
    from autoprog_amd.dist import GradientBucketReducer
    from autoprog_amd.loss import TokenLabelCrossEntropy
    from autoprog_amd.models import create_model
    from autoprog_amd.optim import FlatAdamWEma
    from autoprog_amd.prog.driver import AutoProgDriver
    torch.manual_seed(0)
    np.random.seed(0)
    model = create_model("model_variant", variant="volo_h2_l6", num_classes=16, img_size=96, stem_hidden_dim=16).cuda().train()
    red = GradientBucketReducer(list(model.parameters()), world_size=1, defer_mean=True)
    red.install_sink(model)
    opt = FlatAdamWEma(model, red, lr=1e-3, weight_decay=0.05, ema_decays=[0.9, 0.99])
    loss_fn = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=16)
    g = torch.Generator().manual_seed(1)
    images = torch.randn(8, 3, 96, 96, generator=g).cuda()
    targets = {r: torch.softmax(torch.randn(8, 16, 2 + (r // 16) ** 2, generator=g) * 2, dim=1).cuda() for r in (64, 96)}
    logs = []
    drv = AutoProgDriver(model, loss_fn, opt, red, lambda r: (images, targets[r]), r_list=[64, 96], l_list=[3, 6], dp_list=[0.0, 0.05],
                         grow_epochs=[0, 3], steps_per_epoch=4, search_epochs=1, auto_grow=True, probe_batches=2, time_steps=2, log=logs.append)
    try:
        hist = drv.run(5)
        searches = [h for h in hist if h["kind"] == "search"]
        trains = [h for h in hist if h["kind"] == "train"]
        assert len(searches) == 1 and set(searches[0]["candidates"]) == {"r64_l3", "r64_l6", "r96_l3", "r96_l6"}
        assert searches[0]["chosen"] in [(64, 3), (64, 6), (96, 3), (96, 6)] and searches[0]["w"] >= 0
        assert [t["epoch"] for t in trains] == [1, 2, 3, 4]                       # epoch 0 was the search epoch
        assert (trains[0]["r"], trains[0]["l"]) == searches[0]["chosen"] and (trains[-1]["r"], trains[-1]["l"]) == (96, 6)
        assert all(np.isfinite(t["loss"]) for t in trains)
        before = opt.p.clone()
        with opt.ema_weights(0):
            assert not torch.equal(opt.p, before)
        assert torch.equal(opt.p, before)
    finally:
        red.remove()


def test_driver_gradient_accumulation_equals_one_large_batch():
    """batch splits (reference --batch-splits-list / `update` flag, main_prog.py:567-574,971,1019-1027): one driver update made of
    two micro-batches (backward on loss / 2 each, gradient exchange and optimizer on the second only) against ONE step on the
    concatenated batch.  BatchNorm statistics and the mix-token partner are per micro-batch in the reference too, so the comparison
    uses a model state where neither matters: eval-free check on the Linear / LayerNorm parameters of the transformer stages through
    GradientBucketReducer(accumulate_steps=2) directly -- the slab after two accumulating backward passes equals the sum of the two
    passes' own gradients (to fp32 rounding: the kernels add in place into the same fp32 slab)."""
    model, red, opt, loss_fn, x, target = _setup()
    try:
        halves = [(x[:2].contiguous(), target[:2].contiguous()), (x[2:].contiguous(), target[2:].contiguous())]
        singles = []
        for xi, ti in halves:                                   # each micro-batch on its own
            red.zero_grad()
            np.random.seed(5)
            (loss_fn(model(xi), ti) / 2).backward()
            red.finish()
            singles.append(red.flat.clone())
        red.set_accumulate_steps(2)
        red.zero_grad()
        for i, (xi, ti) in enumerate(halves):
            np.random.seed(5)
            (loss_fn(model(xi), ti) / 2).backward()
            red.finish()
            assert red.is_update_step == (i == 1)
        want = singles[0] + singles[1]
        err = float((red.flat - want).norm() / want.norm())
        assert err < 2e-5, err                                  # fp32 adds in another order (split launches meet in fp32 atomics), nothing more
        red.set_accumulate_steps(1)
    finally:
        red.remove()


def test_weight_gradient_window_matches_one_launch_per_block(monkeypatch):
    """functional's weight-gradient window: the blocks' weight gradients (and LayerNorm parameter gradients) leave in a few launches
    for the whole backward pass instead of one per block, and param.grad holds the same numbers (only the order of fp32 additions
    differs); the window is empty after the pass; AP_WGRAD_WINDOW=0 is the per-block behaviour"""
    from autoprog_amd import functional as AF, ops
    model, red, opt, loss_fn, x, target = _setup()
    try:
        calls = []
        real = ops.gemm_tn_acc_grouped
        monkeypatch.setattr(ops, "gemm_tn_acc_grouped", lambda problems, ln=None: (calls.append((len(problems), len(ln or []))), real(problems, ln=ln))[1])
        grads = {}
        for window in (0, 256):
            monkeypatch.setattr(AF, "WGRAD_WINDOW", window)
            calls.clear()
            np.random.seed(0); torch.manual_seed(0)
            red.zero_grad()
            loss_fn(model(x), target).backward()
            red.finish()
            torch.cuda.synchronize()
            grads[window] = red.flat.detach().clone()
            launches = [c for c in calls if c[0] > 1]
            if window:
                assert len(launches) < n_block_launches and sum(c[0] for c in calls) == n_problems and sum(c[1] for c in calls) == n_ln
                assert not AF._window["units"] and not AF._window["armed"]
            else:
                n_block_launches, n_problems, n_ln = len(launches), sum(c[0] for c in calls), sum(c[1] for c in calls)
                assert n_block_launches >= 3
        assert float(grads[0].abs().max()) > 0
        err = float((grads[256] - grads[0]).norm() / grads[0].norm())
        assert err < 1e-5, err
    finally:
        red.remove()
