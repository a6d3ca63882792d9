"""Launch-order independence of the training step, at the bench geometry and on small ones: the same kernels must give
the same BITS whether they go out eagerly on one stream, eagerly with the weight gradients on the side stream
(MIMO_WGRAD_STREAM), with the side stream's consumers arriving late (MIMO_DEBUG_WGRAD_DELAY_US: an idle kernel in front
of every weight gradient — the hazard class of round 5's max |dz| release race), or as hipGraph replays of the training
forward / backward (MIMO_TRAIN_GRAPH, round 6).  What is compared: the flat gradient buffer after every backward, the
predictions of every step, the parameters and BatchNorm buffers after the last Adam step.

Reference semantics at stake: one optimiser step of `MimoUnetModel.training_step` (mimo/models/mimo_unet.py:115-144) +
backward + Adam is a deterministic function of (parameters, batch, permutations)."""
import os

import pytest
import torch

from oracle import mimo_oracle as O

pytestmark = pytest.mark.gpu


def _model(cfg, state, dropout=(0.0, 0.0, 0.0)):
    from mimo.models.mimo_unet import MimoUnetModel
    m = MimoUnetModel(in_channels=cfg.in_channels, out_channels=cfg.out_channels, num_subnetworks=cfg.num_subnetworks,
                      filter_base_count=cfg.filter_base_count, center_dropout_rate=0.0, final_dropout_rate=0.0,
                      encoder_dropout_rate=dropout[0], core_dropout_rate=dropout[1], decoder_dropout_rate=dropout[2],
                      loss="laplace_nll", weight_decay=0.0, learning_rate=1e-3, seed=0, loss_buffer_size=10,
                      loss_buffer_temperature=0.3)
    m.load_state_dict({"model." + k: v for k, v in state.items()})
    return m.cuda().train()


def _run(monkeypatch, env, cfg, state, batches, *, staged=False, dropout=(0.0, 0.0, 0.0), seed=7):
    """`len(batches)` Adam steps from `state` under the environment `env` (read when the plan is created)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    torch.manual_seed(seed)  # the engine's dropout streams and torch.randperm follow torch's generators
    model = _model(cfg, state, dropout)
    if staged:  # the data-parallel caller's route: mimo_backward_stage 0..7 with a hook between the stages
        model.model.grad_ready_hook = lambda flat, b, e: None
    opt = model.configure_optimizers()["optimizer"]
    grads, preds = [], []
    for i, (image, label, mask, perms) in enumerate(batches):
        opt.zero_grad()
        out = model.training_step_with_perms(image, label, mask, perms)
        out["loss"].backward()
        grads.append(model.model.flat_gradients().clone())
        preds.append(out["preds"].clone())
        opt.step()
    torch.cuda.synchronize()
    assert model.model.numerics_status() == 0
    for k in env:
        monkeypatch.delenv(k)
    return grads, preds, model.model.flat_parameters().clone(), model.model._flat_buffers.clone()


def _same(a, b, what):
    for name, xs, ys in (("gradient buffer", a[0], b[0]), ("predictions", a[1], b[1])):
        for i, (x, y) in enumerate(zip(xs, ys)):
            assert torch.equal(x, y), f"{what}: {name} of step {i} differs by up to {float((x - y).abs().max()):.3e}"
    assert torch.equal(a[2], b[2]), f"{what}: parameters after the last step differ"
    assert torch.equal(a[3], b[3]), f"{what}: BatchNorm buffers after the last step differ"


def _bench_batches(batch, steps, cfg):
    g = torch.Generator(device="cuda").manual_seed(100)
    gc = torch.Generator().manual_seed(101)
    out = []
    for _ in range(steps):  # a new batch and new permutations every step: staged copies must follow the caller's tensors
        image = torch.rand(batch, cfg.in_channels, 256, 256, device="cuda", generator=g)
        label = torch.rand(batch, 1, 256, 256, device="cuda", generator=g)
        out.append((image, label, None, O.draw_perms(batch, cfg.num_subnetworks, generator=gc).cuda()))
    return out


CFG3 = O.NetConfig(in_channels=2, out_channels=2, num_subnetworks=2, filter_base_count=30)


@pytest.mark.parametrize("batch", [4, 32])
def test_stream_protocol_at_the_bench_geometry(batch, monkeypatch):
    """cfg3 (2->1 ch, 256 x 256, S = 2, fbc = 30) at its batch and at its 8-GPU shard, 10 Adam steps: one stream == side
    stream == side stream with every weight gradient 200 us late == hipGraph replay, bit for bit.  (until
    round 6 this comparison was a tool, tests/tools/stream_stress.py, and the suite only covered small geometries.)"""
    state = O.init_state(CFG3, 1)
    batches = _bench_batches(batch, 10, CFG3)
    base = _run(monkeypatch, {"MIMO_WGRAD_STREAM": "0", "MIMO_TRAIN_GRAPH": "0"}, CFG3, state, batches)
    side = _run(monkeypatch, {"MIMO_WGRAD_STREAM": "1", "MIMO_TRAIN_GRAPH": "0"}, CFG3, state, batches)
    _same(base, side, "side stream vs one stream")
    late = _run(monkeypatch, {"MIMO_WGRAD_STREAM": "1", "MIMO_TRAIN_GRAPH": "0", "MIMO_DEBUG_WGRAD_DELAY_US": "200"}, CFG3,
                state, batches)
    _same(base, late, "late weight gradients vs one stream")
    graph = _run(monkeypatch, {"MIMO_WGRAD_STREAM": "1", "MIMO_TRAIN_GRAPH": "1"}, CFG3, state, batches)
    _same(base, graph, "hipGraph replay vs eager")


def _small_case():
    cfg = O.NetConfig(2, 2, 3, 10, encoder_dropout_rate=0.2)
    st = O.init_state(cfg, 91)
    g = torch.Generator().manual_seed(92)
    batches = []
    for _ in range(6):
        image, label = torch.rand(3, 2, 100, 100, generator=g).cuda(), torch.rand(3, 1, 100, 100, generator=g).cuda()
        mask = (torch.rand(3, 1, 100, 100, generator=g) > 0.3).float().cuda()
        batches.append((image, label, mask, O.draw_perms(3, 3, generator=g).cuda()))
    return cfg, st, batches


@pytest.mark.parametrize("staged", [False, True], ids=["whole-backward", "staged-backward"])
def test_training_step_graphs_replay_the_eager_step(staged, monkeypatch):
    """MIMO_TRAIN_GRAPH=1 against =0 on the geometry of round 5's race (S = 3, fbc = 10, 100 x 100, batch 3) with everything
    a captured kernel must not take from the caller: a masked loss, new tensors every step, Dropout2d multipliers drawn in the
    engine from torch's generator — through mimo_backward (one graph) and through mimo_backward_stage (one graph per stage,
    the data-parallel route).  Step 1 runs eagerly (first sighting of the call shape), step 2 captures, steps 3-6 replay."""
    cfg, st, batches = _small_case()
    eager = _run(monkeypatch, {"MIMO_TRAIN_GRAPH": "0"}, cfg, st, batches, staged=staged, dropout=(0.2, 0.0, 0.0))
    graph = _run(monkeypatch, {"MIMO_TRAIN_GRAPH": "1"}, cfg, st, batches, staged=staged, dropout=(0.2, 0.0, 0.0))
    _same(eager, graph, "hipGraph replay vs eager")
    late = _run(monkeypatch, {"MIMO_TRAIN_GRAPH": "1", "MIMO_DEBUG_WGRAD_DELAY_US": "100"}, cfg, st, batches, staged=staged,
                dropout=(0.2, 0.0, 0.0))
    _same(eager, late, "hipGraph replay with late weight gradients vs eager")


def test_training_graphs_follow_call_shape_changes(monkeypatch):
    """The key of a captured step covers what the caller may change between steps: with / without a loss mask, whole /
    staged backward, an eval-mode forward + backward in between (FGSM), gradient accumulation (a live .grad: the engine's
    buffer is saved and added back).  Every step's gradient equals the eager run's."""
    cfg, st, batches = _small_case()

    def sequence(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        torch.manual_seed(3)
        model = _model(cfg, st)
        res = []
        plan_steps = [(0, True), (1, True), (2, True), (3, False), (4, False), (5, False), (0, True), (1, False)]
        for i, (b, with_mask) in enumerate(plan_steps):
            image, label, mask, perms = batches[b]
            model.model.grad_ready_hook = (lambda flat, lo, hi: None) if i in (4, 5) else None
            if i != 6:  # step 6 accumulates into step 5's gradients
                model.zero_grad()
            model.training_step_with_perms(image, label, mask if with_mask else None, perms)["loss"].backward()
            res.append(model.model.flat_gradients().clone())
            if i == 2:  # an eval-mode forward with a backward to the input between two training steps
                model.eval()
                x5 = torch.stack([image[perms[s]] for s in range(3)], 1).requires_grad_(True)
                p1, p2 = model(x5)
                (p1.mean() + p2.mean()).backward()
                res.append(x5.grad.clone())
                model.train()
        torch.cuda.synchronize()
        for k in env:
            monkeypatch.delenv(k)
        return res

    a, b = sequence({"MIMO_TRAIN_GRAPH": "0"}), sequence({"MIMO_TRAIN_GRAPH": "1"})
    for i, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x, y), f"result {i} differs by up to {float((x - y).abs().max()):.3e}"
