"""GraphedStep: the captured HIP graph of a whole training step reproduces the eager steps."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make():
    from vit_torch_amd import CrossEntropyLoss, FusedSGD, VisionTransformer
    torch.manual_seed(5)
    m = VisionTransformer(img_size=32, patch_size=8, embed_dim=128, depth=3, num_heads=4, num_classes=10,
                          compute_dtype="bf16", residual_dtype="auto").cuda()
    m.head = torch.nn.Linear(128, 10, bias=False).cuda()
    return m, CrossEntropyLoss(), FusedSGD(m.parameters(), lr=5e-2, momentum=0.9)


def _batches(n):
    g = torch.Generator("cpu").manual_seed(1)
    return [(torch.randn(64, 3, 32, 32, generator=g).cuda(), torch.randint(0, 10, (64,), generator=g).cuda())
            for _ in range(n)]


def test_graph_replay_matches_eager_steps(lib):
    from vit_torch_amd.graph import GraphedStep
    data = _batches(5)
    m, crit, opt = _make()
    eager = []
    for x, y in data:
        opt.zero_grad()
        loss = crit(m(x), y)
        loss.backward()
        opt.step()
        eager.append(loss.item())
    p_eager = m.engine().pack.flat.clone()

    m2, crit2, opt2 = _make()
    m2.load_state_dict({k: v for k, v in _make()[0].state_dict().items()})      # same seed -> same init
    start = m2.engine().pack.flat.clone()
    step = GraphedStep(m2, crit2, opt2, *data[0], warmup=1)
    # the capture ran warm-up + capture passes on the first batch: restart from the initial state
    with torch.no_grad():
        m2.engine().pack.flat.copy_(start)
    opt2.reset_state()
    graphed = [step(x, y).item() for x, y in data]
    assert graphed == pytest.approx(eager, rel=1e-5, abs=1e-6), (graphed, eager)
    torch.testing.assert_close(m2.engine().pack.flat, p_eager, rtol=1e-5, atol=1e-6)


def test_graph_recaptures_when_the_learning_rate_changes(lib):
    from vit_torch_amd.graph import GraphedStep
    (x, y), = _batches(1)
    m, crit, opt = _make()
    step = GraphedStep(m, crit, opt, x, y, warmup=1)
    g0 = step.graph
    step(x, y)
    assert step.graph is g0
    opt.param_groups[0]["lr"] = 1e-3
    step(x, y)
    assert step.graph is not g0, "a new LR must not be replayed with the old kernel argument"


def test_network_harness_with_hip_graph_matches_eager_harness(lib):
    """Network(hip_graph=True): first batch eager (+capture), same-shape batches replayed, the
    short last batch eager; two epochs with an LR step in between (re-capture)."""
    from vit_torch_amd.network import Network
    data = _batches(4)
    data.append((data[0][0][:40].clone(), data[0][1][:40].clone()))        # ragged last batch
    hist = {}
    for use_graph in (False, True):
        m, _, _ = _make()
        net = Network(m, opt="sgd", lr=5e-2, lr_type="step", lr_step=1, lr_gamma=0.5, hip_graph=use_graph)
        hist[use_graph] = net.fit(data, epochs=2)
        hist[use_graph].append(m.engine().pack.flat.clone())
    for e in range(2):
        assert hist[True][e]["train"]["loss"] == pytest.approx(hist[False][e]["train"]["loss"], rel=1e-5, abs=1e-6)
        assert (hist[True][e]["train"]["correct"] == hist[False][e]["train"]["correct"]).all()
    torch.testing.assert_close(hist[True][2], hist[False][2], rtol=1e-5, atol=1e-6)


def test_graph_static_inputs_written_in_place(lib):
    """A loader may write its batch straight into the graph's static inputs (step.x / step.y) and pass them back:
    no copy is made, and the replay sees the new batch exactly as the copying call does."""
    from vit_torch_amd.graph import GraphedStep
    data = _batches(4)
    ma, crita, opta = _make()
    mb, critb, optb = _make()                      # same seed: same initial weights
    a = GraphedStep(ma, crita, opta, *data[0], warmup=1)
    b = GraphedStep(mb, critb, optb, *data[0], warmup=1)
    for x, y in data[1:]:
        la = a(x, y).item()                        # copied into the static inputs by the call
        b.x.copy_(x)
        b.y.copy_(y)
        lb = b(b.x, b.y).item()                    # written in place by the "loader"
        assert la == lb
    assert torch.equal(ma.engine().pack.flat, mb.engine().pack.flat)


def test_graph_without_the_weight_cast_follows_weights_changed_between_replays(lib):
    """With a fused optimizer over all parameters the fp32 -> bf16 weight cast stays out of the captured graph (the optimizer
    kernel writes master and shadow together); GraphedStep checks the pack's version key eagerly before each replay, so
    weights loaded between replays are still used.  A stock torch optimizer keeps the recorded cast."""
    from vit_torch_amd.graph import GraphedStep
    data = _batches(3)
    m, crit, opt = _make()
    step = GraphedStep(m, crit, opt, *data[0], warmup=1)
    assert step._pack is m.engine().pack and not step._pack.capture_skips_cast
    step(*data[1])
    # new weights through load_state_dict (bumps the parameters' version counters), then one graphed and one eager step
    torch.manual_seed(11)
    other = {k: torch.randn_like(v) * 0.05 for k, v in m.state_dict().items()}
    m.load_state_dict(other)
    opt.reset_state()
    loss_g = step(*data[2]).item()
    m2, crit2, opt2 = _make()
    m2.load_state_dict(other)
    opt2.zero_grad()
    loss_e = crit2(m2(data[2][0]), data[2][1])
    loss_e.backward()
    opt2.step()
    assert loss_g == pytest.approx(loss_e.item(), rel=1e-5, abs=1e-6)
    torch.testing.assert_close(m.engine().pack.flat, m2.engine().pack.flat, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(m.engine().pack.shadow.float(), m2.engine().pack.shadow.float(), rtol=0, atol=0)

    m3, crit3, _ = _make()
    stock = torch.optim.SGD(m3.parameters(), lr=5e-2, momentum=0.9)
    assert GraphedStep(m3, crit3, stock, *data[0], warmup=1)._pack is None


# ---- round 5: the capture contract that came out of round 4's two crashes -----------------------------------
def _small_step(seed=5, **kw):
    from vit_torch_amd import CrossEntropyLoss, FusedSGD, VisionTransformer
    torch.manual_seed(seed)
    m = VisionTransformer(img_size=32, patch_size=16, embed_dim=128, depth=2, num_heads=2, num_classes=10,
                          compute_dtype="bf16", residual_dtype="auto", **kw).cuda()
    m.head = torch.nn.Linear(128, 10, bias=False).cuda()
    m.engine()
    g = torch.Generator("cpu").manual_seed(0)
    x, y = torch.randn(32, 3, 32, 32, generator=g).cuda(), torch.randint(0, 10, (32,), generator=g).cuda()
    return m, CrossEntropyLoss(), FusedSGD(m.parameters(), lr=1e-2, momentum=0.9), x, y


def test_a_second_graphed_step_on_a_live_engine_is_refused_and_close_releases_it():
    """Two captured steps over ONE engine share its flat buffers, optimizer state and cached tables; round 4's capture_end
    segfault sat on a second GraphedStep beside a live first.  The second is refused with a VitmiError; after close() (or
    when the first is garbage) the engine can be captured again, and the new graph's replay equals the eager step."""
    from vit_torch_amd import GraphedStep
    from vit_torch_amd._lib import VitmiError
    m, crit, opt, x, y = _small_step()
    gs = GraphedStep(m, crit, opt, x, y)
    with pytest.raises(VitmiError, match="live GraphedStep"):
        GraphedStep(m, crit, opt, x, y)
    l1 = float(gs(x, y).item())                    # the refused construction left the first graph usable
    gs.close()
    assert gs.graph is None
    gs2 = GraphedStep(m, crit, opt, x, y, warmup=0)      # everything is initialised: no warm-up needed on the same engine
    l2 = float(gs2(x, y).item())
    assert l1 == l1 and l2 == l2 and l2 < l1 + 1.0
    # same weights, one eager step and one replayed step: the same loss
    m3, crit3, opt3, _, _ = _small_step()
    m3.load_state_dict(m.state_dict())
    opt3.zero_grad(); want = crit3(m3(x), y); want.backward(); opt3.step()
    got = gs2(x, y)
    assert float(got.item()) == pytest.approx(float(want.item()), rel=2e-3)


def test_recapture_after_the_engine_was_rebuilt_warms_up_eagerly_first():
    """A model whose configuration moved since the capture gets a NEW engine (new flat buffers, lazily built tables).
    `recapture()` with warmup = 0 must not run that engine's first step inside the capture (lazy initialisation there is
    illegal, raises, and the capture_end behind an invalidated capture is where round 4 crashed): it resolves the engine
    eagerly, sees that it is not the one the graph was captured on, and runs one eager step first."""
    from vit_torch_amd import GraphedStep
    m, crit, opt, x, y = _small_step()
    gs = GraphedStep(m, crit, opt, x, y)
    eng0 = gs._engine
    m.cls_only_last_block = True                   # a configuration change: engine().is_current() turns false
    gs.warm_loss = None
    gs.recapture()                                 # warmup = 0 asked for
    assert gs._engine is not eng0 and gs._engine is m.engine()
    assert gs.warm_loss is not None, "the rebuilt engine's first step must have run eagerly"
    l = float(gs(x, y).item())
    assert l == l


def test_capture_tolerates_hip_calls_from_another_thread():
    """thread_local capture mode: a helper thread (ProcessGroupNCCL's watchdog polling hipEventQuery, a pin-memory thread)
    may call into HIP while the step is being captured.  A thread hammers event queries and allocations during the
    capture; under the default global mode these invalidate the capture."""
    import threading
    from vit_torch_amd import GraphedStep
    m, crit, opt, x, y = _small_step()
    ev = torch.cuda.Event()
    ev.record()
    torch.cuda.synchronize()
    stop = threading.Event()
    errors = []

    def poll():
        try:
            import time
            while not stop.is_set():
                ev.query()                          # hipEventQuery from another thread
                time.sleep(0.0005)
        except Exception as e:                      # pragma: no cover
            errors.append(e)

    t = threading.Thread(target=poll, daemon=True)
    t.start()
    try:
        gs = GraphedStep(m, crit, opt, x, y)
    finally:
        stop.set()
        t.join()
    assert not errors, errors
    l = float(gs(x, y).item())
    assert l == l and abs(l) < 1e3


def test_bf16x3_step_replays_from_a_graph():
    """The three-product parity mode allocates its operand images per call (torch.empty inside the capture: the graph's
    private pool): a replayed step must equal the eager one."""
    from vit_torch_amd import CrossEntropyLoss, FusedSGD, GraphedStep, VisionTransformer
    def make():
        torch.manual_seed(9)
        m = VisionTransformer(img_size=32, patch_size=16, embed_dim=128, depth=2, num_heads=2, num_classes=10,
                              compute_dtype="bf16x3", residual_dtype="fp32").cuda()
        m.head = torch.nn.Linear(128, 10, bias=False).cuda()
        m.engine()
        return m, CrossEntropyLoss(), FusedSGD(m.parameters(), lr=1e-2, momentum=0.9)
    g = torch.Generator("cpu").manual_seed(0)
    data = [(torch.randn(32, 3, 32, 32, generator=g).cuda(), torch.randint(0, 10, (32,), generator=g).cuda()) for _ in range(3)]
    m, crit, opt = make()
    eager = []
    for x, y in [data[0]] + data:
        opt.zero_grad(); loss = crit(m(x), y); loss.backward(); opt.step()
        eager.append(loss.item())
    m2, crit2, opt2 = make()
    gs = GraphedStep(m2, crit2, opt2, *data[0], warmup=1)
    graphed = [gs.warm_loss.item()] + [gs(x, y).item() for x, y in data]
    assert graphed == pytest.approx(eager, rel=1e-5, abs=1e-6), (graphed, eager)
    torch.testing.assert_close(m2.engine().pack.flat, m.engine().pack.flat, rtol=1e-5, atol=1e-6)
