"""LR lambdas of the harness (utils_network.py:35-73, 529-544) against hand-evaluated values."""
import math

import numpy as np
import pytest
import torch


def test_lr_lambdas():
    from vit_torch_amd.network import LRSchedule
    step = LRSchedule.get_step_fn(step=10, gamma=0.5)
    assert [step(e) for e in (0, 9, 10, 19, 20, 30)] == [1.0, 1.0, 0.5, 0.5, 0.25, 0.125]
    exp = LRSchedule.get_exp_fn(gamma=0.99)
    assert exp(0) == 1.0 and abs(exp(10) - 0.99 ** 10) < 1e-15
    cos = LRSchedule.get_cosine(step=20, min_scale=0.1)
    assert abs(cos(0) - 1.0) < 1e-12
    assert abs(cos(5) - (0.45 * (math.cos(0.25 * 2 * math.pi) + 1) + 0.1)) < 1e-12
    assert abs(cos(10) - 1.0) < 1e-12          # mod(10/20, 0.5) = 0: the reference's saw-tooth restart
    ce = LRSchedule.get_cosine_exp(step=20, min_scale=0.1, gamma=0.5)
    assert abs(ce(5) - cos(5) * 0.5 ** 0.25) < 1e-12
    assert LRSchedule.get_base_fn()(7) == 1.0


def test_scheduler_table_and_none_quirk():
    from vit_torch_amd.network import get_lr_scheduler
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=0.1, momentum=0.9)
    sch = get_lr_scheduler(opt, "step", step=2, gamma=0.5)
    lrs = []
    for _ in range(5):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step(); sch.step()
    assert np.allclose(lrs, [0.1, 0.1, 0.05, 0.05, 0.025])
    opt2 = torch.optim.SGD([p], lr=0.1)
    get_lr_scheduler(opt2, "none")
    assert opt2.param_groups[0]["lr"] == 0.0            # 'none' -> lambda e: e
    with pytest.raises(NotImplementedError):
        get_lr_scheduler(opt2, "ca")


def test_network_rejects_non_modules():
    from vit_torch_amd.network import Network
    with pytest.raises(ValueError):
        Network(model="not a module")


def test_bench_parent_launches_ranks_without_touching_the_gpu(monkeypatch):
    """`python bench.py --gpus N` (no torch.distributed.run environment): the parent must start N
    ranks as CHILD processes of `python -m torch.distributed.run ... --master-addr 127.0.0.1`,
    hand them its own arguments, and return their status; it must not call into torch.cuda."""
    import importlib
    import subprocess
    import sys
    import torch
    bench = importlib.import_module("bench")
    calls = {}

    def fake_call(cmd, env=None):
        calls["cmd"], calls["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(torch.cuda, "set_device", lambda *_: (_ for _ in ()).throw(AssertionError("GPU touched")))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 7
    cmd = calls["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(bench.__file__ if bench.__file__ in cmd else [c for c in cmd if c.endswith("bench.py")][0])
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_refuses_a_traffic_file_from_other_kernel_sources(tmp_path):
    """`roofline.traffic` comes from a committed PMC file; bench.py must quote it only when
    the file's kernel-source fingerprint equals that of the sources it runs (tools/pmc_traffic.py records it)."""
    import hashlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    csrc = os.path.join(root, "vit_torch_amd", "csrc")
    h = hashlib.sha256()
    for fn in sorted(os.listdir(csrc)):
        if fn.endswith((".hip", ".h", ".cpp")):
            h.update(fn.encode()); h.update(open(os.path.join(csrc, fn), "rb").read())
    assert bench.kernel_sources_sha256() == h.hexdigest()
    src = open(os.path.join(root, "bench.py")).read()
    assert 'meta.get("kernel_sources_sha256") != kernel_sources_sha256()' in src and "refused, traffic = null" in src
    tool = open(os.path.join(root, "tools", "pmc_traffic.py")).read()
    assert "kernel_sources_sha256" in tool


# ------------------------------------------------------------------ early stopping (utils_network.py:322-328) ---
def _scripted_network(val_accs):
    """A Network over a tiny CPU module whose epochs return scripted validation accuracies."""
    from vit_torch_amd.network import Network

    class Scripted(Network):
        def __init__(self):
            super().__init__(torch.nn.Linear(4, 2), opt="torch_sgd", loss_fn=torch.nn.CrossEntropyLoss(), device="cpu",
                             epochs=len(val_accs), earlystop_epoch=3)
            self.calls = []

        def run_one_epoch(self, dataloader, training=True):
            self.calls.append("train" if training else "val")
            n_val = sum(1 for c in self.calls if c == "val")
            acc = 0.1 if training else val_accs[n_val - 1]
            return {"loss": [1.0], "loss_avg": 1.0, "correct": np.zeros(4, dtype=bool), "acc": acc}

    return Scripted()


def test_fit_stops_early_like_the_reference():
    """After a validation round: once `earlystop_epoch` accuracies exist, stop if none of the last `earlystop_epoch` reaches
    the best so far; the stop takes effect at the top of the NEXT epoch (utils_network.py:256-259, 322-328)."""
    accs = [0.50, 0.60, 0.55, 0.50, 0.65, 0.70]
    net = _scripted_network(accs)
    hist = net.fit([None], [None], earlystop_epoch=2)
    # after epoch 3 the last two are (0.55, 0.50) < best 0.60 -> epoch 4 does not run
    assert len(hist) == 4 and net.stopped_early_after == 4
    assert net.calls == ["train", "val"] * 4
    # a window that still holds the best never stops
    net = _scripted_network(accs)
    assert len(net.fit([None], [None], earlystop_epoch=3)) == 6 and net.stopped_early_after is None
    # the reference's quirk: the constructor's earlystop_epoch (3 here) is ignored, fit()'s own default of 10 rules
    net = _scripted_network([0.9] + [0.1] * 11)
    hist = net.fit([None], [None])
    assert len(hist) == 11 and net.stopped_early_after == 11       # accs 1..10 (ten of them) < 0.9 after epoch 10
    # earlystop_epoch = 0: the window is the whole history, which always holds the best
    net = _scripted_network([0.9, 0.1, 0.1, 0.1])
    assert len(net.fit([None], [None], earlystop_epoch=0)) == 4
    # no validation loader: nothing to stop on
    net = _scripted_network([0.9, 0.1, 0.1, 0.1])
    assert len(net.fit([None], None, earlystop_epoch=1)) == 4


def test_sharded_loader_splits_rows_and_refuses_uneven_batches():
    from vit_torch_amd.network import ShardedLoader
    x, y = torch.arange(24.).view(8, 3), torch.arange(8)
    parts = [list(ShardedLoader([(x, y)], r, 4)) for r in range(4)]
    assert torch.equal(torch.cat([p[0][1] for p in parts]), y)
    assert torch.equal(parts[2][0][0], x[4:6])
    with pytest.raises(ValueError):
        list(ShardedLoader([(x[:6], y[:6])], 0, 4))


def test_network_ddp_needs_a_process_group():
    from vit_torch_amd.network import Network
    with pytest.raises(ValueError, match="process group"):
        Network(torch.nn.Linear(4, 2), opt="torch_sgd", loss_fn=torch.nn.CrossEntropyLoss(), device="cpu",
                ddp={"size": 2, "rank": 0})
