"""RunLog against one of the reference's own log files (tests/golden/ref_stats_log.json):
replaying its per-epoch entries must reproduce its `results` block and its key layout."""
import json
import os

import numpy as np
import pytest

from vit_torch_amd.stats import RunLog, probe_hardware

HERE = os.path.dirname(os.path.abspath(__file__))
REF = json.load(open(os.path.join(HERE, "golden", "ref_stats_log.json")))


class Clock:
    def __init__(self):
        self.t = 0.0

    def __call__(self):
        return self.t


def replay(path=None):
    clk = Clock()
    log = RunLog(path=path, info=REF["info"], clock=clk,
                 telem={k: REF["telem"][k] for k in ("sample_count_train", "sample_count_val", "time_stamp", "mode")})
    for tr, va in zip(REF["train"], REF["val"]):
        for split, r in (("train", tr), ("val", va)):
            clk.t = r["time_start"]
            log.new_round(split)
            clk.t = r["time_finish"]
            log.finish_round(split, epoch=r["epoch"], lr=r["lr"], loss=r["loss"], acc=r["acc"], sample=r["sample"])
    return log, clk


def test_schema_and_entries_match_reference_log():
    log, _ = replay()
    s = log.stats
    assert list(s) == list(REF)
    assert set(s["telem"]) == set(REF["telem"])
    assert set(s["results"]) == set(REF["results"])
    for split in ("train", "val"):
        assert len(s[split]) == len(REF[split])
        for a, b in zip(s[split], REF[split]):
            assert set(a) == set(b)
            for k in b:
                assert a[k] == pytest.approx(b[k], rel=1e-12, abs=1e-9), (split, k)


def test_results_block_reproduces_reference():
    log, _ = replay()
    for k, v in REF["results"].items():
        assert log.results[k] == pytest.approx(v, rel=1e-9, abs=1e-12), k


def test_finish_and_save_round_trip(tmp_path):
    p = str(tmp_path / "logs" / "stats_x.json")
    log, clk = replay(p)
    clk.t = REF["val"][-1]["time_finish"] + 1.0
    log.finish()
    d = json.load(open(p))
    assert d["telem"]["completed"] is True
    assert d["telem"]["time_start"] == REF["train"][0]["time_start"]
    assert d["telem"]["time_elapsed"] == pytest.approx(clk.t - REF["train"][0]["time_start"])
    assert d["results"]["epochs"] == REF["results"]["epochs"]
    with pytest.raises(AssertionError):
        log.save(str(tmp_path / "x.txt"))


def test_hardware_string_is_probed():
    assert probe_hardware(8).startswith("8x") and "3090" not in probe_hardware(8)


def test_network_fit_writes_log(tmp_path):
    """Host logic only: a stub network feeds fit()'s logging path on CPU."""
    from vit_torch_amd.network import Network

    class Stub(Network):
        def __init__(self):
            import torch
            self.epochs = 2
            w = torch.nn.Parameter(torch.zeros(1))
            self.optimizer = torch.optim.SGD([w], lr=0.5)
            self.lr_scheduler = torch.optim.lr_scheduler.LambdaLR(self.optimizer, lambda e: 0.5 ** e)

        def run_one_epoch(self, loader, training=True):
            return {"loss": [1.0], "loss_avg": 1.0 if training else 2.0,
                    "correct": np.array([True, False, True, True]), "acc": 0.75}

    log = RunLog(path=str(tmp_path / "s.json"), telem={"hardware": probe_hardware()})
    hist = Stub().fit([0], [0], log=log)
    d = json.load(open(tmp_path / "s.json"))
    assert [r["lr"] for r in d["train"]] == [0.5, 0.25] and [r["lr"] for r in d["val"]] == [0.0, 0.0]
    assert d["train"][1]["sample"] == 4 and d["val"][0]["loss"] == 2.0 and d["results"]["val.acc"] == 0.75
    assert d["telem"]["completed"] is True and len(hist) == 2
