"""Run log in the reference's on-disk schema.

The reference's `Stats` (/root/reference/utils_stats.py:510-719) writes one JSON per run:

    {"info": {...cli args...}, "telem": {...}, "results": {...}, "train": [...], "val": [...]}

so that runs can be diffed against the logs it ships under `logs/massA/*.json`.  `RunLog`
emits the same keys with the same meaning from what `Network.run_one_epoch` already returns
(one host read per epoch), without the reference's per-batch progress bars and counters:

* per-epoch entry (`StatMetrics.get_stat`, utils_stats.py:487-508):
  `epoch, sample, lr, loss, acc, time, time_start, time_finish, time_cost`
* `telem` (utils_stats.py:551-564, main.py:213-222): `hardware` is probed from the device
  instead of the reference's hard-coded `'1x3090'`
* `results` (utils_stats.py:639-698): best acc per split, mean wall time per split and per
  sample.  Quirk kept on purpose so files diff cleanly: `<split>.loss` is
  `max(losses + [9.999])`, i.e. 9.999 for any sane run (utils_stats.py:671-675).
"""
from __future__ import annotations

import json
import os
import time
from typing import Dict, List, Optional

import numpy as np


def probe_hardware(world_size: int = 1) -> str:
    """'<n>x<device name>' for telem['hardware'] (the reference hard-codes '1x3090', main.py:214)."""
    try:
        import torch
        if torch.cuda.is_available():
            return f"{world_size}x{torch.cuda.get_device_name(0)}"
    except Exception:
        pass
    return f"{world_size}xcpu"


class RunLog:
    splits = ("train", "val")

    def __init__(self, path: Optional[str] = None, info: Optional[dict] = None, telem: Optional[dict] = None,
                 clock=time.time):
        self.path = path
        self._clock = clock
        now = clock()
        self.info: dict = dict(info or {})
        self.telem: dict = {                                   # utils_stats.py:551-564
            "hardware": "<unknown>", "sample_count_train": 1, "sample_count_val": 1, "completed": False,
            "time_stamp": "<unknown>", "time_start": now, "time_finish": None, "time_elapsed": None,
            "time_updated": now, "bs": None, "mode": "<unknown>",
        }
        self.telem.update(telem or {})
        self.logs: Dict[str, List[dict]] = {s: [] for s in self.splits}
        self.epoch = 0
        self.results: dict = {}
        self._open: Dict[str, float] = {}
        self._refresh()

    # ---- one round = one pass over one split ------------------------------------------
    def new_round(self, split: str) -> None:
        assert split in self.splits
        self._open[split] = self._clock()

    def finish_round(self, split: str, epoch: int, lr: float, loss: float, acc: float, sample: int,
                     save: bool = True) -> dict:
        t1 = self._clock()
        t0 = self._open.pop(split, t1)
        rec = {"epoch": int(epoch), "sample": int(sample), "lr": float(lr), "loss": float(loss),
               "acc": float(acc), "time": t1 - t0, "time_start": t0, "time_finish": t1, "time_cost": t1 - t0}
        self.logs[split].append(rec)
        self.epoch = max(self.epoch, int(epoch))
        self._refresh()
        if save and self.path:
            self.save()
        return rec

    def finish(self, save: bool = True) -> None:
        """utils_stats.py:750-781: stamp the run complete."""
        starts = [r["time_start"] for s in self.splits for r in self.logs[s][:1]]
        if starts:
            self.telem["time_start"] = min(starts)
        self.telem["time_finish"] = self._clock()
        self.telem["completed"] = True
        self._refresh()
        if save and self.path:
            self.save()

    # ---- derived block ---------------------------------------------------------------
    def _refresh(self) -> None:
        t = self.telem
        if t["time_start"] and t["time_finish"]:
            t["time_elapsed"] = t["time_finish"] - t["time_start"]
        t["time_updated"] = self._clock()
        split_time, sample_time = {}, {}
        for s in self.splits:
            recs = self.logs[s]
            n = t.get(f"sample_count_{s}") or 1
            split_time[s] = 0.000001
            if recs:
                split_time[s] = float(np.mean([r["time_finish"] - r["time_start"] for r in recs]))
                n = float(np.mean([r["sample"] for r in recs]))
            sample_time[s] = split_time[s] / max(1, int(n))
        self.results = {
            "epochs": self.epoch,
            "epoch.time": sum(split_time.values()),
            "epoch.sample_time": 0.0,                          # never updated upstream either
            **{f"{s}.time": split_time[s] for s in self.splits},
            **{f"{s}.sample_time": sample_time[s] for s in self.splits},
            **{f"{s}.{k}": max([r[k] for r in self.logs[s]] + [d])
               for s in self.splits for k, d in (("acc", 0.0), ("loss", 9.999))},
        }

    @property
    def stats(self) -> dict:
        return {"info": dict(self.info), "telem": dict(self.telem), "results": dict(self.results),
                **{s: list(self.logs[s]) for s in self.splits}}

    def save(self, path: Optional[str] = None) -> bool:
        p = path or self.path
        assert isinstance(p, str) and p.endswith(".json")     # utils_stats.py:709-710
        d = os.path.split(p)[0]
        if d and not os.path.isdir(d):
            os.makedirs(d)
        with open(p, "w") as f:
            json.dump(self.stats, f, indent=4)
        return True

    def console_line(self, split: str) -> str:
        """One line per finished round: the fields of the reference's progress line
        (utils_stats.py:380-423) without the bar."""
        r = self.logs[split][-1]
        return (f"[{split:5s}] epoch {r['epoch']:3d}  sample {r['sample']:6d}  lr {r['lr']:.3e}  "
                f"loss {r['loss']:.4f}  acc {r['acc']:.4f}  time {r['time']:.1f}s")
