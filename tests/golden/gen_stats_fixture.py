"""Copies one of the reference's own run logs (data, not code) as the golden vector for
vit_torch_amd.stats.RunLog: its train/val entries are the inputs, its `results` block the
expected output.  Run here (the reference is not on the GPU box):
    python tests/golden/gen_stats_fixture.py"""
import json
import os

SRC = "/root/reference/logs/massA/stats_210715_212442.json"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_stats_log.json")

if __name__ == "__main__":
    d = json.load(open(SRC))
    assert list(d) == ["info", "telem", "results", "train", "val"]
    json.dump(d, open(DST, "w"), indent=1)
    print("wrote", DST, len(d["train"]), "epochs")
