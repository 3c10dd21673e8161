"""HBM traffic per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the
same command, corrected as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes:
bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (on gfx950 FETCH_SIZE reports half the bytes of a
wide coalesced stream; WRITE_SIZE is exact for 16-B-per-lane stores).
usage: pmc_traffic.py <fetch_dir> <write_dir> [out.json]"""
import csv, glob, json, re, sys
from collections import defaultdict


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        k = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", k)
        k = re.split(r"[<(]|I[Lb]", k)[0] if k.startswith(("void ", "gemm", "ln_", "attn", "colsum", "cast", "patch")) else k
        k = k.replace("void ", "")
        tot[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return tot, cnt


fetch, nf = load(sys.argv[1], "FETCH_SIZE")
write, nw = load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(fetch, key=lambda k: -(2 * fetch[k] + write.get(k, 0))):
    n = nf[k]
    rd, wr = 2 * fetch[k] * 1024 / n, write.get(k, 0) * 1024 / max(nw.get(k, 1), 1)
    out[k] = {"launches": n, "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
              "hbm_bytes_per_launch": round(rd + wr)}
    print(f"{k[:60]:60s} n={n:5d}  read {rd/1e6:9.1f} MB  write {wr/1e6:9.1f} MB per launch")
# steps profiled = launches of the once-per-step optimizer kernel (lets a reader turn
# per-launch bytes into per-step bytes when one GEMM call is several launches)
steps = [v["launches"] for k, v in out.items() if k.startswith("sgd_momentum_kernel")]
# fingerprint of the kernel sources the counters were collected on: bench.py quotes `traffic` only from a file whose
# fingerprint equals that of the sources it runs (a stale file is refused, not quoted)
import hashlib, os
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit_torch_amd", "csrc")
h = hashlib.sha256()
for fn in sorted(os.listdir(csrc)):
    if fn.endswith((".hip", ".h", ".cpp")):
        h.update(fn.encode()); h.update(open(os.path.join(csrc, fn), "rb").read())
out["_meta"] = {"steps_profiled": steps[0] if steps else None, "kernel_sources_sha256": h.hexdigest()}
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
