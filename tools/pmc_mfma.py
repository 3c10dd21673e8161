"""MFMA-busy fraction per kernel from one rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE), as /opt/skills/guides/MI355X_MICROARCH.md reads
them: MFMA_BUSY counts busy cycles summed over the SIMDs that ran the kernel, GUI_ACTIVE is
summed over the 8 XCDs; busy fraction = MFMA_BUSY / (GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs).
usage: pmc_mfma.py <dir> [out.json]"""
import csv, glob, json, re, sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
val = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for r in csv.DictReader(open(f)):
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    k = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", k)
    k = re.split(r"[<(]|I[Lb]", k)[0].replace("void ", "")
    val[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[k] += 1
out = {}
for k, v in sorted(val.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
    act = v.get("GRBM_GUI_ACTIVE", 0.0)
    busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if act <= 0 or busy <= 0:
        continue
    frac = busy / (act / 8.0 * 256 * 4)
    out[k] = {"launches": cnt[k], "mfma_busy_cycles": busy, "gui_active_cycles_sum_xcd": act, "mfma_busy_frac": round(frac, 4)}
    print(f"{k[:60]:60s} n={cnt[k]:5d}  MFMA busy {100 * frac:6.2f} % of the matrix pipes' cycles")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
