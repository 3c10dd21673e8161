"""MFMA-busy fraction per kernel from one rocprofv3 --kernel-trace --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE).
Two normalisations (the clock reconciliation of round 3):
 (a) busy / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) — round 2's figure; GRBM_GUI_ACTIVE reads HIGH on dispatches shorter than
     ~0.3 ms (every kernel of the step), so this UNDER-reads the busy fraction;
 (b) busy cycles per SIMD per second of kernel wall time (from the kernel trace) = clock x busy fraction, printed in MHz:
     divide by the in-kernel clock (`roofline.clock_mhz_under_load` of the bench line) for the fraction.
usage: pmc_mfma.py <dir> [out.json] [clock_mhz]"""
import csv, glob, json, re, sys
from collections import defaultdict


def short(k):
    k = re.sub(r"\(anonymous namespace\)::", "", k)
    k = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", k)
    return re.split(r"[<(]|I[Lb]", k)[0].replace("void ", "")


f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
clock = float(sys.argv[3]) if len(sys.argv) > 3 else None
val = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for r in csv.DictReader(open(f)):
    k = short(r["Kernel_Name"])
    val[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[k] += 1
wall = defaultdict(float)
if kt:
    for r in csv.DictReader(open(kt[0])):
        wall[short(r["Kernel_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
out = {}
for k, v in sorted(val.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
    act = v.get("GRBM_GUI_ACTIVE", 0.0)
    busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if act <= 0 or busy <= 0:
        continue
    frac = busy / (act / 8.0 * 256 * 4)
    rec = {"launches": cnt[k], "mfma_busy_cycles": busy, "gui_active_cycles_sum_xcd": act, "mfma_busy_frac_of_grbm": round(frac, 4)}
    line = f"{k[:48]:48s} n={cnt[k]:5d}  MFMA busy {100 * frac:6.2f} % of GRBM_GUI_ACTIVE/8 (reads low: short dispatches)"
    if wall.get(k):
        mhz = busy / 1024.0 / wall[k] / 1e6
        rec["busy_mhz"] = round(mhz, 1)
        line += f";  {mhz:7.1f} busy-MHz per SIMD"
        if clock:
            rec["mfma_busy_frac_at_clock"] = round(mhz / clock, 4)
            line += f" = {100 * mhz / clock:5.1f} % at the in-kernel clock of {clock:.0f} MHz"
    out[k] = rec
    print(line)
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
