"""Per-kernel summary of rocprofv3 --pmc passes: python tools/pmc_summary.py DIR [DIR...]  (DIR holds *counter_collection.csv).
One line per (kernel, grid): launches, mean duration, and for every counter its mean; derived: MFMA-busy fraction
(SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024): 4 SIMDs x 256 CUs) and shader clock (GRBM_GUI_ACTIVE / 8 / duration)."""
import csv, glob, os, sys
from collections import defaultdict

rows = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            key = (r["Kernel_Name"][:60], r["Grid_Size"])
            rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            rows[key]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = []
for (k, g), c in rows.items():
    dur = sum(c["_dur_us"]) / len(c["_dur_us"])
    n = len(next(v for kk, v in c.items() if kk != "_dur_us"))
    line = dict(kernel=k, grid=g, launches=n, dur_us=dur)
    for name, v in c.items():
        if name != "_dur_us":
            line[name] = sum(v) / len(v)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in line and "GRBM_GUI_ACTIVE" in line:
        line["mfma_busy"] = line["SQ_VALU_MFMA_BUSY_CYCLES"] / (line["GRBM_GUI_ACTIVE"] / 8 * 1024)
        line["clock_GHz"] = line["GRBM_GUI_ACTIVE"] / 8 / dur / 1e3
    out.append(line)
out.sort(key=lambda r: -r["dur_us"] * r["launches"])
for r in out[: int(os.environ.get("TOP", "30"))]:
    extra = " ".join(f"{k}={v:.4g}" for k, v in r.items() if k not in ("kernel", "grid", "launches", "dur_us"))
    print(f"{r['kernel']:<60} grid {r['grid']:>9} x{r['launches']:<5} {r['dur_us']:9.1f} us  {extra}")
