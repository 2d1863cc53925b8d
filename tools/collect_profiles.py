"""Copy the judged summaries of a tools/gpu_profile.sh run from gpurun_out/ into profiles/<tag>/.
python tools/collect_profiles.py [tag]"""
import csv, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r5"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles", tag)
os.makedirs(P, exist_ok=True)

def one(pattern):
    hits = glob.glob(os.path.join(G, pattern), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None

def pmc_rows(dirs):
    rows = []
    for d in dirs:
        f = one(f"{d}/**/*counter_collection.csv")
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            rows.append(dict(kernel=r["Kernel_Name"][:70], counter=r["Counter_Name"], value_KB=float(r["Counter_Value"]),
                             dur_ms=(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, grid=r["Grid_Size"],
                             wg=r["Workgroup_Size"], vgpr=r["VGPR_Count"], agpr=r["Accum_VGPR_Count"], lds=r["LDS_Block_Size"]))
    return rows

def write_pmc(name, dirs):
    """One row per (kernel, counter, launch geometry): launches, mean counter value (KB for FETCH_SIZE / WRITE_SIZE, raw
    counts otherwise) and mean duration -- the per-launch rows of one bench run are ~10 000 lines of repeats."""
    rows = pmc_rows(dirs)
    if not rows:
        return
    import math
    groups = {}
    for r in rows:
        # (launches of one kernel and grid can still be different problems -- the planned fp32 scan uses 512 workgroups for the
        # 50 000-query image-side search AND for a 7 000-query text-side one: a factor-of-two duration bucket keeps them apart)
        bucket = int(math.floor(math.log2(max(r["dur_ms"], 1e-6))))
        key = (r["kernel"], r["counter"], r["grid"], r["wg"], r["vgpr"], r["agpr"], r["lds"], bucket)
        groups.setdefault(key, []).append(r)
    out = []
    for key, rs in groups.items():
        out.append(dict(kernel=key[0], counter=key[1], value_KB=sum(r["value_KB"] for r in rs) / len(rs),
                        dur_ms=sum(r["dur_ms"] for r in rs) / len(rs), grid=key[2], wg=key[3], vgpr=key[4], agpr=key[5],
                        lds=key[6], launches=len(rs)))
    out.sort(key=lambda r: -r["dur_ms"] * r["launches"])
    with open(os.path.join(P, name), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(out[0].keys())); w.writeheader(); w.writerows(out)
    print(name, len(out), "rows (from", len(rows), "launch records)")

for src, dst in ((f"bench_{tag}.json", "bench_default.json"), (f"prof_knn_{tag}.json", "knn_1000000x768.json"),
                 (f"bench_vit-b-16_{tag}.json", "bench_vit-b-16.json"), (f"bench_vit-l-14_{tag}.json", "bench_vit-l-14.json"),
                 (f"bench_mscoco_{tag}.json", "bench_mscoco.json")):
    if os.path.exists(os.path.join(G, src)):
        shutil.copyfile(os.path.join(G, src), os.path.join(P, dst)); print(dst)
for d, dst in ((f"prof_bench_{tag}", "bench_default_kernel_stats.csv"), (f"prof_knn_{tag}", "knn_1000000x768_kernel_stats.csv"),
               (f"prof_vit-b-16_{tag}", "bench_vit-b-16_kernel_stats.csv"), (f"prof_vit-l-14_{tag}", "bench_vit-l-14_kernel_stats.csv"),
               (f"prof_mscoco_{tag}", "bench_mscoco_kernel_stats.csv")):
    f = one(f"{d}/**/*kernel_stats.csv")
    if f:
        shutil.copyfile(f, os.path.join(P, dst)); print(dst)
CS = ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES")
write_pmc("knn_1000000x768_pmc.csv", [f"pmc_knn_{c}_{tag}" for c in CS])
# the headline step's counters: everything in bench_default_pmc.csv; the hand-written GEMM's rows again in encoder_gemm_pmc.csv
# (what bench.py's roofline object reads) and the MFMA rows of all encoder kernels in encoder_mfma_pmc.csv
write_pmc("bench_default_pmc.csv", [f"pmc_bench_{c}_{tag}" for c in CS])
import csv as _csv
src = os.path.join(P, "bench_default_pmc.csv")
if os.path.exists(src):
    rows = list(_csv.DictReader(open(src)))
    for name, keep in (("encoder_gemm_pmc.csv", lambda r: "k_gemm_f16x3t" in r["kernel"]),
                       ("encoder_mfma_pmc.csv", lambda r: r["counter"] in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES")
                                                          and not r["kernel"].startswith(("void (anonymous namespace)::k_scan", "void k_neighbors")))):
        sel = [r for r in rows if keep(r)]
        if sel:
            with open(os.path.join(P, name), "w", newline="") as f:
                w = _csv.DictWriter(f, fieldnames=list(sel[0].keys())); w.writeheader(); w.writerows(sel)
            print(name, len(sel), "rows")
for a in ("vit-b-16", "vit-l-14", "mscoco"):
    write_pmc(f"bench_{a}_pmc.csv", [f"pmc_{a}_{c}_{tag}" for c in CS])
