#!/bin/bash
# PMC passes (one counter group per run, kernel-trace only) over the kNN-only workload.
# usage: tools/pmc_knn.sh TAG "bench args"
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=$1; shift
ARGS="$@"
cd /tmp && export TMPDIR=/tmp
i=0
for G in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $OUT/pmc_${TAG}_$i -- python3 $R/bench.py $ARGS > /dev/null 2> $OUT/pmc_${TAG}_$i.err || echo "group $i failed"
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); dur=collections.defaultdict(list)
for f in glob.glob("$OUT/pmc_${TAG}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "k_scan" not in k and "finalize" not in k and "merge" not in k: continue
        k=k.split("(")[0][-40:]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if r["Counter_Name"] in ("FETCH_SIZE",): dur[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
with open("$OUT/pmc_${TAG}_summary.txt","w") as o:
    for k,v in agg.items():
        o.write(k+"  launches_ms="+str([round(x,1) for x in dur[k]])+"\n")
        for c,x in sorted(v.items()): o.write(f"   {c:28s} {x:.4g}\n")
print(open("$OUT/pmc_${TAG}_summary.txt").read())
PY
