#!/bin/bash
# SQ counters of one kernel (substring match) of a command, three --pmc passes.  bash tools/pmc_sq.sh <kernel-substr> <tag> -- <cmd...>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=$1; TAG=$2; shift 3
OUT=$R/gpurun_out/pmc_sq_$TAG
mkdir -p $OUT
cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"
P3="SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM SQ_INSTS_GDS"
# PMC_SETS="A B C;D E" replaces the three SQ passes by other counter sets (one pass per ';'-separated group), e.g. the
# texture-addresser / vector-cache counters of a gather kernel
if [ -n "$PMC_SETS" ]; then IFS=';' read -r -a SETS <<< "$PMC_SETS"; else SETS=("$P1" "$P2" "$P3"); fi
i=1
for P in "${SETS[@]}"; do
  timeout 400 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pass$i -- "$@" > $OUT/pass$i.log 2>&1
  i=$((i+1))
done
cd $R
python - "$OUT" "$K" <<'PY'
import csv, glob, collections, sys
out, key = sys.argv[1], sys.argv[2]
tot = collections.OrderedDict()
for f in sorted(glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (v, n) in agg.items():
        tot[k] = v / n
txt = "\n".join("%-34s %16.0f" % kv for kv in tot.items())
open(out + "/summary.txt", "w").write("kernel filter: %s (mean per launch)\n%s\n" % (key, txt))
print(txt)
PY
