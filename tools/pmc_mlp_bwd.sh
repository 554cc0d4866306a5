#!/bin/bash
# SQ counters of the MLP backward kernel, two passes (8 SQ slots each); run on the GPU box from the repo root.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_mlp_bwd
mkdir -p $OUT
cd /tmp
rocprofv3 -L > $OUT/avail.txt 2>&1
pick() { for c in "$@"; do grep -qw "$c" $OUT/avail.txt && echo -n "$c "; done; }
P1=$(pick SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU)
P2=$(pick SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS)
P3=$(pick SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_FLAT)
echo "P1=$P1"; echo "P2=$P2"; echo "P3=$P3"
i=1
for P in "$P1" "$P2" "$P3"; do
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pass$i -- python $R/tools/mlp_bwd_bench.py "$@" > $OUT/pass$i.log 2>&1
  i=$((i+1))
done
cd $R
python - <<'PY'
import csv, glob, collections, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_mlp_bwd"
tot = collections.OrderedDict()
for f in sorted(glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "mlp_bwd" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (v, n) in agg.items():
        tot[k] = v / n
for k, v in tot.items():
    print("%-34s %16.0f  (avg per launch)" % (k, v))
open(out + "/summary.txt", "w").write("\n".join("%-34s %16.0f" % kv for kv in tot.items()) + "\n")
PY
