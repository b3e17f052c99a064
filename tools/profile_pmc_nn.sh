#!/bin/bash
# SQ counter passes for the VAE-NN training kernel (GPU box): tools/profile_pmc_nn.sh <runs> <outdir>
R=${1:-256}; OUT=${2:-/root/repo/gpurun_out/pmc_nn}
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $pass --output-format csv -d $OUT/$tag -- python3 /root/repo/tools/probe_nn.py $R > $OUT/$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 $OUT/$tag.log; }
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$OUT/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "nn_train" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc): print(f"{k:28s} n={len(acc[k])} mean={sum(acc[k])/len(acc[k]):.4g}")
PY
