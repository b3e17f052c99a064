#!/bin/bash
# PMC profile of the DP training kernel (GPU box).  usage: tools/profile_pmc.sh <threads> <runs> <outdir>
# Counter passes are separate (SQ: 8 slots per pass); --pmc is never combined with other trace domains.
TH=${1:-1}; R=${2:-2048}; OUT=${3:-/root/repo/gpurun_out/pmc}
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" \
            "SQ_INSTS_VALU_TRANS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $pass --output-format csv -d $OUT/$tag -- python3 /root/repo/tools/probe_scaling.py $TH $R 100 > $OUT/$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 $OUT/$tag.log; }
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$OUT/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "dp_wave" in r["Kernel_Name"] or "dp_train" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc): print(f"{k:28s} n={len(acc[k])} mean={sum(acc[k])/len(acc[k]):.4g}")
PY
