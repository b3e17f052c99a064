#!/bin/bash
# SQ counter profile of one training kernel (GPU box):   tools/profile_pmc.sh <dp|flex|awgn> [runs] [outdir]
#   dp   = the bench kernel (VAE-LE, B = 100, 100 steps per launch)      tools/probe_scaling.py 1 <runs> 100
#   flex = the VAEflex launch (window 100, stride 10, 990 steps)          tools/probe_flex.py <runs>
#   awgn = the AWGN training kernel (B = 350, 30 steps per launch)        tools/probe_awgn.py <runs> 30 0
#   epi  = the compact DP epilogue (one workgroup per run)                tools/probe_epilogue.py <runs>   ("step" = the whole kernel)
# Counter passes are separate rocprofv3 runs (SQ: 8 slots per pass); --pmc is never combined with another trace domain.  Prints (and writes to
# <outdir>/summary.txt) the raw per-launch means and the figures DESIGN.md quotes (per wave-step counts, VALU / LDS busy, bank conflicts, waits).
MODE=${1:-dp}; R=${2:-8192}; OUT=${3:-/root/repo/gpurun_out/pmc_$MODE}
case $MODE in
  dp)   CMD="/root/repo/tools/probe_scaling.py 1 $R 100"; PAT="dp_wave_kernel|dp_train_kernel"; STEPS=100;;
  flex) CMD="/root/repo/tools/probe_flex.py $R"; PAT="dp_wave_kernel|dp_train_kernel"; STEPS=990;;
  awgn) CMD="/root/repo/tools/probe_awgn.py $R 30 0"; PAT="awgn_wave_kernel|awgn_train_kernel"; STEPS=30;;
  epi)  CMD="/root/repo/tools/probe_epilogue.py $R"; PAT="dp_epilogue_compact_kernel<8, true>"; STEPS=1;;
  *) echo "dp|flex|awgn|epi"; exit 1;;
esac
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" \
            "SQ_INSTS_VALU_TRANS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d $OUT/$tag -- python3 $CMD > $OUT/$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 $OUT/$tag.log; }
done
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections, re
acc, names = collections.defaultdict(list), collections.Counter()
for f in glob.glob("$OUT/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if re.search(r"$PAT", r["Kernel_Name"]):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"])); names[r["Kernel_Name"]] += 1
print("# tools/profile_pmc.sh $MODE $R: python3 $CMD; means over the launches of", ", ".join(sorted(names)))
m = {k: sum(v) / len(v) for k, v in acc.items()}
for k in sorted(m): print(f"{k:28s} n={len(acc[k])} mean={m[k]:.4g}")
ws = m.get("SQ_WAVES", 0) * $STEPS
if ws:
    g = lambda k: m.get(k, float("nan"))
    print(f"# derived: per wave-step ({g('SQ_WAVES'):.0f} waves x $STEPS steps): VALU {g('SQ_INSTS_VALU') / ws:.0f} instr, SALU {g('SQ_INSTS_SALU') / ws:.0f}, "
          f"LDS {g('SQ_INSTS_LDS') / ws:.0f} instr = {g('SQ_LDS_IDX_ACTIVE') / ws:.0f} LDS cycles, VMEM wr {g('SQ_INSTS_VMEM_WR') / ws:.1f} rd {g('SQ_INSTS_VMEM_RD') / ws:.1f}")
    print(f"# derived: VALU active / wave cycles {g('SQ_ACTIVE_INST_VALU') / g('SQ_WAVE_CYCLES'):.3f} (x waves per SIMD = SIMD VALU busy); waves waiting (s_waitcnt) {g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.3f}, "
          f"issue-stalled {g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.3f}; LDS bank conflicts / LDS cycles {g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE'):.3f}")
PY
