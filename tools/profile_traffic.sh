#!/bin/bash
# HBM traffic of the bench's dominant kernel from the PMC counters (GPU box): two separate rocprofv3 passes (--pmc only, no other
# trace domain), FETCH_SIZE doubled per the gfx950 correction.  Writes gpurun_out/<dir>/pmc_traffic.json (+ the per-dispatch CSV rows).
OUT=/root/repo/gpurun_out/${1:-pmc_traffic}
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -o pmc -- python3 /root/repo/bench.py --steps 5 --warmup 1 --min-seconds 0 --no-parity --no-extras --no-cpu-baseline > $OUT/$c.log 2>&1 || { echo "pass $c failed"; tail -3 $OUT/$c.log; exit 1; }
done
python3 - <<PY
import csv, glob, json
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$OUT/%s/*counter_collection.csv" % c)[0]
    rows = [r for r in csv.DictReader(open(f)) if "dp_wave_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c]
    w = csv.DictWriter(open("$OUT/pmc_%s_dp_wave_kernel.csv" % c.lower().split("_")[0], "w"), fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
    v = [float(r["Counter_Value"]) for r in rows]
    vals[c] = sum(v) / len(v)
rd, wr = 2 * vals["FETCH_SIZE"] * 1024, vals["WRITE_SIZE"] * 1024
algo = 176 * 8192 * 10000
kname = sorted({r["Kernel_Name"] for r in rows})[0]
kname = kname.replace("void ", "").split("(")[0]
json.dump({"runs": 8192, "threads": 0, "kernel": kname, "FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"], "hbm_read_bytes_per_launch": rd,
           "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": algo, "ratio": (rd + wr) / algo,
           "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_traffic.sh) on python3 bench.py --steps 5 --warmup 1 "
                   "--min-seconds 0 --no-parity --no-extras --no-cpu-baseline; FETCH_SIZE x2 per the gfx950 correction"}, open("$OUT/pmc_traffic.json", "w"), indent=1)
print(open("$OUT/pmc_traffic.json").read())
PY
