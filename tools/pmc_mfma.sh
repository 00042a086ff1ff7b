#!/bin/bash
# MFMA utilisation of the GEMM kernels from counters (BASELINE config 3: "rocprof MFMA util"): one --pmc pass per workload,
#   util = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs),  effective clock = GRBM_GUI_ACTIVE / 8 / duration
# (MI355X_MICROARCH.md: MFMA_BUSY counts cycles, rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs).  Dispatch is serialised under
# --pmc: kernels run alone on the chip (the handle finds that out at create and uses event edges).   tools/pmc_mfma.sh <tag>
tag=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
C="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_mfma_lml -- python3 $ROOT/bench.py --steps 2 --warmup 1 --roofline-steps 1 --no-lookahead --no-cpu-baseline --no-sharded --chains-per-gpu 0 --grad-steps 0 > $OUT/pmc_mfma_lml.log 2>&1
echo "lml pass exit $?"
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_mfma_grad -- python3 $ROOT/tools/trace_n.py 16384 16 grad > $OUT/pmc_mfma_grad.log 2>&1
echo "grad pass exit $?"
python3 - "$OUT" "$tag" <<'PY'
import csv, glob, collections, sys
out, tag = sys.argv[1], sys.argv[2]
lines = []
for name, what in (("pmc_mfma_lml", "bench.py --no-lookahead (Matern-5/2 N=16384 d=16, LML: every kernel alone on the chip)"),
                   ("pmc_mfma_grad", "tools/trace_n.py 16384 16 grad (RBF N=16384 d=16, LML + gradient)")):
    cc = glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True)
    kt = glob.glob(f"{out}/{name}/**/*kernel_trace.csv", recursive=True)
    if not cc:
        lines.append(f"{name}: no counter file"); continue
    dur = {}
    if kt:
        for r in csv.DictReader(open(kt[0])):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    tot = collections.defaultdict(lambda: collections.Counter())
    n = collections.Counter(); secs = collections.Counter(); seen = set()
    for r in csv.DictReader(open(cc[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void migp::", "").replace("migp::", "")[:52]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); n[k] += 1; secs[k] += dur.get(r["Dispatch_Id"], 0.0)
    lines.append(f"== {what}")
    lines.append(f"{'kernel':52s} {'launches':>8s} {'ms total':>9s} {'MFMA util':>9s} {'eff. clock':>10s} {'wait_inst/wave':>14s} {'wait_any/wave':>13s}")
    for k in sorted(tot, key=lambda k: -secs[k]):
        t = tot[k]
        if t["SQ_VALU_MFMA_BUSY_CYCLES"] <= 0 or t["GRBM_GUI_ACTIVE"] <= 0:
            continue
        util = (t["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (t["GRBM_GUI_ACTIVE"] / 8.0)
        clk = t["GRBM_GUI_ACTIVE"] / 8.0 / secs[k] * 1e-9 if secs[k] > 0 else float("nan")
        wc = t["SQ_WAVE_CYCLES"] or float("nan")
        lines.append(f"{k:52s} {n[k]:8d} {secs[k] * 1e3:9.2f} {util:9.3f} {clk:8.2f}GHz {t['SQ_WAIT_INST_ANY'] / wc:14.3f} {t['SQ_WAIT_ANY'] / wc:13.3f}")
open(f"{out}/{tag}_pmc_mfma_util.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
