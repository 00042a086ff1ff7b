#!/bin/bash
# PMC breakdown of one trapezoid GEMM launch: tools/pmc_trap.sh m n k
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_t1 -- python3 $ROOT/tools/dev_gemm_trap.py $1 $2 $3 > $ROOT/gpurun_out/pmc_t1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_t2 -- python3 $ROOT/tools/dev_gemm_trap.py $1 $2 $3 > $ROOT/gpurun_out/pmc_t2.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("pmc_t1", "pmc_t2"):
    f = glob.glob("$ROOT/gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True)
    kt = glob.glob("$ROOT/gpurun_out/%s/**/*kernel_trace.csv" % d, recursive=True)
    if not f: print("no counters in", d); continue
    dur = {}
    for r in csv.DictReader(open(kt[0])): dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        key = (r["Dispatch_Id"], r["Kernel_Name"][:50])
        agg.setdefault(key, {})[r["Counter_Name"]] = agg.setdefault(key, {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    last = {}
    for (did, name), c in agg.items():
        if "gemm" in name.lower(): last[name] = (did, c)
    for name, (did, c) in last.items():
        print(name, "dur_us=%.1f" % (dur.get(did, 0) / 1e3), {k: "%.4g" % v for k, v in c.items()})
PY
