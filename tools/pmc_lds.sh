#!/bin/bash
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_lds_$c -- python3 $ROOT/bench.py --steps 1 --warmup 1 --roofline-steps 1 --no-lookahead --no-cpu-baseline --no-sharded --chains-per-gpu 0 --grad-steps 0 > $ROOT/gpurun_out/pmc_lds_$c.log 2>&1
  echo "$c exit $?"
done
python3 - <<'PY'
import csv, glob, collections, os
root = os.environ["GRAFT_REPO_ROOT"]
for c in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS"):
    fs = glob.glob(f"{root}/gpurun_out/pmc_lds_{c}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(c, "no file"); continue
    tot = collections.Counter(); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void migp::", "")[:44]
        tot[k] += float(r["Counter_Value"]); n[k] += 1
    for k in tot:
        if "gemm" in k or "strip" in k or "leaf" in k or "thin" in k:
            print(f"{c:24s} {k:42s} launches {n[k]:5d} total {tot[k]:.4g} per launch {tot[k]/n[k]:.4g}")
PY
