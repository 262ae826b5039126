#!/bin/bash
# one PMC pass: tools/pmc_one.sh <tag> "<counters>" <cmd...> ; prints (launches, mean) per SpMV kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; ctr=$2; shift 2
OUT=gpurun_out/pmc1_$tag; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT -o p -- "$@" > $OUT/log 2>&1
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if "spmv" in k: print("$tag", k, {c: (len(v), round(sum(v)/len(v))) for c, v in cs.items()})
PY
