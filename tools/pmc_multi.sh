#!/bin/bash
# several PMC passes over one command; prints (launches, mean per launch) for kernels matching $FILTER
# usage: FILTER=wdia tools/pmc_multi.sh tag "CTRS A" "CTRS B" ... -- cmd...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
sets=()
while [ "$1" != "--" ]; do sets+=("$1"); shift; done
shift
i=0
for set in "${sets[@]}"; do
  i=$((i+1)); OUT=gpurun_out/pmcm_${tag}_$i; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT -o p -- "$@" > $OUT/log 2>&1
  python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if "${FILTER:-spmv}" in k: print(k, {c: round(sum(v)/len(v)) for c, v in cs.items()})
PY
done
