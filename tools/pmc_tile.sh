#!/bin/bash
# L1 / L2 / LDS counters of the tile kernels on config 3a (one rocprofv3 pass per counter set, kernel-trace only): mean per launch.
#   usage: FILTER=tile tools/pmc_tile.sh tag -- cmd...      (round 5; a short list of tools/pmc_sets.sh's sets)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift; shift
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS"; do
  i=$((i+1)); OUT=gpurun_out/pmct_${tag}_$i; rm -rf $OUT; mkdir -p $OUT
  timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT -o p -- "$@" > $OUT/log 2>&1
  python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if "${FILTER:-tile_kernel}" in k: print(k[:48], {c: round(sum(v)/len(v)) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
PY
  rm -rf $OUT
done
