#!/bin/bash
# One PMC evidence file (the pmc() passes of tools/refresh_profiles.sh) for one command:
#   gpurun -- 'bash tools/pmc_one.sh gpurun_out/x.txt python bench.py --workload random_spd_10m --steps 8 --warmup 2 --no-cpu-baseline'
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
S=gpurun_out/pmc_one; mkdir -p $S
pmc() { out=$1; shift
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    d=$S/pmc_tmp; rm -rf $d; mkdir -p $d
    timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -o p -- "$@" > /dev/null 2>&1
    python - "$d" <<'PY' >> $out
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    # (a kernel name can cover launches of different sizes -- e.g. spmv_tile_kernel<1> on a second, smaller matrix of the same run: the
    # figure per launch is the mean over the launches within 10 % of the largest, i.e. the full-size ones; n = how many those were)
    for k, cs in agg.items():
        out = {}
        for c, v in cs.items():
            big = [x for x in v if x >= 0.9 * max(v)] if max(v) > 0 else v
            out[c] = (len(big), sum(big) / len(big))
        print(k, out)
PY
  done; }
out=$1; shift
rm -f "$out"
pmc "$out" "$@"
