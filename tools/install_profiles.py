#!/usr/bin/env python
"""Copies gpurun_out/refresh/* into profiles/ and rebuilds profiles/pmc_traffic.json from the PMC passes:
HBM bytes per launch of the dominant kernel (K1 = SpMV fused with the dot) = TCC_EA0_RDREQ x 128 B (all read
requests are 128-byte on gfx950; equals FETCH_SIZE x 2 KiB, see the calibration file) + WRITE_SIZE KiB x 1024."""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "refresh")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
for f in sorted(os.listdir(SRC)):
    if f.startswith(tag + "_") and os.path.isfile(os.path.join(SRC, f)):
        dst = f.replace("_bench_pmc_counters_laplace3d_10m", "_bench_pmc_counters")
        shutil.copy(os.path.join(SRC, f), os.path.join(DST, dst))
        print("installed", dst)


def k1_counters(path, patterns=(r"void sla::spmv_\w+<1[,>]",)):
    """(kernel name, {counter: mean per launch}) of the K1 launch in a pmc_kbench.sh summary: the EPI_DOT SpMV kernel,
    or -- for the forms that take two kernels per (#>) -- the sum over the listed kernels."""
    out, names = {}, []
    for line in open(path):
        m = re.match(r"p\d+ (void sla::\w+<.*?) (\{.*\})\s*$", line)
        if not m or not any(re.match(pat, m.group(1)) for pat in patterns):
            continue
        if m.group(1) not in names:
            names.append(m.group(1))
        for c, (cnt, mean) in eval(m.group(2)).items():
            out[c] = out.get(c, 0.0) + mean
    return " + ".join(names) if names else None, out


traffic = {"_comment": "HBM bytes per launch of the dominant kernel (K1: SpMV fused with the dot) from rocprofv3 PMC passes "
                       "(separate --pmc runs, kernel-trace only; tools/refresh_profiles.sh).  read = TCC_EA0_RDREQ x 128 B "
                       "(no 32-byte requests occur; FETCH_SIZE reads half of that on gfx950, see " + tag + "_kbench_pmc_calibration.txt), "
                       "write = WRITE_SIZE KiB x 1024.  The value-indexed kernels stream a compressed matrix, so the traffic is "
                       "below the algorithmic CSR figure.  bench.py copies the matching entry into roofline.traffic."}
import importlib.util
spec = importlib.util.spec_from_file_location("wl", os.path.join(ROOT, "sparse-linear-algebra_amd", "sla_amd", "workloads.py"))
for w, fname in (("laplace3d_10m", tag + "_bench_pmc_counters.txt"), ("poisson2d_1m", tag + "_bench_pmc_counters_poisson2d_1m.txt"),
                 ("random_spd_1m", tag + "_bench_pmc_counters_random_spd_1m.txt"),
                 ("dense_rows_200k", tag + "_bench_pmc_counters_dense_rows_200k.txt")):
    path = os.path.join(DST, fname)
    if not os.path.exists(path):
        continue
    if w == "dense_rows_200k":   # LDS-panel form: the panel sweep + the finish kernel that carries the fused dot
        name, c = k1_counters(path, (r"void sla::spmv_lpanel_kernel<", r"void sla::lpanel_finish_kernel<1[,>]"))
    else:
        name, c = k1_counters(path)
    if not c or "TCC_EA0_RDREQ_sum" not in c:
        continue
    rd = c["TCC_EA0_RDREQ_sum"] * 128 - c.get("TCC_EA0_RDREQ_32B_sum", 0) * 96
    wr = c["WRITE_SIZE"] * 1024
    bj = os.path.join(DST, tag + ("_bench_default.json" if w == "laplace3d_10m" else "_bench_%s.json" % w))
    algo, alg_bytes = "", None
    if os.path.exists(bj):
        rec = json.load(open(bj))
        algo = rec["config"]["spmv_kernel"].split()[0]
        alg_bytes = rec["roofline"]["bytes_per_launch"]
    traffic["%s/step/n1" % w] = {"kernel": name.replace("void ", "")[:120], "kernel_algo": algo, "read_bytes": int(rd), "write_bytes": int(wr),
                                 "traffic_bytes": int(rd + wr), "algorithmic_bytes": alg_bytes,
                                 "fetch_size_kib": c.get("FETCH_SIZE"), "write_size_kib": c.get("WRITE_SIZE"),
                                 "l2_hit": c.get("TCC_HIT_sum"), "l2_miss": c.get("TCC_MISS_sum"), "source": "profiles/" + fname}
json.dump(traffic, open(os.path.join(DST, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1)[:1500])
