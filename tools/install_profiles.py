#!/usr/bin/env python
"""Copies gpurun_out/refresh/* into profiles/ and rebuilds profiles/pmc_traffic.json from the PMC passes: HBM bytes per
launch of every kernel of the BiCGSTAB step (K1..K5) = TCC_EA0_RDREQ x 128 B (all read requests are 128-byte on gfx950;
equals FETCH_SIZE x 2 KiB, see r01_kbench_pmc_calibration.txt) + WRITE_SIZE KiB x 1024.  A logical K1 / K3 that takes
several launches (column-panel passes: P - 1 launches of spmv_stream_kernel<0> before the fused last pass; LDS panels:
the panel sweep before the finish kernel) is SUMMED over its launches.  K3 is the <2, ...> (EPI_DOT2) instantiation in the reference's split flow and the
<7, ...> (EPI_DOT4: also As . r0hat, s . r0hat) one in the fused single-rank flow; a run holds one of them."""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "refresh")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
for f in sorted(os.listdir(SRC)):
    if f.startswith(tag + "_") and os.path.isfile(os.path.join(SRC, f)):
        shutil.copy(os.path.join(SRC, f), os.path.join(DST, f))
        print("installed", f)


def counters(path):
    """{kernel name: {counter: (launches, mean per launch)}} of a refresh_profiles.sh PMC summary."""
    out = {}
    for line in open(path):
        m = re.match(r"(void sla::\w+<.*?|void sla::\w+\(.*?) (\{.*\})\s*$", line)
        if m:
            out.setdefault(m.group(1), {}).update(eval(m.group(2)))
    return out


def bytes_of(c):
    rd = c["TCC_EA0_RDREQ_sum"][1] * 128 - c.get("TCC_EA0_RDREQ_32B_sum", (0, 0))[1] * 96
    return rd, c["WRITE_SIZE"][1] * 1024


KERNELS = {"K1": r"void sla::spmv_\w+<1[,>]", "K3": r"void sla::spmv_\w+<[27][,>]", "K2": r"void sla::bicg_k2_kernel",
           "K4": r"void sla::bicg_k4_kernel", "K5": r"void sla::bicg_k5_kernel", "K45": r"void sla::bicg_k45_kernel"}
traffic = {"_comment": "HBM bytes per launch of each kernel of the timed BiCGSTAB step from rocprofv3 PMC passes (separate --pmc runs, "
                       "kernel-trace only; tools/refresh_profiles.sh).  read = TCC_EA0_RDREQ x 128 B (no 32-byte requests occur; "
                       "FETCH_SIZE reads half of that on gfx950), write = WRITE_SIZE KiB x 1024.  Memory-side-cache hits are counted, "
                       "not excluded.  bench.py copies the entry of its dominant kernel into roofline.traffic."}
for w, fname, bench in (("laplace3d_10m", tag + "_bench_pmc_counters.txt", tag + "_bench_default.json"),
                        ("random_spd_10m", tag + "_bench_pmc_counters_random_spd_10m.txt", tag + "_bench_random_spd_10m_bicgstab.json"),
                        ("poisson2d_1m", tag + "_bench_pmc_counters_poisson2d_1m.txt", tag + "_bench_poisson2d_1m.json"),
                        ("dense_rows_200k", tag + "_bench_pmc_counters_dense_rows_200k.txt", tag + "_bench_dense_rows_200k.json")):
    path = os.path.join(DST, fname)
    if not os.path.exists(path):
        continue
    cs = counters(path)
    rec = json.load(open(os.path.join(DST, bench))) if os.path.exists(os.path.join(DST, bench)) else {}
    kinfo = rec.get("config", {}).get("spmv_kernel", "")
    algo = kinfo.split()[0] if kinfo else ""
    panels = int(kinfo.split("col_panels=")[1].split()[0]) if "col_panels=" in kinfo else 1
    for kid, pat in KERNELS.items():
        names = [k for k in cs if re.match(pat, k) and "TCC_EA0_RDREQ_sum" in cs[k] and "WRITE_SIZE" in cs[k]]
        if not names:
            continue
        rd, wr = bytes_of(cs[names[0]])
        parts = [names[0]]
        if kid in ("K1", "K3"):
            if panels > 1:   # the P - 1 unfused column-panel passes that precede the fused last pass of one logical SpMV
                pre = [k for k in cs if re.match(r"void sla::spmv_stream_kernel<0[,>]", k) and "WRITE_SIZE" in cs[k]]
                if pre:
                    r0, w0 = bytes_of(cs[pre[0]])
                    rd, wr = rd + (panels - 1) * r0, wr + (panels - 1) * w0
                    parts.append(f"{panels - 1} x {pre[0]}")
            if "ldspanels" in algo:   # the LDS-panel sweep feeding the finish kernel
                pre = [k for k in cs if "spmv_lpanel_kernel" in k and "WRITE_SIZE" in cs[k]]
                fin = [k for k in cs if re.match(r"void sla::lpanel_finish_kernel<%d[,>]" % (1 if kid == "K1" else 2), k) and "WRITE_SIZE" in cs[k]]
                if pre and fin:
                    r0, w0 = bytes_of(cs[pre[0]])
                    r1, w1 = bytes_of(cs[fin[0]])
                    rd, wr, parts = r0 + r1, w0 + w1, [pre[0], fin[0]]
        if kid == "K3" and "K23" in rec.get("kernels", {}) and "K3" not in rec.get("kernels", {}):
            kid = "K23"   # (bicg_fuse23: K2 folded into the plane-march K3)
        krec = rec.get("kernels", {}).get(kid, {})
        traffic["%s/step/n1/%s" % (w, kid)] = {
            "kernel": " + ".join(p.replace("void ", "")[:100] for p in parts), "kernel_algo": algo if kid in ("K1", "K3", "K23") else "",
            "read_bytes": int(rd), "write_bytes": int(wr), "traffic_bytes": int(rd + wr),
            "bytes_streamed_by_design": krec.get("bytes"), "csr_bytes": krec.get("csr_bytes"),
            "l2_hit": cs[names[0]].get("TCC_HIT_sum", (0, None))[1], "l2_miss": cs[names[0]].get("TCC_MISS_sum", (0, None))[1],
            "source": "profiles/" + fname}
json.dump(traffic, open(os.path.join(DST, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1)[:3000])
