"""Is the form the lowering picks the fastest one the library has?  One BiCGSTAB step per form on a workload, same box.
    python tools/form_tournament.py banded_2m|poisson2d_1m|laplace3d_10m|laplace3d_1m [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "banded_2m"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
desc, (dims, (rp, ci, va)) = bench.workload(name)
forms = (("default", {}), ("no march", {"wd_march": 0}), ("gather (wd_lds=0)", {"wd_lds": 0}), ("no wdia-vv", {"wdia_vv": 0}), ("no wdia", {"wdia": 0}),
         ("no wdia, no xwin", {"wdia": 0, "xwin": 0}), ("dictionary codes", {"wdia": 0, "vdict": 0}), ("dictionary codes + xwin", {"wdia": 0, "vdict": 0, "xwin": 2}),
         ("plain CSR", {"wdia": 0, "vdict": 0, "diag": 0}))
seen = set()
for label, opts in forms:
    r = bench.side_block(desc, dims, rp, ci, va, opts, steps, 10)
    algo = r["spmv_kernel"].split()[0]
    if algo in seen and label != "default":
        continue
    seen.add(algo)
    print(f"{name:14s} {label:26s} {r['value']:9.1f} it/s  " + "  ".join(f"{k} {v['ms'] * 1e3:.1f}" for k, v in r["kernels"].items()) + "  " + algo, flush=True)
