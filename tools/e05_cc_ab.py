import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import bench
import form_tournament as ft
for name in ("e05_tiled_10m", "e05_tiled"):
    desc, (dims, (rp, ci, va)) = ft.zoo(name)
    nnz = int(rp[-1])
    for rep in range(2):
        for label, opts in (("default", {}), ("wave_cc=4", {"wave_cc": 4}), ("wave_cc=8", {"wave_cc": 8}), ("wave_cc=16", {"wave_cc": 16})):
            r = bench.side_block(desc, dims, rp, ci, va, opts, 60, 10)
            k1 = r["kernels"]["K1"]["ms"]
            print(f"{name:14s} {label:12s} {r['value']:8.1f} it/s  K1 {k1 * 1e3:6.1f} us = {(12 * nnz + 28 * dims[0]) / k1 / 1e6 / 8000:.3f}  K3 {r['kernels']['K3']['ms']*1e3:6.1f}", flush=True)
