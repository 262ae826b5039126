"""A/B of the plain-CSR wave kernel's instantiations (option stream_wave = PRE * 1000 + OCC * 100 + PPL) on SHORT launches: where the persistent grid
gives a wavefront only a few blocks, the cross-block prefetch of (8, 3, prefetch) buys less than a fourth workgroup per CU."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import bench
import form_tournament as ft
from sla_amd import workloads as wl
for name in (sys.argv[1:] or ["e05_tiled", "lap100_plain", "lap064_plain", "lap128_plain", "lap160_plain"]):
    if name.startswith("lap"):
        g = int(name[3:6]); dims, (rp, ci, va) = wl.laplace3d(g, g, g); desc = name
    else:
        desc, (dims, (rp, ci, va)) = ft.zoo(name)
    nnz = int(rp[-1])
    base = {"wdia": 0, "vdict": 0, "diag": 0, "onchip": 0} if name.startswith("lap") else {}
    for rep in range(2):
        for code in ([int(c) for c in os.environ["WV_CODES"].split(",")] if os.environ.get("WV_CODES") else (1, 604, 408, 1308)):
            r = bench.side_block(desc, dims, rp, ci, va, dict(base, stream_wave=code), 60, 10)
            k1 = r["kernels"]["K1"]["ms"]
            print(f"{name:14s} code={code:5d}  {r['value']:8.1f} it/s  K1 {k1 * 1e3:6.1f} us = {(12 * nnz + 28 * dims[0]) / k1 / 1e6 / 8000:.3f}  K3 {r['kernels']['K3']['ms']*1e3:6.1f}", flush=True)
