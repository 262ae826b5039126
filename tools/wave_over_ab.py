"""A/B of the plain-CSR wave kernel's grid (option wave_over): resident rounds of workgroups per launch.  Short launches (1 M rows:
2.5 blocks per wavefront at the persistent grid) leave whole wavefronts a block short in the last round; an oversubscribed grid lets the
dispatcher hand out the blocks as wavefronts retire.  Same box, interleaved passes; bytes = 12 per entry + 28 per row."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import bench
import form_tournament as ft
from sla_amd import workloads as wl
names = sys.argv[1:] or ["e05_tiled", "lap100_plain", "rand33", "e05_tiled_10m", "lap216_plain"]
for name in names:
    if name.startswith("lap"):
        g = int(name[3:6])
        dims, (rp, ci, va) = wl.laplace3d(g, g, g)
        desc = f"{name}: {dims[0]} rows as plain CSR"
    else:
        desc, (dims, (rp, ci, va)) = ft.zoo(name)
    nnz = int(rp[-1])
    base = {"wdia": 0, "vdict": 0, "diag": 0} if name.startswith("lap") else {}
    for rep in range(3):
        for over in (1, 2, 3, 4, 8):
            r = bench.side_block(desc, dims, rp, ci, va, dict(base, wave_over=over), 60, 10)
            k1 = r["kernels"]["K1"]["ms"]
            print(f"{name:14s} over={over}  {r['value']:8.1f} it/s  K1 {k1 * 1e3:6.1f} us = {(12 * nnz + 28 * dims[0]) / k1 / 1e6 / 8000:.3f}  "
                  f"K3 {r['kernels']['K3']['ms']*1e3:6.1f}  {r.get('kernel_info', '')[:60]}", flush=True)
