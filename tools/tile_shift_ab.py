import os, sys
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
import bench, form_tournament as ft
for name in ("random_spd_1m", "rand100", "rand200", "powerlaw"):
    z = ft.zoo(name)
    desc, (dims, (rp, ci, va)) = z if z else bench.workload(name)
    nnz = int(rp[-1])
    for sh in (17, 18, 19):
        r = bench.side_block(desc, dims, rp, ci, va, {"lpanel": 0, "lflat": 0, "tile_shift": sh}, 40, 10)
        k1 = r["kernels"]["K1"]["ms"]
        print(f"{name:14s} tile_shift={sh}  {r['value']:8.1f} it/s  K1 {k1*1e3:6.1f} us = {(12*nnz+28*dims[0])/k1/1e6/8000:.3f}  {' '.join(t for t in r['spmv_kernel'].split() if t.startswith(('algo','panels','panel_cols','slices')))}", flush=True)
