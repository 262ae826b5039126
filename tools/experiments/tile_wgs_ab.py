"""The CU-wide tile kernel with ONE workgroup per CU (slices of <= 19584 rows, 12 / 20-group chunks) against TWO (half-height slices, 8-group
chunks, 256 registers each: option tile_wgs = 2) on the 1 M-row matrices of the zoo and config 3a; K1 of a bicgstabStep, same box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)   # (ROOT = the repository: run from there after applying the patch)
import bench, form_tournament as ft
for name in (sys.argv[1:] or ["rand100", "rand200", "rand500", "powerlaw", "random_spd_1m", "random_spd_10m"]):
    z = ft.zoo(name)
    desc, (dims, (rp, ci, va)) = z if z else bench.workload(name)
    nnz = int(rp[-1])
    for label, o in (("one workgroup per CU", {}), ("two per CU", {"tile_wgs": 2})):
        for rep in range(2):
            r = bench.side_block(desc, dims, rp, ci, va, dict({"lpanel": 0, "lflat": 0}, **o), 40, 10)
            k1 = r["kernels"]["K1"]["ms"]
            print(f"{name:16s} {label:22s} {r['value']:8.1f} it/s  K1 {k1*1e3:7.1f} us = {(12*nnz+28*dims[0])/k1/1e6/8000:.3f}  {' '.join(t for t in r['spmv_kernel'].split() if t.startswith(('algo=', 'slices=', 'cu_slices')))}", flush=True)
