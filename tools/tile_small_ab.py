"""The CU-wide tile kernel on 1 M-row matrices (one round of the grid, 8 panels): pacing slack and chunk depth, K1 of a bicgstabStep."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import bench, form_tournament as ft
for name in (sys.argv[1:] or ["random_spd_1m", "rand100"]):
    z = ft.zoo(name)
    desc, (dims, (rp, ci, va)) = z if z else bench.workload(name)
    nnz = int(rp[-1])
    for label, o in (("default", {}), ("slack 0", {"tile_slack": 0}), ("slack 1", {"tile_slack": 1}), ("slack 2", {"tile_slack": 2}), ("slack 5", {"tile_slack": 5}),
                     ("12-group chunks", {"tile_depth": 1}), ("20-group chunks", {"tile_depth": 2}), ("2^16 panels", {"tile_shift": 16}), ("2^15 panels", {"tile_shift": 15})):
        r = bench.side_block(desc, dims, rp, ci, va, dict({"lpanel": 0, "lflat": 0}, **o), 40, 10)
        k1 = r["kernels"]["K1"]["ms"]
        print(f"{name:14s} {label:18s} {r['value']:8.1f} it/s  K1 {k1*1e3:6.1f} us = {(12*nnz+28*dims[0])/k1/1e6/8000:.3f}", flush=True)
