"""Same-box A/B of spmv_wave_kernel's instantiations and grid sizes on the FEM-shaped matrix of the zoo (e05r0000 tiled to 1 M / 10 M rows):
K1 of a bicgstabStep per setting.   python tools/e05_wave_ab.py [e05_tiled|e05_tiled_10m]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import form_tournament as ft  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "e05_tiled"
desc, (dims, (rp, ci, va)) = ft.zoo(name)
nnz = int(rp[-1])
for label, opts in (("default", {}), ("8 pairs per lane, 4 workgroups per CU", {"stream_wave": 408}), ("4 pairs, 6 per CU", {"stream_wave": 604}),
                    ("2 pairs, 8 per CU", {"stream_wave": 208}), ("8 pairs, 3 per CU, next chunk prefetched", {"stream_wave": 1308}),
                    ("grid 512", {"spmv_grid": 512}), ("grid 640", {"spmv_grid": 640}), ("grid 1024", {"spmv_grid": 1024}), ("stream kernel", {"stream_wave": 0})):
    r = bench.side_block(desc, dims, rp, ci, va, opts, 60, 10)
    k1 = r["kernels"]["K1"]["ms"]
    print(f"{name:14s} {label:44s} {r['value']:8.1f} it/s  K1 {k1 * 1e3:6.1f} us = {(12 * nnz + 28 * dims[0]) / k1 / 1e6 / 8000:.3f} of peak  grid={[t for t in r['spmv_kernel'].split() if t.startswith('grid=')][0]}", flush=True)
