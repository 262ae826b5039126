"""Same-box A/B of the plain CSR (#>) kernels (options wdia=0 vdict=0 diag=0) on the 216^3 Laplacian (the `general_csr` block of
bench.py) and on a random 1 M-row matrix: spmv_wave_kernel (wavefront-private 128-row blocks, row-pair stores; chunks of 4 or 7 entry
pairs per lane) against spmv_stream_kernel.  Interleaved repetitions; prints K1 / K3 / step timings.
    python tools/wave_ab.py [steps] [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
# WAVE_VARIANTS="name=code[:option=value...],..."  (code = PRE * 1000 + OCC * 100 + PPL of an instantiation in sla_spmv_wave.hip)
variants = [("stream", {"stream_wave": 0})]
for t in os.environ.get("WAVE_VARIANTS", "auto=1,wave=408,wave=1308,wave=604").split(","):
    f = t.split(":")
    opts = {"stream_wave": int(f[0].split("=")[1])}
    for kv in f[1:]:
        k, v = kv.split("=")
        opts[k] = int(v)
    variants.append((t, opts))
desc, (dims, (rp, ci, va)) = bench.workload("laplace3d_10m")
base = {"wdia": 0, "vdict": 0, "diag": 0}
for rep in range(reps):
    for name, extra in variants:
        r = bench.side_block(desc, dims, rp, ci, va, dict(base, **extra), steps, 5)
        print(f"laplace3d_10m {name:22s} {r['value']:8.1f} it/s  K1 {r['k1_ms'] * 1e3:7.1f} us ({r['k1_frac']:.3f})  "
              + "  ".join(f"{k} {v['ms'] * 1e3:.1f}" for k, v in r["kernels"].items()) + "  " + " ".join(r["spmv_kernel"].split()[:2]), flush=True)
del rp, ci, va
desc, (dims, (rp, ci, va)) = bench.workload("random_spd_1m")
for rep in range(reps):
    for name, extra in variants:
        r = bench.side_block(desc, dims, rp, ci, va, dict({"tiles": 0, "panels": 0}, **extra), steps, 5, rhs="A.x*")
        print(f"random_spd_1m {name:22s} {r['value']:8.1f} it/s  K1 {r['k1_ms'] * 1e3:7.1f} us ({r['k1_frac']:.3f})  "
              + "  ".join(f"{k} {v['ms'] * 1e3:.1f}" for k, v in r["kernels"].items()) + "  " + " ".join(r["spmv_kernel"].split()[:2]), flush=True)
