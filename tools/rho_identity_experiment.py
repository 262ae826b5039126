#!/usr/bin/env python
"""How much does evaluating rho_{j+1} = s . r0hat - omega (As . r0hat) instead of (s - omega As) . r0hat change a BiCGSTAB run,
compared with what a mere regrouping of the dot-product sums changes?  (CPU, numpy; the GPU path regroups sums anyway.)
Same problems, right-hand sides, x0 and stopping rule (true residual <= max(1e-6, 1e-4 |r0|), Sparse.hs:1034-1052) as
tests/test_gpu_parity.py::test_bicgstab_fused_k45_flow_vs_reference_split.  Prints iterations to convergence per variant and the
relative distance of the iterates after k steps.
usage: python tools/rho_identity_experiment.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sparse-linear-algebra_amd")):
    sys.path.insert(0, p)
import numpy as np
import scipy.sparse as sp
from sla_amd import workloads as wl
from oracle import oracle as orc


def dotk(k):
    """dot product summed in k contiguous chunks (0: numpy's own order)"""
    if k == 0:
        return lambda a, c: float(a @ c)

    def f(a, c):
        pr = a * c
        m = len(pr) // k * k
        return float(np.sum(pr[:m].reshape(k, -1).sum(axis=1)) + pr[m:].sum())
    return f


def run(A, b, x0, dot, identity, steps=None):
    r = b - A @ x0
    r0 = r.copy()
    tol = max(1e-6, 1e-4 * np.linalg.norm(r0))
    x, p, rho = x0.copy(), r.copy(), dot(r, r0)
    for it in range(1, (steps or 200) + 1):
        v = A @ p
        alpha = rho / dot(v, r0)
        s = r - alpha * v
        t = A @ s
        omega = dot(t, s) / dot(t, t)
        x = x + alpha * p + omega * s
        rn = s - omega * t
        rho_new = (dot(s, r0) - omega * dot(t, r0)) if identity else dot(rn, r0)
        beta = rho_new / rho * alpha / omega
        p, r, rho = rn + beta * (p - omega * v), rn, rho_new
        if steps is None and np.linalg.norm(A @ x - b) <= tol:
            return it, x
    return (steps or 200), x


probs = {"poisson2d 50x40": wl.poisson2d(50, 40), "laplace3d 14x11x13": wl.laplace3d(14, 11, 13),
         "spd 400": wl.random_spd(400, k=3, seed=77), "banded_nonsym 4001": wl.banded_nonsym(4001)}
chunks = (0, 2, 4, 7, 16, 64, 256)
for name, (dims, (rp, ci, va)) in probs.items():
    n = dims[0]
    A = sp.csr_matrix((va, ci, rp), shape=dims)
    Ao = orc.Csr(n, n, rp, ci, va)
    b = orc.spmv(Ao, np.linspace(-1.0, 2.0, n))
    x0 = np.full(n, 0.1)
    it_o = orc.linsolve0(orc.BICGSTAB_, Ao, b, x0)[2]
    ref = {k: run(A, b, x0, dotk(k), False)[0] for k in chunks}
    idn = {k: run(A, b, x0, dotk(k), True)[0] for k in chunks}
    print(f"{name}: oracle {it_o} iterations | reference formula, sums in k chunks {ref} | identity {idn}")
    for k in (5, 10, 20, 40):
        xa = run(A, b, x0, dotk(0), False, k)[1]
        xb = run(A, b, x0, dotk(0), True, k)[1]
        xc = run(A, b, x0, dotk(64), False, k)[1]
        nx = np.linalg.norm(xa)
        print(f"    after {k:2d} steps: identity vs reference {np.linalg.norm(xa - xb) / nx:.1e}   64-chunk sums vs reference {np.linalg.norm(xa - xc) / nx:.1e}")
