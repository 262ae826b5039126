#!/usr/bin/env python
"""One markdown table of a round's bench lines (profiles/<tag>_bench_*.json, written by tools/refresh_profiles.sh on ONE box): the
rows of DESIGN.md section 5 / BASELINE.md -- every figure read from the JSON the driver-shaped command printed, none typed by hand.
    python tools/round_table.py r05"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"


def load(name):
    p = os.path.join(ROOT, "profiles", f"{tag}_bench_{name}.json")
    try:
        txt = open(p).read().strip()
        return json.loads(txt) if txt else None
    except (OSError, ValueError):
        return None


def kern(d, k):
    v = (d.get("kernels") or {}).get(k)
    return f"{v['ms'] * 1e3:.1f} µs, {v['frac']:.2f}" if v else "—"


def row(label, d, extra=""):
    if not d or not d.get("value"):
        return f"| {label} | — | — | — | — | (no line) |"
    r = d.get("roofline", {})
    dom = r.get("kernel", "").split(":")[0]
    domtxt = f"{dom} {r.get('avg_launch_ms', 0) * 1e3:.1f} µs = {r.get('frac', 0):.2f}" if dom else "—"
    if r.get("traffic"):
        domtxt += f"; PMC {r['traffic'] / 1e6:.0f} MB = {r['traffic'] / max(r.get('bytes_per_launch', 1), 1):.3f} ×"
    k1 = kern(d, "K1") if "K1" in (d.get("kernels") or {}) else kern(d, "C1")
    return f"| {label} | {d['ms_per_step'] * 1e3:.1f} µs | **{d['value']:.0f}** | {k1} | {domtxt} | {extra} |"


print(f"| config (`profiles/{tag}_bench_*.json`) | time/iter | iters/s | K1 (SpMV + dot): time, fraction of 8 TB/s on the bytes it streams | dominant kernel | notes |")
print("|---|---|---|---|---|---|")
d = load("default")
if d:
    cb = d.get("cpu_baseline", {})
    e = d.get("end_to_end", {})
    print(row("4 (10 M 7-pt, 216³) bicgstabStep — the driver's command", d,
              (f"K23 (K2 folded into K3) {kern(d, 'K23')}" if 'K23' in (d.get('kernels') or {}) else f"K2 {kern(d, 'K2')}; K3 {kern(d, 'K3')}") + f"; K45 {kern(d, 'K45')}; measured stream ceiling {d.get('hbm_measured_ceiling_gbps', 0) / 1e3:.2f} TB/s"))
    g = d.get("general_csr") or {}
    if g.get("value"):
        ks = g["kernels"]
        print(f"| same matrix as plain CSR (`general_csr`: `wdia=0 vdict=0 diag=0`) | {g['ms_per_step'] * 1e3:.0f} µs | **{g['value']:.0f}** | "
              f"{g['k1_ms'] * 1e3:.1f} µs = {g['k1_csr_gbps'] / 1e3:.2f} TB/s = **{g['k1_csr_frac']:.3f}** (top-level `csr_spmv_frac`) | K3 {ks['K3']['ms'] * 1e3:.1f} µs, {ks['K3']['frac']:.2f} | "
              f"{g['spmv_kernel'].split()[0]}; lowered in {g['lowered_once']['from_csr_s']:.3f} s |")
    r3 = d.get("random_spd_10m") or {}
    if r3.get("value"):
        print(f"| **3a (10 M random SPD, 33 per row)** — `random_spd_10m` block of the same line | {r3['ms_per_step']:.2f} ms | **{r3['value']:.1f}** | "
              f"**{r3['k1_ms']:.3f} ms = {r3['k1_csr_frac']:.3f}** on CSR bytes (`north_star_target`) | K1 / K3 | "
              f"{' '.join(t for t in r3['spmv_kernel'].split() if t.startswith(('algo=', 'exact_fold', 'cu_slices', 'row_owned', 'slices', 'panels')))}; lowered in {r3['lowered_once']['from_csr_s']:.3f} s"
              + (f"; gather-fabric ceiling {d['north_star_target']['ceiling']['k1_ms_at_ceiling']:.2f} ms: K1 at {d['north_star_target']['ceiling']['frac_of_ceiling']:.2f} of it"
                 if (d.get('north_star_target') or {}).get('ceiling') else "") + " |")
    rx = (d.get("north_star_target") or {}).get("relaxed_order_opt_in") or {}
    if rx.get("iters_per_s"):
        cx = rx.get("ceiling") or {}
        print(f"| … 3a with the OPT-IN `tile_relaxed = 1` (row sums in relaxed order: within the rounding bound, not reproducible bit for bit) — same line | | **{rx['iters_per_s']:.1f}** | "
              f"{rx['k1_ms']:.3f} ms = {rx['k1_frac']:.3f} on CSR bytes | K1 / K3 | {rx['spmv_kernel']}"
              + (f"; gather-fabric ceiling {cx['k1_ms_at_ceiling']:.2f} ms: K1 at {cx['frac_of_ceiling']:.2f} of it" if cx else "") + " |")
    if cb:
        print(f"| 4, CPU oracle port (same run, host cores of the GPU box) | | {cb.get('value', 0):.2f} ({cb.get('cores')} thread) / "
              f"{(cb.get('omp') or {}).get('value', 0):.1f} ({(cb.get('omp') or {}).get('cores')} threads OpenMP) | | | {cb.get('sample', '')[:120]} |")
    if e.get("from_csr_s"):
        ls = e.get("linsolve0", {})
        print(f"| 4, whole `linSolve0` call from host arrays (`end_to_end`) | | | | | `from_csr` {e['from_csr_s']:.3f} s; cold call {e.get('cold_linsolve0_s', 0):.3f} s = "
              f"{e.get('cold_over_solve', 0):.2f} × its solve time; `from_coo` {e.get('from_coo_s', 0):.3f} s; {ls.get('iters_per_s_incl_true_residual', 0):.0f} it/s incl. the true residual |")
print(row("4, cgsStep", load("cgs")))
print(row("4, linSolve0 iteration (true residual every iteration)", load("linsolve0")))
def onchip_row(label, name):
    """round 6: sla_solver_step(k) as ONE persistent launch (the default at these sizes) beside the launch flow (SLA_ONCHIP=0) of the same box"""
    on, lf = load(f"onchip_{name}"), load(f"launchflow_{name}")
    if not on or not on.get("value"):
        return f"| {label} | — | — | — | — | (no line) |"
    k = (on.get("kernels") or {}).get("ONCHIP") or {}
    return (f"| {label} | {on['ms_per_step'] * 1e3:.1f} µs | **{on['value']:.0f}** | one launch for all the steps (no K1 of its own) | "
            f"ONCHIP: {on['ms_per_step'] * 1e3:.1f} µs per step, two grid-wide counter barriers each | launch flow on the same box: "
            f"{(lf or {}).get('ms_per_step', 0) * 1e3:.1f} µs = {(lf or {}).get('value', 0):.0f} it/s; {str(k.get('plan', ''))[:150]} |")


d2 = load("poisson2d_1m")
print(row("2 (1 M 5-pt Poisson) bicgstabStep" + (" — ONE persistent on-chip launch (round 6)" if d2 and d2.get("onchip") else " (step graph replay)"), d2))
print(onchip_row("2 (1 M 5-pt Poisson), 200-step window, on-chip against the launch flow", "poisson2d_1m"))
def pair_row(label, on_name, lf_name):
    on, lf = load(on_name), load(lf_name)
    if not on or not on.get("value"):
        return f"| {label} | — | — | — | — | (no line) |"
    return (f"| {label} | {on['ms_per_step'] * 1e3:.1f} µs | **{on['value']:.0f}** | one launch | ONCHIP | launch flow on the same box: "
            f"{(lf or {}).get('ms_per_step', 0) * 1e3:.1f} µs = {(lf or {}).get('value', 0):.0f} it/s |")


print(pair_row("2, cgsStep on chip against the launch flow", "onchip_cgs_poisson2d_1m", "launchflow_cgs_poisson2d_1m"))
print(pair_row("2, `linSolve0 BICGSTAB_` on chip (step + true residual + test per iteration) against the launch flow", "linsolve0_onchip_poisson2d_1m", "linsolve0_launchflow_poisson2d_1m"))
print(onchip_row("4's per-rank slab at N = 8 (216 × 216 × 27), on-chip against the launch flow", "laplace3d_slab8"))
print(onchip_row("108³ (1.26 M rows), on-chip against the launch flow", "laplace3d_1m"))
print(row("5-matrix (2 M banded) bicgstabStep", load("banded_2m")))
lf = load("gmres_banded_2m_launchflow") or {}
print(row("5 (2 M banded) GMRES(30) Arnoldi step — the Gram–Schmidt as ONE persistent launch per step (end of round 6)", load("gmres_banded_2m"),
          f"three launches per step (`arn_orth = 0`) on the same box: {lf.get('ms_per_step', 0) * 1e3:.1f} µs = {lf.get('value', 0):.0f} steps / s" if lf.get("value") else ""))
print(row("3a BiCGSTAB as the headline workload (default: exact fold)", load("random_spd_10m_bicgstab")))
print(row("3a BiCGSTAB, opt-in relaxed order (`SLA_TILE_RELAXED=1`)", load("random_spd_10m_bicgstab_relaxed")))
print(row("3a CGS (default: exact fold)", load("random_spd_10m_cgs")))
print(row("3a CGS, opt-in relaxed order", load("random_spd_10m_cgs_relaxed")))
print(row("3a at 1 M rows (default: exact fold)", load("random_spd_1m")))
print(row("3a at 1 M rows, opt-in relaxed order", load("random_spd_1m_relaxed")))
print(row("3b (200 k rows, 2000 per row)", load("dense_rows_200k")))
for xe in ("window", "allgather"):
    for f in (1, 0):
        print(row(f"4 as ONE slab of 8 (108³ rows) through a 1-rank RCCL communicator, {xe}, {'fused K45' if f else 'split K4 / K5'}", load(f"slab_1rank_rccl_{xe}_fuse{f}")))
d = load("1rank_rccl")
if d and d.get("value"):
    c, r3 = d.get("contract_allgather") or {}, d.get("random_spd_10m") or {}
    print(f"| first contact rehearsed: full-size line through a 1-rank RCCL communicator (pre-flight {sorted((d.get('preflight') or {}).keys())}) | {d['ms_per_step'] * 1e3:.1f} µs | {d['value']:.0f} | | | "
          f"`contract_allgather` {c.get('value', 0):.0f} it/s ({c.get('x_exchange')}, {c.get('allgather_ms') or 0:.3f} ms per all-gather); `random_spd_10m` {r3.get('value', 0):.1f} it/s |")
d = load("loopback_2ranks")
if d and d.get("value"):
    c, r3 = d.get("contract_allgather") or {}, d.get("random_spd_10m") or {}
    print(f"| … on 2 loopback ranks (threads on one GPU: a rehearsal, not a scaling number) | {d['ms_per_step'] * 1e3:.1f} µs | {d['value']:.0f} | | | "
          f"`contract_allgather` {c.get('value', 0):.0f} it/s; `random_spd_10m` {r3.get('value', 0):.1f} it/s, x exchange {(r3.get('x_exchange') or {}).get('mode')} |")
    xr = d.get("exchange_roofline") or {}
    if xr.get("exchanges"):
        ex = "; ".join(f"{k}: {v['bytes_to_busiest_peer']} B to the busiest peer in {v['ms'] * 1e3:.1f} µs (model {v['model_ms'] * 1e3:.1f} µs, {v['bound']}-bound)" for k, v in xr["exchanges"].items())
        print(f"| … its `exchange_roofline` (plan bytes per peer against {xr['link_peak_gbps']:.0f} GB/s per xGMI link; {'real links' if xr['real_links'] else 'rehearsal: no link crossed'}) | | | | | "
              f"{ex}; step measured {xr['step']['measured_ms'] * 1e3:.1f} µs against the model's {xr['step']['model_ms'] * 1e3:.1f} µs |")
for f, inj in (("p2p", "p2p"), ("hang", "p2p_hang"), ("rank1", "p2p_data_rank1")):
    d = load(f"loopback_fault_{f}")
    if d:
        print(f"| … with `SLA_FAULT_INJECT={inj}` | | {d.get('value') or 0:.0f} | | | fallback: {str(d.get('fallback'))[:160]} |")
