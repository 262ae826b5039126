#!/usr/bin/env python3
"""Static resource table of every device kernel of libsla_hip (no GPU needed): registers, AGPRs, spills, scratch, LDS and the occupancy the
compiler derives from them, from hipcc's own `-Rpass-analysis=kernel-resource-usage` remarks with the Makefile's flags.

    python tools/kernel_resources.py [--all] [> profiles/rNN_kernel_resources.txt]      (--all: rocPRIM's sort / scan kernels too)

What to look for: ScratchSize > 0 or VGPR spills in a kernel of the step (none), occupancy of the streaming kernels (HBM-bound sweeps want
>= 4 waves/SIMD in flight), LDS per workgroup of the LDS-resident forms (one workgroup per CU by design: 160 KB)."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sparse-linear-algebra_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FILT = os.environ.get("CXXFILT", "c++filt")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "--offload-arch=gfx950", "-munsafe-fp-atomics",
         "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", "-o", os.devnull]
KEYS = (("TotalSGPRs", "sgpr"), ("VGPRs", "vgpr"), ("AGPRs", "agpr"), ("ScratchSize [bytes/lane]", "scratch"), ("Occupancy [waves/SIMD]", "occ"),
        ("SGPRs Spill", "sspill"), ("VGPRs Spill", "vspill"), ("LDS Size [bytes/block]", "lds"))


def demangle(names):
    try:
        out = subprocess.run([FILT], input="\n".join(names), stdout=subprocess.PIPE, text=True, check=True).stdout.split("\n")
        return [o if o else n for o, n in zip(out, names)]
    except Exception:
        return names


def short(name):
    # sla::spmv_tile_kernel<0, long>(sla::SpmvArgs<long>, ...) -> spmv_tile_kernel<0, long>
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    depth, out = 0, []
    for ch in name:
        if ch == "(" and depth == 0:
            break
        depth += ch == "<"
        depth -= ch == ">"
        out.append(ch)
    return "".join(out).replace("sla::", "").replace("(anonymous namespace)::", "")


def main():
    rows = []
    for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        p = subprocess.run([HIPCC] + FLAGS + [src], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if p.returncode != 0:
            sys.stderr.write(p.stdout[-2000:])
            raise SystemExit("compile failed: " + src)
        cur = None
        for ln in p.stdout.splitlines():
            m = re.search(r"remark: Function Name: (\S+)", ln)
            if m:
                cur = {"file": os.path.basename(src), "name": m.group(1)}
                rows.append(cur)
                continue
            if cur is None:
                continue
            for key, k in KEYS:
                m = re.search(r"remark:\s+" + re.escape(key) + r": (\d+)", ln)
                if m:
                    cur[k] = int(m.group(1))
    # device functions that are not kernels also get a remark block (no occupancy of their own matters): keep kernels = those in rows with all keys
    rows = [r for r in rows if all(k in r for _, k in KEYS)]
    names = demangle([r["name"] for r in rows])
    if "--all" not in sys.argv:
        keep = [i for i, n in enumerate(names) if "rocprim" not in n]
        print("# (%d rocPRIM kernels of the device COO sort / transpose left out: --all lists them)" % (len(rows) - len(keep)))
        rows, names = [rows[i] for i in keep], [names[i] for i in keep]
    print("# hipcc -Rpass-analysis=kernel-resource-usage, gfx950, the Makefile's flags; one line per device kernel")
    print("# occ = waves/SIMD the registers allow (LDS-resident forms run ONE workgroup per CU whatever this says); agpr > 0 with vgpr = 256: the")
    print("# allocator parked values in accumulation registers (v_accvgpr moves, no memory traffic); scratch = bytes/lane of private memory")
    print("%-22s %-64s %5s %5s %5s %4s %7s %7s %8s %7s" % ("file", "kernel", "vgpr", "agpr", "sgpr", "occ", "sspill", "vspill", "scratch", "lds"))
    for r, n in zip(rows, names):
        print("%-22s %-64s %5d %5d %5d %4d %7d %7d %8d %7d" % (r["file"], short(n)[:64], r["vgpr"], r["agpr"], r["sgpr"], r["occ"], r["sspill"],
                                                               r["vspill"], r["scratch"], r["lds"]))
    bad = [(r, n) for r, n in zip(rows, names) if r["scratch"] > 0 or r["vspill"] > 0]
    print("# kernels with scratch or VGPR spills: %d of %d" % (len(bad), len(rows)))
    for r, n in bad:
        print("#   %s: scratch %d B/lane, %d VGPRs spilled" % (short(n), r["scratch"], r["vspill"]))


if __name__ == "__main__":
    main()
