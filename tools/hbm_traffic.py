"""Turns two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of bench.py into per-kernel HBM-side bytes per launch.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
  python tools/hbm_traffic.py gpurun_out/pmc_fetch/*/*_counter_collection.csv gpurun_out/pmc_write/*/*_counter_collection.csv > profiles/rNN_hbm_traffic.json

Corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): the counters are in KiB on gfx950 and FETCH_SIZE reads
half the true value (x2).  The three gemm_nt kernels (gemm_nt_kernel, gemm_nt256_kernel, gemm_nt256p_kernel) are
also summed as the family "gemm_nt_kernel<*>" that bench.py's roofline reports.
"""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_sha16():      # same recipe as bench.py::source_sha16
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "kuzushiji-vision_amd", "csrc", "*.[hc]*"))) + [os.path.join(ROOT, "include", "kzv.h"),
                                                                                                  os.path.join(ROOT, "kuzushiji-vision_amd", "csrc", "Makefile")]:
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z_0-9]+)(<[^>]*>)?", name)
    base, targs = m.group(1), m.group(2) or ""
    if base.startswith(("ln_", "attn_")) and targs:
        return base + "<" + targs[1:-1].split(",")[0].strip() + ">"
    return base


def per_kernel(path: str, counter: str, scale: float):
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"]) * scale
        n[k] += 1
    return tot, n


def main():
    fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE", 1024.0 * 2.0)
    write, nw = per_kernel(sys.argv[2], "WRITE_SIZE", 1024.0)
    out = {}
    for k in sorted(set(fetch) | set(write)):
        out[k] = {"launches": nf.get(k, nw.get(k, 0)),
                  "fetch_bytes_per_launch": fetch[k] / max(1, nf.get(k, 0)),
                  "write_bytes_per_launch": write[k] / max(1, nw.get(k, 0))}
    fam = [k for k in out if k.startswith("gemm_nt")]
    nl = sum(out[k]["launches"] for k in fam)
    if nl:
        out["gemm_nt_kernel<*>"] = {
            "launches": nl,
            "fetch_bytes_per_launch": sum(out[k]["fetch_bytes_per_launch"] * out[k]["launches"] for k in fam) / nl,
            "write_bytes_per_launch": sum(out[k]["write_bytes_per_launch"] * out[k]["launches"] for k in fam) / nl}
    print(json.dumps({
        "source_sha16": source_sha16(),
        "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (two separate passes), tools/hbm_traffic.py",
        "corrections": "counter unit KiB; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section)",
        "note": "FETCH_SIZE counts L2-side fabric requests (Infinity-Cache hits included), i.e. L2 misses, an upper bound on HBM reads",
        "kernels": out}, indent=1))


if __name__ == "__main__":
    main()
