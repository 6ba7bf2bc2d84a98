"""Per-kernel MfmaUtil from one rocprofv3 PMC pass of bench.py:

  rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
  python tools/mfma_util.py gpurun_out/pmc_mfma/*_counter_collection.csv > profiles/rNN_mfma_util.json

MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * SIMDs) * 100 per launch (derived counter, gfx94x formula: ROCm 7.2
ships no gfx950 section -- MI355X_MICROARCH.md "rocprofv3 PMC slots"); mean and max over a kernel's launches."""
import collections
import csv
import json
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.match(r"([A-Za-z_0-9]+)", name).group(1)


def main():
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(sys.argv[1])):
        if r["Counter_Name"] == "MfmaUtil":
            vals[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    out = {k: {"MfmaUtil": {"launches": len(v), "mean": sum(v) / len(v), "max": max(v)}} for k, v in sorted(vals.items()) if max(v) > 0.5}
    print(json.dumps({"command": "rocprofv3 --pmc MfmaUtil --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline",
                      "note": "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * SIMD_NUM) * 100, per launch, averaged per kernel",
                      "kernels": out}, indent=1))


if __name__ == "__main__":
    main()
