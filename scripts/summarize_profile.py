"""Condense rocprofv3 CSV output (kernel stats + PMC counter collection) into a small text/JSON
summary: per-kernel launch count, average duration, and per-launch FETCH_SIZE / WRITE_SIZE."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:70]


def is_product(name):
    """kernels of libpylinac_hip.so (everything else is torch: synthetic input generation, copies, the record cat)"""
    return not (name.startswith(("at::", "rocprim::", "__amd_rocclr", "void at::")) or "at::native" in name)


summary = {}
for f in glob.glob(os.path.join(out, "trace", "*kernel_stats.csv")):
    print("== kernel stats (product kernels only; torch input generation / copies listed in one line at the end):",
          os.path.basename(f))
    other_ns, other_calls = 0.0, 0
    rows = list(csv.DictReader(open(f)))
    prod_total = sum(float(r["TotalDurationNs"]) for r in rows if is_product(short(r["Name"])))
    for row in rows:
        name = short(row["Name"])
        if not is_product(name):
            other_ns += float(row["TotalDurationNs"])
            other_calls += int(row["Calls"])
            continue
        share = 100.0 * float(row["TotalDurationNs"]) / prod_total if prod_total else 0.0
        print(f'{name:72s} calls={row["Calls"]:>6s} avg_ns={float(row["AverageNs"]):12.0f} share_of_product={share:5.1f}%')
        summary.setdefault(name, {})["avg_us"] = float(row["AverageNs"]) / 1e3
        summary[name]["calls"] = int(row["Calls"])
    print(f'(not product: {other_calls} torch / runtime launches, {other_ns / 1e6:.2f} ms in total)')
for counter, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(out, sub, "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter:
                acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    print(f"== {counter} per launch (raw counter units = KiB as reported by rocprofv3)")
    for name, v in sorted(acc.items()):
        avg = sum(v) / len(v)
        print(f"{name:72s} n={len(v):4d} avg={avg:14.1f}")
        summary.setdefault(name, {})[counter] = avg
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
