"""Summarise rocprofv3 output dirs written by tests/prof.sh: per-kernel time stats + per-kernel PMC averages."""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]

def short(name):
    for k in ("k_assoc_walk", "k_assoc_staged", "k_solve", "k_finalize", "k_reset_items", "k_rank_source", "k_rank_target",
              "k_scatter", "k_source_keys", "k_target_keys", "k_scan_local", "k_scan_tops", "k_scan_add", "k_bbox"):
        if k in name:
            return k
    return name[:60]

for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats (rocprofv3 --kernel-trace --stats):", os.path.relpath(f, root))
    rows = list(csv.DictReader(open(f)))
    for r in rows[:25]:
        print(f"{short(r['Name']):28s} calls={r['Calls']:>6s} total_ns={r['TotalDurationNs']:>12s} avg_ns={float(r['AverageNs']):12.1f} pct={r['Percentage']}")

for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"]); c = r["Counter_Name"]
            acc[k][c] += float(r["Counter_Value"]); cnt[k][c] += 1
        print("== PMC per-dispatch averages:", os.path.relpath(f, root))
        for k in sorted(acc, key=lambda k: -sum(acc[k].values()))[:8]:
            print(f"  {k:20s} " + "  ".join(f"{c}={acc[k][c] / cnt[k][c]:.4g} (n={cnt[k][c]})" for c in sorted(acc[k])))

for tag in ("next_vox", "next_feat", "next_map"):
    for f in glob.glob(os.path.join(root, tag, "**", "*kernel_stats.csv"), recursive=True):
        print(f"== kernel stats of the {tag} probe (rocprofv3 --kernel-trace --stats):", os.path.relpath(f, root))
        for r in list(csv.DictReader(open(f)))[:18]:
            nm = r["Name"]
            nm = nm[nm.find("k_"):][:40] if "k_" in nm else nm[:40]
            print(f"{nm:42s} calls={r['Calls']:>6s} total_ns={r['TotalDurationNs']:>12s} avg_ns={float(r['AverageNs']):12.1f} pct={r['Percentage']}")
