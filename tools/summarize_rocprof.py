#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats run into the rows that matter (our kernels first).
usage: summarize_rocprof.py <kernel_stats.csv> [<kernel_trace.csv>] > profiles/<round>_kernel_stats.md"""
import csv
import sys


def short(name):
    name = name.replace("rbg::(anonymous namespace)::", "rbg::")
    return name if len(name) < 110 else name[:107] + "..."


rows = list(csv.DictReader(open(sys.argv[1])))
ours = [r for r in rows if "rbg::" in r["Name"]]
other = [r for r in rows if "rbg::" not in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("| kernel | calls | avg ms | min ms | max ms | total ms | % of GPU time in run |")
print("|---|---|---|---|---|---|---|")
for r in sorted(ours, key=lambda r: -float(r["TotalDurationNs"])):
    print(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs']) / 1e6:.3f} | {float(r['MinNs']) / 1e6:.3f} | "
          f"{float(r['MaxNs']) / 1e6:.3f} | {float(r['TotalDurationNs']) / 1e6:.1f} | {100 * float(r['TotalDurationNs']) / tot:.2f} |")
o = sum(float(r["TotalDurationNs"]) for r in other)
print(f"| (everything else: torch index synthesis, sorts, copies; outside the timed region) | {sum(int(r['Calls']) for r in other)} | | | | {o / 1e6:.1f} | {100 * o / tot:.2f} |")
if len(sys.argv) > 2:
    tr = [r for r in csv.DictReader(open(sys.argv[2])) if "rbg::" in r["Kernel_Name"]]
    if tr:
        print("\nper-dispatch resources (kernel trace):\n")
        print("| kernel | VGPR | accum VGPR | SGPR | LDS B | scratch B | workgroup | grid |")
        print("|---|---|---|---|---|---|---|---|")
        seen = set()
        for r in tr:
            k = short(r["Kernel_Name"])
            if k in seen:
                continue
            seen.add(k)
            print(f"| `{k}` | {r.get('VGPR_Count', '')} | {r.get('Accum_VGPR_Count', '')} | {r.get('SGPR_Count', '')} | {r.get('LDS_Block_Size', '')} | "
                  f"{r.get('Scratch_Size', '')} | {r.get('Workgroup_Size', r.get('Workgroup_Size_X', ''))} | {r.get('Grid_Size', r.get('Grid_Size_X', ''))} |")
