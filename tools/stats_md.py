"""rocprofv3 kernel_stats.csv -> markdown table: python tools/stats_md.py stats.csv [top_n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
print("| kernel | calls | total ms | avg us | % |")
print("|---|---|---|---|---|")
for r in rows[:top]:
    name = r["Name"]
    name = name if len(name) <= 100 else name[:97] + "..."
    print(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
