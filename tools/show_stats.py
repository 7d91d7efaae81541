"""Per-registration view of a rocprofv3 kernel_stats.csv (calls, average, total per registration, share)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 70
tot = sum(float(r['TotalDurationNs']) for r in rows)
regs = [int(r['Calls']) for r in rows if 'k_morton' in r['Name']][0]
ncmd = sum(int(r['Calls']) for r in rows)
print(f"registrations {regs}  gpu ms/reg {tot/regs/1e6:.3f}  commands/reg {ncmd/regs:.1f}")
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:top]:
    n = r['Name'].replace('plade::', '').replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:44]
    print(f"{n:44s} calls/reg {int(r['Calls'])/regs:7.2f} avg_us {float(r['AverageNs'])/1e3:8.2f} us/reg {float(r['TotalDurationNs'])/regs/1e3:8.1f} {100*float(r['TotalDurationNs'])/tot:5.1f}%")
