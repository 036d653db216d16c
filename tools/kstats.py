"""Pretty-print a rocprofv3 kernel_stats.csv: python tools/kstats.py FILE"""
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("msm::", "")
    print(f"{n[:44]:44s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:10.1f} total_ms={float(r['TotalDurationNs'])/1e6:9.3f} {float(r['Percentage']):6.2f}%")
