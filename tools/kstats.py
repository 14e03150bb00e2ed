"""prints name, calls, average us of a rocprofv3 kernel_stats CSV (names contain commas: needs a CSV parser)"""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if len(sys.argv) < 3 or sys.argv[2] in r["Name"]:
        print(f"{float(r['AverageNs'])/1e3:9.1f} us x {r['Calls']:>5s}  {r['Name'][:90]}")
