"""Who launches the stray copies / fills?  Prints, for every kernel of a rocprofv3 kernel trace whose name matches a pattern, the
kernels that ran just before and after it (stream order).  usage: trace_neighbours.py <kernel_trace.csv> <pattern> [n]"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pat = re.compile(sys.argv[2])
short = lambda n: re.sub(r"\(.*$", "", re.sub(r"^void ", "", n)).replace("spg::", "")[:70]
ctx = collections.Counter()
for i, r in enumerate(rows):
    if pat.search(r["Kernel_Name"]):
        prev = short(rows[i - 1]["Kernel_Name"]) if i else "-"
        nxt = short(rows[i + 1]["Kernel_Name"]) if i + 1 < len(rows) else "-"
        ctx[(prev, short(r["Kernel_Name"]), nxt, r.get("Grid_Size", "?"))] += 1
for (p, k, n, g), c in ctx.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 20):
    print(f"{c:4d} x  {p}  ->  [{k} grid {g}]  ->  {n}")
