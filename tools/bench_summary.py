"""Prints the headline and the per-kernel table of a bench.py JSON line.  usage: bench_summary.py <file with the line> [top]"""
import json, sys
d = None
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
top = int(sys.argv[2]) if len(sys.argv) > 2 else 16
print({k: v for k, v in d.items() if k in ("value", "ms_per_step", "n_gpus")})
r = d.get("roofline") or {}
print({k: v for k, v in r.items() if k != "kernels" and k != "traffic_source"})
for k in sorted(r.get("kernels", []), key=lambda k: -k["ms_per_step"])[:top]:
    print(f"{k['kernel'][:48]:48s} n={k['launches_per_step']:6} avg={k['avg_us']:8.1f}us  ms={k['ms_per_step']:6.2f} {k['achieved']:8.1f} {k['unit']:8s} frac={k['frac']}")
