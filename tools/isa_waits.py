"""Static look for serialised memory round trips: compiles a csrc/*.hip for gfx950 to assembly and prints, per kernel, how its VMEM loads are
grouped between `s_waitcnt vmcnt(..)` instructions -- many groups of one or two loads mean dependent round trips (the LayerNorm kernels
were 7 groups before their loads were hoisted).  usage: python tools/isa_waits.py norm.hip [name filter]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "spegnet_amd", "csrc", sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = os.path.join(tempfile.gettempdir(), "isa_waits.s")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-w", "--cuda-device-only", "-S", src,
                "-o", out, "-I", os.path.join(ROOT, "include")], check=True)
name, groups, cur, stores = None, [], 0, 0
def flush():
    if name and (flt in name):
        g = [x for x in groups if x]
        print(f"{len(g):3d} load groups {g[:24]}{'...' if len(g) > 24 else ''}  stores {stores:3d}  {name[:110]}")
for line in open(out):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        flush()
        name, groups, cur, stores = m.group(1), [], 0, 0
        continue
    if name is None:
        continue
    t = line.strip()
    if t.startswith(("global_load", "buffer_load", "flat_load")) and "lds" not in t:
        cur += 1
    elif t.startswith(("global_store", "buffer_store")):
        stores += 1
    elif t.startswith("s_waitcnt") and "vmcnt" in t:
        groups.append(cur); cur = 0
    elif t.startswith("s_endpgm"):
        groups.append(cur); flush(); name = None
