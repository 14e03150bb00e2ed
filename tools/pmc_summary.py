"""Summarises rocprofv3 --pmc counter_collection CSVs per kernel.

  pmc_summary.py traffic <fetch.csv> <write.csv> <out.json> [--source "..."]
      mean FETCH_SIZE / WRITE_SIZE per launch (KiB counters -> bytes; FETCH_SIZE doubled as MI355X_MICROARCH.md "HBM" prescribes for
      gfx950 wide streaming reads) -> JSON {"kernels": {key: {"fetch_bytes_x2", "write_bytes", "launches"}}} keyed by bench.py's op names
      where a mapping exists, and by the demangled kernel name otherwise.
  pmc_summary.py mfma <counters.csv> [--title "..."]
      markdown table: per kernel, launches, mean duration, SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_BF16, SQ_BUSY_CYCLES /
      GRBM_GUI_ACTIVE and the derived MFMA-busy fraction (busy cycles / (4 SIMDs x CUs x active cycles))."""
import argparse, collections, csv, json, re, sys


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("spg::", "")[:120]


def load(path):
    """{kernel: {counter: [values...]}, 'dur': [ns...]}"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    seen = set()
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (r["Dispatch_Id"], k)
            if key not in seen:
                seen.add(key)
                acc[k]["_dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return acc


# bench.py op name -> predicate on the demangled kernel name
BENCH_KEYS = {
    "gemm_nt<bf16,dense>": lambda n: n.startswith("gemm_nt_pipe_kernel<unsigned short, false") or n.startswith("gemm_nt_v3_kernel<false"),
    "gemm_nt<bf16,conv3x3>": lambda n: (n.startswith("conv3x3_halo_kernel") or n.startswith("gemm_nt_pipe_kernel<unsigned short, true")
                                        or n.startswith("gemm_nt_v3_kernel<true")),
    "gemm_tn_group<bf16> (one trunk block's wgrads)": lambda n: n.startswith("gemm_tn_group_kernel") or n.startswith("gemm_tn_group4_kernel"),
    "gemm_tn<bf16,conv3x3> (+reduce)": lambda n: (n.startswith("conv3x3_wgrad_halo_kernel") or n.startswith("gemm_tn_pipe_kernel<unsigned short, true")
                                                  or n.startswith("gemm_tn_pipe4_kernel<unsigned short, true")),
}


def mean(v):
    return sum(v) / max(len(v), 1)


def traffic(a):
    fe, wr = load(a.fetch), load(a.write)
    out = {"source": a.source, "unit": "bytes per launch (mean)", "kernels": {}}
    names = sorted(set(fe) | set(wr))
    for n in names:
        f = fe.get(n, {}).get("FETCH_SIZE", [])
        w = wr.get(n, {}).get("WRITE_SIZE", [])
        out["kernels"][n] = {"fetch_bytes_x2": mean(f) * 1024 * 2 if f else None, "write_bytes": mean(w) * 1024 if w else None,
                             "launches": max(len(f), len(w))}
    for key, pred in BENCH_KEYS.items():
        fs = [v for n in fe if pred(n) for v in fe[n].get("FETCH_SIZE", [])]
        ws = [v for n in wr if pred(n) for v in wr[n].get("WRITE_SIZE", [])]
        if fs and ws:
            out["kernels"][key] = {"fetch_bytes_x2": mean(fs) * 2048, "write_bytes": mean(ws) * 1024, "launches": len(fs)}
    json.dump(out, open(a.out, "w"), indent=1)
    print(f"wrote {a.out}: {len(out['kernels'])} kernels")


def mfma(a):
    d = load(a.counters)
    rows = []
    for n, c in d.items():
        busy, mops = c.get("SQ_VALU_MFMA_BUSY_CYCLES", []), c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", [])
        act = c.get("GRBM_GUI_ACTIVE", []) or c.get("SQ_BUSY_CYCLES", [])
        if not busy:
            continue
        dur = mean(c["_dur_ns"])
        rows.append((sum(c["_dur_ns"]), n, len(busy), dur / 1e3, mean(busy), mean(mops) if mops else float("nan"), mean(act) if act else float("nan")))
    rows.sort(reverse=True)
    if a.title:
        print(f"# {a.title}\n")
    print("MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles); kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs,\n"
          "MI355X_MICROARCH.md 'DVFS give-back').  bf16 MFMA FLOPs = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 (one MOP = 512 FLOPs on gfx94x/95x).\n")
    print("| total ms | launches | avg us | MFMA busy cycles / launch | MFMA MOPS bf16 / launch | GRBM_GUI_ACTIVE / launch | MFMA-busy frac | TFLOP/s from MOPS | kernel |")
    print("|---|---|---|---|---|---|---|---|---|")
    for tot, n, k, us, busy, mops, act in rows[: a.top]:
        cyc = act / 8.0 if act == act else float("nan")
        frac = busy / (1024.0 * cyc) if cyc == cyc and cyc > 0 else float("nan")
        tf = mops * 512 / (us * 1e-6) / 1e12 if mops == mops else float("nan")
        print(f"| {tot/1e6:.2f} | {k} | {us:.1f} | {busy:.3g} | {mops:.3g} | {act:.3g} | {frac:.3f} | {tf:.0f} | `{n}` |")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    t = sub.add_parser("traffic"); t.add_argument("fetch"); t.add_argument("write"); t.add_argument("out"); t.add_argument("--source", default="")
    m = sub.add_parser("mfma"); m.add_argument("counters"); m.add_argument("--title", default=""); m.add_argument("--top", type=int, default=30)
    a = ap.parse_args()
    traffic(a) if a.cmd == "traffic" else mfma(a)
