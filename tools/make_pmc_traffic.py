#!/usr/bin/env python3
"""One record of profiles/pmc_traffic.json from a pmc_summary.py text (tools/final_profile.sh): per-launch L2<->fabric
bytes of one SpMM launch group.  FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled on gfx950
(MI355X_MICROARCH.md: the counter tallies 128-byte fabric reads at 64 B).  The file holds {"records": [...]}, one per
(workload, generator, partvec, ranks, f); a record with the same key is replaced.
usage: make_pmc_traffic.py SUMMARY.txt OUT.json SOURCE-NAME [workload generator ranks f partvec block [kernel-substring]]"""
import importlib.util
import json
import os
import re
import sys


def source_stamp():
    """bench.kernel_source_stamp(): the kernel sources this PMC record is valid for."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.kernel_source_stamp()


def main():
    txt = open(sys.argv[1]).read().splitlines()
    per, cur = {}, None
    only = sys.argv[10] if len(sys.argv) > 10 else None     # keep the kernels whose name (incl. template arguments) contains one of these
    alts = only.split("|") if only is not None else []      # '|'-separated substrings: the kernels of ONE pass (their traffic is summed)
    hit = lambda ln: any(a in ln for a in alts)
    for ln in txt:
        m = re.search(r"counter_collection\.csv \| .*::(\w+)[<(]", ln)
        if m:
            if only is not None and not hit(ln):
                cur = None
                continue
            cur = per.setdefault(m.group(1), {})
            continue
        m = re.match(r"\s+(\w+)\s+n=(\d+)\s+mean=([\d.e+]+)", ln)
        if m and cur is not None:
            cur[m.group(1)] = float(m.group(3))
        if "kernel_trace.csv" in ln:
            cur = None
            m = re.search(r"::(\w+)[<(].*mean=([\d.]+) us", ln)
            if m and (only is None or hit(ln)):
                per.setdefault(m.group(1), {}).setdefault("mean_us", float(m.group(2)))
    read = sum(2 * 1024 * v.get("FETCH_SIZE", 0) for v in per.values())
    write = sum(1024 * v.get("WRITE_SIZE", 0) for v in per.values())
    extra = sys.argv[4:]
    workload, generator, ranks, f, partvec, block = (extra + ["reddit", "rmat", "1", "128", "random", "loc"][len(extra):])[:6]
    out = {"workload": workload, "ranks": ranks, "f": int(f), "generator": generator, "partvec": partvec, "block": block,
           "source_stamp": source_stamp(),
           "kernel": (only if only is not None else ("A_loc.H" if block == "loc" else "A_halo[%s].slab" % block[4:])
                      + " launch group: " + " + ".join(sorted(per))),
           "hbm_bytes_per_launch": int(read + write), "read_bytes": int(read), "write_bytes": int(write),
           "per_kernel": per,
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum in separate passes with --kernel-trace "
                   "only (tools/final_profile.sh), mean over the dispatches of `tools/group_probe.py` (the block exactly as bench.py "
                   "builds it).  FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B fabric reads at 64 B). "
                   "These are L2<->fabric bytes: Infinity-Cache (MALL) hits are included, so true HBM traffic is <= this figure.",
           "source": sys.argv[3] if len(sys.argv) > 3 else sys.argv[1]}
    doc = {"records": []}
    if os.path.exists(sys.argv[2]):
        try:
            old = json.load(open(sys.argv[2]))
            doc = old if "records" in old else {"records": []}
        except Exception:
            pass
    key = lambda r: (r.get("workload"), r.get("generator"), r.get("partvec"), str(r.get("ranks")), r.get("f"), r.get("block", "loc"))
    doc["records"] = [r for r in doc["records"] if key(r) != key(out)] + [out]
    with open(sys.argv[2], "w") as fh:
        json.dump(doc, fh, indent=1)
    print(json.dumps({k: out[k] for k in ("hbm_bytes_per_launch", "read_bytes", "write_bytes")}))


if __name__ == "__main__":
    main()
