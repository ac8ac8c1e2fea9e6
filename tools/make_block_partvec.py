#!/usr/bin/env python3
"""FALLBACK part vector for graphs the reference's partitioner front-ends cannot finish in the build container: a
seeded COMMUNITY-BLOCK vector -- label propagation (partition.label_propagation, the same routine the engine's
vertex order uses) finds communities, which are packed into k parts of equal stored-entry weight, largest first
(LPT).  NOT produced by the reference's tools: files carry the extension `.cb.gz` and the statistics say so.  Runs
anywhere (no /root/reference needed).

usage: python tools/make_block_partvec.py --workload products --generator sbm --k 8"""
import argparse, importlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
PKG = "scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd"


def community_block_partvec(row, col, n, k, iters=8):
    partition = importlib.import_module(PKG + ".partition")
    lab = partition.label_propagation(row, col, n, iters=iters)
    ul, inv = torch.unique(lab, return_inverse=True)
    w = torch.zeros(ul.numel(), dtype=torch.int64, device=row.device).index_add_(0, inv[row], torch.ones_like(row))
    order = torch.argsort(-w, stable=True).cpu().numpy()
    wn = w.cpu().numpy()
    load = np.zeros(k, dtype=np.int64)
    part_of = np.zeros(ul.numel(), dtype=np.int64)
    for c in order:                                   # LPT: heaviest community to the lightest part
        q = int(np.argmin(load))
        part_of[c] = q
        load[q] += wn[c]
    return torch.from_numpy(part_of).to(row.device)[inv]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="products")
    ap.add_argument("--generator", default="sbm")
    ap.add_argument("--k", type=int, default=8)
    ap.add_argument("--device", default="cpu")
    a = ap.parse_args()
    synth, io_ = importlib.import_module(PKG + ".synth"), importlib.import_module(PKG + ".pargcn_io")
    n, row, col, val = synth.make_graph(a.workload, seed=0, device=a.device, generator=a.generator)
    pv = community_block_partvec(row, col, n, a.k).cpu()
    label = a.workload + ("-sbm" if a.generator == "sbm" else "")
    out = os.path.join(ROOT, "tests", "golden", "partvec", "%s.A.mtx.%d.cb" % (label, a.k))
    io_.write_partvec(out, pv.numpy())
    import gzip, shutil                      # committed compressed (partition.read_partvec reads .gz)
    with open(out, "rb") as fi, gzip.open(out + ".gz", "wb", compresslevel=9) as fo:
        shutil.copyfileobj(fi, fo)
    os.remove(out)
    row, col = row.cpu(), col.cpu()
    rec = {}
    for name, v in (("cb", pv), ("rp", synth.random_partvec(n, a.k, seed=0))):
        cut = v[row] != v[col]
        nnz_p = torch.bincount(v[row], minlength=a.k)
        rec[name] = {"boundary_rows_per_aggregation": int(torch.unique(v[row[cut]] * n + col[cut]).numel()),
                     "max_part_vertices": int(torch.bincount(v, minlength=a.k).max()), "max_part_nnz": int(nnz_p.max()),
                     "imbalance_nnz": float(nnz_p.max()) * a.k / float(row.numel())}
    with open(os.path.join(ROOT, "tests", "golden", "partvec", "%s.cb.stats.json" % label), "w") as fh:
        json.dump({"workload": label, "n": n, "nnz": int(row.numel()), "note": "community-block vector (label propagation + "
                   "LPT packing), NOT from the reference's partitioners", "parts": {str(a.k): rec}}, fh, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
