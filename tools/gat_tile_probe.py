#!/usr/bin/env python3
"""What would `attention @ Z` cost on the strip / MFMA-tile decomposition?  (run under gpurun)

The GAT products run as gathers over plain CSR (pgcn_spmm_heads_f32: every entry gathers a 1 KB row).  This probe
builds, per head, the tiled structure of the GCN path with that head's attention values (slow torch build -- only the
product is timed) and runs the existing launch group at f = d, to price the tile route before any fill kernel exists."""
import argparse, importlib, json, os, sys
import numpy as np, torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd"
pkg = lambda m: importlib.import_module(PKG + "." + m)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="reddit")
    ap.add_argument("--heads", type=int, default=4)
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--out", default="gpurun_out/gat_tile_probe.json")
    args = ap.parse_args()
    synth, partition, gat, kernels = pkg("synth"), pkg("partition"), pkg("gat"), pkg("kernels")
    dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
    n, row, col, val = synth.make_graph(args.workload, seed=0, device=dev)
    K = kernels.HipKernels(dev)
    part = partition.build_partition(row, col, val, n, torch.zeros(n, dtype=torch.int64, device=dev), 0, 1, with_transpose=False)
    heads, d = args.heads, args.d
    eng = gat.GatEngine(part, K, dev, None, mode="standard")
    st = eng.new_layer_state(heads, d)
    F = heads * d
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    Z = torch.randn(n, F, device=dev, generator=gen)
    s1 = torch.randn(n, heads, device=dev, generator=gen); s2 = torch.randn(n, heads, device=dev, generator=gen)
    G = torch.randn(n, F, device=dev, generator=gen)
    ref = eng.forward(st, Z, s1, s2).clone()
    Zc = st.Zc
    alpha = eng.planes(st)
    K.gat_edge_softmax(eng.fwd, st.s1, st.s2c, heads, eng.slope, eng.mode_id, n, alpha, st.beta, st.rowstat)
    out = {"n": n, "nnz": eng.nnz, "heads": heads, "d": d}
    o = torch.empty(n, F, device=dev)
    out["heads_kernel_ms"] = timed(lambda: K.spmm_heads(eng.fwd, alpha, Zc, o, heads, d))
    out["per_head_gather_ms"] = timed(lambda: [K.spmm(st.fwd_heads[k], Zc[:, k * d:(k + 1) * d], o[:, k * d:(k + 1) * d]) for k in range(heads)])
    dz = torch.empty(Zc.shape[0], Zc.shape[1], device=dev)
    out["heads_recompute_T_ms"] = timed(lambda: K.spmm_heads_recompute(eng.bwd, st.rowstat, st.s2c, eng.slope, eng.mode_id, G, dz, heads, d))
    # the tiled structure per head (values = that head's attention plane, storage order of eng.fwd)
    rp = eng.fwd.rowptr
    r = torch.repeat_interleave(torch.arange(n, device=dev), rp[1:] - rp[:-1])
    c = eng.fwd.col.to(torch.int64)
    tiled = []
    for k in range(heads):
        h = partition.csr_from_coo(r, c, alpha[k].clone(), n, Zc.shape[0], core=True)
        tiled.append(K.prepare(h))
        if k == 0:
            out["parts"] = {"gather": int(h.col.numel()), "strip": 0 if h.strip is None else h.strip.nnz,
                            "strip_records": 0 if h.strip is None else int(h.strip.rec.shape[0]),
                            "dense3": 0 if h.dense3 is None else h.dense3.nnz,
                            "dense3_blocks": 0 if h.dense3 is None else int(h.dense3.blk_row.numel())}
        del h
    def run_tiled():
        for k in range(heads):
            K.spmm(tiled[k], Zc[:, k * d:(k + 1) * d], o[:, k * d:(k + 1) * d])
    out["tiled_per_head_ms"] = timed(run_tiled)
    run_tiled(); torch.cuda.synchronize()
    out["tiled_vs_heads_max_abs"] = float((o - ref).abs().max())
    out["ref_max_abs"] = float(ref.abs().max())
    def dump():
        os.makedirs(os.path.dirname(os.path.join(ROOT, args.out)), exist_ok=True)
        with open(os.path.join(ROOT, args.out), "w") as fh:
            json.dump(out, fh, indent=1)
    dump()
    # the parts of one head, by CUDA events around the C-ABI calls: run with the pieces switched off
    A0 = tiled[0]
    import dataclasses
    for name, kw in (("gather_only", dict(strip=None, dense=None)), ("strip_only", dict(dense=None, ntasks=0)),
                     ("dense_only", dict(strip=None, ntasks=0))):
        try:
            Ax = dataclasses.replace(A0, launch_cache={}, **kw)
            out["head0_" + name + "_ms"] = timed(lambda: K.spmm(Ax, Zc[:, :d], o[:, :d]))
        except Exception as e:     # (a part switched off leaves slot lists the fix-up still walks: timing only)
            out["head0_" + name + "_ms"] = "failed: %s" % e
        dump()
    # f = 128 through the same structure: two heads' columns at once (what a two-plane record format would cost at best)
    out["head0_f128_ms"] = timed(lambda: K.spmm(dataclasses.replace(A0, launch_cache={}, ws=None), Zc[:, :2 * d], o[:, :2 * d]))
    print(json.dumps(out, indent=1))
    dump()


if __name__ == "__main__":
    main()
