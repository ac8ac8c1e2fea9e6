#!/usr/bin/env python3
"""Time the pieces of one GAT aggregation (forward + backward) on the benchmark graph (run under gpurun).
BASELINE config 5 shape: Reddit-sized R-MAT, 4 heads x 64."""
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
    ap.add_argument("--mode", default="standard")
    args = ap.parse_args()
    synth, partition, gat, kernels = pkg("synth"), pkg("partition"), pkg("gat"), pkg("kernels")
    dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
    n, row, col, val = synth.make_graph(args.workload, seed=0, device=dev)
    K = kernels.HipKernels(dev)
    part = partition.build_partition(row, col, val, n, torch.zeros(n, dtype=torch.int64, device=dev), 0, 1, with_transpose=False)
    heads, d = (1, args.heads * args.d) if args.mode == "reference" else (args.heads, args.d)
    eng = gat.GatEngine(part, K, dev, None, mode=args.mode)
    st = eng.new_layer_state(heads, d)
    F = heads * d
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    Z = torch.randn(n, F, device=dev, generator=gen)
    s1 = torch.randn(n, heads, device=dev, generator=gen); s2 = torch.randn(n, heads, device=dev, generator=gen)
    G = torch.randn(n, F, device=dev, generator=gen)
    nnz = eng.nnz
    out = {"n": n, "nnz": nnz, "heads": heads, "d": d, "mode": args.mode, "rows_block": int(eng.graph.fwd_block.numel())}
    out["forward_ms"] = timed(lambda: eng.forward(st, Z, s1, s2))
    out["backward_ms"] = timed(lambda: eng.backward(st, G))
    Zc, Fh = st.Zc, F
    alpha = eng.planes(st)            # (the engine's fused path keeps no planes; the stand-alone pieces below need them)
    out["softmax_ms"] = timed(lambda: K.gat_edge_softmax(eng.fwd, st.s1, st.s2c, heads, eng.slope, eng.mode_id, n, alpha, st.beta, st.rowstat))
    out["softmax_stats_only_ms"] = timed(lambda: K.gat_edge_softmax(eng.fwd, st.s1, st.s2c, heads, eng.slope, eng.mode_id, n, None, st.beta, st.rowstat))
    o1 = torch.empty(n, F, device=dev); V = torch.empty(n, F + (heads + 3) // 4 * 4, device=dev)
    cov = [True]
    def fwd2():
        cov[0] = K.spmm_heads_forward2(eng.fwd, st.rowstat, st.s2c, eng.slope, eng.mode_id, Zc, o1, V, heads, d)
    out["heads_forward2_ms"] = timed(fwd2)
    out["heads_forward2_covered"] = bool(cov[0])
    o = torch.empty(n, F, device=dev)
    out["spmm_heads_ms"] = timed(lambda: [K.spmm(st.fwd_heads[k], Zc[:, k * d:(k + 1) * d], o[:, k * d:(k + 1) * d]) for k in range(heads)])
    t = (G.view(n, heads, d) * st.out.view(n, heads, d)).sum(-1).contiguous()
    de = torch.empty(max(nnz, 1), heads, device=dev); ds1 = torch.empty(n, heads, device=dev)   # de: entry-major
    out["edge_grad_ms"] = timed(lambda: K.gat_edge_grad(eng.fwd, st.s1, st.s2c, alpha, st.beta, Zc, G, t, heads, d, eng.slope, eng.mode_id, de, ds1))
    ds1p = torch.empty(n, 8, heads, device=dev)
    out["edge_grad_sliced_ms"] = timed(lambda: K.gat_edge_grad_sliced(eng.fwd, st.s1, st.s2c, alpha, st.beta, Zc, G, t, heads, d, eng.slope, eng.mode_id, de, ds1p))
    at = eng._plane_scratch("alpha_t", heads)
    out["permute_ms"] = timed(lambda: K.csr_permute(alpha, eng.perm, at))
    out["weights_t_ms"] = timed(lambda: K.gat_edge_weights_t(eng.bwd, st.s2c, st.rowstat, heads, eng.slope, eng.mode_id, at))
    ds2 = torch.empty(n, heads, device=dev)
    out["row_sums_ms"] = timed(lambda: K.csr_row_sums(eng.bwd, eng.perm, de, heads, ds2))
    ds1t = torch.empty(n, heads, device=dev)
    out["edge_grad_tasks_ms"] = timed(lambda: K.gat_edge_grad_tasks(eng.fwd, st.s1, st.s2c, alpha, st.beta, Zc, G, t, heads, d, eng.slope, eng.mode_id, de, ds1t))
    o2 = torch.empty(n, F, device=dev)
    out["spmm_heads_kernel_ms"] = timed(lambda: K.spmm_heads(eng.fwd, alpha, Zc, o2, heads, d))
    dzc = torch.empty(Zc.shape[0], Zc.shape[1], device=dev)
    out["heads_recompute_T_ms"] = timed(lambda: K.spmm_heads_recompute(eng.bwd, st.rowstat, st.s2c, eng.slope, eng.mode_id, G, dzc, heads, d))
    de_t = torch.empty(max(nnz, 1), heads, device=dev)
    ok = [True]
    def fused():
        ok[0] = K.spmm_heads_grad(eng.bwd, st.rowstat, st.s2c, eng.slope, eng.mode_id, G, Zc, t, dzc, de_t, heads, d)
    out["heads_grad_fused_ms"] = timed(fused)
    out["heads_grad_fused_covered"] = bool(ok[0])
    out["row_sums_fwd_inv_ms"] = timed(lambda: K.csr_row_sums(eng.fwd, eng.inv_perm, de_t, heads, ds1))
    # algorithmic bytes of the streams: softmax 4 (col) + 4 (alpha) per entry and head
    out["softmax_alg_GBs"] = nnz * heads * 8 / out["softmax_ms"] / 1e6
    out["edge_grad_gather_TBs"] = nnz * F * 4 / out["edge_grad_ms"] / 1e9
    out["spmm_gather_TBs"] = nnz * F * 4 / out["spmm_heads_ms"] / 1e9
    print(json.dumps(out, indent=1))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "gat_probe_%s.json" % args.mode), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
