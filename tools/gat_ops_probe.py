#!/usr/bin/env python3
"""Which framework ops launch what inside one training step of the GAT bench line (torch.profiler, 3 steps)."""
import importlib, os, sys, torch, torch.nn as nn
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
PKG = "scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd"
pkg = lambda s: importlib.import_module(PKG + "." + s)
synth, partition, kernels, G, gat = pkg("synth"), pkg("partition"), pkg("kernels"), pkg("PGAT"), pkg("gat")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
n, row, col, val = synth.make_graph("reddit", seed=0, device=dev)
part = partition.build_partition(row, col, val, n, torch.zeros(n, dtype=torch.int64), 0, 1)
K = kernels.HipKernels(dev); eng = gat.GatEngine(part, K, dev, None, mode="standard")
heads, F, L = 4, 256, 3
G.device, G.myrank, G.world_size, G.heads, G._engine_current = dev, 0, 1, heads, eng
pkg("PGCN").tune_dense_gemms(part.n_local, F, dev, fout=F + 2 * heads)
torch.manual_seed(0)
model = nn.Sequential(*[G.PGAT(eng, F, F, heads=heads) for _ in range(L)]).to(dev)
G.initiliaze_parameters(model)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
H = torch.rand(part.n_local, F, device=dev).requires_grad_(True); labels = part.owned.to(dev) % F
def step():
    loss = G.local_loss(model(H), labels, n); opt.zero_grad(); loss.backward(); G.sum_gradients(model); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=90))
