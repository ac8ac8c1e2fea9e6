"""cold-process probe: what does TunableOp do with the shipped result file?"""
import os, sys, time, importlib
t0 = time.time()
import torch
import torch.cuda.tunable as tunable
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
P = importlib.import_module("scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd.PGCN")
dev = torch.device("cuda:0")
torch.zeros(1, device=dev); torch.cuda.synchronize(); print("context %.2fs" % (time.time() - t0))
tunable.enable(True)
tunable.set_filename("/tmp/probe_tunable.csv")
print("validators", tunable.get_validators())
ok = tunable.read_file(P.TUNABLEOP_SHIPPED)
print("read_file ->", ok, "results", len(tunable.get_results()))
tunable.set_max_tuning_duration(30); tunable.set_max_tuning_iterations(20)
tunable.tuning_enable(True)
n, f = 232965, 128
x = torch.zeros((n, f), device=dev); g = torch.zeros((n, f), device=dev); w = torch.zeros((f, f), device=dev)
for name, fn in (("x@w.t", lambda: x @ w.t()), ("g@w", lambda: g @ w), ("weight_grad", lambda: P._LinearNoBias.weight_grad(g, x))):
    t = time.time(); fn(); torch.cuda.synchronize(); print(name, "first call %.2fs" % (time.time() - t), "results now", len(tunable.get_results()))
    t = time.time(); fn(); torch.cuda.synchronize(); print(name, "second call %.4fs" % (time.time() - t))
tunable.tuning_enable(False)
print(open("/tmp/probe_tunable.csv").read() if os.path.exists("/tmp/probe_tunable.csv") else "no file written")
