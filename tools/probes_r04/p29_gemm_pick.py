"""Which stock GEMM path for the n x f x f products of a layer without TunableOp's set-up cost?  Times the three products of a
layer (forward NT, dH NN, dW batched split-K) under (a) PyTorch's default library, (b) preferred_blas_library = hipBLASLt's
heuristic pick, (c) the shipped TunableOp choices; each mode in its own process, reporting the time to the first result too."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, time, json, importlib
t0 = time.time()
import torch
sys.path.insert(0, %r)
mode = sys.argv[1]
P = importlib.import_module("scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd.PGCN")
dev = torch.device("cuda:0")
n, f = 232965, 128
if mode == "hipblaslt":
    torch.backends.cuda.preferred_blas_library("cublaslt")
elif mode == "rocblas":
    torch.backends.cuda.preferred_blas_library("cublas")
x = torch.randn(n, f, device=dev); g = torch.randn(n, f, device=dev); w = torch.randn(f, f, device=dev)
torch.cuda.synchronize(); t1 = time.time()
if mode == "tunableop":
    P.tune_dense_gemms(n, f, dev)
ops = {"fwd x@w.t()": lambda: x @ w.t(), "dH g@w": lambda: g @ w, "dW split-K": lambda: P._LinearNoBias.weight_grad(g, x), "dW plain g.t()@x": lambda: g.t() @ x}
for k, fn in ops.items(): fn()
torch.cuda.synchronize(); t2 = time.time()
res = {}
for k, fn in ops.items():
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    res[k] = round(e0.elapsed_time(e1) / 50 * 1000, 1)
print(json.dumps({"mode": mode, "import_and_alloc_s": round(t1 - t0, 2), "first_products_s": round(t2 - t1, 2), "us": res}))
''' % ROOT
for mode in ("default", "hipblaslt", "rocblas", "tunableop", "default"):
    out = subprocess.run([sys.executable, "-c", CHILD, mode], capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    print(line[0] if line else "FAILED %s: %s" % (mode, out.stderr[-500:]))
