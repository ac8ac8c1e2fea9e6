# how the packed GAT projection's products should be cut: one n x 256 x 264 product, or the recorded n x 256 x 256 kernel + a skinny one
import importlib, sys, torch, time
sys.path.insert(0, '.')
P = importlib.import_module("scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd.PGCN")
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
n = 232965
x = torch.randn(n, 256, device=dev)
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
for N in (256, 264, 272, 288, 320, 8, 16):
    w = torch.randn(N, 256, device=dev); g = torch.randn(n, N, device=dev)
    print('N=%3d  x.W^T: mm_nt %4.0f us torch %4.0f | g.W: mm_nn %4.0f torch %4.0f | wgrad(split-K) %4.0f plain g^T.x %4.0f' % (
        N, t(lambda: P.mm_nt(x, w)), t(lambda: x @ w.t()), t(lambda: P.mm_nn(g, w)), t(lambda: g @ w), t(lambda: P._LinearNoBias.weight_grad(g, x)), t(lambda: g.t() @ x)))
# skinny products as a matrix-vector style reduction over the 256 inputs
w8 = torch.randn(8, 256, device=dev)
print('N=8 via (x[:, None, :] * w8).sum(-1): %4.0f us' % t(lambda: (x.unsqueeze(1) * w8).sum(-1)))
print('N=8 addmm into out view: %4.0f us' % t(lambda: torch.mm(x, w8.t())))
