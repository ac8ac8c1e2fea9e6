#!/usr/bin/env python3
"""Golden vectors for the GAT path FROM THE REFERENCE ITSELF: imports the unmodified
/root/reference/GPU/PGAT.py (its unused ``torchvision`` import is satisfied by an empty stub
module -- torchvision is not installed here), builds its dense ``PGAT`` layers at P=1 with
seeded parameters and records

  * operator level (``ref_gat_<graph>.npz``): H, per layer W / attention vector / output, the
    logits, the loss of run()'s objective (log_softmax + nll_loss, labels i % f, PGAT.py:214-216)
    and every gradient (dH, dW, dattention);
  * run() level (``ref_gat_run_karateA.json``): the 50 printed epoch losses and the final
    parameters of ``run(0, 1, L, f, ...)`` under gloo with a seeded torch RNG.

Build container only (needs /root/reference); the fixtures travel, this script documents them."""
import contextlib, importlib.util, io, json, os, sys, types
import numpy as np, torch, torch.distributed as dist
from scipy.io import mmread

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
SEED = 20260921


def load_ref():
    for name in ("torchvision", "torchvision.datasets", "torchvision.transforms"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision"].datasets = sys.modules["torchvision.datasets"]
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    spec = importlib.util.spec_from_file_location("ref_pgat", os.path.join(REF, "GPU", "PGAT.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    return ref


def operator_level(ref, name, mtx, f, L):
    A = mmread(os.path.join(OUT, mtx))
    n = A.shape[0]
    ref.device = torch.device("cpu")
    ref.myrank, ref.world_size = 0, 1
    ref.send_map, ref.recv_map = ref.compute_communication_maps(A, [0] * n, 0, 1)
    ref.send_buffers, ref.recv_buffers = {}, {}
    Ad = ref.get_partitiont_of_adjacency_matrix(A, [0] * n, 0)
    torch.manual_seed(SEED)
    layers = [ref.PGAT(Ad, f, f) for _ in range(L)]
    gen = torch.Generator().manual_seed(SEED + 1)
    H = (torch.rand(n, f, generator=gen) * 2 - 1).requires_grad_(True)
    ref.X = torch.zeros(H.shape)
    outs, x = [], H
    for layer in layers:
        x = layer(x)
        x.retain_grad()
        outs.append(x)
    labels = torch.arange(0, n) % f
    loss = torch.nn.functional.nll_loss(torch.nn.functional.log_softmax(x, 1), labels)
    loss.backward()
    arrays = {"H": H.detach().numpy(), "dH": H.grad.numpy(), "loss": np.array(loss.item())}
    for i, (layer, o) in enumerate(zip(layers, outs)):
        arrays["W_%d" % i] = layer.linear.weight.detach().numpy()
        arrays["a_%d" % i] = layer.attention.detach().numpy()
        arrays["dW_%d" % i] = layer.linear.weight.grad.numpy()
        arrays["da_%d" % i] = layer.attention.grad.numpy()
        arrays["out_%d" % i] = o.detach().numpy()
        arrays["dout_%d" % i] = o.grad.numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    with open(os.path.join(OUT, name + ".json"), "w") as fh:
        json.dump({"mtx": mtx, "n": n, "f": f, "layers": L, "seed": SEED, "loss": loss.item(),
                   "edges_positive": int((Ad > 0).sum())}, fh, indent=1)
    print(name, "loss", loss.item())


def run_level(ref, name, mtx, f, L):
    A = mmread(os.path.join(OUT, mtx))
    n = A.shape[0]
    pv = os.path.join(OUT, name + ".partvec")
    with open(pv, "w") as fh:
        fh.write(" ".join(["0"] * n) + "\n")
    captured = []
    real_adam = torch.optim.Adam
    def adam(params, **k):                   # run() does not return the model: catch its parameters here
        params = list(params); captured.append(params); return real_adam(params, **k)
    torch.optim.Adam = adam
    torch.manual_seed(SEED)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ref.run(0, 1, L, f, os.path.join(OUT, mtx), pv, "gloo")
    torch.optim.Adam = real_adam
    out = buf.getvalue()
    losses = [float(l.split("Loss")[1]) for l in out.splitlines() if "Loss" in l]
    ref.device = torch.device("cpu")
    torch.manual_seed(SEED)                  # same RNG stream => the parameters run() started from
    Ad = torch.zeros(n, n)
    init = [ref.PGAT(Ad, f, f) for _ in range(L)]
    arrays = {}
    final = captured[0]
    for i, layer in enumerate(init):
        arrays["W0_%d" % i] = layer.linear.weight.detach().numpy().copy()
        arrays["a0_%d" % i] = layer.attention.detach().numpy().copy()
    # nn.Sequential.parameters(): per layer (attention, linear.weight) in registration order
    names = [k for k, _ in torch.nn.Sequential(*init).named_parameters()]
    for k, p in zip(names, final):
        arrays["final_" + k.replace(".", "_")] = p.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    with open(os.path.join(OUT, name + ".json"), "w") as fh:
        json.dump({"mtx": mtx, "n": n, "f": f, "layers": L, "seed": SEED, "losses": losses, "param_names": names,
                   "stdout": out}, fh, indent=1)
    print(name, losses[:3], "...", losses[-1])


def main():
    ref = load_ref()
    torch.set_num_threads(1)
    operator_level(ref, "ref_gat_karateA", "karate.A.mtx", 16, 2)
    operator_level(ref, "ref_gat_gemat11pA", "gemat11p.A.mtx", 8, 2)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29743", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("gloo", rank=0, world_size=1)
    run_level(ref, "ref_gat_run_karateA", "karate.A.mtx", 8, 2)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
