#!/usr/bin/env python3
"""Golden vectors for the mini-batch driver FROM THE REFERENCE ITSELF: runs the unmodified
/root/reference/GPU/PGCN-Mini-batch.py:run() at P=1 under gloo (no quirks at P=1) with a seeded torch RNG
and records the printed epoch losses and the final weights.  Build container only."""
import contextlib, importlib.util, io, json, os, pickle, sys
import numpy as np, torch, torch.distributed as dist

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
SEED = 20260921

def main():
    spec = importlib.util.spec_from_file_location("ref_mb", os.path.join(REF, "GPU", "PGCN-Mini-batch.py"))
    ref = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29741", RANK="0", WORLD_SIZE="1")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=0, world_size=1)
    for name, mtx, n, f, bs in (("ref_minibatch_karateA", "karate.A.mtx", 34, 16, 8),
                                ("ref_minibatch_gemat11pA", "gemat11p.A.mtx", 4929, 8, 1500)):
        pv = os.path.join(OUT, name + ".partvec.pickle")
        with open(pv, "wb") as fh:
            pickle.dump([0] * n, fh)
        captured = []
        orig = ref.SequentialGCN
        real_adam = torch.optim.Adam
        def adam(params, **k):               # run() does not return the model: catch its parameters here
            params = list(params); captured.append(params); return real_adam(params, **k)
        torch.optim.Adam = adam
        torch.manual_seed(SEED)
        w0 = None
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            ref.run(0, 1, 3, f, os.path.join(OUT, mtx), pv, "gloo", bs)
        torch.optim.Adam = real_adam
        out = buf.getvalue()
        losses = [float(l.split("Loss")[1]) for l in out.splitlines() if "Loss" in l]
        final = captured[0]
        torch.manual_seed(SEED)
        init = orig(f, f)          # same RNG stream => the initial weights run() started from
        arrays = {}
        for i, (a, b) in enumerate(zip([init.gcn1, init.gcn2, init.gcn3], final)):
            arrays["w0_%d" % i] = a.linear.weight.detach().numpy().copy()
            arrays["w1_%d" % i] = b.detach().numpy().copy()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
        with open(os.path.join(OUT, name + ".json"), "w") as fh:
            json.dump({"mtx": mtx, "f": f, "batch_size": bs, "seed": SEED, "losses": losses, "stdout": out}, fh, indent=1)
        print(name, losses)
    dist.destroy_process_group()

if __name__ == "__main__":
    main()
