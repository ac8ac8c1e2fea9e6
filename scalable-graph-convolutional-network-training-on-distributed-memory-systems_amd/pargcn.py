"""The aggregation path driven with the CPU engine's training semantics.

Mirror of ``GCN()`` in /root/reference/Parallel-GCN/main.c:166-454 (sigmoid layers,
the loss of :318-323, the output gradient of :325-335, backward with ``A`` -- not
``A^T`` -- :376, dW Allreduce SUM :425, SGD :430), with both aggregations per layer
executed by the HIP engine (``AggregationEngine.forward``).  Dense algebra (n_p x f by
f x f) is plain torch -- plumbing around the graded path.  This is the form the
north-star parity clause names ("outputs match the reference Parallel-GCN CPU path").
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch


def _sigmoid(x):
    return 1 / (1 + torch.exp(-x))          # main.c:79-81


def train(engine, d: Sequence[int], W: Dict[int, torch.Tensor], H0: torch.Tensor, Y: torch.Tensor,
          Ymask: torch.Tensor, epochs: int = 3, alpha: float = 0.01, allreduce=None):
    """``H0, Y, Ymask`` hold OWNED rows only.  ``allreduce(t)`` sums a tensor over ranks in
    place (None for a single rank).  Returns (err per epoch, W, output H_{L-1})."""
    L = len(d) - 1
    n = d[0]
    W = {l: W[l].clone() for l in range(1, L)}
    H = {0: H0}
    Z = {}
    errs = []
    Ym = Ymask.bool()
    for _ in range(epochs):                                        # main.c:231
        for l in range(1, L):                                      # :233
            AH = engine.forward(H[l - 1])                          # :238-299
            Z[l] = AH @ W[l]                                       # :303
            H[l] = _sigmoid(Z[l])                                  # :308
        P = H[L - 1]
        T = torch.where(Ym, (-1.0 * Y.double() * torch.log(P.double())).float(), P)   # :70-73, 318
        err = T.sum().reshape(1)                                   # :320
        if allreduce is not None:
            allreduce(err)                                         # :321
        errs.append(float(err))
        D = torch.where(Ym, P - Y, P) / (P * (1 - P))              # :325-328
        s = _sigmoid(Z[L - 1])
        G = (D * (s * (1 - s))) / float(n)                         # :330-335
        for l in range(L - 1, 0, -1):                              # :338
            AG = engine.forward_symmetric_backward(G)              # :343-404 (A, relies on A = A^T)
            if l != 1:
                s = _sigmoid(Z[l - 1])
                Gn = (AG @ W[l].t()) * (s * (1 - s))               # :407-410
            dW = H[l - 1].t() @ AG                                 # :417
            if allreduce is not None:
                allreduce(dW)                                      # :425
            W[l] = W[l] - alpha * dW                               # :430
            if l != 1:
                G = Gn
    return errs, W, H[L - 1]
