"""Seeded synthetic graphs with the shape of the benchmark datasets.

The real datasets (Reddit, ogbn-products, ...) are not in the image and there is no
network, so ``bench.py`` and the large-size tests use an R-MAT graph of matching
n / nnz / degree skew: (a,b,c,d) = (0.57,0.19,0.19,0.05), ids folded mod n, vertex
ids randomly permuted (so no locality is inherited from the generator), symmetrised,
de-duplicated, self-loops added and normalised exactly like the reference's
``preprocess/GrB-GNN-IDG.py:45-68``:  A_hat = D^-1/2 (A + I) D^-1/2  (fp32).

Pure torch tensor ops: runs on the GPU for the 114 M-edge benchmark graph (a few
hundred ms) and on the CPU for tests.  Plumbing, not the product path.
"""
from __future__ import annotations

import math
from typing import Tuple

import torch

SHAPES = {
    # name: (n, nnz directed without self loops, f, layers)      BASELINE.json configs
    "cora": (2708, 10556, 16, 2),
    "reddit": (232965, 114615892, 128, 3),
    "products": (2449029, 123718280, 128, 3),
    # mid-size case small enough for the reference's 32-bit PaToH / METIS front-ends: part vectors written by
    # them are committed under tests/golden/partvec/ (tools/make_partvecs.py)
    "mid": (131072, 4194304, 64, 2),
}


def _rmat_pairs(m: int, scale: int, gen: torch.Generator, device) -> Tuple[torch.Tensor, torch.Tensor]:
    a, b, c = 0.57, 0.19, 0.19
    r = torch.zeros(m, dtype=torch.int64, device=device)
    cidx = torch.zeros(m, dtype=torch.int64, device=device)
    for _ in range(scale):
        u = torch.rand(m, generator=gen, device=device)
        rb = (u >= a + b).to(torch.int64)
        cb = (((u >= a) & (u < a + b)) | (u >= a + b + c)).to(torch.int64)
        r = (r << 1) | rb
        cidx = (cidx << 1) | cb
    return r, cidx


def rmat_undirected(n: int, nnz_directed: int, seed: int = 0, device="cpu") -> torch.Tensor:
    """Returns sorted unique keys ``row * n + col`` of a symmetric pattern with exactly
    ``nnz_directed`` (rounded down to even) off-diagonal entries."""
    device = torch.device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    target = nnz_directed // 2
    max_pairs = n * (n - 1) // 2
    if target > max_pairs:
        raise ValueError("graph denser than complete")
    scale = max(1, math.ceil(math.log2(max(n, 2))))
    perm = torch.randperm(n, generator=gen, device=device)
    keys = torch.zeros(0, dtype=torch.int64, device=device)
    rounds = 0
    while keys.numel() < target:
        need = target - keys.numel()
        m = int(need * 1.6) + 1024
        r, c = _rmat_pairs(m, scale, gen, device)
        r, c = perm[r % n], perm[c % n]
        ok = r != c
        lo, hi = torch.minimum(r[ok], c[ok]), torch.maximum(r[ok], c[ok])
        keys = torch.unique(torch.cat([keys, lo * n + hi]))
        rounds += 1
        if rounds > 64:
            raise RuntimeError("rmat generator did not converge")
    if keys.numel() > target:   # drop a seeded random subset to hit the size exactly
        drop = torch.randperm(keys.numel(), generator=gen, device=device)[:target]
        keys = keys[torch.sort(drop).values]
    lo, hi = keys // n, keys % n
    full = torch.cat([lo * n + hi, hi * n + lo])
    return torch.sort(full).values


def sbm_undirected(n: int, nnz_directed: int, seed: int = 0, device="cpu", community: int = 2048,
                   p_in: float = 0.7, alpha: float = 2.1) -> torch.Tensor:
    """Degree-corrected planted-partition graph (the second stand-in): communities of
    lognormal size around ``community`` vertices, expected degrees from a truncated Pareto(alpha) tail, an edge
    end picked inside the source's community with probability ``p_in`` and anywhere else (degree weighted)
    otherwise.  Real Reddit / products get their reuse from community structure, not from hubs alone.  Vertex
    ids are randomly permuted: no locality is inherited from the generator.  Returns sorted unique keys
    ``row * n + col`` of a symmetric pattern with exactly ``nnz_directed`` (even) off-diagonal entries."""
    device = torch.device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed + 7919)
    target = nnz_directed // 2
    if target > n * (n - 1) // 2:
        raise ValueError("graph denser than complete")
    # communities: consecutive blocks of lognormal size in an internal numbering
    nc = max(1, n // max(community, 1))
    raw = torch.exp(0.5 * torch.randn(nc, generator=gen, device=device))
    sizes = torch.clamp((raw / raw.sum() * n).long(), min=1)
    sizes[-1] += n - int(sizes.sum())
    if int(sizes[-1]) < 1:
        sizes = torch.full((nc,), n // nc, dtype=torch.int64, device=device)
        sizes[-1] += n - int(sizes.sum())
    cstart = torch.cumsum(sizes, 0) - sizes
    comm = torch.repeat_interleave(torch.arange(nc, device=device), sizes)
    # expected degrees: Pareto tail, truncated
    u = torch.rand(n, generator=gen, device=device, dtype=torch.float64)
    w = torch.clamp((1.0 - u) ** (-1.0 / alpha), max=float(max(n // 40, 2)))
    cw = torch.cumsum(w, 0)                                   # inclusive cumulative weights (internal numbering)
    cw0 = cw - w
    c_lo, c_hi = cw0[cstart], cw[cstart + sizes - 1]          # weight range of every community
    perm = torch.randperm(n, generator=gen, device=device)
    keys = torch.zeros(0, dtype=torch.int64, device=device)
    rounds = 0
    while keys.numel() < target:
        need = target - keys.numel()
        m = int(need * 1.3) + 1024
        src = torch.searchsorted(cw, torch.rand(m, generator=gen, device=device, dtype=torch.float64) * cw[-1]).clamp_(max=n - 1)
        inside = torch.rand(m, generator=gen, device=device) < p_in
        cs = comm[src]
        lo = torch.where(inside, c_lo[cs], torch.zeros_like(c_lo[cs]))
        hi = torch.where(inside, c_hi[cs], cw[-1].expand_as(lo))
        dst = torch.searchsorted(cw, lo + torch.rand(m, generator=gen, device=device, dtype=torch.float64) * (hi - lo)).clamp_(max=n - 1)
        r, c = perm[src], perm[dst]
        ok = r != c
        a, b = torch.minimum(r[ok], c[ok]), torch.maximum(r[ok], c[ok])
        keys = torch.unique(torch.cat([keys, a * n + b]))
        rounds += 1
        if rounds > 64:
            raise RuntimeError("sbm generator did not converge")
    if keys.numel() > target:
        drop = torch.randperm(keys.numel(), generator=gen, device=device)[:target]
        keys = keys[torch.sort(drop).values]
    a, b = keys // n, keys % n
    return torch.sort(torch.cat([a * n + b, b * n + a])).values


def normalized_adjacency(n: int, keys: torch.Tensor):
    """A_hat = D^-1/2 (A + I) D^-1/2 for a symmetric off-diagonal pattern given as sorted
    keys.  Returns COO (row, col, val) sorted by (row, col), self loops included."""
    device = keys.device
    diag = torch.arange(n, dtype=torch.int64, device=device)
    allk = torch.sort(torch.cat([keys, diag * n + diag])).values
    row, col = allk // n, allk % n
    deg = torch.bincount(row, minlength=n).to(torch.float64)   # row sums of A + I (= col sums)
    dinv = (1.0 / torch.sqrt(deg))
    val = (dinv[row] * dinv[col]).to(torch.float32)
    return row, col, val


def make_graph(name_or_n, nnz: int = None, seed: int = 0, device="cpu", generator: str = "rmat"):
    """(n, row, col, val) of a normalised synthetic graph; ``name_or_n`` is a key of
    SHAPES or an explicit vertex count (then ``nnz`` is required).  ``generator``: "rmat" (the
    headline stand-in) or "sbm" (planted partition with a power-law degree tail)."""
    if isinstance(name_or_n, str):
        n, nnz, _, _ = SHAPES[name_or_n]
    else:
        n = int(name_or_n)
    if generator == "sbm":
        keys = sbm_undirected(n, nnz, seed, device, community=max(64, min(2048, n // 16)))
    elif generator == "rmat":
        keys = rmat_undirected(n, nnz, seed, device)
    else:
        raise ValueError("unknown generator %r" % generator)
    row, col, val = normalized_adjacency(n, keys)
    return n, row, col, val


def block_partvec(n: int, P: int) -> torch.Tensor:
    """Contiguous blocks of ceil(n / P) vertex ids."""
    return torch.clamp(torch.arange(n, dtype=torch.int64) // -(-n // P), max=P - 1)


def random_partvec(n: int, P: int, seed: int = 0) -> torch.Tensor:
    """Uniform random part vector (GCN-HP/main.cpp:133-142 ``partition_random``, with a
    fixed seed instead of the clock)."""
    gen = torch.Generator()
    gen.manual_seed(seed)
    return torch.randint(0, P, (n,), generator=gen, dtype=torch.int64)
