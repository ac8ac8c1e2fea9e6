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
    # config 3's shape at quarter scale: the largest products-shaped SBM graph the reference's serial PaToH front-end
    # partitions within two hours here (tools/make_partvecs.py; the full size ran > 4 h) -- its hp vector is committed
    "products4": (612257, 30929570, 128, 3),
    # mid-size case small enough for the reference's 32-bit PaToH / METIS front-ends: part vectors written by
    # them are committed under tests/golden/partvec/ (tools/make_partvecs.py)
    "mid": (131072, 4194304, 64, 2),
    # BASELINE config 4 (8 GPUs; generated rank-locally as binary CSR shards, tools/make_shards.py)
    "papers": (111059956, 1615685872, 64, 2),
}


class PortableRng:
    """Counter-based generator (splitmix64 finaliser on 64-bit integers, wrap-around arithmetic): the SAME
    stream on every device and torch build, unlike torch.Generator whose CPU and GPU streams differ.  Used
    where a graph has to be reproduced exactly in another place (the `mid` workload: part vectors written by
    the reference's partitioners in the build container must fit the graph generated on the GPU box)."""

    def __init__(self, seed: int, device):
        self.device = torch.device(device)
        self.counter = (int(seed) * 0x9E3779B97F4A7C15 + 0x1234567) & 0x7FFFFFFFFFFFFFFF

    def bits(self, m: int) -> torch.Tensor:
        """m non-negative 62-bit integers."""
        x = torch.arange(m, dtype=torch.int64, device=self.device) + self.counter
        self.counter = (self.counter + m + 0x632BE5AB) & 0x7FFFFFFFFFFFFFFF
        for mult in (-7046029254386353131, -4658895280553007687):          # 0x9E37..15, 0xBF58476D1CE4E5B9 as int64
            x = (x ^ ((x >> 30) & 0x3FFFFFFFF)) * mult
        x = x ^ ((x >> 31) & 0x1FFFFFFFF)
        return (x >> 2) & 0x3FFFFFFFFFFFFFFF

    def rand(self, m: int) -> torch.Tensor:
        """float64 uniforms in [0, 1) with 53 bits (exact integer -> float conversions only)."""
        return (self.bits(m) >> 9).to(torch.float64) * (1.0 / 9007199254740992.0)

    def randint(self, m: int, high) -> torch.Tensor:
        return self.bits(m) % high

    def randperm(self, n: int) -> torch.Tensor:
        return torch.argsort(self.bits(n), stable=True)


class _TorchRng:
    """The torch.Generator streams the r01 benchmark graphs were drawn from (device specific)."""

    def __init__(self, seed: int, device):
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed)

    def rand(self, m: int) -> torch.Tensor:
        return torch.rand(m, generator=self.gen, device=self.device)

    def randperm(self, n: int) -> torch.Tensor:
        return torch.randperm(n, generator=self.gen, device=self.device)


def _rmat_pairs(m: int, scale: int, gen, device) -> Tuple[torch.Tensor, torch.Tensor]:
    a, b, c = 0.57, 0.19, 0.19
    r = torch.zeros(m, dtype=torch.int64, device=device)
    cidx = torch.zeros(m, dtype=torch.int64, device=device)
    for _ in range(scale):
        u = gen.rand(m)
        rb = (u >= a + b).to(torch.int64)
        cb = (((u >= a) & (u < a + b)) | (u >= a + b + c)).to(torch.int64)
        r = (r << 1) | rb
        cidx = (cidx << 1) | cb
    return r, cidx


def rmat_undirected(n: int, nnz_directed: int, seed: int = 0, device="cpu", portable: bool = False) -> torch.Tensor:
    """Returns sorted unique keys ``row * n + col`` of a symmetric pattern with exactly
    ``nnz_directed`` (rounded down to even) off-diagonal entries.  ``portable``: draw from the
    device-independent counter-based stream instead of torch.Generator."""
    device = torch.device(device)
    gen = PortableRng(seed, device) if portable else _TorchRng(seed, device)
    target = nnz_directed // 2
    max_pairs = n * (n - 1) // 2
    if target > max_pairs:
        raise ValueError("graph denser than complete")
    scale = max(1, math.ceil(math.log2(max(n, 2))))
    perm = gen.randperm(n)
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
        drop = gen.randperm(keys.numel())[:target]
        keys = keys[torch.sort(drop).values]
    lo, hi = keys // n, keys % n
    full = torch.cat([lo * n + hi, hi * n + lo])
    return torch.sort(full).values


def sbm_undirected(n: int, nnz_directed: int, seed: int = 0, device="cpu", community: int = 2048,
                   p_in: float = 0.7) -> torch.Tensor:
    """Degree-corrected planted-partition graph (the second stand-in): communities of
    uneven size around ``community`` vertices, expected degrees with a Pareto(2) tail, an edge end picked
    inside the source's community with probability ``p_in`` and anywhere (degree weighted) otherwise.  Real
    Reddit / products get their reuse from community structure, not from hubs alone.  Vertex ids are randomly
    permuted: no locality is inherited from the generator.  Drawn from the portable counter-based stream with
    integer weights and correctly rounded float64 operations only, so the graph is the same on every device.
    Returns sorted unique keys ``row * n + col`` of a symmetric pattern with exactly ``nnz_directed`` (even)
    off-diagonal entries."""
    device = torch.device(device)
    gen = PortableRng(seed + 7919, device)
    target = nnz_directed // 2
    if target > n * (n - 1) // 2:
        raise ValueError("graph denser than complete")
    # communities: consecutive blocks of an internal numbering, sizes proportional to 1 + 3 u^2 (1x .. 4x)
    nc = max(1, n // max(community, 1))
    u = gen.rand(nc)
    raw = (1.0 + 3.0 * u * u)
    sizes = torch.clamp((raw * (n / float(raw.sum().item()))).floor().long(), min=1)
    short = n - int(sizes.sum())
    sizes[-1] += short
    if int(sizes[-1]) < 1:
        sizes = torch.full((nc,), n // nc, dtype=torch.int64, device=device)
        sizes[-1] += n - int(sizes.sum())
    cstart = torch.cumsum(sizes, 0) - sizes
    comm = torch.repeat_interleave(torch.arange(nc, device=device), sizes)
    # expected degrees: Pareto(alpha = 2) tail, w = 1 / sqrt(1 - u), truncated; integer weights (x 1024)
    w = torch.clamp(1.0 / torch.sqrt(1.0 - gen.rand(n)), max=float(max(n // 40, 2)))
    wi = torch.clamp((w * 1024.0).floor().long(), min=1)
    cw = torch.cumsum(wi, 0)                                  # inclusive cumulative weights: exact int64
    total = int(cw[-1])
    c_lo = (cw - wi)[cstart]
    c_hi = cw[cstart + sizes - 1]                             # weight range [c_lo, c_hi) of every community
    perm = gen.randperm(n)
    thresh = int(p_in * (1 << 30))
    keys = torch.zeros(0, dtype=torch.int64, device=device)
    rounds = 0
    while keys.numel() < target:
        need = target - keys.numel()
        m = int(need * 1.3) + 1024
        src = torch.searchsorted(cw, gen.randint(m, total), right=True).clamp_(max=n - 1)
        inside = gen.randint(m, 1 << 30) < thresh
        cs = comm[src]
        lo = torch.where(inside, c_lo[cs], torch.zeros_like(cs))
        span = torch.where(inside, c_hi[cs] - c_lo[cs], torch.full_like(cs, total))
        dst = torch.searchsorted(cw, lo + gen.bits(m) % span, right=True).clamp_(max=n - 1)
        r, c = perm[src], perm[dst]
        ok = r != c
        a, b = torch.minimum(r[ok], c[ok]), torch.maximum(r[ok], c[ok])
        keys = torch.unique(torch.cat([keys, a * n + b]))
        rounds += 1
        if rounds > 64:
            raise RuntimeError("sbm generator did not converge")
    if keys.numel() > target:
        drop = gen.randperm(keys.numel())[:target]
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


LEGACY_STREAM = ("cora", "reddit", "products")
PORTABLE_DEFAULT = True


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
        # reddit / products keep the torch.Generator stream of r01 (device specific, numbers stay comparable);
        # every other graph is drawn from the portable stream
        legacy = isinstance(name_or_n, str) and name_or_n in LEGACY_STREAM
        keys = rmat_undirected(n, nnz, seed, device, portable=not legacy and PORTABLE_DEFAULT)
    else:
        raise ValueError("unknown generator %r" % generator)
    row, col, val = normalized_adjacency(n, keys)
    return n, row, col, val


def rmat_shard_keys(n: int, pairs: int, rank: int, partvec: torch.Tensor, seed: int = 0, device="cpu",
                    chunk: int = 1 << 24) -> torch.Tensor:
    """Rank-local generation of a papers100M-scale graph: every rank walks the SAME portable stream of
    ``pairs`` undirected R-MAT pairs chunk by chunk and keeps only the entries of the rows it owns (both
    directions of a pair, self loops of its vertices) -- no rank ever holds the global pattern.  Duplicates
    of a coordinate land on the same rank, so the local de-duplication is exact.  Returns sorted unique keys
    ``row * n + col`` (rows owned by ``rank``).  The union over the ranks is a symmetric pattern with all
    self loops; its size is whatever the de-duplication leaves (no global count is taken)."""
    device = torch.device(device)
    gen = PortableRng(seed, device)
    scale = max(1, math.ceil(math.log2(max(n, 2))))
    perm = gen.randperm(n)
    part = torch.as_tensor(partvec).to(device=device, dtype=torch.int64)
    own = torch.nonzero(part == rank).reshape(-1)
    keys = own * n + own                                         # self loops (A + I)
    done = 0
    while done < pairs:
        m = min(chunk, pairs - done)
        r, c = _rmat_pairs(m, scale, gen, device)
        r, c = perm[r % n], perm[c % n]
        ok = r != c
        r, c = r[ok], c[ok]
        mine_r, mine_c = part[r] == rank, part[c] == rank
        keys = torch.unique(torch.cat([keys, r[mine_r] * n + c[mine_r], c[mine_c] * n + r[mine_c]]))
        done += m
    return keys


def rmat_all_keys(n: int, pairs: int, seed: int = 0, device="cpu", chunk: int = 1 << 24) -> torch.Tensor:
    """The GLOBAL pattern of the stream ``rmat_shard_keys`` walks (sorted unique keys ``row * n + col``, both directions
    of every pair, all self loops): the union of the ranks' key sets, whatever the part vector.  For tools that stand in
    for a whole job on one big device (tools/make_shards.py --only-rank: the global degree vector of a papers100M-scale
    graph is 13 GB of keys on a 288 GB GPU); the ranks of a real job never call this."""
    device = torch.device(device)
    gen = PortableRng(seed, device)
    scale = max(1, math.ceil(math.log2(max(n, 2))))
    perm = gen.randperm(n)
    own = torch.arange(n, dtype=torch.int64, device=device)
    parts = [own * n + own]
    done = 0
    while done < pairs:
        m = min(chunk, pairs - done)
        r, c = _rmat_pairs(m, scale, gen, device)
        r, c = perm[r % n], perm[c % n]
        ok = r != c
        r, c = r[ok], c[ok]
        parts.append(r * n + c)
        parts.append(c * n + r)
        done += m
        if len(parts) > 64:                                       # (bounded list: fold now and then)
            parts = [torch.unique(torch.cat(parts))]
    return torch.unique(torch.cat(parts))


def shard_normalize(n: int, keys: torch.Tensor, degree: torch.Tensor):
    """(row, col, val) of a rank's entries, val = d_r^-1/2 d_c^-1/2 with the GLOBAL degree vector (row counts of
    A + I, summed over the ranks: one all-reduce of an n-vector), like preprocess/GrB-GNN-IDG.py:45-68."""
    row, col = keys // n, keys % n
    dinv = 1.0 / torch.sqrt(degree.to(torch.float64))
    return row, col, (dinv[row] * dinv[col]).to(torch.float32)


def block_partvec(n: int, P: int) -> torch.Tensor:
    """Contiguous blocks of ceil(n / P) vertex ids."""
    return torch.clamp(torch.arange(n, dtype=torch.int64) // -(-n // P), max=P - 1)


def random_partvec(n: int, P: int, seed: int = 0) -> torch.Tensor:
    """Uniform random part vector (GCN-HP/main.cpp:133-142 ``partition_random``, with a
    fixed seed instead of the clock)."""
    gen = torch.Generator()
    gen.manual_seed(seed)
    return torch.randint(0, P, (n,), generator=gen, dtype=torch.int64)
