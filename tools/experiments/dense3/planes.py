"""Host side of the bf16 three-plane MFMA tiles (tools/experiments/dense3): the exact three-way bf16 split of an
fp32 array and the A-operand plane layout of pgcn_spmm_dense_bf16x3_f32.  `integrate.patch` puts these functions into
partition.py (HostDense.planes, tuning.dense_bf16x3); tests/test_dense_bf16x3.py pins them."""
import torch

TR = TC = 128


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    """fp32 -> the nearest bf16 number (ties to even), returned as fp32.  Inf / NaN pass through."""
    b = x.contiguous().view(torch.int32)
    r = (b + 0x7FFF + ((b >> 16) & 1)) & ~0xFFFF                  # (two's-complement int32 add = the usual uint32 trick)
    r = torch.where(torch.isfinite(x), r, b & ~0xFFFF | ((b & 0xFFFF) != 0).to(torch.int32) << 16)
    return r.view(torch.float32)


def bf16_split3(x: torch.Tensor):
    """x = x1 + x2 + x3, every term a bf16 number (returned as fp32): x1 = bf16(x), x2 = bf16(x - x1),
    x3 = bf16(x - x1 - x2).  Both subtractions are exact in fp32 and the last remainder has at most 8 significant
    bits, so the sum is EXACT for |x| >= 2^-110 (below that the lowest bits of x sit under the smallest bf16
    denormal, 2^-133, and are rounded away -- the kernel splits B the same way)."""
    x = x.to(torch.float32)
    x1 = bf16_round(x)
    r = x - x1
    x2 = bf16_round(r)
    x3 = bf16_round(r - x2)
    return x1, x2, x3


def dense_planes(vals: torch.Tensor) -> torch.Tensor:
    """The fp32-MFMA tile image ``vals`` ([ntiles, 128 * 128] in the order of partition.HostDense:
    vals[t][w][s4][kh * 32 + il][e] = A[32 w + il][2 (4 s4 + e) + kh]) as three bf16 planes in the A-operand order of
    v_mfma_f32_32x32x16_bf16: int16 [ntiles, w = 4, ks = 8, p = 3, lane = 64, j = 8] with
    lane = 32 * ((k >> 3) & 1) + (i & 31), ks = k >> 4, j = k & 7 for element A[i = 32 w + ..][k]."""
    nt = vals.shape[0]
    a = vals.view(nt, 4, 16, 2, 32, 4).permute(0, 1, 4, 2, 5, 3).reshape(nt, 4, 32, TC)      # [t][w][il][k]
    planes = torch.stack(bf16_split3(a), 0)                                                  # [p][t][w][il][k]
    bits = (planes.contiguous().view(torch.int32) >> 16).to(torch.int16)                     # bf16 bit patterns
    bits = bits.view(3, nt, 4, 32, 8, 2, 8)                                                  # [p][t][w][il][ks][hk][j]
    return bits.permute(1, 2, 4, 0, 5, 3, 6).reshape(nt, 4, 8, 3, 64, 8).contiguous()
