"""Host replica of the kernels' counter-based dropout rule (csrc/common.h::dropout_hash), used to derive per-site keys
and, in tests, to rebuild the exact masks a training step used."""
import torch

M32 = 0xFFFFFFFF


def lowbias32(x: int) -> int:
    x &= M32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & M32
    x ^= x >> 16
    return x


def site_key(seed: int, site: int) -> int:
    """32-bit key of dropout site `site` for the step seed `seed`."""
    return lowbias32((seed ^ (site * 0x9E3779B1)) & M32) ^ lowbias32((seed >> 32) & M32)


def threshold(p: float) -> int:
    return min(M32, int(round(p * 4294967296.0)))


def keep_mask(key: int, n: int, thr: int, offset: int = 0) -> torch.Tensor:
    """bool[n]: element i kept iff hash(key, offset + i) >= thr (vectorised replica of dropout_keep)."""
    x = (torch.arange(offset, offset + n, dtype=torch.int64) ^ key) & M32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & M32
    x ^= x >> 16
    return x >= thr
