"""Host replica of the kernels' counter-based dropout rule (csrc/common.h::dropout_hash: four 8-bit uniforms per lowbias32
hash), used to derive per-site keys and, in tests, to rebuild the exact masks a training step used."""
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
    """8-bit threshold thr8 = round(p * 256): an element is dropped iff its 8-bit uniform < thr8 (p_eff = thr8 / 256)."""
    return max(0, min(256, int(round(p * 256.0))))


def scale(thr8: int) -> float:
    """1 / (1 - p_eff) for the probability the kernels actually realise, p_eff = thr8 / 256."""
    return 256.0 / max(1, 256 - thr8)


def keep_mask(key: int, n: int, thr: int, offset: int = 0) -> torch.Tensor:
    """bool[n]: element i kept iff byte (i & 3) of hash(key, (offset + i) >> 2) >= thr (replica of csrc/common.h::dropout_keep)."""
    idx = torch.arange(offset, offset + n, dtype=torch.int64)
    x = ((idx >> 2) ^ key) & M32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & M32
    x ^= x >> 16
    return ((x >> (8 * (idx & 3))) & 0xFF) >= thr


def sample_uniform(seed: int, step: int, row: int) -> float:
    """Host replica of the sampler's uniform draw (csrc/sample.hip): u in [0, 1) with 24 bits, a pure function of the 64-bit
    ``seed``, the position being written (``step`` = current length of the id row) and the caption's ``row`` in the batch."""
    lo, hi = seed & M32, (seed >> 32) & M32
    h = lowbias32((lowbias32(lo ^ ((row * 0x9E3779B9) & M32)) + hi + step * 0x85EBCA6B) & M32)
    return (h >> 8) / 16777216.0


def mlm_draws(seed: int, n: int):
    """Host replica of the three per-token draws of csrc/elementwise.hip::lm_inputs_kernel for elements 0 .. n-1 of the label
    tensor: (u_mask float32 in [0, 1), u_rand float32 in [0, 1), random id numerator uint32 -- id = (numerator * vocab) >> 32)."""
    lo, hi = seed & M32, (seed >> 32) & M32
    src = torch.arange(n, dtype=torch.int64)

    def mix(x):
        x = x & M32
        x ^= x >> 16
        x = (x * 0x7FEB352D) & M32
        x ^= x >> 15
        x = (x * 0x846CA68B) & M32
        x ^= x >> 16
        return x
    h1 = mix((mix((src & M32) ^ lo) + hi + ((src >> 32) * 0x9E3779B9)) & M32)
    h2 = mix(h1 ^ 0x85EBCA6B)
    num = mix((h2 + 0x27D4EB2F) & M32)
    return (h1 >> 8).to(torch.float32) / 16777216.0, (h2 >> 8).to(torch.float32) / 16777216.0, num
