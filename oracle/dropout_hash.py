"""TEST INFRASTRUCTURE -- not part of the product path.

numpy restatement of the stateless dropout stream the HIP kernels use
(mac-network_amd/csrc/macx_common.hip.h: hash_mix / site_key / keep_bit).

The reference draws its masks from TensorFlow's stateful RNG (ops.py:312, :674-679, :1054-1059;
mac_cell.py:217, :463): `floor(keep + U[0,1))`.  That stream cannot be reproduced outside TF, so
parity under dropout is defined on IDENTICAL MASKS: the oracle consumes masks produced here, the
product regenerates the same bits in-kernel, and tests/test_gpu_units.py::test_dropout_stream_matches_numpy checks the two
implementations bit-for-bit.
"""
import numpy as np

SITE_MEM_VAR = 1     # ops.py:1054  variational memory mask, once per batch (step 0)
SITE_MEM = 2         # mac_cell.py:217
SITE_READ_KB = 3     # ops.py:678
SITE_READ_MEM = 4    # ops.py:679
SITE_READ_ATT = 5    # ops.py:312 via :142
SITE_WRITE_INFO = 6  # mac_cell.py:463

_M32 = np.uint64(0xFFFFFFFF)


def hash_mix(h):
    h = np.asarray(h, dtype=np.uint64) & _M32
    h = (h * np.uint64(0x9E3779B1)) & _M32
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x85EBCA77)) & _M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE3D)) & _M32
    h ^= h >> np.uint64(16)
    return h


def site_key(seed, site, step):
    a = hash_mix(np.uint64(seed) ^ np.uint64(0xA511E9B3))
    inner = (np.uint64(site) * np.uint64(0x632BE5AB) + np.uint64(step) * np.uint64(0x2545F491) + np.uint64(0x1B873593)) & _M32
    return int(hash_mix(a ^ hash_mix(inner)))


def threshold24(keep):
    if keep >= 1.0:
        return 1 << 24
    return int(np.floor(np.float64(np.float32(keep)) * 16777216.0))


def keep_mask(seed, site, step, keep, first, n, word=0):
    """0/1 float32 mask for flat element indices first .. first+n-1 of a dropout site.  word: the run's mask word
    (include/macx.h, macx_dropout.mask_word -- XORed into the site key; 0 = the plain seed's masks)."""
    idx = (np.arange(n, dtype=np.uint64) + np.uint64(first)) & _M32
    key = np.uint64((site_key(seed, site, step) ^ (int(word) & 0xFFFFFFFF)) & 0xFFFFFFFF)
    h = hash_mix((idx >> np.uint64(1)) ^ key)      # one hash per pair of elements
    bits = np.where((idx & np.uint64(1)) == 1, h >> np.uint64(16), h & np.uint64(0xFFFF))
    return ((bits << np.uint64(8)) < np.uint64(threshold24(keep))).astype(np.float32)


def mask_for(seed, site, step, keep, shape, b0=0, word=0):
    """Mask of a [B, ...] tensor whose flat index starts at global question b0."""
    shape = tuple(int(x) for x in shape)
    per_q = int(np.prod(shape[1:])) if len(shape) > 1 else 1
    n = int(np.prod(shape))
    return keep_mask(seed, site, step, keep, b0 * per_q, n, word).reshape(shape)
