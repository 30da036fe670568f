"""Writes tests/golden/h5/*.h5 with the real HDF5 library (h5py) -- run with an interpreter that has h5py:

    /opt/conda/bin/python3.9 tests/golden/make_h5_fixtures.py          # h5py 3.3.0 / HDF5 1.10.6 in this image

`features_like_reference.h5` repeats the calls of the reference's extractor (extract_features.py:84-110: create_dataset
('features', (N,C,H,W), dtype=float32), then batch-sized slice assignments); the others exercise the remaining structures
macx.h5 parses.  Contents are formulas of the index (see expected() in tests/test_h5.py), so no second copy is stored."""
import os
import sys

import h5py
import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "h5")


def feats(N, C, H, W):
    a = np.arange(N * C * H * W, dtype=np.float64).reshape(N, C, H, W)
    return (np.sin(a * 0.37) * 3.0 + a * 1e-3).astype(np.float32)


def main():
    os.makedirs(HERE, exist_ok=True)
    N, C, H, W = 7, 8, 3, 3
    x = feats(N, C, H, W)
    # 1. what extract_features.py writes: contiguous float32, filled batch by batch (batch_size 3 -> 3 + 3 + 1)
    with h5py.File(os.path.join(HERE, "features_like_reference.h5"), "w") as f:
        d = f.create_dataset("features", (N, C, H, W), dtype=np.float32)
        for i0 in range(0, N, 3):
            d[i0:i0 + 3] = x[i0:i0 + 3]
    # 2. chunked + gzip + shuffle (how feature files are often re-packed), chunks that do not divide the shape
    with h5py.File(os.path.join(HERE, "features_chunked_gzip.h5"), "w") as f:
        f.create_dataset("features", data=x, chunks=(2, 8, 2, 3), compression="gzip", shuffle=True)
    # 3. chunked, no filter, many chunks (multi-level chunk B-tree), plus an integer dataset in a nested group
    big = (np.arange(200 * 6, dtype=np.int64).reshape(200, 6) * 7 - 300)
    with h5py.File(os.path.join(HERE, "chunked_plain_and_groups.h5"), "w") as f:
        f.create_dataset("features", data=x, chunks=(1, 8, 3, 3))
        g = f.create_group("meta").create_group("inner")
        g.create_dataset("ids", data=big.astype(np.int32), chunks=(1, 6))
        g.create_dataset("scalar", data=np.float64(2.5))
        f["meta"].create_dataset("be", data=big[:5].astype(">i2"))
        f.create_dataset("never_written", (4, 2), dtype=np.float32)
    # 4. libver="latest": superblock v3, version-2 object headers, compact link messages, layout message v4
    with h5py.File(os.path.join(HERE, "features_latest.h5"), "w", libver="latest") as f:
        d = f.create_dataset("features", (N, C, H, W), dtype=np.float32)
        d[...] = x
        f.create_group("g").create_dataset("small", data=np.arange(5, dtype=np.uint8))
    # 5. many links in one old-style group (several SNOD nodes under the group B-tree)
    with h5py.File(os.path.join(HERE, "many_links.h5"), "w") as f:
        for i in range(40):
            f.create_dataset("d%02d" % i, data=np.full((3,), i, dtype=np.float32))
    # 6. other element types / filters, attributes (ignored by the reader), a user block in front of the superblock
    with h5py.File(os.path.join(HERE, "types_and_filters.h5"), "w", userblock_size=512) as f:
        f.create_dataset("f8", data=x.astype(np.float64)[:2])
        f.create_dataset("f4_be", data=x.astype(">f4")[:2])
        f.create_dataset("checked", data=x, chunks=(3, 8, 3, 3), fletcher32=True)
        f.create_dataset("gz_only", data=x, chunks=(7, 4, 3, 3), compression="gzip", compression_opts=9)
        f.create_dataset("u2", data=np.arange(12, dtype=np.uint16).reshape(3, 4))
        f["f8"].attrs["note"] = "attributes are skipped"
        f.attrs["version"] = 3
    # 7. structures the reader refuses by name: a version-4 (libver latest) chunk index, a compound type
    with h5py.File(os.path.join(HERE, "refused.h5"), "w", libver="latest") as f:
        f.create_dataset("chunked_v4", data=x, chunks=(2, 8, 3, 3))
        f.create_dataset("compound", data=np.zeros(3, dtype=[("a", "<f4"), ("b", "<i4")]))
        f.create_dataset("ok", data=np.arange(4, dtype=np.float32))
    print("h5py", h5py.__version__, "hdf5", h5py.version.hdf5_version, "->", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    sys.exit(main())
