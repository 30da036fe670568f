"""macx.h5 (HDF5 reading without libhdf5; SURVEY 8f row 4 "h5 features") against files written by the real HDF5 library:
tests/golden/h5/*.h5 come from h5py 3.3.0 / HDF5 1.10.6 (tests/golden/make_h5_fixtures.py), the first one with exactly
the calls of the reference's extractor (extract_features.py:84-110)."""
import os

import numpy as np
import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "h5")
N, C, H, W = 7, 8, 3, 3


def expected():
    a = np.arange(N * C * H * W, dtype=np.float64).reshape(N, C, H, W)
    return (np.sin(a * 0.37) * 3.0 + a * 1e-3).astype(np.float32)


@pytest.fixture()
def h5():
    import macx
    return macx.h5


@pytest.mark.parametrize("name", ["features_like_reference", "features_chunked_gzip", "chunked_plain_and_groups", "features_latest"])
def test_features_dataset_as_the_reference_reads_it(h5, name):
    x = expected()
    f = h5.File(os.path.join(HERE, name + ".h5"), "r")                  # main.py:314
    d = f["features"]
    assert d.shape == (N, C, H, W) and d.dtype == np.float32 and len(d) == N and d.ndim == 4
    for i in (0, 3, N - 1):
        row = d[i]                                                        # main.py:332
        assert row.shape == (C, H, W) and np.array_equal(row, x[i])
    assert np.array_equal(d[2:5], x[2:5]) and np.array_equal(d[[5, 1, 1]], x[[5, 1, 1]]) and np.array_equal(d[...], x)
    assert np.array_equal(d[:, 1, 2], x[:, 1, 2]) and np.array_equal(np.asarray(d), x)
    ids = [4, 0, 6, 6]
    assert np.array_equal(h5.load_image_batch(f, ids), np.stack([x[i] for i in ids], axis=0))      # main.py:325-334
    id2idx = {"a": 2, "b": 5}                                             # NLVR-style id table, main.py:328-331
    out = np.empty((2, C, H, W), np.float32)
    assert h5.load_image_batch(f, ["b", "a"], id2idx=id2idx, out=out) is out and np.array_equal(out, x[[5, 2]])
    f.close()                                                             # main.py:322


def test_contiguous_features_are_memory_mapped(h5):
    """What the extractor writes is one contiguous block: indexing must not read the whole dataset."""
    with h5.File(os.path.join(HERE, "features_like_reference.h5")) as f:
        assert isinstance(f["features"]._array(), np.memmap)
    with h5.File(os.path.join(HERE, "features_latest.h5")) as f:
        assert isinstance(f["features"]._array(), np.memmap)


def test_groups_integers_byte_order_scalars_and_unwritten_datasets(h5):
    big = np.arange(200 * 6, dtype=np.int64).reshape(200, 6) * 7 - 300
    with h5.File(os.path.join(HERE, "chunked_plain_and_groups.h5")) as f:
        assert sorted(f.keys()) == ["features", "meta", "never_written"] and "meta" in f and "nope" not in f
        assert sorted(f["meta"].keys()) == ["be", "inner"]
        ids = f["meta/inner/ids"]                                         # 200 chunks: a multi-level chunk B-tree
        assert ids.dtype == np.int32 and np.array_equal(ids[...], big.astype(np.int32))
        assert np.array_equal(f["meta"]["inner"]["ids"][17], big[17].astype(np.int32))
        sc = f["meta/inner/scalar"]
        assert sc.shape == () and float(sc[()]) == 2.5
        be = f["/meta/be"]
        assert be.dtype == np.dtype(">i2") and np.array_equal(be[...], big[:5].astype(np.int16))
        nw = f["never_written"]
        assert nw.shape == (4, 2) and not nw[...].any()                   # fill value
        with pytest.raises(KeyError):
            f["meta/none"]
    with h5.File(os.path.join(HERE, "features_latest.h5")) as f:
        assert np.array_equal(f["g/small"][...], np.arange(5, dtype=np.uint8))
    with h5.File(os.path.join(HERE, "many_links.h5")) as f:             # several symbol-table nodes under the group B-tree
        assert f.keys() == ["d%02d" % i for i in range(40)] or sorted(f.keys()) == ["d%02d" % i for i in range(40)]
        for i in (0, 17, 39):
            assert np.array_equal(f["d%02d" % i][...], np.full(3, i, np.float32))


def test_rejects_what_is_not_hdf5(h5, tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"not an hdf5 file" * 100)
    with pytest.raises(ValueError, match="not an HDF5"):
        h5.File(str(p))
    with pytest.raises(ValueError, match="read-only"):
        h5.File(os.path.join(HERE, "features_latest.h5"), "w")
    raw = open(os.path.join(HERE, "features_like_reference.h5"), "rb").read()
    q = tmp_path / "cut.h5"
    q.write_bytes(raw[:600])
    with pytest.raises(ValueError):
        h5.File(str(q))["features"][6]


def test_feeds_the_stem_layout(h5):
    """[B,C,H,W] rows from the file are what macx_images_to_nhwc / MACNet take (model.py:67-68)."""
    import torch
    with h5.File(os.path.join(HERE, "features_like_reference.h5")) as f:
        batch = torch.from_numpy(h5.load_image_batch(f, [1, 2]))
    assert batch.shape == (2, C, H, W) and batch.dtype == torch.float32 and batch.is_contiguous()


def test_other_types_filters_user_block_and_named_refusals(h5):
    x = expected()
    with h5.File(os.path.join(HERE, "types_and_filters.h5")) as f:          # 512-byte user block in front of the superblock
        assert sorted(f.keys()) == ["checked", "f4_be", "f8", "gz_only", "u2"]
        assert np.array_equal(f["checked"][...], x)                         # fletcher32-checked chunks
        assert np.array_equal(f["gz_only"][...], x)                         # deflate without shuffle
        assert f["f8"].dtype == np.float64 and np.array_equal(f["f8"][...], x.astype(np.float64)[:2])
        assert f["f4_be"].dtype == np.dtype(">f4") and np.array_equal(f["f4_be"][...], x[:2])
        assert np.array_equal(f["u2"][...], np.arange(12, dtype=np.uint16).reshape(3, 4))
    with h5.File(os.path.join(HERE, "refused.h5")) as f:
        assert np.array_equal(f["ok"][...], np.arange(4, dtype=np.float32))
        with pytest.raises(NotImplementedError, match="layout"):            # version-4 chunk index (libver latest)
            f["chunked_v4"]
        with pytest.raises(NotImplementedError, match="datatype class 6"):  # compound type
            f["compound"]
