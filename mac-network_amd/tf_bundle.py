"""Reader for TensorFlow "V2" checkpoints (`<prefix>.index` + `<prefix>.data-0000N-of-0000M`), the format
`tf.train.Saver` writes in the reference (main.py:236-262, model.py:615-669 variables) -- without TensorFlow.

Host-side plumbing for SURVEY 8f row 4 ("TF checkpoint import by the names in 8b"): the result is the {variable name:
array} dictionary `macx.checkpoint.load_reference` already takes.

Format (tensorflow/core/util/tensor_bundle + the LevelDB table format of tensorflow/core/lib/io/table*):
  * `.index` is an SSTable: data blocks of prefix-compressed (key, value) entries with a restart array, an index block
    whose values are BlockHandles (varint64 offset, size) of the data blocks, and a 48-byte footer (metaindex handle, index
    handle, padding, magic 0xdb4775248b80fb57).  Every block is followed by a 1-byte compression type and a 4-byte masked
    crc32c.  The bundle writer sets kNoCompression; a snappy block is reported, not guessed at.
  * key "" -> BundleHeaderProto {1: num_shards, 2: endianness, 3: version}; every other key is a tensor name ->
    BundleEntryProto {1: dtype, 2: shape {2: dim {1: size}}, 3: shard_id, 4: offset, 5: size, 6: crc32c (fixed32), 7: slices}.
  * tensor bytes sit at [offset, offset + size) of data shard `shard_id`, little endian, row-major.
No TensorFlow exists in this environment, so the reader is checked against an in-repo writer that follows the same
specification (tests/test_tf_bundle.py) -- not against files written by TensorFlow itself; crc32c of blocks and tensors
is verified, which a real file has to satisfy too.
"""
import os
import struct

import numpy as np

MAGIC = 0xDB4775248B80FB57
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
          17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}

_CRC_TABLE = None


def _table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tab = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            tab.append(c)
        _CRC_TABLE = np.array(tab, dtype=np.uint32)
    return _CRC_TABLE


def _apply(cols, v):
    """GF(2) matrix (32 uint32 columns) times each element of the uint32 vector v."""
    out = np.zeros_like(v)
    for j in range(32):
        out ^= np.where((v >> np.uint32(j)) & np.uint32(1), cols[j], np.uint32(0)).astype(np.uint32)
    return out


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli), the checksum of LevelDB blocks and bundle tensors.  Large buffers are cut into 2^k equal
    lanes that advance one byte per numpy step; the lane registers are then merged pairwise with the "advance through n
    zero bytes" operator (the CRC register is linear over GF(2)), so a 60 MB tensor takes about a second."""
    tab = _table()
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    n = buf.size
    state = (crc ^ 0xFFFFFFFF) & 0xFFFFFFFF
    lanes = 1
    while lanes < (1 << 16) and lanes * 2 * 256 <= n:
        lanes *= 2
    done = 0
    if lanes > 1:
        L = n // lanes
        done = L * lanes
        m = buf[:done].reshape(lanes, L)
        s = np.zeros(lanes, dtype=np.uint32)
        s[0] = state
        z = np.uint32(1) << np.arange(32, dtype=np.uint32)          # columns of the zero-byte advance operator
        for i in range(L):
            s = tab[(s ^ m[:, i]) & np.uint32(0xFF)] ^ (s >> np.uint32(8))
            z = tab[z & np.uint32(0xFF)] ^ (z >> np.uint32(8))
        while s.size > 1:
            s = _apply(z, s[0::2]) ^ s[1::2]
            z = _apply(z, z)
        state = int(s[0])
    tl = tab.tolist()
    for b in buf[done:].tolist():
        state = tl[(state ^ b) & 0xFF] ^ (state >> 8)
    return state ^ 0xFFFFFFFF


def mask_crc(c):
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(buf, pos):
    out, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _read_block(f, offset, size, verify):
    f.seek(offset)
    raw = f.read(size + 5)
    if len(raw) != size + 5:
        raise ValueError("truncated table block at %d" % offset)
    body, ctype, crc = raw[:size], raw[size], struct.unpack("<I", raw[size + 1:])[0]
    if verify and mask_crc(crc32c(raw[:size + 1])) != crc:
        raise ValueError("crc mismatch in table block at %d" % offset)
    if ctype != 0:
        raise NotImplementedError("table block at %d is compressed (type %d); the bundle writer does not compress" % (offset, ctype))
    return body


def _block_entries(block):
    n_restarts = struct.unpack("<I", block[-4:])[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def _proto_fields(buf):
    """(field number, wire type, value) of one protobuf message; nested messages stay bytes."""
    pos, n = 0, len(buf)
    while pos < n:
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack("<Q", buf[pos:pos + 8])[0]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = struct.unpack("<I", buf[pos:pos + 4])[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield field, wt, v


def _parse_entry(buf):
    e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=None, sliced=False)
    for field, _, v in _proto_fields(buf):
        if field == 1:
            e["dtype"] = v
        elif field == 2:
            for f2, _, v2 in _proto_fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, v3 in _proto_fields(v2):
                        if f3 == 1:
                            size = v3 - (1 << 64) if v3 >> 63 else v3
                    e["shape"].append(size)
        elif field == 3:
            e["shard_id"] = v
        elif field == 4:
            e["offset"] = v
        elif field == 5:
            e["size"] = v
        elif field == 6:
            e["crc32c"] = v
        elif field == 7:
            e["sliced"] = True
    return e


def read_index(prefix, verify=True):
    """{tensor name: entry dict}, header dict of `<prefix>.index`."""
    path = prefix + ".index"
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        f.seek(size - 48)
        footer = f.read(48)
        if struct.unpack("<Q", footer[40:])[0] != MAGIC:
            raise ValueError("%s is not a TensorFlow checkpoint index (bad table magic)" % path)
        pos = 0
        _, pos = _varint(footer, pos)          # metaindex handle
        _, pos = _varint(footer, pos)
        ioff, pos = _varint(footer, pos)
        isz, pos = _varint(footer, pos)
        entries, header = {}, None
        for _, handle in _block_entries(_read_block(f, ioff, isz, verify)):
            boff, p2 = _varint(handle, 0)
            bsz, _ = _varint(handle, p2)
            for key, value in _block_entries(_read_block(f, boff, bsz, verify)):
                if key == b"":
                    header = {fld: v for fld, _, v in _proto_fields(value)}
                else:
                    entries[key.decode()] = _parse_entry(value)
    if header is None:
        raise ValueError("%s has no bundle header" % path)
    if header.get(2, 0) != 0:
        raise NotImplementedError("big-endian bundle")
    return entries, dict(num_shards=header.get(1, 1), version=header.get(3))


def read_checkpoint(prefix, names=None, verify=True):
    """{variable name: numpy array} of a V2 checkpoint; `names` restricts the tensors read."""
    entries, header = read_index(prefix, verify)
    out, files = {}, {}
    try:
        for name, e in entries.items():
            if names is not None and name not in names:
                continue
            if e["sliced"]:
                raise NotImplementedError("%s is stored as slices (partitioned variable)" % name)
            if e["dtype"] not in DTYPES:
                raise NotImplementedError("%s: dtype enum %d" % (name, e["dtype"]))
            sid = e["shard_id"]
            if sid not in files:
                files[sid] = open("%s.data-%05d-of-%05d" % (prefix, sid, header["num_shards"]), "rb")
            f = files[sid]
            f.seek(e["offset"])
            raw = f.read(e["size"])
            if len(raw) != e["size"]:
                raise ValueError("%s: data shard truncated" % name)
            if verify and e["crc32c"] is not None and mask_crc(crc32c(raw)) != e["crc32c"]:
                raise ValueError("%s: tensor crc mismatch" % name)
            arr = np.frombuffer(raw, dtype=np.dtype(DTYPES[e["dtype"]]).newbyteorder("<"))
            out[name] = arr.reshape(e["shape"]).copy()
    finally:
        for f in files.values():
            f.close()
    return out


# ---------------------------------------------------------------------------------------------------------------------
# writer: the same format, so weights trained here restore in the reference (`saver.restore`, main.py:185-193)

def _put_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field(num, wt, payload):
    tag = _put_varint((num << 3) | wt)
    if wt == 0:
        return tag + _put_varint(payload)
    if wt == 2:
        return tag + _put_varint(len(payload)) + payload
    if wt == 5:
        return tag + struct.pack("<I", payload)
    raise ValueError(wt)


class _BlockBuilder:
    def __init__(self, restart_interval=16):
        self.buf, self.restarts, self.count, self.last, self.interval = bytearray(), [0], 0, b"", restart_interval

    def add(self, key, value):
        shared = 0
        if self.count and self.count % self.interval == 0:
            self.restarts.append(len(self.buf))
        elif self.count:
            lim = min(len(key), len(self.last))
            while shared < lim and key[shared] == self.last[shared]:
                shared += 1
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
        self.last, self.count = key, self.count + 1

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def _emit_block(f, body):
    off = f.tell()
    f.write(body + b"\x00" + struct.pack("<I", mask_crc(crc32c(body + b"\x00"))))
    return _put_varint(off) + _put_varint(len(body))


_ENUM = {np.dtype(v): k for k, v in DTYPES.items()}


def write_checkpoint(prefix, tensors, block_size=4096):
    """Write {variable name: array} as a one-shard V2 checkpoint.  Returns the sorted names."""
    names = sorted(tensors, key=lambda s: s.encode())
    data_path = "%s.data-00000-of-00001" % prefix
    items = [(b"", _field(1, 0, 1) + _field(3, 2, _field(1, 0, 1)))]                 # header: 1 shard, version.producer = 1
    with open(data_path, "wb") as f:
        for name in names:
            a = np.asarray(tensors[name], order="C")
            if a.dtype not in _ENUM:
                raise NotImplementedError("%s: dtype %s" % (name, a.dtype))
            raw = a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes()
            shape = b"".join(_field(2, 2, _field(1, 0, int(s))) for s in a.shape)
            entry = _field(1, 0, _ENUM[a.dtype]) + _field(2, 2, shape)
            if f.tell():
                entry += _field(4, 0, f.tell())
            entry += _field(5, 0, len(raw)) + _field(6, 5, mask_crc(crc32c(raw)))
            items.append((name.encode(), entry))
            f.write(raw)
    with open(prefix + ".index", "wb") as f:
        index, blk = _BlockBuilder(restart_interval=1), _BlockBuilder()
        for key, value in items:
            blk.add(key, value)
            if len(blk.buf) >= block_size:
                index.add(blk.last, _emit_block(f, blk.finish()))
                blk = _BlockBuilder()
        if blk.count:
            index.add(blk.last, _emit_block(f, blk.finish()))
        meta = _emit_block(f, _BlockBuilder().finish())
        idx = _emit_block(f, index.finish())
        foot = meta + idx
        f.write(foot + bytes(40 - len(foot)) + struct.pack("<Q", MAGIC))
    return names
