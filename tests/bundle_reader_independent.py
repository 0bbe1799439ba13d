"""A SECOND reader of the TensorFlow V2 checkpoint ("tensor bundle") format, written for the tests from the format's prose only
(LevelDB table_format.md: blocks of prefix-compressed entries + restart array + 5-byte trailer, 48-byte footer with two block handles
and the magic 0xdb4775248b80fb57; tensor_bundle.proto: BundleHeaderProto / BundleEntryProto; masked CRC-32C as in TFRecords).
It shares NO code with dpig_amd/tfckpt.py or dpig_amd/tfrecord.py -- own varint, own CRC table, own protobuf wire walk -- so that
tests/test_tfckpt.py checks the product's WRITER against something that is not the product's reader.  Still not a
TensorFlow-written file: none exists in this environment (INTEGRATION.md says so)."""
import struct

import numpy as np

MAGIC = 0xdb4775248b80fb57
DTYPES = {1: "<f4", 2: "<f8", 3: "<i4", 4: "u1", 6: "i1", 9: "<i8", 10: "?", 14: "bfloat16", 19: "<f2"}   # tensorflow/core/framework/types.proto

_TABLE = []
for _i in range(256):                       # CRC-32C (Castagnoli), reflected polynomial 0x82F63B78
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ (0x82F63B78 if (_c & 1) else 0)
    _TABLE.append(_c)


def crc32c(data):
    c = 0xFFFFFFFF
    for b in data:
        c = _TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked(data):                           # leveldb / TF mask: rotate right by 15, add a constant
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def varint(buf, pos):
    shift = val = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if b < 0x80:
            return val, pos
        shift += 7


def block(buf, off, size):
    """Contents of the block at (off, size), its trailer checked; compression type 0 only (what the product writes)."""
    body, ctype = buf[off:off + size], buf[off + size]
    (crc,) = struct.unpack_from("<I", buf, off + size + 1)
    if crc != masked(buf[off:off + size + 1]):
        raise ValueError("block checksum mismatch at %d" % off)
    if ctype != 0:
        raise ValueError("compressed block (type %d): not handled by this reader" % ctype)
    return body


def entries(body):
    """(key, value) pairs of one block in file order: shared-prefix varint, unshared varint, value-length varint, key tail, value."""
    (nrestart,) = struct.unpack_from("<I", body, len(body) - 4)
    end = len(body) - 4 - 4 * nrestart
    out, pos, prev = [], 0, b""
    while pos < end:
        shared, pos = varint(body, pos)
        unshared, pos = varint(body, pos)
        vlen, pos = varint(body, pos)
        key = prev[:shared] + bytes(body[pos:pos + unshared])
        pos += unshared
        out.append((key, bytes(body[pos:pos + vlen])))
        pos += vlen
        prev = key
    return out


def fields(msg):
    """Protobuf wire walk: [(field number, wire type, value)]; value = int (varint / fixed) or bytes (length-delimited)."""
    out, pos = [], 0
    while pos < len(msg):
        tag, pos = varint(msg, pos)
        num, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = varint(msg, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", msg, pos)[0]; pos += 8
        elif wt == 2:
            n, pos = varint(msg, pos)
            v = bytes(msg[pos:pos + n]); pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", msg, pos)[0]; pos += 4
        else:
            raise ValueError("wire type %d" % wt)
        out.append((num, wt, v))
    return out


def read_index(path):
    """{key: value bytes} of the .index table, every block of it verified."""
    buf = open(path, "rb").read()
    foot = buf[-48:]
    if struct.unpack("<Q", foot[40:])[0] != MAGIC:
        raise ValueError("not a table: bad magic")
    pos = 0
    _meta_off, pos = varint(foot, pos)
    _meta_size, pos = varint(foot, pos)
    idx_off, pos = varint(foot, pos)
    idx_size, pos = varint(foot, pos)
    out = {}
    for _last_key, handle in entries(block(buf, idx_off, idx_size)):
        off, p = varint(handle, 0)
        size, p = varint(handle, p)
        for k, v in entries(block(buf, off, size)):
            out[k] = v
    return out


def read_bundle(prefix):
    """{variable name: numpy array} of the checkpoint `prefix` (+ '' -> header fields), tensor checksums verified."""
    index = read_index(prefix + ".index")
    header = {n: v for n, _, v in fields(index[b""])}
    nshards = header.get(1, 0)
    out = {}
    for key, val in index.items():
        if key == b"":
            continue
        f = fields(val)
        one = {n: v for n, _, v in f}
        dims = []
        if 2 in one:
            for n, _, v in fields(one[2]):
                if n == 2:                                  # TensorShapeProto.Dim { size = 1 }
                    d = {a: b for a, _, b in fields(v)}
                    dims.append(d.get(1, 0))
        shard, off, size = one.get(3, 0), one.get(4, 0), one.get(5, 0)
        raw = open("%s.data-%05d-of-%05d" % (prefix, shard, nshards), "rb").read()[off:off + size]
        if len(raw) != size or masked(raw) != one.get(6):
            raise ValueError("tensor %r: size or checksum mismatch" % key)
        dt = DTYPES[one.get(1, 0)]
        if dt == "bfloat16":
            arr = (np.frombuffer(raw, dtype="<u2").astype(np.uint32) << 16).view(np.float32)
        else:
            arr = np.frombuffer(raw, dtype=dt)
        out[key.decode()] = arr.reshape(dims)
    return out, header
