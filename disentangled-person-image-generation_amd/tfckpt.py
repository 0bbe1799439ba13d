"""TensorFlow-free reader / writer of TF "V2" checkpoints.

The reference saves and restores with `tf.train.Saver` (`trainer.py:180-213`, `tester.py:17-64`): partial restores
by variable scope ('Encoder' + 'ID_AE' from `pretrained_path`, 'PoseAE' from `pretrained_poseAE_path`), a full
restore from `ckpt_path`.  A V2 checkpoint `<prefix>` is two kinds of file:

  <prefix>.index                    an immutable sorted string table (the LevelDB table format TensorFlow carries
                                    in core/lib/io/table*): data blocks of prefix-compressed (key, value) entries
                                    with a restart array, each followed by a 5-byte trailer (compression type,
                                    masked CRC32C of block + type); an index block mapping separator keys to block
                                    handles; a 48-byte footer (metaindex handle, index handle, padding, magic
                                    0xdb4775248b80fb57).  Key "" holds a BundleHeaderProto {num_shards, endianness,
                                    version}; every other key is a variable name holding a BundleEntryProto
                                    {dtype, shape, shard_id, offset, size, masked crc32c of the bytes, slices}.
  <prefix>.data-SSSSS-of-NNNNN      the tensors' raw little-endian bytes back to back.

This module restates those public formats (tensor_bundle.proto, tensor_shape.proto, types.proto, the LevelDB
table_format document) in plain python / numpy.  UNPINNED: no TensorFlow-written checkpoint exists in this
environment, so the reader has only met files from the writer below and the hand-assembled blocks of
tests/test_tfckpt.py; the constants that can be checked independently (magic, CRC32C vectors, proto field numbers,
snappy framing) are asserted there.

`restore(prefix, scopes)` / `save(prefix)` move values between a checkpoint and the `tflib` parameter registry,
whose names are the reference's variable names (`Encoder/G_encoder/Conv/weights`, `Discriminator.1.Filters`, ...;
SURVEY Appendix F), filters HWIO and FC weights [in, out] exactly as TF stores them.
`restore` / `save` move VARIABLES; entries of a checkpoint that the registry does not know are ignored.  Optimizer slots
(`<var>/Adam`, `<var>/Adam_1`, `beta1_power`, ...) are the trainer's business: `trainer.optimizer_slots` /
`load_optimizer_slots`, switched on by `save_checkpoint(include_optimizer=True)` and `Config.restore_optimizer`.
"""
import os
import struct

import numpy as np

from .tfrecord import _enc_varint, _fields, _ld, _varint, masked_crc32c

TABLE_MAGIC = 0xdb4775248b80fb57
FOOTER_LEN = 48
BLOCK_TRAILER_LEN = 5
RESTART_INTERVAL = 16
BLOCK_SIZE = 262144

# types.proto DataType <-> numpy (the fixed-width types a Saver writes for this model family)
DT_TO_NP = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64,
            10: np.bool_, 17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
NP_TO_DT = {np.dtype(v): k for k, v in DT_TO_NP.items()}


# ---- snappy (block format), decoder only: TensorFlow's table reader accepts snappy blocks --------------------------
def snappy_decompress(buf):
    buf = bytes(buf)
    n, pos = _varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                        # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 2], "little")
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise IOError("snappy: bad copy offset")
        for _ in range(ln):                                  # copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise IOError("snappy: length mismatch")
    return bytes(out)


# ---- the table ---------------------------------------------------------------------------------------------------
def _read_block(buf, offset, size, verify=True):
    end = offset + size
    if end + BLOCK_TRAILER_LEN > len(buf):
        raise IOError("table block runs past the end of the file")
    ctype = buf[end]
    if verify and masked_crc32c(buf[offset:end + 1]) != struct.unpack("<I", buf[end + 1:end + 5])[0]:
        raise IOError("table block CRC mismatch")
    raw = bytes(buf[offset:end])
    if ctype == 1:
        raw = snappy_decompress(raw)
    elif ctype != 0:
        raise IOError("unknown table block compression %d" % ctype)
    return raw


def block_entries(block):
    """(key, value) pairs of one decoded block."""
    n = len(block)
    if n < 4:
        raise IOError("table block too short")
    num_restarts = struct.unpack("<I", block[n - 4:])[0]
    limit = n - 4 - 4 * num_restarts
    if limit < 0:
        raise IOError("bad restart array")
    pos, key = 0, b""
    while pos < limit:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        if shared > len(key) or pos + non_shared + vlen > limit:
            raise IOError("corrupt table entry")
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def _handle(buf, pos=0):
    off, pos = _varint(buf, pos)
    size, pos = _varint(buf, pos)
    return off, size, pos


def read_table(path, verify=True):
    """All (key, value) pairs of a table file, in key order."""
    buf = open(path, "rb").read()
    if len(buf) < FOOTER_LEN:
        raise IOError("%s: too short for a table" % path)
    footer = buf[-FOOTER_LEN:]
    if struct.unpack("<Q", footer[40:])[0] != TABLE_MAGIC:
        raise IOError("%s: not a table file (bad magic number)" % path)
    _, _, pos = _handle(footer)                              # metaindex (filters): unused by the bundle
    ioff, isize, _ = _handle(footer, pos)
    out = []
    for _, handle in block_entries(_read_block(buf, ioff, isize, verify)):
        off, size, _ = _handle(handle)
        out.extend(block_entries(_read_block(buf, off, size, verify)))
    return out


class _BlockBuilder(object):
    def __init__(self):
        self.buf, self.restarts, self.count, self.last = bytearray(), [0], 0, b""

    def add(self, key, value):
        shared = 0
        if self.count % RESTART_INTERVAL == 0:
            if self.count:
                self.restarts.append(len(self.buf))
        else:
            lim = min(len(key), len(self.last))
            while shared < lim and key[shared] == self.last[shared]:
                shared += 1
        self.buf += _enc_varint(shared) + _enc_varint(len(key) - shared) + _enc_varint(len(value))
        self.buf += key[shared:] + value
        self.last = key
        self.count += 1

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def write_table(path, items, block_size=BLOCK_SIZE):
    """items: (key bytes, value bytes) pairs in strictly increasing key order.  Uncompressed blocks."""
    out = bytearray()

    def emit(block):
        off = len(out)
        out.extend(block)
        out.append(0)                                        # kNoCompression
        out.extend(struct.pack("<I", masked_crc32c(bytes(block) + b"\x00")))
        return _enc_varint(off) + _enc_varint(len(block))

    index, cur, prev = _BlockBuilder(), _BlockBuilder(), None
    for key, value in items:
        if prev is not None and key <= prev:
            raise ValueError("table keys must be strictly increasing")
        if cur.count and cur.size() >= block_size:
            index.add(cur.last, emit(cur.finish()))          # the block's last key is a valid separator
            cur = _BlockBuilder()
        cur.add(key, value)
        prev = key
    if cur.count:
        index.add(cur.last, emit(cur.finish()))
    meta = emit(_BlockBuilder().finish())
    idx = emit(index.finish())
    footer = meta + idx
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    out.extend(footer)
    with open(path, "wb") as f:
        f.write(bytes(out))


# ---- the tensor bundle ---------------------------------------------------------------------------------------------
def _parse_header(buf):
    h = {"num_shards": 0, "endianness": 0, "producer": 0}
    for num, wt, val in _fields(memoryview(buf)):
        if num == 1 and wt == 0:
            h["num_shards"] = val
        elif num == 2 and wt == 0:
            h["endianness"] = val
        elif num == 3 and wt == 2:
            for n2, w2, v2 in _fields(val):
                if n2 == 1 and w2 == 0:
                    h["producer"] = v2
    return h


def _parse_entry(buf):
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": 0, "slices": 0, "unknown_rank": False}
    for num, wt, val in _fields(memoryview(buf)):
        if num == 1 and wt == 0:
            e["dtype"] = val
        elif num == 2 and wt == 2:                           # TensorShapeProto
            for n2, w2, v2 in _fields(val):
                if n2 == 2 and w2 == 2:                      # Dim
                    size = 0
                    for n3, w3, v3 in _fields(v2):
                        if n3 == 1 and w3 == 0:
                            size = v3 - (1 << 64) if v3 >= (1 << 63) else v3
                    e["shape"].append(size)
                elif n2 == 3 and w2 == 0:
                    e["unknown_rank"] = bool(v2)
        elif num == 3 and wt == 0:
            e["shard_id"] = val
        elif num == 4 and wt == 0:
            e["offset"] = val
        elif num == 5 and wt == 0:
            e["size"] = val
        elif num == 6 and wt == 5:
            e["crc32c"] = struct.unpack("<I", val)[0]
        elif num == 7:
            e["slices"] += 1
    return e


def _shard_path(prefix, shard, num_shards):
    return "%s.data-%05d-of-%05d" % (prefix, shard, num_shards)


def _index(prefix, verify=True):
    items = read_table(prefix + ".index", verify)
    if not items or items[0][0] != b"":
        raise IOError("%s.index: no bundle header" % prefix)
    header = _parse_header(items[0][1])
    if header["endianness"] != 0:
        raise IOError("big-endian checkpoints are not supported")
    if header["num_shards"] < 1:
        raise IOError("bundle header names no shards")
    return header, [(k.decode("utf-8"), _parse_entry(v)) for k, v in items[1:]]


def list_variables(prefix):
    """[(name, shape tuple, numpy dtype or None)] like tf.train.list_variables."""
    _, entries = _index(prefix)
    return [(n, tuple(e["shape"]), DT_TO_NP.get(e["dtype"])) for n, e in entries]


def load_checkpoint(prefix, names=None, verify=True):
    """{variable name: numpy array}; `names` (an iterable or a predicate) selects a subset."""
    header, entries = _index(prefix, verify)
    want = names if callable(names) or names is None else set(names).__contains__
    shards, out = {}, {}
    for name, e in entries:
        if want is not None and not want(name):
            continue
        if e["slices"]:
            raise IOError("%s: partitioned variables are not supported" % name)
        if e["dtype"] not in DT_TO_NP:
            raise IOError("%s: unsupported dtype enum %d" % (name, e["dtype"]))
        dt = np.dtype(DT_TO_NP[e["dtype"]])
        count = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
        if e["size"] != count * dt.itemsize:
            raise IOError("%s: %d bytes recorded for shape %s of %s" % (name, e["size"], e["shape"], dt))
        sid = e["shard_id"]
        if sid not in shards:
            spath = _shard_path(prefix, sid, header["num_shards"])
            shards[sid] = (np.memmap(spath, dtype=np.uint8, mode="r") if os.path.getsize(spath) else np.zeros(0, np.uint8))
        data = shards[sid]
        if e["offset"] + e["size"] > data.size:
            raise IOError("%s: data shard is truncated" % name)
        raw = np.array(data[e["offset"]:e["offset"] + e["size"]])
        if verify and masked_crc32c(raw) != e["crc32c"]:
            raise IOError("%s: tensor CRC mismatch" % name)
        out[name] = raw.view(dt.newbyteorder("<")).astype(dt, copy=False).reshape(e["shape"])
    return out


def save_checkpoint(prefix, tensors, update_state_file=True):
    """Write {name: array} as a one-shard V2 checkpoint (+ the `checkpoint` state file tf.train.latest_checkpoint reads)."""
    d = os.path.dirname(prefix)
    if d:
        os.makedirs(d, exist_ok=True)
    items = [(b"", b"\x08\x01" + _ld(3, b"\x08\x01"))]       # num_shards = 1, little endian (default), version.producer = 1
    offset = 0
    with open(_shard_path(prefix, 0, 1), "wb") as f:
        for name in sorted(tensors, key=lambda s: s.encode("utf-8")):
            if not name:
                raise ValueError("empty variable name")
            a = np.asarray(tensors[name])                     # (ascontiguousarray would turn a scalar into shape [1])
            if not a.flags.c_contiguous:
                a = a.copy(order="C")
            if a.dtype not in NP_TO_DT:
                raise ValueError("%s: dtype %s has no checkpoint encoding here" % (name, a.dtype))
            raw = a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes()
            shape = b"".join(_ld(2, (b"\x08" + _enc_varint(s)) if s else b"") for s in a.shape)
            entry = b"\x08" + _enc_varint(NP_TO_DT[a.dtype]) + _ld(2, shape)
            if offset:
                entry += b"\x20" + _enc_varint(offset)
            if raw:
                entry += b"\x28" + _enc_varint(len(raw))
            entry += b"\x35" + struct.pack("<I", masked_crc32c(raw))
            items.append((name.encode("utf-8"), entry))
            f.write(raw)
            offset += len(raw)
    write_table(prefix + ".index", items)
    if update_state_file:
        base = os.path.basename(prefix)
        with open(os.path.join(d or ".", "checkpoint"), "w") as f:
            f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))


def latest_checkpoint(model_dir):
    """tf.train.latest_checkpoint: the prefix named by `<model_dir>/checkpoint`, or None."""
    path = os.path.join(model_dir, "checkpoint")
    if not os.path.exists(path):
        return None
    for line in open(path):
        if line.startswith("model_checkpoint_path:"):
            name = line.split(":", 1)[1].strip().strip('"')
            return name if os.path.isabs(name) else os.path.join(model_dir, name)
    return None


# ---- checkpoint <-> tflib registry ---------------------------------------------------------------------------------
def _in_scopes(name, scopes):
    """tf.get_collection(..., scope=s) keeps the names that re.match(s, name): a prefix test for plain scope names."""
    return scopes is None or any(name.startswith(s) for s in scopes)


def restore(prefix, scopes=None, strict=True, verify=True):
    """`tf.train.Saver(var_list).restore(sess, prefix)` for the variables of the tflib registry whose name lies in one
    of `scopes` (None = every variable, trainer.py:188,210-212; ['Encoder', 'ID_AE'] = `pretrained_path`, :180-183;
    ['PoseAE'] = `pretrained_poseAE_path`, :185-187).  Values are copied in place (flat parameter buffers, graphs and
    optimizers keep their addresses).  Like the Saver, a selected variable that the checkpoint lacks, or holds with
    another shape, is an error when `strict`.  Returns the list of restored names."""
    import torch
    from . import tflib as lib
    targets = [n for n in lib._params if _in_scopes(n, scopes)]
    have = {n: (shape, dt) for n, shape, dt in list_variables(prefix)}
    # checkpoint key of a registry name: the name TensorFlow gives the variable (tflib ops create theirs inside
    # tf.name_scope(name): `Discriminator.1/Discriminator.1.Filters`); the bare registry name is accepted too
    key = {}
    for n in targets:
        k = lib.tf_variable_name(n)
        key[n] = k if k in have else (n if n in have else None)
    missing = [n for n in targets if key[n] is None]
    if missing and strict:
        raise Exception("checkpoint %s lacks %d variable(s), e.g. %s" % (prefix, len(missing), lib.tf_variable_name(missing[0])))
    names = [n for n in targets if key[n] is not None]
    bad = [n for n in names if tuple(have[key[n]][0]) != tuple(lib._params[n].shape)]
    if bad:
        raise Exception("checkpoint %s: shape of %s is %s, the model's is %s" % (
            prefix, key[bad[0]], have[key[bad[0]]][0], tuple(lib._params[bad[0]].shape)))
    values = load_checkpoint(prefix, [key[n] for n in names], verify)
    with torch.no_grad():
        for n in names:
            p = lib._params[n]
            v = np.array(values[key[n]], dtype=np.float32, order="C")
            p.data.copy_(torch.from_numpy(v.reshape(-1)).to(p.device).reshape(p.shape))
    if names and any(lib._params[n].is_cuda for n in names):
        # the 'bf16' / 'bf16x3' / 'f32w' kernels read images DERIVED from the masters (bf16 shadows, Winograd images): a restore into a
        # live trainer must re-derive them, or forward / dgrad would run on the filters from before the restore
        from . import hip_ops
        hip_ops.refresh_all_derived()
    return names


def restore_from_config(config):
    """The optional restores of the reference's `init_net` (trainer.py:179-212, tester.py:17-64), in its order: a
    checkpoint prefix per group of variable scopes, then `ckpt_path` for everything."""
    done = []
    for attr, scopes in RESTORE_GROUPS:
        path = getattr(config, attr, None)
        if path:
            done += restore(path, scopes=scopes)
    return done


RESTORE_GROUPS = (("pretrained_path", ["Encoder", "ID_AE"]), ("pretrained_appSample_path", ["Gaussian_FC"]),
                  ("pretrained_poseAE_path", ["PoseAE"]), ("pretrained_poseSample_path", ["PoseGaussian"]),
                  ("ckpt_path", None))


def wants_restore(config):
    return any(getattr(config, attr, None) for attr, _ in RESTORE_GROUPS)


def save(prefix, scopes=None, extra=None):
    """`tf.train.Saver(...).save(sess, prefix)`: the registry's variables (of `scopes`) as a V2 checkpoint; `extra`
    adds further named arrays (e.g. a global step)."""
    from . import tflib as lib
    tensors = {lib.tf_variable_name(n): p.detach().cpu().numpy() for n, p in lib._params.items() if _in_scopes(n, scopes)}
    if extra:
        tensors.update(extra)
    save_checkpoint(prefix, tensors)
    return sorted(tensors)
