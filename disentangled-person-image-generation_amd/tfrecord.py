"""TensorFlow-free reader of the reference's training records (second half of the input-pipeline row).

The reference feeds the trainer from TFRecord files of `tf.train.Example` protos written by
`datasets/convert_market.py` and declared in `datasets/market1501.py:79-141`; decoding, batching and the pose-map
rasterisation happen inside the TF graph (`trainer.py:537-564`).  This module restates the two public formats --
the TFRecord framing (length, masked CRC32C of the length, payload, masked CRC32C of the payload) and the protobuf
wire encoding of `Example{Features{map<string, Feature{BytesList|FloatList|Int64List}>}}` -- in plain Python/numpy,
and assembles the batch dict the trainers consume (`synthetic.make_batch` has the same keys), with the pose maps
rasterised on the device by `dpig_pose_rasterize`.

Nothing here touches TensorFlow.  A writer is included so that tests (and users without the dataset) can
produce records; files written by TensorFlow use exactly this framing and wire format.
"""
import io
import struct

import numpy as np

# ---- CRC32C (Castagnoli), reflected, table driven; TFRecord masks it: rot-right 15, + 0xa282ead8 -----------------
_POLY = 0x82F63B78
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ _POLY if _c & 1 else _c >> 1
    _TABLE.append(_c)
_TABLE = np.array(_TABLE, dtype=np.uint32)


def _crc32c_scalar(data):
    crc = 0xFFFFFFFF
    tab = _TABLE
    for b in bytes(data):
        crc = int(tab[(crc ^ b) & 0xFF]) ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


# The register update is linear over GF(2) in (state, data): the CRC of a long message is assembled from the raw
# (zero-initialised) registers of equal-length chunks, all advanced together one byte per numpy step, and the
# "append n zero bytes" operator Z_n (a 32x32 bit matrix, here 32 column words) that shifts an earlier register past
# the bytes that follow it.  Leading zero bytes leave a zero register at zero, so the front is padded for free.
def _gf2_apply(cols, v):
    out, i = 0, 0
    while v:
        if v & 1:
            out ^= cols[i]
        v >>= 1
        i += 1
    return out


def _zeros_operator(nbytes):
    """Columns of Z_n: register -> register after n zero bytes."""
    result = [1 << i for i in range(32)]                                       # identity
    power = [int(_TABLE[(1 << i) & 0xFF]) ^ ((1 << i) >> 8) for i in range(32)]  # one zero byte
    while nbytes:
        if nbytes & 1:
            result = [_gf2_apply(power, c) for c in result]
        power = [_gf2_apply(power, c) for c in power]
        nbytes >>= 1
    return result


def crc32c(data):
    """CRC-32C (Castagnoli) of a bytes-like object; large inputs take the chunk-parallel numpy path."""
    a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data.reshape(-1).view(np.uint8)
    n = a.size
    if n < 2048:
        return _crc32c_scalar(a.tobytes())
    m = int(min(65536, max(64, n // 512)))                  # chunks advanced in lock step
    L = -(-n // m)
    padded = np.zeros(m * L, dtype=np.uint8)
    padded[m * L - n:] = a
    cols = np.ascontiguousarray(padded.reshape(m, L).T)     # [L, m]: byte i of every chunk is contiguous
    state = np.zeros(m, dtype=np.uint32)
    tab = _TABLE
    for i in range(L):
        state = tab[(state ^ cols[i]) & 0xFF] ^ (state >> 8)
    zl = _zeros_operator(L)
    acc = 0
    for r in state.tolist():
        acc = _gf2_apply(zl, acc) ^ r
    return (_gf2_apply(_zeros_operator(n), 0xFFFFFFFF) ^ acc) ^ 0xFFFFFFFF


def masked_crc32c(data):
    crc = crc32c(data)
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---- TFRecord framing ----------------------------------------------------------------------------------------
def read_records(path_or_file, check_crc=True):
    """Yield the payload of every record of a TFRecord file."""
    f = open(path_or_file, "rb") if isinstance(path_or_file, str) else path_or_file
    try:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) != 12:
                raise IOError("truncated TFRecord header")
            length, lcrc = struct.unpack("<QI", head)
            if check_crc and masked_crc32c(head[:8]) != lcrc:
                raise IOError("TFRecord length CRC mismatch")
            data = f.read(length)
            tail = f.read(4)
            if len(data) != length or len(tail) != 4:
                raise IOError("truncated TFRecord payload")
            if check_crc and masked_crc32c(data) != struct.unpack("<I", tail)[0]:
                raise IOError("TFRecord payload CRC mismatch")
            yield data
    finally:
        if isinstance(path_or_file, str):
            f.close()


def write_records(path_or_file, payloads):
    f = open(path_or_file, "wb") if isinstance(path_or_file, str) else path_or_file
    try:
        for data in payloads:
            head = struct.pack("<Q", len(data))
            f.write(head + struct.pack("<I", masked_crc32c(head)) + data + struct.pack("<I", masked_crc32c(data)))
    finally:
        if isinstance(path_or_file, str):
            f.close()


# ---- protobuf wire format (the subset tf.train.Example uses) ----------------------------------------------------
def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _fields(buf):
    """Yield (field number, wire type, value) of one message; value is an int (varint / fixed) or a memoryview."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val = bytes(buf[pos:pos + 8]); pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos:pos + ln]; pos += ln
        elif wt == 5:
            val = bytes(buf[pos:pos + 4]); pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield num, wt, val


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _packed_varints(v):
    """All varints of a packed repeated field at once -> np.int64 (two's complement of the 64-bit values).  The records' masks are
    8192-element Int64Lists (`pose_mask_r6_*`, datasets/market1501.py:107-108): a python loop over them costs ~7 ms per record side,
    this ~20 us."""
    a = np.frombuffer(v, dtype=np.uint8)
    if a.size == 0:
        return np.zeros(0, np.int64)
    last = a < 0x80                                            # terminator byte of every varint
    if bool(last.all()):                                       # (the common case: 0 / 1 masks, small boxes)
        return a.astype(np.int64)
    if not last[-1]:
        raise ValueError("truncated varint in a packed field")
    ends = np.flatnonzero(last)
    starts = np.empty_like(ends)
    starts[0] = 0
    starts[1:] = ends[:-1] + 1
    if int((ends - starts).max()) > 9:
        raise ValueError("varint longer than 10 bytes")
    idx = np.arange(a.size) - np.repeat(starts, ends - starts + 1)
    contrib = (a & 0x7F).astype(np.uint64) << (np.uint64(7) * idx.astype(np.uint64))
    return np.add.reduceat(contrib, starts).view(np.int64)


def parse_example(payload):
    """tf.train.Example -> {name: list of bytes | np.float32 array | np.int64 array}."""
    out = {}
    buf = memoryview(payload)
    for num, wt, features in _fields(buf):
        if num != 1 or wt != 2:
            continue
        for fnum, fwt, entry in _fields(features):                 # map<string, Feature> entries
            if fnum != 1 or fwt != 2:
                continue
            name, feat = None, None
            for enum, ewt, val in _fields(entry):
                if enum == 1:
                    name = bytes(val).decode("utf-8")
                elif enum == 2:
                    feat = val
            if name is None:
                continue
            value = []
            for knum, kwt, lst in _fields(feat if feat is not None else b""):
                if knum == 1:                                        # BytesList
                    value = [bytes(v) for n_, w_, v in _fields(lst) if n_ == 1]
                elif knum == 2:                                      # FloatList: packed or repeated fixed32
                    vals = []
                    for n_, w_, v in _fields(lst):
                        if n_ != 1:
                            continue
                        vals.append(np.frombuffer(bytes(v), dtype="<f4") if w_ == 2 else np.frombuffer(v, dtype="<f4"))
                    value = np.concatenate(vals).astype(np.float32) if vals else np.zeros(0, np.float32)
                elif knum == 3:                                      # Int64List: packed or repeated varint
                    vals = []
                    for n_, w_, v in _fields(lst):
                        if n_ != 1:
                            continue
                        if w_ == 2:
                            vals.append(_packed_varints(v))
                        else:
                            vals.append(np.array([_signed64(v)], dtype=np.int64))
                    value = np.concatenate(vals) if vals else np.zeros(0, np.int64)
            out[name] = value
    return out


def _enc_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(num, data):
    return _enc_varint((num << 3) | 2) + _enc_varint(len(data)) + data


def encode_example(features):
    """{name: bytes | list of bytes | float array | int array} -> serialized tf.train.Example (packed lists)."""
    entries = b""
    for name in sorted(features):
        v = features[name]
        if isinstance(v, (bytes, bytearray)):
            v = [bytes(v)]
        if isinstance(v, list) and (not v or isinstance(v[0], (bytes, bytearray))):
            feat = _ld(1, b"".join(_ld(1, bytes(b)) for b in v))
        else:
            a = np.asarray(v)
            if a.dtype.kind == "f":
                feat = _ld(2, _ld(1, a.astype("<f4").tobytes()))
            else:
                feat = _ld(3, _ld(1, b"".join(_enc_varint(int(x)) for x in a.reshape(-1))))
        entries += _ld(1, _ld(1, name.encode("utf-8")) + _ld(2, feat))
    return _ld(1, entries)


# ---- the reference's record schema (datasets/market1501.py:79-141) -> trainer batch ---------------------------
def decode_image(raw, fmt, H, W):
    """`slim.tfexample_decoder.Image`: JPEG/PNG bytes -> uint8 [H,W,3] (format 'raw': the bytes are the pixels)."""
    if fmt in ("raw", "RAW"):
        return np.frombuffer(raw, dtype=np.uint8).reshape(H, W, 3)
    from PIL import Image
    img = np.asarray(Image.open(io.BytesIO(raw)).convert("RGB"))
    if img.shape[:2] != (H, W):
        raise ValueError("decoded image is %s, expected %s" % (img.shape[:2], (H, W)))
    return img


def decode_pair(example, which=0, img_H=128, img_W=64, part_indices=range(7)):
    """One side (`which` = 0 source / 1 target) of a pair record as the arrays `_load_batch_pair_pose`
    (trainer.py:537-564) produces, with the part selection of `build_model` (trainer.py:571-578)."""
    s = str(which)
    fmt = example.get("image_format", [b"jpg"])
    fmt = (fmt[0] if len(fmt) else b"jpg").decode()
    img = decode_image(example["image_raw_" + s][0], fmt, img_H, img_W)
    nparts = len(example["part_vis_" + s])
    bbox = np.asarray(example["part_bbox_" + s], dtype=np.int64).reshape(nparts, 4)
    vis = np.asarray(example["part_vis_" + s], dtype=np.int64)
    idx = list(part_indices)
    return {
        "x": (img.astype(np.float32) - 127.5) / 127.5,                                  # process_image(x, 127.5, 127.5)
        "pose_rcv": np.asarray(example["pose_peaks_%s_rcv" % s], dtype=np.float32),
        "mask_r6": np.asarray(example["pose_mask_r6_" + s], dtype=np.float32).reshape(img_H, img_W, 1),
        "part_bbox": bbox[idx].astype(np.int32),
        "part_vis": vis[idx].astype(np.float32),
    }


def host_batch_from_examples(examples, which=0, img_H=128, img_W=64, part_indices=range(7), pin=False):
    """The batch dict as HOST tensors (x, pose_rcv [B, 54], mask_r6, part_bbox, part_vis) -- what `prefetch.DevicePrefetcher`
    uploads; `pin` places them in page-locked memory."""
    import torch
    items = [decode_pair(e, which, img_H, img_W, part_indices) for e in examples]
    out = {}
    for k in ("x", "pose_rcv", "mask_r6", "part_bbox", "part_vis"):
        t = torch.from_numpy(np.stack([it[k] for it in items]))
        if k == "pose_rcv":
            t = t.reshape(t.shape[0], -1).float()
        out[k] = t.pin_memory() if pin else t
    return out


def batch_from_examples(examples, device, which=0, img_H=128, img_W=64, keypoint_num=18, part_indices=range(7), dense_pose=True):
    """Batch dict of device tensors with the keys of `synthetic.make_batch` / `synthetic.to_device`.  The pose arrives as the
    records' (row, col, visibility) triplets `pose_rcv` (is_normalized=False, trainer.py:556-560); with `dense_pose` the
    [B,H,W,18] target map is also rasterised on the device (`pose`) -- without it the trainers feed the keypoints straight to the
    generator's first conv (trainer.pose_input)."""
    from . import utils
    out = {k: v.to(device) for k, v in host_batch_from_examples(examples, which, img_H, img_W, part_indices).items()}
    if dense_pose:
        out["pose"] = utils.pose_target_from_rcv(out["pose_rcv"], keypoint_num, False, img_H, img_W)
    return out


class RecordFeeder(object):
    """The host half of the reference's queue-runner pipeline (`trainer.py:537-564`: `tf.train.batch(..., num_threads=4)` keeps
    decoded batches ready): an endless iterator of host batches (`host_batch_from_examples`) decoded from record payloads on
    `workers` threads, `depth` batches ahead, in record order.  `payloads` is a list of serialized `tf.train.Example`s (e.g.
    `list(read_records(path))`), cycled through in batches of `batch_size`."""

    def __init__(self, payloads, batch_size, which=0, img_H=128, img_W=64, part_indices=range(7), workers=4, depth=4, pin=True):
        import concurrent.futures
        if len(payloads) < 1:
            raise ValueError("RecordFeeder: no records")
        self.payloads, self.B = list(payloads), int(batch_size)
        self.args = (which, img_H, img_W, list(part_indices), pin)
        self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=max(1, int(workers)))
        self.depth, self.cursor, self.pending = max(1, int(depth)), 0, []

    def _decode(self, lo):
        n = len(self.payloads)
        exs = [parse_example(self.payloads[(lo + i) % n]) for i in range(self.B)]
        which, H, W, parts, pin = self.args
        return host_batch_from_examples(exs, which, H, W, parts, pin=pin)

    def __iter__(self):
        return self

    def __next__(self):
        while len(self.pending) < self.depth:
            self.pending.append(self.pool.submit(self._decode, self.cursor))
            self.cursor = (self.cursor + self.B) % len(self.payloads)
        return self.pending.pop(0).result()

    def close(self):
        self.pool.shutdown(wait=False, cancel_futures=True)
