"""TensorFlow-free reader of the reference's TFRecord / tf.train.Example training records:
known-answer vectors of the public formats, round trips, and (GPU) a batch assembled from records."""
import io
import struct

import numpy as np
import pytest

from dpig_amd import tfrecord as T


def test_crc32c_known_answers():
    # RFC 3720 (iSCSI) appendix B.4 test vectors of CRC32C
    assert T.crc32c(b"123456789") == 0xE3069283
    assert T.crc32c(bytes(32)) == 0x8A9136AA
    assert T.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    assert T.crc32c(b"") == 0
    # TFRecord masking: ((crc >> 15) | (crc << 17)) + 0xa282ead8 mod 2^32
    crc = 0xE3069283
    assert T.masked_crc32c(b"123456789") == ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def test_crc32c_chunk_parallel_path_equals_the_byte_loop():
    """Inputs of 2 KB and more take the numpy path (equal-length chunks advanced in lock step, registers combined with
    the GF(2) "append n zero bytes" operator); it must agree with the byte-at-a-time table loop."""
    rng = np.random.RandomState(0)
    for n in (2047, 2048, 2049, 4097, 65536, 100003, 300001):
        d = rng.randint(0, 256, n).astype(np.uint8).tobytes()
        assert T.crc32c(d) == T._crc32c_scalar(d), n
    for d in (bytes(5000), b"\xff" * 7001, bytes(3000) + b"abc", b"abc" + bytes(3000)):
        assert T.crc32c(d) == T._crc32c_scalar(d)
    a = rng.randn(33, 77).astype(np.float32)
    assert T.crc32c(a) == T._crc32c_scalar(a.tobytes())              # arrays are read through their bytes


def test_tfrecord_framing_roundtrip_and_corruption():
    payloads = [b"", b"a", bytes(range(256)) * 5]
    f = io.BytesIO()
    T.write_records(f, payloads)
    raw = f.getvalue()
    assert len(raw) == sum(16 + len(p) for p in payloads)
    assert struct.unpack("<Q", raw[:8])[0] == 0
    assert list(T.read_records(io.BytesIO(raw))) == payloads
    bad = bytearray(raw); bad[-6] ^= 1                       # flip a payload bit of the last record
    with pytest.raises(IOError):
        list(T.read_records(io.BytesIO(bytes(bad))))
    assert len(list(T.read_records(io.BytesIO(bytes(bad)), check_crc=False))) == 3
    with pytest.raises(IOError):
        list(T.read_records(io.BytesIO(raw[:-3])))


def test_example_wire_format_known_bytes():
    """A hand-assembled tf.train.Example: Features{ 'a': Int64List[1, -1, 300] (packed), 'b': FloatList[0.5, -2]
    (UNPACKED, one fixed32 per element), 'c': BytesList['xy', ''] }."""
    int_list = bytes([0x0A, 0x0D, 0x01]) + bytes([0xFF] * 9 + [0x01]) + bytes([0xAC, 0x02])      # 1, -1, 300
    feat_a = bytes([0x1A, len(int_list)]) + int_list
    fl = bytes([0x0D]) + struct.pack("<f", 0.5) + bytes([0x0D]) + struct.pack("<f", -2.0)
    feat_b = bytes([0x12, len(fl)]) + fl
    bl = bytes([0x0A, 0x02]) + b"xy" + bytes([0x0A, 0x00])
    feat_c = bytes([0x0A, len(bl)]) + bl
    def entry(k, feat):
        e = bytes([0x0A, len(k)]) + k + bytes([0x12, len(feat)]) + feat
        return bytes([0x0A, len(e)]) + e
    feats = entry(b"a", feat_a) + entry(b"b", feat_b) + entry(b"c", feat_c)
    ex = T.parse_example(bytes([0x0A, len(feats)]) + feats)
    assert ex["a"].tolist() == [1, -1, 300] and ex["a"].dtype == np.int64
    assert ex["b"].tolist() == [0.5, -2.0] and ex["b"].dtype == np.float32
    assert ex["c"] == [b"xy", b""]
    # the module's own encoder (packed lists) parses back to the same values
    back = T.parse_example(T.encode_example({"a": ex["a"], "b": ex["b"], "c": ex["c"]}))
    assert back["a"].tolist() == [1, -1, 300] and back["b"].tolist() == [0.5, -2.0] and back["c"] == [b"xy", b""]


def _pair_example(rng, H=128, W=64, nparts=37, fmt=b"raw"):
    """A record with the schema of datasets/market1501.py:79-141 (the fields the stage-I trainer reads)."""
    ex = {"image_format": [fmt], "label": np.array([1]), "image_height": np.array([H]), "image_width": np.array([W])}
    truth = {}
    for s in ("0", "1"):
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        rcv = np.zeros((18, 3), np.float32)
        rcv[:, 0] = rng.integers(0, H, 18); rcv[:, 1] = rng.integers(0, W, 18); rcv[:, 2] = rng.uniform(size=18) < 0.85
        mask = (rng.uniform(size=(H * W,)) < 0.35).astype(np.int64)
        bbox = np.zeros((nparts, 4), np.int64)
        bbox[:, 0] = rng.integers(0, H // 2, nparts); bbox[:, 1] = rng.integers(0, W // 2, nparts)
        bbox[:, 2] = bbox[:, 0] + rng.integers(8, H // 2, nparts); bbox[:, 3] = bbox[:, 1] + rng.integers(8, W // 2, nparts)
        vis = (rng.uniform(size=nparts) < 0.9).astype(np.int64)
        ex.update({"image_raw_" + s: [img.tobytes()], "pose_peaks_%s_rcv" % s: rcv.reshape(-1),
                   "pose_mask_r6_" + s: mask, "part_bbox_" + s: bbox.reshape(-1), "part_vis_" + s: vis})
        truth[s] = dict(img=img, rcv=rcv, mask=mask, bbox=bbox, vis=vis)
    return T.encode_example(ex), truth


def test_decode_pair_record_schema():
    rng = np.random.default_rng(5)
    payload, truth = _pair_example(rng)
    f = io.BytesIO(); T.write_records(f, [payload]); f.seek(0)
    ex = T.parse_example(next(T.read_records(f)))
    for which in (0, 1):
        d = T.decode_pair(ex, which)
        t = truth[str(which)]
        assert np.array_equal(d["x"], (t["img"].astype(np.float32) - 127.5) / 127.5)
        assert np.array_equal(d["pose_rcv"], t["rcv"].reshape(-1))
        assert np.array_equal(d["mask_r6"][..., 0], t["mask"].reshape(128, 64))
        assert np.array_equal(d["part_bbox"], t["bbox"][:7]) and np.array_equal(d["part_vis"], t["vis"][:7])


def test_jpeg_image_field_is_decoded():
    from PIL import Image
    img = np.zeros((128, 64, 3), np.uint8); img[32:96, 16:48] = (200, 40, 90)
    buf = io.BytesIO(); Image.fromarray(img).save(buf, format="PNG")           # lossless, so the pixels are exact
    assert np.array_equal(T.decode_image(buf.getvalue(), "png", 128, 64), img)
    buf = io.BytesIO(); Image.fromarray(img).save(buf, format="JPEG", quality=95)
    dec = T.decode_image(buf.getvalue(), "jpg", 128, 64)
    assert dec.shape == (128, 64, 3) and np.abs(dec.astype(int) - img.astype(int)).mean() < 3


@pytest.mark.gpu
def test_batch_from_records_feeds_the_trainer(dev):
    """Records -> batch dict on the device (pose maps rasterised by dpig_pose_rasterize) -> one encoder forward."""
    import torch
    import dpig_amd.tflib as lib
    from dpig_amd import slim
    from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg
    from oracle import ops as O
    rng = np.random.default_rng(6)
    recs = [_pair_example(rng) for _ in range(2)]
    f = io.BytesIO(); T.write_records(f, [r[0] for r in recs]); f.seek(0)
    exs = [T.parse_example(p) for p in T.read_records(f)]
    batch = T.batch_from_examples(exs, dev, which=0)
    assert tuple(batch["x"].shape) == (2, 128, 64, 3) and tuple(batch["pose"].shape) == (2, 128, 64, 18)
    rcv = torch.from_numpy(np.stack([r[1]["0"]["rcv"].reshape(-1) for r in recs])).double()
    ref = O.tf_poseInflate(O.coord2channel_simple_rcv(rcv, 18, False, 128, 64), 18, 4, 128, 64)
    assert torch.equal(batch["pose"].cpu().double(), ref)
    lib.delete_all_params(); slim.reset_scopes()
    np.random.seed(0)
    tr = DPIG_Encoder_GAN_BodyROI_FgBg(Config(batch_size=2, conv_hidden_num=16, z_num=8), dev)
    tr.init_net(batch)
    tr.step = 1
    out = tr.train_step(batch, T.batch_from_examples(exs, dev, which=1))
    assert all(np.isfinite(float(v)) for v in out.values() if hasattr(v, "numel") and v.numel() == 1)
    # the same records without the rasterised map: the keypoints go straight to the generator's first conv (same losses)
    kb = T.batch_from_examples(exs, dev, which=0, dense_pose=False)
    assert "pose" not in kb and tuple(kb["pose_rcv"].shape) == (2, 54)
    a = tr._g_optim_eager(batch, update=False)
    b = tr._g_optim_eager(kb, update=False)
    assert abs(float(a["g_loss"]) - float(b["g_loss"])) <= 1e-5 * abs(float(a["g_loss"]))


from hypothesis import given, settings, strategies as st   # noqa: E402

_feature = st.one_of(st.lists(st.binary(max_size=20), max_size=3),
                     st.lists(st.floats(-1e6, 1e6, width=32), max_size=6).map(lambda v: np.array(v, np.float32)),
                     st.lists(st.integers(-(1 << 62), 1 << 62), max_size=6).map(lambda v: np.array(v, np.int64)))


@settings(max_examples=80, deadline=None)
@given(st.lists(st.dictionaries(st.text(alphabet="abc_019", min_size=1, max_size=12), _feature, max_size=6), max_size=4))
def test_example_and_framing_round_trip_generated(examples):
    """encode_example -> write_records -> read_records -> parse_example returns what went in (bytes lists, float32 and
    int64 arrays incl. negative values and empty lists)."""
    buf = io.BytesIO()
    T.write_records(buf, [T.encode_example(e) for e in examples])
    buf.seek(0)
    got = [T.parse_example(p) for p in T.read_records(buf)]
    assert len(got) == len(examples)
    for g, e in zip(got, examples):
        assert set(g) == set(e)
        for k, v in e.items():
            if isinstance(v, list):
                assert list(g[k]) == v
            else:
                assert np.array_equal(np.asarray(g[k]), v) and (len(v) == 0 or np.asarray(g[k]).dtype == v.dtype)


def _example_message_classes():
    """tf.train.Example built in the protobuf runtime from its published schema (tensorflow/core/example/feature.proto,
    example.proto: BytesList / FloatList / Int64List { repeated value = 1 }, Feature oneof { bytes_list = 1, float_list = 2,
    int64_list = 3 }, Features { map<string, Feature> feature = 1 }, Example { Features features = 1 }) -- an independent
    implementation of the wire format (TensorFlow itself is not installed; its message definitions are these six)."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    f = descriptor_pb2.FileDescriptorProto(name="dpig_test_example.proto", package="dpigtest", syntax="proto3")
    T = descriptor_pb2.FieldDescriptorProto

    def msg(name):
        m = f.message_type.add()
        m.name = name
        return m

    def field(m, name, num, typ, label=T.LABEL_OPTIONAL, type_name=None, packed=None, oneof=None):
        fd = m.field.add()
        fd.name, fd.number, fd.type, fd.label = name, num, typ, label
        if type_name:
            fd.type_name = type_name
        if packed is not None:
            fd.options.packed = packed
        if oneof is not None:
            fd.oneof_index = oneof
        return fd

    field(msg("BytesList"), "value", 1, T.TYPE_BYTES, T.LABEL_REPEATED)
    field(msg("FloatList"), "value", 1, T.TYPE_FLOAT, T.LABEL_REPEATED, packed=True)
    field(msg("Int64List"), "value", 1, T.TYPE_INT64, T.LABEL_REPEATED, packed=True)
    feat = msg("Feature")
    feat.oneof_decl.add().name = "kind"
    field(feat, "bytes_list", 1, T.TYPE_MESSAGE, type_name=".dpigtest.BytesList", oneof=0)
    field(feat, "float_list", 2, T.TYPE_MESSAGE, type_name=".dpigtest.FloatList", oneof=0)
    field(feat, "int64_list", 3, T.TYPE_MESSAGE, type_name=".dpigtest.Int64List", oneof=0)
    feats = msg("Features")
    entry = feats.nested_type.add()
    entry.name = "FeatureEntry"
    entry.options.map_entry = True
    field(entry, "key", 1, T.TYPE_STRING)
    field(entry, "value", 2, T.TYPE_MESSAGE, type_name=".dpigtest.Feature")
    field(feats, "feature", 1, T.TYPE_MESSAGE, T.LABEL_REPEATED, type_name=".dpigtest.Features.FeatureEntry")
    field(msg("Example"), "features", 1, T.TYPE_MESSAGE, type_name=".dpigtest.Features")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(f)
    get = getattr(message_factory, "GetMessageClass", None)
    desc = pool.FindMessageTypeByName("dpigtest.Example")
    return get(desc) if get else message_factory.MessageFactory(pool).GetPrototype(desc)


def test_example_codec_against_the_protobuf_runtime():
    """`parse_example` reads what the protobuf library serialises for tf.train.Example's schema and the library parses what
    `encode_example` writes: the record payload format is pinned against an independent implementation (negative int64 = 10-byte
    varints, packed float / int64 lists, empty lists, several byte strings, unicode keys)."""
    pytest.importorskip("google.protobuf")
    from dpig_amd import tfrecord as R
    Example = _example_message_classes()
    rng = np.random.RandomState(5)
    want = {"image_raw_0": [rng.bytes(300)], "pose_peaks_0": rng.rand(128 * 64 * 18 // 64).astype(np.float32),
            "part_bbox_0": np.array([0, 0, 1, 1, 5, 7, 120, 60, -1, -(2 ** 62), 2 ** 62, 300], dtype=np.int64),
            "many": [b"a", b"", b"xyz" * 40], "empty_f": np.zeros(0, np.float32), "id_é": np.array([7], dtype=np.int64)}
    ex = Example()
    for k, v in want.items():
        f = ex.features.feature[k]
        if isinstance(v, list):
            f.bytes_list.value.extend(v)
        elif v.dtype.kind == "f":
            f.float_list.value.extend(v.tolist())
            if v.size == 0:
                f.float_list.SetInParent()
        else:
            f.int64_list.value.extend(v.tolist())
    got = R.parse_example(ex.SerializeToString())
    assert set(got) == set(want)
    for k, v in want.items():
        if isinstance(v, list):
            assert got[k] == v, k
        else:
            assert got[k].dtype == v.dtype and np.array_equal(got[k], v), k
    back = Example()
    back.ParseFromString(R.encode_example(want))
    assert set(back.features.feature.keys()) == set(want)
    for k, v in want.items():
        f = back.features.feature[k]
        if isinstance(v, list):
            assert list(f.bytes_list.value) == v, k
        elif v.dtype.kind == "f":
            assert np.array_equal(np.array(f.float_list.value, dtype=np.float32), v), k
        else:
            assert list(f.int64_list.value) == v.tolist(), k


def test_packed_varints_vectorised_equals_scalar_decoder():
    """`_packed_varints` (one numpy pass over a packed Int64List) against the byte-by-byte `_varint` loop: 1- to 10-byte
    varints, negatives (two's complement, 10 bytes), the 0 / 1 mask fast path, empty input, truncation."""
    rng = np.random.default_rng(9)
    vals = np.concatenate([rng.integers(0, 2, 300), rng.integers(-5, 300, 200), rng.integers(-2 ** 62, 2 ** 62, 100),
                           np.array([0, 127, 128, 16383, 16384, -1, 2 ** 63 - 1, -2 ** 63])]).astype(np.int64)
    rng.shuffle(vals)
    blob = b"".join(T._enc_varint(int(v) & (2 ** 64 - 1)) for v in vals)
    got = T._packed_varints(memoryview(blob))
    want, pos = [], 0
    while pos < len(blob):
        x, pos = T._varint(blob, pos)
        want.append(T._signed64(x))
    assert got.dtype == np.int64 and got.tolist() == want == vals.tolist()
    mask = rng.integers(0, 2, 8192).astype(np.int64)
    assert np.array_equal(T._packed_varints(bytes(mask.astype(np.uint8))), mask)           # single-byte fast path
    assert T._packed_varints(b"").size == 0
    with pytest.raises(ValueError):
        T._packed_varints(bytes([0x80, 0x80]))                                             # no terminator
    back = T.parse_example(T.encode_example({"m": vals}))
    assert np.array_equal(back["m"], vals)


def test_record_feeder_yields_the_batches_in_record_order():
    rng = np.random.default_rng(10)
    recs = [_pair_example(rng, nparts=7) for _ in range(5)]
    payloads = [r[0] for r in recs]
    feeder = T.RecordFeeder(payloads, batch_size=2, which=1, workers=3, depth=3, pin=False)
    try:
        seen = [next(feeder) for _ in range(6)]                  # 12 records = 2.4 passes over the 5
    finally:
        feeder.close()
    for j, b in enumerate(seen):
        for i in range(2):
            t = recs[(2 * j + i) % 5][1]["1"]
            assert np.array_equal(b["x"][i].numpy(), (t["img"].astype(np.float32) - 127.5) / 127.5)
            assert np.array_equal(b["pose_rcv"][i].numpy(), t["rcv"].reshape(-1))
            assert np.array_equal(b["mask_r6"][i, ..., 0].numpy(), t["mask"].reshape(128, 64).astype(np.float32))
            assert np.array_equal(b["part_bbox"][i].numpy(), t["bbox"][:7]) and tuple(b["pose_rcv"].shape) == (2, 54)
