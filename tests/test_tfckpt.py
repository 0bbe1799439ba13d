"""TF V2 checkpoint reader / writer (dpig_amd/tfckpt.py): the table and bundle formats against
hand-assembled bytes, round trips, corruption detection, and the Saver-like restore into the tflib registry.
No TensorFlow-written file exists in this environment (the module's header says "unpinned"); what can be checked
independently of the writer is checked here against bytes laid out by hand from the format descriptions."""
import struct

import numpy as np
import torch
import pytest

from dpig_amd import tfckpt as C
from dpig_amd.tfrecord import masked_crc32c


def _v(n):                                    # protobuf / leveldb varint, written out independently of the module
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _trailer(block, ctype=0):
    return bytes([ctype]) + struct.pack("<I", masked_crc32c(block + bytes([ctype])))


def test_hand_assembled_table_and_bundle(tmp_path):
    """A complete checkpoint laid out by hand: one fp32 [2,3] variable 'a/w' and one int64 scalar 'step'."""
    w = np.arange(6, dtype="<f4").reshape(2, 3) - 2.5
    step = np.array(1234567890123, dtype="<i8")
    data = w.tobytes() + step.tobytes()
    prefix = str(tmp_path / "model.ckpt-7")
    open(prefix + ".data-00000-of-00001", "wb").write(data)
    # BundleHeaderProto: num_shards(1)=1, version(3){producer(1)=1}
    header = b"\x08\x01" + b"\x1a\x02\x08\x01"
    # BundleEntryProto: dtype(1), shape(2){dim(2){size(1)}}, offset(4), size(5), crc32c(6, fixed32)
    e_w = (b"\x08\x01" + b"\x12\x08" + b"\x12\x02\x08\x02" + b"\x12\x02\x08\x03" + b"\x28\x18" +
           b"\x35" + struct.pack("<I", masked_crc32c(w.tobytes())))
    e_s = (b"\x08\x09" + b"\x12\x00" + b"\x20\x18" + b"\x28\x08" + b"\x35" + struct.pack("<I", masked_crc32c(step.tobytes())))
    # data block: 3 entries, restart interval 16 -> one restart; (shared, non_shared, value_len, key delta, value)
    blk = b""
    for shared, key, val in ((0, b"", header), (0, b"a/w", e_w), (0, b"step", e_s)):
        blk += _v(shared) + _v(len(key)) + _v(len(val)) + key + val
    blk += struct.pack("<I", 0) + struct.pack("<I", 1)
    f = blk + _trailer(blk)
    meta = struct.pack("<I", 0) + struct.pack("<I", 1)                       # empty block
    meta_off = len(f)
    f += meta + _trailer(meta)
    handle = _v(0) + _v(len(blk))
    idx = _v(0) + _v(4) + _v(len(handle)) + b"step" + handle + struct.pack("<I", 0) + struct.pack("<I", 1)
    idx_off = len(f)
    f += idx + _trailer(idx)
    footer = _v(meta_off) + _v(len(meta)) + _v(idx_off) + _v(len(idx))
    footer += b"\x00" * (40 - len(footer)) + bytes.fromhex("57fb808b247547db")   # magic, little endian
    open(prefix + ".index", "wb").write(f + footer)

    assert C.list_variables(prefix) == [("a/w", (2, 3), np.float32), ("step", (), np.int64)]
    got = C.load_checkpoint(prefix)
    assert got["a/w"].dtype == np.float32 and np.array_equal(got["a/w"], w)
    assert got["step"].shape == () and int(got["step"]) == 1234567890123
    # the writer lays the same two variables out to the same bytes
    p2 = str(tmp_path / "again")
    C.save_checkpoint(p2, {"step": step, "a/w": w})
    assert open(p2 + ".data-00000-of-00001", "rb").read() == data
    assert open(p2 + ".index", "rb").read() == f + footer


def test_prefix_compression_restarts_and_many_blocks(tmp_path):
    rng = np.random.RandomState(0)
    keys = sorted({("scope%d/layer_%03d/%s" % (i % 7, i, s)).encode() for i in range(400) for s in ("weights", "biases")})
    items = [(k, bytes(rng.randint(0, 256, rng.randint(0, 40)).astype(np.uint8))) for k in keys]
    for bs in (64, 700, C.BLOCK_SIZE):
        path = str(tmp_path / ("t%d" % bs))
        C.write_table(path, items, block_size=bs)
        assert C.read_table(path) == items
    # a block with shared prefixes, assembled by hand: 'abc' -> 'abd' (shared 2) -> 'abde' (shared 3)
    blk = (_v(0) + _v(3) + _v(1) + b"abc" + b"1") + (_v(2) + _v(1) + _v(1) + b"d" + b"2") + (_v(3) + _v(1) + _v(0) + b"e")
    blk += struct.pack("<I", 0) + struct.pack("<I", 1)
    assert list(C.block_entries(blk)) == [(b"abc", b"1"), (b"abd", b"2"), (b"abde", b"")]
    with pytest.raises(ValueError):
        C.write_table(str(tmp_path / "bad"), [(b"b", b""), (b"a", b"")])


def test_snappy_block_known_answer(tmp_path):
    # 11 bytes: literal 'abc', then an overlapping copy (offset 3, length 8); then a 2-byte-offset copy form
    assert C.snappy_decompress(b"\x0b" + b"\x08abc" + b"\x11\x03") == b"abcabcabcab"
    assert C.snappy_decompress(b"\x08" + b"\x0cabcd" + b"\x0e\x04\x00") == b"abcdabcd"
    long_lit = bytes(range(70))
    assert C.snappy_decompress(b"\x46" + b"\xf0\x45" + long_lit) == long_lit          # literal length in one extra byte
    with pytest.raises(IOError):
        C.snappy_decompress(b"\x05" + b"\x08abc")
    # a table whose data block is snappy-compressed (type byte 1) is read through the decoder
    blk = _v(0) + _v(0) + _v(6) + b"\x08\x01\x1a\x02\x08\x01" + struct.pack("<I", 0) + struct.pack("<I", 1)
    comp = _v(len(blk)) + bytes([(len(blk) - 1) << 2]) + blk                          # one literal
    f = comp + _trailer(comp, 1)
    meta = struct.pack("<I", 0) + struct.pack("<I", 1)
    mo = len(f); f += meta + _trailer(meta)
    h = _v(0) + _v(len(comp))
    idx = _v(0) + _v(0) + _v(len(h)) + h + struct.pack("<I", 0) + struct.pack("<I", 1)
    io_ = len(f); f += idx + _trailer(idx)
    foot = _v(mo) + _v(len(meta)) + _v(io_) + _v(len(idx))
    foot += b"\x00" * (40 - len(foot)) + struct.pack("<Q", C.TABLE_MAGIC)
    path = str(tmp_path / "snappy.index")
    open(path, "wb").write(f + foot)
    assert C.read_table(path) == [(b"", b"\x08\x01\x1a\x02\x08\x01")]


def test_round_trip_dtypes_shapes_and_corruption(tmp_path):
    rng = np.random.RandomState(1)
    tensors = {"Encoder/G_encoder/Conv/weights": rng.randn(3, 3, 21, 16).astype(np.float32),
               "Discriminator.1.Filters": rng.randn(5, 5, 3, 8).astype(np.float32),
               "Discriminator.BN2.moving_mean": np.zeros(8, np.float32),
               "beta1_power": np.array(0.5, np.float32), "step": np.array(41, np.int32),
               "empty": np.zeros((0, 4), np.float32), "ids": rng.randint(-5, 5, (4, 2)).astype(np.int64),
               "half": rng.randn(7).astype(np.float16), "flag": np.array([True, False]), "d": rng.randn(2, 2)}
    prefix = str(tmp_path / "sub" / "model.ckpt-41")
    C.save_checkpoint(prefix, tensors)
    listed = {n: (s, d) for n, s, d in C.list_variables(prefix)}
    assert set(listed) == set(tensors) and listed["empty"] == ((0, 4), np.float32) and listed["step"] == ((), np.int32)
    got = C.load_checkpoint(prefix)
    for n, a in tensors.items():
        assert got[n].dtype == a.dtype and got[n].shape == a.shape and np.array_equal(got[n], a), n
    assert set(C.load_checkpoint(prefix, names=["step", "ids"])) == {"step", "ids"}
    assert set(C.load_checkpoint(prefix, names=lambda n: n.startswith("Discriminator."))) == {
        "Discriminator.1.Filters", "Discriminator.BN2.moving_mean"}
    assert C.latest_checkpoint(str(tmp_path / "sub")) == prefix and C.latest_checkpoint(str(tmp_path)) is None
    # a flipped data byte / index byte / magic is detected
    dpath, ipath = prefix + ".data-00000-of-00001", prefix + ".index"
    raw = bytearray(open(dpath, "rb").read()); raw[100] ^= 1; open(dpath, "wb").write(bytes(raw))
    with pytest.raises(IOError, match="CRC"):
        C.load_checkpoint(prefix)
    assert "step" in C.load_checkpoint(prefix, verify=False)
    raw[100] ^= 1; open(dpath, "wb").write(bytes(raw))
    idx = bytearray(open(ipath, "rb").read()); idx[10] ^= 0x40; open(ipath, "wb").write(bytes(idx))
    with pytest.raises(IOError, match="CRC"):
        C.list_variables(prefix)
    idx[10] ^= 0x40; idx[-1] ^= 1; open(ipath, "wb").write(bytes(idx))
    with pytest.raises(IOError, match="magic"):
        C.list_variables(prefix)
    idx[-1] ^= 1; open(ipath, "wb").write(bytes(idx[:-3]))
    with pytest.raises(IOError):
        C.list_variables(prefix)


def test_saver_like_restore_into_the_registry(tmp_path):
    """trainer.py:180-212: partial restores by scope, the full restore, strictness."""
    import dpig_amd.tflib as lib
    lib.delete_all_params()
    lib.set_device("cpu")
    try:
        rng = np.random.RandomState(2)
        names = {"Encoder/G_encoder/Conv/weights": (3, 3, 4, 8), "Encoder/G_encoder/Conv/biases": (8,),
                 "ID_AE/G/Conv/weights": (3, 3, 8, 8), "PoseAE/G_Pose_Encoder/fully_connected/weights": (54, 16),
                 "Discriminator.1.Filters": (5, 5, 3, 8), "Discriminator.BN2.moving_mean": (8,)}
        saved = {}
        for n, sh in names.items():
            saved[n] = rng.randn(*sh).astype(np.float32)
            lib.param(n, saved[n], trainable="moving" not in n)
        prefix = str(tmp_path / "model.ckpt-3")
        # tflib conv / linear variables are stored under the name TensorFlow gives them: created inside
        # tf.name_scope(name) (reference tflib/ops/conv2d.py:27,88) -> `Discriminator.1/Discriminator.1.Filters`
        keys = [lib.tf_variable_name(n) for n in names]
        assert "Discriminator.1/Discriminator.1.Filters" in keys and "Discriminator.BN2.moving_mean" in keys
        assert C.save(prefix, extra={"step": np.array(3, np.int32)}) == sorted(keys + ["step"])
        flat_alias = lib._params["ID_AE/G/Conv/weights"].data           # in-place copy must keep this storage
        for p in lib._params.values():
            p.data.fill_(7.0)
        done = C.restore(prefix, scopes=["Encoder", "ID_AE"])
        assert sorted(done) == sorted(n for n in names if n.startswith(("Encoder", "ID_AE")))
        assert np.array_equal(flat_alias.numpy(), saved["ID_AE/G/Conv/weights"])
        assert float(lib._params["Discriminator.1.Filters"].detach().min()) == 7.0 == float(
            lib._params["PoseAE/G_Pose_Encoder/fully_connected/weights"].detach().max())

        class Cfg(object):
            pretrained_path = None
            pretrained_poseAE_path = prefix
            ckpt_path = None
        assert C.wants_restore(Cfg) and C.restore_from_config(Cfg) == ["PoseAE/G_Pose_Encoder/fully_connected/weights"]
        assert len(C.restore(prefix)) == len(names)
        for n in names:
            assert np.array_equal(lib._params[n].detach().numpy(), saved[n])
        # a variable the checkpoint lacks: error like Saver.restore, unless not strict
        lib.param("Encoder/G_encoder/Conv_1/weights", np.zeros((3, 3, 8, 8), np.float32))
        with pytest.raises(Exception, match="lacks"):
            C.restore(prefix, scopes=["Encoder"])
        assert len(C.restore(prefix, scopes=["Encoder"], strict=False)) == 2
        lib._params.pop("Encoder/G_encoder/Conv_1/weights")
        lib._params.pop("Discriminator.1.Filters")
        lib.param("Discriminator.1.Filters", np.zeros((5, 5, 3, 16), np.float32))
        with pytest.raises(Exception, match="shape"):
            C.restore(prefix)
    finally:
        lib.delete_all_params()
        lib.set_device(None)


# ---- generated cases --------------------------------------------------------------------------------------------
from hypothesis import given, settings, strategies as st   # noqa: E402


@settings(max_examples=60, deadline=None)
@given(st.dictionaries(st.binary(min_size=1, max_size=24), st.binary(max_size=64), max_size=60),
       st.sampled_from([32, 200, 4096]))
def test_table_round_trip_generated(tmp_path_factory, kv, block_size):
    items = sorted(kv.items())
    path = str(tmp_path_factory.mktemp("tbl") / "t")
    C.write_table(path, items, block_size=block_size)
    assert C.read_table(path) == items


_names = st.text(alphabet="abcXYZ019_./", min_size=1, max_size=20)
_arrays = st.one_of(
    st.tuples(st.sampled_from(["<f4", "<f8", "<i4", "<i8", "<f2", "u1"]),
              st.lists(st.integers(0, 5), max_size=3)).map(
        lambda t: (np.arange(int(np.prod(t[1])) if t[1] else 1).reshape(t[1]) * 3 - 7).astype(t[0])))


@settings(max_examples=40, deadline=None)
@given(st.dictionaries(_names, _arrays, max_size=12))
def test_bundle_round_trip_generated(tmp_path_factory, tensors):
    prefix = str(tmp_path_factory.mktemp("ck") / "m.ckpt")
    C.save_checkpoint(prefix, tensors, update_state_file=False)
    got = C.load_checkpoint(prefix)
    assert set(got) == set(tensors)
    for n, a in tensors.items():
        assert got[n].dtype == a.dtype and got[n].shape == a.shape and np.array_equal(got[n], a)


def test_tf_variable_names_of_the_full_model_key_list():
    """The checkpoint keys of every variable of model 1 as a `tf.train.Saver()` of the reference names them (SURVEY
    Appendix F): slim variables under their scopes, tflib conv / linear variables inside the name scope of their op,
    BatchNorm parameters bare; Adam slots and the optimizer's powers on top."""
    import dpig_amd.tflib as lib
    cases = {
        "Encoder/G_encoder/Conv_3/weights": "Encoder/G_encoder/Conv_3/weights",
        "ID_AE/G/fully_connected_1/biases": "ID_AE/G/fully_connected_1/biases",
        "Discriminator.1.Filters": "Discriminator.1/Discriminator.1.Filters",
        "Discriminator.4.Biases": "Discriminator.4/Discriminator.4.Biases",
        "Discriminator.Output.W": "Discriminator.Output/Discriminator.Output.W",
        "Discriminator.Output.b": "Discriminator.Output/Discriminator.Output.b",
        "Discriminator.BN2.offset": "Discriminator.BN2.offset",
        "Discriminator.BN3.moving_variance": "Discriminator.BN3.moving_variance",
        "Fg_FCDis_Discriminator.Input.Linear.W": "Fg_FCDis_Discriminator.Input.Linear/Fg_FCDis_Discriminator.Input.Linear.W",
        "Bg_FCDis_Discriminator.2.Linear.b": "Bg_FCDis_Discriminator.2.Linear/Bg_FCDis_Discriminator.2.Linear.b",
        "Generator.5.g": "Generator.5/Generator.5.g",
    }
    for reg, tfn in cases.items():
        assert lib.tf_variable_name(reg) == tfn


def test_full_model_checkpoint_key_list_fixture():
    """tests/golden/ckpt_keys_model1.json is the variable list of the reference's stage-I Market model written down from its
    graph-building code by tests/golden/make_ckpt_keys.py (a flat layer table, independent of this repo's model code).  The
    names this repo gives the same variables -- creation order, shapes, TF scoping -- must be exactly that list (the oracle
    builds them here on CPU at full width; tests/test_model_gpu.py checks registry == oracle and the saved key set)."""
    import json
    import os
    import dpig_amd.tflib as lib
    from dpig_amd import synthetic
    from oracle import models as OM
    fix = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ckpt_keys_model1.json")))
    P = OM.ParamStore(seed=1)
    ob = OM.batch_to_torch(synthetic.make_batch(1, seed=1))
    with torch.no_grad():
        _, G = OM.stage1_forward(P, ob)
        OM.dcgan_discriminator(P, G, "dcgan")
    g = [lib.tf_variable_name(n) for n in OM.g_var_names(P)]
    d = [lib.tf_variable_name(n) for n in OM.d_var_names(P)]
    assert g == fix["g_trainable"] and d == fix["d_trainable"]
    for n, t in P.p.items():
        assert list(t.shape) == fix["keys"][lib.tf_variable_name(n)], n
    saved = set(lib.tf_variable_name(n) for n in P.p)
    slots = {k for k in fix["keys"] if k.endswith(("/Adam", "/Adam_1"))}
    extras = {"beta1_power", "beta2_power", "beta1_power_1", "beta2_power_1", "g_lr", "d_lr", "step"}
    assert saved | slots | extras == set(fix["keys"])
    assert slots == {n + s for n in g + d for s in ("/Adam", "/Adam_1")}


def test_optimizer_slots_round_trip_under_tf_names(tmp_path):
    """trainer.optimizer_slots / load_optimizer_slots: Adam moments as `<var>/Adam`, `<var>/Adam_1`, the powers as
    beta^(t+1) under `beta1_power[_k]`; RMSProp as `<var>/RMSProp`, `<var>/RMSProp_1`; all-or-nothing restore."""
    import torch
    import dpig_amd.tflib as lib
    from dpig_amd.trainer import FlatParams, TFAdam, TFRMSProp, load_optimizer_slots, optimizer_slots
    lib.delete_all_params()
    lib.set_device("cpu")
    try:
        rng = np.random.RandomState(5)
        shapes = {"ID_AE/G/Conv/weights": (3, 3, 2, 5), "ID_AE/G/Conv/biases": (5,), "ID_AE/G/fully_connected/weights": (7, 3)}
        params = [lib.param(n, rng.randn(*sh).astype(np.float32)) for n, sh in shapes.items()]
        flat = FlatParams(params)
        lr = torch.tensor([1e-3])
        opt = TFAdam(flat, lr, beta1=0.5, beta2=0.999)
        flat.m.normal_(); flat.v.uniform_()
        opt.state[0] = 7; opt.t = 7
        slots = optimizer_slots(flat, opt, ordinal=1)
        assert set(slots) == {n + s for n in shapes for s in ("/Adam", "/Adam_1")} | {"beta1_power_1", "beta2_power_1",
                                                                                     "dpig_amd/adam_step_1"}
        assert abs(float(slots["beta1_power_1"]) - 0.5 ** 8) < 1e-9 and abs(float(slots["beta2_power_1"]) - 0.999 ** 8) < 1e-7
        prefix = str(tmp_path / "model.ckpt-7")
        C.save(prefix, extra=slots)
        values = C.load_checkpoint(prefix)
        want_m = {n: slots[n + "/Adam"].copy() for n in shapes}
        want_v = {n: slots[n + "/Adam_1"].copy() for n in shapes}
        flat.m.zero_(); flat.v.zero_(); opt.state.zero_(); opt.t = 0
        assert not load_optimizer_slots(flat, opt, values, ordinal=0)          # the generator's powers are not in there
        assert float(flat.m.abs().sum()) == 0.0
        assert not load_optimizer_slots(flat, opt, {k: v for k, v in values.items() if k != "ID_AE/G/Conv/biases/Adam_1"}, 1)
        assert load_optimizer_slots(flat, opt, values, ordinal=1)
        for p, o in zip(flat.params, flat.offsets):
            n = p.numel()
            assert np.array_equal(flat.m[o:o + n].numpy().reshape(p.shape), want_m[p.dpig_name])
            assert np.array_equal(flat.v[o:o + n].numpy().reshape(p.shape), want_v[p.dpig_name])
        assert opt.t == 7 and int(opt.state[0]) == 7 and int(opt.state[1]) == 0
        # a TensorFlow-written checkpoint has no step key: t comes from the powers -- beta2's, because beta1 = 0.5
        # underflows float32 after ~150 updates (0.5^151 == 0.0f) while 0.999^(t+1) stays representable
        tfv = {k: v for k, v in values.items() if not k.startswith("dpig_amd/")}
        assert load_optimizer_slots(flat, opt, tfv, ordinal=1) and opt.t == 7
        for t in (149, 150, 5000, 60000):
            tfv["beta1_power_1"] = np.float32(0.5 ** (t + 1))
            tfv["beta2_power_1"] = np.float32(0.999 ** (t + 1))
            assert load_optimizer_slots(flat, opt, tfv, ordinal=1)
            assert abs(opt.t - t) <= max(1, t // 2000), (t, opt.t)       # (float32 power: exact to ~1e-7 relative)
        assert float(np.float32(0.5 ** 151)) == 0.0
        tfv["beta1_power_1"] = np.float32(0.0)
        tfv["beta2_power_1"] = np.float32(0.0)                 # both underflowed: bias correction == 1 from here on
        assert load_optimizer_slots(flat, opt, tfv, ordinal=1) and opt.t >= 10 ** 6
        ropt = TFRMSProp(flat, lr)
        names = set(optimizer_slots(flat, ropt))
        assert names == {n + s for n in shapes for s in ("/RMSProp", "/RMSProp_1")}
    finally:
        lib.delete_all_params()
        lib.set_device(None)


@settings(max_examples=60, deadline=None)
@given(st.binary(min_size=0, max_size=4096), st.integers(1, 64))
def test_snappy_decoder_against_an_independent_compressor(data, repeat):
    """The one piece of the checkpoint table format an independent implementation exists for in this image: snappy.  Blocks
    produced by Apache Arrow's bundled Google snappy (`pyarrow.compress(codec='snappy')`: literals, 1/2/4-byte-offset copies,
    overlapping copies from repetition) must decode to the original bytes with `tfckpt.snappy_decompress`."""
    pa = pytest.importorskip("pyarrow")
    if not pa.Codec.is_available("snappy"):
        pytest.skip("pyarrow built without snappy")
    raw = data * repeat
    comp = pa.compress(raw, codec="snappy", asbytes=True)
    assert C.snappy_decompress(comp) == raw
    structured = (bytes(range(256)) * 3 + raw[:200] + b"\x00" * 5000 + raw[::-1][:300]) * 2      # long runs -> long-offset copies
    assert C.snappy_decompress(pa.compress(structured, codec="snappy", asbytes=True)) == structured


def _bundle_message_classes():
    """BundleHeaderProto / BundleEntryProto (tensorflow/core/protobuf/tensor_bundle.proto) with TensorShapeProto
    (framework/tensor_shape.proto) and VersionDef (framework/versions.proto), built in the protobuf runtime from their published field
    numbers -- the index file's VALUES are these messages."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    f = descriptor_pb2.FileDescriptorProto(name="dpig_test_bundle.proto", package="dpigb", syntax="proto3")
    T = descriptor_pb2.FieldDescriptorProto

    def msg(name, parent=None):
        m = (parent.nested_type if parent is not None else f.message_type).add()
        m.name = name
        return m

    def field(m, name, num, typ, label=T.LABEL_OPTIONAL, type_name=None):
        fd = m.field.add()
        fd.name, fd.number, fd.type, fd.label = name, num, typ, label
        if type_name:
            fd.type_name = type_name

    shape = msg("TensorShapeProto")
    dim = msg("Dim", shape)
    field(dim, "size", 1, T.TYPE_INT64)
    field(dim, "name", 2, T.TYPE_STRING)
    field(shape, "dim", 2, T.TYPE_MESSAGE, T.LABEL_REPEATED, ".dpigb.TensorShapeProto.Dim")
    field(shape, "unknown_rank", 3, T.TYPE_BOOL)
    ver = msg("VersionDef")
    field(ver, "producer", 1, T.TYPE_INT32)
    field(ver, "min_consumer", 2, T.TYPE_INT32)
    hdr = msg("BundleHeaderProto")
    field(hdr, "num_shards", 1, T.TYPE_INT32)
    field(hdr, "endianness", 2, T.TYPE_INT32)                 # (enum LITTLE = 0, BIG = 1: same wire type)
    field(hdr, "version", 3, T.TYPE_MESSAGE, type_name=".dpigb.VersionDef")
    ent = msg("BundleEntryProto")
    field(ent, "dtype", 1, T.TYPE_INT32)                      # (enum DataType)
    field(ent, "shape", 2, T.TYPE_MESSAGE, type_name=".dpigb.TensorShapeProto")
    field(ent, "shard_id", 3, T.TYPE_INT32)
    field(ent, "offset", 4, T.TYPE_INT64)
    field(ent, "size", 5, T.TYPE_INT64)
    field(ent, "crc32c", 6, T.TYPE_FIXED32)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(f)
    get = getattr(message_factory, "GetMessageClass", None)
    mk = (lambda n: get(pool.FindMessageTypeByName(n))) if get else (lambda n: message_factory.MessageFactory(pool).GetPrototype(pool.FindMessageTypeByName(n)))
    return mk("dpigb.BundleHeaderProto"), mk("dpigb.BundleEntryProto")


def test_bundle_protos_against_the_protobuf_runtime(tmp_path):
    """The header / entry messages `save_checkpoint` writes into the index parse with the protobuf runtime to the values the
    bundle format prescribes (DT_FLOAT = 1, DT_INT32 = 3, DT_INT64 = 9; offsets in write order; masked CRC32C of the bytes), and
    entries the runtime serialises are read back by `_parse_entry`."""
    pytest.importorskip("google.protobuf")
    Header, Entry = _bundle_message_classes()
    rng = np.random.RandomState(3)
    tensors = {"a/w": rng.rand(3, 3, 4, 8).astype(np.float32), "a/b": rng.rand(8).astype(np.float32),
               "step": np.array(7, dtype=np.int32), "big": np.arange(5, dtype=np.int64), "z/empty": np.zeros((0, 4), np.float32)}
    prefix = str(tmp_path / "model.ckpt-1")
    C.save_checkpoint(prefix, tensors)
    items = C.read_table(prefix + ".index")
    h = Header()
    h.ParseFromString(items[0][1])
    assert (h.num_shards, h.endianness, h.version.producer) == (1, 0, 1)
    off = 0
    dt = {np.dtype(np.float32): 1, np.dtype(np.int32): 3, np.dtype(np.int64): 9}
    for key, val in items[1:]:
        e = Entry()
        e.ParseFromString(val)
        a = tensors[key.decode()]
        assert e.dtype == dt[a.dtype] and [d.size for d in e.shape.dim] == list(a.shape) and e.shard_id == 0
        assert e.offset == off and e.size == a.nbytes and e.crc32c == masked_crc32c(a.tobytes())
        off += a.nbytes
        # ... and the other direction: what the runtime writes for the same entry is what the reader understands
        back = C._parse_entry(e.SerializeToString())
        assert (back["dtype"], back["shape"], back["offset"], back["size"], back["crc32c"]) == (e.dtype, list(a.shape), e.offset, e.size, e.crc32c)


def test_writer_against_the_independent_reader(tmp_path):
    """tfckpt.save_checkpoint's files read back by tests/bundle_reader_independent.py -- a reader that shares no code with the module
    (own varint, CRC-32C table, protobuf walk, block parser): names, shapes, dtypes, bytes, every block and tensor checksum.  Enough
    variables that the table has several data blocks (prefix-compressed keys across restart points, a multi-entry index block)."""
    import bundle_reader_independent as R
    rs = np.random.RandomState(3)
    tensors = {"step": np.array(77, dtype=np.int32), "g_lr": np.array(2e-5, dtype=np.float32)}
    for i in range(180):
        scope = "Encoder/G_encoder/Conv_%d" % i if i % 3 else "Discriminator.%d/Discriminator.%d.Filters" % (i, i)
        tensors[scope + "/weights"] = rs.randn(3, 3, 1 + i % 5, 2 + i % 7).astype(np.float32)
        tensors[scope + "/biases"] = rs.randn(2 + i % 7).astype(np.float32)
    tensors["beta1_power"] = np.array(0.9 ** 5, dtype=np.float32)
    tensors["big/weights"] = rs.randn(20480, 16).astype(np.float32)
    prefix = str(tmp_path / "model.ckpt-77")
    C.save_checkpoint(prefix, tensors)
    got, header = R.read_bundle(prefix)
    assert header.get(1) == 1                                        # num_shards
    assert set(got) == set(tensors)
    for k, v in tensors.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k
    # several data blocks (the bundle's block size holds this whole index in one): the same keys through write_table with small
    # blocks -- prefix-compressed keys across restart points, a multi-entry index block -- read by the independent table reader
    items = sorted((k.encode(), ("value of %s" % k).encode() * (1 + len(k) % 3)) for k in tensors)
    small = str(tmp_path / "small.index")
    C.write_table(small, items, block_size=512)
    raw_small = open(small, "rb").read()
    foot = raw_small[-48:]
    p = 0
    _, p = R.varint(foot, p); _, p = R.varint(foot, p)
    io, p = R.varint(foot, p); isz, p = R.varint(foot, p)
    assert len(R.entries(R.block(raw_small, io, isz))) >= 10
    assert R.read_index(small) == dict(items)
    raw = open(prefix + ".index", "rb").read()
    # ... and the independent CRC agrees with the module's on arbitrary bytes
    blob = rs.bytes(1000)
    assert R.masked(blob) == masked_crc32c(blob)
    # a flipped byte in a data block is caught by the independent reader too
    bad = bytearray(raw); bad[10] ^= 1
    open(prefix + ".index", "wb").write(bytes(bad))
    with pytest.raises(ValueError):
        R.read_bundle(prefix)
