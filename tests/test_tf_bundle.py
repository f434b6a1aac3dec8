"""TensorFlow tensor-bundle checkpoints without TensorFlow (cmpc-refseg_amd/tf_bundle.py): what trainval_model.py:46-63,136-142 reads
and writes through tf.train.Saver.  The reference ships no checkpoint and TensorFlow is not installed, so these tests pin the reader /
writer by (a) the published CRC-32C vectors (RFC 3720 B.4, the ones LevelDB's and TensorFlow's crc32c tests use), (b) a byte-level
known answer assembled by hand from the format rules in the module's header, (c) round trips incl. multi-block indexes, and
(d) corruption detection.  "parity unpinned" against real TensorFlow output."""
import importlib
import os
import struct
import types

import numpy as np
import pytest

TB = importlib.import_module("cmpc-refseg_amd.tf_bundle")
CK = importlib.import_module("cmpc-refseg_amd.checkpoint")


def test_crc32c_published_vectors():
    vec = [(bytes(32), 0x8A9136AA), (b"\xff" * 32, 0x62A8AB43), (bytes(range(32)), 0x46DD794E), (bytes(range(31, -1, -1)), 0x113FDB5C),
           (b"123456789", 0xE3069283)]
    for data, want in vec:
        assert TB.crc32c(data) == want                      # the library's cmpc_crc32c (SSE4.2 or table)
        assert TB._crc32c_py(0, data) == want               # the pure-Python fallback
    a, b = b"hello ", b"world"
    assert TB.crc32c(b, TB.crc32c(a)) == TB.crc32c(a + b)  # continuation
    big = np.random.default_rng(0).integers(0, 256, 1 << 20, dtype=np.uint8)
    assert TB.crc32c(big) == TB.crc32c(big[: 12345].tobytes() + big[12345:].tobytes())
    # LevelDB's mask: rotate right by 15, add the delta; unmask inverts it
    c = TB.crc32c(b"foo")
    assert TB.unmask_crc(TB.mask_crc(c)) == c and TB.mask_crc(c) != c
    assert TB.mask_crc(0) == 0xA282EAD8


def test_library_exports_crc32c():
    lib = importlib.import_module("cmpc-refseg_amd")._lib.load()
    assert lib.cmpc_crc32c(0, b"123456789", 9) == 0xE3069283


def test_known_answer_bytes(tmp_path):
    """One float32 [2] variable "v" = (1, 2): every byte of the two files, assembled by hand from the format."""
    p = str(tmp_path / "ka")
    TB.write_bundle(p, {"v": np.array([1.0, 2.0], np.float32)})
    data = open(p + ".data-00000-of-00001", "rb").read()
    assert data == struct.pack("<2f", 1.0, 2.0)
    idx = open(p + ".index", "rb").read()
    header = bytes([0x08, 0x01, 0x1A, 0x02, 0x08, 0x01])                     # num_shards=1; version{producer=1}
    entry = (bytes([0x08, 0x01,                                                # dtype DT_FLOAT
                    0x12, 0x04, 0x12, 0x02, 0x08, 0x02,                        # shape{dim{size=2}}
                    0x28, 0x08,                                                # size 8 (offset 0 and shard 0 are proto3 defaults)
                    0x35]) + struct.pack("<I", TB.mask_crc(TB.crc32c(data))))  # crc32c fixed32, masked
    block = (bytes([0, 0, len(header)]) + header +                             # key "" (shared 0, non-shared 0)
             bytes([0, 1, len(entry)]) + b"v" + entry +                        # key "v"
             struct.pack("<II", 0, 1))                                         # one restart at 0
    assert idx[: len(block)] == block
    assert idx[len(block)] == 0                                                # no compression
    assert struct.unpack_from("<I", idx, len(block) + 1)[0] == TB.mask_crc(TB.crc32c(block + b"\0"))
    assert len(idx) >= 48 and struct.unpack_from("<Q", idx, len(idx) - 8)[0] == 0xDB4775248B80FB57
    r = TB.read_bundle(p)
    assert list(r) == ["v"] and r["v"].dtype == np.float32 and np.array_equal(r["v"], [1.0, 2.0])


def test_round_trip_multi_block_and_dtypes(tmp_path):
    rng = np.random.default_rng(1)
    vs = {f"text_objseg/vis_trans_c{l}_head{h}/{n}": rng.standard_normal((3, 5)).astype(np.float32) for l in (3, 4, 5) for h in range(1, 6) for n in ("DW", "biases")}
    vs.update({k + "/Adam": v * 0 for k, v in list(vs.items())})
    vs.update({"global_step": np.asarray(123456789012, np.int64), "beta1_power": np.asarray(0.5, np.float32), "h": rng.standard_normal(7).astype(np.float16),
               "i32": np.arange(6, dtype=np.int32).reshape(2, 3), "empty": np.zeros((0, 4), np.float32), "d": rng.standard_normal((2, 2, 2)),
               "flag": np.array([True, False])})
    p = str(tmp_path / "m-1")
    TB.write_bundle(p, vs, block_size=256)                                    # forces many data blocks and index entries
    r = TB.read_bundle(p)
    assert sorted(r) == sorted(vs)
    for k, v in vs.items():
        assert r[k].dtype == np.asarray(v).dtype and r[k].shape == np.asarray(v).shape and np.array_equal(r[k], v), k
    lv = TB.list_variables(p)
    assert lv["global_step"] == (np.dtype(np.int64), ()) and lv["i32"] == (np.dtype(np.int32), (2, 3))
    sub = TB.read_bundle(p, names=["h", "global_step"])
    assert sorted(sub) == ["global_step", "h"]
    with pytest.raises(KeyError):
        TB.read_bundle(p, names=["nope"])


def test_corruption_is_detected(tmp_path):
    p = str(tmp_path / "c")
    TB.write_bundle(p, {"w": np.arange(64, dtype=np.float32)})
    d = bytearray(open(p + ".data-00000-of-00001", "rb").read()); d[17] ^= 1
    open(p + ".data-00000-of-00001", "wb").write(bytes(d))
    with pytest.raises(ValueError, match="checksum"):
        TB.read_bundle(p)
    assert TB.read_bundle(p, verify=False)["w"].shape == (64,)
    TB.write_bundle(p, {"w": np.arange(64, dtype=np.float32)})
    i = bytearray(open(p + ".index", "rb").read()); i[3] ^= 0x40
    open(p + ".index", "wb").write(bytes(i))
    with pytest.raises(ValueError):
        TB.read_bundle(p)
    open(p + ".index", "wb").write(b"not a table")
    with pytest.raises(ValueError, match="magic"):
        TB.read_bundle(p)


def test_snappy_blocks_of_foreign_tables():
    # literal "abcd", then a copy of 8 bytes at offset 4 (overlapping its own output), then literal "xy"
    comp = bytes([14, (4 - 1) << 2]) + b"abcd" + bytes([((8 - 4) << 2) | 1, 4]) + bytes([(2 - 1) << 2]) + b"xy"
    assert TB._snappy_decompress(comp) == b"abcdabcdabcdxy"


def test_saver_writes_and_reads_tensorflow_checkpoints(tmp_path):
    """checkpoint.Saver(fmt="tf") without a GPU: `<prefix>-<step>.index/.data-*`, the `checkpoint` state file, rotation, and a restore
    that round-trips parameters, Adam slots and the step through the TensorFlow files."""
    import torch
    idx = {"text_objseg/c5_lateral/DW": (0, (1, 1, 2, 3)), "text_objseg/c5_lateral/biases": (8, (3,))}
    eng = types.SimpleNamespace(index=idx, params=torch.arange(12.0), m=torch.ones(12), v=torch.full((12,), 2.0), step=0, pack=lambda: None)
    model = types.SimpleNamespace(eng=eng, device=torch.device("cpu"), backbone_vars={"conv1/weights": torch.zeros(7, 7, 3, 4)},
                                  load_backbone=lambda named: setattr(model, "backbone_vars", dict(named)))
    orig, orig_dev = torch.cuda.synchronize, torch.cuda.device
    torch.cuda.synchronize = lambda *a, **k: None
    import contextlib
    torch.cuda.device = lambda d: contextlib.nullcontext()
    try:
        sv = CK.Saver(max_to_keep=2, fmt="tf")
        paths = []
        for step in (5, 10, 15):
            eng.step = step
            paths.append(sv.save(model, str(tmp_path / "snap")))
        assert [os.path.basename(p) for p in paths] == ["snap-5", "snap-10", "snap-15"]
        assert sorted(os.listdir(tmp_path)) == ["checkpoint", "snap-10.data-00000-of-00001", "snap-10.index", "snap-15.data-00000-of-00001", "snap-15.index"]
        assert CK.latest_checkpoint(str(tmp_path / "snap")) == paths[-1]
        assert TB.read_checkpoint_state(str(tmp_path)) == paths[-1]
        lv = TB.list_variables(paths[-1])
        # the names tf.global_variables() has in the reference graph (checkpoint.py header; parity-unpinned: no TF-written file to compare)
        assert lv["text_objseg/c5_lateral/DW"] == (np.dtype(np.float32), (1, 1, 2, 3)) and lv["text_objseg/Variable_1"] == (np.dtype(np.int32), ())
        assert {"text_objseg/text_objseg/c5_lateral/DW/Adam", "text_objseg/text_objseg/c5_lateral/biases/Adam_1", "text_objseg/beta1_power",
                "text_objseg/beta2_power", "conv1/weights"} <= set(lv)
        assert not {"global_step", "beta1_power", "text_objseg/c5_lateral/DW/Adam"} & set(lv)
        eng.params, eng.m, eng.v, eng.step = torch.zeros(12), torch.zeros(12), torch.zeros(12), 0
        CK.Saver().restore(model, paths[-1])
        live = torch.tensor([0, 1, 2, 3, 4, 5, 8, 9, 10])                      # the elements the two variables cover
        assert torch.equal(eng.params[live], torch.arange(12.0)[live]) and torch.equal(eng.m[live], torch.ones(9)) and torch.equal(eng.v[live], torch.full((9,), 2.0))
        assert eng.step == 15 and float(eng.params[6]) == 0.0
        # rounds 1-2 wrote `<var>/Adam`, `global_step`: still accepted on restore
        old = {"text_objseg/c5_lateral/DW": np.full((1, 1, 2, 3), 7, np.float32), "text_objseg/c5_lateral/biases": np.zeros(3, np.float32),
               "text_objseg/c5_lateral/DW/Adam": np.full((1, 1, 2, 3), 3, np.float32), "text_objseg/c5_lateral/DW/Adam_1": np.full((1, 1, 2, 3), 4, np.float32),
               "text_objseg/c5_lateral/biases/Adam": np.zeros(3, np.float32), "text_objseg/c5_lateral/biases/Adam_1": np.zeros(3, np.float32),
               "global_step": np.asarray(42, np.int64), "conv1/weights": np.zeros((7, 7, 3, 4), np.float32)}
        CK.restore_variables(model, old)
        assert eng.step == 42 and float(eng.params[0]) == 7 and float(eng.m[0]) == 3 and float(eng.v[5]) == 4
        # a weights-only file restores but says so; a file missing a selected variable raises, with or without a filter
        weights_only = {k: v for k, v in old.items() if "Adam" not in k and k != "global_step"}
        with pytest.warns(UserWarning, match="weights only"):
            CK.restore_variables(model, weights_only)
        with pytest.raises(KeyError):
            CK.restore_variables(model, {k: v for k, v in old.items() if k != "text_objseg/c5_lateral/biases"})
        with pytest.raises(KeyError):
            CK.restore_variables(model, {"bn_conv1/gamma": np.ones(4, np.float32)}, var_filter=CK.is_backbone_var)    # lacks conv1/weights
        CK.restore_variables(model, {"conv1/weights": np.ones((7, 7, 3, 4), np.float32)}, var_filter=CK.is_backbone_var)
        assert float(model.backbone_vars["conv1/weights"].sum()) == 7 * 7 * 3 * 4
    finally:
        torch.cuda.synchronize, torch.cuda.device = orig, orig_dev


def test_v1_checkpoint_known_answer_and_round_trip(tmp_path):
    """V1 (single-file, TensorSliceWriter) checkpoints -- the format of deeplab_resnet_init.ckpt (trainval_model.py:50).  A byte-level known
    answer assembled by hand from saved_tensor_slice.proto (one float32 [2] tensor "v" = (1, 2)), a multi-block round trip of several
    dtypes incl. a scalar, a hand-built SLICED tensor (two SavedSlice entries with explicit extents) and corruption detection.  No V1 file
    written by TensorFlow exists here: parity unpinned."""
    p = str(tmp_path / "m.ckpt")
    TB.write_v1_checkpoint(p, {"v": np.array([1.0, 2.0], np.float32)})
    raw = open(p, "rb").read()
    meta = bytes([0x0A, 0x11,                                                  # SavedTensorSlices.meta (len 17)
                  0x0A, 0x0F,                                                  #   tensor (SavedSliceMeta, len 15)
                  0x0A, 0x01]) + b"v" + bytes([0x12, 0x04, 0x12, 0x02, 0x08, 0x02,   # name "v"; shape{dim{size=2}}
                  0x18, 0x01,                                                  #   type DT_FLOAT
                  0x22, 0x02, 0x0A, 0x00])                                     #   slice{extent{}}  (the whole dimension)
    data = bytes([0x12, 0x13,                                                  # SavedTensorSlices.data (SavedSlice, len 19)
                  0x0A, 0x01]) + b"v" + bytes([0x12, 0x02, 0x0A, 0x00,         # name; slice{extent{}}
                  0x1A, 0x0A, 0x2A, 0x08]) + struct.pack("<2f", 1.0, 2.0)      # data{float_val (packed) = 1, 2}
    key = bytes([0x00]) + b"v" + bytes([0x00, 0x01]) + bytes([0x01, 0x01]) + bytes([0x80, 0x7F])    # OrderedCode: 0, "v", rank 1, start 0, length -1
    block = (bytes([0, 0, len(meta)]) + meta + bytes([0, len(key), len(data)]) + key + data + struct.pack("<II", 0, 1))
    assert raw[: len(block)] == block and raw[len(block)] == 0
    assert TB.is_v1_checkpoint(p) and not TB.is_v1_checkpoint(__file__)
    r = TB.read_v1_checkpoint(p)
    assert list(r) == ["v"] and r["v"].dtype == np.float32 and np.array_equal(r["v"], [1.0, 2.0])
    rng = np.random.default_rng(1)
    vs = {"conv1/weights": rng.standard_normal((7, 7, 3, 64)).astype(np.float32), "bn_conv1/moving_variance": rng.random(64).astype(np.float32),
          "global_step": np.asarray(7, np.int32), "counts": (np.arange(12, dtype=np.int64).reshape(3, 4) - 6), "dbl": rng.standard_normal((3, 2))}
    TB.write_v1_checkpoint(p, vs, block_size=1024)
    r = TB.read_v1_checkpoint(p)
    assert set(r) == set(vs)
    for k in vs:
        assert r[k].dtype == vs[k].dtype and r[k].shape == vs[k].shape and np.array_equal(r[k], vs[k]), k
    assert list(TB.read_v1_checkpoint(p, names=["global_step"])) == ["global_step"]
    with pytest.raises(KeyError):
        TB.read_v1_checkpoint(p, names=["nope"])
    # a partitioned tensor: rows [0,2) and [2,3) of a [3,2] float tensor as two SavedSlice entries
    full = np.arange(6, dtype=np.float32).reshape(3, 2)
    shp = TB._msg(TB._f_bytes(2, TB._f_varint(1, 3)), TB._f_bytes(2, TB._f_varint(1, 2)))
    def ext(start, length): return TB._f_bytes(1, TB._msg(TB._f_varint(1, start), TB._f_varint(2, length)))
    s1, s2 = TB._msg(ext(0, 2), TB._f_bytes(1, b"")), TB._msg(ext(2, 1), TB._f_bytes(1, b""))
    metab = TB._f_bytes(1, TB._f_bytes(1, TB._msg(TB._f_bytes(1, b"p"), TB._f_bytes(2, shp), TB._f_varint(3, 1), TB._f_bytes(4, s1), TB._f_bytes(4, s2))))
    def sl(s, vals): return TB._f_bytes(2, TB._msg(TB._f_bytes(1, b"p"), TB._f_bytes(2, s), TB._f_bytes(3, TB._f_bytes(5, vals.astype("<f4").tobytes()))))
    open(p, "wb").write(TB._write_table([(b"", metab), (b"\x00p\x00\x01a", sl(s1, full[:2])), (b"\x00p\x00\x01b", sl(s2, full[2:]))], 4096))
    assert np.array_equal(TB.read_v1_checkpoint(p)["p"], full)
    open(p, "wb").write(TB._write_table([(b"", metab), (b"\x00p\x00\x01a", sl(s1, full[:2]))], 4096))
    with pytest.raises(ValueError):
        TB.read_v1_checkpoint(p)                                               # a slice is missing
    TB.write_v1_checkpoint(p, vs, block_size=1024)
    b = bytearray(open(p, "rb").read()); b[100] ^= 0x40; open(p, "wb").write(bytes(b))
    with pytest.raises(ValueError):
        TB.read_v1_checkpoint(p)                                               # block checksum


def test_saver_restores_backbone_from_v1_checkpoint(tmp_path):
    """trainval_model.py:50-54: Saver(var_list = backbone variables).restore(sess, deeplab_resnet_init.ckpt) -- a V1 file."""
    import contextlib
    import torch
    calls = {}
    bb = {"conv1/weights": np.ones((7, 7, 3, 8), np.float32), "bn_conv1/gamma": np.full(8, 2.0, np.float32)}
    eng = types.SimpleNamespace(index={}, pack=lambda: None, step=0)
    model = types.SimpleNamespace(eng=eng, device=torch.device("cpu"), backbone_vars={k: torch.zeros(v.shape) for k, v in bb.items()},
                                  load_backbone=lambda named: calls.setdefault("bb", named))
    p = str(tmp_path / "deeplab_resnet_init.ckpt")
    TB.write_v1_checkpoint(p, dict(bb, **{"fc1_voc12_c0/weights": np.zeros((3, 3, 8, 21), np.float32)}))
    orig, orig_dev = torch.cuda.synchronize, torch.cuda.device
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.device = lambda d: contextlib.nullcontext()
    try:
        CK.Saver(var_filter=CK.is_backbone_var).restore(model, p)
    finally:
        torch.cuda.synchronize, torch.cuda.device = orig, orig_dev
    assert set(calls["bb"]) >= set(bb) and all(np.array_equal(np.asarray(calls["bb"][k]), bb[k]) for k in bb)
