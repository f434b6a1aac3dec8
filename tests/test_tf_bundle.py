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
