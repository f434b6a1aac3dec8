"""TensorFlow "tensor bundle" checkpoints (the V2 format tf.train.Saver writes: `<prefix>.index` + `<prefix>.data-00000-of-00001`),
read and written without TensorFlow -- what trainval_model.py:46-63 restores (`deeplab_resnet_init.ckpt`, snapshots) and :136-142 saves.

Format, restated from TensorFlow's published sources (tensorflow/core/util/tensor_bundle/tensor_bundle.{h,cc}, tensorflow/core/lib/io/
{format,block,table}.cc, tensorflow/core/protobuf/tensor_bundle.proto; TensorFlow itself is not installed here and the reference ships
no checkpoint, so this module is pinned by round trips and by hand-assembled known-answer bytes only -- "parity unpinned"):

  * the index is a LevelDB-style sorted string table: data blocks of prefix-compressed entries
    [shared varint32][non_shared varint32][value_len varint32][key suffix][value] followed by a uint32 restart array and its
    length; every block is followed by a 5-byte trailer (compression type 0 = none / 1 = snappy, masked CRC-32C of contents + type);
    an index block maps separator keys to (offset, size) handles of the data blocks; the 48-byte footer holds the metaindex and index
    handles and the magic 0xdb4775248b80fb57;
  * key "" holds a BundleHeaderProto {num_shards = 1, endianness = 2, version = 3}; every other key is a variable name whose value is
    a BundleEntryProto {dtype = 1, shape = 2, shard_id = 3, offset = 4, size = 5, crc32c = 6 (fixed32, masked), slices = 7};
  * the data shard holds the tensors' little-endian bytes at those offsets.
Partitioned variables (slices) and string tensors are not handled (the reference's graphs have none).
"""
from __future__ import annotations

import os
import struct
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

MAGIC = 0xDB4775248B80FB57
_MASK_DELTA = 0xA282EAD8
# tensorflow/core/framework/types.proto
_DT = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_, 17: np.uint16,
       19: np.float16, 22: np.uint32, 23: np.uint64}
_DT_OF = {np.dtype(v): k for k, v in _DT.items()}


# ---- CRC-32C ----------------------------------------------------------------------------------------------------------------
_CRC_TABLE: Optional[List[int]] = None


def _crc32c_py(crc: int, data: bytes) -> int:
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = []
        for i in range(256):
            r = i
            for _ in range(8):
                r = (r >> 1) ^ (0x82F63B78 if r & 1 else 0)
            t.append(r)
        _CRC_TABLE = t
    c = crc ^ 0xFFFFFFFF
    t = _CRC_TABLE
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def crc32c(data, crc: int = 0) -> int:
    """CRC-32C of a bytes-like object / contiguous array, continuing from `crc`; the library's SSE4.2 routine (cmpc_crc32c) when the
    library is built, pure Python otherwise."""
    a = np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray, memoryview)) else np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    if a.size == 0:
        return crc
    try:
        from . import _lib
        fn = _lib.load().cmpc_crc32c
    except Exception:
        return _crc32c_py(crc, a.tobytes())
    return int(fn(crc, a.ctypes.data, a.size))


def mask_crc(c: int) -> int:
    return (((c >> 15) | (c << 17)) + _MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(m: int) -> int:
    r = (m - _MASK_DELTA) & 0xFFFFFFFF
    return ((r >> 17) | (r << 15)) & 0xFFFFFFFF


# ---- varints / protobuf wire format -----------------------------------------------------------------------------------------
def _put_varint(v: int) -> bytes:
    if v < 0:
        v += 1 << 64
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _get_varint(b, pos: int) -> Tuple[int, int]:
    r, shift = 0, 0
    while True:
        if pos >= len(b):
            raise ValueError("truncated varint")
        c = b[pos]
        pos += 1
        r |= (c & 0x7F) << shift
        if c < 0x80:
            return r, pos
        shift += 7
        if shift > 63:
            raise ValueError("varint too long")


def _fields(b) -> Iterable[Tuple[int, int, object]]:
    """(field number, wire type, value) of one protobuf message; value = int (varint / fixed) or bytes (length-delimited)."""
    pos = 0
    while pos < len(b):
        key, pos = _get_varint(b, pos)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(b, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", b, pos)[0]; pos += 8
        elif wt == 2:
            n, pos = _get_varint(b, pos)
            if pos + n > len(b):
                raise ValueError("truncated length-delimited field")
            v = bytes(b[pos: pos + n]); pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", b, pos)[0]; pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield f, wt, v


def _msg(*parts: bytes) -> bytes:
    return b"".join(parts)


def _f_varint(f: int, v: int) -> bytes:
    return _put_varint(f << 3) + _put_varint(v)


def _f_bytes(f: int, v: bytes) -> bytes:
    return _put_varint((f << 3) | 2) + _put_varint(len(v)) + v


def _f_fixed32(f: int, v: int) -> bytes:
    return _put_varint((f << 3) | 5) + struct.pack("<I", v)


def _signed64(v: int) -> int:
    return v - (1 << 64) if v >= 1 << 63 else v


# ---- snappy (blocks of foreign checkpoints may be compressed; ours are not) -------------------------------------------------
def _snappy_decompress(b: bytes) -> bytes:
    n, pos = _get_varint(b, 0)
    out = bytearray()
    while pos < len(b):
        tag = b[pos]; pos += 1
        kind = tag & 3
        if kind == 0:                                   # literal
            ln = tag >> 2
            if ln >= 60:
                k = ln - 59
                ln = int.from_bytes(b[pos: pos + k], "little"); pos += k
            ln += 1
            out += b[pos: pos + ln]; pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | b[pos]; pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = int.from_bytes(b[pos: pos + 2], "little"); pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(b[pos: pos + 4], "little"); pos += 4
        if off == 0 or off > len(out):
            raise ValueError("corrupt snappy block")
        for _ in range(ln):                              # copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("snappy: length mismatch")
    return bytes(out)


# ---- table -------------------------------------------------------------------------------------------------------------------
def _read_block(f: bytes, off: int, size: int, verify: bool) -> bytes:
    if off + size + 5 > len(f):
        raise ValueError("block handle points past the end of the index file")
    body, ctype = f[off: off + size], f[off + size]
    if verify:
        want = struct.unpack_from("<I", f, off + size + 1)[0]
        if mask_crc(crc32c(f[off: off + size + 1])) != want:
            raise ValueError("index block checksum mismatch")
    if ctype == 0:
        return body
    if ctype == 1:
        return _snappy_decompress(body)
    raise ValueError(f"unknown block compression {ctype}")


def _block_entries(blk: bytes) -> Iterable[Tuple[bytes, bytes]]:
    if len(blk) < 4:
        raise ValueError("block too short")
    nrest = struct.unpack_from("<I", blk, len(blk) - 4)[0]
    end = len(blk) - 4 - 4 * nrest
    if end < 0:
        raise ValueError("bad restart array")
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(blk, pos)
        non_shared, pos = _get_varint(blk, pos)
        vlen, pos = _get_varint(blk, pos)
        if shared > len(key) or pos + non_shared + vlen > end:
            raise ValueError("corrupt block entry")
        key = key[:shared] + blk[pos: pos + non_shared]; pos += non_shared
        yield key, blk[pos: pos + vlen]
        pos += vlen


def _table_entries(f: bytes, verify: bool) -> Iterable[Tuple[bytes, bytes]]:
    if len(f) < 48 or struct.unpack_from("<Q", f, len(f) - 8)[0] != MAGIC:
        raise ValueError("not a TensorFlow checkpoint index (bad table magic)")
    foot = f[len(f) - 48:]
    _, p = _get_varint(foot, 0); _, p = _get_varint(foot, p)          # metaindex handle (unused)
    ioff, p = _get_varint(foot, p); isz, p = _get_varint(foot, p)
    for _, handle in _block_entries(_read_block(f, ioff, isz, verify)):
        off, q = _get_varint(handle, 0); sz, q = _get_varint(handle, q)
        yield from _block_entries(_read_block(f, off, sz, verify))


class _BlockBuilder:
    def __init__(self, restart_interval: int = 16):
        self.buf, self.restarts, self.count, self.last, self.ri = bytearray(), [0], 0, b"", restart_interval

    def add(self, key: bytes, value: bytes):
        shared = 0
        if self.count < self.ri:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf)); self.count = 0
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
        self.last = key; self.count += 1

    def size(self) -> int:
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self) -> bytes:
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def _write_table(entries: List[Tuple[bytes, bytes]], block_size: int) -> bytes:
    out = bytearray()

    def emit(block: bytes) -> bytes:
        off = len(out)
        out.extend(block); out.append(0)
        out.extend(struct.pack("<I", mask_crc(crc32c(block + b"\0"))))
        return _put_varint(off) + _put_varint(len(block))

    index = _BlockBuilder(1)
    cur = _BlockBuilder()
    for k, v in entries:                                     # sorted by key
        cur.add(k, v)
        if cur.size() >= block_size:
            index.add(cur.last, emit(cur.finish())); cur = _BlockBuilder()   # separator = the block's last key (valid: >= it, < the next)
    if cur.count or not entries:
        index.add(cur.last, emit(cur.finish()))
    meta = emit(_BlockBuilder().finish())
    idx = emit(index.finish())
    foot = meta + idx
    out.extend(foot + b"\0" * (40 - len(foot)) + struct.pack("<Q", MAGIC))
    return bytes(out)


# ---- bundle ------------------------------------------------------------------------------------------------------------------
def _parse_entry(v: bytes):
    dtype, shape, shard, off, size, crc, sliced = 0, [], 0, 0, 0, None, False
    for f, wt, x in _fields(v):
        if f == 1: dtype = x
        elif f == 2:
            for f2, _, d in _fields(x):
                if f2 == 2:
                    sz = 0
                    for f3, _, y in _fields(d):
                        if f3 == 1: sz = _signed64(y)
                    shape.append(sz)
                elif f2 == 3 and d:
                    raise ValueError("tensor of unknown rank in checkpoint")
        elif f == 3: shard = x
        elif f == 4: off = x
        elif f == 5: size = x
        elif f == 6: crc = x
        elif f == 7: sliced = True
    return dtype, tuple(shape), shard, off, size, crc, sliced


def list_variables(prefix: str, verify: bool = True) -> Dict[str, Tuple[np.dtype, Tuple[int, ...]]]:
    """tf.train.list_variables: name -> (dtype, shape)."""
    with open(prefix + ".index", "rb") as fh:
        f = fh.read()
    out = {}
    for k, v in _table_entries(f, verify):
        if k == b"":
            continue
        dt, shape, *_ = _parse_entry(v)
        out[k.decode()] = (np.dtype(_DT[dt]) if dt in _DT else None, shape)
    return out


def read_bundle(prefix: str, names: Optional[Iterable[str]] = None, verify: bool = True) -> Dict[str, np.ndarray]:
    """tf.train.load_checkpoint(prefix).get_tensor(name) for every (or the selected) variable.  verify: check the table blocks' and the
    tensors' CRC-32C.  Raises ValueError on a corrupt or unsupported file, KeyError on a selected name the checkpoint lacks."""
    with open(prefix + ".index", "rb") as fh:
        f = fh.read()
    want = None if names is None else set(names)
    entries, nshards = {}, 1
    for k, v in _table_entries(f, verify):
        if k == b"":
            for fno, _, x in _fields(v):
                if fno == 1: nshards = x
                elif fno == 2 and x != 0:
                    raise ValueError("big-endian checkpoint")
            continue
        name = k.decode()
        if want is None or name in want:
            entries[name] = _parse_entry(v)
    if want is not None and want - set(entries):
        raise KeyError(f"checkpoint {prefix} lacks {sorted(want - set(entries))[:3]}")
    shards: Dict[int, np.memmap] = {}
    out: Dict[str, np.ndarray] = {}
    for name, (dt, shape, shard, off, size, crc, sliced) in entries.items():
        if sliced:
            raise ValueError(f"{name}: partitioned variables are not supported")
        if dt not in _DT:
            raise ValueError(f"{name}: unsupported dtype enum {dt}")
        if shard not in shards:
            path = f"{prefix}.data-{shard:05d}-of-{nshards:05d}"
            shards[shard] = np.memmap(path, dtype=np.uint8, mode="r") if os.path.getsize(path) else np.zeros(0, np.uint8)
        dtype = np.dtype(_DT[dt])
        n = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if size != n * dtype.itemsize or off + size > shards[shard].size:
            raise ValueError(f"{name}: entry size {size} does not match shape {shape} / data file")
        raw = np.array(shards[shard][off: off + size])                  # copy out of the mapping
        if verify and crc is not None and mask_crc(crc32c(raw)) != crc:
            raise ValueError(f"{name}: tensor checksum mismatch")
        out[name] = raw.view(dtype).reshape(shape)
    return out


def write_bundle(prefix: str, variables: Dict[str, np.ndarray], block_size: int = 262144) -> None:
    """tf.train.Saver().save's two files for `variables` (name -> array), one data shard."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    entries: List[Tuple[bytes, bytes]] = [(b"", _msg(_f_varint(1, 1), _f_bytes(3, _f_varint(1, 1))))]      # num_shards 1, little endian, producer 1
    off = 0
    with open(f"{prefix}.data-00000-of-00001", "wb") as fh:
        for name in sorted(variables, key=lambda s: s.encode()):
            if not name:
                raise ValueError("empty variable name")
            a = np.asarray(variables[name], order="C")               # (ascontiguousarray would turn a scalar into shape (1,))
            if a.dtype not in _DT_OF:
                raise ValueError(f"{name}: dtype {a.dtype} has no TensorFlow checkpoint encoding here")
            if a.dtype.byteorder == ">":
                a = a.astype(a.dtype.newbyteorder("<"))
            raw = a.tobytes()
            shape = _msg(*[_f_bytes(2, _f_varint(1, int(d))) for d in a.shape])
            e = _msg(_f_varint(1, _DT_OF[a.dtype]), _f_bytes(2, shape), _f_varint(4, off) if off else b"", _f_varint(5, len(raw)),
                     _f_fixed32(6, mask_crc(crc32c(raw))))
            entries.append((name.encode(), e))
            fh.write(raw); off += len(raw)
    with open(prefix + ".index", "wb") as fh:
        fh.write(_write_table(entries, block_size))


# ---- V1 checkpoints: tf.train.Saver(write_version=V1), one file written by TensorSliceWriter --------------------------------------
# The file IS a sorted string table (same block / footer format as the V2 index above).  Entry "" holds a SavedTensorSlices message whose
# `meta` (field 1) lists every tensor: SavedSliceMeta {name = 1, shape = 2 (TensorShapeProto), type = 3, slice = 4}.  Every other entry
# holds a SavedTensorSlices whose `data` (field 2) is a SavedSlice {name = 1, slice = 2 (TensorSliceProto: repeated Extent {start = 1,
# length = 2}; an extent with neither is the whole dimension), data = 3 (TensorProto with the VALUES in its typed repeated field: float_val
# = 5, double_val = 6, int_val = 7, int64_val = 10, bool_val = 11, half_val = 13; or tensor_content = 4)}
# (tensorflow/core/util/saved_tensor_slice.proto, tensor_slice_writer.{h,cc}).  The entry keys (OrderedCode of name + slice) are not
# needed for reading.  This is the format of `deeplab_resnet_init.ckpt` (trainval_model.py:50); restated from the published format --
# no V1 file written by TensorFlow exists here: PARITY UNPINNED.
def is_v1_checkpoint(path: str) -> bool:
    if not os.path.isfile(path) or os.path.getsize(path) < 48:
        return False
    with open(path, "rb") as fh:
        fh.seek(-8, os.SEEK_END)
        return struct.unpack("<Q", fh.read(8))[0] == MAGIC


def _shape_of(b: bytes) -> Tuple[int, ...]:
    shape = []
    for f2, _, d in _fields(b):
        if f2 == 2:
            sz = 0
            for f3, _, y in _fields(d):
                if f3 == 1: sz = _signed64(y)
            shape.append(sz)
    return tuple(shape)


def _extents(b: bytes, shape: Tuple[int, ...]):
    """TensorSliceProto -> [(start, length)] per dimension."""
    ext = []
    for f, _, e in _fields(b):
        if f != 1:
            continue
        start, length = 0, None
        for f2, _, y in _fields(e):
            if f2 == 1: start = _signed64(y)
            elif f2 == 2: length = _signed64(y)
        d = len(ext)
        ext.append((start, (shape[d] - start) if length is None or length < 0 else length))
    while len(ext) < len(shape):
        ext.append((0, shape[len(ext)]))
    return ext


_V1_FIELD = {5: (np.float32, 5, "<f4"), 6: (np.float64, 1, "<f8")}        # packed fixed-width fields of TensorProto: field -> (dtype, wire, fmt)


def _tensor_values(b: bytes, dtype: np.dtype) -> np.ndarray:
    """The values of a TensorProto written by TensorSliceWriter::SaveData (typed repeated field) or as tensor_content."""
    parts: List[np.ndarray] = []
    for f, wt, x in _fields(b):
        if f == 4 and wt == 2:                                  # tensor_content
            parts.append(np.frombuffer(x, dtype=dtype))
        elif f in _V1_FIELD and np.dtype(_V1_FIELD[f][0]) == dtype:
            parts.append(np.frombuffer(x, dtype=_V1_FIELD[f][2]) if wt == 2 else np.array([struct.unpack("<f" if f == 5 else "<d", struct.pack("<I" if f == 5 else "<Q", x))[0]], dtype))
        elif f in (7, 10, 11, 13) and wt in (0, 2):             # int_val / int64_val / bool_val / half_val: varints (packed or not)
            vals = []
            if wt == 0:
                vals.append(x)
            else:
                pos = 0
                while pos < len(x):
                    v, pos = _get_varint(x, pos)
                    vals.append(v)
            if f == 13:
                parts.append(np.array(vals, dtype=np.uint16).view(np.float16))
            else:
                parts.append(np.array([_signed64(v) for v in vals], dtype=np.int64).astype(dtype))
    return np.concatenate(parts) if parts else np.zeros(0, dtype)


def read_v1_checkpoint(path: str, names: Optional[Iterable[str]] = None, verify: bool = True) -> Dict[str, np.ndarray]:
    with open(path, "rb") as fh:
        f = fh.read()
    want = None if names is None else set(names)
    meta: Dict[str, Tuple[np.dtype, Tuple[int, ...]]] = {}
    out: Dict[str, np.ndarray] = {}
    filled: Dict[str, int] = {}
    for k, v in _table_entries(f, verify):
        for fno, _, x in _fields(v):
            if fno == 1:                                        # SavedTensorSliceMeta
                for f1, _, t in _fields(x):
                    if f1 != 1:
                        continue
                    name, shape, dt = "", (), 1
                    for f2, _, y in _fields(t):
                        if f2 == 1: name = y.decode()
                        elif f2 == 2: shape = _shape_of(y)
                        elif f2 == 3: dt = y
                    if dt not in _DT:
                        raise ValueError(f"{name}: unsupported dtype enum {dt}")
                    meta[name] = (np.dtype(_DT[dt]), shape)
            elif fno == 2:                                      # SavedSlice
                name, sl, data = "", b"", b""
                for f1, _, y in _fields(x):
                    if f1 == 1: name = y.decode()
                    elif f1 == 2: sl = y
                    elif f1 == 3: data = y
                if name not in meta:
                    raise ValueError(f"V1 checkpoint: data entry for {name!r} before / without its metadata")
                if want is not None and name not in want:
                    continue
                dtype, shape = meta[name]
                ext = _extents(sl, shape)
                vals = _tensor_values(data, dtype)
                n = int(np.prod([l for _, l in ext], dtype=np.int64)) if ext else 1
                if vals.size != n:
                    raise ValueError(f"{name}: slice holds {vals.size} values, its extents say {n}")
                if name not in out:
                    out[name] = np.zeros(shape, dtype)
                    filled[name] = 0
                out[name][tuple(slice(s, s + l) for s, l in ext)] = vals.reshape([l for _, l in ext])
                filled[name] += n
    for name, (dtype, shape) in meta.items():
        if want is not None and name not in want:
            continue
        total = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if filled.get(name, 0) != total:
            raise ValueError(f"{name}: {filled.get(name, 0)} of {total} values present")
    if want is not None and want - set(out):
        raise KeyError(f"checkpoint {path} lacks {sorted(want - set(out))[:3]}")
    return out


def _ordered_num(v: int) -> bytes:          # OrderedCode::WriteNumIncreasing
    b = v.to_bytes(8, "big").lstrip(b"\0")
    return bytes([len(b)]) + b


def _ordered_str(s: bytes) -> bytes:        # OrderedCode::WriteString
    return s.replace(b"\0", b"\0\xff").replace(b"\xff", b"\xff\0") + b"\0\x01"


def write_v1_checkpoint(path: str, variables: Dict[str, np.ndarray], block_size: int = 262144) -> None:
    """A V1 checkpoint of whole (unsliced) float32 / float64 / int32 / int64 tensors, as TensorSliceWriter lays it out."""
    metas, entries = [], []
    for name in sorted(variables):
        a = np.asarray(variables[name])
        a = a if a.flags.c_contiguous else a.copy()                       # (np.ascontiguousarray would turn a scalar into shape (1,))
        if a.dtype not in _DT_OF:
            raise ValueError(f"{name}: unsupported dtype {a.dtype}")
        shp = _msg(*[_f_bytes(2, _f_varint(1, d)) for d in a.shape])
        full = _msg(*[_f_bytes(1, b"") for _ in a.shape])                  # every extent the whole dimension
        metas.append(_f_bytes(1, _msg(_f_bytes(1, name.encode()), _f_bytes(2, shp), _f_varint(3, _DT_OF[a.dtype]), _f_bytes(4, full))))
        if a.dtype == np.float32: vals = _f_bytes(5, a.astype("<f4").tobytes())
        elif a.dtype == np.float64: vals = _f_bytes(6, a.astype("<f8").tobytes())
        else: vals = _f_bytes(7 if a.dtype == np.int32 else 10, b"".join(_put_varint(int(x) & ((1 << 64) - 1)) for x in a.reshape(-1)))
        # key: OrderedCode(0, name, rank, then (start, length) = (0, -1) per dimension as single-byte signed numbers 0x80, 0x7f)
        key = _ordered_num(0) + _ordered_str(name.encode()) + _ordered_num(a.ndim) + b"\x80\x7f" * a.ndim
        entries.append((key, _f_bytes(2, _msg(_f_bytes(1, name.encode()), _f_bytes(2, full), _f_bytes(3, vals)))))
    entries.sort(key=lambda kv: kv[0])
    table = [(b"", _f_bytes(1, _msg(*metas)))] + entries
    with open(path, "wb") as fh:
        fh.write(_write_table(table, block_size))


def write_checkpoint_state(directory: str, latest: str, all_paths: Iterable[str]) -> None:
    """The `checkpoint` text file tf.train.Saver keeps beside its snapshots (tf.train.latest_checkpoint reads it)."""
    with open(os.path.join(directory, "checkpoint"), "w") as fh:
        fh.write(f'model_checkpoint_path: "{latest}"\n')
        for p in all_paths:
            fh.write(f'all_model_checkpoint_paths: "{p}"\n')


def read_checkpoint_state(directory: str) -> Optional[str]:
    path = os.path.join(directory, "checkpoint")
    if not os.path.exists(path):
        return None
    for line in open(path):
        if line.startswith("model_checkpoint_path:"):
            p = line.split(":", 1)[1].strip().strip('"')
            return p if os.path.isabs(p) else os.path.join(directory, p)
    return None


def read_checkpoint_state_all(directory: str) -> List[str]:
    """all_model_checkpoint_paths of the `checkpoint` state file, in the order written (oldest save first); [] without the file."""
    path = os.path.join(directory, "checkpoint")
    if not os.path.exists(path):
        return []
    out = []
    for line in open(path):
        if line.startswith("all_model_checkpoint_paths:"):
            out.append(line.split(":", 1)[1].strip().strip('"'))
    return out
