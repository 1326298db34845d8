"""TensorFlow-free reader / writer for the reference's tfrecord files.

The reference stores its VQA examples as tf.Example protos in TFRecord shards
(data/tools/vqa_v2/generator_tf_record_memft_genome.py:184-194) and parses them with
tf.parse_single_example (vqa/datasets/input_ops_vqa_tf_record_memft.py:28-57).  TensorFlow is not
available here, so this module implements the two public formats directly:
  * TFRecord framing:  uint64 length | uint32 masked_crc32c(length) | data | uint32 masked_crc32c(data)
  * tf.Example:  Example{1: Features{1: map<string, Feature>}}, Feature = oneof {1: BytesList,
    2: FloatList (packed or not), 3: Int64List (packed varints or not)}
Only what the VQA schema needs (bytes / float / int64 lists) is supported.
"""
from __future__ import annotations

import glob
import os
import struct

import numpy as np

# ----------------------------------------------------------------------------- crc32c (Castagnoli)
_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        poly, tab = 0x82F63B78, []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ poly if c & 1 else c >> 1
            tab.append(c)
        _CRC_TABLE = tab
    return _CRC_TABLE


def crc32c(data: bytes) -> int:
    tab, c = _crc_table(), 0xFFFFFFFF
    for b in data:
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ----------------------------------------------------------------------------- TFRecord framing
def read_records(path, verify=False):
    with open(path, "rb") as f:
        while True:
            head = f.read(12)
            if len(head) == 0:
                return
            if len(head) < 12:
                raise IOError("truncated TFRecord header in %s" % path)
            (length,), (lcrc,) = struct.unpack("<Q", head[:8]), struct.unpack("<I", head[8:])
            data = f.read(length)
            tail = f.read(4)
            if len(data) < length or len(tail) < 4:
                raise IOError("truncated TFRecord in %s" % path)
            if verify:
                if masked_crc32c(head[:8]) != lcrc or masked_crc32c(data) != struct.unpack("<I", tail)[0]:
                    raise IOError("TFRecord CRC mismatch in %s" % path)
            yield data


def write_records(path, records):
    with open(path, "wb") as f:
        for data in records:
            head = struct.pack("<Q", len(data))
            f.write(head + struct.pack("<I", masked_crc32c(head)) + data + struct.pack("<I", masked_crc32c(data)))


# ----------------------------------------------------------------------------- protobuf wire format
def _varint(buf, i):
    x = shift = 0
    while True:
        b = buf[i]
        i += 1
        x |= (b & 0x7F) << shift
        if not b & 0x80:
            return x, i
        shift += 7


def _fields(buf):
    i, n = 0, len(buf)
    while i < n:
        key, i = _varint(buf, i)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(buf, i)
        elif wt == 1:
            v, i = buf[i:i + 8], i + 8
        elif wt == 2:
            ln, i = _varint(buf, i)
            v, i = buf[i:i + ln], i + ln
        elif wt == 5:
            v, i = buf[i:i + 4], i + 4
        else:
            raise ValueError("unsupported wire type %d" % wt)
        yield num, wt, v


def _signed64(x):
    return x - (1 << 64) if x >= (1 << 63) else x


def parse_example(data: bytes) -> dict:
    """tf.Example bytes -> {name: list[bytes] | np.float32[] | np.int64[]}"""
    out = {}
    for num, wt, features in _fields(data):
        if num != 1:
            continue
        for n2, _, entry in _fields(features):          # map<string, Feature> entries
            if n2 != 1:
                continue
            key, feat = None, b""
            for n3, _, v in _fields(entry):
                if n3 == 1:
                    key = bytes(v).decode("utf-8")
                elif n3 == 2:
                    feat = v
            val = []
            for kind, _, lst in _fields(feat):
                if kind == 1:                            # BytesList
                    val = [bytes(v) for n4, _, v in _fields(lst) if n4 == 1]
                elif kind == 2:                          # FloatList
                    fl = []
                    for n4, w4, v in _fields(lst):
                        if n4 == 1:
                            fl.extend(np.frombuffer(bytes(v), "<f4").tolist())
                    val = np.asarray(fl, np.float32)
                elif kind == 3:                          # Int64List
                    il = []
                    for n4, w4, v in _fields(lst):
                        if n4 != 1:
                            continue
                        if w4 == 0:
                            il.append(_signed64(v))
                        else:
                            j, b = 0, bytes(v)
                            while j < len(b):
                                x, j = _varint(b, j)
                                il.append(_signed64(x))
                    val = np.asarray(il, np.int64)
            out[key] = val
    return out


def _enc_varint(x):
    x &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        out.append(b | (0x80 if x else 0))
        if not x:
            return bytes(out)


def _ld(num, payload):
    return _enc_varint((num << 3) | 2) + _enc_varint(len(payload)) + payload


def make_example(features: dict) -> bytes:
    """{name: bytes | str | list[int] | list[float] | np.ndarray} -> serialized tf.Example"""
    entries = b""
    for key in sorted(features):
        v = features[key]
        if isinstance(v, (bytes, str)):
            b = v.encode() if isinstance(v, str) else v
            feat = _ld(1, _ld(1, b))
        else:
            a = np.atleast_1d(np.asarray(v))
            if a.dtype.kind == "f":
                feat = _ld(2, _ld(1, a.astype("<f4").tobytes()))
            else:
                feat = _ld(3, _ld(1, b"".join(_enc_varint(int(x)) for x in a)))
        entries += _ld(1, _ld(1, key.encode()) + _ld(2, feat))
    return _ld(1, entries)


# ----------------------------------------------------------------------------- VQA split loader
def load_vqa_split(tf_record_dir, split, num_answers, verify=False):
    """Reads `<tf_record_dir>/<split>/<split>-*` shards (the layout of
    input_ops_vqa_tf_record_memft.create, :15-16) into an input_ops_vqa.SplitData."""
    from .input_ops_vqa import SplitData
    files = sorted(glob.glob(os.path.join(tf_record_dir, split, "{}-*".format(split))))
    if not files:
        raise FileNotFoundError("no tfrecord shards under %s" % os.path.join(tf_record_dir, split))
    qid, image_id, image_idx, q, qoff, aid, asc, aoff = [], [], [], [], [0], [], [], [0]
    for fn in files:
        for rec in read_records(fn, verify):
            ex = parse_example(rec)
            qid.append(int(ex["qid"][0]))
            image_id.append(ex["image_id"][0].decode())
            image_idx.append(int(ex["image_idx"][0]))
            seq = np.asarray(ex["q_intseq/list"], np.int32)
            assert len(seq) == int(ex["q_intseq/len"][0])
            q.append(seq)
            qoff.append(qoff[-1] + len(seq))
            aid.append(np.asarray(ex["answers/ids"], np.int32))
            asc.append(np.asarray(ex["answers/scores"], np.float32))
            aoff.append(aoff[-1] + len(aid[-1]))
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)
    return SplitData(np.asarray(qid, np.int64), np.asarray(image_id), np.asarray(image_idx, np.int64),
                     cat(q, np.int32), np.asarray(qoff, np.int64), cat(aid, np.int32), cat(asc, np.float32),
                     np.asarray(aoff, np.int64), num_answers)
