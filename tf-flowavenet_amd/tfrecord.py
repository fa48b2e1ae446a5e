"""TFRecord files of the reference (``tfrecord.py:10-88`` writes them, ``dataset.py:47-68`` parses them)
without TensorFlow: the container framing (length, masked CRC-32C, payload, masked CRC-32C) and the
four fields of the ``tf.train.Example`` the reference stores -

    audio      FloatList  [audio_len]         audio_len  Int64List [1]
    mel        FloatList  [frames * mels]     mel_shape  Int64List [2]      (speaker_id Int64List [1], optional)

- decoded with a minimal protobuf wire-format reader (varints + length-delimited fields; packed and
unpacked repeated scalars), so datasets preprocessed by the reference can feed ``train.py``
(SURVEY section 8f-4).  ``TFRecordCreator`` mirrors the reference's writer (same split, same file
names) for data produced by ``preprocessing.preprocess``.  Pure host-side I/O: no arithmetic on the
path beyond the checksum.
"""
from __future__ import annotations

import os
import struct

import numpy as np

# ---------------------------------------------------------------- CRC-32C (Castagnoli), masked as TFRecord does
_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = np.arange(256, dtype=np.uint32)
        for _ in range(8):
            t = np.where(t & 1, (t >> 1) ^ np.uint32(0x82F63B78), t >> 1)
        _CRC_TABLE = t
    return _CRC_TABLE


def crc32c(data: bytes) -> int:
    t = _crc_table()
    crc = 0xFFFFFFFF
    for b in data:
        crc = int(t[(crc ^ b) & 0xFF]) ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def masked_crc(data: bytes) -> int:
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ---------------------------------------------------------------- protobuf wire format (just enough)
def _varint(buf, i):
    shift = val = 0
    while True:
        b = buf[i]
        i += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, i
        shift += 7


def _fields(buf):
    """(field number, wire type, value) of one message; length-delimited values are memoryviews."""
    i, n = 0, len(buf)
    while i < n:
        key, i = _varint(buf, i)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(buf, i)
        elif wt == 1:
            v, i = bytes(buf[i:i + 8]), i + 8
        elif wt == 2:
            ln, i = _varint(buf, i)
            v, i = buf[i:i + ln], i + ln
        elif wt == 5:
            v, i = bytes(buf[i:i + 4]), i + 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield num, wt, v


def _feature(buf):
    """tf.train.Feature -> ndarray (float32 / int64) or list of bytes."""
    for num, _, v in _fields(buf):
        if num == 2:      # FloatList { repeated float value = 1 [packed] }
            out = [np.frombuffer(bytes(x), "<f4") if wt == 2 else np.frombuffer(x, "<f4") for n_, wt, x in _fields(v) if n_ == 1]
            return np.concatenate(out) if out else np.zeros(0, np.float32)
        if num == 3:      # Int64List { repeated int64 value = 1 [packed] }
            vals = []
            for n_, wt, x in _fields(v):
                if n_ != 1:
                    continue
                if wt == 0:
                    vals.append(x)
                else:
                    j = 0
                    while j < len(x):
                        val, j = _varint(x, j)
                        vals.append(val)
            return np.array([u - (1 << 64) if u >= (1 << 63) else u for u in vals], dtype=np.int64)
        if num == 1:      # BytesList
            return [bytes(x) for n_, _, x in _fields(v) if n_ == 1]
    return None


def parse_example(record: bytes) -> dict:
    """Serialized tf.train.Example -> {feature name: array}."""
    out = {}
    for num, _, feats in _fields(memoryview(record)):
        if num != 1:                                   # Example.features
            continue
        for n2, _, entry in _fields(feats):            # Features.feature map entries
            if n2 != 1:
                continue
            key = val = None
            for n3, _, x in _fields(entry):
                if n3 == 1:
                    key = bytes(x).decode("utf-8")
                elif n3 == 2:
                    val = _feature(x)
            out[key] = val
    return out


def read_records(path, check_crc=True):
    with open(path, "rb") as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise ValueError("%s: truncated record header" % path)
            (length,), (lcrc,) = struct.unpack("<Q", head[:8]), struct.unpack("<I", head[8:])
            if check_crc and masked_crc(head[:8]) != lcrc:
                raise ValueError("%s: corrupt record length" % path)
            data = f.read(length)
            (dcrc,) = struct.unpack("<I", f.read(4))
            if len(data) < length or (check_crc and masked_crc(data) != dcrc):
                raise ValueError("%s: corrupt record" % path)
            yield data


def read_samples(path, check_crc=False):
    """Yield ``(audio [T] float32, mel [frames, mels] float32, speaker_id)`` like ``Dataset._load_sample``
    (dataset.py:47-68) before its crop.  The payload CRC of multi-megabyte records is skipped by default
    (a pure-Python CRC is slow); the framing is still validated by the lengths."""
    for rec in read_records(path, check_crc=check_crc):
        ex = parse_example(rec)
        n = int(ex["audio_len"][0])
        shape = tuple(int(v) for v in ex["mel_shape"])
        spk = int(ex["speaker_id"][0]) if ex.get("speaker_id") is not None else 0
        yield ex["audio"][:n].astype(np.float32), ex["mel"].reshape(shape).astype(np.float32), spk


# ---------------------------------------------------------------- writer (mirrors tfrecord.py:17-88)
def _enc_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(num, payload):
    return _enc_varint((num << 3) | 2) + _enc_varint(len(payload)) + payload


def serialize_example(audio, mel, speaker_id=None) -> bytes:
    def floats(a):
        return _ld(2, _ld(1, np.ascontiguousarray(a, "<f4").tobytes()))

    def ints(a):
        return _ld(3, _ld(1, b"".join(_enc_varint(int(v)) for v in a)))

    feats = {"audio": floats(audio), "audio_len": ints([len(audio)]), "mel_shape": ints(mel.shape), "mel": floats(mel.reshape(-1))}
    if speaker_id is not None:
        feats["speaker_id"] = ints([speaker_id])
    body = b"".join(_ld(1, _ld(1, k.encode()) + _ld(2, v)) for k, v in feats.items())
    return _ld(1, body)


def write_records(path, records):
    with open(path, "wb") as f:
        for data in records:
            head = struct.pack("<Q", len(data))
            f.write(head + struct.pack("<I", masked_crc(head)) + data + struct.pack("<I", masked_crc(data)))


class TFRecordCreator:
    """``train.txt`` (+ ``audios/``, ``mels/``) -> ``train.tfrecord`` / ``test.tfrecord`` (tfrecord.py:76-88)."""

    def __init__(self, metadata_filename, hparams):
        self._hparams = hparams
        self._metadata_filename = metadata_filename
        self._basedir = os.path.dirname(metadata_filename)

    def _records(self, meta):
        for audio_filename, mel_filename, _, speaker_id, _ in meta:
            audio = np.load(os.path.join(self._basedir, "audios", audio_filename))
            mel = np.load(os.path.join(self._basedir, "mels", mel_filename))
            yield serialize_example(audio, mel, int(speaker_id) if self._hparams.gin_channels > 0 else None)

    def create_tfrecords(self):
        from sklearn.model_selection import train_test_split
        with open(self._metadata_filename, encoding="utf-8") as f:
            metadata = [line.strip().split("|") for line in f if line.strip()]
        idx = np.arange(len(metadata))
        tr, te = train_test_split(idx, test_size=self._hparams.test_size, random_state=self._hparams.split_random_state)
        write_records(os.path.join(self._basedir, "train.tfrecord"), self._records([metadata[i] for i in tr]))
        write_records(os.path.join(self._basedir, "test.tfrecord"), self._records([metadata[i] for i in te]))
