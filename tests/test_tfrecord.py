"""TFRecord container + tf.train.Example codec without TensorFlow: known answers
for the checksum, a cross-check against the protobuf runtime with the Example schema declared on the
fly, and the reference's file layout end to end."""
import os
import struct

import numpy as np
import pytest

from tf_flowavenet_amd import tfrecord as T
from conftest import small_hparams


def test_crc32c_known_answers_and_tf_mask():
    assert T.crc32c(b"123456789") == 0xE3069283          # the standard CRC-32C check value
    assert T.crc32c(b"") == 0
    assert T.crc32c(bytes(32)) == 0x8A9136AA             # RFC 3720 B.4: 32 bytes of zeros
    c = T.crc32c(b"abc")
    assert T.masked_crc(b"abc") == ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _example_classes():
    """tf.train.Example / Features / Feature / *List declared with the protobuf runtime (field numbers of
    tensorflow/core/example/{example,feature}.proto)."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    f = descriptor_pb2.FileDescriptorProto(name="fwn_example_test.proto", package="fwn_t", syntax="proto3")

    def msg(name):
        m = f.message_type.add()
        m.name = name
        return m

    def field(m, name, num, typ, label=1, type_name=None, packed=None):
        fd = m.field.add()
        fd.name, fd.number, fd.type, fd.label = name, num, typ, label
        if type_name:
            fd.type_name = type_name
        return fd

    D = descriptor_pb2.FieldDescriptorProto
    field(msg("BytesList"), "value", 1, D.TYPE_BYTES, D.LABEL_REPEATED)
    field(msg("FloatList"), "value", 1, D.TYPE_FLOAT, D.LABEL_REPEATED)
    field(msg("Int64List"), "value", 1, D.TYPE_INT64, D.LABEL_REPEATED)
    feat = msg("Feature")
    for n, (nm, tn) in enumerate((("bytes_list", "BytesList"), ("float_list", "FloatList"), ("int64_list", "Int64List")), 1):
        field(feat, nm, n, D.TYPE_MESSAGE, type_name=".fwn_t." + tn)
    feats = msg("Features")
    entry = feats.nested_type.add()
    entry.name = "FeatureEntry"
    entry.options.map_entry = True
    field(entry, "key", 1, D.TYPE_STRING)
    field(entry, "value", 2, D.TYPE_MESSAGE, type_name=".fwn_t.Feature")
    field(feats, "feature", 1, D.TYPE_MESSAGE, D.LABEL_REPEATED, type_name=".fwn_t.Features.FeatureEntry")
    field(msg("Example"), "features", 1, D.TYPE_MESSAGE, type_name=".fwn_t.Features")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(f)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("fwn_t.Example"))


def test_example_codec_agrees_with_the_protobuf_runtime():
    Example = _example_classes()
    rng = np.random.default_rng(0)
    audio, mel = rng.standard_normal(777).astype(np.float32), rng.random((13, 80)).astype(np.float32)
    # protobuf runtime -> our parser
    ex = Example()
    ex.features.feature["audio"].float_list.value.extend(audio.tolist())
    ex.features.feature["audio_len"].int64_list.value.append(len(audio))
    ex.features.feature["mel_shape"].int64_list.value.extend(mel.shape)
    ex.features.feature["mel"].float_list.value.extend(mel.reshape(-1).tolist())
    ex.features.feature["speaker_id"].int64_list.value.append(-3)
    got = T.parse_example(ex.SerializeToString())
    np.testing.assert_array_equal(got["audio"], audio)
    np.testing.assert_array_equal(got["mel"].reshape(13, 80), mel)
    assert got["audio_len"].tolist() == [777] and got["mel_shape"].tolist() == [13, 80] and got["speaker_id"].tolist() == [-3]
    # our writer -> protobuf runtime
    back = Example()
    back.ParseFromString(T.serialize_example(audio, mel, speaker_id=5))
    np.testing.assert_array_equal(np.array(back.features.feature["audio"].float_list.value, np.float32), audio)
    assert list(back.features.feature["mel_shape"].int64_list.value) == [13, 80]
    assert list(back.features.feature["speaker_id"].int64_list.value) == [5]


def test_container_framing_and_corruption_detection(tmp_path):
    path = str(tmp_path / "a.tfrecord")
    recs = [b"", b"hello", bytes(range(256)) * 3]
    T.write_records(path, recs)
    assert list(T.read_records(path)) == recs
    raw = bytearray(open(path, "rb").read())
    assert struct.unpack("<Q", raw[:8])[0] == 0                     # first record: empty payload
    raw[12 + 4 + 12 + 2] ^= 0xFF                                     # flip a payload byte of the second record
    open(path, "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        list(T.read_records(path))


def test_creator_and_dataset_round_trip(tmp_path):
    """preprocess-style directory -> train/test.tfrecord (the reference's split) -> crops like dataset.py."""
    from tf_flowavenet_amd import train as TL
    hp = small_hparams(hop_size=16, max_time_steps=64, batch_size=3, test_size=2, num_mels=8)
    base = tmp_path / "training_data"
    os.makedirs(base / "audios"); os.makedirs(base / "mels")
    rng = np.random.default_rng(1)
    lines = []
    for i in range(6):
        frames = 6 + i
        np.save(base / "audios" / ("a%d.npy" % i), rng.standard_normal(frames * 16).astype(np.float32))
        np.save(base / "mels" / ("m%d.npy" % i), rng.random((frames, 8)).astype(np.float32))
        lines.append("a%d.npy|m%d.npy|%d|0|text" % (i, i, frames * 16))
    (base / "train.txt").write_text("\n".join(lines), encoding="utf-8")
    T.TFRecordCreator(str(base / "train.txt"), hp).create_tfrecords()
    tr = list(T.read_samples(str(base / "train.tfrecord"), check_crc=True))
    te = list(T.read_samples(str(base / "test.tfrecord"), check_crc=True))
    assert len(tr) == 4 and len(te) == 2
    for audio, mel, spk in tr + te:
        assert audio.shape[0] == mel.shape[0] * 16 and mel.shape[1] == 8 and spk == 0
    ds = TL.Dataset.from_tfrecords(str(base / "train.tfrecord"), str(base / "test.tfrecord"), hp, seed=0)
    mels, audios = ds.next_train()
    assert mels.shape == (3, 4, 8) and audios.shape == (3, 64)
    # every crop is a window of one of the stored utterances, audio aligned with its mel frames
    ok = 0
    for k in range(3):
        for audio, mel, _ in tr:
            for s in range(mel.shape[0] - 4):
                if np.array_equal(mel[s:s + 4], mels[k]) and np.array_equal(audio[s * 16:s * 16 + 64], audios[k]):
                    ok += 1
    assert ok >= 3
