"""Mel front-end (preprocessing.py:49-97): the oracle's pins, the host-side
filterbank against the oracle's independent restatement, and (GPU) fwn_mel_spectrogram / the
preprocessing surface against the oracle."""
import os
import wave

import numpy as np
import pytest

from oracle import mel_np as onp
from tf_flowavenet_amd import preprocessing as P
from tf_flowavenet_amd.hparams import default_hparams, hparams8000


def _wav(seed, n, scale=0.1):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 22050.0
    return (scale * rng.standard_normal(n) + 0.3 * np.sin(2 * np.pi * 440.0 * t) * np.hanning(n)).astype(np.float32)


def test_oracle_stft_matches_scipy():
    import scipy.signal as ss
    y = _wav(0, 6000).astype(np.float64)
    for n_fft, hop in ((1024, 256), (512, 96)):
        yp = np.pad(y, (n_fft // 2, n_fft // 2), mode="reflect")
        _, _, z = ss.stft(yp, window="hann", nperseg=n_fft, noverlap=n_fft - hop, boundary=None, padded=False)
        want = (np.abs(z.T) * (n_fft / 2)) ** 2        # scipy scales by 1 / sum(window) = 2 / n_fft
        got = onp.stft_power(y, n_fft, hop)
        assert got.shape == (1 + len(y) // hop, n_fft // 2 + 1)
        np.testing.assert_allclose(got[:want.shape[0]], want, rtol=1e-9, atol=1e-12 * want.max())


def test_oracle_filterbank_closed_form_properties():
    hp = default_hparams()
    fb = onp.mel_filterbank(hp.sample_rate, hp.n_fft, hp.num_mels, hp.fmin, hp.fmax)
    assert fb.shape == (80, 513) and (fb >= 0).all()
    freqs = np.linspace(0, hp.sample_rate / 2, 513)
    assert fb[:, freqs < hp.fmin].sum() == 0 and fb[:, freqs > hp.fmax].sum() == 0
    # mel scale: linear below 1 kHz at 200/3 Hz per mel, then log with 27 steps per factor 6.4
    np.testing.assert_allclose(onp.hz_to_mel([0.0, 1000.0, 6400.0]), [0.0, 15.0, 42.0], atol=1e-12)
    np.testing.assert_allclose(onp.mel_to_hz(onp.hz_to_mel([125.0, 999.0, 1001.0, 7600.0])), [125.0, 999.0, 1001.0, 7600.0])
    # Slaney normalisation: a continuous triangle of height 2/(hi-lo) has unit area
    edges = onp.mel_to_hz(np.linspace(onp.hz_to_mel(hp.fmin), onp.hz_to_mel(hp.fmax), 82))
    wide = (edges[2:] - edges[:-2]) > 8 * freqs[1]
    area = fb.sum(axis=1) * freqs[1]
    np.testing.assert_allclose(area[wide], 1.0, atol=0.03)
    # each filter peaks between its neighbours' edges
    peak = freqs[fb.argmax(axis=1)]
    assert (np.diff(peak) >= 0).all() and (np.abs(peak - edges[1:-1]) <= freqs[1]).all()


def test_oracle_process_utterance_shapes_and_range():
    hp = default_hparams()
    y = _wav(1, 22050 + 77)
    audio, mel = onp.process_utterance(y, hp)
    n = 1 + len(y) // hp.hop_size
    assert mel.shape == (n, 80) and audio.shape == (n * hp.hop_size,)
    assert mel.min() >= 0 and mel.max() <= 1 and abs(np.abs(audio).max() - hp.rescaling_max) < 1e-12
    # a pure tone lights up the filter that contains it
    t = np.arange(8192) / hp.sample_rate
    mel_tone = onp.melspectrogram(0.5 * np.sin(2 * np.pi * 1000.0 * t), hp)
    edges = onp.mel_to_hz(np.linspace(onp.hz_to_mel(hp.fmin), onp.hz_to_mel(hp.fmax), 82))
    assert abs(edges[1:-1][mel_tone[8:-8].mean(axis=0).argmax()] - 1000.0) < 40.0


@pytest.mark.parametrize("hp", [default_hparams(), hparams8000()], ids=["22k", "8k"])
def test_host_filterbank_and_window_match_oracle(hp):
    fb = P.mel_filterbank(hp)
    want = onp.mel_filterbank(hp.sample_rate, hp.n_fft, hp.num_mels, hp.fmin, hp.fmax)
    assert fb.dtype == np.float32 and fb.shape == want.shape
    np.testing.assert_allclose(fb, want, rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(P.hann_window(hp.n_fft), onp.hann_periodic(hp.n_fft), atol=1e-7)


def test_read_wav_round_trip(tmp_path):
    pcm = (np.random.default_rng(3).uniform(-1, 1, 4000) * 32767).astype("<i2")
    path = str(tmp_path / "a.wav")
    with wave.open(path, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(22050); w.writeframes(pcm.tobytes())
    np.testing.assert_array_equal(P.read_wav(path, 22050), pcm.astype(np.float32) / 32768.0)
    with pytest.raises(ValueError):
        P.read_wav(path, 16000)


# ------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("hp,b,t", [(default_hparams(), 3, 16128), (default_hparams(), 1, 7001), (hparams8000(), 2, 2320)],
                         ids=["22k_b3", "22k_ragged", "8k"])
def test_mel_kernel_matches_oracle(hp, b, t):
    """fwn_mel_spectrogram vs the fp64 oracle: normalised log-mel in [0, 1], fp32 DFT sums of
    n_fft terms -> 2e-4 absolute (0.02 dB) away from the 1e-4 power floor."""
    import torch
    wav = np.stack([_wav(10 + i, t) for i in range(b)])
    got = P.MelSpectrogram(hp)(torch.from_numpy(wav)).cpu().numpy()
    want = np.stack([onp.melspectrogram(w.astype(np.float64), hp) for w in wav])
    assert got.shape == want.shape == (b, 1 + t // hp.hop_size, hp.num_mels)
    assert np.abs(got - want).max() < 2e-4, np.abs(got - want).max()
    # edge cases: a clip of zeros sits on the floor of the log
    z = P.MelSpectrogram(hp)(torch.zeros(1, t)).cpu().numpy()
    floor = np.clip((20 * np.log10(1e-4) - hp.ref_level_db - hp.min_level_db) / -hp.min_level_db, 0, 1)
    np.testing.assert_allclose(z, floor, atol=1e-6)


@pytest.mark.gpu
def test_preprocess_file_contract(tmp_path):
    hp = default_hparams()
    book = tmp_path / "in" / "book1"
    os.makedirs(book / "wavs")
    lines = []
    for i, n in enumerate((9000, 12345)):
        pcm = (np.clip(_wav(20 + i, n), -1, 1) * 32767).astype("<i2")
        with wave.open(str(book / "wavs" / ("u%d.wav" % i)), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(hp.sample_rate); w.writeframes(pcm.tobytes())
        lines.append("u%d|raw text %d|normalised text %d" % (i, i, i))
    (book / "metadata.csv").write_text("\n".join(lines), encoding="utf-8")
    out = tmp_path / "out"
    meta = P.preprocess(str(tmp_path / "in"), str(out), hp)
    rows = (out / "train.txt").read_text(encoding="utf-8").strip().split("\n")
    assert len(meta) == len(rows) == 2 and rows[0].split("|")[0] == "dataset-audio-00001.npy"
    for r, n in zip(rows, (9000, 12345)):
        a_name, m_name, steps, spk, text = r.split("|")
        audio, mel = np.load(out / "audios" / a_name), np.load(out / "mels" / m_name)
        assert audio.dtype == mel.dtype == np.float32 and int(steps) == len(audio) == mel.shape[0] * hp.hop_size
        assert mel.shape == (1 + n // hp.hop_size, hp.num_mels)
        src = book / "wavs" / ("u%d.wav" % (int(a_name[-9:-4]) - 1))
        a0, m0 = onp.process_utterance(P.read_wav(str(src), hp.sample_rate), hp)
        np.testing.assert_allclose(audio, a0, atol=1e-6)
        assert np.abs(mel - m0).max() < 2e-4
