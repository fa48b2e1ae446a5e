"""Mel front-end - the reference's ``preprocessing.py`` on MI355X.

Same names, file layout and value ranges as the reference (``preprocessing.py:13-106``):
``preprocess(in_dir, out_dir)`` writes ``audios/dataset-audio-%05d.npy`` (float32 samples, length a
multiple of ``hop_size``), ``mels/dataset-mel-%05d.npy`` (float32 ``[N, num_mels]`` in [0, 1]) and
``train.txt`` (``audio|mel|timesteps|speaker|text``); ``_process_utterance`` is the per-clip step.
The spectrogram (``librosa.feature.melspectrogram`` in the reference, ``:58-64``) runs in
``libfwn.so`` (``fwn_mel_spectrogram``); there is no CPU fallback.

Differences (deliberate): wavs are read with the stdlib ``wave`` module (16-bit PCM at
``hparams.sample_rate``; librosa's resampling loader is not a dependency), the TFRecord writer
(``tfrecord.py``) is out of scope, and clips are processed on one device stream instead of a
process pool.
"""
from __future__ import annotations

import os
import wave

import numpy as np

from . import _lib


def _mel_points(fmin, fmax, n):
    """n frequencies evenly spaced on the Slaney mel scale (linear to 1 kHz, log above)."""
    lin, knee, step = 200.0 / 3.0, 1000.0, np.log(6.4) / 27.0

    def to_mel(f):
        return f / lin if f < knee else knee / lin + np.log(f / knee) / step

    m = np.linspace(to_mel(float(fmin)), to_mel(float(fmax)), n)
    return np.where(m < knee / lin, m * lin, knee * np.exp(step * (m - knee / lin)))


def mel_filterbank(hparams):
    """Triangular, area-normalised (Slaney) filters: float32 [num_mels, n_fft/2 + 1]."""
    nb = hparams.n_fft // 2 + 1
    edges = _mel_points(hparams.fmin, hparams.fmax, hparams.num_mels + 2)
    freqs = np.arange(nb) * (hparams.sample_rate / float(hparams.n_fft))
    lo, mid, hi = edges[:-2, None], edges[1:-1, None], edges[2:, None]
    tri = np.minimum((freqs[None, :] - lo) / (mid - lo), (hi - freqs[None, :]) / (hi - mid))
    return (np.maximum(tri, 0.0) * (2.0 / (hi - lo))).astype(np.float32)


def hann_window(n):
    """Periodic Hann window (scipy ``get_window('hann', n, fftbins=True)``), float32."""
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)).astype(np.float32)


class MelSpectrogram:
    """``wav [B, T] -> mel [B, 1 + T // hop, num_mels]`` on the device (preprocessing.py:58-69)."""

    def __init__(self, hparams, device="cuda"):
        import torch
        self._hp = hparams
        self._lib = _lib.load()          # fails loudly when libfwn.so is missing
        self._device = torch.device(device)
        self._win = torch.from_numpy(hann_window(hparams.n_fft)).to(self._device)
        self._fb = torch.from_numpy(mel_filterbank(hparams)).to(self._device)

    def __call__(self, wav):
        import torch
        hp = self._hp
        w = torch.as_tensor(wav)
        squeeze = w.dim() == 1
        if squeeze:
            w = w[None]
        if w.dim() != 2:
            raise ValueError("wav must have shape [T] or [B, T], got %r" % (tuple(w.shape),))
        w = w.to(device=self._device, dtype=torch.float32).contiguous()
        b, t = int(w.shape[0]), int(w.shape[1])
        frames = 1 + t // hp.hop_size
        mel = torch.empty(b, frames, hp.num_mels, dtype=torch.float32, device=self._device)
        rc = self._lib.fwn_mel_spectrogram(w.data_ptr(), b, t, self._win.data_ptr(), self._fb.data_ptr(),
                                           hp.n_fft, hp.hop_size, hp.num_mels, float(hp.ref_level_db),
                                           float(hp.min_level_db), mel.data_ptr(),
                                           torch.cuda.current_stream(self._device).cuda_stream)
        _lib.check(rc, "fwn_mel_spectrogram")
        return mel[0] if squeeze else mel


def _process_utterance(wav, hparams, mel_fn=None):
    """One clip (preprocessing.py:49-89, minus the file I/O): float samples -> (audio, mel).

    audio: float32 [N * hop_size], zero padded / trimmed to the mel's frame count (:71-84);
    mel: float32 [N, num_mels] in [0, 1]."""
    wav = np.asarray(wav, dtype=np.float32)
    peak = float(np.abs(wav).max())
    if not peak > 0.0:
        raise ValueError("silent clip: the reference divides by max|wav| (preprocessing.py:52)")
    wav = wav / peak * hparams.rescaling_max
    mel_fn = mel_fn or MelSpectrogram(hparams)
    mel = mel_fn(wav).cpu().numpy()
    hop = hparams.hop_size
    pad = (len(wav) // hop + 1) * hop - len(wav)
    out = np.pad(wav, (pad // 2, pad // 2 + pad % 2))
    n = mel.shape[0]
    assert len(out) >= n * hop
    return out[:n * hop].astype(np.float32), mel.astype(np.float32)


def read_wav(path, sample_rate):
    """16-bit PCM wav at ``sample_rate`` -> float32 mono in [-1, 1)."""
    with wave.open(path, "rb") as w:
        if w.getsampwidth() != 2:
            raise ValueError("%s: only 16-bit PCM is supported" % path)
        if w.getframerate() != sample_rate:
            raise ValueError("%s: sample rate %d != hparams.sample_rate %d (resample first)"
                             % (path, w.getframerate(), sample_rate))
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").astype(np.float32) / 32768.0
        if w.getnchannels() > 1:
            pcm = pcm.reshape(-1, w.getnchannels()).mean(axis=1)
    return pcm


def build_from_path(in_dir, out_dir, hparams, num_workers=1):
    """``in_dir/<book>/metadata.csv`` + ``wavs/<id>.wav`` (single speaker, preprocessing.py:30-45)."""
    del num_workers     # one device stream does the work of the reference's process pool
    mel_fn = MelSpectrogram(hparams)
    books = sorted(os.path.join(in_dir, f) for f in os.listdir(in_dir) if os.path.isdir(os.path.join(in_dir, f)))
    metadata, index = [], 1
    for book in books:
        with open(os.path.join(book, "metadata.csv"), encoding="utf-8") as f:
            lines = f.read().strip().split("\n")
        for line in lines:
            parts = line.strip().split("|")
            text = parts[2] if len(parts) > 2 else ""
            wav = read_wav(os.path.join(book, "wavs", "%s.wav" % parts[0]), hparams.sample_rate)
            audio, mel = _process_utterance(wav, hparams, mel_fn)
            audio_filename = "dataset-audio-%05d.npy" % index
            mel_filename = "dataset-mel-%05d.npy" % index
            np.save(os.path.join(out_dir, "audios", audio_filename), audio, allow_pickle=False)
            np.save(os.path.join(out_dir, "mels", mel_filename), mel, allow_pickle=False)
            metadata.append((audio_filename, mel_filename, len(audio), 0, text))
            index += 1
    return metadata


def write_metadata(metadata, out_dir, hparams):
    with open(os.path.join(out_dir, "train.txt"), "w", encoding="utf-8") as f:
        for m in metadata:
            f.write("|".join(str(x) for x in m) + "\n")
    frames = sum(m[2] for m in metadata)
    print("Wrote %d utterances, %d time steps (%.2f hours)"
          % (len(metadata), frames, frames / hparams.sample_rate / 3600))
    print("Max input length:  %d" % max(len(m[4]) for m in metadata))
    print("Max output length: %d" % max(m[2] for m in metadata))


def preprocess(in_dir, out_dir, hparams=None, num_workers=1):
    if hparams is None:
        from .hparams import hparams as hparams_default
        hparams = hparams_default
    for sub in ("", "audios", "mels"):
        os.makedirs(os.path.join(out_dir, sub), exist_ok=True)
    metadata = build_from_path(in_dir, out_dir, hparams, num_workers)
    write_metadata(metadata, out_dir, hparams)
    return metadata


if __name__ == "__main__":
    import argparse
    parser = argparse.ArgumentParser(description="Preprocessing",
                                     formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--in_dir", "-i", type=str, default="./", help="In Directory")
    parser.add_argument("--out_dir", "-o", type=str, default="./", help="Out Directory")
    args = parser.parse_args()
    preprocess(args.in_dir, args.out_dir)
