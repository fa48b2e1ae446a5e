"""TEST INFRASTRUCTURE ONLY - fp64 NumPy restatement of the reference's mel front-end.

Follows ``preprocessing.py:49-97`` (``_process_utterance``).  The arithmetic itself lives in the
third-party ``librosa`` (``requirements.txt:4``, version unpinned, absent from this image and from
/root/reference): ``librosa.feature.melspectrogram`` = ``filters.mel(...) @ |stft(...)|**2`` with its
published defaults, restated below (Slaney mel scale + Slaney area normalisation, centred STFT with
reflect padding, periodic Hann window, power 2).

PARITY UNPINNED: the reference holds no test, fixture or stored spectrogram for this row and librosa
cannot be imported here.  Pins used instead (tests/test_mel.py): the STFT against
``scipy.signal.stft``; filterbank closed-form properties.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import numpy as np


def hz_to_mel(f):
    """librosa.hz_to_mel(htk=False): linear below 1 kHz (200/3 Hz per mel), log above."""
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm=1) -> [n_mels, 1+n_fft//2]."""
    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return w * enorm[:, None]


def hann_periodic(n):
    """scipy.signal.get_window('hann', n, fftbins=True)."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def stft_power(y, n_fft, hop):
    """|librosa.stft(y, n_fft, hop, center=True, pad_mode='reflect', window='hann')|**2 -> [frames, bins]."""
    y = np.asarray(y, dtype=np.float64)
    yp = np.pad(y, (n_fft // 2, n_fft // 2), mode="reflect")
    frames = 1 + (len(yp) - n_fft) // hop
    win = hann_periodic(n_fft)
    idx = np.arange(n_fft)[None, :] + hop * np.arange(frames)[:, None]
    spec = np.fft.rfft(yp[idx] * win[None, :], axis=1)
    return spec.real ** 2 + spec.imag ** 2


def melspectrogram(y, hp):
    """preprocessing.py:58-69: [frames, num_mels] in [0, 1]."""
    fb = mel_filterbank(hp.sample_rate, hp.n_fft, hp.num_mels, hp.fmin, hp.fmax)
    mel = stft_power(y, hp.n_fft, hp.hop_size) @ fb.T                        # (N, D), :58-64
    mel = 20.0 * np.log10(np.maximum(1e-4, mel)) - hp.ref_level_db           # :67 (20 log10 of a POWER: kept)
    return np.clip((mel - hp.min_level_db) / (-hp.min_level_db), 0.0, 1.0)   # :68


def process_utterance(wav, hp):
    """preprocessing.py:49-89 without the file I/O: (audio [N*hop], mel [N, num_mels])."""
    wav = np.asarray(wav, dtype=np.float64)
    wav = wav / np.abs(wav).max() * hp.rescaling_max                         # :52
    mel = melspectrogram(wav, hp)
    pad = (len(wav) // hp.hop_size + 1) * hp.hop_size - len(wav)             # :71
    out = np.pad(wav, (pad // 2, pad // 2 + pad % 2))                        # :72-76
    n = mel.shape[0]
    assert len(out) >= n * hp.hop_size                                       # :78
    return out[:n * hp.hop_size], mel                                        # :83
