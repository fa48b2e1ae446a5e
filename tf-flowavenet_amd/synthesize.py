"""``synthesize`` CLI - same flags and file contract as the reference's synthesize.py:51-63:

    python -m tf_flowavenet_amd.synthesize --saved_dir logs/pretrained/ --mels_dir mels/ --output_dir output/

``mels_dir/*.npy`` float32 [F, num_mels] in [0,1]  ->  ``output_dir/<name>.wav`` (16-bit mono PCM at
``hparams.sample_rate``).  Differences: the checkpoint is this package's own format (``*.npz`` /
``*.safetensors`` holding the parameter names of ``weights.param_shapes``, or the reference's own variable
names ``vocoder/FloWaveNet/...:0`` as dumped from a TF checkpoint - ``weights.from_reference_names``; the TF
tensor-bundle format itself is out of scope), the wav writer is the stdlib ``wave`` module (librosa is not a dependency), ``z`` is
seedable (``--seed``; TF's Philox stream cannot be reproduced), and mels of equal length are
batched into one launch.
"""
from __future__ import annotations

import argparse
import glob
import os
import wave

import numpy as np


def load_checkpoint(saved_dir):
    """Latest readable ``*.npz`` / ``*.safetensors`` in ``saved_dir`` -> dict name -> ndarray.  ``...ckpt-<step>.npz``
    files (train.py's) are ordered by step number, anything else by modification time; a file that cannot be read
    (e.g. truncated by a crash) is skipped with a message."""
    import re
    files = glob.glob(os.path.join(saved_dir, "*.npz")) + glob.glob(os.path.join(saved_dir, "*.safetensors"))
    if not files:
        raise FileNotFoundError("no *.npz / *.safetensors checkpoint in %r" % saved_dir)

    def order(path):
        m = re.search(r"ckpt-(\d+)\.npz$", path)
        return (1, int(m.group(1)), 0.0) if m else (0, 0, os.path.getmtime(path))

    from .weights import from_reference_names
    last_error = None
    for path in sorted(files, key=order, reverse=True):
        try:
            if path.endswith(".npz"):
                with np.load(path) as f:
                    raw = {k: f[k] for k in f.files}
            else:
                from safetensors.numpy import load_file
                raw = load_file(path)
        except Exception as e:
            print("Skipping unreadable checkpoint {} ({}: {})".format(path, type(e).__name__, e))
            last_error = e
            continue
        print("Loading checkpoint {}".format(path))
        return from_reference_names(raw)
    raise FileNotFoundError("no readable checkpoint in %r (last error: %s)" % (saved_dir, last_error))


def max_clips_per_call(hparams, t):
    """How many clips of t samples one ``reverse`` call may take: every activation buffer is addressed with 32-bit
    offsets below 2 GiB (csrc/api.hip check_model: n_layer * (B T / 2) * 512 and B * T * num_mels bytes)."""
    lim = min(((1 << 31) - 1) // (hparams.n_layer * 256), ((1 << 31) - 1) // hparams.num_mels)
    return int((lim - 1) // t)


def write_wav(path, audio, sample_rate):
    pcm = np.clip(np.asarray(audio, dtype=np.float64), -1.0, 1.0)
    pcm = (pcm * 32767.0).round().astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(int(sample_rate))
        w.writeframes(pcm.tobytes())


def synthesize(args, hparams, model=None):
    import torch
    from .model import FloWaveNet
    if model is None:
        model = FloWaveNet(hparams).load_params(load_checkpoint(args.saved_dir))
    os.makedirs(args.output_dir, exist_ok=True)
    names = sorted(f for f in os.listdir(args.mels_dir) if f.endswith(".npy"))
    mels = {n: np.load(os.path.join(args.mels_dir, n)).astype(np.float32) for n in names}
    by_len = {}
    for n in names:
        by_len.setdefault(mels[n].shape[0], []).append(n)
    gen = torch.Generator(device="cpu").manual_seed(int(args.seed))
    hop = hparams.hop_size
    align = max(1, (1 << hparams.n_block) // np.gcd(1 << hparams.n_block, hop))
    for frames, group in sorted(by_len.items()):
        pad = (-frames) % align                      # T must divide by 2^n_block (model.py:226)
        t = (frames + pad) * hop
        per_call = min(int(args.batch), max_clips_per_call(hparams, t))      # long utterances: fewer clips per launch
        if per_call < 1:
            raise ValueError("an utterance of %d samples exceeds what one call can address (%d samples): split the mel"
                             % (t, max_clips_per_call(hparams, 1)))
        for i in range(0, len(group), per_call):
            chunk = group[i:i + per_call]
            c = np.stack([np.pad(mels[n], ((0, pad), (0, 0)), mode="edge") for n in chunk])
            z = torch.randn(len(chunk), t, 1, generator=gen) * hparams.temp     # synthesize.py:14
            wav = model.reverse(z.cuda(), torch.from_numpy(c).cuda()).squeeze(-1).cpu().numpy()
            for n, w in zip(chunk, wav):
                write_wav(os.path.join(args.output_dir, n[:-4] + ".wav"), w[:frames * hop], hparams.sample_rate)
    return names


def main(argv=None):
    from .hparams import hparams
    parser = argparse.ArgumentParser()
    parser.add_argument("--saved_dir", default="logs/pretrained/", help="Folder with model checkpoint")
    parser.add_argument("--mels_dir", default="mels/", help="folder to contain mels to synthesize audio from using the model")
    parser.add_argument("--output_dir", default="output/", help="folder to contain synthesized audio files")
    parser.add_argument("--seed", type=int, default=hparams.tf_random_seed, help="seed of the latent z")
    parser.add_argument("--batch", type=int, default=8, help="equal-length mels per launch")
    args = parser.parse_args(argv)
    synthesize(args, hparams)


if __name__ == "__main__":
    main()
