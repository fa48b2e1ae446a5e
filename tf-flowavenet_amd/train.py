"""``train`` CLI - the reference's training loop (train.py:153-283) on MI355X.

    python -m tf_flowavenet_amd.train --base_dir data/ --input training_data/train.txt
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        -m tf_flowavenet_amd.train --base_dir data/            # one process per GPU, RCCL gradient all-reduce

Same flags as the reference (``--base_dir --input --restore --summary_interval --checkpoint_interval
--eval_interval --train_steps``) plus ``--log_dir`` / ``--seed``.  Input is the output of
``preprocessing.preprocess``: ``train.txt`` with ``audios/*.npy`` and ``mels/*.npy`` beside it.

Differences (deliberate): the TFRecord round trip (tfrecord.py, dataset.py:20-44) is skipped - the
``.npy`` files are read directly and cropped like ``Dataset._load_sample`` (dataset.py:70-76)
(``--input training_data/train.tfrecord`` reads the reference's TFRecords instead, ``tfrecord.py``); the
train / test split is the reference's ``train_test_split(test_size, random_state)`` (tfrecord.py:81-82);
summaries are JSON lines (``<log_dir>/train/summary.jsonl``, ``test/summary.jsonl``) and evaluation
audio is written as wav files instead of TensorBoard events; checkpoints are ``.npz`` files that
``synthesize.py`` loads (parameters in the reference's layouts, plus the Adam slots and global step).
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import collections
import time

import numpy as np


class Dataset:
    """Random fixed-length crops of (mel, audio) pairs (dataset.py:47-85)."""
    # memory-mapped utterances kept open per rank (_load).  Every numpy memmap pins a dup'd file descriptor, two per
    # utterance: 64 pairs = 128 descriptors, far below the usual soft RLIMIT_NOFILE of 1024 that the GPU runtime, RCCL
    # and the log files also draw on.
    MAX_OPEN = 64

    def __init__(self, metadata_path, hparams, seed=None, rank=0):
        self._hp = hparams
        self._basedir = os.path.dirname(metadata_path)
        with open(metadata_path, "rt", encoding="utf-8") as f:
            meta = [m.split("|") for m in f.read().strip().split("\n") if m]
        self._frames = hparams.max_time_steps // hparams.hop_size            # dataset.py:13-14
        self._steps = self._frames * hparams.hop_size
        # dataset.py:73 draws the crop start from [0, frames - max_time_frames): needs strictly longer clips
        meta = [m for m in meta if int(m[2]) // hparams.hop_size > self._frames]
        if not meta:
            raise ValueError("no utterance longer than max_time_steps=%d in %s" % (hparams.max_time_steps, metadata_path))
        idx = np.arange(len(meta))
        if len(meta) > hparams.test_size:                                    # tfrecord.py:81-82
            from sklearn.model_selection import train_test_split
            tr, te = train_test_split(idx, test_size=hparams.test_size, random_state=hparams.split_random_state)
        else:
            tr, te = idx, idx
        self.train_meta, self.test_meta = [meta[i] for i in tr], [meta[i] for i in te]
        base = hparams.shuffle_random_seed if seed is None else seed
        self._rng = np.random.RandomState(base + 7919 * rank)
        self._cache, self._lru = {}, collections.OrderedDict()

    @classmethod
    def from_tfrecords(cls, train_path, test_path, hparams, seed=None, rank=0):
        """The reference's own data files (tfrecord.py:76-88): ``train.tfrecord`` / ``test.tfrecord``."""
        from . import tfrecord
        self = cls.__new__(cls)
        self._hp, self._basedir = hparams, os.path.dirname(train_path)
        self._frames = hparams.max_time_steps // hparams.hop_size
        self._steps = self._frames * hparams.hop_size
        self._cache, self._lru = {}, collections.OrderedDict()      # TFRecord samples stay resident (not in the LRU)

        def load(path, tag):
            metas = []
            for k, (audio, mel, spk) in enumerate(tfrecord.read_samples(path)):
                if mel.shape[0] > self._frames:
                    key = "%s-%d" % (tag, k)
                    self._cache[key] = (audio, mel)
                    metas.append([key, key, str(len(audio)), str(spk), ""])
            return metas

        self.train_meta = load(train_path, "train")
        self.test_meta = load(test_path, "test") if test_path and os.path.exists(test_path) else self.train_meta
        if not self.train_meta:
            raise ValueError("no utterance longer than max_time_steps=%d in %s" % (hparams.max_time_steps, train_path))
        base = hparams.shuffle_random_seed if seed is None else seed
        self._rng = np.random.RandomState(base + 7919 * rank)
        return self

    def _load(self, m):
        """Utterances read from ``audios/`` / ``mels/`` are memory-mapped (a crop touches a few pages; nothing of an
        LJSpeech-sized set stays resident per rank), with at most ``MAX_OPEN`` maps kept open (LRU)."""
        hit = self._cache.get(m[0])
        if hit is not None:
            if m[0] in self._lru:
                self._lru.move_to_end(m[0])
            return hit
        pair = (np.load(os.path.join(self._basedir, "audios", m[0]), mmap_mode="r"),
                np.load(os.path.join(self._basedir, "mels", m[1]), mmap_mode="r"))
        self._cache[m[0]] = pair
        self._lru[m[0]] = None
        while len(self._lru) > self.MAX_OPEN:
            old, _ = self._lru.popitem(last=False)
            # Dropping the pair releases the descriptors: a np.memmap holds a buffer export on its mmap object (an explicit
            # mmap.close() raises BufferError while the array exists), so the map and its file descriptor go when CPython
            # drops the LAST reference to the arrays - here, unless a caller kept a view.  _batch() copies its crops out
            # (np.empty + slice assignment) and keeps none; tests/test_train_cli.py bounds the open descriptors.
            del self._cache[old]
        return pair

    def _batch(self, metas):
        hp = self._hp
        mels = np.empty((len(metas), self._frames, hp.num_mels), dtype=np.float32)
        audios = np.empty((len(metas), self._steps), dtype=np.float32)
        for k, m in enumerate(metas):
            audio, mel = self._load(m)
            start = self._rng.randint(0, mel.shape[0] - self._frames)        # dataset.py:73-76
            mels[k] = mel[start:start + self._frames]
            audios[k] = audio[start * hp.hop_size:start * hp.hop_size + self._steps]
            del audio, mel              # the batch holds copies: no view outlives this iteration (see _load's close)
        return mels, audios

    def next_train(self):
        pick = self._rng.randint(0, len(self.train_meta), size=self._hp.batch_size)
        return self._batch([self.train_meta[i] for i in pick])

    def next_test(self):
        pick = self._rng.randint(0, len(self.test_meta), size=self._hp.batch_size)
        return self._batch([self.test_meta[i] for i in pick])

    def eval_sample(self):
        """One utterance, capped at eval_max_time_steps (train.py:118-131)."""
        hp = self._hp
        m = self.test_meta[self._rng.randint(0, len(self.test_meta))]
        audio, mel = self._load(m)
        frames = min(int(hp.eval_max_time_steps // hp.hop_size), mel.shape[0])
        while frames > 1 and (frames * hp.hop_size) % (1 << hp.n_block):      # model.py:226: T % 2^n_block == 0
            frames -= 1
        return np.array(mel[:frames]), np.array(audio[:frames * hp.hop_size])   # copies, not views of the map


def save_checkpoint(path, trainer):
    views = trainer.opt.master_views()
    out = {k: v.detach().cpu().numpy() for k, v in views.items()}
    out["__opt/m"] = trainer.opt.m.cpu().numpy()
    out["__opt/v"] = trainer.opt.v.cpu().numpy()
    out["__opt/global_step"] = np.asarray(trainer.opt.global_step, dtype=np.int64)
    tmp = path + ".tmp"              # outside every restore glob (*.npz): a crash mid-write never shadows a good checkpoint
    with open(tmp, "wb") as f:
        np.savez(f, **out)
        f.flush()
        os.fsync(f.fileno())
    os.replace(tmp, path)


def checkpoint_files(save_dir):
    """``flowavenet_model.ckpt-<step>.npz`` files of save_dir, oldest step first (by step number, not mtime)."""
    import re
    found = []
    for path in glob.glob(os.path.join(save_dir, "flowavenet_model.ckpt-*.npz")):
        m = re.search(r"ckpt-(\d+)\.npz$", path)
        if m:
            found.append((int(m.group(1)), path))
    return [p for _, p in sorted(found)]


def _checkpoint_step(path):
    import re
    return int(re.search(r"ckpt-(\d+)\.npz$", path).group(1))


def restore_checkpoint(save_dir, trainer):
    """Highest-step readable ``flowavenet_model.ckpt-<step>.npz`` -> masters, Adam slots, global step.  Returns the
    step or None.

    Only a file that cannot be READ (truncated by a crash: zip / EOF / OS errors) is skipped, with a message.  A file
    that reads but does not fit this model - a missing parameter, another shape: different hparams - raises: silently
    starting from step 0 would go on to overwrite the run's checkpoints.  In a data-parallel job rank 0 picks the file
    and every rank must load that very step; disagreement (a file one rank cannot read) aborts the job."""
    import zipfile
    import torch
    import torch.distributed as dist
    group = getattr(trainer.opt, "group", None)
    multi = dist.is_available() and dist.is_initialized() and group is not False and dist.get_world_size(group) > 1
    files = list(reversed(checkpoint_files(save_dir)))
    want = None
    if multi:      # the step rank 0 is about to try first; ranks that see other files fail below instead of diverging
        box = [(_checkpoint_step(files[0]) if files else -1) if dist.get_rank(group) == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        want = box[0]
    got = None
    for path in files:
        if want is not None and got is None and _checkpoint_step(path) > want >= 0:
            continue                    # newer than what rank 0 sees (it is still being written there)
        try:
            with np.load(path) as f:
                names = set(f.files)
                arrays = {k: f[k] for k in names}
        except (zipfile.BadZipFile, EOFError, OSError, ValueError) as e:      # I/O and container-format errors only
            print("Skipping unreadable checkpoint {} ({}: {})".format(path, type(e).__name__, e))
            continue                    # (data parallel: if only this rank cannot read it, the agreement check below aborts)
        views = trainer.opt.master_views()
        missing = [k for k in list(views) + ["__opt/m", "__opt/v", "__opt/global_step"] if k not in names]
        if missing:
            raise KeyError("checkpoint {} does not belong to this model: {} entries missing, e.g. {!r} (other hparams?)"
                           .format(path, len(missing), missing[0]))
        for k, v in views.items():
            if int(np.prod(arrays[k].shape)) != v.numel():
                raise ValueError("checkpoint {}: {!r} has shape {}, the model expects {} (other hparams?)"
                                 .format(path, k, tuple(arrays[k].shape), tuple(v.shape)))
        loaded = {k: torch.from_numpy(arrays[k]).reshape(v.shape) for k, v in views.items()}
        m, v_, gs = torch.from_numpy(arrays["__opt/m"]), torch.from_numpy(arrays["__opt/v"]), int(arrays["__opt/global_step"])
        if m.numel() != trainer.opt.m.numel() or v_.numel() != trainer.opt.v.numel():
            raise ValueError("checkpoint {}: Adam slots of {} elements, the model has {}".format(path, m.numel(), trainer.opt.m.numel()))
        got = (path, loaded, m, v_, gs)
        break
    if multi:
        mine = torch.tensor([got[4] if got else -1], dtype=torch.int64, device=trainer.opt.m.device)
        lo, hi = mine.clone(), mine.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
        if int(lo) != int(hi):
            raise RuntimeError("ranks disagree on the checkpoint to restore (steps {} .. {}; this rank: {}): every rank "
                               "must read the same file of {}".format(int(lo), int(hi), int(mine), save_dir))
    if got is not None:
        path, loaded, m, v_, gs = got
        views = trainer.opt.master_views()
        print("Loading checkpoint {}".format(path))
        for k, v in views.items():
            v.copy_(loaded[k])
        trainer.opt.m.copy_(m)
        trainer.opt.v.copy_(v_)
        trainer.opt.global_step = gs
        return gs
    return None


def train(log_dir, args, hparams, input_path, device="cuda", params=None):
    import torch
    import torch.distributed as dist
    from . import weights
    from .model import FloWaveNet
    from .optim import learning_rate
    from .synthesize import write_wav
    from .training import Trainer

    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    save_dir = os.path.join(log_dir, "pretrained")
    train_logdir, test_logdir = os.path.join(log_dir, "train"), os.path.join(log_dir, "test")
    for d in (save_dir, train_logdir, test_logdir):
        os.makedirs(d, exist_ok=True)
    checkpoint_path = os.path.join(save_dir, "flowavenet_model.ckpt")
    metadata_filename = os.path.join(args.base_dir, input_path)
    if rank == 0:
        print("Checkpoint_path: {}".format(checkpoint_path))
        print("Loading training data from: {}".format(metadata_filename))
    seed = getattr(args, "seed", None)
    if metadata_filename.endswith(".tfrecord"):       # the reference's own files (train.py:161-162)
        dataset = Dataset.from_tfrecords(metadata_filename, os.path.join(os.path.dirname(metadata_filename), "test.tfrecord"),
                                         hparams, seed=seed, rank=rank)
    else:
        dataset = Dataset(metadata_filename, hparams, seed=seed, rank=rank)
    if params is None:     # the reference's initialisers: he-uniform convs, g = 1, ZeroConv1d all zeros (modules.py:21-22,47-49)
        params = weights.synthetic_params(hparams, hparams.tf_random_seed if seed is None else seed, zero_conv="zeros")
    trainer = Trainer(hparams, params, device=device)
    step = None
    if args.restore:
        step = restore_checkpoint(save_dir, trainer)
    if step is None:
        if rank == 0:
            print("Starting new training!" if not args.restore else "No checkpoint found.")
            print("Init ActNorm layer...", end="")
        mels, audios = dataset.next_train()
        trainer.ddi(audios, mels)                                             # train.py:221,229 (init=True)
        init_loss = float(trainer.step(audios, mels)[0])                     # ... which also applies an update
        step = trainer.opt.global_step
        if rank == 0:
            print(" OK. Init loss: {:.5f}".format(init_loss))
    if rank == 0:
        print("FloWaveNet training set to a maximum of {} steps".format(args.train_steps))

    def log(path, rec):
        with open(os.path.join(path, "summary.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")

    while step < args.train_steps:
        start_time = time.time()
        mels, audios = dataset.next_train()
        loss, log_p, logdet, gnorm = trainer.step(audios, mels)
        step = trainer.opt.global_step
        total_loss = float(loss)                                             # synchronises: the step time is real
        step_duration = time.time() - start_time
        if rank == 0:
            print("Step {:7d} [{:.3f} sec/step, loss={:.5f}, log_p={:.5f}, logdet={:.5f}]".format(
                step, step_duration, total_loss, float(log_p), float(logdet)), end="\r")
        if rank == 0 and step % args.summary_interval == 0:
            print("\nWriting summary at step {}".format(step))
            log(train_logdir, {"step": step, "losses/total_loss": total_loss, "losses/log_p": float(log_p),
                               "losses/logdet": float(logdet), "learning_rate": learning_rate(step - 1),
                               "gradient_global_norm": float(gnorm)})
            tm, ta = dataset.next_test()                                     # get_test_losses, train.py:85-91
            model = FloWaveNet(hparams, device=device, cond_mode=1).load_params(trainer.opt.master_views())
            tlp, tld = model.forward(torch.from_numpy(ta).reshape(ta.shape[0], -1, 1), torch.from_numpy(tm))
            log(test_logdir, {"step": step, "losses/total_loss": -(float(tlp) + float(tld)),
                              "losses/log_p": float(tlp), "losses/logdet": float(tld)})
        if rank == 0 and (step % args.checkpoint_interval == 0 or step == args.train_steps):
            save_checkpoint("%s-%d.npz" % (checkpoint_path, step), trainer)
        if rank == 0 and step % args.eval_interval == 0:
            print("\nEvaluating at step {}".format(step))                     # predict_random_samples, train.py:114-139
            mel, wav = dataset.eval_sample()
            model = FloWaveNet(hparams, device=device, cond_mode=1).load_params(trainer.opt.master_views())
            g = torch.Generator().manual_seed(step)
            z = torch.randn(1, len(wav), 1, generator=g) * hparams.temp
            pred = model.reverse(z, torch.from_numpy(mel[None])).reshape(-1).cpu().numpy()
            write_wav(os.path.join(train_logdir, "predictions-%d.wav" % step), pred, hparams.sample_rate)
            write_wav(os.path.join(train_logdir, "targets-%d.wav" % step), wav, hparams.sample_rate)
    return save_dir


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--base_dir", default="")
    parser.add_argument("--input", default="training_data/train.txt")
    parser.add_argument("--input_dir", default="training_data/", help="folder to contain inputs sentences/targets")
    parser.add_argument("--restore", type=lambda s: str(s).lower() not in ("false", "0", "no"), default=True,
                        help="Set this to False to do a fresh training")
    parser.add_argument("--summary_interval", type=int, default=500, help="Steps between running summary ops")
    parser.add_argument("--checkpoint_interval", type=int, default=2000, help="Steps between writing checkpoints")
    parser.add_argument("--eval_interval", type=int, default=5000, help="Steps between eval on test data")
    parser.add_argument("--train_steps", type=int, default=2000000, help="total number of model training steps")
    parser.add_argument("--log_dir", default="logs")
    parser.add_argument("--seed", type=int, default=None)
    args = parser.parse_args(argv)
    import torch
    import torch.distributed as dist
    from .hparams import hparams
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    if world > 1:
        dist.init_process_group("nccl")
    try:
        train(os.path.join(args.base_dir, args.log_dir), args, hparams, args.input)
    finally:
        if world > 1:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
