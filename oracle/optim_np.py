"""fp64 NumPy restatement of the reference's data-parallel update (TEST INFRASTRUCTURE ONLY).

average_gradients (utils.py:34-60): per variable, mean over towers.
build_model tail (train.py:75-81): x 1/scale, clip_by_global_norm(., 1), Adam.apply_gradients.
get_optimizer (train.py:15-24): step-wise learning rate.
TF-1.12 semantics: clip: g * clip / max(||g||, clip); Adam:
lr_t = lr*sqrt(1-b2^t)/(1-b1^t), m <- b1 m + (1-b1) g, v <- b2 v + (1-b2) g^2,
theta <- theta - lr_t * m / (sqrt(v) + eps)   (eps outside the bias correction).
"""
import numpy as np


def learning_rate(step: int) -> float:
    """train.py:17-20."""
    lr = 0.001
    if step >= 200000:
        lr = 0.001 / 2
    if step >= 400000:
        lr = 0.001 / 4
    if step >= 600000:
        lr = 0.001 / 6
    return lr


def average_gradients(tower_grads):
    """utils.py:34-60: list (towers) of lists (variables) of arrays -> list of means."""
    return [np.mean(np.stack(gs, 0), 0) for gs in zip(*tower_grads)]


def clip_by_global_norm(grads, clip_norm=1.0):
    """train.py:27-32 / tf.clip_by_global_norm."""
    gn = np.sqrt(sum(float(np.sum(np.square(g, dtype=np.float64))) for g in grads))
    return [g * (clip_norm / max(gn, clip_norm)) for g in grads], gn


def adam_step(theta, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer.apply_gradients for one tensor (step counts from 1)."""
    lr_t = lr * np.sqrt(1.0 - b2 ** step) / (1.0 - b1 ** step)
    m = b1 * m + (1.0 - b1) * g
    v = b2 * v + (1.0 - b2) * g * g
    return theta - lr_t * m / (np.sqrt(v) + eps), m, v


def data_parallel_update(theta, tower_grads_scaled, m, v, step, scale=64.0):
    """One train.py:75-81 update on flat vectors: tower gradients of (scale * loss)."""
    g = np.mean(np.stack(tower_grads_scaled, 0), 0) / scale
    (g,), gn = clip_by_global_norm([g], 1.0)
    theta, m, v = adam_step(theta, g, m, v, step, learning_rate(step - 1))
    return theta, m, v, gn
