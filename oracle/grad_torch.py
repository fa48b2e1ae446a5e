"""TEST INFRASTRUCTURE ONLY - gradient oracle: autograd through the independent torch formulation
(oracle/flowavenet_torch.py) of loss = -(log_p + logdet) (train.py:56-66), fp64, with respect to the
reference's raw parameters (weight-norm V / g / bias, ZeroConv kernel / bias / scale, ActNorm b / logs,
up-sampling kernels).  PARITY UNPINNED like the rest of the oracle (TF 1.12 cannot run here).
Only tests/ may import this module."""
import numpy as np
import torch

from . import flowavenet_torch as OT


def loss_and_grads(params, x, c, hp):
    """params: dict name -> ndarray (reference layouts); x [B,T,1] or [B,T]; c [B,F,mels].
    Returns (loss, log_p, logdet, grads dict name -> ndarray)."""
    leaves = {k: torch.tensor(np.asarray(v, dtype=np.float64), requires_grad=True) for k, v in params.items()}
    fp = OT.fold(leaves, hp)
    xt = torch.as_tensor(np.asarray(x, dtype=np.float64)).reshape(np.shape(x)[0], -1, 1)
    ct = torch.as_tensor(np.asarray(c, dtype=np.float64))
    log_p, logdet, _ = OT.forward(fp, xt, ct, hp, as_tensors=True)
    loss = -(log_p + logdet)
    loss.backward()
    grads = {k: (v.grad.numpy() if v.grad is not None else np.zeros(v.shape)) for k, v in leaves.items()}
    return float(loss.detach()), float(log_p.detach()), float(logdet.detach()), grads
