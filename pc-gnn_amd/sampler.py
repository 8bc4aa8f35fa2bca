"""Pick step: the label-balanced sampler of the reference (src/utils.py:274-278) on the device.

``random.choices(idx_train, weights=deg/LF, k)`` is a running sum of the weights
plus one ``bisect_right`` per draw.  The weights do not change between epochs, so
the fp64 running sum is formed once on the host exactly as CPython forms it
(sequentially), uploaded, and every epoch is a single kernel of k binary searches
(``pcg_pick``).  Given the same uniform draws the picked ids are bit-identical to
the reference's; without them the draws come from Philox4x32-10 on the device.
"""
from typing import Optional, Sequence

import numpy as np
import torch

from . import ops


class PickSampler:
    def __init__(self, idx_train: Sequence[int], y_train: np.ndarray, degree_train: np.ndarray, device, seed: int = 0):
        y = np.asarray(y_train)
        lf = (y.sum() - len(y)) * y + len(y)                       # utils.py:276
        self.weights = np.asarray(degree_train, dtype=np.int64) / lf  # utils.py:277
        self.cum_host = np.cumsum(self.weights)                    # itertools.accumulate, fp64, sequential
        self.device = torch.device(device)
        self.cum = torch.from_numpy(self.cum_host).to(self.device)
        self.idx_train = torch.from_numpy(np.asarray(idx_train, dtype=np.int32)).to(self.device)
        self.seed = seed

    def pick(self, size: int, epoch: int = 0, uniforms: Optional[torch.Tensor] = None,
             out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """int32 device tensor of `size` training-node ids, drawn with replacement."""
        return ops.pick(self.cum, self.idx_train, size, uniforms, self.seed, epoch, out)

    def pick_shuffled(self, size: int, out_ids: torch.Tensor, labels_all: Optional[torch.Tensor] = None,
                      out_labels: Optional[torch.Tensor] = None, epoch: int = 0,
                      epoch_counter: Optional[torch.Tensor] = None, bump: bool = False, n_epochs: int = 1) -> torch.Tensor:
        """pick + random.shuffle + label lookup of one epoch (or of n_epochs consecutive ones, `size` draws each) in one launch
        (utils.py:274-278, model_handler.py:131-133)."""
        return ops.pick_shuffled(self.cum, self.idx_train, size, self.seed, epoch, out_ids, labels_all, out_labels,
                                 epoch_counter, bump, n_epochs)


def pick_step(idx_train, y_train, adj_list, size, device="cuda", uniforms=None, seed=0, epoch=0):
    """Reference signature (utils.py:274): returns a Python list like the reference does.
    (The training harness keeps the ids on the device instead - PickSampler.pick.)"""
    deg = np.array([len(adj_list[v]) for v in idx_train])
    s = PickSampler(idx_train, y_train, deg, device, seed)
    u = None if uniforms is None else torch.as_tensor(uniforms, dtype=torch.float64, device=s.device)
    return s.pick(size, epoch, u).cpu().tolist()
