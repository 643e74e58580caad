"""A checker-backed stand-in for engine.HipBackend, used ONLY by the world_size-2 gloo test to
exercise the collective choreography of engine.ShardedAttention / attention_mpi on CPU.
It lives in tests/ on purpose: the product package has no CPU compute path."""
import numpy as np
import torch

import oracle as O


def _r4(x):
    return (x + 3) // 4 * 4


class OracleBackend:
    name = "oracle-standin"

    def __init__(self):
        self.o = O.Oracle()
        self.comm_device = torch.device("cpu")

    def to_device(self, a, dtype=None):
        return torch.as_tensor(a).to(dtype=dtype)

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype)

    def cvt_d2f(self, x64):
        rows, cols = x64.shape
        out = torch.zeros((rows, _r4(cols)), dtype=torch.float32)
        out[:, :cols] = x64.to(torch.float32)          # RNE, as attention-mpi.c:31-64
        return out

    def cvt_f2d(self, x32, cols):
        return x32[:, :cols].to(torch.float64)

    def shard_partial(self, Qf, Kf, Vf, dk, dv):
        c, lm, ls = self.o.shard_partial_f32(Qf[:, :dk].numpy(), Kf[:, :dk].numpy().reshape(-1, dk),
                                             Vf[:, :dv].numpy().reshape(-1, dv))
        contrib = torch.zeros((Qf.shape[0], _r4(dv)), dtype=torch.float32)
        contrib[:, :dv] = torch.from_numpy(c)
        return contrib, torch.from_numpy(lm), torch.from_numpy(ls)

    def merge_rescale(self, contrib, lsum, lmax, gmax, dv):
        corr = torch.exp(lmax - gmax)
        lsum.mul_(corr)
        contrib.mul_(corr[:, None])

    def merge_normalise(self, contrib, gsum, dv):
        inv = torch.where(gsum == 0, torch.zeros_like(gsum), 1.0 / gsum)
        contrib.mul_(inv[:, None])

    def merge_gathered(self, contrib, stats, self_index, dv):
        lmax, lsum = stats[:, 0, :], stats[:, 1, :]
        gmax = lmax.max(dim=0).values
        gsum = (lsum * torch.exp(lmax - gmax)).sum(dim=0)
        w = torch.where(gsum == 0, torch.zeros_like(gsum), torch.exp(lmax[self_index] - gmax) / gsum)
        contrib.mul_(w[:, None])

    def finish_f64(self, contrib, lsum, dv):
        inv = torch.where(lsum == 0, torch.zeros_like(lsum), 1.0 / lsum)
        return (contrib[:, :dv] * inv[:, None]).to(torch.float64)
