"""numpy stand-in for the per-shard HIP stages, with the interface contrack_amd.dist.run_sharded expects.
TEST INFRASTRUCTURE: lets the distributed driver (halo exchange, table all-gather, replicated resolve,
extent all-reduce) run under gloo on CPU.  Built on tests/cpu_tables.py and the oracle's threshold."""
import numpy as np
import torch

import cpu_tables
from contrack_amd import _native
from oracle import cpu_oracle


class CpuShardEngine:
    def __init__(self, anom, thr, gorl, wrow):
        self.anom, self.thr, self.gorl, self.wrow = anom, thr, gorl, wrow
        self.T, self.ny, self.nx = anom.shape
        self.prev_lab = None
        self.wlo, self.whi, self.wshift, self.limb_bits = _native.weights_to_limbs(wrow, npix=self.ny * self.nx, with_bits=True)

    def label2d(self, has_prev):
        self.has_prev = has_prev
        if self.T:
            self.mask = cpu_oracle.threshold_mask(self.anom, self.thr, self.gorl).astype(bool)
        else:
            self.mask = np.zeros((0, self.ny, self.nx), dtype=bool)
        self.last_lab = cpu_tables.label_step(self.mask[-1])[0] if self.T else np.zeros((self.ny, self.nx), np.int32)

    def halo_nbytes(self):
        return self.ny * self.nx * 4

    def halo_export(self):
        return torch.from_numpy(self.last_lab.astype(np.int32).reshape(-1).view(np.uint8).copy())

    def halo_template(self):
        return torch.empty(self.halo_nbytes(), dtype=torch.uint8)

    def halo_import(self, tensor):
        self.prev_lab = tensor.numpy().view(np.int32).reshape(self.ny, self.nx).copy()

    def overlap(self):
        self.tb = cpu_tables.build_tables(self.mask, self.wlo, self.whi, prev_lab=self.prev_lab if self.has_prev else None)

    def tables(self):
        return cpu_tables.pack_blob(self.tb, self.wshift, self.has_prev and self.prev_lab is not None, limb_bits=self.limb_bits)

    def extents(self, result, shard, t_begin):
        comp_label, ops = result.arrays()
        co, _ = result.shard_offsets()
        n_labels = result.info()["n_labels"]
        self.ids = cpu_tables.fold_ids(self.tb["labs"], comp_label[co[shard]:co[shard + 1]], ops, t_begin)
        tmin = np.full(n_labels + 1, np.iinfo(np.int32).max, dtype=np.int32)
        tmax = np.full(n_labels + 1, np.iinfo(np.int32).min, dtype=np.int32)
        for k in range(self.T):
            u = np.unique(self.ids[k])
            u = u[u > 0]
            tmin[u] = np.minimum(tmin[u], t_begin + k)
            tmax[u] = np.maximum(tmax[u], t_begin + k)
        self.tmin, self.tmax = torch.from_numpy(tmin), torch.from_numpy(tmax)
        return self.tmin, self.tmax

    def write(self, persistence):
        tmin, tmax = self.tmin.numpy().astype(np.int64), self.tmax.numpy().astype(np.int64)
        present = tmax >= tmin
        alive = present & (tmax - tmin + 1 >= persistence)
        alive[0] = False
        lut = np.where(alive, np.arange(len(alive)), 0).astype(np.int32)
        self.flag = lut[self.ids] if self.T else np.zeros((0, self.ny, self.nx), np.int32)
        return int(alive.sum()), bool((self.flag == 0).any())
