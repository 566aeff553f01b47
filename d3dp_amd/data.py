"""Training / evaluation batch assembly with the pose pools resident in HBM.

The reference's ``ChunkedGenerator_Seq`` (common/generators.py:12-171) copies every chunk of every batch through
float64 numpy buffers on the host, flips it there, and the training loop then converts and uploads it
(main.py:355-362).  An MI355X holds the whole Human3.6M training set (1.56 M frames x 17 joints x 5 floats =
0.5 GB) in a corner of its 288 GB, so here the pools are uploaded once and a batch is ONE gather launch
(include/d3dp_hip.h: d3dp_batch_gather) driven by a small int32 table; the host keeps only the lineage
bookkeeping, which is restated so that batches come out in the reference's order:

  * pairs: per sequence ceil(n/F) chunks, centred (offset = (n_chunks*F - n)//2), frames outside the sequence repeat
    the edge frame; with ``augment`` every chunk appears a second time flipped (generators.py:41-50);
  * one ``RandomState(1234).permutation(pairs)`` per epoch when shuffling (generators.py:93-95) -- the RandomState
    object is what checkpoints store (main.py:546) and restore (main.py:338);
  * batches of ``batch_size`` consecutive pairs, the last one short (generators.py:104).

Flipping negates x and swaps left/right joints for 2D and 3D (generators.py:120-140) and negates camera
parameters 2 and 7 (generators.py:145-148).  ``zero_root`` folds main.py:364-365 (trajectory split + root zeroing)
into the same pass.
"""
from __future__ import annotations

from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .clips import flip_perm


def build_pairs(lengths: Sequence[int], chunk_length: int, augment: bool) -> np.ndarray:
    """(n_pairs, 4) int64 rows (seq_idx, start_frame, end_frame, flip) in the reference's order."""
    rows: List[Tuple[int, int, int, int]] = []
    for i, n in enumerate(lengths):
        n_chunks = (n + chunk_length - 1) // chunk_length
        offset = (n_chunks * chunk_length - n) // 2
        bounds = np.arange(n_chunks + 1) * chunk_length - offset
        plain = [(i, int(a), int(b), 0) for a, b in zip(bounds[:-1], bounds[1:])]
        rows += plain
        if augment:
            rows += [(i, a, b, 1) for (i, a, b, _) in plain]
    return np.asarray(rows, dtype=np.int64).reshape(-1, 4)


class ChunkLineage:
    """Host bookkeeping of ``ChunkedGenerator_Seq``: which (sequence, chunk, flip) goes into which batch of which
    epoch.  Pure numpy; the device batcher below turns each epoch's pairs into one int32 table."""

    def __init__(self, lengths, batch_size, chunk_length, shuffle=True, random_seed=1234, augment=False, endless=False):
        self.batch_size, self.chunk_length = int(batch_size), int(chunk_length)
        self.lengths = [int(n) for n in lengths]
        self.offsets = np.concatenate(([0], np.cumsum(self.lengths)[:-1])).astype(np.int64)
        self.pairs = build_pairs(self.lengths, self.chunk_length, augment)
        self.num_batches = (len(self.pairs) + self.batch_size - 1) // self.batch_size
        self.random = np.random.RandomState(random_seed)
        self.shuffle, self.endless, self.state, self.augment = shuffle, endless, None, augment

    # ---- the reference generator's small API ---------------------------------------------------------------------
    def num_frames(self):
        return self.num_batches * self.batch_size

    def batch_num(self):
        return self.num_batches

    def random_state(self):
        return self.random

    def set_random_state(self, random):
        self.random = random

    def augment_enabled(self):
        return self.augment

    def next_pairs(self):
        if self.state is None:
            pairs = self.random.permutation(self.pairs) if self.shuffle else self.pairs
            return 0, pairs
        return self.state

    def table(self, pairs: np.ndarray) -> np.ndarray:
        """(n_pairs, 4) int32 rows (pool offset of the sequence, its length, chunk start, flip)."""
        seq = pairs[:, 0]
        t = np.stack((self.offsets[seq], np.asarray(self.lengths, dtype=np.int64)[seq], pairs[:, 1], pairs[:, 3]), axis=1)
        return t.astype(np.int32)


class ChunkedBatcher(ChunkLineage):
    """Device-resident counterpart of ``ChunkedGenerator_Seq`` (same constructor arguments and methods)."""

    def __init__(self, batch_size, cameras, poses_3d, poses_2d, chunk_length, pad=0, causal_shift=0, shuffle=True,
                 random_seed=1234, augment=False, kps_left=None, kps_right=None, joints_left=None, joints_right=None,
                 endless=False, device=None, zero_root=False):
        assert poses_3d is None or len(poses_3d) == len(poses_2d), (len(poses_3d), len(poses_2d))
        assert cameras is None or len(cameras) == len(poses_2d)
        for i in range(len(poses_2d)):
            assert poses_3d is None or poses_2d[i].shape[0] == poses_3d[i].shape[0]
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda":
            raise _lib.D3DPHipError("ChunkedBatcher keeps its pools on an MI355X (device %s); there is no CPU fallback"
                                    % self.device)
        super().__init__([p.shape[0] for p in poses_2d], batch_size, chunk_length, shuffle, random_seed, augment, endless)
        self.pad, self.causal_shift, self.zero_root = pad, causal_shift, bool(zero_root)
        self.cameras = None if cameras is None else [np.asarray(c) for c in cameras]
        self.J = int(poses_2d[0].shape[-2])
        dev = self.device
        self.pool2d = torch.from_numpy(np.concatenate([np.asarray(p, dtype=np.float32) for p in poses_2d])).to(dev)
        self.pool3d = None if poses_3d is None else torch.from_numpy(
            np.concatenate([np.asarray(p, dtype=np.float32) for p in poses_3d])).to(dev)
        ident = list(range(self.J))
        self.perm2d = torch.tensor(flip_perm(kps_left, kps_right, self.J) if kps_left is not None else ident,
                                   dtype=torch.int32, device=dev)
        self.perm3d = torch.tensor(flip_perm(joints_left, joints_right, self.J) if joints_left is not None else ident,
                                   dtype=torch.int32, device=dev)

    # ---- batches ---------------------------------------------------------------------------------------------------
    def _tables(self, pairs: np.ndarray) -> torch.Tensor:
        """device table for a whole epoch: one upload."""
        return torch.from_numpy(self.table(pairs)).to(self.device)

    def gather(self, table: torch.Tensor) -> Tuple[Optional[torch.Tensor], torch.Tensor]:
        nb = table.shape[0]
        F, J = self.chunk_length, self.J
        out2d = torch.empty((nb, F, J, 2), dtype=torch.float32, device=self.device)
        out3d = None if self.pool3d is None else torch.empty((nb, F, J, 3), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().d3dp_batch_gather(self.pool2d.data_ptr(), _lib.ptr(self.pool3d), table.data_ptr(),
                                                     self.perm2d.data_ptr(), self.perm3d.data_ptr(), out2d.data_ptr(),
                                                     _lib.ptr(out3d), nb, F, J, int(self.zero_root),
                                                     _lib.current_stream()), "d3dp_batch_gather")
        return out3d, out2d

    def _cams(self, chunk: np.ndarray):
        if self.cameras is None:
            return None
        cam = np.stack([np.asarray(self.cameras[s], dtype=np.float64) for s in chunk[:, 0]])
        flip = chunk[:, 3].astype(bool)
        cam[flip, 2] *= -1
        cam[flip, 7] *= -1
        return cam

    def next_epoch(self) -> Iterator[Tuple[Optional[np.ndarray], Optional[torch.Tensor], torch.Tensor]]:
        """Yields (cameras or None, batch_3d or None, batch_2d) like the reference generator; the pose batches are
        fp32 GPU tensors."""
        bs = self.batch_size
        while True:
            first, pairs = self.next_pairs()
            table = self._tables(pairs)                    # the whole epoch's index table: one upload
            for k in range(first, self.num_batches):
                rows = slice(k * bs, (k + 1) * bs)
                b3, b2 = self.gather(table[rows])
                if self.endless:
                    self.state = (k + 1, pairs)            # where an endless generator resumes
                yield self._cams(pairs[rows]), b3, b2
            if not self.endless:
                return
            self.state = None


class UnchunkedSequences:
    """Counterpart of ``UnchunkedGenerator_Seq`` (generators.py:174-250): whole sequences, one at a time, on the GPU;
    with ``augment`` a flipped copy is appended as the second batch element."""

    def __init__(self, cameras, poses_3d, poses_2d, pad=0, causal_shift=0, augment=False, kps_left=None, kps_right=None,
                 joints_left=None, joints_right=None, device=None):
        assert poses_3d is None or len(poses_3d) == len(poses_2d)
        assert cameras is None or len(cameras) == len(poses_2d)
        self.device = torch.device(device if device is not None else "cuda")
        self.augment = False                       # like the reference: the constructor argument is ignored (:197)
        self.kps_left, self.kps_right, self.joints_left, self.joints_right = kps_left, kps_right, joints_left, joints_right
        self.pad, self.causal_shift = pad, causal_shift
        self.cameras = [] if cameras is None else cameras
        self.poses_3d = [] if poses_3d is None else [torch.as_tensor(np.asarray(p, dtype=np.float32), device=self.device) for p in poses_3d]
        self.poses_2d = [torch.as_tensor(np.asarray(p, dtype=np.float32), device=self.device) for p in poses_2d]

    def num_frames(self):
        return sum(int(p.shape[0]) for p in self.poses_2d)

    def augment_enabled(self):
        return self.augment

    def set_augment(self, augment):
        self.augment = augment

    def _mirrored(self, x, left, right):
        """x with its first coordinate negated and the left / right joints swapped (the flip augmentation)."""
        y = x.clone()
        y[..., 0] *= -1
        return y[:, :, flip_perm(left, right, y.shape[2])]

    def next_epoch(self):
        n = len(self.poses_2d)
        for i in range(n):
            cam = self.cameras[i] if i < len(self.cameras) else None
            b3 = self.poses_3d[i][None] if i < len(self.poses_3d) else None
            b2 = self.poses_2d[i][None]
            b_cam = None if cam is None else np.asarray(cam)[None]
            if self.augment:                               # second batch element: the mirrored copy (camera: cx and the
                if b_cam is not None:                      # tangential term change sign)
                    b_cam = np.repeat(b_cam, 2, axis=0)
                    b_cam[1, [2, 7]] *= -1
                if b3 is not None:
                    b3 = torch.cat((b3, self._mirrored(b3, self.joints_left, self.joints_right)), dim=0)
                b2 = torch.cat((b2, self._mirrored(b2, self.kps_left, self.kps_right)), dim=0)
            yield b_cam, b3, b2
