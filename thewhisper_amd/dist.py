"""Multi-GPU layout: one process per MI355X, streams sharded across ranks, no collective on the data path.

Streams (and chunks of an offline file) are independent and the model fits one GPU 90x over, so every rank
holds a full replica.  ``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm,
"gloo" on CPU for the tests) is used only for the launch barrier, the max-over-ranks timing and the optional
fixed-shape gather of results - an 8 KB message per rank, latency-bound on the 7 x 153 GB/s xGMI links.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch


def shard_streams(n_streams: int, rank: int, world: int) -> List[int]:
    """Sticky assignment stream_id -> rank = stream_id % world (per-stream scheduler state and KV slots stay local)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return [s for s in range(n_streams) if s % world == rank]


class Replicas:
    """Thin wrapper over the default process group; a no-op when WORLD_SIZE == 1."""

    def __init__(self, backend: Optional[str] = None, device: Optional[torch.device] = None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device or (torch.device("cuda", self.local_rank) if torch.cuda.is_available() else torch.device("cpu"))
        self.dist = None
        self._coll_device = self.device
        if self.world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            # TW_DIST_BACKEND=gloo: plumbing tests of the N > 1 path on a box with fewer GPUs than ranks (RCCL refuses two
            # ranks on one device); the collectives here are three scalars, so the backend does not matter for the numbers
            backend = backend or os.environ.get("TW_DIST_BACKEND") or ("nccl" if self.device.type == "cuda" else "gloo")
            if not dist.is_initialized():
                kw = {"device_id": self.device} if backend == "nccl" else {}
                dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)
            self.dist = dist
            self._coll_device = self.device if backend == "nccl" else torch.device("cpu")

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def max_float(self, x: float) -> float:
        if self.dist is None:
            return float(x)
        t = torch.tensor([x], dtype=torch.float64, device=self._coll_device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_int(self, x: int) -> int:
        if self.dist is None:
            return int(x)
        t = torch.tensor([x], dtype=torch.int64, device=self._coll_device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t.item())

    def gather_floats(self, x: float) -> List[float]:
        """One float per rank, in rank order, on every rank (per-rank throughput for the benchmark line)."""
        if self.dist is None:
            return [float(x)]
        t = torch.tensor([x], dtype=torch.float64, device=self._coll_device)
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    def gather_tokens(self, stream_ids: Sequence[int], tokens: np.ndarray, n_streams: int) -> Optional[Dict[int, np.ndarray]]:
        """Optional result gather: every rank contributes a fixed-shape int32 [cap, 1 + L] buffer (stream id + tokens);
        rank 0 returns {stream_id: tokens}.  One all_gather; never on the per-token path."""
        tokens = np.asarray(tokens, dtype=np.int32)
        L = tokens.shape[1] if tokens.ndim == 2 else 0
        if self.dist is None:
            return {int(s): tokens[i] for i, s in enumerate(stream_ids)}
        cap = (n_streams + self.world - 1) // self.world
        buf = torch.full((cap, 1 + L), -1, dtype=torch.int32)
        for i, s in enumerate(stream_ids):
            buf[i, 0] = int(s)
            buf[i, 1:] = torch.from_numpy(tokens[i])
        buf = buf.to(self._coll_device)
        out = [torch.empty_like(buf) for _ in range(self.world)]
        self.dist.all_gather(out, buf)
        if self.rank != 0:
            return None
        res: Dict[int, np.ndarray] = {}
        for t in out:
            a = t.cpu().numpy()
            for row in a:
                if row[0] >= 0:
                    res[int(row[0])] = row[1:].copy()
        return res

    def close(self):
        if self.dist is not None and self.dist.is_initialized():
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None
