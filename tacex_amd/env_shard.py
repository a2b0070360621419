"""Environment sharding across the GPUs of one node + the single observation all-gather.

The reference has no collectives: multi-GPU there is IsaacLab's `--distributed` flag that pins
one process per GPU (scripts/reinforcement_learning/skrl/train.py:116-117).  Environments are independent units
(every op of the tactile path is per-env), so the path shards by contiguous env ranges with NO data-path
collective; the only exchange is collecting the policy observation of all shards - exactly one all-gather per
step (RCCL over xGMI on the GPU box, `backend="nccl"`; gloo in the CPU tests).

xGMI is a full mesh of point-to-point links, so an all-gather is one hop and per-link bound
(t ~ shard_bytes / 153 GB/s): the payload is the low-resolution policy observation (+ markers, indentation),
not the full-resolution fp32 frame (472 MB per 512-env shard would cost ~3 ms, as much as rendering it).
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import torch
import torch.distributed as dist


def shard_range(num_envs_total: int, rank: int, world_size: int) -> tuple[int, int]:
    """Contiguous env range [lo, hi) of `rank`; the first `num_envs_total % world_size` ranks get one extra env."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    q, r = divmod(num_envs_total, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


@dataclass
class ShardInfo:
    rank: int
    world_size: int
    local_rank: int
    lo: int
    hi: int

    @property
    def num_local(self) -> int:
        return self.hi - self.lo


def init_from_env(num_envs_total: int, backend: str | None = None) -> ShardInfo:
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run contract); single process if unset."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    force = os.environ.get("TACEX_FORCE_DIST") == "1" and "RANK" in os.environ  # exercise the collective path with 1 rank
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    lo, hi = shard_range(num_envs_total, rank, world)
    return ShardInfo(rank, world, local, lo, hi)


class ObservationGather:
    """Packs the per-shard observation pieces into ONE contiguous buffer and all-gathers it in ONE collective.

    pieces: dict name -> per-env shape (without the env dim). All shards must hold the same number of envs
    (all_gather_into_tensor needs equal sizes; use num_envs_total % world_size == 0).
    dtypes: optional dict name -> torch.dtype for pieces that differ from `dtype` (e.g. a uint8 policy image next to
    float32 markers); the buffer is then a byte buffer with every piece aligned to 8 bytes, still ONE collective.
    collective: None = a collective whenever world_size > 1 or a process group exists; False = never (a rank that works alone while
    the other ranks of the group wait - bench.py's strong-scaling base - must not start a collective the others do not join).
    """

    def __init__(self, pieces: dict[str, tuple[int, ...]], num_local: int, world_size: int, device, dtype=torch.float32,
                 dtypes: dict | None = None, collective: bool | None = None):
        self.pieces = {k: tuple(v) for k, v in pieces.items()}
        self.dtypes = {k: (dtypes or {}).get(k, dtype) for k in self.pieces}
        self.bytes_mode = len(set(self.dtypes.values())) > 1
        buf_dtype = torch.uint8 if self.bytes_mode else next(iter(self.dtypes.values()), dtype)
        isz = {k: (torch.empty((), dtype=d).element_size() if self.bytes_mode else 1) for k, d in self.dtypes.items()}
        self.sizes = {k: int(torch.Size(v).numel()) * isz[k] for k, v in self.pieces.items()}  # buffer elements per env
        self._off, self._cat = {}, []  # _cat: (name | None for padding, width) in buffer order
        o = 0
        for k, n in self.sizes.items():
            if self.bytes_mode and o % 8:
                pad = 8 - o % 8
                self._cat.append((None, pad))
                o += pad
            self._off[k] = o
            self._cat.append((k, n))
            o += n
        if self.bytes_mode and o % 8:
            self._cat.append((None, 8 - o % 8))
            o += 8 - o % 8
        self.row = o
        self.num_local, self.world = num_local, world_size
        self.local = torch.zeros((num_local, self.row), device=device, dtype=buf_dtype)
        # one rank, no process group: the "gathered" buffer IS the send buffer (no copy); otherwise the collective fills it
        self._collective = (world_size > 1 or (dist.is_available() and dist.is_initialized())) if collective is None else bool(collective)
        if not self._collective and world_size != 1:
            raise ValueError("collective=False needs world_size == 1")
        self._alias = world_size == 1 and not self._collective
        self.full = self.local if self._alias else torch.zeros((world_size * num_local, self.row), device=device, dtype=buf_dtype)
        self._pads = {w: torch.zeros((num_local, w), device=device, dtype=buf_dtype) for k, w in self._cat if k is None}

    def _as_buf(self, name: str, value: torch.Tensor) -> torch.Tensor:
        v = value.reshape(self.num_local, -1)
        if v.dtype != self.dtypes[name]:
            v = v.to(self.dtypes[name])
        return v.contiguous().view(torch.uint8) if self.bytes_mode else v

    def slot(self, name: str) -> torch.Tensor:
        """(num_local, n) strided view into the packed send buffer, in the buffer's dtype (bytes in mixed-dtype mode)."""
        o, n = self._off[name], self.sizes[name]
        return self.local[:, o:o + n]

    def pack(self, name: str, value: torch.Tensor):
        self.wait()
        self.slot(name).copy_(self._as_buf(name, value))

    def pack_all(self, values: dict[str, torch.Tensor]):
        """Fill the whole send buffer with ONE kernel (a concatenation along the row) instead of one strided copy per
        piece: at 256 envs every extra launch costs ~5 us of a ~600 us step."""
        self.wait()  # the previous step's collective still reads the send buffer
        torch.cat([self._pads[w] if k is None else self._as_buf(k, values[k]) for k, w in self._cat], dim=1, out=self.local)

    def payload_bytes(self) -> int:
        return self.local.numel() * self.local.element_size()

    def gather_async(self):
        """Issue the step's collective WITHOUT stalling the compute stream: RCCL runs it on its own stream behind the
        packing kernel, so it overlaps the next step's rendering.  `wait()` (called by `pack*` and `views`) orders the
        compute stream behind it before the send buffer is refilled or the result is read."""
        self.wait()
        if self._collective:
            self._work = dist.all_gather_into_tensor(self.full, self.local, async_op=True)
        elif not self._alias:
            self.full.copy_(self.local)

    def wait(self):
        w = getattr(self, "_work", None)
        if w is not None:
            w.wait()
            self._work = None

    def gather(self) -> dict[str, torch.Tensor]:
        """Blocking form: collective + views of the gathered pieces."""
        self.gather_async()
        return self.views()

    def views(self) -> dict[str, torch.Tensor]:
        self.wait()
        out = {}
        for k, shape in self.pieces.items():
            o, n = self._off[k], self.sizes[k]
            piece = self.full[:, o:o + n]
            if self.bytes_mode:
                piece = piece.view(self.dtypes[k])
            out[k] = piece.reshape((self.full.shape[0],) + shape)
        return out
