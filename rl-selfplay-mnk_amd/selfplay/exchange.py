"""The exchange step of the sharded rollout through the C ABI: an RCCL communicator (one rank per GPU, xGMI inside a
node) and ``mnk_allgather_records`` on a caller-chosen HIP stream.

The reference has no distributed code.  What crosses the links is the content of its ``RolloutBuffer``
(``/root/reference/src/alg/rollout_buffer.py:14-44``) in this build's packed forms: the packed records of a
chunk, or the message "chunk-start planes | action log | chunk-start meta" (``selfplay/random_rollout.py``) that
``mnk_replay_actions`` expands on the receiver.  Env shards are independent, so this all-gather is the only
collective on the path.

``torch.distributed`` is used for what it is good at -- rendezvous: rank 0 creates the 128-byte communicator id
and the process group (any backend) broadcasts it.  The collective itself is issued by ``libmnk_hip.so`` on the
same RCCL the process already has loaded, on the stream the caller names, so the all-gather is ordered against the
rollout kernel by plain stream / event dependencies like every other entry point of the ABI.
"""
import ctypes
from typing import Optional

import torch

import mnk_hip


class RecordExchange:
    """One RCCL communicator over all ranks of ``group`` (default: the world), this rank on its current device."""

    def __init__(self, rank: int, world: int, comm_id: bytes):
        if len(comm_id) != mnk_hip.COMM_ID_BYTES:
            raise ValueError(f"communicator id must be {mnk_hip.COMM_ID_BYTES} bytes")
        self.rank, self.world = int(rank), int(world)
        handle = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(comm_id, mnk_hip.COMM_ID_BYTES)
        mnk_hip.call("mnk_comm_init", ctypes.byref(handle), buf, self.world, self.rank)
        self._comm = handle
        # False: ncclAllGather (RCCL picks the algorithm); True: one grouped send + receive per peer, every message
        # once over each of the rank's own xGMI links (mnk_allgather_records_direct)
        self.direct = False

    @staticmethod
    def new_id() -> bytes:
        buf = ctypes.create_string_buffer(mnk_hip.COMM_ID_BYTES)
        mnk_hip.call("mnk_comm_unique_id", buf)
        return buf.raw

    @classmethod
    def from_process_group(cls, group=None) -> "RecordExchange":
        """Rendezvous over an initialised ``torch.distributed`` group: rank 0's id is broadcast to the others."""
        import torch.distributed as dist

        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.new_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(rank, world, box[0])

    def all_gather(self, send: torch.Tensor, recv: torch.Tensor, stream: Optional[torch.cuda.Stream] = None,
                   direct: Optional[bool] = None) -> None:
        """recv[r] = rank r's ``send`` for every r; ``recv`` holds ``world * send.numel()`` elements of the same
        dtype.  Enqueued on ``stream`` (default: the current stream of ``send``'s device); returns at once.
        ``direct`` (default: ``self.direct``) picks the per-peer send / receive form of the exchange."""
        nbytes = send.numel() * send.element_size()
        if recv.numel() * recv.element_size() != self.world * nbytes:
            raise ValueError(f"recv holds {recv.numel() * recv.element_size()} bytes, need {self.world} x {nbytes}")
        if send.device != recv.device or send.device.type != "cuda":
            raise ValueError("send and recv must live on the same GPU")
        s = stream.cuda_stream if stream is not None else mnk_hip.stream_ptr(send.device)
        entry = "mnk_allgather_records_direct" if (self.direct if direct is None else direct) else "mnk_allgather_records"
        mnk_hip.call(entry, self._comm, mnk_hip.ptr(send), mnk_hip.ptr(recv), nbytes, s)

    def close(self) -> None:
        if self._comm is not None and self._comm.value:
            mnk_hip.call("mnk_comm_destroy", self._comm)
        self._comm = None

    def __del__(self):  # best effort; close() explicitly before the process group goes away
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
