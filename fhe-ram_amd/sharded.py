"""Row-sharded FHE-RAM across the GPUs of one node.

One process per GPU.  Rank g owns rows r = g (mod G) of every sub-RAM; the address digits and the
evaluation keys are replicated.  Data-path communication per operation:

  read / read_prepare_write : ONE all-gather of word_size GLWEs per rank (98 304 B each on the
                              device), then the root finishes (top log2 G packing levels,
                              coordinate-1 products, trace);
  write                     : ONE broadcast of word_size GLWEs (the un-rotated ct_lo) from the root.

The packing combine is not a sum (it contains key-switches and limb renormalisation), so this is an
all-gather + local finish, not a reduce: the result is bit-identical to the unsharded path
(tests/test_gpu_sharded.py).

`ShardedRam` is engine-agnostic: the engine is the HIP context (`fheram_amd.Ram(shard=, n_shards=)`);
the communicator is torch.distributed (backend "nccl" = RCCL over xGMI on device buffers, or "gloo"
on host buffers) or an in-process stand-in for single-GPU tests.
"""
from typing import List, Optional

import numpy as np


class LocalComm:
    """In-process communicator for tests: G 'ranks' living in one process exchange through a list."""

    def __init__(self, n_shards: int):
        self.n = n_shards
        self._slots: List[Optional[np.ndarray]] = [None] * n_shards
        self._bcast = None

    # collective emulation: callers deposit, then collect
    def deposit(self, rank, buf):
        self._slots[rank] = np.array(buf, copy=True)

    def gathered(self):
        assert all(s is not None for s in self._slots)
        out = np.stack(self._slots)
        self._slots = [None] * self.n
        return out


class TorchComm:
    """torch.distributed communicator.  Host mode (gloo): numpy int64 buffers.  Device mode (nccl/RCCL):
    int32 CUDA tensors whose data_ptr is handed to the C ABI."""

    def __init__(self, device_buffers: bool):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.device = device_buffers
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def alloc(self, n_glwe: int, glwe_len: int):
        if self.device:
            return self.torch.empty((n_glwe, glwe_len), dtype=self.torch.int32, device="cuda")
        return np.zeros((n_glwe, glwe_len), dtype=np.int64)

    def handle(self, buf):
        """what the engine takes: ndarray, or (device pointer, True)"""
        return (buf.data_ptr(), True) if self.device else buf

    # Device mode: the evaluator enqueues on its own HIP stream, RCCL on torch's current stream.  The two
    # are ordered with events (fheram_stream_signal / fheram_stream_wait); the host never blocks.
    def _stream(self):
        return self.torch.cuda.current_stream().cuda_stream

    def engine_done(self, engine):
        """the collective that follows must see what the engine has enqueued so far"""
        if self.device:
            engine.stream_signal(self._stream())

    def engine_next(self, engine):
        """what the engine enqueues next must see the collective's result"""
        if self.device:
            engine.stream_wait(self._stream())

    def all_gather(self, buf, out):
        """out: [world * n_glwe][glwe_len], persistent (the engine reads it asynchronously)"""
        if self.device:
            self.dist.all_gather_into_tensor(out, buf)
            return out
        t = self.torch.from_numpy(buf)
        outs = [self.torch.from_numpy(out[r * buf.shape[0]:(r + 1) * buf.shape[0]]) for r in range(self.world)]
        self.dist.all_gather(outs, t)
        return out

    def broadcast(self, buf, root: int):
        if self.device:
            self.dist.broadcast(buf, src=root)
            return buf
        t = self.torch.from_numpy(buf)
        self.dist.broadcast(t, src=root)
        return buf


class ShardedRam:
    """Ram::read / read_prepare_write / write (ram.rs:172-294) over a row-sharded RAM; one instance per rank.

    engine: object with read_partial / read_finish / write_begin / write_root / write_shard (fheram_amd.Ram
            created with shard/n_shards; device buffers also need stream_signal / stream_wait) and .params."""

    def __init__(self, engine, comm: TorchComm, root: int = 0, download: bool = True):
        """download=False leaves the result of a read on the root's device (engine.result() fetches it)."""
        self.engine, self.comm, self.root, self.download = engine, comm, root, download
        p = engine.params
        self.ws, self.glen = p.word_size(), p.glwe_len()
        self._part = comm.alloc(self.ws, self.glen)
        self._gath = comm.alloc(comm.world * self.ws, self.glen)
        self._ctlo = comm.alloc(self.ws, self.glen)

    def _read(self, address, keys, prepare_write):
        c = self.comm
        c.engine_next(self.engine)        # an earlier collective may still be reading _part
        self.engine.read_partial(address, keys, prepare_write, out=c.handle(self._part))
        c.engine_done(self.engine)        # also orders the gather behind an earlier read_finish that reads _gath
        gathered = c.all_gather(self._part, self._gath)                  # the one exchange step of a read
        if c.rank == self.root:
            c.engine_next(self.engine)
            return self.engine.read_finish(address, keys, c.handle(gathered), prepare_write, download=self.download)
        return None

    def read(self, address, keys):
        return self._read(address, keys, False)

    def read_prepare_write(self, address, keys):
        return self._read(address, keys, True)

    def write(self, w, address, keys):
        c = self.comm
        self.engine.write_begin(address, keys)     # trace(ct_hi) of the local rows + inverse of coordinate 0: no ct_lo needed
        if c.rank == self.root:
            self.engine.write_root(w, address, keys, out=c.handle(self._ctlo))
        c.engine_done(self.engine)                 # root: ct_lo is ready; others: an earlier write_shard has read _ctlo
        c.broadcast(self._ctlo, self.root)                                # the one exchange step of a write
        c.engine_next(self.engine)
        self.engine.write_shard(address, keys, c.handle(self._ctlo))
