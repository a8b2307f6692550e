"""Row-sharded consistency matrix across the GPUs of one node.

One process per GPU (torch.distributed, backend "nccl" = RCCL on ROCm).  The chain and the
candidate list are replicated; rank r solves the cells (i, j >= i) of the rows the library's own
assignment gives it (ipc_row_assignment: balanced by cost -- a row's cost is the number of poses its
cells sweep -- costliest row first to the least loaded rank; IPC_ROW_BALANCE=cyclic keeps i % world),
producing bit rows [rows_per_rank, words] (row i sits at slot[i] % rows_per_rank of its owner's shard).
ipc_solve_rows blocks the host once per step (the counts of the cells to solve again; the first step of a candidate
list once more for the plan, see include/ipc_amd.h) and returns with everything else enqueued.  ONE all-gather of those bit rows over xGMI reassembles the matrix on
every rank (N^2/8 bytes in total: 197 KB for C2, 78 MB for C5), after which every rank
assembles the symmetric matrix and runs the (cheap, sequential-in-k) set-max redundantly, so no
second collective is needed to publish the result.

The `backend` object is what does the solving: in production it is the HIP engine
(ipc_amd.consensus.IPC); the world_size-2 gloo tests on CPU plug in a stand-in that produces
the same shard layout, which is how the sharding / gather / reassembly logic is covered
without a GPU.
"""
import contextlib

import numpy as np
import torch
import torch.distributed as dist


class EngineBackend:
    """HIP engine over device pointers; tensors live on the engine's GPU."""

    def __init__(self, engine):
        self.e = engine
        self.device = torch.device("cuda", engine.device)
        self.N, self.words = engine.N, engine.words
        # A stream of its own, made torch's current stream for the duration of a step: torch's
        # default stream is the NULL stream, whose handle (0) the C ABI reads as "the engine's own
        # non-blocking stream" -- work launched there is NOT ordered with the collective, which
        # torch orders against its current stream.
        self.stream = torch.cuda.Stream(device=self.device)

    def stream_ctx(self):
        return torch.cuda.stream(self.stream)

    def empty_words(self, n):
        return torch.empty(n, dtype=torch.int64, device=self.device)

    def empty_bytes(self, n):
        return torch.empty(n, dtype=torch.uint8, device=self.device)

    def _stream(self):
        h = torch.cuda.current_stream(self.device).cuda_stream
        if not h:
            raise RuntimeError("EngineBackend must run under its own stream (use stream_ctx())")
        return h

    def solve_rows(self, rank, world, upper):
        self.e.solve_rows(rank, world, upper.data_ptr(), self._stream())

    def assemble(self, gathered, world, bits):
        self.e.assemble_matrix(gathered.data_ptr(), world, bits.data_ptr(), self._stream())

    def set_max(self, bits, accepted):
        self.e.set_max(bits.data_ptr(), accepted.data_ptr(), self._stream())


class ShardedMatrix:
    def __init__(self, backend, rank=0, world=1, group=None, force_gather=False):
        """force_gather: go through the collective even at world 1 (a 1-rank RCCL group on one GPU executes the same
        all_gather_into_tensor call, stream ordering and tensor views as an 8-rank node does -- tests/test_gpu_dist_rccl.py)."""
        self.b, self.rank, self.world, self.group = backend, rank, world, group
        self.gather = world > 1 or force_gather
        N, words = backend.N, backend.words
        self.rpr = (N + world - 1) // world
        self.upper = backend.empty_words(self.rpr * words)
        self.gathered = backend.empty_words(world * self.rpr * words) if self.gather else self.upper
        self.bits = backend.empty_words(N * words)
        self.accepted = backend.empty_bytes(N)

    def step(self):
        """One pass of the hot path: solve my rows, all-gather, assemble, set-max; everything is
        enqueued on the backend's stream (the collective is ordered against it by torch)."""
        ctx = self.b.stream_ctx() if hasattr(self.b, "stream_ctx") else contextlib.nullcontext()
        with ctx:
            self.b.solve_rows(self.rank, self.world, self.upper)
            if self.gather:
                dist.all_gather_into_tensor(self.gathered, self.upper, group=self.group)
            self.b.assemble(self.gathered, self.world, self.bits)
            self.b.set_max(self.bits, self.accepted)

    def result(self):
        N, words = self.b.N, self.b.words
        if hasattr(self.b, "stream"):
            self.b.stream.synchronize()
        bits = self.bits.cpu().numpy().view(np.uint64).reshape(N, words)
        return bits, self.accepted.cpu().numpy()
