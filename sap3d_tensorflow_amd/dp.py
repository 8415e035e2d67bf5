"""Control plane of data-parallel runs: one process per GPU, torch.distributed (gloo) only ships the
RCCL unique id and brackets timed regions; the gradient all-reduce itself runs inside libp3dhip
(RCCL on a side stream).  The reference is single-device (train.py:73): nothing to mirror."""
import contextlib
import os
import sys


@contextlib.contextmanager
def _stdout_to_stderr():
    """gloo's C++ side announces its connections on file descriptor 1 ("[Gloo] Rank 0 is connected to 7 peer ranks"); a bench run's
    standard output is ONE JSON line, so whatever the rendezvous prints goes to standard error."""
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


class Plane:
    """rank / world from the torch.distributed.run environment; no-ops when world == 1."""

    def __init__(self, backend="gloo"):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            with _stdout_to_stderr():
                dist.init_process_group(backend, rank=self.rank, world_size=self.world)
                dist.barrier()          # the full mesh connects (and says so) at the first collective
            self.dist = dist

    def share_from_rank0(self, make):
        """make() is called on rank 0 only; every rank gets its result (the RCCL unique id)."""
        if not self.dist:
            return make()
        box = [make() if self.rank == 0 else None]
        self.dist.broadcast_object_list(box, src=0)
        return box[0]

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def max_over_ranks(self, value):
        if not self.dist:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])

    def sum_arrays(self, arrays):
        """In-place sum of numpy arrays over ranks (tests: what the in-library all-reduce computes)."""
        if not self.dist:
            return arrays
        import torch
        for a in arrays:
            t = torch.from_numpy(a)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return arrays

    def shard(self, global_batch):
        """Clips [lo, hi) of the global batch owned by this rank (even split on dim 0)."""
        if global_batch % self.world:
            raise ValueError("global batch %d does not divide over %d ranks" % (global_batch, self.world))
        per = global_batch // self.world
        return self.rank * per, (self.rank + 1) * per

    def close(self):
        if self.dist:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None
