"""Multi-GPU plumbing for bench.py: one process per GPU (torch.distributed; "nccl" is RCCL on ROCm, "gloo" on CPU).

The data path has NO collective: the FM-index is replicated per GPU and every rank aligns its own contiguous shard of
reads (align_reads_inexact_parallel's static chunking, inexact_match.c:115-116, applied across GPUs).  The only
communication is the timing protocol of the benchmark (barrier, then MAX of the step time and SUM of the work counters)
and the exchange of two checksums per rank for the cross-shard parity check.
"""
import os


def env_rank():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def shard_bounds(n_reads, world, rank):
    """Contiguous shard [lo, hi) of rank: chunk = [rank*n/world, (rank+1)*n/world) like inexact_match.c:115-116."""
    return rank * n_reads // world, (rank + 1) * n_reads // world


class Group:
    def __init__(self, backend=None, device=None):
        self.rank, self.local_rank, self.world = env_rank()
        self.dist = None
        self.device = device
        if self.world > 1:
            import torch
            import torch.distributed as dist
            backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                self.device = torch.device("cuda", self.local_rank)
                dist.init_process_group("nccl", device_id=self.device)
            else:
                self.device = torch.device("cpu")
                dist.init_process_group(backend)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def reduce_step(self, dt, kernel_ms, visits):
        """-> (max step seconds over ranks, max kernel ms over ranks, total visits over ranks)."""
        if self.dist is None:
            return dt, kernel_ms, visits
        import torch
        t = torch.tensor([dt, kernel_ms], dtype=torch.float64, device=self.device)
        v = torch.tensor([visits], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        self.dist.all_reduce(v, op=self.dist.ReduceOp.SUM)
        return float(t[0]), float(t[1]), float(v[0])

    def all_gather_pairs(self, x, y):
        """-> [(x_r, y_r) for every rank r] (two integers per rank; used for the cross-shard parity check, not on the data path)."""
        if self.dist is None:
            return [(int(x), int(y))]
        import torch
        mine = torch.tensor([int(x), int(y)], dtype=torch.int64, device=self.device)
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine)
        return [(int(t[0]), int(t[1])) for t in out]

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
