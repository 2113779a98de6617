"""-M multi-sample sharding across the GPUs of one node (one process per GPU).

The path shards by SAMPLE (each alignment file is an independent problem, /root/reference/src/emsar_main.c:380-488),
so there is no data-path collective: torch.distributed is used only to line the ranks up around the timed
region and to take the max of their times.  Backend "nccl" (= RCCL) on GPUs, "gloo" in the CPU tests.
"""
import os


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard(n_samples, rank, world):
    """Sample i runs on rank i mod world (the same rule as emsar-hip -M: sample i -> GPU i mod G)."""
    return list(range(rank, n_samples, world))


class Group:
    """Thin wrapper so that bench.py and the tests share the barrier / reduction code."""

    def __init__(self, backend=None, device=None):
        self.rank, self.world, self.local_rank = env_rank()
        self.dist = None
        self.device = device
        if self.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                kw = {}
                if backend == "nccl" and device is not None:
                    kw["device_id"] = device
                dist.init_process_group(backend or "gloo", **kw)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max(self, values):
        """Element-wise max over ranks of a list of floats."""
        if self.dist is None:
            return list(values)
        import torch
        t = torch.tensor(list(values), dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(x) for x in t]

    def sum(self, values):
        if self.dist is None:
            return list(values)
        import torch
        t = torch.tensor(list(values), dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [float(x) for x in t]

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None
