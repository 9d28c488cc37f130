"""Cross-GPU plumbing around the sweep.  The exchange itself -- ONE all-reduce(sum) of the packed sufficient statistics per
statistics pass -- runs INSIDE libdpmmhip.so (dpmm_comm_init + dpmm_step_stats / dpmm_suffstats_host: RCCL on the ctx stream;
with a non-RCCL process group, dpmm_comm_init_host: the same library path over a host all-reduce of the group);
this module ships the RCCL unique id, gathers labels for results, and sums the small evaluation tables.

Replaces the reference's two-level tree reduce of `thin_suff_stats` dicts
(create_suff_stats_dict_node_leader / update_suff_stats_posterior!,
src/local_clusters_actions.jl:171-254; aggregate_suff_stats, priors/niw.jl:64-66,
priors/multinomial_prior.jl:41-43).  One process per GPU; torch.distributed is used as plumbing
only (backend "nccl" == RCCL over xGMI on the GPU box; "gloo" in the CPU tests).
N counts travel as Float64 integers (exact below 2^53), so one dtype, one collective.
"""
import os

import numpy as np

from .sampler import LocalComm


class TorchDistComm:
    def __init__(self, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.backend = dist.get_backend()
        self.device = int(os.environ.get("LOCAL_RANK", 0)) if device is None else device

    def attach(self, worker):
        """Multi-GPU: attach the collective INSIDE libdpmmhip.so; from then on the worker's statistics calls return rows summed
        over all ranks.  Backend "nccl": an RCCL communicator on the ctx stream (dpmm_comm_init; torch.distributed only ships the
        128-byte unique id).  Any other backend (gloo: hosts without a GPU fabric, ranks sharing one GPU): the library stages its
        device buffers through pinned memory and this group's all_reduce sums them (dpmm_comm_init_host) -- same library code path
        around the transport.  Never silently unattached: a multi-rank run whose statistics stay local would diverge."""
        if self.world == 1 and self.backend != "nccl":
            return
        if self.backend == "nccl":
            uid = [worker.comm_unique_id() if self.rank == 0 else None]
            self.dist.broadcast_object_list(uid, src=0)
            worker.comm_init(uid[0], self.rank, self.world)
            return
        if not hasattr(worker, "comm_init_host"):
            raise RuntimeError(f"backend {self.backend!r}: the worker has no host-transport attachment and world = {self.world}")

        def allreduce(arr):          # arr: numpy view of the library's pinned staging -> summed in place
            self.dist.all_reduce(self.torch.from_numpy(arr))
        worker.comm_init_host(self.rank, self.world, allreduce)

    def allreduce_np(self, arr):
        """Sum of a Float64 numpy array over the ranks (CPU test workers; the GPU path reduces inside libdpmmhip.so)."""
        t = self.torch.from_numpy(np.ascontiguousarray(arr, np.float64).copy())
        if self.backend == "nccl":
            t = t.to(f"cuda:{self.device}")
        self.dist.all_reduce(t)
        return t.cpu().numpy()

    def allgather_bytes(self, buf):
        objs = [None] * self.world
        self.dist.all_gather_object(objs, bytes(buf))
        return objs

    def gather_labels(self, worker):
        lab, sub = worker.get_labels()
        objs = [None] * self.world
        self.dist.all_gather_object(objs, (lab, sub))
        return np.concatenate([o[0] for o in objs]), np.concatenate([o[1] for o in objs])

    def reduce_counts(self, table):
        t = self.torch.from_numpy(np.ascontiguousarray(table, np.int64))
        if self.backend == "nccl":
            t = t.to(f"cuda:{self.device}")
        self.dist.all_reduce(t)
        return t.cpu().numpy()

    def reduce_f64(self, arr, op="sum"):
        """All-reduce of a small Float64 vector (`op` in sum / min / max): the 1-D 2-means sums and the percentile
        extremes of smart splits."""
        t = self.torch.from_numpy(np.ascontiguousarray(arr, np.float64).copy())
        if self.backend == "nccl":
            t = t.to(f"cuda:{self.device}")
        rop = {"sum": self.dist.ReduceOp.SUM, "min": self.dist.ReduceOp.MIN, "max": self.dist.ReduceOp.MAX}[op]
        self.dist.all_reduce(t, op=rop)
        return t.cpu().numpy()

    def broadcast_int(self, v):
        obj = [int(v)]
        self.dist.broadcast_object_list(obj, src=0)
        return obj[0]


def default_comm():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return TorchDistComm()
    except ImportError:
        pass
    return LocalComm()
