"""Population sharding across the GPUs of one node (one process per GPU, torch.distributed).

The candidate step has no candidate<->candidate data flow: within an iteration
every candidate reads (A, b, strategy) and its own state.  So the active candidates are
block-partitioned over the ranks, every rank runs the batched HIP phases for its block only,
and the per-candidate scalar records the host bookkeeping needs (Rayleigh dots, solve status,
norms, residuals -- the inputs of landscape energy / stuckness) plus the updated candidate
rows are exchanged with ONE kind of collective: an all-gather (RCCL over xGMI with the
`nccl` backend; `gloo` in the CPU tests).  The host orchestration is replicated: every rank
holds the same candidate list and consumes the same RNG streams, so bookkeeping is identical
on all ranks by construction.  A is replicated (uploaded by each rank).
"""
from __future__ import annotations

import os

import numpy as np


class PopulationComm:
    def __init__(self, device_tensors: bool | None = None):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.torch = torch
        self.dist = dist
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()
        backend = dist.get_backend()
        self.on_device = (backend == "nccl") if device_tensors is None else device_tensors
        self.device = None
        if self.on_device:
            self.device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        self.collectives = 0
        self.bytes_gathered = 0

    # ---- partition -------------------------------------------------------------------------
    def owners(self, n_items: int) -> np.ndarray:
        """Contiguous block partition of range(n_items) over the ranks (list order preserved)."""
        base, extra = divmod(n_items, self.world)
        sizes = [base + (1 if r < extra else 0) for r in range(self.world)]
        return np.repeat(np.arange(self.world), sizes)

    # ---- collective --------------------------------------------------------------------------
    def allgather_rows(self, local: np.ndarray, counts) -> np.ndarray:
        """local: (counts[rank], width) float64 -> (sum(counts), width), concatenated in rank order."""
        torch = self.torch
        width = local.shape[1] if local.ndim == 2 else 0
        cmax = int(max(counts)) if len(counts) else 0
        if cmax == 0 or width == 0:
            return np.zeros((int(sum(counts)), width))
        buf = np.zeros((cmax, width), dtype=np.float64)
        buf[: local.shape[0]] = local
        t = torch.from_numpy(buf)
        if self.on_device:
            t = t.to(self.device)
        out = torch.empty((self.world * cmax, width), dtype=torch.float64, device=t.device)
        self.dist.all_gather_into_tensor(out, t)
        self.collectives += 1
        self.bytes_gathered += out.numel() * 8
        full = out.cpu().numpy().reshape(self.world, cmax, width)
        return np.concatenate([full[r, : counts[r]] for r in range(self.world)], axis=0)

    # ---- candidate rows, device to device ---------------------------------------------------------
    class _DevArray:
        """__cuda_array_interface__ view of a population array of the library's context (float64 pairs)."""
        def __init__(self, ptr, rows, cols):
            self.__cuda_array_interface__ = {"shape": (rows, cols), "typestr": "<f8", "data": (ptr, False), "version": 2}

    def sync_rows_device(self, ctx, which: int, slots_by_rank, length: int) -> None:
        """After a sharded step every rank holds fresh rows only for its own candidates.  All-gather them over RCCL straight
        out of / into the contexts' population arrays (one gather kernel, one all-gather, one scatter kernel) -- no host
        bounce.  slots_by_rank[r] = the slots rank r updated, in list order (identical on every rank)."""
        torch = self.torch
        ptr, ld, cap = ctx.pop_device_ptr(which)                 # joins the context's stream
        X = torch.as_tensor(PopulationComm._DevArray(ptr, cap, 2 * ld), device=self.device)
        cmax = max((len(s) for s in slots_by_rank), default=0)
        if cmax == 0:
            return
        w = 2 * length
        send = torch.zeros((cmax, w), dtype=torch.float64, device=self.device)
        mine = slots_by_rank[self.rank]
        if len(mine):
            send[: len(mine)] = X[torch.as_tensor(mine, dtype=torch.long, device=self.device), :w]
        out = torch.empty((self.world * cmax, w), dtype=torch.float64, device=self.device)
        self.dist.all_gather_into_tensor(out, send)
        self.collectives += 1
        self.bytes_gathered += out.numel() * 8
        for r, sl in enumerate(slots_by_rank):
            if len(sl):                                           # own rows too: the same bytes, one code path
                X[torch.as_tensor(sl, dtype=torch.long, device=self.device), :w] = out[r * cmax: r * cmax + len(sl)]
        torch.cuda.current_stream(self.device).synchronize()     # the library's stream may read the rows from here on

    def barrier(self):
        self.dist.barrier()


def init_from_env(backend: str | None = None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun) if world > 1."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and os.environ.get("MAUS_FORCE_COMM", "0") != "1":     # MAUS_FORCE_COMM=1: a one-rank group (RCCL smoke test)
        return None
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            # torch's wheel bundles its own HIP runtime under the same soname as /opt/rocm's (libamdhip64.so.7).  Loaded
            # first, it also serves libmaus_hip.so; loaded second it finds the device already held by the other runtime
            # ("No HIP GPUs are available", tools/probe_torch_hip_order.py).  So: process group before the first context.
            if not torch.cuda.is_available():
                from . import _cabi
                hint = (" -- libmaus_hip.so was loaded before torch initialised its HIP runtime: call dist.init_from_env() "
                        "before creating a Context / MAUS_Solver") if _cabi._lib is not None else ""
                raise RuntimeError("backend 'nccl' needs torch.cuda, which sees no GPU" + hint)
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend=backend)
    return PopulationComm()
