"""Population sharding across the GPUs of one node (one process per GPU).

The candidate step has no candidate<->candidate data flow: within an iteration every candidate reads
(A, b, strategy) and its own state (AMS:576).  So the active candidates are block-partitioned over the ranks in list
order, every rank runs the batched HIP phases for its block only, and the per-candidate scalar records the host
bookkeeping needs (Rayleigh dots, solve status, norms, residuals -- the inputs of landscape energy / stuckness,
AMS:424-475) plus the updated candidate rows are exchanged with ONE kind of collective: an all-gather.  The host
orchestration is replicated: every rank holds the same candidate list and consumes the same RNG streams, so
bookkeeping is identical on all ranks by construction.  A is replicated (uploaded by each rank); what only one rank
needs to compute (start-up diagnostics, the Hermitian eigendecomposition) is computed by rank 0 and broadcast.

Two transports behind one interface:

  'rccl'  (GPUs)  the collectives of libmaus_hip itself (csrc/comm.hip: RCCL over xGMI on the context's own stream).
                  No torch in the process: the ranks are started by any launcher that sets RANK / LOCAL_RANK /
                  WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run does); the 128-byte RCCL id travels
                  from rank 0 to the others over a socket on MASTER_ADDR.
  'gloo'  (CPU)   torch.distributed with the gloo backend: the multi-process tests (tests/test_dist_gloo.py, over the
                  NumPy device double) and rehearsals of several ranks on one GPU.
"""
from __future__ import annotations

import contextlib
import os
import pickle
import socket
import struct
import time

import numpy as np

_MAGIC = b"MAUSRCCL1"


def _token() -> bytes:
    """What tells this job's id server from anything else that may listen near MASTER_PORT.  MAUS_JOB_SECRET -- a random string
    the launcher exports to every rank (bench.py's self-launch does) -- when there is one; otherwise the launcher's run id."""
    sec = os.environ.get("MAUS_JOB_SECRET", "")
    return (sec + "|" + os.environ.get("TORCHELASTIC_RUN_ID", "") + ":" + os.environ.get("MASTER_PORT", "") + ":" +
            os.environ.get("WORLD_SIZE", "")).encode()


def _recv_exact(conn, n: int) -> bytes:
    buf = b""
    while len(buf) < n:
        chunk = conn.recv(n - len(buf))
        if not chunk:
            break
        buf += chunk
    return buf


def _exchange_unique_id(rank: int, world: int, make_id, timeout: float = 300.0) -> bytes:
    """Rank 0 creates the RCCL unique id and serves it to the other ranks of this node over TCP.  The port is the first
    free one above MASTER_PORT (the launcher's own store listens ON MASTER_PORT); clients probe the same range.  Hello =
    magic + rank + token length + token, read in full; the reply echoes the client's rank and a digest of the token before the
    id, so neither side believes a foreign peer that merely knows the magic."""
    import hashlib
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    base = int(os.environ.get("MASTER_PORT", "29500"))
    ports = [base + 1 + k for k in range(32)]
    tok = _token()
    proof = hashlib.sha256(b"reply" + tok).digest()[:16]
    if world == 1:
        return make_id()
    head = len(_MAGIC) + 8
    if rank == 0:
        uid = make_id()
        srv = None
        for p in ports:
            try:
                srv = socket.create_server((addr, p), reuse_port=False)
                break
            except OSError:
                continue
        if srv is None:
            raise RuntimeError(f"dist: no free port in {ports[0]}..{ports[-1]} on {addr} for the RCCL id exchange")
        srv.settimeout(timeout)
        served = set()
        deadline = time.time() + timeout
        try:
            while len(served) < world - 1:
                if time.time() > deadline:
                    raise RuntimeError(f"dist: only {len(served)} of {world - 1} ranks fetched the RCCL id within {timeout:.0f} s")
                conn, _ = srv.accept()
                with conn:
                    conn.settimeout(10.0)
                    try:
                        hello = _recv_exact(conn, head)
                        if len(hello) != head or not hello.startswith(_MAGIC):
                            continue
                        r, ln = struct.unpack("<ii", hello[len(_MAGIC):])
                        if not (0 <= ln <= 4096) or _recv_exact(conn, ln) != tok or not (0 < r < world):
                            continue
                        conn.sendall(_MAGIC + struct.pack("<i", r) + proof + uid)
                        served.add(r)
                    except OSError:
                        continue
        finally:
            srv.close()
        return uid
    deadline = time.time() + timeout
    hello = _MAGIC + struct.pack("<ii", rank, len(tok)) + tok
    want = len(_MAGIC) + 4 + len(proof) + 128
    while time.time() < deadline:
        for p in ports:
            try:
                with socket.create_connection((addr, p), timeout=2.0) as s:
                    s.settimeout(5.0)
                    s.sendall(hello)
                    buf = _recv_exact(s, want)
                    if (len(buf) == want and buf.startswith(_MAGIC) and struct.unpack("<i", buf[len(_MAGIC):len(_MAGIC) + 4])[0] == rank
                            and buf[len(_MAGIC) + 4:len(_MAGIC) + 4 + len(proof)] == proof):
                        return buf[-128:]
            except OSError:
                continue
        time.sleep(0.2)
    raise RuntimeError(f"dist: rank {rank} could not fetch the RCCL id from rank 0 on {addr}:{ports[0]}..{ports[-1]}")


class RootFailure(RuntimeError):
    """Work that one rank does for all of them (start-up diagnostics, the Hermitian decomposition) failed there."""


class PopulationComm:
    """rank / world, the block partition, and the collectives of a sharded run.  `transport`: 'rccl' or 'gloo'."""

    def __init__(self, transport: str | None = None, device_tensors: bool | None = None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank)))
        if transport is None:
            transport = "gloo" if device_tensors is False else _default_transport()
        if transport not in ("rccl", "gloo"):
            raise ValueError(f"PopulationComm: unknown transport {transport!r}")
        self.transport = transport
        self.on_device = transport == "rccl"     # candidate rows travel device to device; one rank per GPU is guaranteed
        self.in_root_call = False
        self.collectives = 0
        self.bytes_gathered = 0
        self.seconds = 0.0                       # host wall time inside collectives (bench: per-rank report)
        self.ctx = None
        self._uid = None
        if transport == "gloo":
            import torch
            import torch.distributed as dist
            if not dist.is_initialized():
                raise RuntimeError("torch.distributed is not initialised (dist.init_from_env('gloo'))")
            self.torch, self.dist = torch, dist
            self.rank, self.world = dist.get_rank(), dist.get_world_size()
        else:
            from . import _cabi
            self._uid = _exchange_unique_id(self.rank, self.world, _cabi.comm_unique_id)

    # ---- binding to the device context (rccl: the communicator lives in the library, on the context's stream) ----------
    def attach(self, ctx) -> None:
        if self.transport != "rccl" or ctx is self.ctx:
            return
        if self.ctx is not None:
            raise RuntimeError("PopulationComm: already attached to another context (one communicator per context)")
        ctx.comm_init(self.rank, self.world, self._uid)
        self.ctx = ctx

    def _need_ctx(self):
        if self.ctx is None:
            raise RuntimeError("PopulationComm('rccl') is not attached to a device context yet (DeviceEngine(comm=...) does it)")
        return self.ctx

    # ---- partition -------------------------------------------------------------------------
    def owners(self, n_items: int) -> np.ndarray:
        """Contiguous block partition of range(n_items) over the ranks (list order preserved)."""
        base, extra = divmod(n_items, self.world)
        sizes = [base + (1 if r < extra else 0) for r in range(self.world)]
        return np.repeat(np.arange(self.world), sizes)

    # ---- collectives --------------------------------------------------------------------------
    def _account(self, t0, nbytes):
        self.collectives += 1
        self.bytes_gathered += int(nbytes)
        self.seconds += time.perf_counter() - t0

    def allgather_rows(self, local: np.ndarray, counts) -> np.ndarray:
        """local: (counts[rank], width) float64 -> (sum(counts), width), concatenated in rank order."""
        width = local.shape[1] if local.ndim == 2 else 0
        cmax = int(max(counts)) if len(counts) else 0
        if cmax == 0 or width == 0:
            return np.zeros((int(sum(counts)), width))
        t0 = time.perf_counter()
        buf = np.zeros((cmax, width), dtype=np.float64)
        buf[: local.shape[0]] = local
        if self.transport == "rccl":
            full = self._need_ctx().comm_allgather_records(buf, self.world)
        else:
            t = self.torch.from_numpy(buf)
            out = self.torch.empty((self.world * cmax, width), dtype=self.torch.float64)
            self.dist.all_gather_into_tensor(out, t)
            full = out.numpy().reshape(self.world, cmax, width)
        self._account(t0, self.world * cmax * width * 8)
        return np.concatenate([full[r, : counts[r]] for r in range(self.world)], axis=0)

    def sync_rows_device(self, ctx, which: int, slots_by_rank, length: int) -> None:
        """After a sharded step every rank holds fresh rows only for its own candidates.  slots_by_rank[r] = the slots rank
        r updated, in list order (identical on every rank): packed, all-gathered over RCCL and scattered straight into the
        context's population array -- one pack kernel, one all-gather, one unpack kernel, no host bounce."""
        if self.transport != "rccl":
            raise RuntimeError("sync_rows_device needs the 'rccl' transport")
        t0 = time.perf_counter()
        self._need_ctx().comm_allgather_rows(which, slots_by_rank, length)
        self._account(t0, sum(len(s) for s in slots_by_rank) * length * 16)

    def bcast_array(self, arr: np.ndarray, root: int = 0) -> np.ndarray:
        """In-place broadcast of a C-contiguous array that has the same shape and dtype on every rank."""
        arr = np.ascontiguousarray(arr)
        if arr.nbytes == 0:
            return arr
        t0 = time.perf_counter()
        if self.transport == "rccl":
            self._need_ctx().comm_bcast(arr, root)
        else:
            t = self.torch.from_numpy(arr.view(np.uint8).reshape(-1))
            self.dist.broadcast(t, src=root)
        self._account(t0, arr.nbytes)
        return arr

    def bcast_object(self, obj, root: int = 0):
        """A picklable object from `root` to every rank (start-up diagnostics)."""
        payload = pickle.dumps(obj) if self.rank == root else b""
        ln = self.bcast_array(np.array([len(payload)], dtype=np.int64), root)
        buf = np.frombuffer(payload, dtype=np.uint8).copy() if self.rank == root else np.empty(int(ln[0]), dtype=np.uint8)
        self.bcast_array(buf, root)
        return obj if self.rank == root else pickle.loads(buf.tobytes())

    def root_call(self, fn, root: int = 0):
        """fn() on `root` only, its (picklable) result on every rank.  Whatever fn raises on the root -- a device allocation
        that fails, MemoryError, an error of the library -- is raised on EVERY rank as RootFailure: the status travels first,
        so no rank is left waiting inside a collective that the root never enters."""
        status = None
        if self.rank == root:
            self.in_root_call = True            # (engine.bind_matrix: no collective from inside work the others do not take part in)
            try:
                status = ("ok", fn())
            except Exception as e:                                  # noqa: BLE001 -- the point is that nothing escapes un-broadcast
                status = ("err", f"{type(e).__name__}: {e}")
            finally:
                self.in_root_call = False
        status = self.bcast_object(status, root)
        if status[0] != "ok":
            raise RootFailure(f"rank {root} failed in work it does for all ranks: {status[1]}")
        return status[1]

    def bcast_eigvecs(self, ctx, evecs, n: int, root: int = 0) -> None:
        """The eigenvector matrix of the Hermitian shortcut (AMS:161): decomposed by `root` only, resident on every rank
        afterwards.  rccl: uploaded once and broadcast device to device; gloo: host broadcast, then each rank uploads.  The
        root's own preparation (an upload or a read-back of n x n) runs under root_call: its failure reaches every rank."""
        if self.transport == "rccl":
            t0 = time.perf_counter()
            # evecs None: already resident on the root's device (device_eigh)
            self.root_call(lambda: ctx.set_eigvecs(evecs) if evecs is not None else None, root)
            self._need_ctx().comm_bcast_eigvecs(n, root)
            self._account(t0, 16 * n * n)
            return
        held = {}

        def fetch():
            held["V"] = np.ascontiguousarray(evecs if evecs is not None else ctx.get_eigvecs(), dtype=np.complex128)
        self.root_call(fetch, root)
        V = held["V"] if self.rank == root else np.empty((n, n), dtype=np.complex128)
        self.bcast_array(V.view(np.float64), root)
        ctx.set_eigvecs(V)

    def max_over_ranks(self, value: float) -> float:
        return float(self.allgather_rows(np.array([[float(value)]]), [1] * self.world).max())

    def barrier(self):
        if self.transport == "rccl":
            self.allgather_rows(np.zeros((1, 1)), [1] * self.world)
        else:
            self.dist.barrier()

    def stats(self):
        return {"collectives": self.collectives, "bytes": self.bytes_gathered, "ms": self.seconds * 1e3}

    @contextlib.contextmanager
    def all_blas_threads(self):
        """Work that only ONE rank does while the others wait (start-up diagnostics, the Hermitian eigh) gets the node's
        cores: launchers such as torch.distributed.run export OMP_NUM_THREADS=1 to every rank, which would run a 73-second
        decomposition (n = 8192) on one thread."""
        try:
            from threadpoolctl import threadpool_limits
        except Exception:
            yield
            return
        from .engine import cpu_allowance
        with threadpool_limits(limits=cpu_allowance()):       # (not the visible cores: engine.cap_blas_threads)
            yield


def _default_transport() -> str:
    env = os.environ.get("MAUS_DIST_BACKEND")
    if env:
        return {"nccl": "rccl", "rccl": "rccl", "gloo": "gloo"}[env]
    from . import _cabi
    return "rccl" if _cabi.device_count() > 0 else "gloo"


def init_from_env(backend: str | None = None):
    """The communicator of this process from RANK / WORLD_SIZE / MASTER_* (as set by torch.distributed.run or any other
    one-process-per-GPU launcher), or None for a single rank.  backend: 'rccl' (alias 'nccl') | 'gloo' | None = rccl when
    this process sees a GPU, gloo otherwise.  MAUS_FORCE_COMM=1 builds a one-rank communicator (RCCL smoke test)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and os.environ.get("MAUS_FORCE_COMM", "0") != "1":
        return None
    transport = {"nccl": "rccl", "rccl": "rccl", "gloo": "gloo", None: None}[backend]
    if transport is None:
        transport = _default_transport()
    if transport == "gloo":
        import torch.distributed as dist
        if not dist.is_initialized():
            dist.init_process_group(backend="gloo")
    return PopulationComm(transport)
