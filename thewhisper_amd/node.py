"""Node-level front end: one serving process per MI355X, sessions sticky to a GPU (BASELINE config 4: 128 streaming
sessions on the 8 GPUs of a node, 16 per GPU).

Replaces the process layout of the reference's demo server (R:examples/server.py:22-115: ONE process, ONE shared
pipeline, one request at a time).  Streams are independent and the model fits a GPU ninety times over, so the layout is
SURVEY.md section 8e's: a full replica per GPU, ``session -> rank = session_index % world`` (``dist.shard_streams``; the
per-session scheduler state and the session's place in the hub's passes stay local to that rank), no collective anywhere -
the ranks never talk to each other.  The HTTP process (``gateway.create_app(NodeRouter(...))``) only routes:

    front process (FastAPI)  --pipe-->  worker rank r  =  SessionHost(BatchingHub(AMDWhisperBackend on cuda:r))

Each worker serves its pipe from a small thread pool so that the requests of its sessions meet in the hub's passes exactly
as they do in the single-GPU gateway.  ``python -m thewhisper_amd.gateway --gpus N ...`` builds this.

A worker that dies (a GPU fault takes its process down) takes its sessions with it - their scheduler state lived there: requests
in flight fail with ``WorkerGone`` (HTTP 503), the sessions answer 404 from then on, new sessions and stateless requests go to the
surviving ranks, ``health()`` reports which ranks are alive.  Workers are not respawned here: that is the process supervisor's job.
"""
from __future__ import annotations

import importlib
import itertools
import multiprocessing as mp
import os
import threading
from concurrent.futures import Future, ThreadPoolExecutor
from concurrent.futures import TimeoutError as _FutureTimeout
from typing import Any, Dict, Optional, Tuple

import numpy as np

__all__ = ["NodeRouter", "WorkerGone", "worker_main"]


def _resolve(spec: str):
    mod, fn = spec.split(":")
    return getattr(importlib.import_module(mod), fn)


def worker_main(rank: int, world: int, conn, host_factory: str, factory_kwargs: Dict[str, Any], threads: int = 32):
    """Entry point of one worker process.  ``host_factory`` = "module:function"; called as ``f(rank=, world=, **kwargs)`` it
    returns the rank's ``gateway.SessionHost`` (building the backend on ITS GPU).  Protocol on ``conn``: requests
    ``(req_id, op, args)``, replies ``(req_id, ok, payload)``; ``op`` = a ``SessionHost`` method name, or "stop"."""
    os.environ["THEWHISPER_RANK"] = str(rank)
    try:
        host = _resolve(host_factory)(rank=rank, world=world, **factory_kwargs)
        conn.send((0, True, {"rank": rank, "sample_rate": host.sample_rate}))
    except BaseException as e:  # noqa: BLE001
        conn.send((0, False, f"worker {rank} failed to start: {e!r}"))
        return
    send_lock = threading.Lock()

    def reply(req_id, ok, payload):
        with send_lock:
            conn.send((req_id, ok, payload))

    def serve(req_id, op, args):
        try:
            reply(req_id, True, getattr(host, op)(*args))
        except BaseException as e:  # noqa: BLE001
            # the argument travels as TEXT: an exception argument that cannot be pickled would raise inside this handler, no
            # reply would ever be sent and the router's caller would wait for ever
            try:
                reply(req_id, False, (type(e).__name__, str(e.args[0]) if e.args else ""))
            except BaseException:  # noqa: BLE001
                try:
                    reply(req_id, False, ("RuntimeError", "worker failed and its error could not be reported"))
                except BaseException:  # noqa: BLE001  (the pipe itself is gone: the router's reader notices)
                    pass

    pool = ThreadPoolExecutor(max_workers=threads, thread_name_prefix=f"tw-rank{rank}")
    while True:
        try:
            req_id, op, args = conn.recv()
        except EOFError:
            break
        if op == "stop":
            reply(req_id, True, None)
            break
        pool.submit(serve, req_id, op, args)
    pool.shutdown(wait=False)
    hub = getattr(host, "hub", None)
    if hub is not None:
        hub.close()


class _Worker:
    def __init__(self, ctx, rank: int, world: int, host_factory: str, factory_kwargs: Dict[str, Any]):
        self.rank = rank
        self.conn, child = ctx.Pipe()
        self.proc = ctx.Process(target=worker_main, args=(rank, world, child, host_factory, factory_kwargs), daemon=True,
                                name=f"thewhisper-rank{rank}")
        self.proc.start()
        child.close()
        self.send_lock = threading.Lock()
        self.pending: Dict[int, Future] = {}
        self.pending_lock = threading.Lock()
        self.hello: Future = Future()
        self.alive = True                    # False once the pipe closed (the process died or was stopped)
        self.reader = threading.Thread(target=self._read, name=f"tw-router-rank{rank}", daemon=True)
        self.reader.start()

    def _read(self):
        while True:
            try:
                req_id, ok, payload = self.conn.recv()
            except (EOFError, OSError):
                err = WorkerGone(f"worker {self.rank} went away")
                with self.pending_lock:
                    self.alive = False
                    futs, self.pending = list(self.pending.values()), {}
                for f in futs:
                    _resolve_future(f, None, err)
                if not self.hello.done():
                    self.hello.set_exception(err)
                return
            if req_id == 0:
                (self.hello.set_result if ok else lambda m: self.hello.set_exception(RuntimeError(m)))(payload)
                continue
            with self.pending_lock:
                fut = self.pending.pop(req_id, None)
            if fut is None:
                continue
            _resolve_future(fut, payload if ok else None, None if ok else _rebuild_error(payload))


def _resolve_future(fut, result, exc):
    """``fut``: a ``concurrent.futures.Future`` (blocking callers) or ``(loop, asyncio.Future)`` (``NodeRouter.acall``: the
    reader thread hands the reply to the event loop, no pool thread is parked per request)."""
    if isinstance(fut, tuple):
        loop, afut = fut

        def done():
            if not afut.done():
                afut.set_exception(exc) if exc is not None else afut.set_result(result)

        try:
            loop.call_soon_threadsafe(done)
        except RuntimeError:      # the loop is closed: nobody is waiting any more
            pass
    elif not fut.done():
        fut.set_exception(exc) if exc is not None else fut.set_result(result)


class WorkerGone(RuntimeError):
    """The serving process of a GPU is no longer there: its sessions are lost (their state lived in that process)."""


def _rebuild_error(payload: Tuple[str, Any]) -> BaseException:
    from .gateway import HostBusy

    name, arg = payload
    if name == "KeyError":
        return KeyError(arg)
    if name == "HostBusy":
        return HostBusy(arg)
    if name == "ValueError":
        return ValueError(arg)
    return RuntimeError(f"{name}: {arg}")


class NodeRouter:
    """The ``SessionHost`` interface over ``world`` worker processes (one per GPU).  Sessions are assigned round-robin at
    creation (``index % world``, the rule of ``dist.shard_streams``) and never move; stateless ``transcribe`` requests are
    spread round-robin."""

    def __init__(self, world: int, host_factory: str, factory_kwargs: Optional[Dict[str, Any]] = None, start_timeout_s: float = 600.0,
                 call_timeout_s: float = 600.0):
        if world < 1:
            raise ValueError("world must be >= 1")
        self.call_timeout_s = float(call_timeout_s)   # a worker that neither answers nor exits (a hung GPU) is reported as gone
        ctx = mp.get_context("spawn")      # a forked child would inherit the parent's HIP state
        self.world = world
        self.workers = [_Worker(ctx, r, world, host_factory, dict(factory_kwargs or {})) for r in range(world)]
        hellos = [w.hello.result(timeout=start_timeout_s) for w in self.workers]
        self.sample_rate = hellos[0]["sample_rate"]
        self._ids = itertools.count(1)
        self._id_lock = threading.Lock()
        self._session_rank: Dict[str, int] = {}
        self._created = 0
        self._rr = 0
        self._lock = threading.Lock()

    # -- plumbing ------------------------------------------------------------------------------------
    def call(self, rank: int, op: str, *args, timeout: Optional[float] = None):
        w = self.workers[rank]
        with self._id_lock:
            req_id = next(self._ids)
        fut: Future = Future()
        with w.pending_lock:
            if not w.alive:
                raise WorkerGone(f"worker {rank} went away")
            w.pending[req_id] = fut
        try:
            with w.send_lock:
                w.conn.send((req_id, op, args))
        except (OSError, ValueError) as e:       # the pipe broke between the check and the send
            with w.pending_lock:
                w.pending.pop(req_id, None)
            raise WorkerGone(f"worker {rank} went away") from e
        try:
            return fut.result(timeout=self.call_timeout_s if timeout is None else timeout)
        except _FutureTimeout as e:
            with w.pending_lock:
                w.pending.pop(req_id, None)
            raise WorkerGone(f"worker {rank} did not answer within {self.call_timeout_s if timeout is None else timeout:.0f} s") from e

    async def acall(self, rank: int, op: str, *args, timeout: Optional[float] = None):
        """``call`` for an asyncio caller (the gateway's routes): the request is written to the pipe from the event loop (a pipe
        write of <= a few tens of KB; the worker's receive loop drains it continuously) and the reply is awaited - no thread of
        the front process waits with it.  Measured with 128 closed-loop sessions on 8 stub ranks: the routing process did
        ~410 calls/s with a pool thread per request (four GIL hand-offs each) and ~2x that this way (tests/test_node_scale.py)."""
        import asyncio

        w = self.workers[rank]
        with self._id_lock:
            req_id = next(self._ids)
        loop = asyncio.get_running_loop()
        afut = loop.create_future()
        with w.pending_lock:
            if not w.alive:
                raise WorkerGone(f"worker {rank} went away")
            w.pending[req_id] = (loop, afut)
        try:
            with w.send_lock:
                w.conn.send((req_id, op, args))
        except (OSError, ValueError) as e:
            with w.pending_lock:
                w.pending.pop(req_id, None)
            raise WorkerGone(f"worker {rank} went away") from e
        t = self.call_timeout_s if timeout is None else timeout
        try:
            return await asyncio.wait_for(afut, t)
        except asyncio.TimeoutError as e:
            with w.pending_lock:
                w.pending.pop(req_id, None)
            raise WorkerGone(f"worker {rank} did not answer within {t:.0f} s") from e

    async def _asession_call(self, sid: str, op: str, *args):
        try:
            return await self.acall(self.rank_of(sid), op, sid, *args)
        except KeyError:
            with self._lock:
                self._session_rank.pop(sid, None)
            raise

    async def aadd_chunk(self, sid: str, audio_np: np.ndarray) -> None:
        return await self._asession_call(sid, "add_chunk", np.ascontiguousarray(audio_np))

    async def aprocess(self, sid: str):
        return await self._asession_call(sid, "process")

    async def aclear(self, sid: str) -> None:
        return await self._asession_call(sid, "clear")

    async def atranscribe(self, audio: np.ndarray, sr: int):
        return await self.acall(self._pick_stateless_rank(), "transcribe", np.ascontiguousarray(audio), sr)

    def _session_call(self, sid: str, op: str, *args):
        """A request of session ``sid`` on its rank.  A worker that no longer knows the session (idle expiry inside the rank's
        SessionHost, or a restart) answers KeyError: the routing entry is dropped with it, so clients that never call /end do not
        leave their ids in this process for ever."""
        try:
            return self.call(self.rank_of(sid), op, sid, *args)
        except KeyError:
            with self._lock:
                self._session_rank.pop(sid, None)
            raise

    def _alive_ranks(self):
        return [w.rank for w in self.workers if w.alive]

    def rank_of(self, sid: str) -> int:
        with self._lock:
            r = self._session_rank.get(sid)
            if r is not None and not self.workers[r].alive:    # the session's state died with its process
                del self._session_rank[sid]
                r = None
        if r is None:
            raise KeyError(sid)
        return r

    # -- SessionHost interface -------------------------------------------------------------------------
    def create(self) -> str:
        import base64

        from .gateway import HostBusy

        sid = base64.urlsafe_b64encode(os.urandom(16)).decode("ascii")
        with self._lock:
            rank = self._created % self.world          # dist.shard_streams: stream_id % world
            self._created += 1
        if not self.workers[rank].alive:               # a GPU's process died: its share of the new sessions goes to the survivors
            alive = self._alive_ranks()
            if not alive:
                raise HostBusy("no serving process is alive")
            rank = alive[rank % len(alive)]
        self.call(rank, "create", sid)
        with self._lock:
            self._session_rank[sid] = rank
        return sid

    def add_chunk(self, sid: str, audio_np: np.ndarray) -> None:
        return self._session_call(sid, "add_chunk", np.ascontiguousarray(audio_np))

    def process(self, sid: str):
        return self._session_call(sid, "process")

    def clear(self, sid: str) -> None:
        return self._session_call(sid, "clear")

    def end(self, sid: str) -> None:
        with self._lock:
            rank = self._session_rank.pop(sid, None)
        if rank is not None and self.workers[rank].alive:
            try:
                self.call(rank, "end", sid)
            except WorkerGone:
                pass

    def _pick_stateless_rank(self) -> int:
        from .gateway import HostBusy

        with self._lock:
            rank = self._rr % self.world
            self._rr += 1
        if not self.workers[rank].alive:
            alive = self._alive_ranks()
            if not alive:
                raise HostBusy("no serving process is alive")
            rank = alive[rank % len(alive)]
        return rank

    def transcribe(self, audio: np.ndarray, sr: int):
        return self.call(self._pick_stateless_rank(), "transcribe", np.ascontiguousarray(audio), sr)

    def health(self) -> Dict[str, Any]:
        per = []
        for r in range(self.world):
            try:
                per.append({**self.call(r, "health", timeout=30), "alive": True})
            except Exception:  # noqa: BLE001  (gone, or not answering within the timeout)
                per.append({"passes": None, "rows": None, "sessions": 0, "vad_launches": None, "alive": False})
        return {"passes": sum(p["passes"] or 0 for p in per), "rows": sum(p.get("rows") or 0 for p in per), "sessions": sum(p["sessions"] for p in per),
                "ranks": per, "world": self.world, "alive": sum(1 for p in per if p["alive"])}

    def close(self):
        for w in self.workers:
            if not w.alive:
                continue
            try:
                self.call(w.rank, "stop", timeout=10)
            except Exception:  # noqa: BLE001
                pass
        for w in self.workers:
            w.proc.join(timeout=20)
            if w.proc.is_alive():
                w.proc.terminate()


def default_host_factory(rank: int, world: int, model: str, chunk_length_s: int = 10, max_batch: int = 16, language: str = "en",
                         use_vad: bool = False, torch_dtype: Optional[str] = None, prefetch_cus: int = 0, **_):  # pragma: no cover - needs weights and GPUs
    """What ``python -m thewhisper_amd.gateway --gpus N`` runs in every worker: the backend of GPU ``rank`` behind a hub."""
    os.environ["THEWHISPER_DEVICE"] = f"cuda:{rank}"
    from .gateway import SessionHost
    from .serving import BatchingHub
    from .streaming import AMDWhisperBackend

    import torch

    dtype = {None: None, "bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[torch_dtype]
    backend = AMDWhisperBackend(model, chunk_length_s=chunk_length_s, language=language, batch_size=max_batch, torch_dtype=dtype)
    vad = None
    if use_vad:
        from .vad import VadService

        vad = VadService(max_streams=1024, device=rank)
    return SessionHost(BatchingHub(backend, max_batch=max_batch, prefetch_cus=prefetch_cus), vad=vad)
