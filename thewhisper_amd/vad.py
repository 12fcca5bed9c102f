"""Voice-activity gate on the MI355X with silero-vad's calling contract.

The reference gates ``StreamingPipeline.add_new_chunk`` with silero-vad loaded through ``torch.hub`` - a network download -
and calls it as ``prob = self.vad_model(torch_frame_512, 16000).item()`` on consecutive 512-sample frames, the model keeping
state between calls (R:thestage_speechkit/streaming/streaming_pipeline.py:533-538, :589-622).  ``EnergyVAD`` honours that
contract with a different detector: an adaptive-noise-floor energy rule (stated in csrc/k_vad.hip, restated in
oracle/whisper_oracle.py::energy_vad) computed by ``tw_vad_energy``.  It is NOT silero and makes no claim of equal
decisions; it exists so that the default ``use_vad=True`` path of the reference scheduler can run offline and so that a
serving tick can gate all sessions in one launch (``BatchedVAD``).

    sp = StreamingPipeline(backend=..., use_vad=False)       # constructor must not reach torch.hub
    attach_vad(sp)                                           # sp.vad_model = EnergyVAD(); sp.use_vad = True

Serving (``VadService``): the reference calls the detector once per 512-sample frame and synchronises on every result
(``.item()``); with N sessions that is N x 31 launches + syncs per second of audio on the request threads.  ``VadService``
owns ONE ``BatchedVAD`` with a state slot per session; a session's ``VadStream`` is the silero-shaped object its scheduler
calls, but ``prefetch`` evaluates ALL frames an ``add_new_chunk`` is about to ask for in one request, and the service thread
answers the requests of all sessions that arrived within its window with one launch per frame count (the gateway's
``add_chunk`` route does exactly that, gateway.py).
"""
from __future__ import annotations

import ctypes as C
import collections
import queue
import threading
import time
from concurrent.futures import Future
from typing import Callable, Deque, List, Optional, Tuple

import numpy as np
import torch

from . import _cabi

FRAME = 512


class BatchedVAD:
    """``n_streams`` independent detectors; ``probs(frames)`` runs the next frame(s) of every stream in ONE launch."""

    def __init__(self, n_streams: int = 1, device: int = 0):
        if not torch.cuda.is_available():
            raise RuntimeError("thewhisper_amd.vad needs an MI355X (no CPU fallback)")
        self.lib = _cabi.load_library()
        self.n = int(n_streams)
        self.device = torch.device("cuda", device)
        self.state = torch.zeros((self.n, 2), dtype=torch.float32, device=self.device)

    def reset_states(self, stream: Optional[int] = None):
        if stream is None:
            self.state.zero_()
        else:
            self.state[stream].zero_()

    def probs(self, pcm) -> torch.Tensor:
        """pcm: float32 [n_streams, k*512] (numpy or torch, host or device) -> device tensor [n_streams, k] of speech
        probabilities; the per-stream state advances by k frames."""
        x = torch.as_tensor(pcm, dtype=torch.float32)
        if x.dim() == 1:
            x = x[None]
        if x.shape[0] != self.n or x.shape[1] < FRAME or x.shape[1] % FRAME:
            raise ValueError(f"expected [{self.n}, k*{FRAME}] samples, got {tuple(x.shape)}")
        x = x.to(self.device).contiguous()
        k = x.shape[1] // FRAME
        out = torch.empty((self.n, k), dtype=torch.float32, device=self.device)
        st = torch.cuda.current_stream(self.device).cuda_stream
        stride0 = x.stride(0) if self.n > 1 else x.shape[1]   # a size-1 dimension of a contiguous tensor may carry any stride
        rc = self.lib.tw_vad_energy(self.device.index or 0, C.c_void_p(x.data_ptr()), stride0, self.n, k,
                                    C.c_void_p(self.state.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(int(st) if st else None))
        if rc != 0:
            raise RuntimeError(f"tw_vad_energy failed ({rc}): {self.lib.tw_last_error(None).decode()}")
        return out


class EnergyVAD:
    """Single-stream detector with silero's call signature: ``vad(frame_512, sampling_rate) -> 0-dim tensor`` (``.item()``)."""

    def __init__(self, device: int = 0):
        self._b = BatchedVAD(1, device)

    def reset_states(self):
        self._b.reset_states()

    def __call__(self, x, sampling_rate: int = 16000) -> torch.Tensor:
        if sampling_rate != 16000:
            raise ValueError("EnergyVAD expects 16 kHz audio")
        x = torch.as_tensor(x, dtype=torch.float32).reshape(-1)
        if x.numel() != FRAME:
            raise ValueError(f"EnergyVAD expects {FRAME}-sample frames (as silero-vad at 16 kHz), got {x.numel()}")
        return self._b.probs(x[None])[0, 0]


def attach_vad(streaming_pipeline, vad=None):
    """Switch the voice-activity gate of a reference ``StreamingPipeline`` (constructed with ``use_vad=False`` so that its
    constructor does not call torch.hub) to ``vad`` (default: a fresh ``EnergyVAD``)."""
    streaming_pipeline.vad_model = vad if vad is not None else EnergyVAD()
    streaming_pipeline.use_vad = True
    return streaming_pipeline


class VadStream:
    """One session's detector (silero's calling contract) backed by a slot of a ``VadService``."""

    def __init__(self, service: "VadService", slot: int):
        self._svc = service
        self.slot = slot
        self._ready: Deque[Tuple[np.ndarray, float]] = collections.deque()   # (frame, probability) evaluated ahead of the calls
        self.launch_requests = 0    # requests this stream sent to the service (1 per add_chunk when prefetched; tests)

    def prefetch(self, samples: np.ndarray) -> int:
        """Evaluate every complete 512-sample frame of ``samples`` (the scheduler's left-over VAD buffer + the new chunk) in ONE
        request; the following ``__call__``s on exactly these frames, in order, are answered from the result.  Returns the
        number of frames evaluated."""
        x = np.ascontiguousarray(np.asarray(samples, dtype=np.float32).reshape(-1))
        k = len(x) // FRAME
        if k == 0:
            return 0
        x = x[: k * FRAME]
        p = self._svc.submit(self.slot, x).result()
        self.launch_requests += 1
        for i in range(k):
            self._ready.append((x[i * FRAME : (i + 1) * FRAME], float(p[i])))
        return k

    def __call__(self, x, sampling_rate: int = 16000) -> torch.Tensor:
        if sampling_rate != 16000:
            raise ValueError("VadStream expects 16 kHz audio")
        f = np.asarray(x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else x, dtype=np.float32).reshape(-1)
        if f.size != FRAME:
            raise ValueError(f"expected {FRAME}-sample frames (as silero-vad at 16 kHz), got {f.size}")
        if self._ready:
            g, p = self._ready.popleft()
            if np.array_equal(g, f):
                return torch.tensor(p)
            # the caller is NOT replaying the prefetched frames (the detector state already advanced over them): drop them
            self._ready.clear()
            raise RuntimeError("VadStream: frames were prefetched but a different frame was asked for; call reset_states()")
        self.launch_requests += 1
        return torch.tensor(float(self._svc.submit(self.slot, f).result()[0]))

    def reset_states(self):
        self._ready.clear()
        self._svc.reset(self.slot)

    def close(self):
        self._svc.release(self.slot)


class VadService:
    """All sessions' voice-activity detectors of one GPU behind one worker thread: requests (slot, k frames) that arrive within
    ``window_s`` are grouped by k and every group is ONE ``tw_vad_energy`` launch (state gathered / scattered by slot).
    ``kernel(pcm [g, k*512] float32 host, state [g, 2] float32 host) -> (probs [g, k], new state)`` replaces the GPU launch in
    the GPU-less tests (the numpy restatement of the same rule)."""

    def __init__(self, max_streams: int = 64, device: int = 0, window_s: float = 0.001, kernel: Optional[Callable] = None):
        self.max_streams = int(max_streams)
        self.window_s = window_s
        self._kernel = kernel
        self._bv = None if kernel is not None else BatchedVAD(self.max_streams, device)
        self._host_state = np.zeros((self.max_streams, 2), np.float32) if kernel is not None else None
        self._free = list(range(self.max_streams - 1, -1, -1))
        self._lock = threading.Lock()
        self._q: "queue.Queue" = queue.Queue()
        self.launches = 0
        self.requests = 0
        self._closed = False
        self._thread = threading.Thread(target=self._run, name="thewhisper-vad", daemon=True)
        self._thread.start()

    # -- sessions ------------------------------------------------------------------------------------
    def open_stream(self) -> VadStream:
        with self._lock:
            if not self._free:
                raise RuntimeError("VadService: no free detector slot")
            slot = self._free.pop()
        self.reset(slot)
        return VadStream(self, slot)

    def release(self, slot: int):
        with self._lock:
            if slot not in self._free:
                self._free.append(slot)

    def reset(self, slot: int):
        self.submit(slot, None).result()

    def submit(self, slot: int, frames: Optional[np.ndarray]) -> Future:
        if self._closed:
            raise RuntimeError("VadService is closed")
        fut: Future = Future()
        self._q.put((slot, frames, fut))
        return fut

    def close(self):
        self._closed = True
        self._q.put(None)
        self._thread.join(timeout=10)

    # -- worker --------------------------------------------------------------------------------------
    def _run(self):
        if self._bv is not None:
            torch.cuda.set_device(self._bv.device)
        while True:
            item = self._q.get()
            if item is None:
                return
            batch = [item]
            deadline = time.monotonic() + self.window_s
            while True:
                try:
                    nxt = self._q.get(timeout=max(0.0, deadline - time.monotonic()))
                except queue.Empty:
                    break
                if nxt is None:
                    self._q.put(None)
                    break
                batch.append(nxt)
            try:
                self._serve(batch)
            except Exception as e:  # noqa: BLE001
                for _, _, fut in batch:
                    if not fut.done():
                        fut.set_exception(e)

    def _serve(self, batch: List[Tuple[int, Optional[np.ndarray], Future]]):
        # requests of ONE slot must stay in arrival order (the detector has state); two requests of a slot never share a launch
        rounds: List[List[Tuple[int, Optional[np.ndarray], Future]]] = []
        for it in batch:
            for r in rounds:
                if all(o[0] != it[0] for o in r):
                    r.append(it)
                    break
            else:
                rounds.append([it])
        for r in rounds:
            for slot, frames, fut in [x for x in r if x[1] is None]:
                self._reset_slot(slot)
                fut.set_result(None)
            by_k = collections.defaultdict(list)
            for slot, frames, fut in [x for x in r if x[1] is not None]:
                by_k[len(frames) // FRAME].append((slot, frames, fut))
            for k, grp in by_k.items():
                self.requests += len(grp)
                self.launches += 1
                slots = [g[0] for g in grp]
                pcm = np.stack([g[1] for g in grp])
                probs = self._launch(slots, pcm)
                for i, (_, _, fut) in enumerate(grp):
                    fut.set_result(probs[i])

    def _reset_slot(self, slot: int):
        if self._bv is not None:
            self._bv.state[slot].zero_()
        else:
            self._host_state[slot] = 0

    def _launch(self, slots: List[int], pcm: np.ndarray) -> np.ndarray:
        if self._bv is None:
            p, st = self._kernel(pcm, self._host_state[slots])
            self._host_state[slots] = st
            return np.asarray(p, dtype=np.float32)
        bv = self._bv
        idx = torch.as_tensor(slots, dtype=torch.long, device=bv.device)
        sub = BatchedVAD.__new__(BatchedVAD)     # a view of the service's detector over the slots of this launch
        sub.lib, sub.n, sub.device = bv.lib, len(slots), bv.device
        sub.state = bv.state.index_select(0, idx).contiguous()
        out = sub.probs(pcm)
        bv.state.index_copy_(0, idx, sub.state)
        return out.cpu().numpy()
