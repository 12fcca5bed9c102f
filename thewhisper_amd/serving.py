"""Cross-stream batching front end (BASELINE config 4: 16 streams per GPU).

The reference's demo server shares ONE StreamingPipeline across sessions and cannot batch
(R:examples/server.py:22-115).  Here every session keeps its own, unmodified, reference
``StreamingPipeline`` (per-stream scheduler state), and all sessions of a GPU share one
``BatchingHub``: each session's backend proxy implements the reference's
``TranscriptionBackend.transcribe`` contract (R:thestage_speechkit/streaming/streaming_pipeline.py:51-64),
but instead of running the pipeline itself it parks the request; a single worker thread drains up to
``max_batch`` parked requests and runs them as ONE ``ASRPipeline`` call on a list of buffers - HF's chunk
iterator then collates the 10 s chunks of different streams into one batched encoder/decoder pass
(HF:pipelines/base.py:1319-1340), i.e. one ``tw_encode`` / ``tw_generate_greedy`` over <=16 streams.
Results are identical to per-stream calls (each stream's tokens depend only on its own buffer).
"""
from __future__ import annotations

import queue
import threading
import time
from concurrent.futures import Future
from typing import Any, Dict, List, Optional, Tuple

import numpy as np

from .streaming import AMDWhisperBackend


class _StreamBackend:
    """What a session's StreamingPipeline sees: a blocking ``transcribe`` with the reference signature."""

    def __init__(self, hub: "BatchingHub", stream_id: int):
        self.hub = hub
        self.stream_id = stream_id

    def transcribe(self, audio: np.ndarray, buffer_start_time: float, sample_rate: int) -> List[Dict[str, Any]]:
        return self.hub.submit(audio, buffer_start_time, sample_rate).result()


class BatchingHub:
    def __init__(self, backend: AMDWhisperBackend, max_batch: Optional[int] = None, max_wait_s: float = 0.004,
                 max_pending: int = 1024):
        self.backend = backend
        eng = backend.asr_pipeline.model.engine
        self.max_batch = int(max_batch or eng.max_batch)
        if self.max_batch > eng.max_batch:
            raise ValueError("max_batch exceeds the engine capacity")
        self.max_wait_s = max_wait_s
        self._q: "queue.Queue[Optional[Tuple[np.ndarray, float, int, Future]]]" = queue.Queue(maxsize=max_pending)
        self._closed = False
        self._lock = threading.Lock()
        self._next_id = 0
        self.batches: List[int] = []          # sizes of the batches that were run (introspection / tests)
        self._worker = threading.Thread(target=self._run, name="thewhisper-batcher", daemon=True)
        self._worker.start()

    # -- session side --------------------------------------------------------------------------------
    def stream_backend(self) -> _StreamBackend:
        self._next_id += 1
        return _StreamBackend(self, self._next_id - 1)

    def submit(self, audio: np.ndarray, buffer_start_time: float, sample_rate: int) -> Future:
        """Parks one request; raises ``RuntimeError`` after ``close()`` and ``queue.Full`` when ``max_pending`` requests wait."""
        fut: Future = Future()
        with self._lock:
            if self._closed:
                raise RuntimeError("BatchingHub is closed")
            self._q.put_nowait((np.asarray(audio), float(buffer_start_time), int(sample_rate), fut))
        return fut

    def close(self):
        """Stops the worker; requests still parked are failed (their sessions would otherwise wait forever)."""
        with self._lock:
            if self._closed:
                return
            self._closed = True
        self._q.put(None)
        self._worker.join(timeout=30)
        self._fail_pending(RuntimeError("BatchingHub closed before the request was served"))

    def _fail_pending(self, exc: Exception):
        while True:
            try:
                item = self._q.get_nowait()
            except queue.Empty:
                return
            if item is not None and not item[3].done():
                item[3].set_exception(exc)

    # -- worker --------------------------------------------------------------------------------------
    def _run(self):
        try:   # the batcher owns a GPU context: pin this thread's torch device to it
            import torch

            dev = getattr(self.backend.asr_pipeline.model.engine, "device", None)
            if dev is not None and dev.type == "cuda":
                torch.cuda.set_device(dev)
        except Exception:  # noqa: BLE001
            pass
        while True:
            item = self._q.get()
            if item is None:
                return
            batch = [item]
            deadline = time.monotonic() + self.max_wait_s
            while len(batch) < self.max_batch:
                try:
                    nxt = self._q.get(timeout=max(0.0, deadline - time.monotonic()))
                except queue.Empty:
                    break
                if nxt is None:
                    self._q.put(None)
                    break
                batch.append(nxt)
            self._execute(batch)

    def _execute(self, batch):
        self.batches.append(len(batch))
        try:
            results = self.backend.transcribe_many([(a, t0, sr) for a, t0, sr, _ in batch], batch_size=self.max_batch)
            for (_, _, _, fut), res in zip(batch, results):
                fut.set_result(res)
        except Exception as e:  # noqa: BLE001
            if len(batch) == 1:
                if not batch[0][3].done():
                    batch[0][3].set_exception(e)   # as the reference would raise in that session
                return
            # one malformed buffer must not fail its neighbours: run the members of the batch one by one, so that only
            # the offending session sees the exception
            for a, t0, sr, fut in batch:
                if fut.done():
                    continue
                try:
                    fut.set_result(self.backend.transcribe_many([(a, t0, sr)], batch_size=self.max_batch)[0])
                except Exception as e1:  # noqa: BLE001
                    fut.set_exception(e1)
