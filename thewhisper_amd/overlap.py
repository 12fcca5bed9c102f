"""Encoder / decoder overlap across consecutive batches on ONE MI355X.

The decode loop of a batch is a chain of ~8000 dependent, latency-bound launches that never fills the chip, while the
encoder of the next batch is dense MFMA work.  Two HIP streams alone do not overlap them - the encoder's workgroups take
every CU for ~100 us at a time and the decode chain starves (measured: wall time = sum).  Confining the encoder to a few
CUs with a CU-masked stream (``hipExtStreamCreateWithCUMask``) does: with 64 of 256 CUs the encoder + cross-K/V of 16 ten
second chunks take ~75 ms instead of 24 ms, which hides completely behind the 210 ms decode loop running on the other 192
CUs, and the decode loop slows down by only ~2 % (its launches never use more than 320 workgroups).  Measured on 5 batches of
16 streams (tools/dbg/dbg_overlap.py): 1179 ms sequential, 1149 / 1121 / 1106 / 1101 / 1108 / 1186 ms with 24 / 32 / 48 /
64 / 96 / 128 encoder CUs.

Two contexts of the same model alternate roles (the cross-K/V arena of a context is read by its decode loop, so the next
batch needs its own): while context k % 2 decodes batch k, context (k + 1) % 2 runs log-mel + encoder + cross-K/V of batch
k + 1.  Results are those of the sequential order - every batch still goes through exactly the same kernels - only the
schedule changes.  The reference has no counterpart (R:examples/server.py shares one pipeline and processes one request at
a time); this belongs to the serving layer of SURVEY.md section 8f rank 1.
"""
from __future__ import annotations

import ctypes as C
import queue
import threading
from typing import Any, Callable, Iterable, List, Optional, Sequence

import torch

from .engine import WhisperEngine

__all__ = ["EncoderOverlap", "masked_stream"]

class _Hip:
    """Streams and events through libthewhisper's own entry points (tw_stream_* / tw_event_*): one HIP runtime - the one torch
    mapped and the library is linked against - owns every handle.  (dlopen("libamdhip64.so") by bare name could map a second
    runtime whose handles mean nothing to the first.)"""

    def __init__(self):
        from . import _cabi

        self.lib = _cabi.load_library()

    def stream_create_masked(self, device: int, mask) -> int:
        st = C.c_void_p()
        _chk(self.lib.tw_stream_create_masked(device, mask, len(mask), C.byref(st)), "tw_stream_create_masked")
        return int(st.value)

    def event_create(self, device: int) -> int:
        ev = C.c_void_p()
        _chk(self.lib.tw_event_create(device, C.byref(ev)), "tw_event_create")
        return int(ev.value)

    def event_record(self, ev: int, st: int):
        _chk(self.lib.tw_event_record(C.c_void_p(ev), C.c_void_p(st)), "tw_event_record")

    def stream_wait_event(self, st: int, ev: int):
        _chk(self.lib.tw_stream_wait_event(C.c_void_p(st), C.c_void_p(ev)), "tw_stream_wait_event")

    def stream_synchronize(self, st: int):
        _chk(self.lib.tw_stream_synchronize(C.c_void_p(st)), "tw_stream_synchronize")

    def stream_destroy(self, st: int):
        self.lib.tw_stream_destroy(C.c_void_p(st))

    def event_destroy(self, ev: int):
        self.lib.tw_event_destroy(C.c_void_p(ev))


_hip = None


def _hiplib():
    global _hip
    if _hip is None:
        _hip = _Hip()
    return _hip


def _chk(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc})")


def masked_stream(first_cu: int, last_cu: int, n_cus: int, device: int = 0) -> int:
    """A HIP stream whose kernels may only run on compute units [first_cu, last_cu) of `device`."""
    words = (n_cus + 31) // 32
    mask = (C.c_uint32 * words)(*([0] * words))
    for i in range(first_cu, last_cu):
        mask[i // 32] |= 1 << (i % 32)
    return _hiplib().stream_create_masked(device, mask)


class EncoderOverlap:
    """Runs ``encode_fn(engine, batch)`` of batch k + 1 on ``encoder_cus`` compute units while ``decode_fn(engine, batch,
    encoded)`` of batch k runs on the others.  ``engines``: two contexts built from the same weights, same device."""

    def __init__(self, engines: Sequence[WhisperEngine], encoder_cus: int = 64, cu_range: Optional[Sequence[int]] = None,
                 decoder_cus: Optional[int] = None):
        """``cu_range = (first, last)`` restricts the whole pipeline to a slice of the chip (several pipelines side by side);
        default: all compute units.  The last ``encoder_cus`` CUs of the range run the encoder stage, the first ``decoder_cus``
        (default: all the others) the decode loop."""
        if len(engines) != 2:
            raise ValueError("EncoderOverlap needs exactly two contexts")
        self.engines = list(engines)
        self.device = engines[0].device
        n_cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        lo, hi = (0, n_cus) if cu_range is None else (int(cu_range[0]), int(cu_range[1]))
        if not (0 <= lo < hi <= n_cus) or not 0 < encoder_cus < hi - lo:
            raise ValueError(f"bad CU partition: range [{lo}, {hi}) of {n_cus}, {encoder_cus} encoder CUs")
        self.encoder_cus, self.n_cus = int(encoder_cus), int(n_cus)
        hip = _hiplib()
        dev = self.device.index or 0
        n_dec = hi - lo - encoder_cus if decoder_cus is None else int(decoder_cus)
        if not 0 < n_dec <= hi - lo - encoder_cus:
            raise ValueError(f"bad CU partition: {n_dec} decoder CUs next to {encoder_cus} encoder CUs in [{lo}, {hi})")
        self.s_dec = masked_stream(lo, lo + n_dec, n_cus, dev)
        self.s_enc = masked_stream(hi - encoder_cus, hi, n_cus, dev)
        self.s_all = masked_stream(lo, hi, n_cus, dev)     # the first batch's encoder stage: nothing else is running yet
        self._events: List[int] = [hip.event_create(dev) for _ in range(2)]

    def close(self):
        hip = _hiplib()
        for e in self.engines:
            e.raw_stream = None
        for ev in self._events:
            hip.event_destroy(ev)
        for s in (self.s_dec, self.s_enc, self.s_all):
            if s is not None:
                hip.stream_destroy(s)   # synchronises first
        self._events, self.s_dec, self.s_enc, self.s_all = [], None, None, None

    def run(self, batches: Iterable[Any], encode_fn: Callable[[WhisperEngine, Any], Any],
            decode_fn: Callable[[WhisperEngine, Any, Any], Any]) -> List[Any]:
        """Processes the batches in order and returns ``decode_fn``'s results in order.

        ``encode_fn`` only enqueues work; it must RETURN every torch tensor it handed to the engine (e.g. the log-mel tensor):
        the pipeline keeps the returned object alive until the batch has been decoded.  Dropping such a tensor earlier hands
        its memory back to torch's caching allocator while the CU-masked stream - which the allocator knows nothing about -
        is still reading it, and the next batch's tensor lands on top of it."""
        hip = _hiplib()
        batches = list(batches)
        free = [threading.Semaphore(1), threading.Semaphore(1)]  # context i may be (re)used by the encode stage
        ready: "queue.Queue" = queue.Queue()
        stop = threading.Event()                                 # set when the consumer gives up (a stage failed)
        dev_index = self.device.index or 0

        def producer():
            try:
                torch.cuda.set_device(dev_index)   # torch allocations of this thread; the tw_* calls guard their own device
                for i, b in enumerate(batches):
                    k = i % 2
                    free[k].acquire()                      # decode of batch i - 2 has returned
                    if stop.is_set():
                        return
                    eng = self.engines[k]
                    # the first batch has no decode loop to hide behind: its encoder stage gets the whole range (17 instead of 25 ms
                    # on the decoder's 160 CUs for 16 x 10 s: the pipeline fill every timed region pays once)
                    s_i = self.s_all if i == 0 else self.s_enc
                    eng.raw_stream = s_i
                    enc = encode_fn(eng, b)                # asynchronous launches
                    hip.event_record(self._events[k], s_i)
                    ready.put((i, k, enc, None))
            except BaseException as e:  # noqa: BLE001  (hand the failure to the consumer)
                ready.put((-1, -1, None, e))

        th = threading.Thread(target=producer, name="tw-encoder-stage", daemon=True)
        th.start()
        out: List[Any] = []
        try:
            for _ in range(len(batches)):
                i, k, enc, err = ready.get()
                if err is not None:
                    raise err
                eng = self.engines[k]
                hip.stream_wait_event(self.s_dec, self._events[k])
                eng.raw_stream = self.s_dec
                out.append(decode_fn(eng, batches[i], enc))  # blocking: returns when the batch is decoded
                hip.stream_synchronize(self.s_dec)
                free[k].release()
        finally:
            stop.set()
            for f in free:
                f.release()
            th.join(timeout=60)
            for e in self.engines:   # back to torch's current stream for whoever uses the contexts next
                e.raw_stream = None
        return out
