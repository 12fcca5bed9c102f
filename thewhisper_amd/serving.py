"""Cross-stream batching front end (BASELINE config 4: 16 streams per GPU).

The reference's demo server shares ONE StreamingPipeline across sessions and cannot batch
(R:examples/server.py:22-115).  Here every session keeps its own, unmodified, reference
``StreamingPipeline`` (per-stream scheduler state), and all sessions of a GPU share one
``BatchingHub``: each session's backend proxy implements the reference's
``TranscriptionBackend.transcribe`` contract (R:thestage_speechkit/streaming/streaming_pipeline.py:51-64),
but instead of running the pipeline itself it parks the request; a single worker thread owns the engine.

Scheduling unit = one *seek pass of one chunk* (``thewhisper_amd.shortform``): a request is pre-processed into its
chunks' decoding states, every engine pass takes up to ``max_batch`` unfinished chunks - new ones and ones that
need a further seek iteration alike, whichever session they belong to - and a request is post-processed and answered
as soon as ITS chunks are done.  (Until round 3 a batch of requests went through one ``ASRPipeline`` call: HF's seek
loop then runs its later iterations with ever fewer rows while the finished sessions wait for the slowest one - a
quarter of the engine's throughput on the synthetic streams, profiles/r02_SUMMARY.md.)  Backends whose call is not
eligible for the restated loop (``AMDWhisperBackend.job_codec() is None``) are served in whole-call batches as before.
Results are identical to per-stream calls either way (each stream's tokens depend only on its own buffer).
"""
from __future__ import annotations

import collections
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
        return self.hub.submit(audio, buffer_start_time, sample_rate, stream_id=self.stream_id).result()


class _Prefetcher:
    """Encoder stage of NEW arrivals under the decode loop of the running pass.

    A pass's members are only known when the previous pass has ended (a chunk that needs a further seek iteration continues
    where the decoded timestamps say), so THEIR encoder stage cannot be hidden; a new request's first iteration always starts
    at frame 0 and can.  This thread is enabled only while ``Pass.run`` blocks in the greedy loop: it takes requests off the
    hub's queue, opens them (HF's ``preprocess`` with the log-mel on the sibling context) and enqueues encoder + cross-K/V of
    their chunks on a CU-masked stream into the sibling's slots; the hub's next pass adopts those rows (``Pass.adopt`` ->
    tw_adopt_cross_kv, ~30 us per 10 s chunk) behind the rows that continue.  Same kernels, same weights, same results."""

    def __init__(self, hub: "BatchingHub", codec, engine, n_cus: int):
        import torch

        from .overlap import masked_stream

        self.hub, self.codec, self.main = hub, codec, engine
        self.cap = int(hub.max_batch)
        self.side = engine.sibling(self.cap)
        self.stream = None
        self.main_stream = None
        dev = getattr(engine, "device", None)
        if dev is not None and dev.type == "cuda":     # (the CPU stand-in engine of the tests has no streams)
            total = torch.cuda.get_device_properties(dev).multi_processor_count
            n_cus = max(8, min(int(n_cus), total - 8))
            self.stream = masked_stream(total - n_cus, total, total, dev.index or 0)
            self.side.raw_stream = self.stream
            # ... and the passes themselves on the OTHER compute units: a decode launch whose workgroups land behind the encoder's
            # ~100 us tiles holds up the whole dependent chain (measured with the main stream unmasked: decode loop 194.7 instead of
            # 178 ms per pass; thewhisper_amd/overlap.py masks both streams for the same reason)
            self.main_prev = engine.raw_stream
            self.main_stream = masked_stream(0, total - n_cus, total, dev.index or 0)
            engine.raw_stream = self.main_stream
        self.lock = threading.Lock()          # held by this thread while it works on one request
        self.enabled = threading.Event()
        self.stop = False
        self.pre: "collections.deque" = collections.deque()    # (work, side slot, segment tensor), slot order
        self.ahead: List[Any] = []            # chunks of requests in flight that sit the running pass out: encoded first, in one call
        self.dead: List[Any] = []             # pre-encoded chunks whose request has failed (drop): the OBJECTS, compared with `is` -
                                              # ids of freed works are reused by CPython and would discard a live chunk's row
        self.new_jobs: List[Any] = []
        self.count = 0                        # filled slots of the sibling context
        self.thread = threading.Thread(target=self._run, name="thewhisper-prefetch", daemon=True)
        self.thread.start()

    # -- hub side (the batcher thread) ----------------------------------------------------------------
    def resume(self):
        self.enabled.set()

    def pause(self):
        """Returns once this thread is neither holding a request nor touching the sibling context."""
        self.enabled.clear()
        with self.lock:
            pass

    def drain_jobs(self) -> List[Any]:
        jobs, self.new_jobs = self.new_jobs, []
        return jobs

    def take(self, free: int):
        """Up to `free` pre-encoded rows from the front (consecutive sibling slots): ([works], first slot, [tensors to keep])."""
        works, keep, slot0 = [], [], None
        while self.pre and len(works) < free:
            if any(self.pre[0][0] is w for w in self.dead):   # a chunk of a request that has failed meanwhile (drop): nobody waits for it
                if works:
                    break                           # ... and the run of CONSECUTIVE slots ends in front of it
                gone = self.pre.popleft()[0]
                self.dead = [w for w in self.dead if w is not gone]
                continue
            w, slot, seg = self.pre.popleft()
            if slot0 is None:
                slot0 = slot
            works.append(w)
            keep.append(seg)
        if not self.pre:
            self.count = 0        # tw_adopt_cross_kv orders the sibling's next launches behind the copies
            self.dead.clear()
        return works, slot0, keep

    def drop(self, dead_works):
        """(batcher thread, prefetcher paused)  Forget the chunks of failed requests: not encoded ahead any more, and their
        pre-encoded rows are skipped by `take` (their sibling slots stay unused until the queue has drained).  Under the lock a chunk
        is in `ahead`, in `pre` or not with the prefetcher at all (`_encode_ahead` moves it under the same lock), so only those in `pre`
        need remembering - as OBJECTS (no id of a freed work can match a live one), each forgotten when `take` pops it."""
        with self.lock:       # (the prefetcher may be running: a failed pass.run() is reported after resume())
            self.ahead = [w for w in self.ahead if not any(w is x for x in dead_works)]
            self.dead += [w for (w, _slot, _seg) in self.pre if any(w is x for x in dead_works) and not any(w is x for x in self.dead)]

    def shutdown(self):
        self.stop = True
        self.enabled.set()
        self.thread.join(timeout=30)
        if self.stream is not None:
            try:
                from .overlap import _hiplib

                self.side.raw_stream = None
                _hiplib().stream_destroy(self.stream)     # synchronises first
                self.main.raw_stream = self.main_prev
                _hiplib().stream_destroy(self.main_stream)
            except Exception:  # noqa: BLE001
                pass
        self.side.close()

    # -- the thread -------------------------------------------------------------------------------------
    def _run(self):
        import torch

        from . import shortform

        import contextlib

        fe = self.codec.pipe.feature_extractor
        fe._tls.engine = self.side              # this thread's log-mel calls go to the sibling context (feature_extraction.py)
        ctx = contextlib.nullcontext()
        if self.stream is not None:             # ... and its torch ops to the side stream, never to the null stream
            dev = self.main.device
            torch.cuda.set_device(dev)
            ctx = torch.cuda.stream(torch.cuda.ExternalStream(self.stream, device=dev))
        with ctx:
            while True:
                self.enabled.wait()
                if self.stop:
                    return
                item = _NOTHING
                with self.lock:
                    if self.enabled.is_set() and self.ahead:
                        works, self.ahead = self.ahead[: self.cap - self.count], []
                        if works:
                            self._encode_ahead(works, shortform)
                        continue
                    if self.enabled.is_set() and self.count < self.cap:
                        try:
                            item = self.hub._take(timeout=0.004)
                        except queue.Empty:
                            item = _NOTHING
                    if item is None:              # close(): the batcher thread must see it too
                        self.hub._stopping = True
                        try:
                            self.hub._q.put_nowait(None)
                        except queue.Full:
                            pass
                        return
                    if item is not _NOTHING:
                        self._prefetch(item, shortform)
                if item is _NOTHING:
                    time.sleep(0.001)

    def _encode_ahead(self, works, shortform):
        """Next seek iteration of chunks that are NOT in the running pass (more requests in flight than a pass has rows): their
        seek is final until they run again, so their encoder stage goes under this pass's decode loop, one batched call."""
        import torch

        n = len(works)
        try:
            segs = torch.stack([shortform.next_segment(w, self.side.T) for w in works], dim=0)
            self.side.encode(segs, slot0=self.count) if self.count else self.side.encode(segs)
            self.side.cross_kv(n, slot0=self.count) if self.count else self.side.cross_kv(n)
        except Exception:  # noqa: BLE001  (the batcher encodes them itself, and reports what fails there)
            return
        for i, w in enumerate(works):
            self.pre.append((w, self.count + i, segs))
        self.count += n
        self.hub.prefetched += n
        self.hub.prefetched_ahead += n

    def _prefetch(self, item, shortform):
        import torch

        audio, t0, sr, fut = item
        try:
            job = self.codec.open(audio, t0, sr)
        except Exception as e:  # noqa: BLE001  (a malformed buffer fails its own session only)
            self.hub._answer(fut, exc=e)
            return
        job.future = fut
        n = len(job.works)
        if self.count + n > self.cap:
            # does not fit the sibling's free slots: the batcher takes it through the ordinary path (opened already).  Its log-mel
            # was enqueued on THIS thread's stream and the batcher slices it on its own: nothing but the legacy null stream's
            # implicit ordering would hold the two apart, so the hand-over waits for the features (rare path, ~0.1 ms)
            if self.stream is not None:
                torch.cuda.current_stream().synchronize()
            self.new_jobs.append(job)
            return
        try:
            segs = torch.stack([shortform.first_segment(w, self.side.T) for w in job.works], dim=0)
            self.side.encode(segs, slot0=self.count) if self.count else self.side.encode(segs)
            self.side.cross_kv(n, slot0=self.count) if self.count else self.side.cross_kv(n)
        except Exception as e:  # noqa: BLE001
            self.hub._answer(fut, exc=e)
            return
        for i, w in enumerate(job.works):
            self.pre.append((w, self.count + i, segs))
        self.count += n
        self.new_jobs.append(job)
        self.hub.prefetched += n


_NOTHING = object()


class BatchingHub:
    def __init__(self, backend: AMDWhisperBackend, max_batch: Optional[int] = None, max_wait_s: float = 0.004,
                 max_pending: int = 1024, continuous: bool = True, gather_s: float = 0.15, prefetch_cus: int = 0):
        self.backend = backend
        eng = backend.asr_pipeline.model.engine
        self.max_batch = int(max_batch or eng.max_batch)
        if self.max_batch > eng.max_batch:
            raise ValueError("max_batch exceeds the engine capacity")
        self.max_wait_s = max_wait_s
        # Sessions in a closed loop (answer -> next chunk -> next request) split into two cohorts that take turns in half-empty
        # passes: the requests answered by pass k come back while pass k + 1 - started the moment pass k ended, with whoever
        # was already waiting - is running (128 sessions on 8 stub ranks, tests/test_node_scale.py: exactly 8.0 of 16 rows per
        # pass).  So a pass with free rows also waits for the sessions that were JUST ANSWERED, as long as they are on their way:
        # the hub keeps a running estimate of the sessions' turnaround (answer -> next request of the same stream; only
        # samples below 0.25 s count, i.e. closed-loop callers - a real-time session comes back after 0.5 s of audio and is not
        # waited for) and holds the pass until the answered streams have asked again or 1.25 x that estimate (the 90th
        # percentile of the recent samples; at most `gather_s`) has passed since the last answer.  Costs the requests already waiting at most that, only while rows are free.
        self.gather_s = float(gather_s)
        self._last_answer_t = 0.0
        self._turn_ema = 0.0                       # seconds; 0 = no closed-loop caller seen (reported by /health)
        self._turns: "collections.deque[Tuple[float, float]]" = collections.deque(maxlen=64)   # recent (when, turnaround) samples
        self.turn_horizon_s = 10.0                 # samples older than this no longer describe who is calling: the estimate decays to 0
        self._answered: Dict[int, float] = {}      # stream id -> when its last request was answered (cleared when it asks again)
        self._q: "queue.Queue[Optional[Tuple[np.ndarray, float, int, Future]]]" = queue.Queue(maxsize=max_pending)
        self._closed = False
        self._lock = threading.Lock()
        self._next_id = 0
        self.batches: "collections.deque[int]" = collections.deque(maxlen=4096)   # sizes of the LAST passes / batches that were run
        self.passes = 0   # monotonic count of engine passes (continuous) / pipeline batches (classic): what /health reports
        self.rows = 0     # ... and of the rows (chunks / requests) they carried: rows / passes = how full the passes are
        self.latencies: "collections.deque[float]" = collections.deque(maxlen=4096)  # submit -> answer, seconds
        self._codec = None
        self._carry = None
        # > 0: while a pass decodes, the requests that ARRIVE are opened (log-mel) and the first seek iteration of their chunks is
        # encoded on a sibling context of the same weights, on a stream confined to that many compute units; the next pass adopts
        # them (their encoder stage is then hidden under the decode loop).  Continuous mode on the real engine only.
        self.prefetch_cus = int(prefetch_cus)
        self._prefetcher: Optional["_Prefetcher"] = None
        self._stopping = False
        self.prefetched = 0     # rows whose encoder stage ran under another pass's decode loop
        self.prefetched_ahead = 0   # ... of which: rows of requests in flight that sat a pass out (more requests than rows per pass)
        # wall time of the batcher loop by phase, summed over the passes (seconds): where a pass's time goes besides the decode loop
        self.phase_s: Dict[str, float] = {"pause": 0.0, "adopt": 0.0, "encode_left": 0.0, "intake": 0.0, "encode_fresh": 0.0,
                                          "run": 0.0, "after": 0.0}
        if continuous and hasattr(backend, "job_codec"):
            self._codec = backend.job_codec()
        self._post_q: "queue.Queue" = queue.Queue()
        self._poster = None
        if self._codec is not None:
            self._poster = threading.Thread(target=self._post_loop, name="thewhisper-postprocess", daemon=True)
            self._poster.start()
        self._worker = threading.Thread(target=self._run, name="thewhisper-batcher", daemon=True)
        self._worker.start()

    # -- session side --------------------------------------------------------------------------------
    def stream_backend(self) -> _StreamBackend:
        self._next_id += 1
        return _StreamBackend(self, self._next_id - 1)

    def submit(self, audio: np.ndarray, buffer_start_time: float, sample_rate: int, stream_id: Optional[int] = None) -> Future:
        """Parks one request; raises ``RuntimeError`` after ``close()`` and ``queue.Full`` when ``max_pending`` requests wait."""
        fut: Future = Future()
        fut.t_submit = time.monotonic()  # type: ignore[attr-defined]
        fut.stream_id = stream_id        # type: ignore[attr-defined]
        with self._lock:
            if self._closed:
                raise RuntimeError("BatchingHub is closed")
            if stream_id is not None:
                t_ans = self._answered.pop(stream_id, None)
                if t_ans is not None and fut.t_submit - t_ans < 0.25:      # type: ignore[attr-defined]
                    self._turns.append((fut.t_submit, fut.t_submit - t_ans))  # type: ignore[attr-defined]
                    self._refresh_turnaround(fut.t_submit)                 # type: ignore[attr-defined]
            self._q.put_nowait((np.asarray(audio), float(buffer_start_time), int(sample_rate), fut))
        return fut

    def _refresh_turnaround(self, now: float):
        """(lock held)  The gather window has to cover the SLOW returners (a pass started without them leaves them a pass of their
        own): 90th percentile of the turnaround samples taken within the last `turn_horizon_s`.  Samples only ever arrive from
        closed-loop callers (< 0.25 s), so without the horizon one burst of them (a benchmark, a fast client) would keep later
        real-time sessions waiting for streams that are not coming back; with none recent the estimate is 0 = no gathering."""
        while self._turns and now - self._turns[0][0] > self.turn_horizon_s:
            self._turns.popleft()
        if not self._turns:
            self._turn_ema = 0.0
            return
        srt = sorted(d for _, d in self._turns)
        self._turn_ema = srt[min(len(srt) - 1, int(0.9 * len(srt)))]

    def close(self):
        """Stops the worker; requests still parked are failed (their sessions would otherwise wait forever)."""
        with self._lock:
            if self._closed:
                return
            self._closed = True
        self._q.put(None)
        self._worker.join(timeout=30)
        if self._poster is not None:
            self._post_q.put(None)
            self._poster.join(timeout=30)
        self._fail_pending(RuntimeError("BatchingHub closed before the request was served"))

    def _fail_pending(self, exc: Exception):
        while True:
            try:
                item = self._q.get_nowait()
            except queue.Empty:
                return
            if item is not None and not item[3].done():
                item[3].set_exception(exc)

    def _answer(self, fut: Future, result=None, exc: Optional[BaseException] = None):
        if fut.done():
            return
        now = time.monotonic()
        self.latencies.append(now - getattr(fut, "t_submit", now))
        sid = getattr(fut, "stream_id", None)
        with self._lock:
            self._last_answer_t = now
            if sid is not None:
                self._answered[sid] = now
                if len(self._answered) > 256:    # streams that went away: forget answers older than any turnaround that counts
                    self._answered = {k: t for k, t in self._answered.items() if now - t < 0.25}
        if exc is not None:
            fut.set_exception(exc)
        else:
            fut.set_result(result)

    # -- worker --------------------------------------------------------------------------------------
    def _run(self):
        try:   # the batcher owns a GPU context: pin this thread's torch device to it
            import torch

            dev = getattr(self.backend.asr_pipeline.model.engine, "device", None)
            if dev is not None and dev.type == "cuda":
                torch.cuda.set_device(dev)
        except Exception:  # noqa: BLE001
            pass
        if self._codec is not None:
            # the plan is learned when the FIRST request arrives, not in the constructor: until then this thread does not touch
            # the engine (whoever built the hub may still be using the backend directly, and a context is not thread-safe)
            first = self._q.get()
            if first is None:
                return
            self._carry = first
            try:
                ok = self._codec.learn()
            except Exception:  # noqa: BLE001  (the warm-up request failed: serve whole-call batches, errors surface per request)
                ok = False
            if ok:
                self._run_continuous()
                return
            self._codec = None
        self._run_classic()

    def _gather_until(self, deadline: float) -> float:
        """Until when a pass with free rows keeps its intake open: `deadline`, or - while streams answered a moment ago have
        not asked again - 1.5 x the measured turnaround after the last answer (see __init__)."""
        if self._turn_ema <= 0.0:
            return deadline
        now = time.monotonic()
        with self._lock:
            self._refresh_turnaround(now)    # the estimate decays: no closed-loop caller for `turn_horizon_s` = no gathering
            if self._turn_ema <= 0.0:
                return deadline
            pending = any(now - t < 0.25 for t in self._answered.values())
            last = self._last_answer_t
        if not pending:
            return deadline
        return max(deadline, last + min(self.gather_s, 1.25 * self._turn_ema))

    def _take(self, block: bool = True, timeout: Optional[float] = None):
        """Next parked request: the one taken off the queue before the plan was learned first, then the queue."""
        if self._carry is not None:
            item, self._carry = self._carry, None
            return item
        if not block:
            return self._q.get_nowait()
        return self._q.get(timeout=timeout)

    # .. continuous: the unit is one seek pass of one chunk ..........................................
    def _run_continuous(self):
        from . import shortform

        codec = self._codec
        eng = self.backend.asr_pipeline.model.engine
        jobs: List[Any] = []          # requests in flight, oldest first
        stop = False
        pf = None
        if self.prefetch_cus > 0 and hasattr(eng, "sibling"):
            pf = self._prefetcher = _Prefetcher(self, codec, eng, self.prefetch_cus)

        def fail_jobs_of(works, exc):
            nonlocal jobs
            hit = [j for j in jobs if any(w in works for w in j.works)]
            for j in hit:
                self._answer(j.future, exc=exc)
            jobs = [j for j in jobs if j not in hit]
            if pf is not None and hit:
                # their other chunks must not be adopted or encoded ahead any more: the futures are answered, the rows would only
                # occupy later passes (and a chunk of a dead job that exceeds MAX_SEEK_PASSES would fail the LIVE jobs of its pass)
                pf.drop([w for j in hit for w in j.works])

        def shutdown():
            if pf is not None:
                pf.shutdown()
                jobs.extend(pf.drain_jobs())
            for j in jobs:
                self._answer(j.future, exc=RuntimeError("BatchingHub closed before the request was served"))

        ph = self.phase_s
        t_mark = time.perf_counter()

        def lap(name):
            nonlocal t_mark
            now = time.perf_counter()
            ph[name] += now - t_mark
            t_mark = now

        while True:
            t_mark = time.perf_counter()
            pas = shortform.Pass(eng, codec.plan)
            if pf is not None:
                pf.pause()            # from here to resume() this thread is the only user of the queue and of the sibling context
                jobs.extend(j for j in pf.drain_jobs() if j not in jobs)
            lap("pause")
            if self._stopping:
                shutdown()
                return
            # 1. rows whose next iteration was encoded under the previous pass's decode loop (_Prefetcher: arrivals, and - with more
            #    requests in flight than a pass has rows - the chunks that sat that pass out): a copy each
            adopted: List[Any] = []

            def room() -> int:          # rows this pass can still take
                return min(pas.free, self.max_batch - len(pas.works))

            if pf is not None and pf.pre and room() > 0:
                ws, slot0, keep = pf.take(room())
                try:
                    if ws:
                        pas.adopt(ws, pf.side, slot0)
                    pas._keep.extend(keep)
                    adopted = ws
                except Exception as e:  # noqa: BLE001
                    fail_jobs_of(ws, e)
                    continue
            lap("adopt")
            # 1b. the chunks that need a further seek iteration: their encoder stage is enqueued NOW (asynchronous), so the GPU is
            #    already busy while the sessions answered after the last pass are on their way back
            pre_works = set(id(w) for w, _, _ in pf.pre) if pf is not None else set()
            taken = set(id(w) for w in adopted)
            waiting = [w for j in jobs for w in j.works if not w.done and id(w) not in pre_works and id(w) not in taken]
            left = waiting[: room()]
            if left:
                try:
                    pas.add(left)
                except Exception as e:  # noqa: BLE001  (engine failure)
                    fail_jobs_of(left, e)
                    continue
            if pf is not None:
                pf.ahead = waiting[len(left):]     # they sit this pass out: encoded on the side while it decodes
            lap("encode_left")
            # 2. intake: block when there is nothing to decode; otherwise linger while rows are free - for max_wait_s, or for as
            #    long as the encoder stage of step 1 keeps the GPU busy anyway (~1.2 ms per chunk), whichever is longer
            fresh: List[Any] = []
            linger = max(self.max_wait_s, 0.0012 * len(left))
            deadline = time.monotonic() + linger if (left or adopted) else None
            while room() - len(fresh) > 0:
                try:
                    if deadline is None:
                        item = self._take()
                        deadline = time.monotonic() + self.max_wait_s
                    else:
                        # (re-read every time: the poster thread answers the previous pass's requests while this loop waits)
                        item = self._take(timeout=max(0.0, self._gather_until(deadline) - time.monotonic()))
                except queue.Empty:
                    break
                if item is None:
                    stop = True
                    break
                audio, t0, sr, fut = item
                try:
                    job = codec.open(audio, t0, sr)
                except Exception as e:  # noqa: BLE001  (a malformed buffer fails its own session only)
                    self._answer(fut, exc=e)
                    continue
                job.future = fut
                jobs.append(job)
                fresh.extend(job.works)
            if stop:
                shutdown()
                return
            lap("intake")
            # 3. the late arrivals join the pass as a further group (slots after the others'), then ONE greedy loop over all
            works = list(adopted) + list(left)
            try:
                take = fresh[: room()]
                if take:
                    pas.add(take)
                    works += take
                if not works:
                    continue
                self.batches.append(len(works))
                self.passes += 1
                self.rows += len(works)
                if pf is not None:
                    pf.resume()       # arrivals from now on are encoded on the side while this pass decodes
                lap("encode_fresh")
                pas.run()
                lap("run")
                ph["greedy"] = ph.get("greedy", 0.0) + getattr(pas, "greedy_s", 0.0)
                for w in works:
                    if w.passes > shortform.MAX_SEEK_PASSES:
                        raise RuntimeError(f"a chunk needed more than {shortform.MAX_SEEK_PASSES} seek passes (the decoder keeps seeking to frame 0)")
            except Exception as e:  # noqa: BLE001  (engine failure: every request that had a chunk in this pass fails)
                fail_jobs_of(works, e)
                continue
            # 4. finished requests leave for post-processing (tokenizer state machine, word merge: host Python that overlaps
            #    the next pass - the engine call releases the GIL)
            still = []
            for j in jobs:
                if j.done:
                    self._post_q.put(j)
                else:
                    still.append(j)
            jobs = still
            lap("after")

    def _post_loop(self):
        while True:
            job = self._post_q.get()
            if job is None:
                return
            try:
                self._answer(job.future, result=self._codec.close(job))
            except Exception as e:  # noqa: BLE001
                self._answer(job.future, exc=e)

    # .. classic: the unit is one pipeline call over a batch of requests .............................
    def _run_classic(self):
        while True:
            item = self._take()
            if item is None:
                return
            batch = [item]
            deadline = time.monotonic() + self.max_wait_s
            while len(batch) < self.max_batch:
                try:
                    nxt = self._q.get(timeout=max(0.0, self._gather_until(deadline) - time.monotonic()))
                except queue.Empty:
                    break
                if nxt is None:
                    self._q.put(None)
                    break
                batch.append(nxt)
            self._execute(batch)

    def _execute(self, batch):
        self.batches.append(len(batch))
        self.passes += 1
        self.rows += len(batch)
        try:
            results = self.backend.transcribe_many([(a, t0, sr) for a, t0, sr, _ in batch], batch_size=self.max_batch)
            for (_, _, _, fut), res in zip(batch, results):
                self._answer(fut, result=res)
        except (ValueError, TypeError) as e:
            if len(batch) == 1:
                self._answer(batch[0][3], exc=e)   # as the reference would raise in that session
                return
            # one malformed buffer must not fail its neighbours: run the members of the batch one by one, so that only
            # the offending session sees the exception (input errors only - an engine fault is not retried B more times)
            for a, t0, sr, fut in batch:
                if fut.done():
                    continue
                try:
                    self._answer(fut, result=self.backend.transcribe_many([(a, t0, sr)], batch_size=self.max_batch)[0])
                except Exception as e1:  # noqa: BLE001
                    self._answer(fut, exc=e1)
        except Exception as e:  # noqa: BLE001
            for _, _, _, fut in batch:
                self._answer(fut, exc=e)
