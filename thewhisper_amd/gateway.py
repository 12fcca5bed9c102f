"""HTTP gateway that speaks the wire format of the reference's remote backend.

The reference's ``RemoteAPITimestampsBackend`` (R:thestage_speechkit/streaming/streaming_pipeline.py:66-337) posts each
rolling buffer as a mono 16-bit WAV (multipart field ``file``, name ``chunk.wav``) with optional ``Authorization: Bearer``,
``X-Lang-Id`` and ``X-Model-Name`` headers, and expects JSON with ``transcription`` (or ``text``) and
``metadata.chunks = [{"text", "timestamp": [start, end]}, ...]`` (word level, seconds relative to the buffer).  A reference
installation pointed at this gateway (``TRITON_URL=http://host:port/transcribe``, ``use_remote_api=True``) therefore runs on
the MI355X backend without a code change, and - unlike the reference's demo server, which shares one pipeline and handles
one request at a time (R:examples/server.py:22-115) - concurrent requests of different sessions are executed as ONE batched
encoder/decoder pass through ``BatchingHub``.

Host-side Python only; the hot path stays behind ``AMDWhisperBackend``.  ``python -m thewhisper_amd.gateway --model ...``
"""
import hmac
import io
import wave
from typing import Any, Dict, List, Optional, Tuple, Union

import numpy as np

from .serving import BatchingHub
from .streaming import AMDWhisperBackend

__all__ = ["create_app", "decode_wav", "words_to_response", "multipart_file", "SessionHost", "HostBusy"]


def multipart_file(body: bytes, content_type: str, field: str = "file") -> bytes:
    """Content of the multipart/form-data part named ``field`` (what ``httpx.post(files={"file": (...)})`` of the reference's
    client produces, R:...streaming_pipeline.py:126-134).  Parsed here because the optional ``python-multipart`` package that
    FastAPI's ``UploadFile`` needs is not a dependency of this repository."""
    ct = content_type or ""
    if "multipart/form-data" not in ct or "boundary=" not in ct:
        raise ValueError("expected multipart/form-data with a boundary")
    boundary = ct.split("boundary=", 1)[1].split(";", 1)[0].strip().strip('"').encode()
    for part in body.split(b"--" + boundary):
        head, sep, content = part.partition(b"\r\n\r\n")
        if not sep:
            continue
        headers = head.decode("latin-1").lower()
        if "content-disposition" in headers and f'name="{field}"' in headers:
            return content[:-2] if content.endswith(b"\r\n") else content
    raise ValueError(f"multipart body has no part named '{field}'")


def decode_wav(data: bytes) -> "Tuple[np.ndarray, int]":
    """WAV bytes -> (float32 mono in [-1, 1], sample rate).  Accepts what the reference client sends (16-bit PCM, mono;
    R:...streaming_pipeline.py:93-112) plus 8/32-bit PCM and multi-channel input (averaged)."""
    try:
        with wave.open(io.BytesIO(data), "rb") as wf:
            ch, width, sr, n = wf.getnchannels(), wf.getsampwidth(), wf.getframerate(), wf.getnframes()
            raw = wf.readframes(n)
    except (wave.Error, EOFError) as e:
        raise ValueError(f"not a PCM WAV file: {e}") from e
    if width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32767.0  # inverse of the client's x * 32767 (:101)
    elif width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif width == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    else:
        raise ValueError(f"unsupported sample width {width}")
    if ch > 1:
        x = x[: (len(x) // ch) * ch].reshape(-1, ch).mean(axis=1)
    return np.ascontiguousarray(x, dtype=np.float32), int(sr)


def words_to_response(words: List[Dict[str, Any]], model_name: str = "") -> Dict[str, Any]:
    """Backend words (R:...streaming_pipeline.py:415-433: text/start/end relative to the buffer) -> the JSON the reference's
    client parses (:126-134 for the text, :286-337 for ``metadata.chunks``)."""
    chunks = [{"text": w["text"], "timestamp": [float(w["start"]), float(w["end"])]} for w in words]
    text = "".join(w["text"] for w in words).strip()
    out: Dict[str, Any] = {"transcription": text, "text": text, "metadata": {"chunks": chunks}}
    if model_name:
        out["model"] = model_name
    return out


def _default_scheduler_factory(session_backend, chunk_length_s):
    """The per-session scheduler is the reference's own, unmodified ``StreamingPipeline`` (rolling buffer, truncation,
    committed / uncommitted words: R:thestage_speechkit/streaming/streaming_pipeline.py:443-988), fed through its backend
    injection seam (:460, :561-562).  It lives in the reference package, so session routes need ``thestage_speechkit``
    importable (the gateway then runs inside the reference's environment) or an explicit ``scheduler_factory``."""
    try:
        from thestage_speechkit.streaming import StreamingPipeline
    except Exception as e:  # noqa: BLE001
        raise RuntimeError("session routes need the reference package `thestage_speechkit` (its StreamingPipeline is the "
                           "per-session scheduler) or create_app(scheduler_factory=...)") from e
    return StreamingPipeline(backend=session_backend, chunk_length_s=chunk_length_s, min_process_chunk_s=0.5, use_vad=False)


class HostBusy(Exception):
    """The host cannot take the request now (session table or request queue full, hub closed): HTTP 503 + Retry-After."""


class SessionHost:
    """Everything behind the routes for ONE GPU: the session table (per-session reference scheduler + lock + last use), the
    shared ``BatchingHub`` and, optionally, the shared voice-activity service.  ``thewhisper_amd.node.NodeRouter`` exposes
    the same methods over N such hosts in N processes (one per MI355X)."""

    def __init__(self, backend: Union[BatchingHub, AMDWhisperBackend], scheduler_factory=None, max_sessions: int = 1024,
                 session_ttl_s: float = 900.0, vad=None):
        import threading

        self.hub = backend if isinstance(backend, BatchingHub) else None
        self.base_backend = backend.backend if self.hub is not None else backend
        self.sample_rate = self.base_backend.sample_rate
        self.make_scheduler = scheduler_factory or _default_scheduler_factory
        self.max_sessions = int(max_sessions)
        self.session_ttl_s = float(session_ttl_s)
        self.vad = vad                      # thewhisper_amd.vad.VadService or None (sessions then run with use_vad=False)
        self.sessions: Dict[str, Dict[str, Any]] = {}
        self._creating = 0                  # sessions being built outside the table lock: they count against max_sessions
        self._lock = threading.Lock()
        # a bare backend has ONE tw_ctx, which is not thread-safe (include/thewhisper.h): serialise the engine calls of the
        # request threads.  Behind a hub the worker thread is the only caller.
        self._engine_lock = threading.Lock()

    # -- sessions ------------------------------------------------------------------------------------
    def _evict_idle(self, now: float) -> List[Dict[str, Any]]:
        """Sessions nobody ended (a client that went away) leave the table after ``session_ttl_s`` without a request.  Called
        with the table lock held; the caller ``_drop``s what is returned after releasing it."""
        dead = [k for k, v in self.sessions.items() if now - v["last_used"] > self.session_ttl_s and not v["lock"].locked()]
        return [self.sessions.pop(k) for k in dead]

    @staticmethod
    def _drop(sess: Dict[str, Any]):
        """Releases what a session holds.  The session is already out of the table; a request that is still running on it (an
        ``/end`` racing a ``/process``) finishes first - its scheduler may be asking the detector stream for frames.  A request
        that had looked the session up but not yet taken its lock finds it ``closed`` and answers 404: its detector slot may
        already belong to another session."""
        with sess["lock"]:
            sess["closed"] = True
            vs = sess.get("vad")
            if vs is not None:
                vs.close()

    def create(self, session_id: Optional[str] = None) -> str:
        import base64
        import os
        import threading
        import time

        sid = session_id or base64.urlsafe_b64encode(os.urandom(16)).decode("ascii")     # as R:examples/server.py:122
        now = time.monotonic()
        with self._lock:
            dead = self._evict_idle(now)
            full = len(self.sessions) + self._creating >= self.max_sessions
            if not full:
                self._creating += 1         # the scheduler is built outside the lock (it may be slow); the place is taken now
        for d in dead:
            self._drop(d)
        if full:
            raise HostBusy("too many sessions")
        sess = None
        try:
            session_backend = self.hub.stream_backend() if self.hub is not None else _LockedBackend(self.base_backend, self._engine_lock)
            sched = self.make_scheduler(session_backend, self.base_backend.chunk_length_s)
            vs = None
            if self.vad is not None:
                from .vad import attach_vad

                vs = self.vad.open_stream()
                try:
                    attach_vad(sched, vs)
                except BaseException:
                    vs.close()          # the detector slot must not outlive a session that was never created
                    raise
            sess = {"scheduler": sched, "lock": threading.Lock(), "last_used": now, "vad": vs, "closed": False}
        finally:
            with self._lock:
                self._creating -= 1
                if sess is not None:
                    self.sessions[sid] = sess
        return sid

    def _get(self, sid: str) -> Dict[str, Any]:
        import time

        with self._lock:
            s = self.sessions.get(sid)
            if s is not None:
                s["last_used"] = time.monotonic()
        if s is None:
            raise KeyError(sid)
        return s

    def add_chunk(self, sid: str, audio_np: np.ndarray) -> None:
        s = self._get(sid)
        with s["lock"]:
            if s["closed"]:
                raise KeyError(sid)
            sched, vs = s["scheduler"], s["vad"]
            if vs is not None and getattr(sched, "use_vad", False):
                # the reference evaluates its detector frame by frame on (left-over samples + this chunk)
                # (R:...streaming_pipeline.py:589-622): ask for all of those frames in ONE request, which the VadService merges
                # with the other sessions' into one launch; the scheduler's per-frame calls are then answered from it
                left = getattr(sched, "_vad_buffer", np.zeros(0, np.float32))
                vs.prefetch(np.concatenate([np.asarray(left, np.float32), np.asarray(audio_np, np.float32)]))
            sched.add_new_chunk(audio_np)

    def process(self, sid: str):
        s = self._get(sid)
        with s["lock"]:
            if s["closed"]:
                raise KeyError(sid)
            return s["scheduler"].process_new_chunk()    # blocks in the hub while the shared passes decode

    def clear(self, sid: str) -> None:
        s = self._get(sid)
        with s["lock"]:
            if s["closed"]:
                raise KeyError(sid)
            if hasattr(s["scheduler"], "clear"):
                s["scheduler"].clear()   # per-session state: clearing is safe here (the reference's shared pipeline leaves it commented out)
            if s["vad"] is not None:
                s["vad"].reset_states()

    def end(self, sid: str) -> None:
        with self._lock:
            s = self.sessions.pop(sid, None)
        if s is not None:
            self._drop(s)

    # -- stateless -----------------------------------------------------------------------------------
    def transcribe(self, audio: np.ndarray, sr: int) -> List[Dict[str, Any]]:
        import queue as _queue

        if self.hub is not None:
            try:
                fut = self.hub.submit(audio, 0.0, sr)
            except _queue.Full as e:
                raise HostBusy("request queue full") from e
            except RuntimeError as e:
                raise HostBusy(str(e)) from e
            return fut.result()
        with self._engine_lock:
            return self.base_backend.transcribe(audio, 0.0, sr)

    def health(self) -> Dict[str, Any]:
        return {"passes": (self.hub.passes if self.hub is not None else None), "rows": (self.hub.rows if self.hub is not None else None),
                "last_rows": (list(self.hub.batches)[-24:] if self.hub is not None else None),
                "turnaround_ms": (round(1e3 * self.hub._turn_ema, 1) if self.hub is not None else None),
                "sessions": len(self.sessions),
                "vad_launches": (self.vad.launches if self.vad is not None else None)}


class _LockedBackend:
    """``TranscriptionBackend`` view of a bare backend for request threads: one engine call at a time."""

    def __init__(self, backend, lock):
        self._b, self._lock = backend, lock

    def transcribe(self, audio, buffer_start_time, sample_rate):
        with self._lock:
            return self._b.transcribe(audio, buffer_start_time, sample_rate)


def create_app(backend, auth_token: str = "", model_name: str = "", lang_id: Optional[str] = None, path: str = "/transcribe",
               scheduler_factory=None, max_sessions: int = 1024, session_ttl_s: float = 900.0, vad=None,
               host_threads: Optional[int] = None):
    """FastAPI application.  ``backend``: a ``BatchingHub`` (concurrent requests share passes), a bare ``AMDWhisperBackend``, a
    ready ``SessionHost`` or a ``thewhisper_amd.node.NodeRouter`` (one host process per GPU).  ``auth_token``: when set,
    requests must carry ``Authorization: Bearer <token>``.  ``lang_id``: when set, a request's ``X-Lang-Id`` must equal it
    (the engine is built for one language prompt).  ``vad``: a ``thewhisper_amd.vad.VadService`` - sessions are then gated
    like the reference's default ``use_vad=True`` (energy rule, not silero - vad.py), all sessions' frames in shared launches.

    Two surfaces:
      * ``POST /transcribe`` - stateless, the wire format of the reference's remote backend (module docstring);
      * ``POST /session/create/``, ``/session/{id}/add_chunk``, ``/process``, ``/clear``, ``/end`` - the routes of the
        reference's demo server (R:examples/server.py:118-163) with the same request / response shapes, except that every
        session owns its scheduler state (the reference shares ONE StreamingPipeline across sessions, :25, :90, :98, and
        cannot batch) and all sessions of a GPU share the hub.  Sessions nobody ends are dropped after ``session_ttl_s``.
    Every route body runs in a thread pool: a session's lock is held for the whole batched decode of ``/process``, and
    taking it on the event loop would stall every other session (and defeat the batching the routes exist for).  The pool is
    this application's OWN (``host_threads`` workers, default ``min(max_sessions, 512)``): a ``/process`` call keeps its thread for
    a whole pass, so the pool bounds the requests in flight - Starlette's shared default pool has 40 threads, i.e. 5 rows per
    GPU in an 8-GPU node whose passes take 16."""
    import asyncio
    import base64
    import functools
    import queue as _queue
    from concurrent.futures import ThreadPoolExecutor

    from fastapi import FastAPI, Header, HTTPException, Request, WebSocket

    app = FastAPI(title="thewhisper-amd gateway")
    n_threads = int(host_threads or min(max(int(max_sessions), 64), 512))
    pool = ThreadPoolExecutor(max_workers=n_threads, thread_name_prefix="tw-gateway")
    app.state.host_threads = n_threads

    async def run_in_threadpool(fn, *a):
        return await asyncio.get_running_loop().run_in_executor(pool, functools.partial(fn, *a))

    if hasattr(backend, "create") and hasattr(backend, "add_chunk"):
        host = backend                       # SessionHost or NodeRouter
    else:
        host = SessionHost(backend, scheduler_factory=scheduler_factory, max_sessions=max_sessions, session_ttl_s=session_ttl_s, vad=vad)
    app.state.host = host
    sample_rate = host.sample_rate

    def check_auth(authorization: Optional[str]):
        if auth_token and not hmac.compare_digest((authorization or "").encode(), f"Bearer {auth_token}".encode()):
            raise HTTPException(status_code=401, detail="invalid or missing bearer token")

    async def guarded(fn, *a):
        """Run a host method - natively if the host has an async form of it (``NodeRouter.aprocess`` ...: nothing to block on in
        this process, the work happens in a worker process), in the thread pool otherwise - and map its failures as the
        reference's server does (:86-89, :128-133)."""
        try:
            afn = getattr(host, "a" + fn.__name__, None)
            if afn is not None:
                return await afn(*a)
            return await run_in_threadpool(fn, *a)
        except KeyError as e:
            raise HTTPException(status_code=404, detail=f"Session {e.args[0]} not found") from e
        except (HostBusy, _queue.Full) as e:      # session table / request queue full, hub closed
            raise HTTPException(status_code=503, detail=str(e) or "request queue full", headers={"Retry-After": "1"}) from e
        except HTTPException:
            raise
        except Exception as e:  # noqa: BLE001
            if type(e).__name__ == "WorkerGone":   # node.py: the GPU's serving process died under the request; its sessions are gone
                raise HTTPException(status_code=503, detail=str(e), headers={"Retry-After": "1"}) from e
            raise HTTPException(status_code=500, detail=str(e)) from e

    @app.post("/session/create/")
    async def session_create(authorization: Optional[str] = Header(default=None)):
        check_auth(authorization)
        try:
            return {"session_id": await run_in_threadpool(host.create)}
        except HostBusy as e:
            raise HTTPException(status_code=503, detail=str(e), headers={"Retry-After": "1"}) from e
        except RuntimeError as e:
            if type(e).__name__ == "WorkerGone":   # node.py: the chosen GPU's serving process died under the call
                raise HTTPException(status_code=503, detail=str(e), headers={"Retry-After": "1"}) from e
            raise HTTPException(status_code=500, detail=f"Failed to initialize model: {e}") from e

    @app.post("/session/{session_id}/end")
    async def session_end(session_id: str, authorization: Optional[str] = Header(default=None)):
        check_auth(authorization)
        await guarded(host.end, session_id)
        return {"status": "success"}

    @app.post("/session/{session_id}/add_chunk")
    async def session_add_chunk(session_id: str, audio_data: str, authorization: Optional[str] = Header(default=None)):
        """``audio_data``: base64 of float32 PCM, a query parameter exactly as in the reference (R:examples/server.py:135-144)."""
        check_auth(authorization)
        try:
            audio_np = np.frombuffer(base64.b64decode(audio_data), dtype=np.float32)
        except Exception as e:  # noqa: BLE001
            raise HTTPException(status_code=500, detail=str(e)) from e
        await guarded(host.add_chunk, session_id, audio_np)
        return {"status": "success"}

    @app.post("/session/{session_id}/process")
    async def session_process(session_id: str, authorization: Optional[str] = Header(default=None)):
        check_auth(authorization)
        words, uncommited_words = await guarded(host.process, session_id)
        return {"words": words, "uncommited_words": uncommited_words}   # key spelling as in the reference (:153)

    @app.post("/session/{session_id}/clear")
    async def session_clear(session_id: str, authorization: Optional[str] = Header(default=None)):
        check_auth(authorization)
        await guarded(host.clear, session_id)
        return {"status": "success"}

    @app.post(path)
    async def post_transcribe(request: Request, authorization: Optional[str] = Header(default=None),
                              x_lang_id: Optional[str] = Header(default=None), x_model_name: Optional[str] = Header(default=None)):
        check_auth(authorization)
        if lang_id and x_lang_id and x_lang_id != lang_id:
            raise HTTPException(status_code=400, detail=f"this gateway serves language '{lang_id}'")
        if model_name and x_model_name and x_model_name != model_name:
            raise HTTPException(status_code=404, detail=f"model '{x_model_name}' is not loaded (serving '{model_name}')")
        body = await request.body()
        ctype = request.headers.get("content-type", "")
        try:
            wav = body if ctype.startswith("audio/") else multipart_file(body, ctype)
            audio, sr = decode_wav(wav)
        except ValueError as e:
            raise HTTPException(status_code=400, detail=str(e)) from e
        if sr != sample_rate:
            raise HTTPException(status_code=400, detail=f"expected {sample_rate} Hz audio, got {sr} Hz")
        if len(audio) == 0:
            return words_to_response([], model_name)
        words = await guarded(host.transcribe, audio, sr)   # blocks until the (shared) passes have decoded it
        return words_to_response(words, model_name)

    @app.websocket("/ws/stream")
    async def ws_stream(ws: WebSocket):
        """One streaming session over a WebSocket (SURVEY.md section 8f rank 1 names the surface; the reference's server imports
        ``WebSocket`` without using it, R:examples/server.py:1).  Binary message = float32 PCM chunk at 16 kHz: it is added to
        the session (``add_chunk``) and processed (``process``); every message is answered with the JSON of the ``/process``
        route.  Text message "clear" / "end" as the routes.  ``?token=`` carries the bearer token.  Closing ends the session."""
        if auth_token and not hmac.compare_digest((ws.query_params.get("token") or "").encode(), auth_token.encode()):
            await ws.close(code=4401)
            return
        await ws.accept()
        try:
            sid = await run_in_threadpool(host.create)
        except Exception as e:  # noqa: BLE001
            await ws.send_json({"error": str(e)})
            await ws.close(code=1013)
            return
        try:
            while True:
                msg = await ws.receive()
                if msg["type"] == "websocket.disconnect":
                    break
                if msg.get("bytes") is not None:
                    audio_np = np.frombuffer(msg["bytes"], dtype=np.float32)
                    await guarded(host.add_chunk, sid, audio_np)
                    words, uncommited_words = await guarded(host.process, sid)
                    await ws.send_json({"words": words, "uncommited_words": uncommited_words})
                elif msg.get("text") == "clear":
                    await run_in_threadpool(host.clear, sid)
                    await ws.send_json({"status": "success"})
                elif msg.get("text") == "end":
                    break
        except Exception as e:  # noqa: BLE001
            try:
                await ws.send_json({"error": str(e)})
            except Exception:  # noqa: BLE001
                pass
        finally:
            await run_in_threadpool(host.end, sid)
            try:
                await ws.close()
            except Exception:  # noqa: BLE001
                pass

    @app.get("/health")
    async def health():
        h = await run_in_threadpool(host.health)
        return {"status": "ready", "model": model_name, "batches": h.get("passes"), **h}

    return app


def build_host(argv: Optional[List[str]] = None):
    """Parses the command line of ``python -m thewhisper_amd.gateway`` and builds what it serves: ``(host, args)`` with ``host``
    a ``SessionHost`` over a ``BatchingHub`` over ``AMDWhisperBackend(args.model, ...)`` - the reference's constructor form, a
    checkpoint NAME OR PATH (R:thestage_speechkit/nvidia/asr_pipeline.py:47-70) - or a ``NodeRouter`` for ``--gpus N``."""
    import argparse

    ap = argparse.ArgumentParser(description="MI355X Whisper gateway speaking the TheWhisper remote-backend wire format")
    ap.add_argument("--model", required=True, help="HF checkpoint name or path (e.g. TheStageAI/thewhisper-large-v3)")
    ap.add_argument("--chunk-length-s", type=int, default=10)
    ap.add_argument("--max-batch", type=int, default=16)
    ap.add_argument("--gpus", type=int, default=1, help="serving processes, one per MI355X (sessions sticky by index %% gpus; thewhisper_amd/node.py)")
    ap.add_argument("--vad", action="store_true", help="gate sessions with the on-device energy VAD (the reference's default use_vad=True; vad.py)")
    ap.add_argument("--host", default="0.0.0.0")
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--auth-token", default="")
    ap.add_argument("--language", default="en")
    ap.add_argument("--dtype", default=None, choices=[None, "bf16", "fp16", "fp32"], help="compute dtype of the engine context (default: fp16 - what the reference's streaming backend asks its platform "
                         "pipelines for, R:thestage_speechkit/streaming/streaming_pipeline.py:369-370; a float16 CONTEXT since round 4, "
                         "rounds 1-3 ran such requests in bf16: pass bf16 for that arithmetic)")
    ap.add_argument("--prefetch-cus", type=int, default=0,
                    help="compute units of the side stream that encodes, under the running pass's decode loop, the rows that sit that "
                         "pass out (arrivals; with more requests in flight than --max-batch also the chunks waiting for their next "
                         "seek iteration).  96 for deployments with about two sessions per pass row; 0 (default) = off")
    args = ap.parse_args(argv)
    import torch

    dtype = {None: None, "bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    if args.gpus > 1:
        from .node import NodeRouter

        host = NodeRouter(args.gpus, "thewhisper_amd.node:default_host_factory",
                          dict(model=args.model, chunk_length_s=args.chunk_length_s, max_batch=args.max_batch, language=args.language,
                               use_vad=args.vad, torch_dtype=args.dtype, prefetch_cus=args.prefetch_cus))
    else:
        backend = AMDWhisperBackend(args.model, chunk_length_s=args.chunk_length_s, language=args.language, batch_size=args.max_batch,
                                    torch_dtype=dtype)
        vad = None
        if args.vad:
            from .vad import VadService

            vad = VadService(max_streams=1024)
        host = SessionHost(BatchingHub(backend, max_batch=args.max_batch, prefetch_cus=args.prefetch_cus), vad=vad)
    return host, args


def main(argv: Optional[List[str]] = None):  # pragma: no cover - serves until interrupted
    import uvicorn

    host, args = build_host(argv)
    app = create_app(host, auth_token=args.auth_token, model_name=args.model, lang_id=args.language,
                     host_threads=max(64, 2 * args.gpus * args.max_batch))
    # add_chunk carries base64 PCM in the QUERY STRING, as the reference's route does (R:examples/server.py:135-144): 0.5 s of audio is
    # 43 KB of URL, above h11's default 16 KB request-line limit
    uvicorn.run(app, host=args.host, port=args.port, h11_max_incomplete_event_size=1 << 20)


if __name__ == "__main__":  # pragma: no cover
    main()
