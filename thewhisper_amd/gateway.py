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

__all__ = ["create_app", "decode_wav", "words_to_response", "multipart_file"]


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


def create_app(backend: Union[BatchingHub, AMDWhisperBackend], auth_token: str = "", model_name: str = "",
               lang_id: Optional[str] = None, path: str = "/transcribe", scheduler_factory=None, max_sessions: int = 1024):
    """FastAPI application.  ``backend``: a ``BatchingHub`` (concurrent requests share batches) or a bare
    ``AMDWhisperBackend``.  ``auth_token``: when set, requests must carry ``Authorization: Bearer <token>``.
    ``lang_id``: when set, a request's ``X-Lang-Id`` must equal it (the engine is built for one language prompt).

    Two surfaces:
      * ``POST /transcribe`` - stateless, the wire format of the reference's remote backend (module docstring);
      * ``POST /session/create/``, ``/session/{id}/add_chunk``, ``/process``, ``/clear``, ``/end`` - the routes of the
        reference's demo server (R:examples/server.py:118-163) with the same request / response shapes, except that every
        session owns its scheduler state (the reference shares ONE StreamingPipeline across sessions, :25, :90, :98, and
        cannot batch) and all sessions of the process share the hub, i.e. one batched engine call per tick.
        ``scheduler_factory(session_backend, chunk_length_s)`` builds the per-session scheduler (default: the reference's
        StreamingPipeline)."""
    import base64
    import os
    import threading

    from fastapi import FastAPI, Header, HTTPException, Request
    from fastapi.concurrency import run_in_threadpool

    app = FastAPI(title="thewhisper-amd gateway")
    hub = backend if isinstance(backend, BatchingHub) else None
    base_backend = backend.backend if hub is not None else backend
    sample_rate = base_backend.sample_rate
    sessions: Dict[str, Any] = {}
    sessions_lock = threading.Lock()
    make_scheduler = scheduler_factory or _default_scheduler_factory

    def check_auth(authorization: Optional[str]):
        if auth_token and not hmac.compare_digest((authorization or "").encode(), f"Bearer {auth_token}".encode()):
            raise HTTPException(status_code=401, detail="invalid or missing bearer token")

    def get_session(session_id: str):
        with sessions_lock:
            s = sessions.get(session_id)
        if s is None:
            raise HTTPException(status_code=404, detail=f"Session {session_id} not found")   # R:examples/server.py:86-89
        return s

    @app.post("/session/create/")
    async def session_create(authorization: Optional[str] = Header(default=None)):
        check_auth(authorization)
        session_id = base64.urlsafe_b64encode(os.urandom(16)).decode("ascii")     # as R:examples/server.py:122
        try:
            sched = make_scheduler(hub.stream_backend() if hub is not None else base_backend, base_backend.chunk_length_s)
        except RuntimeError as e:
            raise HTTPException(status_code=500, detail=f"Failed to initialize model: {e}") from e
        with sessions_lock:
            if len(sessions) >= max_sessions:
                raise HTTPException(status_code=503, detail="too many sessions")
            sessions[session_id] = {"scheduler": sched, "lock": threading.Lock()}
        return {"session_id": session_id}

    @app.post("/session/{session_id}/end")
    async def session_end(session_id: str, authorization: Optional[str] = Header(default=None)):
        check_auth(authorization)
        with sessions_lock:
            sessions.pop(session_id, None)
        return {"status": "success"}

    @app.post("/session/{session_id}/add_chunk")
    async def session_add_chunk(session_id: str, audio_data: str, authorization: Optional[str] = Header(default=None)):
        """``audio_data``: base64 of float32 PCM, a query parameter exactly as in the reference (R:examples/server.py:135-144)."""
        check_auth(authorization)
        s = get_session(session_id)
        try:
            audio_np = np.frombuffer(base64.b64decode(audio_data), dtype=np.float32)
            with s["lock"]:
                s["scheduler"].add_new_chunk(audio_np)
            return {"status": "success"}
        except HTTPException:
            raise
        except Exception as e:  # noqa: BLE001
            raise HTTPException(status_code=500, detail=str(e)) from e

    @app.post("/session/{session_id}/process")
    async def session_process(session_id: str, authorization: Optional[str] = Header(default=None)):
        check_auth(authorization)
        s = get_session(session_id)

        def work():
            with s["lock"]:
                return s["scheduler"].process_new_chunk()    # blocks in the hub while the shared batch is decoded

        try:
            words, uncommited_words = await run_in_threadpool(work)
            return {"words": words, "uncommited_words": uncommited_words}   # key spelling as in the reference (:153)
        except Exception as e:  # noqa: BLE001
            raise HTTPException(status_code=500, detail=str(e)) from e

    @app.post("/session/{session_id}/clear")
    async def session_clear(session_id: str, authorization: Optional[str] = Header(default=None)):
        check_auth(authorization)
        s = get_session(session_id)
        with s["lock"]:
            if hasattr(s["scheduler"], "clear"):
                s["scheduler"].clear()   # per-session state: clearing is safe here (the reference's shared pipeline leaves it commented out)
        return {"status": "success"}

    def transcribe(audio: np.ndarray, sr: int) -> List[Dict[str, Any]]:
        if hub is not None:
            return hub.submit(audio, 0.0, sr).result()
        return backend.transcribe(audio, 0.0, sr)

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
        words = await run_in_threadpool(transcribe, audio, sr)   # blocks until the (shared) batch has been decoded
        return words_to_response(words, model_name)

    @app.get("/health")
    async def health():
        return {"status": "ready", "model": model_name, "batches": (len(hub.batches) if hub is not None else None),
                "sessions": len(sessions)}

    return app


def main(argv: Optional[List[str]] = None):  # pragma: no cover - needs weights and a GPU
    import argparse

    import uvicorn

    ap = argparse.ArgumentParser(description="MI355X Whisper gateway speaking the TheWhisper remote-backend wire format")
    ap.add_argument("--model", required=True, help="HF checkpoint name or path (e.g. TheStageAI/thewhisper-large-v3)")
    ap.add_argument("--chunk-length-s", type=int, default=10)
    ap.add_argument("--max-batch", type=int, default=16)
    ap.add_argument("--host", default="0.0.0.0")
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--auth-token", default="")
    ap.add_argument("--language", default="en")
    args = ap.parse_args(argv)
    backend = AMDWhisperBackend(args.model, chunk_length_s=args.chunk_length_s, language=args.language, batch_size=args.max_batch)
    hub = BatchingHub(backend, max_batch=args.max_batch)
    uvicorn.run(create_app(hub, auth_token=args.auth_token, model_name=args.model, lang_id=args.language), host=args.host, port=args.port)


if __name__ == "__main__":  # pragma: no cover
    main()
