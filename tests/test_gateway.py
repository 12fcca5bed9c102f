"""thewhisper_amd/gateway.py on CPU (oracle-backed engine): the reference's remote-backend wire format in, the same words out
as a direct backend call; concurrent requests share batches; when the reference checkout is present its own
RemoteAPITimestampsBackend is pointed at the gateway."""
import io
import os
import threading
import wave

import time

import numpy as np
import pytest
import torch

from oracle import whisper_oracle as wo
from tests.test_pipeline_glue import build_amd_pipeline, normalise

torch.set_grad_enabled(False)
fastapi = pytest.importorskip("fastapi")
from fastapi.testclient import TestClient  # noqa: E402


def wav_bytes(audio: np.ndarray, sr: int = 16000) -> bytes:
    """What the reference client sends (R:thestage_speechkit/streaming/streaming_pipeline.py:93-112): clip, * 32767, int16."""
    pcm = (np.clip(audio.astype(np.float32), -1.0, 1.0) * 32767.0).astype(np.int16)
    buf = io.BytesIO()
    with wave.open(buf, "wb") as wf:
        wf.setnchannels(1)
        wf.setsampwidth(2)
        wf.setframerate(sr)
        wf.writeframes(pcm.tobytes())
    return buf.getvalue()


@pytest.fixture(scope="module")
def served():
    from thewhisper_amd import AMDWhisperBackend
    from thewhisper_amd.gateway import create_app
    from thewhisper_amd.serving import BatchingHub

    pipe = build_amd_pipeline("micro", 10, 4)
    backend = AMDWhisperBackend(None, chunk_length_s=10, asr_pipeline=pipe)
    hub = BatchingHub(backend, max_batch=4, max_wait_s=0.3)
    app = create_app(hub, auth_token="s3cret", model_name="micro", lang_id="en")
    yield backend, hub, TestClient(app)
    hub.close()


def test_wav_roundtrip_and_multipart_parser():
    from thewhisper_amd.gateway import decode_wav, multipart_file

    a = wo.synth_audio(4000, 1, "speechlike")
    x, sr = decode_wav(wav_bytes(a))
    assert sr == 16000 and x.dtype == np.float32 and len(x) == 4000
    assert np.abs(x - np.clip(a, -1, 1)).max() <= 1.0 / 32767 + 1e-7     # one int16 step
    body = (b"--XyZ\r\nContent-Disposition: form-data; name=\"other\"\r\n\r\nabc\r\n"
            b"--XyZ\r\nContent-Disposition: form-data; name=\"file\"; filename=\"chunk.wav\"\r\nContent-Type: audio/wav\r\n\r\n"
            b"RIFF\r\n--binary\r\n\r\n--XyZ--\r\n")
    assert multipart_file(body, 'multipart/form-data; boundary="XyZ"') == b"RIFF\r\n--binary\r\n"
    with pytest.raises(ValueError):
        multipart_file(body, "application/json")
    with pytest.raises(ValueError):
        decode_wav(b"not a wav")


def test_gateway_returns_the_backend_words(served):
    backend, hub, client = served
    audio = wo.synth_audio(16000 * 7, 5, "speechlike")
    # the gateway sees the int16-quantised audio, so the direct call gets the same samples
    from thewhisper_amd.gateway import decode_wav

    q, _ = decode_wav(wav_bytes(audio))
    want = backend.transcribe(q.copy(), 0.0, 16000)
    r = client.post("/transcribe", files={"file": ("chunk.wav", wav_bytes(audio), "audio/wav")},
                    headers={"Authorization": "Bearer s3cret", "X-Lang-Id": "en", "X-Model-Name": "micro"})
    assert r.status_code == 200, r.text
    data = r.json()
    got = [{"text": c["text"], "start": c["timestamp"][0], "end": c["timestamp"][1]} for c in data["metadata"]["chunks"]]
    assert normalise(got) == normalise(want)
    assert data["transcription"] == "".join(w["text"] for w in want).strip() == data["text"]
    # raw audio/wav body is accepted too
    r2 = client.post("/transcribe", content=wav_bytes(audio), headers={"Authorization": "Bearer s3cret", "Content-Type": "audio/wav"})
    assert r2.status_code == 200 and r2.json()["metadata"] == data["metadata"]


def test_gateway_rejects_bad_requests(served):
    _, _, client = served
    audio = wo.synth_audio(16000, 2, "noise")
    f = {"file": ("chunk.wav", wav_bytes(audio), "audio/wav")}
    assert client.post("/transcribe", files=f).status_code == 401
    assert client.post("/transcribe", files=f, headers={"Authorization": "Bearer nope"}).status_code == 401
    ok = {"Authorization": "Bearer s3cret"}
    assert client.post("/transcribe", files=f, headers={**ok, "X-Lang-Id": "de"}).status_code == 400
    assert client.post("/transcribe", files=f, headers={**ok, "X-Model-Name": "other"}).status_code == 404
    assert client.post("/transcribe", files={"file": ("chunk.wav", wav_bytes(audio, 8000), "audio/wav")}, headers=ok).status_code == 400
    assert client.post("/transcribe", files={"file": ("chunk.wav", b"junk", "audio/wav")}, headers=ok).status_code == 400
    assert client.post("/transcribe", files={"nofile": ("x", b"y", "text/plain")}, headers=ok).status_code == 400
    assert client.get("/health").json()["status"] == "ready"


def test_concurrent_requests_share_batches(served):
    backend, hub, client = served
    clips = [wo.synth_audio(16000 * s, 10 + s, "speechlike") for s in (3, 5, 4, 6)]
    before = len(hub.batches)
    out = [None] * 4

    def go(i):
        out[i] = client.post("/transcribe", files={"file": ("chunk.wav", wav_bytes(clips[i]), "audio/wav")},
                             headers={"Authorization": "Bearer s3cret"}).json()

    th = [threading.Thread(target=go, args=(i,)) for i in range(4)]
    [t.start() for t in th]
    [t.join(300) for t in th]
    assert all(o is not None and "metadata" in o for o in out)
    assert max(list(hub.batches)[before:]) > 1            # different HTTP requests were decoded in one batched pass


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference checkout not present")
def test_reference_remote_backend_against_the_gateway(served, monkeypatch):
    """The reference's own client class, unmodified, talking to the gateway (httpx.post routed to the test client)."""
    from oracle.make_golden import _import_reference

    backend, hub, client = served
    _, sp = _import_reference()

    def fake_post(url, headers=None, files=None, timeout=None):
        return client.post("/transcribe", headers=headers, files=files)

    monkeypatch.setattr(sp.httpx, "post", fake_post)
    remote = sp.RemoteAPITimestampsBackend(api_url="http://gateway/transcribe", auth_token="s3cret", model_name="micro", lang_id="en")
    audio = wo.synth_audio(16000 * 8, 21, "speechlike")
    from thewhisper_amd.gateway import decode_wav

    q, _ = decode_wav(wav_bytes(audio))
    want = backend.transcribe(q.copy(), 12.5, 16000)          # what LocalWhisperBackend-equivalent code returns for this buffer
    got = remote.transcribe(audio, 12.5, 16000)
    assert normalise(got) == normalise(want)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference checkout not present")
def test_session_routes_mirror_the_reference_server(served):
    """R:examples/server.py:118-163: create / add_chunk (base64 float32 as a query parameter) / process / clear / end.  Two
    sessions stream the golden clip in lock step through their own scheduler state and the shared hub; the words each one
    gets equal the reference's single-stream run."""
    import base64
    import json

    from oracle.make_golden import _import_reference

    _import_reference()      # makes thestage_speechkit importable: the per-session scheduler is the reference's StreamingPipeline
    backend, hub, client = served
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pipeline_golden.json")))["streaming_micro_c10"]
    audio = wo.synth_audio(16000 * g["seconds"], g["seed"], g["kind"])[: 16000 * 6]
    H = {"Authorization": "Bearer s3cret"}
    assert client.post("/session/create/").status_code == 401
    sids = [client.post("/session/create/", headers=H).json()["session_id"] for _ in range(2)]
    assert len(set(sids)) == 2 and client.get("/health").json()["sessions"] == 2
    committed = {s: [] for s in sids}
    last = {s: [] for s in sids}
    step = g["step_samples"] * 10      # 0.5 s per request keeps the test short; the scheduler state machine is the same
    for i in range(0, len(audio), step):
        chunk = base64.b64encode(audio[i : i + step].astype(np.float32).tobytes()).decode("ascii")
        for s in sids:
            assert client.post(f"/session/{s}/add_chunk", params={"audio_data": chunk}, headers=H).json() == {"status": "success"}
        res = {}

        def proc(s):
            res[s] = client.post(f"/session/{s}/process", headers=H)

        th = [threading.Thread(target=proc, args=(s,)) for s in sids]
        [t.start() for t in th]
        [t.join(120) for t in th]
        for s in sids:
            assert res[s].status_code == 200, res[s].text
            body = res[s].json()
            assert set(body) == {"words", "uncommited_words"}
            committed[s] += body["words"]
            last[s] = body["uncommited_words"]
    assert committed[sids[0]] == committed[sids[1]] and last[sids[0]] == last[sids[1]]
    assert len(last[sids[0]]) + len(committed[sids[0]]) > 0
    assert all({"text", "start", "end"} <= set(w) for w in last[sids[0]])
    assert client.post(f"/session/{sids[0]}/clear", headers=H).json() == {"status": "success"}
    assert client.post(f"/session/{sids[0]}/end", headers=H).json() == {"status": "success"}
    assert client.post(f"/session/{sids[0]}/process", headers=H).status_code == 404
    assert client.post("/session/nope/add_chunk", params={"audio_data": ""}, headers=H).status_code == 404
    client.post(f"/session/{sids[1]}/end", headers=H)
    assert client.get("/health").json()["sessions"] == 0


def test_websocket_session_streams_chunks_and_answers_like_the_process_route():
    from fastapi.testclient import TestClient

    from tests.node_factory import TinyScheduler
    from thewhisper_amd import AMDWhisperBackend
    from thewhisper_amd.gateway import create_app
    from thewhisper_amd.serving import BatchingHub

    pipe = build_amd_pipeline("micro", 10, 2)
    backend = AMDWhisperBackend(None, chunk_length_s=10, asr_pipeline=pipe)
    hub = BatchingHub(backend, max_batch=2, max_wait_s=0.05)
    app = create_app(hub, auth_token="tok", scheduler_factory=TinyScheduler)
    client = TestClient(app)
    clip = wo.synth_audio(32000, 9, "speechlike")
    with client.websocket_connect("/ws/stream?token=tok") as ws:
        replies = []
        for i in range(0, len(clip), 8000):
            ws.send_bytes(clip[i : i + 8000].astype(np.float32).tobytes())
            replies.append(ws.receive_json())
        assert client.get("/health").json()["sessions"] == 1
        ws.send_text("clear")
        assert ws.receive_json() == {"status": "success"}
        ws.send_text("end")
    assert normalise(replies[-1]["uncommited_words"]) == normalise(backend.transcribe(clip, 0.0, 16000))
    assert normalise(replies[0]["uncommited_words"]) == normalise(backend.transcribe(clip[:8000], 0.0, 16000))   # one reply per chunk
    t_end = time.monotonic() + 5.0    # ("end" has no reply: the handler drops the session on its own clock)
    while client.get("/health").json()["sessions"] != 0 and time.monotonic() < t_end:
        time.sleep(0.02)
    assert client.get("/health").json()["sessions"] == 0                 # closing ended the session
    with pytest.raises(Exception):
        with client.websocket_connect("/ws/stream?token=wrong"):
            pass
    hub.close()


def test_session_table_limit_ttl_and_end_during_a_request():
    """SessionHost without an engine: places in the table are taken while a scheduler is still being built (a burst of creates
    cannot overshoot max_sessions), idle sessions leave after the TTL, and ending a session waits for the request that is
    running on it before the detector stream is released."""
    import time

    from thewhisper_amd.gateway import HostBusy, SessionHost

    class Backend:
        sample_rate, chunk_length_s = 16000, 10

        def transcribe(self, audio, t0, sr):
            return []

    gate, building = threading.Event(), threading.Event()

    class SlowScheduler:
        use_vad = True
        _vad_buffer = np.zeros(0, np.float32)

        def __init__(self, backend, chunk_length_s):
            building.set()
            gate.wait(10)

        def add_new_chunk(self, a):
            pass

        def process_new_chunk(self):
            in_request.set()
            release.wait(10)
            assert not vad.closed, "the detector stream was released under a running request"
            return [], []

    class VadStreamStub:
        closed = False

        def prefetch(self, a):
            pass

        def reset_states(self):
            pass

        def close(self):
            VadStreamStub.closed = True

    class VadStub:
        launches = 0

        def open_stream(self):
            return vad

    vad = VadStreamStub()
    in_request, release = threading.Event(), threading.Event()
    import thewhisper_amd.vad as vadmod

    host = SessionHost(Backend(), scheduler_factory=SlowScheduler, max_sessions=1, session_ttl_s=0.05, vad=VadStub())
    orig_attach, vadmod.attach_vad = vadmod.attach_vad, (lambda sched, vs: None)
    try:
        out = {}
        t = threading.Thread(target=lambda: out.setdefault("sid", host.create()))
        t.start()
        assert building.wait(10)
        with pytest.raises(HostBusy):      # the first session is not in the table yet, but its place is taken
            host.create()
        gate.set()
        t.join(10)
        sid = out["sid"]
        assert host.health()["sessions"] == 1
        # /end while /process is running: the stream is closed only after the request returned
        p = threading.Thread(target=lambda: out.setdefault("res", host.process(sid)))
        p.start()
        assert in_request.wait(10)
        e = threading.Thread(target=host.end, args=(sid,))
        e.start()
        time.sleep(0.05)
        assert not vad.closed and host.health()["sessions"] == 0
        release.set()
        p.join(10)
        e.join(10)
        assert vad.closed and out["res"] == ([], [])
        # TTL: an idle session makes room for a new one
        VadStreamStub.closed = False
        release.set()
        s2 = host.create()
        time.sleep(0.1)
        s3 = host.create()
        assert s2 != s3 and host.health()["sessions"] == 1 and VadStreamStub.closed
        with pytest.raises(KeyError):
            host.process(s2)
    finally:
        vadmod.attach_vad = orig_attach
