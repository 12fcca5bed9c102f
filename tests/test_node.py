"""Node-level front end (thewhisper_amd/node.py, BASELINE config 4's layout) on CPU: two worker PROCESSES (the micro model on
the oracle-backed engine each) behind one NodeRouter / one FastAPI app.  Sessions are sticky (index % world), each rank
batches ITS sessions, results equal a direct backend call, a dead session id is a 404 whichever rank it would hash to."""
import base64
import threading

import numpy as np
import pytest
import torch

from oracle import whisper_oracle as wo

torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def router():
    from thewhisper_amd.node import NodeRouter

    r = NodeRouter(2, "tests.node_factory:make_host", {"max_batch": 4}, start_timeout_s=300)
    yield r
    r.close()


def test_sessions_are_sticky_and_batched_per_rank(router):
    fastapi = pytest.importorskip("fastapi")
    from fastapi.testclient import TestClient

    from tests.test_pipeline_glue import build_amd_pipeline, normalise
    from thewhisper_amd import AMDWhisperBackend
    from thewhisper_amd.gateway import create_app

    client = TestClient(create_app(router, model_name="micro"))
    sids = [client.post("/session/create/").json()["session_id"] for _ in range(6)]
    assert [router.rank_of(s) for s in sids] == [0, 1, 0, 1, 0, 1]          # dist.shard_streams: index % world
    clips = [wo.synth_audio(48000 + 1600 * k, 60 + k, "speechlike") for k in range(6)]
    out = [None] * 6
    gate = threading.Barrier(6)

    def run(k):
        for i in range(0, len(clips[k]), 8000):      # small chunks, as the reference's client sends them (a query parameter)
            q = base64.b64encode(clips[k][i : i + 8000].astype(np.float32).tobytes()).decode()
            assert client.post(f"/session/{sids[k]}/add_chunk", params={"audio_data": q}).status_code == 200
        gate.wait()
        out[k] = client.post(f"/session/{sids[k]}/process").json()

    th = [threading.Thread(target=run, args=(k,)) for k in range(6)]
    [t.start() for t in th]
    [t.join(600) for t in th]
    direct = AMDWhisperBackend(None, chunk_length_s=10, asr_pipeline=build_amd_pipeline("micro", 10, 1))
    for k in range(6):
        assert normalise(out[k]["uncommited_words"]) == normalise(direct.transcribe(clips[k], 0.0, 16000))
    h = client.get("/health").json()
    assert h["world"] == 2 and [r["sessions"] for r in h["ranks"]] == [3, 3]
    assert all(r["passes"] >= 1 for r in h["ranks"])                          # both GPUs' hubs decoded
    assert h["passes"] < 6 + 2                                               # ... in shared passes, not one per session (+ warm-ups)
    # stickiness survives: the same session keeps answering from its rank; ending it frees it there only
    assert client.post(f"/session/{sids[0]}/end").json() == {"status": "success"}
    assert client.post(f"/session/{sids[0]}/process").status_code == 404
    assert [r["sessions"] for r in client.get("/health").json()["ranks"]] == [2, 3]
    # stateless requests are spread over the ranks
    w0 = router.transcribe(clips[0], 16000)
    w1 = router.transcribe(clips[0], 16000)
    assert normalise(w0) == normalise(w1) == normalise(direct.transcribe(clips[0], 0.0, 16000))


def test_a_dead_worker_takes_only_its_sessions_with_it(router):
    """Rank 1's process is killed: its sessions answer 404, a request in flight fails with WorkerGone (503 at the HTTP layer), new
    sessions and stateless requests go to the surviving rank, health says who is alive, close() does not hang."""
    fastapi = pytest.importorskip("fastapi")
    from fastapi.testclient import TestClient

    from thewhisper_amd.gateway import create_app
    from thewhisper_amd.node import WorkerGone

    client = TestClient(create_app(router, model_name="micro"))
    a, b = router.create(), router.create()
    ra, rb = router.rank_of(a), router.rank_of(b)
    assert {ra, rb} == {0, 1}
    dead_sid, live_sid = (a, b) if ra == 1 else (b, a)
    router.workers[1].proc.kill()
    router.workers[1].proc.join(30)
    for _ in range(200):                      # the reader thread notices the closed pipe
        if not router.workers[1].alive:
            break
        import time
        time.sleep(0.05)
    assert not router.workers[1].alive
    with pytest.raises(WorkerGone):
        router.call(1, "health")
    assert client.post(f"/session/{dead_sid}/process").status_code == 404
    clip = wo.synth_audio(16000, 5, "speechlike")
    q = base64.b64encode(clip[:8000].astype(np.float32).tobytes()).decode()
    assert client.post(f"/session/{live_sid}/add_chunk", params={"audio_data": q}).status_code == 200
    assert client.post(f"/session/{live_sid}/process").status_code == 200
    h = client.get("/health").json()
    assert h["alive"] == 1 and [r["alive"] for r in h["ranks"]] == [True, False]
    new = [client.post("/session/create/").json()["session_id"] for _ in range(3)]
    assert [router.rank_of(s) for s in new] == [0, 0, 0]
    assert isinstance(router.transcribe(clip, 16000), list) and isinstance(router.transcribe(clip, 16000), list)
    assert client.post(f"/session/{dead_sid}/end").json() == {"status": "success"}      # ending a lost session is not an error
    for s in new + [live_sid]:
        router.end(s)
