"""-m gpu: the serving-side rows of SURVEY.md section 8f on the real engine: the HTTP gateway (remote-backend wire format of the
reference), the `thestage_speechkit.amd` package as shipped under integration/, and the ASRPipeline(engine=...) sharing."""
import io
import json
import os
import sys
import threading
import wave

import numpy as np
import pytest
import torch

from oracle import whisper_oracle as wo

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "pipeline_golden.json")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")


def _wav(audio, sr=16000):
    pcm = (np.clip(audio.astype(np.float32), -1.0, 1.0) * 32767.0).astype(np.int16)
    buf = io.BytesIO()
    with wave.open(buf, "wb") as wf:
        wf.setnchannels(1); wf.setsampwidth(2); wf.setframerate(sr); wf.writeframes(pcm.tobytes())
    return buf.getvalue()


def test_gateway_on_gpu_speaks_the_reference_wire_format():
    """What the reference's RemoteAPITimestampsBackend sends (R:thestage_speechkit/streaming/streaming_pipeline.py:93-153,
    :266-337: int16 WAV in a multipart field `file`) against the real engine behind the BatchingHub; concurrent requests
    share a batch; the words equal a direct backend call on the same (int16-quantised) audio."""
    fastapi = pytest.importorskip("fastapi")
    from fastapi.testclient import TestClient

    from tests.test_pipeline_glue import build_amd_pipeline, normalise
    from thewhisper_amd import AMDWhisperBackend
    from thewhisper_amd.gateway import create_app, decode_wav
    from thewhisper_amd.serving import BatchingHub

    pipe = build_amd_pipeline("micro", 10, 4, device="cuda", engine_factory=None)
    backend = AMDWhisperBackend(None, chunk_length_s=10, asr_pipeline=pipe)
    hub = BatchingHub(backend, max_batch=4, max_wait_s=0.5)
    client = TestClient(create_app(hub, auth_token="tok", model_name="micro", lang_id="en"))
    clips = [wo.synth_audio(16000 * 4 + 800 * k, 30 + k, "speechlike") for k in range(4)]
    direct = [normalise(backend.transcribe(decode_wav(_wav(c))[0], 0.0, 16000)) for c in clips]
    out = [None] * 4
    gate = threading.Barrier(4)

    def post(k):
        gate.wait()
        out[k] = client.post("/transcribe", files={"file": ("chunk.wav", _wav(clips[k]), "audio/wav")},
                             headers={"Authorization": "Bearer tok", "X-Lang-Id": "en", "X-Model-Name": "micro"})

    th = [threading.Thread(target=post, args=(k,)) for k in range(4)]
    [t.start() for t in th]
    [t.join(120) for t in th]
    hub.close()
    for k in range(4):
        assert out[k].status_code == 200, out[k].text
        body = out[k].json()
        assert [c["text"] for c in body["metadata"]["chunks"]] == [w["text"] for w in direct[k]]
        for c, w in zip(body["metadata"]["chunks"], direct[k]):
            assert abs(c["timestamp"][0] - w["start"]) < 1e-6 and abs(c["timestamp"][1] - w["end"]) < 1e-6
        assert body["transcription"] == "".join(w["text"] for w in direct[k]).strip()
    assert max(hub.batches) > 1
    assert client.post("/transcribe", files={"file": ("chunk.wav", _wav(clips[0]), "audio/wav")}).status_code == 401


def test_thestage_speechkit_amd_package_as_shipped():
    """integration/thestage_speechkit/amd is the file that goes into the reference tree; on this box (no reference checkout)
    it imports as a namespace package and hands out the MI355X ASRPipeline, which reproduces the reference's golden output."""
    if "thestage_speechkit" in sys.modules:
        pytest.skip("a real thestage_speechkit is already imported in this process")
    sys.path.insert(0, os.path.join(ROOT, "integration"))
    try:
        from thestage_speechkit.amd import ASRPipeline as AmdPlatformPipeline
    finally:
        sys.path.pop(0)
        for k in [k for k in sys.modules if k.startswith("thestage_speechkit")]:
            del sys.modules[k]
    from oracle import hf_reference as hr
    from tests.test_pipeline_glue import normalise
    from thewhisper_amd import ASRPipeline

    assert AmdPlatformPipeline is ASRPipeline
    g = json.load(open(GOLD))["micro_c10"]
    dims = wo.PRESETS[g["preset"]]
    pipe = AmdPlatformPipeline(hr.build_hf_model(dims, wo.make_weights(dims, 0)), feature_extractor=hr.build_feature_extractor(dims, 10),
                               tokenizer=hr.build_tokenizer(dims), chunk_length_s=10, device="cuda", torch_dtype=torch.float32,
                               batch_size=g["batch_size"])
    audio = wo.synth_audio(16000 * g["seconds"], g["seed"], g["kind"])
    gk = {"num_beams": 1, "do_sample": False, "use_cache": True, "language": "en", "max_new_tokens": g["max_new_tokens"]}
    out = normalise(pipe(audio.copy(), generate_kwargs=dict(gk), chunk_length_s=9, return_timestamps=True))
    assert out == g["outputs"]["True"]
    # the log-mel tensor the pipeline hands to generate() never left the device
    feats = pipe.feature_extractor(audio[:160000], sampling_rate=16000, return_tensors="pt", return_attention_mask=True)
    assert feats["input_features"].is_cuda and feats["input_features"].shape == (1, dims.n_mels, 1000)
    assert isinstance(pipe.feature_extractor(audio[:160000], sampling_rate=16000, return_tensors="np")["input_features"], np.ndarray)


def test_eval_harness_drives_the_gpu_pipeline():
    """the eval port (benchmark/eval_utils.py, the metric definitions of R:benchmark/eval_utils.py:112-154) through
    thewhisper_amd.ASRPipeline on the MI355X, called exactly as benchmark/run_evaluation.py calls it.  No transcribed corpus
    exists offline, so the references are built from the pipeline's own output: identical text -> WER = CER = 0; one word
    dropped from every reference -> the corpus WER that edit implies; RTFx = audio seconds / generation wall seconds."""
    sys.path.insert(0, os.path.join(ROOT, "benchmark"))
    try:
        from eval_utils import evaluate_dataset, get_normalizer, mean_over_tasks
    finally:
        sys.path.pop(0)
    from tests.test_pipeline_glue import build_amd_pipeline

    pipe = build_amd_pipeline("micro", 10, 4, device="cuda", engine_factory=None)
    audio = [wo.synth_audio(16000 * (4 + k), 70 + k, ["speechlike", "noise"][k % 2]) for k in range(6)]
    gk = {"num_beams": 1, "task": "transcribe", "do_sample": False, "max_new_tokens": 24}
    gen = lambda x, generate_kwargs: pipe(x, generate_kwargs=generate_kwargs, batch_size=4)  # noqa: E731  (run_evaluation.py:62)
    texts = [o["text"] for o in pipe([a.copy() for a in audio], generate_kwargs={**gk, "language": "en"}, batch_size=4)]
    norm = get_normalizer("en")
    assert all(len(norm(t).split()) >= 2 for t in texts), texts
    exact = evaluate_dataset(gen, [a.copy() for a in audio], texts, language="en", generate_kwargs=gk, batch_size=4)
    assert exact["wer"] == 0.0 and exact["cer"] == 0.0
    assert exact["rtfx"] > 1.0 and abs(exact["dataset_duration_hours"] * 3600 - sum(len(a) for a in audio) / 16000) < 1e-6
    refs = [" ".join(norm(t).split()[1:]) for t in texts]        # the hypothesis now has one inserted word per utterance
    ins = evaluate_dataset(gen, [a.copy() for a in audio], refs, language="en", generate_kwargs=gk, batch_size=4)
    assert abs(ins["wer"] - len(refs) / sum(len(r.split()) for r in refs)) < 1e-12
    assert mean_over_tasks({"a": exact, "b": ins})["wer"] == ins["wer"] / 2


def test_hub_prefetch_on_gpu_gives_the_direct_results():
    """BatchingHub(prefetch_cus=96): arrivals during a pass are encoded on the sibling context / CU-masked stream and adopted by the
    next pass.  Words equal direct backend calls; rows did take the prefetch route."""
    import time

    from tests.test_pipeline_glue import build_amd_pipeline, normalise
    from thewhisper_amd import AMDWhisperBackend
    from thewhisper_amd.serving import BatchingHub

    pipe = build_amd_pipeline("micro", 10, 4, device="cuda", engine_factory=None)
    backend = AMDWhisperBackend(None, chunk_length_s=10, asr_pipeline=pipe)
    clips = [wo.synth_audio(16000 * 6 + 1600 * k, 70 + k, ["speechlike", "noise", "sine"][k % 3]) for k in range(12)]
    direct = [normalise(backend.transcribe(c.copy(), 0.5 * k, 16000)) for k, c in enumerate(clips)]
    hub = BatchingHub(backend, max_batch=4, max_wait_s=0.002, prefetch_cus=96)
    futs = []
    for k, c in enumerate(clips):               # a steady trickle: most requests arrive while some pass is decoding
        futs.append(hub.submit(c.copy(), 0.5 * k, 16000))
        time.sleep(0.004)
    got = [normalise(f.result(timeout=300)) for f in futs]
    prefetched, rows = hub.prefetched, hub.rows
    hub.close()
    assert got == direct
    assert prefetched >= 1, (prefetched, rows)
    print(f"\nHUB PREFETCH: {prefetched} of {rows} rows were encoded under another pass's decode loop")
