"""SURVEY 8f-2, voice-activity gate.  CPU: the stated energy rule (oracle restatement) behaves as a VAD and, injected into the
reference's unmodified StreamingPipeline, gates which audio reaches the backend.  GPU (-m gpu): tw_vad_energy == the
restatement (probabilities within 2e-5, decisions at the reference's threshold 0.1 identical away from the threshold), batched
== per stream, silero's calling contract."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from oracle import whisper_oracle as wo

REF = "/root/reference"


def _talk_silence(seed=0, secs=(1.0, 2.0, 1.5, 2.0, 1.0)):
    """silence / speech / silence / speech / silence with a -60 dB noise bed"""
    rng = np.random.default_rng(seed)
    parts = []
    for i, s in enumerate(secs):
        n = int(16000 * s)
        bed = rng.standard_normal(n).astype(np.float32) * 1e-3
        parts.append(bed + (wo.synth_audio(n, seed + i, "speechlike") if i % 2 else 0))
    x = np.concatenate(parts).astype(np.float32)
    return x[: len(x) // 512 * 512]


def test_energy_rule_separates_speech_from_silence():
    x = _talk_silence()
    p, st = wo.energy_vad(x[None])
    frames_per = [int(16000 * s) // 512 for s in (1.0, 2.0, 1.5, 2.0)]
    b = np.cumsum([0] + frames_per)
    speech = p[0] > 0.1
    assert speech[b[0] + 2 : b[1] - 2].mean() < 0.05          # leading silence
    assert speech[b[1] + 2 : b[2] - 2].mean() > 0.7           # first utterance
    assert speech[b[2] + 8 : b[3] - 2].mean() < 0.2           # pause (after the floor settled)
    assert speech[b[3] + 2 : b[4] - 2].mean() > 0.7
    # state carries across calls: frame-by-frame == one bulk call
    st2, out = None, []
    for f in range(40):
        q, st2 = wo.energy_vad(x[None, f * 512 : (f + 1) * 512], st2)
        out.append(q[0, 0])
    assert np.array_equal(np.array(out, np.float32), p[0, :40])
    assert np.all(wo.energy_vad(np.zeros((1, 5120), np.float32))[0] == 0)   # digital silence is below the absolute gate


class _OracleVad:
    """silero-shaped callable over the restatement (tests only)."""

    def __init__(self):
        self.state = None

    def __call__(self, x, sr):
        p, self.state = wo.energy_vad(np.asarray(x, dtype=np.float32)[None], self.state)
        return torch.tensor(p[0, 0])


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_gate_under_the_reference_scheduler():
    """The reference's StreamingPipeline with its VAD state machine switched on (R:...:640-721): only audio around speech
    reaches the backend, and nothing is sent during the long trailing silence."""
    from oracle.make_golden import _import_reference

    _, sp = _import_reference()
    sent = []

    class Spy:
        def transcribe(self, audio, buffer_start_time, sample_rate):
            sent.append((len(audio), float(buffer_start_time)))
            return []

    s = sp.StreamingPipeline(backend=Spy(), chunk_length_s=10, min_process_chunk_s=0.5, use_vad=False)
    s.vad_model, s.use_vad = _OracleVad(), True          # what thewhisper_amd.vad.attach_vad does
    x = _talk_silence(3, (1.5, 2.5, 6.0))
    for i in range(0, len(x), 800):
        s(x[i : i + 800])
    total_sent = max(n for n, _ in sent) if sent else 0
    assert sent, "speech never reached the backend"
    assert total_sent < 16000 * 5.5                        # the 6 s of trailing silence were not appended to the buffer
    # with the gate off everything is forwarded
    sent_all = []

    class Spy2:
        def transcribe(self, audio, buffer_start_time, sample_rate):
            sent_all.append(len(audio))
            return []

    s2 = sp.StreamingPipeline(backend=Spy2(), chunk_length_s=10, min_process_chunk_s=0.5, use_vad=False)
    for i in range(0, len(x), 800):
        s2(x[i : i + 800])
    assert max(sent_all) > total_sent


@pytest.mark.gpu
def test_vad_kernel_matches_the_restatement():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from thewhisper_amd.vad import BatchedVAD, EnergyVAD, attach_vad

    xs = np.stack([_talk_silence(s)[: 512 * 200] for s in range(5)])
    xs[3] *= 0.01                                           # a quiet talker
    xs[4] = 0                                               # digital silence
    ref, ref_state = wo.energy_vad(xs)
    vad = BatchedVAD(5)
    got = vad.probs(xs).cpu().numpy()
    assert np.abs(got - ref).max() < 2e-5
    clear = np.abs(ref - 0.1) > 1e-3
    assert np.array_equal((got > 0.1)[clear], (ref > 0.1)[clear])
    assert np.abs(vad.state.cpu().numpy() - ref_state).max() < 1e-4
    # streaming contract: one 512-sample frame per call, state kept, .item()
    single = EnergyVAD()
    seq = [single(torch.from_numpy(xs[0, f * 512 : (f + 1) * 512]), 16000).item() for f in range(60)]
    assert np.abs(np.array(seq, np.float32) - ref[0, :60]).max() < 2e-5
    single.reset_states()
    assert abs(single(xs[0, :512], 16000).item() - ref[0, 0]) < 2e-5
    with pytest.raises(ValueError):
        single(xs[0, :400], 16000)
    # chunked batched calls == one bulk call (a serving tick gates all sessions in one launch)
    v2 = BatchedVAD(5)
    parts = [v2.probs(xs[:, i * 512 * 20 : (i + 1) * 512 * 20]).cpu().numpy() for i in range(10)]
    assert np.array_equal(np.concatenate(parts, axis=1), got)
    host = types.SimpleNamespace(vad_model=None, use_vad=False)
    attach_vad(host)
    assert host.use_vad and isinstance(host.vad_model, EnergyVAD)


# ---- VadService: all sessions' detectors behind one worker, frames of an add_chunk in one request --------------------------
def _np_kernel(pcm, state):
    return wo.energy_vad(pcm, state)


def test_vad_service_batches_sessions_and_equals_sequential_detectors():
    import threading

    from thewhisper_amd.vad import VadService

    svc = VadService(max_streams=8, window_s=0.05, kernel=_np_kernel)
    xs = [_talk_silence(s)[: 512 * 40] for s in range(4)]
    streams = [svc.open_stream() for _ in range(4)]
    got = [[] for _ in range(4)]
    gate = threading.Barrier(4)

    def session(k):
        for i in range(0, len(xs[k]), 512 * 2):            # a tick = two frames, asked for in ONE request
            gate.wait()
            n = streams[k].prefetch(xs[k][i : i + 1024])
            assert n == 2
            for f in range(2):                             # the scheduler's frame-by-frame calls replay the prefetched frames
                got[k].append(streams[k](torch.from_numpy(xs[k][i + 512 * f : i + 512 * (f + 1)]), 16000).item())

    th = [threading.Thread(target=session, args=(k,)) for k in range(4)]
    [t.start() for t in th]
    [t.join(60) for t in th]
    for k in range(4):
        ref, _ = wo.energy_vad(xs[k][None])
        assert np.array_equal(np.array(got[k], np.float32), ref[0])
    assert svc.requests == 4 * 20 and svc.launches < svc.requests / 2      # sessions of a tick shared launches
    assert all(s.launch_requests == 20 for s in streams)                  # one request per tick, not one per frame
    # a frame that was not prefetched is evaluated directly; reset clears the slot's state
    streams[0].reset_states()
    p0 = streams[0](xs[0][:512], 16000).item()
    assert p0 == wo.energy_vad(xs[0][None, :512])[0][0, 0]
    # slots are recycled
    for s in streams:
        s.close()
    assert len(svc._free) == 8
    svc.close()


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_gateway_sessions_are_gated_by_the_shared_vad_service():
    """SessionHost + VadService under the reference's unmodified StreamingPipeline: the add_chunk route asks for all frames of a
    chunk at once (one service request per chunk), the scheduler's state machine sees exactly the per-frame decisions of a
    sequential detector, and silence does not reach the backend."""
    from oracle.make_golden import _import_reference
    from thewhisper_amd.gateway import SessionHost
    from thewhisper_amd.vad import VadService

    _, sp = _import_reference()
    sent = {0: [], 1: []}

    class Spy:
        sample_rate, chunk_length_s = 16000, 10

        def __init__(self, k=None):
            self.k = k

        def transcribe(self, audio, buffer_start_time, sample_rate):
            sent[self.k].append(len(audio))
            return []

    made = []

    def factory(backend, chunk_length_s):
        k = len(made)
        s = sp.StreamingPipeline(backend=Spy(k), chunk_length_s=chunk_length_s, min_process_chunk_s=0.5, use_vad=False)
        made.append(s)
        return s

    svc = VadService(max_streams=4, window_s=0.002, kernel=_np_kernel)
    host = SessionHost(Spy(), scheduler_factory=factory, vad=svc)
    a, b = host.create(), host.create()
    xa, xb = _talk_silence(3, (1.5, 2.5, 6.0)), _talk_silence(4, (0.5, 1.0, 3.0))
    for i in range(0, max(len(xa), len(xb)), 800):
        for sid, x in ((a, xa), (b, xb)):
            if i < len(x):
                host.add_chunk(sid, x[i : i + 800])
                host.process(sid)
    # reference run of the same state machine with a sequential detector
    want = []
    class RefSpy:
        def transcribe(self, audio, buffer_start_time, sample_rate):
            want.append(len(audio))
            return []

    s_ref = sp.StreamingPipeline(backend=RefSpy(), chunk_length_s=10, min_process_chunk_s=0.5, use_vad=False)
    s_ref.vad_model, s_ref.use_vad = _OracleVad(), True
    for i in range(0, len(xa), 800):
        s_ref(xa[i : i + 800])
    assert sent[0] == want and sent[0], "the gated session must send exactly what the sequentially gated scheduler sends"
    assert max(sent[0]) < 16000 * 5.5
    streams = [host.sessions[a]["vad"], host.sessions[b]["vad"]]
    assert streams[0].launch_requests <= len(range(0, len(xa), 800))      # <= one request per add_chunk
    host.end(a); host.end(b)
    assert len(svc._free) == 4
    svc.close()


@pytest.mark.gpu
def test_vad_service_on_gpu_equals_per_stream_detectors():
    """The serving form on the MI355X: requests of several sessions merged into shared tw_vad_energy launches (state gathered and
    scattered by slot) give every session exactly what its own sequential detector gives."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import threading

    from thewhisper_amd.vad import BatchedVAD, VadService

    xs = [_talk_silence(s)[: 512 * 60] for s in range(6)]
    want = []
    for x in xs:
        want.append(BatchedVAD(1).probs(x[None]).cpu().numpy()[0])
    svc = VadService(max_streams=16, window_s=0.02)
    streams = [svc.open_stream() for _ in range(6)]
    got = [[] for _ in range(6)]
    gate = threading.Barrier(6)

    def session(k):
        step = 512 * (1 + k % 3)                             # sessions tick with different chunk sizes: 1, 2 or 3 frames
        for i in range(0, len(xs[k]), step):
            gate.wait() if i == 0 else None
            n = streams[k].prefetch(xs[k][i : i + step])
            for f in range(n):
                got[k].append(streams[k](xs[k][i + 512 * f : i + 512 * (f + 1)], 16000).item())

    th = [threading.Thread(target=session, args=(k,)) for k in range(6)]
    [t.start() for t in th]
    [t.join(120) for t in th]
    for k in range(6):
        assert np.array_equal(np.array(got[k], np.float32), want[k]), k
    assert svc.launches < svc.requests                        # requests shared launches
    svc.close()
