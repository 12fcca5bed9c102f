"""Host logic of thewhisper_amd/overlap.py without a GPU: the HIP entry points are stubbed, the engines are fakes.  Checks
the hand-off protocol (a context is never used by both stages at once, results come back in order, the encoder stage of
batch k+1 runs while batch k decodes, failures propagate)."""
import threading
import time
import types

import pytest
import torch

from thewhisper_amd import overlap as ov


class _FakeHip:
    """Stands in for overlap._Hip (the tw_stream_* / tw_event_* entry points of the library)."""

    def __init__(self):
        self.events, self.calls = {}, []
        self._n = 100

    def _new(self):
        self._n += 1
        return self._n

    def stream_create_masked(self, device, mask):
        self.calls.append(("stream", len(mask), [int(m) for m in mask]))
        return self._new()

    def event_create(self, device):
        return self._new()

    def event_record(self, ev, st):
        self.calls.append(("record", ev, st))

    def stream_wait_event(self, st, ev):
        self.calls.append(("wait", st, ev))

    def stream_synchronize(self, st):
        pass

    def event_destroy(self, ev):
        pass

    def stream_destroy(self, st):
        pass


class _FakeEngine:
    def __init__(self, name):
        self.name, self.raw_stream, self.device = name, None, torch.device("cpu")
        self.busy = threading.Lock()


@pytest.fixture()
def fake(monkeypatch):
    hip = _FakeHip()
    monkeypatch.setattr(ov, "_hiplib", lambda: hip)
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda d: types.SimpleNamespace(multi_processor_count=256))
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    return hip


def test_cu_masks_partition_the_chip(fake):
    o = ov.EncoderOverlap([_FakeEngine("a"), _FakeEngine("b")], encoder_cus=64)
    (_, w0, dec_mask), (_, w1, enc_mask), (_, w2, all_mask) = [c for c in fake.calls if c[0] == "stream"]
    assert w0 == w1 == w2 == 8
    dec_bits = sum(bin(m).count("1") for m in dec_mask)
    enc_bits = sum(bin(m).count("1") for m in enc_mask)
    assert (dec_bits, enc_bits) == (192, 64)
    assert all(a & b == 0 for a, b in zip(dec_mask, enc_mask))
    assert all(a | b == c for a, b, c in zip(dec_mask, enc_mask, all_mask))   # the first batch's encoder stage: the whole range
    o.close()
    with pytest.raises(ValueError):
        ov.EncoderOverlap([_FakeEngine("a")], encoder_cus=64)
    with pytest.raises(ValueError):
        ov.EncoderOverlap([_FakeEngine("a"), _FakeEngine("b")], encoder_cus=256)
    o2 = ov.EncoderOverlap([_FakeEngine("a"), _FakeEngine("b")], encoder_cus=32, cu_range=(128, 256))
    masks = [c[2] for c in fake.calls if c[0] == "stream"][-3:-1]
    assert sum(bin(m).count("1") for m in masks[0]) == 96 and all(m == 0 for m in masks[0][:4])
    o2.close()


def test_pipeline_order_exclusive_use_and_overlap(fake):
    engs = [_FakeEngine("a"), _FakeEngine("b")]
    o = ov.EncoderOverlap(engs, encoder_cus=64)
    log, overlapped = [], []
    decoding = threading.Event()

    def enc(e, b):
        assert e.busy.acquire(blocking=False), "context used by both stages at once"
        assert e.raw_stream in (o.s_enc, o.s_all)
        if decoding.is_set():
            overlapped.append(b)
        time.sleep(0.01)
        log.append(("enc", b, e.name, e.raw_stream))
        e.busy.release()
        return b * 10

    def dec(e, b, encoded):
        assert encoded == b * 10
        assert e.busy.acquire(blocking=False), "context used by both stages at once"
        assert e.raw_stream == o.s_dec
        decoding.set()
        time.sleep(0.03)
        decoding.clear()
        log.append(("dec", b, e.name))
        e.busy.release()
        return b + 1000

    out = o.run(list(range(7)), enc, dec)
    assert out == [1000 + i for i in range(7)]
    assert [x[1] for x in log if x[0] == "dec"] == list(range(7))
    assert [x[2] for x in log if x[0] == "dec"] == ["a", "b", "a", "b", "a", "b", "a"]
    first_enc = next(x for x in log if x[0] == "enc" and x[1] == 0)
    assert first_enc[3] == o.s_all                      # pipeline fill: batch 0 encodes on every CU of the range (nothing else runs yet)
    assert all(x[3] == o.s_enc for x in log if x[0] == "enc" and x[1] > 0)
    assert len(overlapped) >= 3                         # later encoder stages ran while a decode was in progress
    assert all(e.raw_stream is None for e in engs)      # handed back to torch's current stream
    # every decode waited on the event its encoder stage recorded
    waits = [c for c in fake.calls if c[0] == "wait"]
    assert len(waits) == 7 and all(w[1] == o.s_dec for w in waits)
    o.close()


def test_failures_propagate(fake):
    o = ov.EncoderOverlap([_FakeEngine("a"), _FakeEngine("b")], encoder_cus=64)

    def bad_enc(e, b):
        if b == 2:
            raise RuntimeError("encoder stage failed")
        return None

    with pytest.raises(RuntimeError, match="encoder stage failed"):
        o.run(list(range(4)), bad_enc, lambda e, b, x: b)

    def bad_dec(e, b, x):
        raise ValueError("decode failed")

    with pytest.raises(ValueError, match="decode failed"):
        o.run(list(range(3)), lambda e, b: None, bad_dec)
    o.close()
