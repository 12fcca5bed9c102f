"""benchmark/eval_utils.py: corpus WER / CER / RTFx with the reference's definitions (R:benchmark/eval_utils.py:43-154)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "benchmark"))
import eval_utils as eu  # noqa: E402


def test_edit_distance_and_corpus_rates():
    assert eu.edit_distance("kitten", "sitting") == 3 and eu.edit_distance("", "abc") == 3 and eu.edit_distance("abc", "abc") == 0
    assert eu.edit_distance("a b c".split(), "a x c d".split()) == 2
    # corpus level: errors and lengths are summed BEFORE dividing (jiwer / evaluate), not averaged per utterance
    refs = ["the cat sat on the mat", "hello"]
    hyps = ["the cat sat on mat", "hello there world"]
    assert abs(eu.corpus_wer(hyps, refs) - (1 + 2) / (6 + 1)) < 1e-12
    assert abs(eu.corpus_cer(["abcd"], ["abxd"]) - 0.25) < 1e-12
    assert eu.corpus_wer([], []) == 0.0


def test_basic_normalizer_and_metrics():
    n = eu._basic_normalize
    assert n("Hello,   WORLD! [noise] It's <unk> fine.") == "hello world it s fine"
    m = eu.compute_text_metrics(["Hello, world!"], ["hello world"], "xx")
    assert m["wer"] == 0.0 and m["cer"] == 0.0


def test_evaluate_dataset_batches_and_rtfx():
    calls = []

    def fake_pipe(batch, generate_kwargs):
        calls.append((len(batch), dict(generate_kwargs)))
        return [{"text": "a b c"} for _ in batch]

    audio = [np.zeros(16000 * 2, np.float32) for _ in range(5)]
    m = eu.evaluate_dataset(fake_pipe, audio, ["a b c"] * 4 + ["a b d"], language="en", generate_kwargs={"num_beams": 1}, batch_size=2)
    assert [c[0] for c in calls] == [2, 2, 1] and calls[0][1]["language"] == "en"
    assert abs(m["wer"] - 1 / 15) < 1e-12 and abs(m["dataset_duration_hours"] - 10 / 3600) < 1e-12 and m["rtfx"] > 0
    assert eu.mean_over_tasks({"a": {"wer": 0.1, "rtfx": 10.0}, "b": {"wer": 0.3, "cer": 0.2, "rtfx": 30.0}}) == {"wer": 0.2, "cer": 0.2, "rtfx": 20.0}
