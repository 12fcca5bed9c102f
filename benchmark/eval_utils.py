"""WER / CER / RTFx evaluation with the reference's metric definitions.

The reference's harness (R:benchmark/eval_utils.py:43-154, R:benchmark/run_evaluation.py:94-134) reports, per dataset,
corpus-level WER and CER after text normalisation (`evaluate`'s "wer"/"cer" = jiwer: total edit operations over all
utterances / total reference length) and RTFx = total audio seconds / generation wall time, then the mean over datasets.
Its data side (Open-ASR through `datasets`) and its normaliser / metric packages (`whisper_normalizer`, `evaluate`) need the
network or are not installed here, so this module keeps the DEFINITIONS and drops those dependencies: datasets are local
manifests (benchmark/run_evaluation.py), the edit distance is computed here, and the normaliser is `whisper_normalizer` when
importable, else a stated basic rule (lower case, punctuation -> space, whitespace collapsed).
"""
from __future__ import annotations

import re
import time
import unicodedata
from typing import Any, Callable, Dict, Iterable, List, Optional, Sequence


def _basic_normalize(text: str) -> str:
    text = unicodedata.normalize("NFKC", text).lower()
    text = re.sub(r"[<\[][^>\]]*[>\]]", "", text)          # bracketed annotations, as whisper's BasicTextNormalizer drops them
    text = "".join(" " if unicodedata.category(c)[0] in "MSP" else c for c in text)
    return re.sub(r"\s+", " ", text).strip()


def get_normalizer(language: Optional[str]) -> Callable[[str], str]:
    try:  # the reference's choice (R:benchmark/eval_utils.py:24-35) when the package is there
        if language == "en":
            from whisper_normalizer.english import EnglishTextNormalizer

            return EnglishTextNormalizer()
        from whisper_normalizer.basic import BasicTextNormalizer

        return BasicTextNormalizer(remove_diacritics=True)
    except Exception:  # noqa: BLE001
        return _basic_normalize


def edit_distance(a: Sequence, b: Sequence) -> int:
    """Levenshtein distance (substitutions, deletions, insertions all cost 1), O(len(a) * len(b)) time, O(len(b)) memory."""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i] + [0] * len(b)
        for j, y in enumerate(b, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y))
        prev = cur
    return prev[-1]


def corpus_wer(predictions: Iterable[str], references: Iterable[str]) -> float:
    """sum of word-level edit distances / sum of reference word counts (what `evaluate.load("wer")` returns)."""
    err = n = 0
    for p, r in zip(predictions, references):
        rw = r.split()
        err += edit_distance(p.split(), rw)
        n += len(rw)
    return err / n if n else 0.0


def corpus_cer(predictions: Iterable[str], references: Iterable[str]) -> float:
    """sum of character-level edit distances / sum of reference lengths (what `evaluate.load("cer")` returns)."""
    err = n = 0
    for p, r in zip(predictions, references):
        err += edit_distance(p, r)
        n += len(r)
    return err / n if n else 0.0


def compute_text_metrics(predictions: List[str], references: List[str], language: Optional[str]) -> Dict[str, float]:
    norm = get_normalizer(language)
    p, r = [norm(t) for t in predictions], [norm(t) for t in references]
    return {"wer": corpus_wer(p, r), "cer": corpus_cer(p, r)}


def evaluate_dataset(asr_generator: Callable, audio_list: List[Any], references: List[str], sampling_rate: int = 16000,
                     language: Optional[str] = "en", generate_kwargs: Optional[Dict[str, Any]] = None,
                     batch_size: int = 64) -> Dict[str, float]:
    """One dataset: batches of `batch_size` utterances through `asr_generator(list_of_arrays, generate_kwargs=...)` (an
    ASRPipeline), corpus WER / CER, RTFx = total audio s / generation wall s (R:benchmark/eval_utils.py:112-154)."""
    gk = dict(generate_kwargs or {})
    gk["language"] = language
    preds: List[str] = []
    start = time.time()
    for i in range(0, len(audio_list), max(1, int(batch_size))):
        out = asr_generator(audio_list[i : i + batch_size], generate_kwargs=gk)
        for item in (out if isinstance(out, list) else [out]):
            preds.append(item if isinstance(item, str) else item.get("text", "") if isinstance(item, dict) else str(item))
    gen_time = time.time() - start
    metrics = compute_text_metrics(preds, references, language)
    total_audio_s = float(sum(len(a) for a in audio_list)) / float(sampling_rate)
    metrics["dataset_duration_hours"] = total_audio_s / 3600.0
    if total_audio_s > 0 and gen_time > 0:
        metrics["rtfx"] = total_audio_s / gen_time
    return metrics


def mean_over_tasks(results: Dict[str, Dict[str, float]]) -> Dict[str, float]:
    out = {}
    for k in ("wer", "cer", "rtfx"):
        v = [m[k] for m in results.values() if m.get(k) is not None]
        if v:
            out[k] = sum(v) / len(v)
    return out
