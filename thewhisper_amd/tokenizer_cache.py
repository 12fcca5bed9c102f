"""Cache of ``tokenizer.all_special_ids`` for the lifetime of a pipeline.

HF's ``WhisperTokenizer._decode_with_timestamps`` evaluates the ``all_special_ids`` property - a ``convert_tokens_to_ids`` over
every special token - once per decoded word when word timestamps are requested (``_split_tokens_on_unicode``,
HF:models/whisper/tokenization_whisper.py:279-286, :1315-1345).  Measured on the MI355X box for one batched pipeline call of 16
ten-second streams: 1756 evaluations = 111 ms of the 149 ms the whole post-processing takes (11 % of the call).  The value
only depends on the tokenizer's special-token tables, which do not change while a pipeline lives; the cached subclass
re-derives it whenever those tables change, so results are identical by construction.
"""
from __future__ import annotations


def _key(tok):
    return (len(tok._extra_special_tokens), tuple(str(v) for v in tok._special_tokens_map.values()))


def cache_special_ids(tokenizer):
    cls = type(tokenizer)
    if getattr(cls, "_tw_special_id_cache", False) or not hasattr(tokenizer, "_extra_special_tokens") \
            or not hasattr(tokenizer, "_special_tokens_map"):
        return tokenizer
    base_ids = cls.all_special_ids.fget if isinstance(getattr(cls, "all_special_ids", None), property) else None
    if base_ids is None:
        return tokenizer

    def all_special_ids(self):
        k = _key(self)
        hit = self.__dict__.get("_tw_special_ids")
        if hit is None or hit[0] != k:
            hit = (k, list(base_ids(self)))
            self.__dict__["_tw_special_ids"] = hit
        return hit[1]

    cached = type(cls.__name__, (cls,), {"all_special_ids": property(all_special_ids), "_tw_special_id_cache": True})
    try:
        tokenizer.__class__ = cached
    except TypeError:   # exotic tokenizer classes that cannot be re-classed keep HF's behaviour
        pass
    return tokenizer
