"""Host-side caches of tokenizer work for the lifetime of a pipeline: ``all_special_ids`` and short ``decode`` calls.

HF's ``WhisperTokenizer._decode_with_timestamps`` evaluates the ``all_special_ids`` property - a ``convert_tokens_to_ids`` over
every special token - once per decoded word when word timestamps are requested (``_split_tokens_on_unicode``,
HF:models/whisper/tokenization_whisper.py:279-286, :1315-1345).  Measured on the MI355X box for one batched pipeline call of 16
ten-second streams: 1756 evaluations = 111 ms of the 149 ms the whole post-processing takes (11 % of the call).  The value
only depends on the tokenizer's special-token tables, which do not change while a pipeline lives; the cached subclass
re-derives it whenever those tables change, so results are identical by construction.

The same function decodes every token of a transcript on its own (``tokenizer.decode([id], decode_with_timestamps=True)``,
one call per token and a second one per partial UTF-8 sequence): ~840 calls of ~25 us per stream and engine call, i.e. most of
what is left of the post-processing.  Decoding a list of at most four ids with default options is a pure function of the ids,
the two flags and the tokenizer's tables, so the cached subclass memoises exactly those calls (same key discipline: the memo
is dropped when the special-token tables change; anything else - other argument types, other options - goes to HF's decode).
"""
from __future__ import annotations

import threading

_MEMO_MAX = 1 << 18
_tls = threading.local()   # .key = (tokenizer id, table key) while a `_decode_asr` call of this thread is in flight


def _frozen(v):
    # list-valued table entries are COPIED into the key (a list mutated in place would otherwise compare equal to itself)
    return tuple(_frozen(x) for x in v) if isinstance(v, (list, tuple)) else v


def _key(tok):
    # everything a cached value depends on, by value: the special-token tables (replacing, adding or mutating an entry -> new
    # key -> the caches are rebuilt), a counter the cached subclass bumps whenever tokens are added through the tokenizer's API
    # (add_tokens() with ordinary tokens changes decode results too; asking the Rust tokenizer for its vocabulary size costs
    # 6 ms per call - 250 x the memoised decode - so the mutation is caught where it happens instead) and the one decode option
    # that is read from the tokenizer instead of the call
    return (len(tok._extra_special_tokens), _frozen(list(tok._extra_special_tokens)), _frozen(list(tok._special_tokens_map.values())),
            tok.__dict__.get("_tw_vocab_version", 0), getattr(tok, "clean_up_tokenization_spaces", None))


def _cur_key(tok):
    """The table key, computed once per `_decode_asr` call (the ~2000 memoised decode() calls inside one post-processing pass
    cannot see the tables change: they run on one thread) and per call otherwise (the by-value key costs ~50 us)."""
    held = getattr(_tls, "key", None)
    if held is not None and held[0] == id(tok):
        return held[1]
    return _key(tok)


def cache_special_ids(tokenizer):
    cls = type(tokenizer)
    if getattr(cls, "_tw_special_id_cache", False) or not hasattr(tokenizer, "_extra_special_tokens") \
            or not hasattr(tokenizer, "_special_tokens_map"):
        return tokenizer
    base_ids = cls.all_special_ids.fget if isinstance(getattr(cls, "all_special_ids", None), property) else None
    if base_ids is None:
        return tokenizer
    base_decode = cls.decode

    def all_special_ids(self):
        k = _cur_key(self)
        hit = self.__dict__.get("_tw_special_ids")
        if hit is None or hit[0] != k:
            hit = (k, list(base_ids(self)))
            self.__dict__["_tw_special_ids"] = hit
        return hit[1]

    def decode(self, token_ids, skip_special_tokens=False, clean_up_tokenization_spaces=None, output_offsets=False,
               time_precision=0.02, decode_with_timestamps=False, normalize=False, basic_normalize=False,
               remove_diacritics=False, **kwargs):
        short = (type(token_ids) is list and 0 < len(token_ids) <= 4 and clean_up_tokenization_spaces is None
                 and not output_offsets and time_precision == 0.02 and not normalize and not basic_normalize
                 and not remove_diacritics and not kwargs and type(skip_special_tokens) is bool
                 and type(decode_with_timestamps) is bool)
        if short:
            for t in token_ids:
                if type(t) is not int:
                    short = False
                    break
        if not short:
            return base_decode(self, token_ids, skip_special_tokens=skip_special_tokens,
                               clean_up_tokenization_spaces=clean_up_tokenization_spaces, output_offsets=output_offsets,
                               time_precision=time_precision, decode_with_timestamps=decode_with_timestamps,
                               normalize=normalize, basic_normalize=basic_normalize, remove_diacritics=remove_diacritics,
                               **kwargs)
        k = _cur_key(self)
        memo = self.__dict__.get("_tw_decode_memo")
        if memo is None or memo[0] != k or len(memo[1]) > _MEMO_MAX:
            memo = (k, {})
            self.__dict__["_tw_decode_memo"] = memo
        mk = (tuple(token_ids), skip_special_tokens, decode_with_timestamps)
        text = memo[1].get(mk)
        if text is None:
            text = base_decode(self, list(token_ids), skip_special_tokens=skip_special_tokens,
                               decode_with_timestamps=decode_with_timestamps)
            if type(text) is str:
                memo[1][mk] = text
        return text

    members = {"all_special_ids": property(all_special_ids), "decode": decode, "_tw_special_id_cache": True}
    base_asr = getattr(cls, "_decode_asr", None)
    if base_asr is not None:
        def _decode_asr(self, *a, **k):   # the outer call of a pipeline's post-processing: validate the tables once for it
            prev = getattr(_tls, "key", None)
            _tls.key = (id(self), _key(self))
            try:
                return base_asr(self, *a, **k)
            finally:
                _tls.key = prev

        members["_decode_asr"] = _decode_asr
    base_add = getattr(cls, "_add_tokens", None)
    if base_add is not None:
        def _add_tokens(self, *a, **k):   # add_tokens() and add_special_tokens() both end here: new vocabulary -> new cache key
            self.__dict__["_tw_vocab_version"] = self.__dict__.get("_tw_vocab_version", 0) + 1
            return base_add(self, *a, **k)

        members["_add_tokens"] = _add_tokens
    cached = type(cls.__name__, (cls,), members)
    try:
        tokenizer.__class__ = cached
    except TypeError:   # exotic tokenizer classes that cannot be re-classed keep HF's behaviour
        pass
    return tokenizer
