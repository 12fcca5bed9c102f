"""Chunk-merge fix the reference installs at import time (R:thestage_speechkit/__init__.py:5-139).

HF's ``_find_longest_common_sequence`` (HF:models/whisper/tokenization_whisper.py:1153-1270) compares
``(start, end)`` timestamp tuples with ``<=`` when merging strided chunks with word timestamps; when the
left token's end timestamp is ``None`` (an open-ended last word) that raises ``TypeError``.  The
reference replaces the comparison with one that treats such a token as "in order".

To keep a single source of truth for the (long) merge routine we do not restate HF's function: we
re-compile HF's own source with the two tuple comparisons routed through ``_ordered`` below, and
install the result over the private symbol exactly like the reference does.  When running inside the
reference package (``thestage_speechkit`` already imported) its own patch is left in place.
"""
from __future__ import annotations

import inspect
import logging
import re
import sys
import textwrap

from transformers.models.whisper import tokenization_whisper as _tw


def _ordered(left, right) -> bool:
    """The reference's ``compare`` (R:thestage_speechkit/__init__.py:75-78): a left token whose end timestamp is
    ``None`` (open-ended last word) counts as "in order"; otherwise the plain tuple comparison HF uses."""
    if left[1] is None:
        return True
    return left <= right


_PATTERN = re.compile(
    r"(left_token_timestamp_sequence\[left_start \+ idx\])\s*<=\s*"
    r"(token_timestamp_sequences\[seq_idx \+ 1\]\[right_start \+ idx\])"
)


def _build_patched():
    src = textwrap.dedent(inspect.getsource(_tw._find_longest_common_sequence))
    # HF 5.15.0 :1229-1232  `left_token_timestamp_sequence[left_start + idx] <= token_timestamp_sequences[seq_idx + 1][...]`
    new_src, n = _PATTERN.subn(r"_ordered(\1, \2)", src)
    if n != 1:
        return None
    ns = dict(_tw.__dict__)
    ns["_ordered"] = _ordered
    exec(compile(new_src, _tw.__file__, "exec"), ns)  # noqa: S102 - HF's own source, one operator rewritten
    return ns["_find_longest_common_sequence"]


def install() -> bool:
    if "thestage_speechkit" in sys.modules:  # the reference already installed its version
        return False
    if getattr(_tw._find_longest_common_sequence, "_thewhisper_patched", False):
        return False
    try:
        fn = _build_patched()
    except Exception:  # noqa: BLE001 - source unavailable (frozen build) or not compilable
        fn = None
    if fn is None:
        # An untested transformers layout: keep HF's own function rather than failing the import.  Only the corner the
        # reference patches is affected (merging strided chunks with WORD timestamps when a left token's end timestamp is
        # None raises TypeError inside HF); everything else behaves as upstream.  Tested range: transformers 5.15.x.
        STATUS["installed"] = False
        logging.getLogger(__name__).warning(
            "thewhisper_amd.lcs_patch: transformers %s has a different _find_longest_common_sequence; the reference's "
            "chunk-merge fix (R:thestage_speechkit/__init__.py:75-94) was NOT applied (tested with transformers 5.15.x)",
            getattr(sys.modules.get("transformers"), "__version__", "?"))
        return False
    fn._thewhisper_patched = True
    _tw._find_longest_common_sequence = fn
    STATUS["installed"] = True
    return True


STATUS = {"installed": None}   # True: our re-compiled function is active; False: HF layout not recognised (warning logged)
install()
