#!/usr/bin/env python3
"""Adds the ``amd`` platform to a checkout of TheStageAI/TheWhisper.

    python integration/apply.py --reference /path/to/TheWhisper [--out /path/to/patched/copy]

Without ``--out`` the checkout is patched in place; with it, ``thestage_speechkit/`` and ``examples/`` are copied there
first (the reference tree itself is never written to).  Two changes, both idempotent:

1. ``thestage_speechkit/amd/__init__.py``  <- integration/thestage_speechkit/amd/__init__.py
2. ``thestage_speechkit/streaming/streaming_pipeline.py``: the platform switch of ``LocalWhisperBackend.__init__``
   (R:thestage_speechkit/streaming/streaming_pipeline.py:358-367) gains

        elif platform == "amd":
            from ..amd import ASRPipeline

            device = "cuda"

   (ROCm torch exposes the MI355X as "cuda", like the nvidia branch; THEWHISPER_DEVICE selects another device.)

``examples/run_streaming.py --platform amd`` and ``StreamingPipeline(platform="amd")`` then work as on the other platforms.
"""
from __future__ import annotations

import argparse
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ANCHOR = '        else:\n            raise ValueError(f"Invalid platform: {platform}")'
BRANCH = ('        elif platform == "amd":\n'
          '            from ..amd import ASRPipeline\n'
          '\n'
          '            device = "cuda"\n')


def apply(reference: str, out: str | None = None) -> str:
    root = reference
    if out:
        os.makedirs(out, exist_ok=True)
        for sub in ("thestage_speechkit", "examples"):
            src = os.path.join(reference, sub)
            if os.path.isdir(src):
                shutil.copytree(src, os.path.join(out, sub), dirs_exist_ok=True,
                                ignore=shutil.ignore_patterns("__pycache__", "*.pyc"))
        for dp, dns, fns in os.walk(out):            # a copy of a read-only checkout keeps its modes: make the copy writable
            for n in dns + fns:
                q = os.path.join(dp, n)
                os.chmod(q, os.stat(q).st_mode | 0o200)
        root = out
    pkg = os.path.join(root, "thestage_speechkit")
    if not os.path.isdir(pkg):
        raise SystemExit(f"{pkg} not found: not a TheWhisper checkout")
    os.makedirs(os.path.join(pkg, "amd"), exist_ok=True)
    shutil.copyfile(os.path.join(HERE, "thestage_speechkit", "amd", "__init__.py"), os.path.join(pkg, "amd", "__init__.py"))
    sp = os.path.join(pkg, "streaming", "streaming_pipeline.py")
    text = open(sp).read()
    if 'platform == "amd"' not in text:
        if text.count(ANCHOR) != 1:
            raise SystemExit("the platform switch of LocalWhisperBackend.__init__ was not found (reference layout changed)")
        text = text.replace(ANCHOR, BRANCH + ANCHOR)
        os.chmod(sp, os.stat(sp).st_mode | 0o200)     # (a copy of a read-only checkout keeps its modes)
        open(sp, "w").write(text)
    return root


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", required=True)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    print(apply(a.reference, a.out))
