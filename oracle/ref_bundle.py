"""The reference checkout on a box that has none (TEST INFRASTRUCTURE - nothing under thewhisper_amd/ imports this).

`/root/reference` exists in the build container only; the GPU box gets a snapshot of this repository.  So that the parity
tests and bench.py's `cpu_baseline` leg can run THE REFERENCE'S OWN CODE on the MI355X box too - its scheduler
(R:thestage_speechkit/streaming/streaming_pipeline.py:443-988 `StreamingPipeline`), its stepper
(R:thestage_speechkit/streaming/streams.py:16-81 `ArrayStream`) and its pipeline class
(R:thestage_speechkit/nvidia/asr_pipeline.py:30-92) - `__graft_entry__.build()` packs the reference's Python package into ONE
compressed archive under `oracle/_ref/` (git-ignored: it never enters the history; not gpurun-ignored: it travels with the
snapshot like the built `.so`).  Nothing of it is copied into the tree as source; a consumer unpacks it into a temporary
directory at run time and imports it from there.

    reference_dir()     -> a directory that contains `thestage_speechkit/`, or None
    import_reference()  -> (ASRPipeline, streaming_pipeline module, streams module) of the reference, imported from there
"""
from __future__ import annotations

import hashlib
import io
import os
import sys
import tarfile
import tempfile
import types
from typing import Optional

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
BUNDLE = os.path.join(ROOT, "oracle", "_ref", "thestage_speechkit.tar.gz")
PACKAGE = "thestage_speechkit"


def _has_package(d: Optional[str]) -> bool:
    return bool(d) and os.path.isfile(os.path.join(d, PACKAGE, "__init__.py"))


def make_bundle(ref: str = REF, out: str = BUNDLE) -> Optional[str]:
    """Pack `<ref>/thestage_speechkit/**/*.py` (+ the licence) into `out`; None when there is no checkout to pack (the GPU box).
    Deterministic bytes for an unchanged checkout (sorted members, zeroed times), so rebuilding does not churn the snapshot."""
    if not _has_package(ref):
        return None
    members = []
    for base, dirs, files in os.walk(os.path.join(ref, PACKAGE)):
        dirs[:] = sorted(d for d in dirs if d != "__pycache__")
        for f in sorted(files):
            if f.endswith(".py"):
                members.append(os.path.relpath(os.path.join(base, f), ref))
    if os.path.isfile(os.path.join(ref, "LICENSE")):
        members.append("LICENSE")
    buf = io.BytesIO()
    with tarfile.open(fileobj=buf, mode="w:gz", compresslevel=9) as tar:
        for rel in members:
            ti = tar.gettarinfo(os.path.join(ref, rel), arcname=rel)
            ti.mtime, ti.uid, ti.gid, ti.uname, ti.gname = 0, 0, 0, "", ""
            with open(os.path.join(ref, rel), "rb") as fh:
                tar.addfile(ti, fh)
    data = buf.getvalue()
    # (gzip stamps its header with the current time: zero it, bytes 4..7)
    data = data[:4] + b"\0\0\0\0" + data[8:]
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not (os.path.isfile(out) and open(out, "rb").read() == data):
        with open(out, "wb") as fh:
            fh.write(data)
    return out


def reference_dir() -> Optional[str]:
    """Where the reference package can be imported from: $TW_REFERENCE_DIR, /root/reference, or the bundle unpacked into a
    temporary directory named after the bundle's digest (unpacked once per box)."""
    env = os.environ.get("TW_REFERENCE_DIR")
    if _has_package(env):
        return env
    if _has_package(REF):
        return REF
    if not os.path.isfile(BUNDLE):
        return None
    digest = hashlib.sha256(open(BUNDLE, "rb").read()).hexdigest()[:12]
    # (round-5 advice) a predictable name under a world-writable directory is only trusted when it is a directory THIS user owns and
    # nobody else can write to; anything else is ignored and the bundle is unpacked into a fresh private directory (mkdtemp: 0700)
    dst = os.path.join(tempfile.gettempdir(), f"tw_reference_{digest}_{os.getuid()}")

    def _mine(d):
        try:
            st = os.stat(d)
        except OSError:
            return False
        return st.st_uid == os.getuid() and not (st.st_mode & 0o022)

    if not (_mine(dst) and _has_package(dst)):
        tmp = tempfile.mkdtemp(prefix="tw_reference_", dir=tempfile.gettempdir())
        with tarfile.open(BUNDLE, "r:gz") as tar:
            for m in tar.getmembers():      # plain relative file members only
                if not m.isfile() or m.name.startswith("/") or ".." in m.name.split("/"):
                    raise RuntimeError(f"unexpected member in {BUNDLE}: {m.name}")
            try:
                tar.extractall(tmp, filter="data")
            except TypeError:               # (Python < 3.10.12 without the extraction filters)
                tar.extractall(tmp)
        if os.path.lexists(dst) and not _mine(dst):
            return tmp if _has_package(tmp) else None     # somebody else's directory sits on the name: use the private copy
        try:
            os.rename(tmp, dst)
        except OSError:                     # another process unpacked it meanwhile
            pass
    return dst if _has_package(dst) else None


def which() -> str:
    """For reports: where the reference came from on this box."""
    d = reference_dir()
    if d is None:
        return "absent"
    if d == REF:
        return "/root/reference (checkout)"
    if d == os.environ.get("TW_REFERENCE_DIR"):
        return f"$TW_REFERENCE_DIR = {d}"
    return "oracle/_ref bundle of /root/reference/thestage_speechkit (packed by __graft_entry__.build, unpacked to a temporary directory)"


def stub_audio_io() -> None:
    """`thestage_speechkit.streaming.streams` imports sounddevice and librosa (microphone / file input), absent here and not on
    the hot path.  transformers must be imported BEFORE the stubs (its availability probe trips on a module without a spec,
    SURVEY.md section 8c)."""
    import transformers  # noqa: F401

    if "sounddevice" not in sys.modules:
        sd = types.ModuleType("sounddevice")
        sd.InputStream = type("InputStream", (), {})
        sys.modules["sounddevice"] = sd
    if "librosa" not in sys.modules:
        lb = types.ModuleType("librosa")
        lb.load = lb.resample = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("librosa stub"))
        sys.modules["librosa"] = lb


def import_reference(ref: Optional[str] = None):
    """(nvidia.ASRPipeline, streaming.streaming_pipeline, streaming.streams) of the reference; raises if it is not available."""
    ref = ref or reference_dir()
    if ref is None:
        raise RuntimeError("the reference package is not available on this box (no /root/reference, no oracle/_ref bundle)")
    if ref not in sys.path:
        sys.path.insert(0, ref)
    stub_audio_io()
    from thestage_speechkit.nvidia import ASRPipeline
    from thestage_speechkit.streaming import streaming_pipeline as sp
    from thestage_speechkit.streaming import streams

    return ASRPipeline, sp, streams


if __name__ == "__main__":
    print(make_bundle(), which())
