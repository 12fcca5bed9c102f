"""``thestage_speechkit.amd`` - the MI355X platform package, sibling of ``thestage_speechkit.nvidia`` and ``.apple``.

This is the file a maintainer drops into the reference tree (integration/apply.py does it for a checkout);
``LocalWhisperBackend(platform="amd")`` imports ``ASRPipeline`` from here exactly as it imports
``..nvidia.ASRPipeline`` / ``..apple.ASRPipeline`` (R:thestage_speechkit/streaming/streaming_pipeline.py:358-367).
"""
from thewhisper_amd import ASRPipeline  # noqa: F401  drop-in sibling of nvidia.ASRPipeline / apple.ASRPipeline

__all__ = ["ASRPipeline"]
