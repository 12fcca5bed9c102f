"""thewhisper_amd - MI355X (gfx950) backend for TheStageAI/TheWhisper's Whisper hot path.

Drops into the reference as ``thestage_speechkit/amd`` (sibling of ``nvidia/`` and ``apple/``; see
INTEGRATION.md): ``ASRPipeline`` mirrors R:thestage_speechkit/nvidia/asr_pipeline.py and
``AMDWhisperBackend`` implements the ``TranscriptionBackend.transcribe`` contract of
R:thestage_speechkit/streaming/streaming_pipeline.py:51-64.  All arithmetic of the path (log-mel,
encoder, cached-KV decoder, logits processors, DTW) runs in hand-written HIP kernels behind the C
ABI of include/thewhisper.h; there is no CPU or eager-PyTorch fallback.
"""
from __future__ import annotations

__version__ = "0.1.0"

__all__ = ["ASRPipeline", "AMDWhisperBackend", "WhisperEngine", "AMDWhisperForConditionalGeneration", "BatchingHub",
           "EncoderOverlap", "create_gateway_app"]


def __getattr__(name):  # lazy: importing the package must not pull torch/transformers
    if name == "WhisperEngine":
        from .engine import WhisperEngine

        return WhisperEngine
    if name == "ASRPipeline":
        from .asr_pipeline import ASRPipeline

        return ASRPipeline
    if name == "AMDWhisperForConditionalGeneration":
        from .model import AMDWhisperForConditionalGeneration

        return AMDWhisperForConditionalGeneration
    if name == "AMDWhisperBackend":
        from .streaming import AMDWhisperBackend

        return AMDWhisperBackend
    if name == "BatchingHub":
        from .serving import BatchingHub

        return BatchingHub
    if name == "EncoderOverlap":
        from .overlap import EncoderOverlap

        return EncoderOverlap
    if name == "create_gateway_app":
        from .gateway import create_app

        return create_app
    raise AttributeError(name)
