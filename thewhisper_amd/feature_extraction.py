"""``AMDWhisperFeatureExtractor``: HF ``WhisperFeatureExtractor`` whose log-mel runs in the HIP kernel (A1).

HF's pipeline calls ``feature_extractor(chunk, sampling_rate=..., return_tensors="pt", return_attention_mask=True)``
per chunk (HF:pipelines/automatic_speech_recognition.py:67-72); everything except the arithmetic
(padding to n_samples, attention-mask rescaling, BatchFeature packing) is inherited, and
``_torch_extract_fbank_features`` (HF:models/whisper/feature_extraction_whisper.py:135-168) is replaced by
``tw_logmel``.  No CPU fallback: without an attached engine the call raises.
"""
from __future__ import annotations

import threading

import numpy as np
from transformers import WhisperFeatureExtractor


class AMDWhisperFeatureExtractor(WhisperFeatureExtractor):
    _engine = None
    _tls = threading.local()

    @classmethod
    def from_hf(cls, fe: WhisperFeatureExtractor) -> "AMDWhisperFeatureExtractor":
        if isinstance(fe, cls):
            return fe
        if not isinstance(fe, WhisperFeatureExtractor):
            raise TypeError("expected a WhisperFeatureExtractor")
        if fe.n_fft != 400 or fe.hop_length != 160 or fe.sampling_rate != 16000 or getattr(fe, "dither", 0.0) != 0.0:
            raise NotImplementedError("the MI355X log-mel kernel implements n_fft=400, hop=160, 16 kHz, no dither")
        new = cls(feature_size=fe.feature_size, sampling_rate=fe.sampling_rate, hop_length=fe.hop_length,
                  chunk_length=fe.chunk_length, n_fft=fe.n_fft, padding_value=fe.padding_value,
                  return_attention_mask=fe.return_attention_mask)
        return new

    def attach_engine(self, engine) -> None:
        if engine.n_mels != self.feature_size:
            raise ValueError("engine / feature extractor mel-bin mismatch")
        self._engine = engine

    def __call__(self, raw_speech, *args, return_tensors=None, **kwargs):
        # With return_tensors="pt" (what the ASR pipeline asks for, HF:pipelines/automatic_speech_recognition.py:67-72) the
        # log-mel tensor stays where tw_logmel wrote it - in HBM - and travels through BatchFeature / the pipeline's collate
        # function as a device tensor straight into tw_encode: no device -> host -> device round trip per chunk.
        self._tls.keep_on_device = return_tensors == "pt"
        try:
            return super().__call__(raw_speech, *args, return_tensors=return_tensors, **kwargs)
        finally:
            self._tls.keep_on_device = False

    def _torch_extract_fbank_features(self, waveform: np.ndarray, device: str = "cpu"):  # noqa: ARG002
        # a thread may direct ITS calls to another context of the same model (serving.py: the prefetch thread computes the
        # log-mel of new arrivals on the sibling context / side stream while the main context is inside its decode loop)
        engine = getattr(self._tls, "engine", None) or self._engine
        if engine is None:
            raise RuntimeError("AMDWhisperFeatureExtractor has no engine attached (no CPU fallback)")
        import torch

        w = np.asarray(waveform, dtype=np.float32)
        squeeze = w.ndim == 1
        if squeeze:
            w = w[None]
        outs = []
        mb = engine.max_batch
        for i in range(0, w.shape[0], mb):
            x = torch.from_numpy(np.ascontiguousarray(w[i : i + mb]))
            outs.append(engine.logmel(x, out_dtype=torch.float32))
        mel = outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)
        if getattr(self._tls, "keep_on_device", False):
            return mel[0] if squeeze else mel
        out = mel.cpu().numpy()
        return out[0] if squeeze else out

    # the numpy path must not silently take over either
    def _np_extract_fbank_features(self, waveform_batch, device):  # noqa: ARG002
        return self._torch_extract_fbank_features(np.asarray(waveform_batch), device)
