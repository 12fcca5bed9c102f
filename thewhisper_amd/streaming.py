"""``AMDWhisperBackend`` - the ``platform="amd"`` transcription backend for the reference's StreamingPipeline.

Implements the ``TranscriptionBackend.transcribe(audio, buffer_start_time, sample_rate) -> [{"text","start","end"}]``
contract (R:thestage_speechkit/streaming/streaming_pipeline.py:51-64) exactly as ``LocalWhisperBackend`` does
for the other platforms (R:...:340-435): fixed greedy generate kwargs with ``max_new_tokens=128``, word
timestamps always on, the zlib gibberish filter (>2.2 -> ``[]``) and the open-ended last-word fix-up.
Use it through the reference's injection seam ``StreamingPipeline(backend=AMDWhisperBackend(...))`` or via the
three-line ``platform == "amd"`` patch shown in INTEGRATION.md.
"""
from __future__ import annotations

import zlib
from typing import Any, Dict, List, Optional

import numpy as np
import torch


def _compression_ratio(text: str) -> float:
    """R:thestage_speechkit/streaming/streaming_pipeline.py:41-43."""
    text_bytes = text.encode("utf-8")
    return len(text_bytes) / len(zlib.compress(text_bytes))


class AMDWhisperBackend:
    """Duck-typed ``TranscriptionBackend`` (the ABC lives in the reference package)."""

    def __init__(
        self,
        model,
        model_size: str = "S",
        chunk_length_s: int = 10,
        torch_dtype: "torch.dtype | None" = None,
        language: str = "en",
        feature_extractor=None,
        tokenizer=None,
        revision: str = "main",
        asr_pipeline=None,
        **pipeline_kwargs,
    ):
        from .asr_pipeline import ASRPipeline

        if torch_dtype is None:
            # the reference defaults to fp16 (R:...:369-370); the MI355X engine's production dtype is bf16
            torch_dtype = torch.bfloat16
        self.chunk_length_s: float = chunk_length_s
        self.sample_rate: int = 16000
        self.device: str = "cuda"  # ROCm exposes MI355X as "cuda", as on the nvidia platform (R:...:365)
        self.language: str = language
        self.asr_pipeline = asr_pipeline or ASRPipeline(
            model,
            model_size=model_size,
            chunk_length_s=chunk_length_s,
            torch_dtype=torch_dtype,
            device=pipeline_kwargs.pop("device", self.device),
            feature_extractor=feature_extractor,
            tokenizer=tokenizer,
            revision=revision,
            **pipeline_kwargs,
        )

    def _generate_kwargs(self) -> Dict[str, Any]:
        return {"use_cache": True, "num_beams": 1, "do_sample": False, "max_new_tokens": 128, "language": self.language}

    def transcribe(self, audio: np.ndarray, buffer_start_time: float, sample_rate: int) -> List[Dict[str, Any]]:
        result: Dict[str, Any] = self.asr_pipeline(
            audio,
            return_timestamps="word",
            generate_kwargs=self._generate_kwargs(),
            chunk_length_s=self.chunk_length_s,
        )
        return self._to_tokens(result, len(audio) / sample_rate, buffer_start_time)

    def transcribe_many(self, requests, batch_size: Optional[int] = None) -> List[List[Dict[str, Any]]]:
        """Several streams' rolling buffers in ONE pipeline call: [(audio, buffer_start_time, sample_rate), ...].
        HF collates the chunks of different buffers into batched engine calls; per-stream results are unchanged."""
        audios = [np.asarray(a) for a, _, _ in requests]
        kw = {} if batch_size is None else {"batch_size": int(batch_size)}
        results = self.asr_pipeline(
            audios, return_timestamps="word", generate_kwargs=self._generate_kwargs(), chunk_length_s=self.chunk_length_s, **kw
        )
        return [self._to_tokens(res, len(a) / sr, t0) for res, (a, t0, sr) in zip(results, requests)]

    def job_codec(self) -> "Optional[JobCodec]":
        """Per-request pre/post-processing around ``shortform.run_pass`` (what ``serving.BatchingHub`` schedules), or None
        when this backend's call cannot run the restated short-form loop (it then serves whole-call batches)."""
        try:
            return JobCodec(self)
        except _NotEligible:
            return None

    @staticmethod
    def _to_tokens(result: Dict[str, Any], audio_duration: float, buffer_start_time: float) -> List[Dict[str, Any]]:
        if _compression_ratio(result["text"]) > 2.2:
            return []
        generated_tokens: List[Dict[str, Any]] = []
        max_word_duration: float = 1.0
        for token in result["chunks"]:
            if token["timestamp"][1] is None:
                if audio_duration - token["timestamp"][0] < max_word_duration:
                    token["timestamp"] = (token["timestamp"][0], audio_duration)
                else:
                    token["timestamp"] = (token["timestamp"][0], token["timestamp"][0] + max_word_duration)
            generated_tokens.append(
                {
                    "text": token["text"],
                    "start": token["timestamp"][0] + buffer_start_time,
                    "end": token["timestamp"][1] + buffer_start_time,
                }
            )
        return generated_tokens


class _NotEligible(Exception):
    pass


class BufferJob:
    """One ``transcribe`` request on its way through the hub: its chunks' decoding states and what post-processing needs."""

    __slots__ = ("works", "meta", "audio_duration", "buffer_start_time", "future", "t_submit")

    def __init__(self, works, meta, audio_duration, buffer_start_time):
        self.works = works                        # [shortform.ChunkWork] - one per <= chunk_length_s piece of the buffer
        self.meta = meta                          # [(is_last, stride)] as HF's chunk iterator produced them
        self.audio_duration = audio_duration
        self.buffer_start_time = buffer_start_time
        self.future = None
        self.t_submit = 0.0

    @property
    def done(self) -> bool:
        return all(w.done for w in self.works)


class JobCodec:
    """Splits ``AMDWhisperBackend.transcribe`` into the three stages HF's pipeline runs per call - ``preprocess`` (chunking +
    log-mel), the model, ``postprocess`` (tokenizer state machine, word timestamps, LCS merge) - so that a scheduler can put
    the model stage of MANY requests into shared passes (``shortform.run_pass``).  Stages 1 and 3 are HF's own methods on
    the backend's pipeline object, called with exactly the parameters ``pipeline.__call__`` would derive; the model stage's
    output dictionaries have the shape HF's un-batching hands to ``postprocess`` for a batch of one."""

    def __init__(self, backend: AMDWhisperBackend):
        pipe = backend.asr_pipeline
        model = getattr(pipe, "model", None)
        if model is None or not hasattr(model, "last_plan") or not hasattr(pipe, "_sanitize_parameters"):
            raise _NotEligible()
        self.backend = backend
        self.pipe = pipe
        pre, _fwd, post = pipe._sanitize_parameters(return_timestamps="word", generate_kwargs=backend._generate_kwargs(),
                                                     chunk_length_s=backend.chunk_length_s)
        self.pre = {**pipe._preprocess_params, **pre}
        self.post = {**pipe._postprocess_params, **post}
        self.plan = None

    def learn(self) -> bool:
        """One short request through the ordinary pipeline call: HF's ``generate`` runs once with this backend's options and
        the model object records the short-form plan (model.py).  False if the call turned out not to be eligible."""
        model = self.pipe.model
        model.last_plan = None
        self.backend.transcribe(np.zeros(self.backend.sample_rate, dtype=np.float32), 0.0, self.backend.sample_rate)
        plan = model.last_plan
        if plan is None or not plan.return_token_timestamps or not plan.return_segments:
            return False
        self.plan = plan
        return True

    def open(self, audio: np.ndarray, buffer_start_time: float, sample_rate: int) -> BufferJob:
        """Stage 1 (HF:pipelines/automatic_speech_recognition.py:346-482 via ``pipe.preprocess``)."""
        from .shortform import ChunkWork

        audio = np.asarray(audio)
        works, meta = [], []
        for item in self.pipe.preprocess(audio, **self.pre):
            feats = item["input_features"]
            am = item.get("attention_mask")
            nf = int(am.sum()) if am is not None else None
            works.append(ChunkWork(feats[0], nf))
            meta.append((item["is_last"], item.get("stride")))
        if not works:
            raise ValueError("empty audio buffer")
        return BufferJob(works, meta, len(audio) / sample_rate, buffer_start_time)

    def close(self, job: BufferJob) -> List[Dict[str, Any]]:
        """Stage 3 (``pipe.postprocess`` + the reference backend's word fix-ups, R:...:412-433)."""
        from .shortform import work_tokens

        outs = []
        for w, (is_last, stride) in zip(job.works, job.meta):
            seq, _raw, seg = work_tokens(self.plan, w)
            o = {"is_last": is_last, "tokens": seq.unsqueeze(0), "token_timestamps": seg.unsqueeze(0)}
            if stride is not None:
                o["stride"] = stride
            outs.append(o)
        result = self.pipe.postprocess(outs, **self.post)
        return AMDWhisperBackend._to_tokens(result, job.audio_duration, job.buffer_start_time)
