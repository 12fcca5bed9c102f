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
        reuse_committed_prefix: bool = False,
        reuse_margin_s: float = 1.0,
        draft_previous_tick: Optional[bool] = None,
        **pipeline_kwargs,
    ):
        """``reuse_committed_prefix`` (never the default): the reference scheduler hands over, every 0.5 s,
        a rolling buffer most of which the previous call already transcribed (R:...streaming_pipeline.py:770-796 re-decodes all
        of it).  With the option on, a call whose buffer EXTENDS the previous call's (same ``buffer_start_time``, at least as
        long, one chunk) hands the previous call's tokens whose word timestamps end at least ``reuse_margin_s`` before the old
        buffer's end to the greedy loop as forced output - one batched prefill (tw_greedy_opts::n_forced) instead of one decode
        step each - and decodes only the tail.  The audio those tokens belong to is unchanged, but the encoder is not causal and
        the log-mel clamp is global, so the forced tokens need not be what a fresh decode would pick: an approximation, whose
        delta bench.py / tests measure.  ``reuse_stats`` counts calls, reused calls and forced tokens.
        The option needs ONE BACKEND PER STREAM (the state of the previous call lives in the backend): besides the start time and the
        length, a call must begin with exactly the samples of its predecessor's buffer to reuse anything - a backend shared between
        sessions (gateway._LockedBackend, two schedulers on one pipeline) therefore never forces tokens of other audio, it just
        decodes afresh - and ``reset()`` (what a scheduler's ``clear()`` should call) forgets the previous call.

        ``draft_previous_tick`` (round 6): the EXACT form of the same idea.  The previous call's tokens - all of them, no margin -
        are handed to the greedy loop as a DRAFT (tw_greedy_opts::n_draft): the engine runs prompt + draft through the decoder in
        batched launches WITH the logits and Whisper's logits processors of every position, keeps the draft up to the first
        position where its own arg-max differs, takes that arg-max, offers the rest of the draft again behind it and decodes
        step by step from where nothing more is confirmed.  The call returns what the plain backend returns - same ids, same
        word timestamps - whatever the draft was (a draft of other audio only costs time), so the eligibility test is a matter
        of speed, not of correctness.  ``reuse_stats`` counts drafted and confirmed tokens.  Not together with
        ``reuse_committed_prefix``.  Because the result is the plain call's, this is ON by default (``None`` = on unless
        ``reuse_committed_prefix`` was asked for); ``False`` gives the reference's behaviour literally - every tick decoded from
        scratch (R:thestage_speechkit/streaming/streaming_pipeline.py:388-435)."""
        from .asr_pipeline import ASRPipeline

        if torch_dtype is None:
            # as the reference (R:...:369-370): float16 - a float16 context since round 4 (model.py: build_engine); bf16 before
            torch_dtype = torch.float16
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
        if draft_previous_tick is None:
            draft_previous_tick = not reuse_committed_prefix
        if reuse_committed_prefix and draft_previous_tick:
            raise ValueError("reuse_committed_prefix (approximate) and draft_previous_tick (exact) are alternatives")
        self.reuse_committed_prefix = bool(reuse_committed_prefix)
        self.draft_previous_tick = bool(draft_previous_tick)
        self.reuse_margin_s = float(reuse_margin_s)
        self.reuse_stats = {"calls": 0, "reused": 0, "forced_tokens": 0, "decoded_tokens": 0, "draft_tokens": 0, "confirmed_tokens": 0,
                            "verify_launches": 0}
        self._reuse_codec = None      # JobCodec (learned plan), built on the first reuse-enabled call
        self._last = None             # what the previous call left: start time, samples, first-iteration tokens + timestamps

    def _generate_kwargs(self) -> Dict[str, Any]:
        return {"use_cache": True, "num_beams": 1, "do_sample": False, "max_new_tokens": 128, "language": self.language}

    def transcribe(self, audio: np.ndarray, buffer_start_time: float, sample_rate: int) -> List[Dict[str, Any]]:
        if self.reuse_committed_prefix or self.draft_previous_tick:
            words = self._transcribe_with_reuse(np.asarray(audio), float(buffer_start_time), int(sample_rate))
            if words is not None:
                return words
        result: Dict[str, Any] = self.asr_pipeline(
            audio,
            return_timestamps="word",
            generate_kwargs=self._generate_kwargs(),
            chunk_length_s=self.chunk_length_s,
        )
        return self._to_tokens(result, len(audio) / sample_rate, buffer_start_time)

    # -- decoder-side reuse between the ticks of one stream (opt-in) -------------------------------
    def reset(self) -> None:
        """Forget the previous call (``reuse_committed_prefix``): the next buffer is decoded afresh.  For the owner of the stream to
        call when its scheduler restarts at t = 0 (R:thestage_speechkit/streaming/streaming_pipeline.py:953-976 ``clear``)."""
        self._last = None

    def _forced_prefix(self, audio: np.ndarray, buffer_start_time: float, sample_rate: int) -> Optional[np.ndarray]:
        """Tokens of the previous call that may be forced in this one, or None.  Eligible: same buffer start, buffer not shorter
        (the scheduler appended audio; a trimmed buffer starts elsewhere and is decoded afresh).  Kept: the longest prefix whose
        token timestamps (DTW, seconds from the buffer start) end ``reuse_margin_s`` before the OLD buffer's end - the audio
        behind them has had its right context for at least that long."""
        last = self._last
        n_samples = len(audio)
        if last is None or abs(last["start"] - buffer_start_time) > 1e-6 or n_samples < last["n_samples"] or last["sr"] != sample_rate:
            return None
        # ... and it must BE the previous buffer plus new audio: same start and length alone also match another session's buffer or
        # a restarted stream (0.64 MB compared per 10 s: ~0.1 ms)
        prev = last["audio"]
        if audio.dtype != prev.dtype or not np.array_equal(audio[: len(prev)], prev):
            return None
        ids, ts = last["ids"], last["ts"]
        if ids is None or len(ids) == 0:
            return None
        if self.draft_previous_tick:      # exact whatever is offered: everything the previous call produced
            return np.asarray(ids, dtype=np.int32)
        if ts is None:
            return None
        limit = last["n_samples"] / sample_rate - self.reuse_margin_s
        keep = 0
        for i in range(len(ids)):
            if ts[i] > limit:
                break
            keep = i + 1
        if keep < 2:
            return None
        return np.asarray(ids[:keep], dtype=np.int32)

    def _transcribe_with_reuse(self, audio: np.ndarray, buffer_start_time: float, sample_rate: int):
        """The call through the restated short-form loop (shortform.py) with a forced prefix on the first seek iteration; None
        when this backend cannot (no learned plan, a buffer longer than one chunk): the ordinary call then runs."""
        from . import shortform

        if self._reuse_codec is None:
            codec = self.job_codec()
            if codec is None or not codec.learn():        # (the plan is learned from one ORDINARY call of this backend: JobCodec.learn)
                self.reuse_committed_prefix = self.draft_previous_tick = False
                return None                               # not eligible on this pipeline: stays the plain backend
            self._reuse_codec = codec
        codec = self._reuse_codec
        self.reuse_stats["calls"] += 1
        job = codec.open(audio, buffer_start_time, sample_rate)
        eng = self.asr_pipeline.model.engine
        if len(job.works) != 1:
            self._last = None
            forced = None
        else:
            forced = self._forced_prefix(audio, buffer_start_time, sample_rate)
            budget = int(codec.plan.greedy.get("max_new_tokens", 128))
            if forced is not None and len(forced) >= budget - 1:
                forced = forced[: max(0, budget - 2)]
            if forced is not None and self.draft_previous_tick:
                if len(forced) >= 1:
                    job.works[0].draft = forced
                    self.reuse_stats["reused"] += 1
                    self.reuse_stats["draft_tokens"] += int(len(forced))
                forced = None
            elif forced is not None and len(forced) >= 2:
                job.works[0].forced = forced
                self.reuse_stats["reused"] += 1
                self.reuse_stats["forced_tokens"] += int(len(forced))
        while not job.done:
            for w in [w for w in job.works if not w.done]:
                shortform.run_pass(eng, codec.plan, [w])
                if w.passes > shortform.MAX_SEEK_PASSES:
                    raise RuntimeError(f"a chunk needed more than {shortform.MAX_SEEK_PASSES} seek passes")
        if len(job.works) == 1 and job.works[0].first_pass is not None:
            ids, ts = job.works[0].first_pass
            dr = job.works[0].draft_result
            if dr is not None:
                self.reuse_stats["confirmed_tokens"] += int(dr["accepted"])
                self.reuse_stats["verify_launches"] += int(dr["launches"])
            self.reuse_stats["decoded_tokens"] += int(len(ids)) - (int(len(forced)) if forced is not None else 0) - (int(dr["accepted"]) if dr else 0)
            self._last = {"start": buffer_start_time, "n_samples": len(audio), "sr": sample_rate, "ids": ids, "ts": ts,
                          "audio": np.array(audio, copy=True)}
        return codec.close(job)

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
        b = self.backend
        mode = (b.reuse_committed_prefix, b.draft_previous_tick)     # (the ORDINARY call: the reuse / draft paths bypass HF's generate)
        b.reuse_committed_prefix = b.draft_previous_tick = False
        try:
            b.transcribe(np.zeros(b.sample_rate, dtype=np.float32), 0.0, b.sample_rate)
        finally:
            b.reuse_committed_prefix, b.draft_previous_tick = mode
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
