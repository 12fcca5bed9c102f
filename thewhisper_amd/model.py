"""HF-shaped model object whose arithmetic runs in libthewhisper_gfx950.so.

The reference slots a foreign engine under Hugging Face ``generate()`` in two ways: a whole-class
swap (``elastic_models...WhisperForConditionalGeneration``, R:thestage_speechkit/nvidia/asr_pipeline.py:48-56)
and an encoder/decoder module swap (R:thestage_speechkit/apple/model.py:601-614).  This backend keeps
Whisper's *control flow* from HF (``WhisperGenerationMixin.generate``: init tokens, seek loop, segment
slicing - per chunk, host side) and replaces everything underneath it:

* ``_EngineGreedyMixin.generate`` is inserted in the MRO between ``WhisperGenerationMixin`` and
  ``GenerationMixin``; it is what ``generate_with_fallback`` reaches through ``super().generate``
  (HF:models/whisper/generation_whisper.py:1027) and runs encoder + cross-K/V + the whole greedy
  loop with the Whisper logits processors on the GPU (A2-A10).
* ``_extract_token_timestamps`` is overridden with the on-device median-filter + DTW (A11).

Unsupported generation options (beam search, sampling, custom processors ...) raise
``NotImplementedError``: there is no eager/CPU fallback.
"""
from __future__ import annotations

import logging
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from transformers import WhisperConfig, WhisperForConditionalGeneration
from transformers.generation.logits_process import (
    SuppressTokensAtBeginLogitsProcessor,
    SuppressTokensLogitsProcessor,
    WhisperTimeStampLogitsProcessor,
)
from transformers.generation.utils import GenerateEncoderDecoderOutput, GenerationMixin
from transformers.modeling_outputs import Seq2SeqLMOutput

logger = logging.getLogger(__name__)


def dims_from_config(cfg: WhisperConfig) -> Dict[str, int]:
    if cfg.encoder_attention_heads != cfg.decoder_attention_heads or cfg.encoder_ffn_dim != cfg.decoder_ffn_dim:
        raise NotImplementedError("encoder/decoder head or FFN sizes differ")
    return dict(
        d_model=cfg.d_model, enc_layers=cfg.encoder_layers, dec_layers=cfg.decoder_layers,
        heads=cfg.encoder_attention_heads, ffn=cfg.encoder_ffn_dim, vocab=cfg.vocab_size, n_mels=cfg.num_mel_bins,
        max_source_positions=1500, max_target_positions=cfg.max_target_positions,
    )


def _default_engine_factory(dims, T, max_batch, dtype, alignment_heads, device_index):
    from .engine import WhisperEngine

    return WhisperEngine(dims, T, max_batch=max_batch, dtype=dtype, alignment_heads=alignment_heads, device=device_index)


class _EngineGreedyMixin(GenerationMixin):
    """Replaces ``GenerationMixin.generate`` (-> ``_sample``, HF:generation/utils.py:2783-2946) for the one
    configuration Whisper transcription uses: greedy, num_beams=1, short-form segment in, token ids out."""

    def generate(  # type: ignore[override]
        self,
        inputs: Optional[torch.Tensor] = None,
        generation_config=None,
        logits_processor=None,
        stopping_criteria=None,
        prefix_allowed_tokens_fn=None,
        synced_gpus=None,
        decoder_input_ids: Optional[torch.Tensor] = None,
        attention_mask: Optional[torch.Tensor] = None,
        **kwargs,
    ):
        eng = self._require_engine()
        gc = generation_config if generation_config is not None else self.generation_config
        self._check_supported(gc, stopping_criteria, prefix_allowed_tokens_fn, kwargs)
        if inputs is None:
            inputs = kwargs.pop("input_features", None)
        if inputs is None or decoder_input_ids is None:
            raise NotImplementedError("the MI355X engine needs `input_features` and `decoder_input_ids`")
        B = int(inputs.shape[0])
        if B > eng.max_batch:
            raise ValueError(f"batch of {B} chunks exceeds the engine capacity max_batch={eng.max_batch}; "
                             "construct ASRPipeline with a matching batch_size")
        n_prompt = int(decoder_input_ids.shape[1])
        opts = self._parse_logits_processors(logits_processor, n_prompt)
        max_target = int(self.config.max_target_positions)
        if getattr(gc, "max_new_tokens", None) is not None:
            max_len = n_prompt + int(gc.max_new_tokens)
        else:
            max_len = int(gc.max_length)
        max_len = min(max_len, max_target)
        min_new = int(getattr(gc, "min_new_tokens", None) or 0)
        if getattr(gc, "min_length", 0) and int(gc.min_length) > n_prompt:
            min_new = max(min_new, int(gc.min_length) - n_prompt)
        eos = gc.eos_token_id
        if isinstance(eos, (list, tuple)):
            if len(eos) != 1:
                raise NotImplementedError("multiple eos_token_id values")
            eos = eos[0]
        pad = gc.pad_token_id if gc.pad_token_id is not None else eos

        eng.encode(inputs)
        eng.cross_kv(B)
        want_align = bool(getattr(gc, "return_token_timestamps", False))
        prompt = decoder_input_ids.detach().to("cpu", torch.int32).numpy()
        greedy_kw = dict(max_new_tokens=max_len - n_prompt, min_new_tokens=min_new, max_length=max_target, eos_id=int(eos),
                         pad_id=int(pad), want_alignment=want_align, **opts)
        probe = getattr(self, "_plan_probe", None)
        if probe is not None:   # a call that is learning its short-form plan (shortform.py): what was the engine asked to do?
            probe.append({"prompt": prompt.copy(), "greedy": dict(greedy_kw),
                          "dict": bool(getattr(gc, "return_dict_in_generate", False))})
        out = eng.generate_greedy(prompt, **greedy_kw)
        seq = torch.from_numpy(out["sequences"]).to(decoder_input_ids.device, torch.long)
        self._last_greedy = {"B": B, "n_prompt": n_prompt, "len": int(seq.shape[1])}
        if getattr(gc, "return_dict_in_generate", False):
            return GenerateEncoderDecoderOutput(sequences=seq)
        return seq

    # -- helpers ---------------------------------------------------------------------------------
    @staticmethod
    def _check_supported(gc, stopping_criteria, prefix_allowed_tokens_fn, kwargs):
        bad = []
        if getattr(gc, "num_beams", 1) not in (None, 1) or kwargs.get("num_beams", 1) not in (None, 1):
            bad.append("num_beams > 1")
        if getattr(gc, "do_sample", False):
            bad.append("do_sample")
        if stopping_criteria:
            bad.append("custom stopping_criteria")
        if prefix_allowed_tokens_fn is not None:
            bad.append("prefix_allowed_tokens_fn")
        for k in ("assistant_model", "encoder_outputs", "decoder_attention_mask", "streamer", "past_key_values"):
            if kwargs.get(k) is not None:
                bad.append(k)
        if getattr(gc, "repetition_penalty", None) not in (None, 1.0):
            bad.append("repetition_penalty")
        if getattr(gc, "no_repeat_ngram_size", None) not in (None, 0):
            bad.append("no_repeat_ngram_size")
        if getattr(gc, "num_return_sequences", 1) not in (None, 1):
            bad.append("num_return_sequences > 1")
        if bad:
            raise NotImplementedError("not supported by the MI355X greedy engine: " + ", ".join(bad))

    @staticmethod
    def _parse_logits_processors(processors, n_prompt: int) -> Dict[str, Any]:
        opts: Dict[str, Any] = dict(timestamps=False, begin_suppress=(), suppress=(), no_timestamps_id=0,
                                    max_initial_timestamp_index=None)
        for p in processors or []:
            if isinstance(p, SuppressTokensAtBeginLogitsProcessor):
                if p.begin_index != n_prompt:
                    raise NotImplementedError("begin_index differs from the prompt length")
                opts["begin_suppress"] = tuple(int(x) for x in p.begin_suppress_tokens.tolist())
            elif isinstance(p, SuppressTokensLogitsProcessor):
                opts["suppress"] = tuple(int(x) for x in p.suppress_tokens.tolist())
            elif isinstance(p, WhisperTimeStampLogitsProcessor):
                if p.begin_index != n_prompt or not p._detect_timestamp_from_logprob:
                    raise NotImplementedError("unsupported WhisperTimeStampLogitsProcessor configuration")
                opts["timestamps"] = True
                opts["no_timestamps_id"] = int(p.no_timestamps_token_id)
                opts["max_initial_timestamp_index"] = p.max_initial_timestamp_index
            else:
                raise NotImplementedError(f"logits processor {type(p).__name__} is not implemented on the MI355X engine")
        return opts


class AMDWhisperForConditionalGeneration(WhisperForConditionalGeneration, _EngineGreedyMixin):
    """``WhisperForConditionalGeneration`` with the hot path on the MI355X engine.

    MRO: AMDWhisper -> WhisperForConditionalGeneration -> WhisperGenerationMixin -> _EngineGreedyMixin ->
    GenerationMixin -> ...; the HF torch modules are kept only as the container the checkpoint loads
    into - after ``build_engine`` their storage is released.
    """

    _engine = None
    _engine_factory: Callable = staticmethod(_default_engine_factory)

    # -- engine lifetime ---------------------------------------------------------------------------
    def build_engine(self, chunk_length_s: int = 30, max_batch: int = 1, dtype: Optional[torch.dtype] = None,
                     engine_factory: Optional[Callable] = None, release_torch_weights: bool = True,
                     decoder_weights: Optional[str] = None):
        """Create the context for T = 50*chunk_length_s encoder frames and upload the weights (A0 is applied by
        the library: tw_finalize_weights interpolates the positional table like patch_hf_model)."""
        if chunk_length_s > 30 or chunk_length_s < 1:
            raise ValueError("chunk_length_s must be in [1, 30]")
        T = int(1500 * (chunk_length_s / 30))  # same expression as R:thestage_speechkit/nvidia/asr_pipeline.py:16
        p0 = next(self.parameters())
        dt = dtype or p0.dtype
        # float16 requests run in a float16 context since round 4 (the reference's streaming default,
        # R:thestage_speechkit/streaming/streaming_pipeline.py:369-370); THEWHISPER_FP16_AS_BF16=1 restores the bf16 mapping
        import os

        eng_dtype = {torch.float32: "f32", torch.float16: "f16"}.get(dt, "bf16")
        if eng_dtype == "f16" and os.environ.get("THEWHISPER_FP16_AS_BF16", "0") == "1":
            logger.warning("fp16 requested: THEWHISPER_FP16_AS_BF16=1, the MI355X engine computes in bf16")
            eng_dtype = "bf16"
        if decoder_weights == "fp8":
            if eng_dtype == "f32":
                raise ValueError("decoder_weights='fp8' needs a bf16 (or fp16) torch_dtype")
            eng_dtype = "fp8"
        heads = getattr(self.generation_config, "alignment_heads", None) or []
        factory = engine_factory or type(self)._engine_factory
        dev_index = p0.device.index if p0.device.type == "cuda" and p0.device.index is not None else 0
        eng = factory(dims_from_config(self.config), T, int(max_batch), eng_dtype, [tuple(h) for h in heads], dev_index)
        sd = {k: v for k, v in self.state_dict().items() if k != "proj_out.weight"}
        eng.load_state_dict(sd)
        self._engine = eng
        self._engine_T = T
        # mirror what patch_hf_model leaves behind, so HF's short-form test `frames <= 2*max_source_positions` holds
        self.config.max_source_positions = T
        if release_torch_weights:
            for p in self.parameters():
                p.data = torch.empty(0, dtype=p.dtype, device=p.device)
        return eng

    def attach_engine(self, engine):
        """Run on an engine whose weights were loaded elsewhere (weights generated on the device, or one context shared by
        several pipelines); the torch modules of this object are then only the container HF's generate() needs."""
        self._engine = engine
        self._engine_T = int(engine.T)
        self.config.max_source_positions = int(engine.T)
        return engine

    def _require_engine(self):
        if self._engine is None:
            raise RuntimeError("AMDWhisperForConditionalGeneration has no engine: call build_engine() "
                               "(thewhisper_amd.ASRPipeline does it); there is no eager fallback")
        return self._engine

    @property
    def engine(self):
        return self._require_engine()

    # -- Whisper control flow: short-form fast path ---------------------------------------------------
    #: False = every call goes through HF's WhisperGenerationMixin.generate (A/B switch; THEWHISPER_FAST_GENERATE=0 too)
    fast_generate: bool = True
    _plans: Optional[Dict[Any, Any]] = None
    _plan_probe: Optional[list] = None
    last_plan = None   # the ShortFormPlan of the most recent eligible call (serving.py picks it up after its warm-up call)

    _FAST_KW = {"input_features", "attention_mask", "generation_config", "return_timestamps", "return_token_timestamps",
                "return_segments", "language", "task", "is_multilingual", "use_cache", "num_beams", "do_sample",
                "max_new_tokens", "min_new_tokens", "max_length", "temperature", "return_dict_in_generate"}
    _GC_FIELDS = ("max_new_tokens", "max_length", "min_new_tokens", "min_length", "eos_token_id", "pad_token_id", "suppress_tokens",
                  "begin_suppress_tokens", "no_timestamps_token_id", "max_initial_timestamp_index", "return_timestamps", "task",
                  "language", "num_beams", "do_sample", "temperature", "no_speech_threshold", "logprob_threshold",
                  "compression_ratio_threshold", "condition_on_prev_tokens", "prompt_condition_type", "forced_decoder_ids",
                  "is_multilingual", "return_dict_in_generate", "force_unique_generate_call", "num_return_sequences",
                  "repetition_penalty", "no_repeat_ngram_size", "decoder_start_token_id", "prev_sot_token_id")

    def _plan_key(self, kwargs) -> Optional[Tuple]:
        """Hashable fingerprint of everything that shapes a short-form call, or None if the call is not eligible for the
        restated control flow (it then runs HF's own ``generate``)."""
        import os

        if not self.fast_generate or os.environ.get("THEWHISPER_FAST_GENERATE", "1") == "0" or self._engine is None:
            return None
        if any(k not in self._FAST_KW for k in kwargs):
            return None           # prompt_ids, thresholds, logits_processor, stopping criteria, assistant model, ...
        feats = kwargs.get("input_features")
        if not isinstance(feats, torch.Tensor) or feats.dim() != 3 or int(feats.shape[-1]) != 2 * int(self._engine_T):
            return None           # long-form (or malformed) input: HF's loop (and its error messages)
        if int(feats.shape[0]) > self._engine.max_batch or int(feats.shape[0]) < 1:
            return None
        gc = kwargs.get("generation_config") or self.generation_config
        lang = kwargs.get("language", getattr(gc, "language", None))
        if lang is None and getattr(gc, "is_multilingual", False):
            return None           # language detection is a forward pass whose result selects the prompt per row
        if kwargs.get("temperature") not in (None, 0, 0.0) or kwargs.get("return_dict_in_generate"):
            return None
        if kwargs.get("return_token_timestamps") and kwargs.get("attention_mask") is None:
            return None
        def frz(v):
            if isinstance(v, (list, tuple)):
                return tuple(frz(x) for x in v)
            if isinstance(v, dict):
                return tuple(sorted((str(k), frz(x)) for k, x in v.items()))
            if isinstance(v, torch.Tensor):
                return tuple(v.flatten().tolist())
            return v if isinstance(v, (int, float, str, bool, type(None))) else repr(v)
        kw = tuple(sorted((k, frz(v)) for k, v in kwargs.items() if k not in ("input_features", "attention_mask", "generation_config")))
        gcf = tuple(frz(getattr(gc, f, None)) for f in self._GC_FIELDS)
        ah = getattr(gc, "alignment_heads", None)
        return (kw, gcf, frz(ah), kwargs.get("attention_mask") is None)

    def generate(self, *args, **kwargs):  # type: ignore[override]
        """``WhisperGenerationMixin.generate`` with a fast path: once a set of call options has been seen, eligible short-form
        batches run the restated seek loop of ``thewhisper_amd.shortform`` (identical results, see that module); the first
        call of each kind - and every call that is not eligible - runs HF's code unchanged."""
        from . import shortform

        key = None if args else self._plan_key(kwargs)
        if key is None:
            return super().generate(*args, **kwargs)
        if self._plans is None:
            self._plans = {}
        plan = self._plans.get(key, False)
        if plan is None:          # learned before: HF's flow did something the plan cannot express
            return super().generate(**kwargs)
        if plan is False:         # first call of this kind: run HF's flow and learn from what it asked the engine to do
            self._plan_probe = []
            try:
                out = super().generate(**kwargs)
                recs = self._plan_probe
            finally:
                self._plan_probe = None
            self._plans[key] = self._learn_plan(kwargs, recs)
            self.last_plan = self._plans[key]
            return out
        self.last_plan = plan
        return shortform.generate_shortform(self._require_engine(), plan, kwargs["input_features"], kwargs.get("attention_mask"))

    def _learn_plan(self, kwargs, recs):
        from . import shortform

        if not recs:
            return None
        g0, p0 = recs[0]["greedy"], recs[0]["prompt"][0]
        for r in recs:   # every inner call of the seek loop must have been the same request, every row the same prompt
            if r["greedy"] != g0 or r["dict"] != recs[0]["dict"] or r["prompt"].shape[1] != len(p0) or (r["prompt"] != p0[None]).any():
                return None
        gc = kwargs.get("generation_config") or self.generation_config
        nts = getattr(gc, "no_timestamps_token_id", None)
        rt = kwargs.get("return_timestamps")
        if rt is None:
            rt = getattr(gc, "return_timestamps", False)
        return shortform.ShortFormPlan(
            init_tokens=tuple(int(x) for x in p0), greedy=dict(g0), eos=int(g0["eos_id"]), pad=int(g0["pad_id"]),
            timestamp_begin=(int(nts) + 1) if nts is not None else int(self.config.vocab_size) + 1,   # HF:...:1409-1415
            return_timestamps=bool(rt), return_token_timestamps=bool(kwargs.get("return_token_timestamps")),
            return_segments=bool(kwargs.get("return_segments", False)), result_is_dict=bool(recs[0]["dict"]))

    # -- teacher-forced forward (language detection, parity tests) ---------------------------------
    def forward(self, input_features=None, decoder_input_ids=None, encoder_outputs=None, **kwargs):  # type: ignore[override]
        eng = self._require_engine()
        if decoder_input_ids is None:
            raise NotImplementedError("forward() needs decoder_input_ids")
        if input_features is not None:
            B = int(input_features.shape[0])
            eng.encode(input_features)
            eng.cross_kv(B)
        else:
            # the encoder states live inside the engine (A2-A5): a tensor of `encoder_outputs` cannot be handed to it, and decoding
            # against whatever the engine encoded LAST would silently answer for other audio.  HF's own callers of this form
            # (detect_language, HF:models/whisper/generation_whisper.py:1610-1683) pass input_features whenever they have them.
            raise NotImplementedError("forward() needs input_features: the MI355X engine keeps the encoder states internally and "
                                      "does not accept `encoder_outputs`")
        B = int(decoder_input_ids.shape[0])
        eng.decoder_reset(B)
        ids = decoder_input_ids.detach().to("cpu").numpy()
        cols = []
        for j in range(ids.shape[1]):
            cols.append(eng.decode_step(ids[:, j].tolist()))
        logits = torch.stack(cols, dim=1)
        return Seq2SeqLMOutput(logits=logits)

    # -- A11 -------------------------------------------------------------------------------------
    def _extract_token_timestamps(self, generate_outputs, alignment_heads, time_precision=0.02, num_frames=None,
                                  num_input_ids=None):
        """On-device replacement of HF:models/whisper/generation_whisper.py:241-381 (5.15.0 semantics)."""
        eng = self._require_engine()
        st = self._last_greedy
        seq = generate_outputs["sequences"]
        B, L = int(seq.shape[0]), int(seq.shape[1])
        if st["B"] != B or st["len"] != L:
            raise RuntimeError("token timestamps requested for a generate() call the engine no longer holds")
        n_in = int(num_input_ids if num_input_ids is not None else st["n_prompt"])
        # which columns HF keeps depends on the type and uniformity of `num_frames` (shortform.hf_kept_columns); the engine crops once per row
        from .shortform import columns_as_num_frames, hf_kept_columns

        nf = None if num_frames is None else columns_as_num_frames(hf_kept_columns(num_frames, B, int(eng.T)))
        if L - 1 <= n_in:  # one generated token: no cross-attention rows after the prompt -> zeros (HF :341-344)
            return torch.zeros((B, L), dtype=torch.float32, device=seq.device)
        ts = eng.token_timestamps(B, n_in, L, nf, time_precision)
        return torch.from_numpy(ts).to(seq.device)

    # -- construction helpers --------------------------------------------------------------------
    @classmethod
    def from_hf(cls, hf_model: WhisperForConditionalGeneration) -> "AMDWhisperForConditionalGeneration":
        """Re-class an already loaded HF model (weights shared, no copy)."""
        if isinstance(hf_model, cls):
            return hf_model
        if not isinstance(hf_model, WhisperForConditionalGeneration):
            raise TypeError("expected a transformers WhisperForConditionalGeneration")
        hf_model.__class__ = cls
        hf_model._engine = None
        return hf_model
