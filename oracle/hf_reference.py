"""Harness around the *installed* Hugging Face Whisper (transformers 5.15.0) on CPU.

TEST INFRASTRUCTURE ONLY (see oracle/whisper_oracle.py header).  The reference repo contains no
model arithmetic for its NVIDIA path: ``thestage_speechkit.nvidia.ASRPipeline`` with
``model_size=None`` calls ``transformers.WhisperForConditionalGeneration`` directly
(R:thestage_speechkit/nvidia/asr_pipeline.py:57-60).  This module builds that HF model from the
oracle's seeded weights, plus the synthetic tokenizer / generation config of SURVEY.md section 8c,
so that (a) the numpy oracle can be pinned against it and (b) ``bench.py`` can time the
reference's PyTorch-CPU arithmetic beside the MI355X numbers.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import numpy as np

from .whisper_oracle import SpecialTokens, WhisperDims, interpolate_positions

# Checkpoint-free fixtures (synthetic tokenizer, generation config, alignment heads) are not arithmetic of the path; they
# live in one place, thewhisper_amd/synthetic.py, because benchmarks and examples of the product need them too.
from thewhisper_amd import synthetic as _syn

LANGS = _syn.LANGS


def default_alignment_heads(dims: WhisperDims):
    return _syn.default_alignment_heads(dims.dec_layers, dims.heads)


def build_hf_config(dims: WhisperDims):
    from transformers import WhisperConfig

    st = SpecialTokens()
    return WhisperConfig(
        vocab_size=dims.vocab,
        num_mel_bins=dims.n_mels,
        d_model=dims.d_model,
        encoder_layers=dims.enc_layers,
        decoder_layers=dims.dec_layers,
        encoder_attention_heads=dims.heads,
        decoder_attention_heads=dims.heads,
        encoder_ffn_dim=dims.ffn,
        decoder_ffn_dim=dims.ffn,
        max_source_positions=dims.max_source_positions,
        max_target_positions=dims.max_target_positions,
        bos_token_id=st.eos,
        eos_token_id=st.eos,
        pad_token_id=st.eos,
        decoder_start_token_id=st.sot,
        activation_function="gelu",
        dropout=0.0,
        attention_dropout=0.0,
        activation_dropout=0.0,
        use_cache=True,
    )


def fill_generation_config(gc, dims: WhisperDims, alignment_heads=None):
    """Hand-filled multilingual generation config."""
    return _syn.fill_generation_config(gc, dims.dec_layers, dims.heads, dims.max_target_positions, alignment_heads)


def build_hf_model(dims: WhisperDims, weights: Dict[str, np.ndarray], dtype=None, alignment_heads=None, model_cls=None):
    """HF ``WhisperForConditionalGeneration`` (or ``model_cls``) carrying the oracle weights."""
    import torch
    from transformers import WhisperForConditionalGeneration

    cls = model_cls or WhisperForConditionalGeneration
    cfg = build_hf_config(dims)
    try:
        from transformers.initialization import no_init_weights

        with no_init_weights():
            model = cls(cfg)
    except Exception:  # pragma: no cover
        model = cls(cfg)
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights.items()}
    sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    bad = [m for m in missing if "proj_out" not in m]
    if bad or unexpected:
        raise RuntimeError(f"state_dict mismatch: missing={bad} unexpected={unexpected}")
    model.tie_weights()
    model.eval()
    if dtype is not None:
        model = model.to(dtype)
    fill_generation_config(model.generation_config, dims, alignment_heads)
    return model


def build_tokenizer(dims: WhisperDims):
    """In-memory synthetic ``WhisperTokenizer`` with the large-v3 special-token id layout."""
    return _syn.build_tokenizer(dims.vocab)


def build_feature_extractor(dims: WhisperDims, chunk_length_s: int):
    from transformers import WhisperFeatureExtractor

    return WhisperFeatureExtractor(feature_size=dims.n_mels, chunk_length=chunk_length_s)


def patch_chunk_length(model, chunk_length_s: int):
    """A0 + version-drift fix D1: interpolate encoder positions to
    T = 50*chunk and make the 5.x ``embed_positions(arange(num_embeddings))`` lookup agree."""
    import torch

    if chunk_length_s >= 30:
        return model
    T = int(1500 * (chunk_length_s / 30))
    model.config.max_source_positions = T
    pos = model.model.encoder.embed_positions.weight.detach().float().cpu().numpy()
    new = torch.from_numpy(interpolate_positions(pos, T)).to(model.model.encoder.embed_positions.weight.dtype)
    model.model.encoder.embed_positions.weight.data = new
    model.model.encoder.embed_positions.num_embeddings = T
    return model
