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

LANGS = [
    "en", "zh", "de", "es", "ru", "ko", "fr", "ja", "pt", "tr", "pl", "ca", "nl", "ar", "sv", "it", "id", "hi",
    "fi", "vi", "he", "uk", "el", "ms", "cs", "ro", "da", "hu", "ta", "no", "th", "ur", "hr", "bg", "lt", "la",
    "mi", "ml", "cy", "sk", "te", "fa", "lv", "bn", "sr", "az", "sl", "kn", "et", "mk", "br", "eu", "is", "hy",
    "ne", "mn", "bs", "kk", "sq", "sw", "gl", "mr", "pa", "si", "km", "sn", "yo", "so", "af", "oc", "ka", "be",
    "tg", "sd", "gu", "am", "yi", "lo", "uz", "fo", "ht", "ps", "tk", "nn", "mt", "sa", "lb", "my", "bo", "tl",
    "mg", "as", "tt", "haw", "ln", "ha", "ba", "jw", "su", "yue",
]


def default_alignment_heads(dims: WhisperDims):
    """Synthetic alignment heads (the upstream checkpoints' lists are not available offline):
    the upper half of the decoder layers, rotating heads - 10 pairs for 32-layer models like
    large-v3, fewer for small ones."""
    n = min(10, max(2, dims.dec_layers * 2))
    out = []
    for j in range(n):
        layer = dims.dec_layers - 1 - (j % max(1, dims.dec_layers // 2))
        head = (3 * j + 1) % dims.heads
        if [layer, head] not in out:
            out.append([layer, head])
    return out


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
    st = SpecialTokens()
    gc.lang_to_id = {f"<|{l}|>": st.lang_en + i for i, l in enumerate(LANGS)}
    gc.task_to_id = {"transcribe": st.transcribe, "translate": st.translate}
    gc.no_timestamps_token_id = st.no_timestamps
    gc.prev_sot_token_id = st.sot_prev
    gc.is_multilingual = True
    gc.alignment_heads = alignment_heads if alignment_heads is not None else default_alignment_heads(dims)
    gc.max_initial_timestamp_index = 50
    gc.suppress_tokens = []
    gc.begin_suppress_tokens = [220, st.eos]
    gc.max_length = dims.max_target_positions
    gc.forced_decoder_ids = None
    gc.bos_token_id = st.eos
    gc.eos_token_id = st.eos
    gc.pad_token_id = st.eos
    gc.decoder_start_token_id = st.sot
    gc.return_timestamps = False
    return gc


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


def _bytes_to_unicode():
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\u00a1"), ord("\u00ac") + 1)) + list(range(ord("\u00ae"), ord("\u00ff") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


def build_tokenizer(dims: WhisperDims):
    """In-memory synthetic ``WhisperTokenizer`` with the large-v3 special-token id layout
    (ctor HF:models/whisper/tokenization_whisper.py:206-276)."""
    from transformers import WhisperTokenizer

    st = SpecialTokens()
    vocab: Dict[str, int] = {}
    # byte-level alphabet first (the GPT-2 byte<->unicode table used by the ByteLevel pre-tokenizer/decoder)
    for ch in _bytes_to_unicode().values():
        vocab[ch] = len(vocab)
    i = 0
    while len(vocab) < st.eos:
        tok = f"Ġw{i}"
        if tok not in vocab:
            vocab[tok] = len(vocab)
        i += 1
    specials = ["<|endoftext|>", "<|startoftranscript|>"]
    specials += [f"<|{l}|>" for l in LANGS]
    specials += ["<|translate|>", "<|transcribe|>", "<|startoflm|>", "<|startofprev|>", "<|nospeech|>", "<|notimestamps|>"]
    n_ts = dims.vocab - (st.eos + len(specials))
    specials += [f"<|{k * 0.02:.2f}|>" for k in range(n_ts)]
    for s in specials:
        vocab[s] = len(vocab)
    assert len(vocab) == dims.vocab, (len(vocab), dims.vocab)
    assert vocab["<|notimestamps|>"] == st.no_timestamps
    tok = WhisperTokenizer(
        vocab=vocab,
        merges=[],
        language="en",
        task="transcribe",
        # timestamps are ordinary added tokens upstream: `timestamp_begin = all_special_ids[-1] + 1`
        additional_special_tokens=[t for t in specials[1:] if vocab[t] <= st.no_timestamps],
        pad_token="<|endoftext|>",
        bos_token="<|endoftext|>",
        eos_token="<|endoftext|>",
        unk_token="<|endoftext|>",
    )
    return tok


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
