"""Checkpoint-free fixtures: a synthetic Whisper tokenizer / generation config / random weights of a named architecture.

There is no network on the build and benchmark boxes (no checkpoints, no tokenizer files), so benchmarks, examples and
tests fabricate a model of the right SHAPE: the large-v3 special-token id layout (the reference
hard-codes 50364 = <|notimestamps|> itself, R:thestage_speechkit/apple/model.py:333), a byte-level vocabulary padded
with filler words, a hand-filled multilingual generation config, and uniform random weights.  Nothing here is arithmetic
of the hot path; with real checkpoints none of it is used.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

LANGS = [
    "en", "zh", "de", "es", "ru", "ko", "fr", "ja", "pt", "tr", "pl", "ca", "nl", "ar", "sv", "it", "id", "hi",
    "fi", "vi", "he", "uk", "el", "ms", "cs", "ro", "da", "hu", "ta", "no", "th", "ur", "hr", "bg", "lt", "la",
    "mi", "ml", "cy", "sk", "te", "fa", "lv", "bn", "sr", "az", "sl", "kn", "et", "mk", "br", "eu", "is", "hy",
    "ne", "mn", "bs", "kk", "sq", "sw", "gl", "mr", "pa", "si", "km", "sn", "yo", "so", "af", "oc", "ka", "be",
    "tg", "sd", "gu", "am", "yi", "lo", "uz", "fo", "ht", "ps", "tk", "nn", "mt", "sa", "lb", "my", "bo", "tl",
    "mg", "as", "tt", "haw", "ln", "ha", "ba", "jw", "su", "yue",
]

# large-v3 special-token ids
EOS, SOT, LANG_EN, TRANSLATE, TRANSCRIBE, SOT_PREV, NO_SPEECH, NO_TIMESTAMPS = 50257, 50258, 50259, 50359, 50360, 50362, 50363, 50364

DIMS: Dict[str, Dict[str, int]] = {
    "large-v3": dict(d_model=1280, enc_layers=32, dec_layers=32, heads=20, ffn=5120, vocab=51866, n_mels=128,
                     max_source_positions=1500, max_target_positions=448),
    "large-v3-turbo": dict(d_model=1280, enc_layers=32, dec_layers=4, heads=20, ffn=5120, vocab=51866, n_mels=128,
                           max_source_positions=1500, max_target_positions=448),
    "tiny.en": dict(d_model=384, enc_layers=4, dec_layers=4, heads=6, ffn=1536, vocab=51864, n_mels=80,
                    max_source_positions=1500, max_target_positions=448),
}


def default_alignment_heads(dec_layers: int, heads: int) -> List[List[int]]:
    """Synthetic alignment heads (the upstream checkpoints' lists are not available offline): the upper half of the decoder
    layers, rotating heads - 10 pairs for 32-layer models like large-v3, fewer for small ones."""
    n = min(10, max(2, dec_layers * 2))
    out: List[List[int]] = []
    for j in range(n):
        layer = dec_layers - 1 - (j % max(1, dec_layers // 2))
        head = (3 * j + 1) % heads
        if [layer, head] not in out:
            out.append([layer, head])
    return out


def build_config(dims: Dict[str, int]):
    from transformers import WhisperConfig

    return WhisperConfig(
        vocab_size=dims["vocab"], num_mel_bins=dims["n_mels"], d_model=dims["d_model"],
        encoder_layers=dims["enc_layers"], decoder_layers=dims["dec_layers"],
        encoder_attention_heads=dims["heads"], decoder_attention_heads=dims["heads"],
        encoder_ffn_dim=dims["ffn"], decoder_ffn_dim=dims["ffn"],
        max_source_positions=dims.get("max_source_positions", 1500), max_target_positions=dims.get("max_target_positions", 448),
        bos_token_id=EOS, eos_token_id=EOS, pad_token_id=EOS, decoder_start_token_id=SOT,
        activation_function="gelu", dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, use_cache=True,
    )


def fill_generation_config(gc, dec_layers: int, heads: int, max_target_positions: int = 448, alignment_heads=None):
    """Hand-filled multilingual generation config."""
    gc.lang_to_id = {f"<|{l}|>": LANG_EN + i for i, l in enumerate(LANGS)}
    gc.task_to_id = {"transcribe": TRANSCRIBE, "translate": TRANSLATE}
    gc.no_timestamps_token_id = NO_TIMESTAMPS
    gc.prev_sot_token_id = SOT_PREV
    gc.is_multilingual = True
    gc.alignment_heads = alignment_heads if alignment_heads is not None else default_alignment_heads(dec_layers, heads)
    gc.max_initial_timestamp_index = 50
    gc.suppress_tokens = []
    gc.begin_suppress_tokens = [220, EOS]
    gc.max_length = max_target_positions
    gc.forced_decoder_ids = None
    gc.bos_token_id = EOS
    gc.eos_token_id = EOS
    gc.pad_token_id = EOS
    gc.decoder_start_token_id = SOT
    gc.return_timestamps = False
    return gc


def _bytes_to_unicode():
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


def build_tokenizer(vocab_size: int):
    """In-memory ``WhisperTokenizer`` with the large-v3 special-token id layout (ctor
    HF:models/whisper/tokenization_whisper.py:206-276): 256 byte symbols, filler words up to id 50256, then the specials
    and the timestamp tokens <|0.00|> ... up to ``vocab_size``."""
    from transformers import WhisperTokenizer

    vocab: Dict[str, int] = {}
    for ch in _bytes_to_unicode().values():  # the GPT-2 byte<->unicode table used by the ByteLevel pre-tokenizer/decoder
        vocab[ch] = len(vocab)
    i = 0
    while len(vocab) < EOS:
        tok = f"Ġw{i}"
        if tok not in vocab:
            vocab[tok] = len(vocab)
        i += 1
    specials = ["<|endoftext|>", "<|startoftranscript|>"]
    specials += [f"<|{l}|>" for l in LANGS]
    specials += ["<|translate|>", "<|transcribe|>", "<|startoflm|>", "<|startofprev|>", "<|nospeech|>", "<|notimestamps|>"]
    n_ts = vocab_size - (EOS + len(specials))
    specials += [f"<|{k * 0.02:.2f}|>" for k in range(n_ts)]
    for s in specials:
        vocab[s] = len(vocab)
    assert len(vocab) == vocab_size, (len(vocab), vocab_size)
    assert vocab["<|notimestamps|>"] == NO_TIMESTAMPS
    return WhisperTokenizer(
        vocab=vocab, merges=[], language="en", task="transcribe",
        # timestamps are ordinary added tokens upstream: `timestamp_begin = all_special_ids[-1] + 1`
        additional_special_tokens=[t for t in specials[1:] if vocab[t] <= NO_TIMESTAMPS],
        pad_token="<|endoftext|>", bos_token="<|endoftext|>", eos_token="<|endoftext|>", unk_token="<|endoftext|>",
    )


def random_state_dict(dims: Dict[str, int], device, seed: int = 0):
    """Random-init weights of the named architecture, generated on ``device`` in the HF state_dict layout."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    d, f, v = dims["d_model"], dims["ffn"], dims["vocab"]

    def uni(shape, amp):
        return (torch.rand(shape, device=device, generator=g, dtype=torch.float32) - 0.5) * (2 * amp)

    sd = {}

    def lin(name, o, i, bias=True):
        sd[name + ".weight"] = uni((o, i), 1.7 / i ** 0.5)
        if bias:
            sd[name + ".bias"] = uni((o,), 0.05)

    def ln(name):
        sd[name + ".weight"] = 1.0 + uni((d,), 0.1)
        sd[name + ".bias"] = uni((d,), 0.05)

    def attn(p):
        lin(p + ".k_proj", d, d, False)
        lin(p + ".v_proj", d, d)
        lin(p + ".q_proj", d, d)
        lin(p + ".out_proj", d, d)

    e = "model.encoder"
    sd[e + ".conv1.weight"] = uni((d, dims["n_mels"], 3), 1.7 / (3 * dims["n_mels"]) ** 0.5)
    sd[e + ".conv1.bias"] = uni((d,), 0.05)
    sd[e + ".conv2.weight"] = uni((d, d, 3), 1.7 / (3 * d) ** 0.5)
    sd[e + ".conv2.bias"] = uni((d,), 0.05)
    sd[e + ".embed_positions.weight"] = uni((dims.get("max_source_positions", 1500), d), 0.5)
    for i in range(dims["enc_layers"]):
        p = f"{e}.layers.{i}"
        attn(p + ".self_attn"); ln(p + ".self_attn_layer_norm"); lin(p + ".fc1", f, d); lin(p + ".fc2", d, f); ln(p + ".final_layer_norm")
    ln(e + ".layer_norm")
    dd = "model.decoder"
    sd[dd + ".embed_tokens.weight"] = uni((v, d), 0.12)
    sd[dd + ".embed_positions.weight"] = uni((dims.get("max_target_positions", 448), d), 0.12)
    for i in range(dims["dec_layers"]):
        p = f"{dd}.layers.{i}"
        attn(p + ".self_attn"); ln(p + ".self_attn_layer_norm"); attn(p + ".encoder_attn"); ln(p + ".encoder_attn_layer_norm")
        lin(p + ".fc1", f, d); lin(p + ".fc2", d, f); ln(p + ".final_layer_norm")
    ln(dd + ".layer_norm")
    return sd


def skeleton_model(dims: Dict[str, int], device="cuda", dtype=None, alignment_heads: Optional[Sequence] = None):
    """An ``AMDWhisperForConditionalGeneration`` of the named architecture WITHOUT materialised weights (parameter storage is
    released): the container HF's pipeline / generate() need around an engine that was loaded separately
    (``model.attach_engine``).  The hot path never touches the torch modules."""
    import torch
    from transformers.initialization import no_init_weights

    from .model import AMDWhisperForConditionalGeneration

    cfg = build_config(dims)
    old = torch.get_default_dtype()
    try:
        torch.set_default_dtype(dtype or torch.bfloat16)
        with no_init_weights(), torch.device("meta"):
            model = AMDWhisperForConditionalGeneration(cfg)
    finally:
        torch.set_default_dtype(old)
    dev = torch.device(device)
    for mod in model.modules():  # meta -> empty real tensors on the target device (HF's pipeline calls model.to(device))
        for name, p in list(mod._parameters.items()):
            if p is not None:
                mod._parameters[name] = torch.nn.Parameter(torch.empty(0, dtype=p.dtype, device=dev), requires_grad=False)
        for name, b in list(mod._buffers.items()):
            if b is not None:
                mod._buffers[name] = torch.empty(0, dtype=b.dtype, device=dev)
    model.eval()
    fill_generation_config(model.generation_config, dims["dec_layers"], dims["heads"], dims.get("max_target_positions", 448),
                           [list(h) for h in alignment_heads] if alignment_heads is not None else None)
    return model
