"""CPU oracle for the Whisper hot path (log-mel -> encoder -> cached-KV decoder -> greedy -> DTW).

TEST INFRASTRUCTURE ONLY.  Nothing under ``thewhisper_amd/`` may import this module; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and only as the
checker / the reported CPU baseline.

This is a plain-numpy restatement of the arithmetic that the reference delegates to the un-vendored
third-party dependency ``transformers`` (pinned ``==4.52.3`` in R:pyproject.toml:35,:45; the version
installed in this image, and therefore the parity target, is **5.15.0**).  Citations:

* ``R:<path>:<lines>``  -> /root/reference/<path>
* ``HF:<path>:<lines>`` -> site-packages/transformers/<path> (5.15.0)

Parity pinning: the reference ships no tests, fixtures or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against the reference arithmetic itself:
``tests/test_oracle_vs_hf.py`` runs the installed HF implementation live on CPU and
``tests/golden/`` holds vectors produced by driving the reference's own
``thestage_speechkit.nvidia.ASRPipeline`` (``oracle/make_golden.py``).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# --------------------------------------------------------------------------------------
# Model dimensions
# --------------------------------------------------------------------------------------


@dataclass(frozen=True)
class WhisperDims:
    """Architecture sizes (HF:models/whisper/configuration_whisper.py)."""

    d_model: int
    enc_layers: int
    dec_layers: int
    heads: int
    ffn: int
    vocab: int
    n_mels: int
    max_source_positions: int = 1500
    max_target_positions: int = 448

    @property
    def head_dim(self) -> int:
        return self.d_model // self.heads


PRESETS: Dict[str, WhisperDims] = {
    # SURVEY.md section 8 dimension table
    "tiny.en": WhisperDims(384, 4, 4, 6, 1536, 51864, 80),
    "large-v3": WhisperDims(1280, 32, 32, 20, 5120, 51866, 128),
    "large-v3-turbo": WhisperDims(1280, 32, 4, 20, 5120, 51866, 128),
    # Small configurations the numpy oracle finishes in well under a second.  They keep the
    # large-v3 special-token layout (vocab 51866) so the logits processors see real ids.
    "micro": WhisperDims(128, 2, 2, 2, 256, 51866, 128),
    "micro80": WhisperDims(192, 2, 2, 3, 384, 51864, 80),
}


@dataclass
class SpecialTokens:
    """Large-v3 special-token layout (R:thestage_speechkit/apple/model.py:333 hard-codes 50364)."""

    eos: int = 50257
    sot: int = 50258
    lang_en: int = 50259
    translate: int = 50359
    transcribe: int = 50360
    sot_prev: int = 50362
    no_speech: int = 50363
    no_timestamps: int = 50364

    @property
    def timestamp_begin(self) -> int:
        return self.no_timestamps + 1


# --------------------------------------------------------------------------------------
# Seeded weights in the HF state_dict layout
# --------------------------------------------------------------------------------------


def uniform_at(seed: int, offset: int, shape) -> np.ndarray:
    """float32 uniforms number ``offset, offset + 1, ...`` of the stream ``np.random.default_rng(seed).random(..., dtype=float32)``
    produces when drawn in one go: PCG64 yields two float32 draws per 64-bit step, low half first, so the generator is advanced
    ``offset // 2`` steps and, at an odd offset, one draw (the low half, which belongs to the value before) is discarded."""
    g = np.random.Generator(np.random.PCG64(seed))
    g.bit_generator.advance(offset // 2)
    if offset & 1:
        g.random(1, dtype=np.float32)
    return g.random(shape, dtype=np.float32)


def make_weights(dims: WhisperDims, seed: int = 0, scale: float = 1.0, q_gain: float = 1.0) -> Dict[str, np.ndarray]:
    """Deterministic float32 weights keyed by HF parameter names.

    Independent of torch/transformers RNG so that the oracle, the HF harness and the HIP engine
    can all be fed the *same* tensors on any machine.  Linear/conv/embedding weights are uniform
    with std ~ 0.05*scale (large enough that attention is not uniform and argmax margins are
    not vanishing); LayerNorm gains are 1 +- 0.1, biases +-0.05.

    ``q_gain`` multiplies every attention query projection (weight and bias) after generation (the random
    stream is unchanged).  With unit gain a 32-layer random model attends almost uniformly, its output barely
    depends on the audio and greedy decoding collapses onto one token; ``scale=0.5, q_gain=8`` (the full-depth
    goldens, oracle/make_golden_full.py) gives peaked attention, audio-dependent token paths and an active
    timestamp grammar - a harder parity target.
    """
    d, f, v = dims.d_model, dims.ffn, dims.vocab
    w: Dict[str, np.ndarray] = {}
    # The values are those of ONE generator, np.random.default_rng(seed), drawing float32 uniforms tensor after tensor in the order
    # below.  They are produced by several threads: every tensor gets its own copy of the generator, advanced to the tensor's place in
    # that stream (PCG64 yields two float32 draws per 64-bit step, low half first), so the result is bit-identical to the sequential
    # loop (tests/test_oracle_weights.py) while the 1.5 G parameters of large-v3 take tens of seconds instead of two minutes on the
    # 8-core test containers (the loop is bound by first-touch page faults, which parallelise).
    drawn = [0]

    def uni(shape, amp, plus_one=False):
        n = int(np.prod(shape))
        ticket = (shape if isinstance(shape, tuple) else (shape,), float(amp), drawn[0], plus_one)
        drawn[0] += n
        return ticket

    def fill(ticket):
        shape, amp, off, plus_one = ticket
        x = uniform_at(seed, off, shape)
        x -= np.float32(0.5)
        x *= np.float32(2.0 * amp)             # the same float32 operations, in the same order, as (u - 0.5) * (2 * amp)
        if plus_one:
            x = (1.0 + x).astype(np.float32)
        return x

    def lin(name, out_f, in_f, bias=True):
        amp = scale * 1.7 / math.sqrt(in_f)
        w[name + ".weight"] = uni((out_f, in_f), amp)
        if bias:
            w[name + ".bias"] = uni((out_f,), 0.05)

    def ln(name):
        w[name + ".weight"] = uni((d,), 0.1, plus_one=True)
        w[name + ".bias"] = uni((d,), 0.05)

    def attn(prefix):
        lin(prefix + ".k_proj", d, d, bias=False)
        lin(prefix + ".v_proj", d, d)
        lin(prefix + ".q_proj", d, d)
        lin(prefix + ".out_proj", d, d)

    e = "model.encoder"
    w[e + ".conv1.weight"] = uni((d, dims.n_mels, 3), scale * 1.7 / math.sqrt(3 * dims.n_mels))
    w[e + ".conv1.bias"] = uni((d,), 0.05)
    w[e + ".conv2.weight"] = uni((d, d, 3), scale * 1.7 / math.sqrt(3 * d))
    w[e + ".conv2.bias"] = uni((d,), 0.05)
    w[e + ".embed_positions.weight"] = sinusoids(dims.max_source_positions, d)
    for i in range(dims.enc_layers):
        p = f"{e}.layers.{i}"
        attn(p + ".self_attn")
        ln(p + ".self_attn_layer_norm")
        lin(p + ".fc1", f, d)
        lin(p + ".fc2", d, f)
        ln(p + ".final_layer_norm")
    ln(e + ".layer_norm")

    dd = "model.decoder"
    w[dd + ".embed_tokens.weight"] = uni((v, d), scale * 0.12)
    w[dd + ".embed_positions.weight"] = uni((dims.max_target_positions, d), scale * 0.12)
    for i in range(dims.dec_layers):
        p = f"{dd}.layers.{i}"
        attn(p + ".self_attn")
        ln(p + ".self_attn_layer_norm")
        attn(p + ".encoder_attn")
        ln(p + ".encoder_attn_layer_norm")
        lin(p + ".fc1", f, d)
        lin(p + ".fc2", d, f)
        ln(p + ".final_layer_norm")
    ln(dd + ".layer_norm")
    from concurrent.futures import ThreadPoolExecutor

    keys = [k for k, t in w.items() if isinstance(t, tuple)]
    with ThreadPoolExecutor(max_workers=max(1, min(8, os.cpu_count() or 1))) as pool:
        for k, x in zip(keys, pool.map(fill, [w[k] for k in keys])):
            w[k] = x
    if q_gain != 1.0:
        for k in w:
            if ".q_proj." in k:
                w[k] = (w[k] * np.float32(q_gain)).astype(np.float32)
    return w


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> np.ndarray:
    """Sinusoidal encoder positions (HF:models/whisper/modeling_whisper.py sinusoids(); R:thestage_speechkit/apple/mlx_modules.py:27-33)."""
    log_inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = np.exp(-log_inc * np.arange(channels // 2, dtype=np.float64))
    t = np.arange(length, dtype=np.float64)[:, None] * inv[None, :]
    return np.concatenate([np.sin(t), np.cos(t)], axis=1).astype(np.float32)


def interpolate_positions(pos: np.ndarray, new_len: int) -> np.ndarray:
    """A0: ``F.interpolate(mode='linear', align_corners=False)`` of the encoder positional table
    ``[1500,d] -> [new_len,d]`` (R:thestage_speechkit/nvidia/asr_pipeline.py:15-27)."""
    old_len = pos.shape[0]
    if new_len == old_len:
        return pos.astype(np.float32).copy()
    scale = np.float32(old_len) / np.float32(new_len)
    # torch area_pixel_compute_source_index, align_corners=False: src = scale*(dst+0.5)-0.5, clamped at 0
    dst = np.arange(new_len, dtype=np.float32)
    src = scale * (dst + np.float32(0.5)) - np.float32(0.5)
    src = np.maximum(src, np.float32(0.0))
    i0 = np.floor(src).astype(np.int64)
    i0 = np.minimum(i0, old_len - 1)
    i1 = np.minimum(i0 + 1, old_len - 1)
    lam1 = (src - i0.astype(np.float32)).astype(np.float32)
    lam0 = np.float32(1.0) - lam1
    p = pos.astype(np.float32)
    return (lam0[:, None] * p[i0] + lam1[:, None] * p[i1]).astype(np.float32)


# --------------------------------------------------------------------------------------
# A1: log-mel features
# --------------------------------------------------------------------------------------

N_FFT = 400
HOP = 160
SAMPLE_RATE = 16000


def _hz_to_mel_slaney(freq):
    freq = np.asarray(freq, dtype=np.float64)
    mels = 3.0 * freq / 200.0
    logstep = 27.0 / np.log(6.4)
    out = np.where(freq >= 1000.0, 15.0 + np.log(np.maximum(freq, 1e-30) / 1000.0) * logstep, mels)
    return out


def _mel_to_hz_slaney(mels):
    mels = np.asarray(mels, dtype=np.float64)
    freq = 200.0 * mels / 3.0
    logstep = np.log(6.4) / 27.0
    return np.where(mels >= 15.0, 1000.0 * np.exp(logstep * (mels - 15.0)), freq)


def mel_filter_bank(n_mels: int) -> np.ndarray:
    """Slaney-scale, slaney-normalised triangular bank ``[201, n_mels]`` float64
    (HF:audio_utils.py:638-729 with the arguments of HF:models/whisper/feature_extraction_whisper.py:95-103)."""
    n_bins = 1 + N_FFT // 2
    mel_min = _hz_to_mel_slaney(0.0)
    mel_max = _hz_to_mel_slaney(8000.0)
    mel_freqs = np.linspace(mel_min, mel_max, n_mels + 2)
    filter_freqs = _mel_to_hz_slaney(mel_freqs)
    fft_freqs = np.linspace(0, SAMPLE_RATE // 2, n_bins)
    filter_diff = np.diff(filter_freqs)
    slopes = filter_freqs[None, :] - fft_freqs[:, None]
    down = -slopes[:, :-2] / filter_diff[:-1]
    up = slopes[:, 2:] / filter_diff[1:]
    bank = np.maximum(0.0, np.minimum(down, up))
    enorm = 2.0 / (filter_freqs[2 : n_mels + 2] - filter_freqs[:n_mels])
    return bank * enorm[None, :]


def hann_window(n: int = N_FFT) -> np.ndarray:
    """``torch.hann_window(n)`` (periodic) in float64."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n, dtype=np.float64) / n)


def log_mel(pcm: np.ndarray, n_mels: int, n_samples: Optional[int] = None) -> np.ndarray:
    """A1.  ``pcm`` float [B, n] (or [n]); zero-pad/truncate to ``n_samples``; returns float32
    ``[B, n_mels, n_samples//160]``.

    Restates HF:models/whisper/feature_extraction_whisper.py:135-168 (``torch.stft`` n_fft=400,
    hop=160, periodic Hann, centre reflect padding; drop the last frame; |.|^2; mel; log10 clamp
    1e-10; per-clip ``max-8`` floor; ``(x+4)/4``).  The DFT is done in float64 (an exact-math
    reference; torch's float32 FFT differs from it at the 1e-6 level).
    """
    pcm = np.asarray(pcm)
    if pcm.ndim == 1:
        pcm = pcm[None]
    b, n = pcm.shape
    if n_samples is None:
        n_samples = n
    x = np.zeros((b, n_samples), dtype=np.float64)
    m = min(n, n_samples)
    x[:, :m] = pcm[:, :m].astype(np.float32).astype(np.float64)
    pad = N_FFT // 2
    xp = np.pad(x, ((0, 0), (pad, pad)), mode="reflect")
    n_frames = n_samples // HOP  # == (1 + n_samples//HOP) - 1, last STFT frame dropped
    idx = np.arange(n_frames)[:, None] * HOP + np.arange(N_FFT)[None, :]
    frames = xp[:, idx] * hann_window()[None, None, :]  # [B, F, 400]
    spec = np.fft.rfft(frames, n=N_FFT, axis=-1)  # [B, F, 201]
    power = spec.real**2 + spec.imag**2
    bank = mel_filter_bank(n_mels).astype(np.float32).astype(np.float64)  # HF casts the bank to f32
    mel = np.einsum("km,bfk->bmf", bank, power)
    log_spec = np.log10(np.maximum(mel, 1e-10))
    mx = log_spec.max(axis=(1, 2), keepdims=True)
    log_spec = np.maximum(log_spec, mx - 8.0)
    return ((log_spec + 4.0) / 4.0).astype(np.float32)


# --------------------------------------------------------------------------------------
# A2-A8: encoder / decoder arithmetic
# --------------------------------------------------------------------------------------


def _erf(x: np.ndarray) -> np.ndarray:
    try:
        from scipy.special import erf  # exact to double precision

        return erf(x)
    except Exception:  # pragma: no cover
        return np.vectorize(math.erf)(x)


def gelu(x: np.ndarray) -> np.ndarray:
    """Exact (erf) GELU, ``activation_function='gelu'``."""
    return (0.5 * x * (1.0 + _erf(x / math.sqrt(2.0)))).astype(x.dtype)


def layer_norm(x: np.ndarray, g: np.ndarray, b: np.ndarray, eps: float = 1e-5) -> np.ndarray:
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return ((x - mu) / np.sqrt(var + eps) * g + b).astype(x.dtype)


def softmax(x: np.ndarray, axis: int = -1) -> np.ndarray:
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


@dataclass
class DecoderCache:
    """Per-batch decoder state: growing self K/V and per-chunk cross K/V (A5, A7)."""

    self_k: List[np.ndarray]
    self_v: List[np.ndarray]
    cross_k: List[np.ndarray]
    cross_v: List[np.ndarray]
    length: int = 0


class OracleWhisper:
    """Numpy Whisper (HF:models/whisper/modeling_whisper.py:360-505, :540-795, :1080).

    ``dtype`` float32 mirrors the reference CPU path; float64 gives a high-precision reference
    for tolerance budgeting.  ``T`` (encoder frames) may be < 1500: the positional table is
    interpolated exactly as R:thestage_speechkit/nvidia/asr_pipeline.py:15-27 (A0).
    """

    def __init__(self, dims: WhisperDims, weights: Dict[str, np.ndarray], T: Optional[int] = None, dtype=np.float32):
        self.dims = dims
        self.dtype = dtype
        self.T = T or dims.max_source_positions
        self.w = {k: np.asarray(v).astype(dtype) for k, v in weights.items()}
        pos = np.asarray(weights["model.encoder.embed_positions.weight"], dtype=np.float32)
        self.enc_pos = interpolate_positions(pos, self.T).astype(dtype)

    # ---- helpers -------------------------------------------------------------------------
    def _lin(self, x, name, bias=True):
        y = x @ self.w[name + ".weight"].T
        if bias:
            y = y + self.w[name + ".bias"]
        return y

    def _ln(self, x, name):
        return layer_norm(x, self.w[name + ".weight"], self.w[name + ".bias"])

    # decoder projections go through these two hooks so that OracleWhisperMXFP8 can restate the quantised arithmetic
    def _dec_proj(self, x, ln_name, lin_name, bias=True):
        """pre-LayerNorm followed by a linear layer (HF:models/whisper/modeling_whisper.py:441-470)."""
        memo = getattr(self, "_ln_memo", None)
        if memo is None or memo[0] is not x or memo[1] != ln_name:  # q/k/v share one normalised input
            memo = (x, ln_name, self._ln(x, ln_name))
            self._ln_memo = memo
        return self._lin(memo[2], lin_name, bias)

    def _dec_lin(self, x, lin_name):
        return self._lin(x, lin_name)

    def _dec_logits(self, x):
        d = "model.decoder"
        return self._ln(x, d + ".layer_norm") @ self.w[d + ".embed_tokens.weight"].T  # tied proj_out (HF :965, :1080)

    def _heads(self, x):  # [B, n, d] -> [B, H, n, hd]
        b, n, _ = x.shape
        return x.reshape(b, n, self.dims.heads, self.dims.head_dim).transpose(0, 2, 1, 3)

    def _merge(self, x):  # [B, H, n, hd] -> [B, n, d]
        b, h, n, hd = x.shape
        return x.transpose(0, 2, 1, 3).reshape(b, n, h * hd)

    # ---- A2: conv stem ---------------------------------------------------------------------
    def conv_stem(self, mel: np.ndarray) -> np.ndarray:
        """mel [B, n_mels, 2T] -> [B, T, d]  (HF:models/whisper/modeling_whisper.py:612-625)."""
        x = mel.astype(self.dtype)
        b, c, n = x.shape
        if n != 2 * self.T:
            raise ValueError(f"expected {2 * self.T} mel frames, got {n}")  # HF :612-617
        w1, b1 = self.w["model.encoder.conv1.weight"], self.w["model.encoder.conv1.bias"]
        xp = np.pad(x, ((0, 0), (0, 0), (1, 1)))
        # out[b,co,t] = sum_{ci,k} w1[co,ci,k] * xp[b,ci,t+k]
        cols = np.stack([xp[:, :, k : k + n] for k in range(3)], axis=-1)  # [B, ci, n, 3]
        h = np.einsum("bitk,oik->bto", cols, w1) + b1
        h = gelu(h)  # [B, 2T, d]
        w2, b2 = self.w["model.encoder.conv2.weight"], self.w["model.encoder.conv2.bias"]
        hp = np.pad(h, ((0, 0), (1, 1), (0, 0)))
        cols2 = np.stack([hp[:, k : k + n : 2, :] for k in range(3)], axis=-1)  # [B, T, ci, 3]
        y = np.einsum("btik,oik->bto", cols2, w2) + b2
        y = gelu(y)
        return (y + self.enc_pos[None]).astype(self.dtype)

    # ---- A3/A4: encoder --------------------------------------------------------------------
    def encoder_layer(self, x: np.ndarray, i: int) -> np.ndarray:
        p = f"model.encoder.layers.{i}"
        scale = self.dtype(self.dims.head_dim**-0.5)
        h = self._ln(x, p + ".self_attn_layer_norm")
        q = self._heads(self._lin(h, p + ".self_attn.q_proj") * scale)  # scale before QK^T (HF :309)
        k = self._heads(self._lin(h, p + ".self_attn.k_proj", bias=False))
        v = self._heads(self._lin(h, p + ".self_attn.v_proj"))
        a = softmax(q @ k.transpose(0, 1, 3, 2)) @ v
        x = x + self._lin(self._merge(a), p + ".self_attn.out_proj")
        h = self._ln(x, p + ".final_layer_norm")
        h = gelu(self._lin(h, p + ".fc1"))
        return (x + self._lin(h, p + ".fc2")).astype(self.dtype)

    def encode(self, mel: np.ndarray, return_layers: bool = False):
        x = self.conv_stem(mel)
        layers = [x]
        for i in range(self.dims.enc_layers):
            x = self.encoder_layer(x, i)
            layers.append(x)
        out = self._ln(x, "model.encoder.layer_norm")
        return (out, layers) if return_layers else out

    # ---- A5: cross K/V ---------------------------------------------------------------------
    def new_cache(self, enc: np.ndarray) -> DecoderCache:
        ck, cv = [], []
        for i in range(self.dims.dec_layers):
            p = f"model.decoder.layers.{i}.encoder_attn"
            ck.append(self._heads(self._lin(enc, p + ".k_proj", bias=False)))
            cv.append(self._heads(self._lin(enc, p + ".v_proj")))
        b = enc.shape[0]
        hd, H = self.dims.head_dim, self.dims.heads
        empty = lambda: np.zeros((b, H, 0, hd), dtype=self.dtype)  # noqa: E731
        L = self.dims.dec_layers
        return DecoderCache([empty() for _ in range(L)], [empty() for _ in range(L)], ck, cv, 0)

    # ---- A6-A8: decoder forward over n new tokens -----------------------------------------
    def decode(self, ids: np.ndarray, cache: DecoderCache, want_cross: Optional[Sequence[Tuple[int, int]]] = None):
        """ids int [B, n] appended at position ``cache.length``.  Returns (logits [B, n, V] float,
        cross-attention probabilities of the requested (layer, head) pairs [B, Ha, n, T] or None)."""
        ids = np.asarray(ids)
        b, n = ids.shape
        past = cache.length
        d = "model.decoder"
        x = self.w[d + ".embed_tokens.weight"][ids] + self.w[d + ".embed_positions.weight"][past : past + n][None]
        scale = self.dtype(self.dims.head_dim**-0.5)
        causal = np.triu(np.full((n, past + n), -np.inf, dtype=self.dtype), k=past + 1)
        cross = [] if want_cross is not None else None
        want = {}
        if want_cross is not None:
            for j, (l, h) in enumerate(want_cross):
                want.setdefault(int(l), []).append((j, int(h)))
            cross = [None] * len(want_cross)
        for i in range(self.dims.dec_layers):
            p = f"{d}.layers.{i}"
            ln = p + ".self_attn_layer_norm"
            q = self._heads(self._dec_proj(x, ln, p + ".self_attn.q_proj") * scale)
            k = self._heads(self._dec_proj(x, ln, p + ".self_attn.k_proj", bias=False))
            v = self._heads(self._dec_proj(x, ln, p + ".self_attn.v_proj"))
            cache.self_k[i] = np.concatenate([cache.self_k[i], k], axis=2)
            cache.self_v[i] = np.concatenate([cache.self_v[i], v], axis=2)
            s = q @ cache.self_k[i].transpose(0, 1, 3, 2) + causal[None, None]
            a = softmax(s) @ cache.self_v[i]
            x = x + self._dec_lin(self._merge(a), p + ".self_attn.out_proj")
            q = self._heads(self._dec_proj(x, p + ".encoder_attn_layer_norm", p + ".encoder_attn.q_proj") * scale)
            pr = softmax(q @ cache.cross_k[i].transpose(0, 1, 3, 2))
            if i in want:
                for j, hh in want[i]:
                    cross[j] = pr[:, hh]  # [B, n, T]
            a = pr @ cache.cross_v[i]
            x = x + self._dec_lin(self._merge(a), p + ".encoder_attn.out_proj")
            h_ = gelu(self._dec_proj(x, p + ".final_layer_norm", p + ".fc1"))
            x = (x + self._dec_lin(h_, p + ".fc2")).astype(self.dtype)
        cache.length = past + n
        logits = self._dec_logits(x)
        cross_out = None
        if cross is not None:
            cross_out = np.stack(cross, axis=1)  # [B, Ha, n, T]
        return logits, cross_out


# --------------------------------------------------------------------------------------
# BASELINE config 5: MXFP8 decoder projections (restates thewhisper_amd/csrc/k_decode.hip: sk_quant_mx8,
# quant_mx8_kernel, skinny_mfma_kernel<W8>; there is no reference implementation of this mode - the
# reference's fp8 engines are closed TensorRT plans, R:thestage_speechkit/nvidia/asr_pipeline.py:48-56)
# --------------------------------------------------------------------------------------


def bf16_round(x: np.ndarray) -> np.ndarray:
    """float32 -> nearest bfloat16 (ties to even) -> float32."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def e4m3_rne(v: np.ndarray) -> np.ndarray:
    """Round to the OCP e4m3 grid, ties to even; valid for |v| < 256 (callers scale into that range)."""
    v = np.asarray(v, dtype=np.float64)
    a = np.abs(v)
    e = np.floor(np.log2(np.maximum(a, 2.0**-6)))  # binade (values below the smallest normal share the subnormal spacing)
    step = 2.0 ** (e - 3)
    return np.sign(v) * np.rint(a / step) * step


def mx8_quant_dequant(x: np.ndarray) -> np.ndarray:
    """Quantise-dequantise the last axis (K, multiple of 128) in the blocks of 32 that the gfx950 scaled MFMA scales
    together under the engine's operand layout: within each 128-wide step, k = (2h + mm) * 32 + (2u + kk) * 8 + e with
    mm, kk in {0, 1}, e < 8 form block (h, u) (thewhisper_amd/csrc/k_decode.hip: sk_quant_mx8).  Scale = 2^(E - 7 - 127)
    with E the biased exponent of the block maximum (one exponent above the OCP MX convention: the hardware convert does
    not saturate), elements RNE to e4m3."""
    x = np.asarray(x, dtype=np.float32)
    K = x.shape[-1]
    assert K % 128 == 0
    lead = x.shape[:-1]
    v = x.reshape(*lead, K // 128, 2, 2, 2, 2, 8)  # [.., s, h, mm, u, kk, e]
    amax = np.abs(v).max(axis=(-4, -2, -1), keepdims=True)  # over (mm, kk, e) for each (s, h, u)
    Eb = np.floor(np.log2(np.maximum(amax, 2.0**-126))).astype(np.int64) + 127
    sb = np.maximum(Eb - 7, 1)
    X = np.ldexp(1.0, sb - 127)
    q = e4m3_rne(v.astype(np.float64) / X)
    return (q * X).astype(np.float32).reshape(x.shape)


def kv8_quant_dequant(x: np.ndarray) -> np.ndarray:
    """fp8 cross-attention K / V^T caches of the TW_BF16_MXFP8 engine (thewhisper_amd/csrc/k_gemm.hip: gemm_epilogue_kv8):
    per (stream, head, key) the 64 head dims share one power-of-two scale 2^(sb-127), sb = max(E - 7, 1) with E the biased
    exponent of the largest bf16 magnitude (the rule of mx8_quant_dequant), elements RNE to e4m3.  Last axis = head dim."""
    xb = bf16_round(x)
    amax = np.abs(xb).max(axis=-1, keepdims=True)
    Eb = np.floor(np.log2(np.maximum(amax, 2.0**-126))).astype(np.int64) + 127
    sb = np.maximum(Eb - 7, 1)
    X = np.ldexp(1.0, sb - 127)
    return (e4m3_rne(xb.astype(np.float64) / X) * X).astype(np.float32)


class OracleWhisperMXFP8(OracleWhisper):
    """bf16 activations, decoder projection weights in MXFP8 with the pre-LayerNorm folded into the weights, as the
    TW_BF16_MXFP8 engine computes:  y = rstd * (Q(x) . Q(W')^T - mean * gW) + cb,  W' = bf16(g * W).
    ``act_quant=False`` restates the TW_BF16_W8A16 engine: the same quantised weights, widened to bf16 (exactly), against the
    UNQUANTISED bf16 activations:  y = rstd * (x . Q(W')^T - mean * gW) + cb."""

    def __init__(self, dims, weights, T=None, cross_q_ahead=True, act_quant=True):
        super().__init__(dims, weights, T=T, dtype=np.float32)
        self._qa = mx8_quant_dequant if act_quant else (lambda v: np.asarray(v, dtype=np.float32))
        self._fold_cache: Dict[Tuple[str, str], Tuple[np.ndarray, np.ndarray, np.ndarray]] = {}
        self._wq_cache: Dict[str, np.ndarray] = {}
        self._ahead_cache: Dict[int, Tuple[np.ndarray, np.ndarray]] = {}
        # the engine accumulates the cross-attention query ahead of its LayerNorm ("cross query ahead"):
        # x1 W'^T = x0 W'^T + attn (W' Wo)^T + W' bo, with W' Wo composed once (fp32), rounded to bf16 and quantised like
        # every other weight; False restates the plain sequence (LayerNorm'd x1 times the quantised W')
        self.cross_q_ahead = cross_q_ahead

    def _folded(self, ln_name, w_name, b_name):
        key = (ln_name, w_name)
        if key not in self._fold_cache:
            g = bf16_round(self.w[ln_name + ".weight"])
            beta = bf16_round(self.w[ln_name + ".bias"])
            W = bf16_round(self.w[w_name])
            Wf = bf16_round(W * g[None, :])
            gw = (W.astype(np.float64) @ g.astype(np.float64)).astype(np.float32)
            cb = W.astype(np.float64) @ beta.astype(np.float64)
            if b_name is not None:
                cb = cb + bf16_round(self.w[b_name]).astype(np.float64)
            self._fold_cache[key] = (mx8_quant_dequant(Wf), gw, cb.astype(np.float32))
            self._fold_cache[(ln_name, w_name, "bf16")] = Wf
        return self._fold_cache[key]

    def _cross_q_ahead(self, i, x0, attn_b, x1):
        """Pre-softmax cross-attention query of layer i the way the fused launches form it (api.hip: decode_core)."""
        p = f"model.decoder.layers.{i}"
        ln, wn = p + ".encoder_attn_layer_norm", p + ".encoder_attn.q_proj"
        Wq, gw, cb = self._folded(ln, wn + ".weight", wn + ".bias")
        if i not in self._ahead_cache:
            Wf = self._fold_cache[(ln, wn + ".weight", "bf16")]
            Wo = bf16_round(self.w[p + ".self_attn.out_proj.weight"])
            bo = bf16_round(self.w[p + ".self_attn.out_proj.bias"])
            Wc = bf16_round((Wf.astype(np.float32) @ Wo.astype(np.float32)).astype(np.float32))
            self._ahead_cache[i] = (mx8_quant_dequant(Wc), (Wf.astype(np.float32) @ bo.astype(np.float32)).astype(np.float32))
        Wc_q, c0 = self._ahead_cache[i]
        u = (self._qa(bf16_round(x0)) @ Wq.T + c0) + self._qa(attn_b) @ Wc_q.T
        xb = bf16_round(x1)
        mean = xb.mean(axis=-1, keepdims=True)
        var = np.maximum((xb.astype(np.float64) ** 2).mean(axis=-1, keepdims=True) - mean.astype(np.float64) ** 2, 0.0)
        rstd = (1.0 / np.sqrt(var + 1e-5)).astype(np.float32)
        return rstd * (u - mean * gw) + cb

    def _folded_apply(self, x, ln_name, w_name, b_name):
        xb = bf16_round(x)
        Wq, gw, cb = self._folded(ln_name, w_name, b_name)
        mean = xb.mean(axis=-1, keepdims=True)
        var = np.maximum((xb.astype(np.float64) ** 2).mean(axis=-1, keepdims=True) - mean.astype(np.float64) ** 2, 0.0)
        rstd = (1.0 / np.sqrt(var + 1e-5)).astype(np.float32)
        acc = self._qa(xb) @ Wq.T
        return rstd * (acc - mean * gw) + cb

    def _dec_proj(self, x, ln_name, lin_name, bias=True):
        return self._folded_apply(x, ln_name, lin_name + ".weight", lin_name + ".bias" if bias else None)

    def _dec_lin(self, x, lin_name):
        if lin_name not in self._wq_cache:
            self._wq_cache[lin_name] = mx8_quant_dequant(bf16_round(self.w[lin_name + ".weight"]))
        return self._qa(bf16_round(x)) @ self._wq_cache[lin_name].T + bf16_round(self.w[lin_name + ".bias"])

    def _dec_logits(self, x):
        d = "model.decoder"
        return self._folded_apply(x, d + ".layer_norm", d + ".embed_tokens.weight", None)

    def new_cache(self, enc):
        """Cross K/V as the engine's GEMM produces them: bf16 encoder states x bf16 weights, fp32 accumulation, rounded to bf16
        and stored as e4m3 with one power-of-two scale per (key, head) (kv8_quant_dequant)."""
        r = bf16_round
        e = r(enc)
        ck, cv = [], []
        for i in range(self.dims.dec_layers):
            p = f"model.decoder.layers.{i}.encoder_attn"
            ck.append(kv8_quant_dequant(self._heads(r(e @ r(self.w[p + ".k_proj.weight"]).T))))
            cv.append(kv8_quant_dequant(self._heads(r(e @ r(self.w[p + ".v_proj.weight"]).T + r(self.w[p + ".v_proj.bias"])))))
        b = enc.shape[0]
        empty = lambda: np.zeros((b, self.dims.heads, 0, self.dims.head_dim), dtype=np.float32)  # noqa: E731
        L = self.dims.dec_layers
        return DecoderCache([empty() for _ in range(L)], [empty() for _ in range(L)], ck, cv, 0)

    @staticmethod
    def _attend(q, K, V, mask=None):
        """softmax(q K^T) V the way the decode attention kernels evaluate it: fp32 scores, unnormalised probabilities
        rounded to bf16 for the P.V product, fp32 sum of the unrounded probabilities as the normaliser."""
        sc = q @ K.transpose(0, 1, 3, 2)
        if mask is not None:
            sc = sc + mask
        p = np.exp(sc - sc.max(axis=-1, keepdims=True))
        return (bf16_round(p) @ V) / p.sum(axis=-1, keepdims=True), p / p.sum(axis=-1, keepdims=True)

    def decode(self, ids, cache, want_cross=None):
        """Same dataflow as OracleWhisper.decode with the engine's storage roundings made explicit: every tensor the
        TW_BF16_MXFP8 engine keeps in HBM (residual stream, q/k/v, attention outputs, FFN hidden) is rounded to bf16 where
        the engine rounds it, so that the fp8 quantiser sees the same inputs (a value that bf16 rounding moves across an
        e4m3 rounding boundary would otherwise change by a whole fp8 step)."""
        ids = np.asarray(ids)
        b, n = ids.shape
        past = cache.length
        d = "model.decoder"
        r = bf16_round
        x = r(r(self.w[d + ".embed_tokens.weight"])[ids] + r(self.w[d + ".embed_positions.weight"])[past : past + n][None])
        scale = np.float32(self.dims.head_dim**-0.5)
        causal = np.triu(np.full((n, past + n), -np.inf, dtype=np.float32), k=past + 1)
        cross = [None] * len(want_cross) if want_cross is not None else None
        want = {}
        if want_cross is not None:
            for j, (l, h) in enumerate(want_cross):
                want.setdefault(int(l), []).append((j, int(h)))
        for i in range(self.dims.dec_layers):
            p = f"{d}.layers.{i}"
            ln = p + ".self_attn_layer_norm"
            q = self._heads(r(self._dec_proj(x, ln, p + ".self_attn.q_proj") * scale))
            k = self._heads(r(self._dec_proj(x, ln, p + ".self_attn.k_proj", bias=False)))
            v = self._heads(r(self._dec_proj(x, ln, p + ".self_attn.v_proj")))
            cache.self_k[i] = np.concatenate([cache.self_k[i], k], axis=2)
            cache.self_v[i] = np.concatenate([cache.self_v[i], v], axis=2)
            a, _ = self._attend(q, cache.self_k[i], cache.self_v[i], causal[None, None])
            x0, attn_b = x, r(self._merge(a))
            x = r(x0 + self._dec_lin(attn_b, p + ".self_attn.out_proj"))
            if self.cross_q_ahead:
                q = self._heads(r(self._cross_q_ahead(i, x0, attn_b, x) * scale))
            else:
                q = self._heads(r(self._dec_proj(x, p + ".encoder_attn_layer_norm", p + ".encoder_attn.q_proj") * scale))
            a, pr = self._attend(q, r(cache.cross_k[i]), r(cache.cross_v[i]))
            if i in want:
                for j, hh in want[i]:
                    cross[j] = pr[:, hh]
            x = r(x + self._dec_lin(r(self._merge(a)), p + ".encoder_attn.out_proj"))
            h_ = r(gelu(self._dec_proj(x, p + ".final_layer_norm", p + ".fc1")))
            x = r(x + self._dec_lin(h_, p + ".fc2"))
        cache.length = past + n
        logits = self._dec_logits(x)
        return logits, (np.stack(cross, axis=1) if cross is not None else None)


# --------------------------------------------------------------------------------------
# A10: logits processors (decisions are discrete -> must match exactly)
# --------------------------------------------------------------------------------------


@dataclass
class GreedyOptions:
    """Options of one short-form greedy call (A9/A10)."""

    eos: int = 50257
    pad: int = 50257
    max_new_tokens: int = 128
    min_new_tokens: int = 0
    max_length: int = 448
    begin_suppress: Tuple[int, ...] = (220, 50257)
    suppress: Tuple[int, ...] = ()
    timestamps: bool = False
    no_timestamps_id: int = 50364
    max_initial_timestamp_index: Optional[int] = 50
    alignment_heads: Optional[Sequence[Tuple[int, int]]] = None


def logsumexp(x: np.ndarray) -> float:
    x = np.asarray(x, dtype=np.float32)
    m = x.max()
    if not np.isfinite(m):
        return float(m)
    return float(m + np.log(np.exp(x - m, dtype=np.float32).sum(dtype=np.float32)))


def apply_logits_processors(scores: np.ndarray, seq: Sequence[int], begin_index: int, opt: GreedyOptions) -> np.ndarray:
    """One row of fp32 ``scores`` [V] given the tokens so far ``seq`` (prompt included).

    Order as HF builds it: MinNewTokens (default list, HF:generation/utils.py _get_logits_processor)
    then [SuppressTokensAtBegin, SuppressTokens, WhisperTimeStamp]
    (HF:models/whisper/generation_whisper.py:1774-1812; processors
    HF:generation/logits_process.py:1816-1866, :1869-1906, :1909-2047).
    """
    s = scores.astype(np.float32).copy()
    ninf = np.float32(-np.inf)
    cur_len = len(seq)
    if opt.min_new_tokens > 0 and cur_len - begin_index < opt.min_new_tokens:
        s[opt.eos] = ninf
    if opt.begin_suppress and cur_len == begin_index:
        s[list(opt.begin_suppress)] = ninf
    if opt.suppress:
        s[list(opt.suppress)] = ninf
    if opt.timestamps:
        ts_begin = opt.no_timestamps_id + 1
        s[opt.no_timestamps_id] = ninf
        sampled = list(seq[begin_index:])
        last_ts = len(sampled) >= 1 and sampled[-1] >= ts_begin
        penult_ts = len(sampled) < 2 or sampled[-2] >= ts_begin
        if last_ts:
            if penult_ts:
                s[ts_begin:] = ninf
            else:
                s[: opt.eos] = ninf
        stamps = [t for t in sampled if t >= ts_begin]
        if stamps:
            if last_ts and not penult_ts:
                ts_last = stamps[-1]
            else:
                ts_last = stamps[-1] + 1
            s[ts_begin:ts_last] = ninf
        if cur_len == begin_index:
            s[:ts_begin] = ninf
            if opt.max_initial_timestamp_index is not None:
                last_allowed = ts_begin + opt.max_initial_timestamp_index
                s[last_allowed + 1 :] = ninf
        # "if sum of probability over timestamps is above any other token, sample timestamp":
        # log_softmax subtracts the same constant on both sides of the comparison.
        lse_all = logsumexp(s)
        ts_lp = logsumexp(s[ts_begin:]) - lse_all
        text_lp = float(s[:ts_begin].max()) - lse_all
        if ts_lp > text_lp:
            s[:ts_begin] = ninf
    return s


def greedy_generate(
    model: OracleWhisper,
    enc: np.ndarray,
    prompt: np.ndarray,
    opt: GreedyOptions,
    teacher: Optional[np.ndarray] = None,
    begin_index: Optional[int] = None,
):
    """A9 inner loop: HF:generation/utils.py:2783-2946 for num_beams=1, do_sample=False.
    ``begin_index`` < n0: the prompt's tail from there on is OUTPUT that is already known (a generation that is being continued:
    the logits processors' begin index and the new-token budget count from ``begin_index``, as they did when those tokens were
    produced; the engine's tw_greedy_opts::n_forced).

    ``prompt`` int [B, n0].  Returns dict(sequences [B, n0+G] padded with ``pad``, logits list
    (raw fp32 last-position logits per step, [B, V]), cross [B, Ha, n0+G-1, T] or None).
    ``teacher`` (optional [B, G]) forces the chosen tokens (teacher forcing for tolerance tests).
    """
    prompt = np.asarray(prompt, dtype=np.int64)
    b, n0 = prompt.shape
    cache = model.new_cache(enc)
    seqs = [list(map(int, prompt[i])) for i in range(b)]
    unfinished = np.ones(b, dtype=bool)
    nb = n0 if begin_index is None else int(begin_index)
    max_len = min(opt.max_length, nb + opt.max_new_tokens)
    raw_logits: List[np.ndarray] = []
    cross_rows: List[np.ndarray] = []
    feed = prompt
    step = 0
    while True:
        logits, cross = model.decode(feed, cache, want_cross=opt.alignment_heads)
        if cross is not None:
            cross_rows.append(cross)
        last = logits[:, -1].astype(np.float32)
        raw_logits.append(last)
        nxt = np.zeros(b, dtype=np.int64)
        for i in range(b):
            sc = apply_logits_processors(last[i], seqs[i], nb, opt)
            tok = int(np.argmax(sc))  # first maximal index, as torch.argmax
            if teacher is not None:
                tok = int(teacher[i, step])
            if not unfinished[i]:
                tok = opt.pad
            nxt[i] = tok
            seqs[i].append(tok)
        step += 1
        unfinished &= nxt != opt.eos
        if len(seqs[0]) >= max_len or not unfinished.any():
            break
        feed = nxt[:, None]
    out = {
        "sequences": np.asarray(seqs, dtype=np.int64),
        "logits": raw_logits,
        "cross": np.concatenate(cross_rows, axis=2) if cross_rows else None,
    }
    return out


# --------------------------------------------------------------------------------------
# A11: word-timestamp alignment (median filter + DTW)
# --------------------------------------------------------------------------------------


def median_filter(x: np.ndarray, width: int = 7) -> np.ndarray:
    """Median filter along the last axis with reflect padding (HF:models/whisper/generation_whisper.py:43-61)."""
    if width <= 0 or width % 2 != 1:
        raise ValueError("`filter_width` should be an odd number")
    pad = width // 2
    if x.shape[-1] <= pad:
        return x
    xp = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(pad, pad)], mode="reflect")
    win = np.lib.stride_tricks.sliding_window_view(xp, width, axis=-1)
    return np.sort(win, axis=-1)[..., pad]


def dtw(matrix: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """``_dynamic_time_warping`` (HF:models/whisper/generation_whisper.py:64-115): float32 cost table
    filled column-major with strict-< tie breaks, then backtrace."""
    n, m = matrix.shape
    cost = np.full((n + 1, m + 1), np.inf, dtype=np.float32)
    trace = -np.ones((n + 1, m + 1), dtype=np.int8)
    cost[0, 0] = 0
    mat = np.asarray(matrix, dtype=np.float64)
    for j in range(1, m + 1):
        for i in range(1, n + 1):
            c0, c1, c2 = cost[i - 1, j - 1], cost[i - 1, j], cost[i, j - 1]
            if c0 < c1 and c0 < c2:
                c, t = c0, 0
            elif c1 < c0 and c1 < c2:
                c, t = c1, 1
            else:
                c, t = c2, 2
            cost[i, j] = mat[i - 1, j - 1] + np.float64(c)  # rounded to f32 on store, as numpy does
            trace[i, j] = t
    i, j = n, m
    trace[0, :] = 2
    trace[:, 0] = 1
    ti, tj = [], []
    while i > 0 or j > 0:
        ti.append(i - 1)
        tj.append(j - 1)
        t = trace[i, j]
        if t == 0:
            i -= 1
            j -= 1
        elif t == 1:
            i -= 1
        else:
            j -= 1
    return np.array(ti[::-1]), np.array(tj[::-1])


def hf_kept_columns(num_frames, batch: int, t_cols: int) -> List[int]:
    """Columns of the alignment matrix HF's ``_extract_token_timestamps`` keeps per row (HF:models/whisper/
    generation_whisper.py:310-330 and :357-359), as the literal Python slices it applies:

      * an ``int``: ``weights[..., : n // 2]`` once, for the whole batch;
      * a list / tuple / array / tensor whose values are ALL EQUAL: the same slice for the whole batch (:317-321) AND, in the
        per-row loop, ``weights[b, ..., : num_frames[b] // 2]`` again on the already cropped matrix (:357-359).  For a
        non-negative bound the second slice changes nothing; for a NEGATIVE one (``num_frames - seek < 0``: a seek iteration
        past the end of a clip shorter than the chunk, :1152-1155) it removes ``|n // 2|`` columns a second time - with fewer
        columns than that left, none remain and every token gets the empty-matrix DTW's time index -1;
      * values that differ: the per-row slice only.

    Floor division; a negative bound counts from the end, as Python slices do."""
    def crop(length: int, k: int) -> int:
        return min(length, k) if k >= 0 else max(0, length + k)

    if num_frames is None:
        return [t_cols] * batch
    if isinstance(num_frames, (int, np.integer)):
        return [crop(t_cols, int(num_frames) // 2)] * batch
    nf = [int(x) for x in (num_frames.tolist() if hasattr(num_frames, "tolist") else list(num_frames))]
    if len(nf) != batch:
        nf = [int(x) for x in np.repeat(nf, batch // len(nf))]
    if len(set(nf)) == 1:
        k = nf[0] // 2
        return [crop(crop(t_cols, k), k)] * batch
    return [crop(t_cols, n // 2) for n in nf]


def token_timestamps(
    cross: np.ndarray, num_input_ids: int, num_frames=None, time_precision: float = 0.02,
    median_width: int = 7, columns: Optional[Sequence[int]] = None,
) -> np.ndarray:
    """``_extract_token_timestamps`` (HF:models/whisper/generation_whisper.py:241-381), 5.15.0 semantics (D4):
    ``cross`` float32 [B, Ha, N_rows, T] (rows = prompt + generated[:-1]); crop the time axis as HF does for this
    ``num_frames`` (an int, or one value per row: ``hf_kept_columns``), drop the prompt rows, z-score over the token axis,
    median filter over time, mean over heads, DTW on ``-matrix``, ``[0]*prompt ++ jump_times ++ [last]``.
    ``columns`` (instead of ``num_frames``): the number of leading time columns to keep per row, given outright - the C ABI's
    level (tw_token_timestamps takes one bound per row; the batch-level rules above are the host mirror's business).
    Returns float32 [B, N_rows + 1]."""
    cross = np.asarray(cross, dtype=np.float32)
    b, _, n_rows, t_cols = cross.shape
    out = np.zeros((b, n_rows + 1), dtype=np.float32)
    cols = [int(c) for c in columns] if columns is not None else hf_kept_columns(num_frames, b, t_cols)
    for bi in range(b):
        w = cross[bi][..., : cols[bi]]
        w = w[:, num_input_ids:, :]
        if w.shape[1] == 0:
            continue
        if w.shape[2] == 0:
            # HF runs its DTW on the empty matrix: the back-trace walks the token axis at time index -1 (:88-115)
            jump_times = np.full(w.shape[1], np.float32(-1 * time_precision), dtype=np.float32)
        else:
            with np.errstate(divide="ignore", invalid="ignore"):
                std = w.std(axis=-2, keepdims=True)  # population std (unbiased=False)
                mean = w.mean(axis=-2, keepdims=True)
                w = (w - mean) / std
            w = median_filter(w, median_width)
            mat = w.mean(axis=0)
            text_idx, time_idx = dtw(-mat.astype(np.float64))
            jumps = np.pad(np.diff(text_idx), (1, 0), constant_values=1).astype(bool)
            jump_times = (time_idx[jumps] * time_precision).astype(np.float32)
        out[bi] = np.concatenate([np.zeros(num_input_ids, np.float32), jump_times, jump_times[-1:]])
    return out


# --------------------------------------------------------------------------------------
# Synthetic inputs
# --------------------------------------------------------------------------------------


def energy_vad(pcm: np.ndarray, state: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
    """Restatement of thewhisper_amd/csrc/k_vad.hip (the stand-in for the reference's silero-vad gate,
    R:thestage_speechkit/streaming/streaming_pipeline.py:533-538, :589-622; NOT silero - see the kernel header).
    pcm [B, k*512] float32, state [B, 2] (noise floor dB, started flag) or None -> (probabilities [B, k], new state)."""
    x = np.asarray(pcm, dtype=np.float32)
    if x.ndim == 1:
        x = x[None]
    B, k = x.shape[0], x.shape[1] // 512
    st = np.zeros((B, 2), np.float32) if state is None else np.array(state, dtype=np.float32)
    out = np.zeros((B, k), np.float32)
    for b in range(B):
        nf, started = np.float32(st[b, 0]), st[b, 1] != 0
        for f in range(k):
            fr = x[b, f * 512 : (f + 1) * 512].astype(np.float64)
            e = np.float32(10.0) * np.log10(np.float32((fr * fr).sum() * (1.0 / 512.0)) + np.float32(1e-10)).astype(np.float32)
            nf = np.float32(min(e, nf + np.float32(0.02))) if started else np.float32(min(e, np.float32(-40.0)))
            started = True
            p = np.float32(1.0) / (np.float32(1.0) + np.exp(-((e - nf) - np.float32(9.0)) * np.float32(0.5), dtype=np.float32)) \
                if e > np.float32(-60.0) else np.float32(0.0)
            out[b, f] = p
        st[b] = (nf, 1.0 if started else 0.0)
    return out, st


def synth_audio(n: int, seed: int = 0, kind: str = "noise") -> np.ndarray:
    """Seeded float32 mono 16 kHz test clips: gaussian noise (sigma 0.1, clipped), zeros, 440 Hz sine."""
    if kind == "noise":
        x = np.random.default_rng(seed).standard_normal(n).astype(np.float32) * np.float32(0.1)
        return np.clip(x, -1.0, 1.0).astype(np.float32)
    if kind == "zeros":
        return np.zeros(n, dtype=np.float32)
    if kind == "sine":
        t = np.arange(n, dtype=np.float64) / SAMPLE_RATE
        return (0.5 * np.sin(2 * np.pi * 440.0 * t)).astype(np.float32)
    if kind == "speechlike":  # amplitude-modulated band noise: gives non-uniform mel + attention
        rng = np.random.default_rng(seed)
        t = np.arange(n, dtype=np.float64) / SAMPLE_RATE
        env = 0.55 + 0.45 * np.sin(2 * np.pi * 3.1 * t + rng.random() * 6.28)
        car = np.sin(2 * np.pi * (180 + 40 * np.sin(2 * np.pi * 0.7 * t)) * t)
        x = env * (0.3 * car + 0.05 * rng.standard_normal(n))
        return np.clip(x, -1, 1).astype(np.float32)
    raise ValueError(kind)
