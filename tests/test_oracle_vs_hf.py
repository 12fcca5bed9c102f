"""Pins the numpy oracle (oracle/whisper_oracle.py) against the arithmetic the reference actually runs:
the installed Hugging Face Whisper (transformers 5.15.0) on CPU.  CPU-only, a few seconds."""
import json

import numpy as np
import pytest
import torch

from oracle import hf_reference as hr
from oracle import whisper_oracle as wo

torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def micro():
    dims = wo.PRESETS["micro"]
    w = wo.make_weights(dims, 0)
    return dims, w, hr.build_hf_model(dims, w)


@pytest.mark.parametrize("kind", ["noise", "sine", "zeros", "speechlike"])
@pytest.mark.parametrize("n_mels", [80, 128])
def test_logmel_matches_hf_feature_extractor(kind, n_mels):
    from transformers import WhisperFeatureExtractor

    fe = WhisperFeatureExtractor(feature_size=n_mels, chunk_length=10)
    pcm = wo.synth_audio(16000 * 7, 1, kind)  # shorter than the chunk: exercises zero padding
    ref = fe(pcm, sampling_rate=16000, return_tensors="np").input_features
    got = wo.log_mel(pcm, n_mels, 160000)
    assert got.shape == ref.shape == (1, n_mels, 1000)
    # torch.stft is a float32 FFT; the oracle's DFT is float64: agreement at float32 round-off
    assert np.abs(got - ref).max() < 5e-5


def test_mel_filter_bank_matches_hf():
    from transformers.audio_utils import mel_filter_bank

    for n in (80, 128):
        ref = mel_filter_bank(201, n, 0.0, 8000.0, 16000, norm="slaney", mel_scale="slaney")
        assert np.allclose(wo.mel_filter_bank(n), ref, rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("T", [500, 750, 1000])
def test_position_interpolation_matches_patch_hf_model(T):
    """A0: F.interpolate(mode='linear', align_corners=False) as in R:thestage_speechkit/nvidia/asr_pipeline.py:15-27."""
    import torch.nn.functional as F

    pos = wo.sinusoids(1500, 64)
    ref = F.interpolate(torch.from_numpy(pos).unsqueeze(0).transpose(1, 2), size=T, mode="linear", align_corners=False)
    ref = ref.transpose(1, 2).squeeze(0).numpy()
    assert np.abs(wo.interpolate_positions(pos, T) - ref).max() < 1e-6


def test_encoder_and_teacher_forced_logits(micro):
    dims, w, hf = micro
    pcm = wo.synth_audio(480000, 0, "speechlike")
    mel = wo.log_mel(pcm, dims.n_mels)
    om = wo.OracleWhisper(dims, w)
    enc = om.encode(mel)
    ref = hf.model.encoder(torch.from_numpy(mel)).last_hidden_state.numpy()
    assert np.abs(enc - ref).max() < 2e-5
    ids = np.array([[50258, 50259, 50360, 50364, 100, 2000, 31000, 50257]])
    lg, _ = om.decode(ids, om.new_cache(enc))
    ref_lg = hf(input_features=torch.from_numpy(mel), decoder_input_ids=torch.from_numpy(ids)).logits.numpy()
    assert np.abs(lg - ref_lg).max() < 2e-5
    # incremental decoding with the KV cache equals the full pass
    cache = om.new_cache(enc)
    om.decode(ids[:, :4], cache)
    step, _ = om.decode(ids[:, 4:5], cache)
    assert np.abs(step[:, 0] - lg[:, 4]).max() < 2e-5


def test_greedy_timestamp_grammar_and_token_timestamps_match_hf(micro):
    """Inner greedy loop + the three Whisper logits processors + DTW, against HF's generate (first seek iteration)."""
    from transformers.generation.utils import GenerationMixin

    dims, w, hf = micro
    pcm = np.stack([wo.synth_audio(480000, s, k) for s, k in [(0, "speechlike"), (2, "noise")]])
    mel = wo.log_mel(pcm, dims.n_mels)
    calls = []
    orig = GenerationMixin.generate

    def spy(self, *a, **k):
        out = orig(self, *a, **k)
        calls.append(out)
        return out

    GenerationMixin.generate = spy
    try:
        hf.generate(input_features=torch.from_numpy(mel), attention_mask=torch.ones(2, 3000, dtype=torch.long),
                    return_timestamps=True, return_token_timestamps=True, language="en", max_new_tokens=24,
                    num_beams=1, do_sample=False, use_cache=True)
    finally:
        GenerationMixin.generate = orig
    first = calls[0]
    om = wo.OracleWhisper(dims, w)
    heads = [tuple(h) for h in hf.generation_config.alignment_heads]
    opt = wo.GreedyOptions(max_new_tokens=24, timestamps=True, alignment_heads=heads)
    res = wo.greedy_generate(om, om.encode(mel), np.array([[50258, 50259, 50360]] * 2), opt)
    assert np.array_equal(res["sequences"], first["sequences"].numpy())
    ts = wo.token_timestamps(res["cross"], 3, num_frames=[3000, 3000])
    assert np.array_equal(ts, first["token_timestamps"].numpy())


def test_dtw_and_median_filter_match_hf_private_helpers():
    from transformers.models.whisper.generation_whisper import _dynamic_time_warping, _median_filter

    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 3, 9, 40)).astype(np.float32)
    assert np.array_equal(wo.median_filter(x, 7), _median_filter(torch.from_numpy(x), 7).numpy())
    m = rng.standard_normal((13, 57))
    a, b = wo.dtw(m)
    ra, rb = _dynamic_time_warping(m)
    assert np.array_equal(a, ra) and np.array_equal(b, rb)


def test_logits_processor_edge_cases():
    """min_new_tokens masks eos first; the first sampled token must be a timestamp <= max_initial_timestamp_index."""
    opt = wo.GreedyOptions(timestamps=True, min_new_tokens=2)
    s = np.zeros(51866, np.float32)
    s[opt.eos] = 10.0
    out = wo.apply_logits_processors(s, [50258, 50259, 50360], 3, opt)
    assert out[opt.eos] == -np.inf
    ts0 = opt.no_timestamps_id + 1
    assert np.isneginf(out[:ts0]).all() and np.isneginf(out[ts0 + 51 :]).all() and np.isfinite(out[ts0 : ts0 + 51]).all()
    # after "<ts> text": a following timestamp may not go backwards
    out = wo.apply_logits_processors(s, [50258, 50259, 50360, ts0 + 10, 400], 3, opt)
    assert np.isneginf(out[ts0 : ts0 + 11]).all() and np.isfinite(out[ts0 + 11])


@pytest.mark.parametrize("num_frames", [None, 3000, 1067, -1721, [3000, 3000], [-1721, -1721], [-400, -400], [-2999, -2999],
                                        [-3000, -3000], [1067, 2900], [-1721, 900], "t:[-1721]", "t:[-400, -400]", "t:[-400, 700]",
                                        "a:[-1000, -1000]"])
def test_token_timestamp_cropping_matches_hf_for_every_num_frames_flavour(micro, num_frames):
    """HF crops the alignment matrix with Python slices whose effect depends on the TYPE and UNIFORMITY of `num_frames`
    (int: once; equal values: twice - which matters for negative bounds, `num_frames - seek < 0` in a seek iteration past the end
    of a short clip; different values: once per row; nothing left: the empty-matrix DTW).  The restatement reproduces HF's own
    `_extract_token_timestamps` (HF:models/whisper/generation_whisper.py:241-381) bit for bit in every case."""
    import warnings

    dims, w, hf = micro
    rng = np.random.default_rng(5)
    B, Ha, N, T = 2, 2, 9, 1500
    heads = [tuple(h) for h in hf.generation_config.alignment_heads][:Ha]
    if isinstance(num_frames, str):
        kind, lst = num_frames.split(":")
        vals = json.loads(lst)
        nf_hf = torch.tensor(vals) if kind == "t" else np.array(vals)
        nf_or = vals
    else:
        nf_hf = nf_or = num_frames
    if isinstance(nf_or, list) and len(nf_or) == 1:
        B = 1
    probs = rng.random((B, Ha, N, T)).astype(np.float32)
    probs /= probs.sum(-1, keepdims=True)
    # HF takes the cross attentions as a tuple over generation steps of per-layer tensors [B, H, 1, T]
    L, H = dims.dec_layers, dims.heads
    steps = []
    for i in range(N):
        layers = [torch.zeros(B, H, 1, T) for _ in range(L)]
        for a, (l, h) in enumerate(heads):
            layers[l][:, h, 0, :] = torch.from_numpy(probs[:, a, i, :])
        steps.append(tuple(layers))
    seqs = torch.zeros((B, N + 1), dtype=torch.long)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from transformers.generation.utils import GenerateEncoderDecoderOutput

        go = GenerateEncoderDecoderOutput(sequences=seqs, cross_attentions=tuple(steps))
        ref = hf._extract_token_timestamps(go, heads, num_frames=nf_hf, num_input_ids=3)
    got = wo.token_timestamps(probs, 3, nf_or)
    assert np.array_equal(got, ref.numpy()), (num_frames, got[:, :8], ref.numpy()[:, :8])
