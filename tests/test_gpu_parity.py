"""Parity tests proper: the HIP path (through the C ABI) against the numpy oracle and against the committed golden
vectors produced by the reference's own pipeline.  Run on the MI355X box: ``pytest -m gpu``.

Tolerances (stated once):
  * log-mel: max-abs 2e-4 vs the float64 oracle (the reference's own float32 FFT differs from it by ~2e-5).
  * strict-f32 mode: encoder hidden / teacher-forced logits relative-L2 <= 2e-5, greedy ids IDENTICAL,
    alignment rows max-abs 1e-4, token timestamps identical (+-one 0.02 s frame allowed).
  * bf16 mode: relative-L2 <= 3e-2 (measured ~6e-3) and top-1 agreement on every teacher-forced step whose
    oracle top1-top2 margin exceeds 0.25.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import whisper_oracle as wo
from tests.util import PROMPT, clips, dims_variant, make_engine, rel_l2

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")


# ---------------------------------------------------------------- A1
@pytest.mark.parametrize("n_mels", [80, 128])
def test_logmel(n_mels):
    dims = dims_variant("micro", n_mels=n_mels, enc_layers=0, dec_layers=0)
    eng = make_engine(dims, wo.make_weights(dims, 0), T=500, max_batch=4, dtype="f32")
    pcm = clips(160000, ["noise", "sine", "zeros", "speechlike"])
    got = eng.logmel(torch.from_numpy(pcm).cuda(), out_dtype=torch.float32).cpu().numpy()
    assert np.abs(got - wo.log_mel(pcm, n_mels)).max() < 2e-4
    # ragged input: 6.25 s of audio zero-padded to the 10 s chunk, and an empty (all padding) clip
    short = pcm[:, :100000]
    got = eng.logmel(torch.from_numpy(short).cuda(), n_samples=160000, out_dtype=torch.float32).cpu().numpy()
    assert np.abs(got - wo.log_mel(short, n_mels, 160000)).max() < 2e-4
    got = eng.logmel(torch.from_numpy(pcm).cuda(), n_valid=[0, 1, 399, 160000], out_dtype=torch.float32).cpu().numpy()
    ref = np.stack([wo.log_mel(pcm[i, :n], n_mels, 160000)[0] for i, n in enumerate([0, 1, 399, 160000])])
    assert np.abs(got - ref).max() < 2e-4
    eng.close()


def test_logmel_golden_reference_rows():
    z = np.load(os.path.join(GOLD, "micro_c10.npz"))
    meta = json.load(open(os.path.join(GOLD, "pipeline_golden.json")))["micro_c10"]
    dims = dims_variant("micro", enc_layers=0, dec_layers=0)
    eng = make_engine(dims, wo.make_weights(dims, 0), T=500, max_batch=1, dtype="f32")
    clip = wo.synth_audio(16000 * meta["seconds"], meta["seed"], meta["kind"])[:160000]
    got = eng.logmel(torch.from_numpy(clip).cuda(), out_dtype=torch.float32).cpu().numpy()
    assert np.abs(got[0, ::16, ::25] - z["mel_rows"]).max() < 2e-4   # rows produced by the reference's HF feature extractor
    eng.close()


# ---------------------------------------------------------------- A2-A4
ENC_CASES = [
    ("micro", 100, 2, "f32", 2, 2e-5, {}), ("micro", 500, 3, "f32", 2, 2e-5, {}), ("micro80", 100, 2, "f32", 2, 2e-5, {}),
    ("micro", 100, 2, "bf16", 2, 3e-2, {}), ("micro", 500, 4, "bf16", 2, 3e-2, {}),
    ("large-v3", 500, 1, "f32", 1, 2e-5, {}), ("large-v3", 500, 4, "bf16", 1, 3e-2, {}),
    ("tiny.en", 1500, 1, "f32", 4, 5e-5, {}), ("micro", 100, 1, "f32", 0, 2e-5, {}),
    # chunk lengths whose frame count is no multiple of the 64-key tiles: 1 s (less than one tile), 25 s, 29 s
    ("micro", 50, 2, "f32", 2, 2e-5, {}), ("micro", 1250, 1, "f32", 2, 2e-5, {}), ("micro", 1450, 2, "bf16", 2, 3e-2, {}),
    # float16 contexts (round 4; the reference's streaming default dtype): 10 mantissa bits, bounds 1/8 of bf16's
    ("micro", 100, 2, "f16", 2, 4e-3, {}), ("micro", 500, 4, "f16", 2, 4e-3, {}), ("large-v3", 500, 4, "f16", 1, 4e-3, {}),
    ("micro", 1450, 2, "f16", 2, 4e-3, {}), ("large-v3", 500, 1, "f16", 1, 4e-3, {}),
    # 20 s chunks (T = 1000; R:README.md:49 advertises 10 / 15 / 20 / 30 s engines), every element type, micro and the real width
    ("micro", 1000, 2, "f32", 2, 2e-5, {}), ("micro", 1000, 3, "bf16", 2, 3e-2, {}), ("micro", 1000, 2, "f16", 2, 4e-3, {}),
    ("large-v3", 1000, 1, "f32", 1, 2e-5, {}), ("large-v3", 1000, 2, "bf16", 1, 3e-2, {}), ("large-v3", 1000, 2, "f16", 1, 4e-3, {}),
]


@pytest.mark.parametrize("preset,T,B,dtype,layers,tol,over", ENC_CASES)
def test_encoder(preset, T, B, dtype, layers, tol, over):
    dims = dims_variant(preset, enc_layers=layers, dec_layers=0, **over)
    w = wo.make_weights(dims, 1)
    eng = make_engine(dims, w, T=T, max_batch=B, dtype=dtype)
    mel = wo.log_mel(clips(T * 320, B), dims.n_mels)
    ref = wo.OracleWhisper(dims, w, T=T).encode(mel)
    got = eng.encode(torch.from_numpy(mel).cuda(), return_hidden=True).cpu().numpy()
    assert np.isfinite(got).all() and rel_l2(got, ref) < tol
    eng.close()


def test_encoder_golden_reference_rows():
    z = np.load(os.path.join(GOLD, "micro80_c30.npz"))
    meta = json.load(open(os.path.join(GOLD, "pipeline_golden.json")))["micro80_c30"]
    dims = dims_variant("micro80", dec_layers=0)
    w = {k: v for k, v in wo.make_weights(wo.PRESETS["micro80"], 0).items() if "decoder.layers" not in k}
    eng = make_engine(dims, w, T=1500, max_batch=1, dtype="f32")
    clip = wo.synth_audio(16000 * meta["seconds"], meta["seed"], meta["kind"])[:480000]
    mel = eng.logmel(torch.from_numpy(clip).cuda(), out_dtype=torch.float32)
    got = eng.encode(mel, return_hidden=True).cpu().numpy()
    assert np.abs(got[0, ::50, ::8] - z["enc_rows"]).max() < 1e-4    # rows of the reference's HF encoder output
    eng.close()


# ---------------------------------------------------------------- A5-A8
DEC_CASES = [
    ("micro", 100, 3, "f32", 2, 2e-5), ("micro", 500, 1, "f32", 2, 2e-5), ("micro", 100, 8, "f32", 2, 2e-5),
    ("micro", 100, 16, "bf16", 2, 3e-2), ("large-v3", 500, 1, "f32", 1, 2e-5), ("large-v3", 500, 2, "bf16", 1, 3e-2),
    ("large-v3", 500, 16, "bf16", 1, 3e-2), ("tiny.en", 1500, 5, "f32", 4, 5e-5), ("micro", 100, 2, "f32", 0, 2e-5),
    ("micro", 750, 2, "f32", 2, 2e-5), ("micro", 750, 3, "bf16", 2, 3e-2),   # 15 s chunks: two key chunks, second one partial
    ("micro", 100, 17, "f32", 2, 2e-5), ("micro", 100, 40, "bf16", 2, 3e-2), ("micro", 100, 64, "f32", 1, 2e-5),  # > 16 streams: groups of 16
    ("large-v3", 500, 32, "bf16", 1, 3e-2),
    # round 5: the operand-ring projection kernels (more than 16 streams at the real width: K = 1280 / 5120 - the micro presets never
    # reach them) against the oracle: two and four groups of 16 streams, every element type (the strict-f32 rings to 2e-5)
    ("large-v3", 100, 64, "f32", 1, 2e-5), ("large-v3", 100, 24, "f32", 1, 2e-5), ("large-v3", 100, 64, "bf16", 1, 3e-2),
    ("large-v3", 100, 40, "f16", 1, 4e-3),
    ("micro", 50, 2, "f32", 2, 2e-5), ("micro", 1250, 2, "f32", 2, 2e-5), ("micro", 1450, 3, "bf16", 2, 3e-2),   # ragged last key tile
    # float16 contexts
    ("micro", 100, 16, "f16", 2, 4e-3), ("large-v3", 500, 2, "f16", 1, 4e-3), ("large-v3", 500, 16, "f16", 1, 4e-3),
    ("micro", 750, 3, "f16", 2, 4e-3), ("micro", 100, 40, "f16", 2, 4e-3), ("micro", 1450, 3, "f16", 2, 4e-3),
    # 20 s chunks (T = 1000 keys: 15 full 64-key tiles + 40): the cross attention's two-chunk path with a ragged tail
    ("micro", 1000, 2, "f32", 2, 2e-5), ("micro", 1000, 3, "bf16", 2, 3e-2), ("micro", 1000, 17, "f16", 2, 4e-3),
    ("large-v3", 1000, 1, "f32", 1, 2e-5), ("large-v3", 1000, 16, "bf16", 1, 3e-2), ("large-v3", 1000, 4, "f16", 1, 4e-3),
]


@pytest.mark.parametrize("preset,T,B,dtype,layers,tol", DEC_CASES)
def test_decoder_teacher_forced(preset, T, B, dtype, layers, tol):
    dims = dims_variant(preset, enc_layers=1, dec_layers=layers)
    w = wo.make_weights(dims, 2)
    eng = make_engine(dims, w, T=T, max_batch=B, dtype=dtype)
    mel = wo.log_mel(clips(T * 320, B), dims.n_mels)
    om = wo.OracleWhisper(dims, w, T=T)
    enc = om.encode(mel)
    eng.encode(torch.from_numpy(mel).cuda())
    eng.cross_kv(B)
    eng.decoder_reset(B)
    ids = np.concatenate([np.tile(np.array(PROMPT), (B, 1)), np.random.default_rng(3).integers(0, 50000, size=(B, 6))], axis=1)
    cache = om.new_cache(enc)
    for s in range(ids.shape[1]):
        ref, _ = om.decode(ids[:, s : s + 1], cache)
        ref = ref[:, 0]
        got = eng.decode_step(ids[:, s].tolist()).cpu().numpy()
        assert rel_l2(got, ref) < tol
        if dtype in ("bf16", "f16"):  # top-1 agreement wherever the oracle's margin is clear
            srt = np.sort(ref, axis=-1)
            clear = (srt[:, -1] - srt[:, -2]) > (0.25 if dtype == "bf16" else 0.03)
            assert np.array_equal(got.argmax(-1)[clear], ref.argmax(-1)[clear])
    eng.close()


_PLAIN_SEQUENCE = r"""
import sys, numpy as np, torch
from oracle import whisper_oracle as wo
from tests.util import PROMPT, clips, dims_variant, make_engine
preset, T, B, dtype, layers, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), sys.argv[6]
dims = dims_variant(preset, enc_layers=1, dec_layers=layers)
eng = make_engine(dims, wo.make_weights(dims, 2), T=T, max_batch=B, dtype=dtype)
eng.encode(torch.from_numpy(wo.log_mel(clips(T * 320, B), dims.n_mels)).cuda())
eng.cross_kv(B); eng.decoder_reset(B)
ids = np.concatenate([np.tile(np.array(PROMPT), (B, 1)), np.random.default_rng(3).integers(0, 50000, size=(B, 3))], axis=1)
np.save(out, np.stack([eng.decode_step(ids[:, s].tolist()).cpu().numpy() for s in range(ids.shape[1])]))
"""


@pytest.mark.parametrize("preset,T,B,dtype,layers,tol", [("micro", 100, 3, "f32", 2, 2e-5), ("tiny.en", 500, 20, "f32", 4, 5e-5),
                                                         ("large-v3", 500, 16, "bf16", 2, 1.5e-2)])
def test_cross_query_ahead_matches_the_plain_launch_sequence(preset, T, B, dtype, layers, tol, tmp_path):
    """decode_core accumulates the cross-attention query ahead of its LayerNorm (x0 W'^T in the QKV launch, attn (W' Wo)^T in
    the out-projection launch, LayerNorm applied inside the cross attention: DESIGN.md section 4).  TW_FUSE_CQ=0 runs the
    plain 8-launch sequence on the same context layout; the logits of both agree to rounding (f32) / to the bf16 rounding of
    W' Wo and of the un-normalised accumulator (bf16).  The plain sequence runs in a child process (the switch is read once)."""
    import subprocess
    import sys
    out = str(tmp_path / "plain.npy")
    env = dict(os.environ, TW_FUSE_CQ="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, "-c", _PLAIN_SEQUENCE, preset, str(T), str(B), dtype, str(layers), out], check=True, env=env, cwd=root)
    plain = np.load(out)
    dims = dims_variant(preset, enc_layers=1, dec_layers=layers)
    eng = make_engine(dims, wo.make_weights(dims, 2), T=T, max_batch=B, dtype=dtype)
    eng.encode(torch.from_numpy(wo.log_mel(clips(T * 320, B), dims.n_mels)).cuda())
    eng.cross_kv(B)
    eng.decoder_reset(B)
    ids = np.concatenate([np.tile(np.array(PROMPT), (B, 1)), np.random.default_rng(3).integers(0, 50000, size=(B, 3))], axis=1)
    for s in range(ids.shape[1]):
        got = eng.decode_step(ids[:, s].tolist()).cpu().numpy()
        assert rel_l2(got, plain[s]) < tol, (s, rel_l2(got, plain[s]))
    eng.close()


FP8_CASES = [("micro", 100, 3, 0), ("micro", 100, 3, 1), ("micro", 100, 16, 2), ("large-v3", 500, 2, 1), ("tiny.en", 1500, 1, 2),
             ("micro", 50, 2, 1), ("micro", 1250, 2, 2)]   # fp8 cross-K/V: less than one 64-key group, three groups with a ragged last one


@pytest.mark.parametrize("flavour", ["fp8a8", "fp8a16"])
@pytest.mark.parametrize("preset,T,B,layers", FP8_CASES)
def test_decoder_mxfp8_teacher_forced(preset, T, B, layers, flavour):
    """BASELINE config 5: decoder projection weights in MXFP8 on v_mfma_scale_f32_16x16x128_f8f6f4.  Checked against the
    numpy restatement of the same quantised arithmetic and against the unquantised f32 oracle (fp8 noise budget).
    Quantisation is chaotic: a 1-ulp bf16 difference upstream (fp32 summation order in the cross-K/V GEMM or a softmax)
    moves a value across an e4m3 rounding boundary and changes it by 6 %, so only the first token through at most one
    decoder layer reproduces the restatement to rounding error; deeper/later comparisons share the quantised weights but
    decorrelate in the activations and are held to the fp8 noise level instead.
    flavour "fp8a16" (TW_BF16_W8A16): the same quantised weights widened to bf16 in registers against UNQUANTISED activations
    on the bf16 MFMA; its restatement is `OracleWhisperMXFP8(act_quant=False)`.  Only the e4m3 cross-K/V remain chaotic there, so
    the engine must be at least as close to exact arithmetic as W8A8 is allowed to be, and closer to its restatement."""
    a16 = flavour == "fp8a16"
    dims = dims_variant(preset, enc_layers=1, dec_layers=layers)
    w = wo.make_weights(dims, 2)
    eng = make_engine(dims, w, T=T, max_batch=B, dtype=flavour)
    mel = wo.log_mel(clips(T * 320, B), dims.n_mels)
    om = wo.OracleWhisper(dims, w, T=T)
    oq = wo.OracleWhisperMXFP8(dims, w, T=T, act_quant=not a16)
    # the decoder is compared on the ENGINE's encoder states (bf16): see the docstring
    enc = eng.encode(torch.from_numpy(mel).cuda(), return_hidden=True).cpu().numpy()
    eng.cross_kv(B)
    eng.decoder_reset(B)
    ids = np.concatenate([np.tile(np.array(PROMPT), (B, 1)), np.random.default_rng(3).integers(0, 50000, size=(B, 3))], axis=1)
    cache, cache_q = om.new_cache(enc), oq.new_cache(enc)
    for s in range(ids.shape[1]):
        ref = om.decode(ids[:, s : s + 1], cache)[0][:, 0]
        refq = oq.decode(ids[:, s : s + 1], cache_q)[0][:, 0]
        got = eng.decode_step(ids[:, s].tolist()).cpu().numpy()
        if preset == "micro" and layers <= 1 and s == 0:
            # layers = 0 (embedding + logits): bit-level agreement of layouts, scales, roundings.  With one decoder layer the
            # fp8 cross-K/V and the fp8 activation operands come in: the engine's and the restatement's bf16 values differ by
            # 1 ulp wherever the fp32 summation order of a projection differs, and such a value can land on the other side of
            # an e4m3 rounding boundary (a 6 % step for that element).  Perturbing 5 % of the attention outputs of the
            # restatement by one bf16 ulp moves its logits by 1-2e-2 (tools/dbg/dbg_fuse_fp8.py); measured here 2.2e-2
            # (7e-3 with TW_FUSE_CQ=0 against the plain order), which still pins layouts and scale placement (a wrong scale
            # or a permuted fragment is an error of order 1) ...
            assert rel_l2(got, refq) < (2e-3 if layers == 0 else 3e-2), rel_l2(got, refq)
            if layers == 1:
                # ... and the restatement of the engine's order of operations ("cross query ahead") is the closer one:
                # against the plain order (LayerNorm, then the quantised query weight) the same logits are 4e-2 away
                op = wo.OracleWhisperMXFP8(dims, w, T=T, cross_q_ahead=False, act_quant=not a16)
                refp = op.decode(ids[:, :1], op.new_cache(enc))[0][:, 0]
                assert rel_l2(got, refq) < rel_l2(got, refp), (rel_l2(got, refq), rel_l2(got, refp))
        assert rel_l2(got, refq) < (4e-2 if a16 else 6e-2), (s, rel_l2(got, refq))
        assert rel_l2(got, ref) < (8e-2 if a16 else 1e-1), (s, rel_l2(got, ref))       # fp8 quantisation noise vs exact arithmetic
        assert rel_l2(refq, ref) > (5e-3 if a16 else 1e-2)                             # ... which the restatement does model
        srt = np.sort(ref, axis=-1)
        clear = (srt[:, -1] - srt[:, -2]) > 1.0
        assert np.array_equal(got.argmax(-1)[clear], ref.argmax(-1)[clear])
    eng.close()


def test_w8a16_groups_of_streams_match_single_group():
    """TW_BF16_W8A16 is the fp8 flavour that takes more than 16 streams (2 / 4 groups of 16 per weight pass, k_decode.hip CG):
    40 streams holding 40 different clips - teacher-forced logits of every stream equal (to summation-order noise) what a
    16-stream context computes for the same clip, whichever group and lane the stream sits in."""
    dims = dims_variant("micro", enc_layers=1, dec_layers=2)
    w = wo.make_weights(dims, 5)
    T, B = 100, 40
    pcm = clips(T * 320, B)
    ids = np.concatenate([np.tile(np.array(PROMPT), (B, 1)), np.random.default_rng(7).integers(0, 50000, size=(B, 4))], axis=1)

    def run(eng, sel):
        n = len(sel)
        eng.encode(eng.logmel(torch.from_numpy(pcm[sel]).cuda()))
        eng.cross_kv(n)
        eng.decoder_reset(n)
        return np.stack([eng.decode_step(ids[sel, s].tolist()).cpu().numpy() for s in range(ids.shape[1])], axis=1)

    big = make_engine(dims, w, T=T, max_batch=B, dtype="fp8a16")
    got = run(big, np.arange(B))
    big.close()
    small = make_engine(dims, w, T=T, max_batch=16, dtype="fp8a16")
    for lo in (0, 16, 32):
        sel = np.arange(lo, min(lo + 16, B))
        ref = run(small, sel)
        assert rel_l2(got[sel], ref) < 1e-2, (lo, rel_l2(got[sel], ref))   # (fc2 splits K over 8 instead of 16 wavefronts above 16 streams: bf16 rounding flips)
    small.close()
    with pytest.raises(RuntimeError, match="max_batch"):          # W8A8 (activation quantisation in registers) stays at one group
        make_engine(dims, w, T=T, max_batch=B, dtype="fp8a8")


def test_the_greedy_loop_on_its_own_stream_gives_the_callers_stream_results(monkeypatch):
    """`WhisperEngine.generate_greedy` runs the loop on a stream confined to 160 compute units when the caller manages no streams
    (engine.py: _decode_stream; THEWHISPER_DECODE_CUS=0 keeps the caller's stream).  Same kernels, ordered behind the encoder stage
    by an event: ids, lengths and token timestamps are identical, also when calls alternate with encoder stages of new audio (the
    ordering between the two streams is what could go wrong) and for a forced-prefix call (which keeps the caller's stream)."""
    dims = dims_variant("micro", enc_layers=2, dec_layers=2)
    w = wo.make_weights(dims, 21)
    T, B = 150, 3
    heads = [(0, 1), (1, 0)]
    kinds = (["speechlike", "noise", "sine"], ["noise", "sine", "speechlike"], ["sine", "speechlike", "zeros"])
    mels = [torch.from_numpy(wo.log_mel(clips(T * 320, k), dims.n_mels)).cuda() for k in kinds]
    prompt = np.tile(np.array(PROMPT), (B, 1))

    def run():
        eng = make_engine(dims, w, T=T, max_batch=B, dtype="bf16", heads=heads, use_graph=True)
        out = []
        for mel in mels:
            eng.encode(mel)
            eng.cross_kv(B)
            g = eng.generate_greedy(prompt, max_new_tokens=24, timestamps=True, want_alignment=True)
            ts = eng.token_timestamps(B, prompt.shape[1], g["length"], [2 * T] * B)
            out.append((g["sequences"].copy(), ts.copy()))
        forced = np.concatenate([prompt, out[-1][0][:, prompt.shape[1] : prompt.shape[1] + 4].astype(np.int32)], axis=1)
        if not (forced[:, prompt.shape[1] :] == 50257).any():
            g = eng.generate_greedy(forced, max_new_tokens=24, timestamps=True, want_alignment=True, n_forced=4)
            out.append((g["sequences"].copy(), None))
        used = eng.__dict__.get("_dec_stream")
        eng.close()
        return out, used

    monkeypatch.setenv("THEWHISPER_DECODE_CUS", "0")
    plain, used0 = run()
    monkeypatch.setenv("THEWHISPER_DECODE_CUS", "160")
    masked, used1 = run()
    assert used0 is None and used1 is not None
    assert len(plain) == len(masked)
    for (a, ta), (b, tb) in zip(plain, masked):
        assert np.array_equal(a, b)
        assert (ta is None and tb is None) or np.array_equal(ta, tb)


@pytest.mark.parametrize("dtype", ["bf16", "fp8a16"])
def test_sibling_context_and_adopted_cross_kv_are_bit_identical(dtype):
    """tw_create_sibling + tw_adopt_cross_kv (the serving loop's prefetch path): clips encoded by a sibling context (shared
    weights, own workspace) on a CU-masked stream and adopted into slots 1.. of the main context decode to EXACTLY the logits the
    main context computes when it encodes them itself - same kernels, same weights."""
    from thewhisper_amd.overlap import masked_stream

    dims = dims_variant("micro", enc_layers=2, dec_layers=2)
    w = wo.make_weights(dims, 9)
    T, B = 100, 5
    mel = torch.from_numpy(wo.log_mel(clips(T * 320, B), dims.n_mels)).cuda()
    ids = np.concatenate([np.tile(np.array(PROMPT), (B, 1)), np.random.default_rng(1).integers(0, 50000, size=(B, 3))], axis=1)
    eng = make_engine(dims, w, T=T, max_batch=B, dtype=dtype)

    def logits():
        eng.decoder_reset(B)
        return np.stack([eng.decode_step(ids[:, s].tolist()).cpu().numpy() for s in range(ids.shape[1])], axis=1)

    eng.encode(mel)
    eng.cross_kv(B)
    ref = logits()
    side = eng.sibling(max_batch=4)
    n_cus = torch.cuda.get_device_properties(0).multi_processor_count
    side.raw_stream = masked_stream(n_cus - 64, n_cus, n_cus, 0)
    side.encode(mel[1:3])                      # clips 1, 2 -> sibling slots 0, 1
    side.encode(mel[3:5], slot0=2)             # clips 3, 4 -> sibling slots 2, 3
    side.cross_kv(2)
    side.cross_kv(2, slot0=2)
    eng.encode(mel[:1])                        # clip 0 the ordinary way ...
    eng.cross_kv(1)
    eng.adopt_cross_kv(side, 0, 3, 1)          # ... sibling slots 0..2 -> main slots 1..3
    eng.adopt_cross_kv(side, 3, 1, 4)          # ... sibling slot 3 -> main slot 4
    got = logits()
    assert np.array_equal(got, ref)
    with pytest.raises(RuntimeError, match="tw_adopt_cross_kv"):
        eng.adopt_cross_kv(side, 3, 2, 0)      # the sibling holds 4 clips
    with pytest.raises(RuntimeError, match="shares another context's weights"):
        side.load_weight("model.encoder.conv1.bias", torch.zeros(dims.d_model))
    from thewhisper_amd.overlap import _hiplib
    st, side.raw_stream = side.raw_stream, None
    _hiplib().stream_destroy(st)
    eng.close()                                # closes the sibling first
    assert side.ctx is None


def test_mxfp8_greedy_generation_runs_and_is_deterministic():
    """The fp8 context runs the whole path (graph replay, sampler, alignment, DTW); ids are deterministic across calls and
    batch positions (there is no reference for fp8 token ids: with random weights they legitimately differ from bf16)."""
    dims = wo.PRESETS["micro"]
    w = wo.make_weights(dims, 0)
    heads = [(dims.dec_layers - 1, 0), (dims.dec_layers - 1, 1)]
    eng = make_engine(dims, w, T=750, max_batch=4, dtype="fp8", heads=heads, use_graph=True)
    pcm = np.tile(clips(750 * 320, 1), (4, 1))
    eng.encode(eng.logmel(torch.from_numpy(pcm).cuda()))
    eng.cross_kv(4)
    prompt = np.tile(np.array(PROMPT, dtype=np.int32), (4, 1))
    a = eng.generate_greedy(prompt, max_new_tokens=20, min_new_tokens=20, timestamps=True, want_alignment=True)
    b = eng.generate_greedy(prompt, max_new_tokens=20, min_new_tokens=20, timestamps=True, want_alignment=True)
    assert np.array_equal(a["sequences"], b["sequences"])
    assert all(np.array_equal(a["sequences"][0], a["sequences"][i]) for i in range(1, 4))   # same clip in every stream
    L = int(a["length"])
    ts = eng.token_timestamps(4, 3, L, [1500] * 4)
    assert np.all(np.diff(ts[:, 3:L], axis=1) >= 0) and ts.max() <= 15.0 + 1e-6
    eng.close()


def test_logits_golden_reference_topk():
    z = np.load(os.path.join(GOLD, "micro_c10.npz"))
    meta = json.load(open(os.path.join(GOLD, "pipeline_golden.json")))["micro_c10"]
    dims = wo.PRESETS["micro"]
    eng = make_engine(dims, wo.make_weights(dims, 0), T=500, max_batch=1, dtype="f32")
    clip = wo.synth_audio(16000 * meta["seconds"], meta["seed"], meta["kind"])[:160000]
    eng.encode(eng.logmel(torch.from_numpy(clip).cuda(), out_dtype=torch.float32))
    eng.cross_kv(1)
    eng.decoder_reset(1)
    for j, tok in enumerate(z["teacher_ids"][0].tolist()):
        lg = eng.decode_step([tok]).cpu().numpy()[0]
        top = np.argsort(-lg)[:8]
        assert np.array_equal(top, z["logits_top_idx"][j])                 # the reference model's top-8 ids
        assert np.abs(lg[top] - z["logits_top"][j]).max() < 1e-4
    eng.close()


# ---------------------------------------------------------------- A9-A11
GREEDY_CASES = [("micro", 100, 1, 24, False, 0), ("micro", 100, 3, 24, True, 0), ("micro", 500, 2, 40, True, 40),
                ("micro80", 100, 2, 24, False, 0), ("micro", 100, 16, 20, True, 0), ("micro", 750, 2, 24, True, 0),
                ("micro", 100, 33, 12, True, 0),
                # 200 positions: the step graphs of the 64 / 128 / 192 / 256-key self-attention buckets (1, 2 and 4 wavefronts)
                ("micro", 100, 2, 200, True, 200), ("micro", 100, 3, 150, False, 150),
                # the maxima of the boundary: all 448 decoder positions (every self-attention bucket up to 448 keys; the loop must stop at
                # max_length whatever max_new_tokens says), and the 64 streams a context can hold
                ("micro", 100, 2, 500, True, 500), ("micro", 100, 64, 6, True, 0),
                # the largest word-timestamp problem of the path: 445 tokens x 1500 frames (30 s chunk), one row with a negative frame bound
                ("micro", 1500, 2, 500, True, 500),
                # 20 s chunks (R:README.md:49): ids, alignment rows and word timestamps at T = 1000
                ("micro", 1000, 2, 40, True, 40)]


@pytest.mark.parametrize("preset,T,B,max_new,graph,min_new", GREEDY_CASES)
def test_greedy_ids_alignment_and_timestamps(preset, T, B, max_new, graph, min_new):
    dims = wo.PRESETS[preset]
    w = wo.make_weights(dims, 0)
    heads = [(dims.dec_layers - 1, 0), (dims.dec_layers - 1, 1)]
    eng = make_engine(dims, w, T=T, max_batch=B, dtype="f32", heads=heads, use_graph=graph)
    pcm = clips(T * 320, B)
    eng.encode(eng.logmel(torch.from_numpy(pcm).cuda(), out_dtype=torch.float32))
    eng.cross_kv(B)
    prompt = np.tile(np.array(PROMPT, dtype=np.int32), (B, 1))
    out = eng.generate_greedy(prompt, max_new_tokens=max_new, min_new_tokens=min_new, timestamps=True, want_alignment=True)
    om = wo.OracleWhisper(dims, w, T=T)
    opt = wo.GreedyOptions(max_new_tokens=max_new, min_new_tokens=min_new, timestamps=True, alignment_heads=heads)
    ref = wo.greedy_generate(om, om.encode(wo.log_mel(pcm, dims.n_mels)), prompt, opt)
    assert np.array_equal(out["sequences"], ref["sequences"])              # greedy token ids identical
    L = out["length"]
    al = eng.get_alignment(B, L - 1)
    assert np.abs(al - ref["cross"]).max() < 1e-4
    assert np.abs(al.sum(-1) - 1.0).max() < 1e-4                            # softmax rows
    nf = [max(2 * T - 8 * i, 40) for i in range(B)]                          # ragged valid-frame counts
    if B >= 2:
        nf[1] = -(T // 2) - 1                                              # negative bound: HF slices from the end (seek loop)
    ts = eng.token_timestamps(B, 3, L, nf)
    assert np.abs(ts - wo.token_timestamps(ref["cross"], 3, nf)).max() <= 0.0201
    # a bound that leaves NO column (HF's double crop of a negative uniform bound ends here, shortform.hf_kept_columns): the
    # reference still runs its DTW, on the empty matrix - every generated token at -1 frame; and bounds of exactly one / all columns
    nf2 = [(-2 * T - 10, 2, 2 * T + 50)[i % 3] for i in range(B)]
    want = wo.token_timestamps(ref["cross"], 3, None, columns=[(0, 1, T)[i % 3] for i in range(B)])
    ts2 = eng.token_timestamps(B, 3, L, nf2)
    assert np.array_equal(ts2[0], want[0]) and (ts2[0, 3:] == np.float32(-0.02)).all() and (ts2[0, :3] == 0).all()
    assert np.abs(ts2 - want).max() <= 0.0201
    eng.close()


FORCED_CASES = [("micro", 100, 1, 60, 37, "f32", True), ("micro", 100, 1, 150, 120, "f32", False), ("micro", 500, 3, 40, 25, "f32", True),
                ("micro", 100, 16, 30, 9, "f32", True), ("micro", 100, 1, 60, 37, "bf16", True), ("micro", 750, 2, 40, 30, "fp8a16", True),
                ("micro", 100, 2, 24, 1, "f32", True), ("micro", 100, 5, 90, 70, "bf16", False), ("micro", 500, 3, 40, 25, "f16", True)]


@pytest.mark.parametrize("preset,T,B,max_new,n_forced,dtype,graph", FORCED_CASES)
def test_forced_prefix_prefill_continues_the_same_generation(preset, T, B, max_new, n_forced, dtype, graph):
    """tw_greedy_opts::n_forced: the first `n_forced` tokens of a finished generation handed back as
    forced output - processed by the batched prefill (rows = streams x positions per launch, up to 64 rows: several launches for
    the long cases) - continue into EXACTLY the same sequence, alignment rows and token timestamps as the step-by-step run:
    strict f32 against the oracle (whose `begin_index` restates the semantics: processors and the token budget count from the
    real prompt) and against the engine's own unforced run in every dtype."""
    dims = wo.PRESETS[preset]
    w = wo.make_weights(dims, 0)
    heads = [(dims.dec_layers - 1, 0), (dims.dec_layers - 1, 1)]
    eng = make_engine(dims, w, T=T, max_batch=B, dtype=dtype, heads=heads, use_graph=graph)
    pcm = clips(T * 320, B)
    eng.encode(eng.logmel(torch.from_numpy(pcm).cuda(), out_dtype=torch.float32))
    eng.cross_kv(B)
    prompt = np.tile(np.array(PROMPT, dtype=np.int32), (B, 1))
    kw = dict(max_new_tokens=max_new, min_new_tokens=max_new, timestamps=True, want_alignment=True)
    full = eng.generate_greedy(prompt, **kw)
    L = full["length"]
    al_full = eng.get_alignment(B, L - 1)
    nf = [2 * T] * B
    ts_full = eng.token_timestamps(B, 3, L, nf)
    forced = full["sequences"][:, : 3 + n_forced].astype(np.int32)
    out = eng.generate_greedy(forced, n_forced=n_forced, **kw)
    al = eng.get_alignment(B, L - 1)
    ts = eng.token_timestamps(B, 3, L, nf)
    if dtype == "f32":
        om = wo.OracleWhisper(dims, w, T=T)
        opt = wo.GreedyOptions(max_new_tokens=max_new, min_new_tokens=max_new, timestamps=True, alignment_heads=heads)
        ref = wo.greedy_generate(om, om.encode(wo.log_mel(pcm, dims.n_mels)), forced, opt, begin_index=3)
        assert np.array_equal(out["sequences"], ref["sequences"])
        assert np.abs(al - ref["cross"]).max() < 1e-4
        assert np.array_equal(out["sequences"], full["sequences"]) and out["length"] == L
        assert np.abs(al - al_full).max() < 1e-5 and np.array_equal(ts, ts_full)
    else:
        # reduced precision (round 6): the prefill launches hold 16-64 rows where a step holds B - other instantiations of the
        # projection kernel - but every instantiation sums K in ONE order (k_decode.hip), so the forced call reproduces the
        # step-by-step call bit for bit here too: ids, alignment rows, timestamps
        assert np.array_equal(out["sequences"], full["sequences"]) and out["length"] == L
        assert np.array_equal(al, al_full) and np.array_equal(ts, ts_full)
    # errors: more forced tokens than prompt, forced <eos>, nothing left to generate
    with pytest.raises(RuntimeError, match="n_forced"):
        eng.generate_greedy(forced, n_forced=forced.shape[1], **kw)
    bad = forced.copy(); bad[0, -1] = 50257
    with pytest.raises(RuntimeError, match="forced token"):
        eng.generate_greedy(bad, n_forced=max(1, n_forced), **kw)
    with pytest.raises(RuntimeError, match="nothing to generate"):
        eng.generate_greedy(full["sequences"][:, :L].astype(np.int32), n_forced=L - 3, **kw)
    eng.close()


@pytest.mark.parametrize("timestamps", [False, True])
def test_suppress_lists_match_oracle(timestamps):
    """SuppressTokens / SuppressTokensAtBegin (HF:generation/logits_process.py:1816-1906): a suppress list that contains the
    tokens an unconstrained run picks, so the mask has to change the result, plus ids at the bitmap word edges."""
    dims = wo.PRESETS["micro"]
    w = wo.make_weights(dims, 0)
    eng = make_engine(dims, w, T=100, max_batch=2, dtype="f32", use_graph=True)
    pcm = clips(100 * 320, 2)
    mel = wo.log_mel(pcm, dims.n_mels)
    om = wo.OracleWhisper(dims, w, T=100)
    enc = om.encode(mel)
    prompt = np.tile(np.array(PROMPT, dtype=np.int32), (2, 1))
    free = wo.greedy_generate(om, enc, prompt, wo.GreedyOptions(max_new_tokens=12, timestamps=timestamps))
    picked = sorted({int(t) for t in free["sequences"][:, prompt.shape[1]:].ravel() if t < 50257})
    sup = tuple(picked[: max(1, len(picked) // 2)] + [0, 31, 32, 63, 64, 50256, 1234])
    begin = (220, 50257) + tuple(picked[-1:])
    opt = wo.GreedyOptions(max_new_tokens=12, timestamps=timestamps, suppress=sup, begin_suppress=begin)
    ref = wo.greedy_generate(om, enc, prompt, opt)
    assert not np.array_equal(ref["sequences"], free["sequences"])
    eng.encode(torch.from_numpy(mel).cuda())
    eng.cross_kv(2)
    for _ in range(2):  # second call replays the captured step graph with the same bitmap buffer
        out = eng.generate_greedy(prompt, max_new_tokens=12, timestamps=timestamps, suppress=sup, begin_suppress=begin)
        assert np.array_equal(out["sequences"], ref["sequences"])
    out = eng.generate_greedy(prompt, max_new_tokens=12, timestamps=timestamps)  # list removed again: bitmap is rebuilt per call
    assert np.array_equal(out["sequences"], free["sequences"])
    eng.close()


def test_eos_and_padding_semantics():
    """A row that emits eos keeps receiving pad while the others continue (HF:generation/utils.py:2929-2936)."""
    dims = wo.PRESETS["micro"]
    w = wo.make_weights(dims, 0)
    # bias the tied embedding so that eos wins quickly on some rows but not all
    w = dict(w)
    emb = w["model.decoder.embed_tokens.weight"].copy()
    emb[50257] *= 6.0
    w["model.decoder.embed_tokens.weight"] = emb
    eng = make_engine(dims, w, T=100, max_batch=4, dtype="f32")
    pcm = clips(32000, 4)
    eng.encode(eng.logmel(torch.from_numpy(pcm).cuda(), out_dtype=torch.float32))
    eng.cross_kv(4)
    prompt = np.tile(np.array(PROMPT + [50364], dtype=np.int32), (4, 1))
    out = eng.generate_greedy(prompt, max_new_tokens=30, timestamps=False)
    om = wo.OracleWhisper(dims, w, T=100)
    ref = wo.greedy_generate(om, om.encode(wo.log_mel(pcm, dims.n_mels)), prompt, wo.GreedyOptions(max_new_tokens=30))
    assert np.array_equal(out["sequences"], ref["sequences"])
    eng.close()


# ---------------------------------------------------------------- end to end through the drop-in API
def _assert_same_transcript(out, gold, ts_tol):
    """Text and chunk texts identical; timestamps identical (ts_tol = 0) or within one 0.02 s alignment frame."""
    assert out["text"] == gold["text"]
    if "chunks" not in gold:
        assert "chunks" not in out
        return 0.0
    assert [c["text"] for c in out["chunks"]] == [c["text"] for c in gold["chunks"]]
    worst = 0.0
    for a, b in zip(out["chunks"], gold["chunks"]):
        for x, y in zip(a["timestamp"], b["timestamp"]):
            assert (x is None) == (y is None)
            if x is not None:
                worst = max(worst, abs(x - y))
    assert worst <= ts_tol, f"timestamp deviation {worst}"
    return worst


@pytest.mark.parametrize("name", ["micro_c10", "micro80_c30", "tiny_en_c30"])   # tiny_en_c30 = BASELINE configs[0] literally
def test_pipeline_on_gpu_matches_reference_golden(name):
    """thewhisper_amd.ASRPipeline on cuda (strict-f32 engine) reproduces what the reference's nvidia.ASRPipeline (HF
    branch, CPU) returned for the same audio: text and segment timestamps byte for byte, word timestamps (DTW on float32
    cross-attention statistics) within one 0.02 s frame."""
    from tests.test_pipeline_glue import build_amd_pipeline, normalise

    g = json.load(open(os.path.join(GOLD, "pipeline_golden.json")))[name]
    pipe = build_amd_pipeline(g["preset"], g["chunk_s"], g["batch_size"], device="cuda", engine_factory=None, weight_kw=g.get("weight_kw"))
    audio = wo.synth_audio(16000 * g["seconds"], g["seed"], g["kind"])
    gk = {"num_beams": 1, "do_sample": False, "use_cache": True, "language": "en", "max_new_tokens": g["max_new_tokens"]}
    for rt in (False, True, "word"):
        out = normalise(pipe(audio.copy(), generate_kwargs=dict(gk), chunk_length_s=g["chunk_s"] - 1, return_timestamps=rt))
        if rt == "word":
            dev = _assert_same_transcript(out, g["outputs"][str(rt)], 0.0201)
            print(f"{name}: max word-timestamp deviation vs reference = {dev:.3f} s")
        else:
            assert out == g["outputs"][str(rt)], f"return_timestamps={rt}"


def test_streaming_backend_on_gpu_matches_reference_golden():
    from tests.test_pipeline_glue import build_amd_pipeline, normalise
    from thewhisper_amd import AMDWhisperBackend

    g = json.load(open(os.path.join(GOLD, "pipeline_golden.json")))["streaming_micro_c10"]
    pipe = build_amd_pipeline("micro", 10, 1, device="cuda", engine_factory=None)
    backend = AMDWhisperBackend(None, chunk_length_s=10, asr_pipeline=pipe)
    audio = wo.synth_audio(16000 * g["seconds"], g["seed"], g["kind"])
    for c in g["calls"]:
        got = normalise(backend.transcribe(audio[c["offset"] : c["offset"] + c["n"]], c["t0"], 16000))
        assert [w["text"] for w in got] == [w["text"] for w in c["result"]], (c["n"], c["offset"])
        for a, b in zip(got, c["result"]):
            assert abs(a["start"] - b["start"]) <= 0.0201 and abs(a["end"] - b["end"]) <= 0.0201


# ---------------------------------------------------------------- full size, size-independent properties
def test_large_v3_full_size_properties():
    """whisper-large-v3 at BASELINE.json's size (32+32 layers, 10 s chunks, bf16): the oracle is too slow here, so check
    properties: determinism, batch invariance of greedy ids, softmax rows sum to 1, monotone in-range timestamps."""
    import bench

    dims = bench.DIMS["large-v3"]
    from thewhisper_amd.engine import WhisperEngine

    heads = bench.alignment_heads(dims)
    eng = WhisperEngine(dims, 500, max_batch=4, dtype="bf16", alignment_heads=heads)
    eng.load_state_dict(bench.random_state_dict(dims, torch.device("cuda", 0), seed=0))
    pcm = torch.from_numpy(clips(160000, ["speechlike", "noise", "speechlike", "sine"])).cuda()
    pcm[2] = pcm[0]                                                        # stream 2 repeats stream 0

    def run(nb):
        eng.encode(eng.logmel(pcm[:nb]))
        eng.cross_kv(nb)
        prompt = np.tile(np.array(PROMPT, dtype=np.int32), (nb, 1))
        out = eng.generate_greedy(prompt, max_new_tokens=48, timestamps=True, want_alignment=True)
        ts = eng.token_timestamps(nb, 3, out["length"], [1000] * nb)
        al = eng.get_alignment(nb, out["length"] - 1)
        return out["sequences"], ts, al

    s4, t4, a4 = run(4)
    s4b, t4b, _ = run(4)
    assert np.array_equal(s4, s4b) and np.array_equal(t4, t4b)              # deterministic (no atomics / split-K)
    assert np.array_equal(s4[0], s4[2]) and np.array_equal(t4[0], t4[2])    # identical streams -> identical results
    s1, t1, _ = run(1)
    n = min(s1.shape[1], s4.shape[1])
    assert np.array_equal(s1[0, :n], s4[0, :n])                             # batch-size invariance (B=1 kernels vs B=4)
    assert (s4 >= 0).all() and (s4 < dims["vocab"]).all()
    assert np.abs(a4.sum(-1) - 1.0).max() < 2e-3
    assert (np.diff(t4[:, 3:], axis=1) >= -1e-6).all() and t4.min() >= 0 and t4.max() <= 10.0
    eng.close()


def test_large_v3_maximum_context_properties():
    """The largest context the boundary admits - whisper-large-v3, 30 s chunks (T = 1500), 64 streams, bf16: ~16 GB of cross-K/V, four
    groups of 16 streams per projection launch, three 512-key rounds per cross-attention head.  Properties as above."""
    import bench

    dims = bench.DIMS["large-v3"]
    from thewhisper_amd.engine import WhisperEngine

    heads = bench.alignment_heads(dims)
    eng = WhisperEngine(dims, 1500, max_batch=64, dtype="bf16", alignment_heads=heads)
    eng.load_state_dict(bench.random_state_dict(dims, torch.device("cuda", 0), seed=0))
    kinds = ["speechlike", "noise", "sine", "speechlike"]
    pcm = torch.from_numpy(clips(480000, [kinds[i % 4] for i in range(64)])).cuda()
    pcm[17] = pcm[0]
    pcm[63] = pcm[0]

    def run(nb):
        eng.encode(eng.logmel(pcm[:nb]))
        eng.cross_kv(nb)
        prompt = np.tile(np.array(PROMPT, dtype=np.int32), (nb, 1))
        out = eng.generate_greedy(prompt, max_new_tokens=24, timestamps=True, want_alignment=True)
        ts = eng.token_timestamps(nb, 3, out["length"], [3000] * nb)
        return out["sequences"], ts, eng.get_alignment(nb, out["length"] - 1)

    s, t, a = run(64)
    s2, t2, _ = run(64)
    assert np.array_equal(s, s2) and np.array_equal(t, t2)
    for j in (17, 63):                                                     # same audio in another group of 16 streams -> same result
        assert np.array_equal(s[0], s[j]) and np.array_equal(t[0], t[j])
    s1, _, _ = run(1)
    s16, _, _ = run(16)
    n = min(s1.shape[1], s16.shape[1])
    assert np.array_equal(s1[0, :n], s16[0, :n])      # 1 and 16 streams take the same kernels: identical ids
    # 64 streams decode through other instantiations (four groups of 16 per weight pass, fc2 on 8 instead of 16 wavefronts) that sum K
    # in the SAME order (round 6, k_decode.hip): the clip's ids and timestamps are those of the one-stream run, to the last token
    n = min(s1.shape[1], s.shape[1])
    assert np.array_equal(s1[0, :n], s[0, :n])
    assert (s >= 0).all() and (s < dims["vocab"]).all()
    assert a.shape[-1] == 1500 and np.abs(a.sum(-1) - 1.0).max() < 2e-3
    assert (np.diff(t[:, 3:], axis=1) >= -1e-6).all() and t.min() >= 0 and t.max() <= 30.0
    eng.close()


def test_batching_hub_on_gpu_matches_reference_golden():
    """§8f row 1: four concurrent sessions share one MI355X engine through the BatchingHub; every session still gets the
    reference's per-stream result, and requests from different sessions are executed as one batch."""
    import threading

    from tests.test_pipeline_glue import build_amd_pipeline, normalise
    from thewhisper_amd import AMDWhisperBackend
    from thewhisper_amd.serving import BatchingHub

    g = json.load(open(os.path.join(GOLD, "pipeline_golden.json")))["streaming_micro_c10"]
    pipe = build_amd_pipeline("micro", 10, 4, device="cuda", engine_factory=None)
    hub = BatchingHub(AMDWhisperBackend(None, chunk_length_s=10, asr_pipeline=pipe), max_batch=4, max_wait_s=0.5)
    audio = wo.synth_audio(16000 * g["seconds"], g["seed"], g["kind"])
    picks = g["calls"][::5]
    out = [[None] * len(picks) for _ in range(4)]
    barrier = threading.Barrier(4)

    def session(k):
        be = hub.stream_backend()
        for i, c in enumerate(picks):
            barrier.wait()
            out[k][i] = be.transcribe(audio[c["offset"] : c["offset"] + c["n"]], c["t0"], 16000)

    th = [threading.Thread(target=session, args=(k,)) for k in range(4)]
    [t.start() for t in th]
    [t.join(300) for t in th]
    hub.close()
    for k in range(4):
        for i, c in enumerate(picks):
            got = normalise(out[k][i])
            assert [w["text"] for w in got] == [w["text"] for w in c["result"]]
            for a, b in zip(got, c["result"]):
                assert abs(a["start"] - b["start"]) <= 0.0201 and abs(a["end"] - b["end"]) <= 0.0201
    assert max(hub.batches) > 1


def test_encoder_overlap_gives_the_sequential_results():
    """thewhisper_amd/overlap.py: two contexts, encoder stage of batch k+1 on a CU-masked stream under the decode loop of batch
    k.  Same kernels, different schedule: token ids, alignment rows and timestamps equal those of one context run sequentially."""
    from thewhisper_amd.overlap import EncoderOverlap

    dims = wo.PRESETS["micro"]
    w = wo.make_weights(dims, 0)
    heads = [(dims.dec_layers - 1, 0), (dims.dec_layers - 1, 1)]
    engs = [make_engine(dims, w, T=100, max_batch=3, dtype="f32", heads=heads, use_graph=True) for _ in range(2)]
    batches = [torch.from_numpy(np.ascontiguousarray(clips(100 * 320, 3)[::s])).cuda() for s in (1, -1, 1, -1, 1)]
    prompt = np.tile(np.array(PROMPT, dtype=np.int32), (3, 1))

    def enc(e, pc):
        mel = e.logmel(pc, out_dtype=torch.float32)
        e.encode(mel)
        e.cross_kv(pc.shape[0])
        return mel  # must outlive the asynchronous launches (EncoderOverlap.run keeps it until the batch is decoded)

    def dec(e, pc, _):
        out = e.generate_greedy(prompt, max_new_tokens=16, timestamps=True, want_alignment=True)
        L = out["length"]
        return out["sequences"].copy(), e.get_alignment(3, L - 1).copy(), e.token_timestamps(3, 3, L, [200] * 3).copy()

    seq = []
    for b in batches:
        enc(engs[0], b)
        seq.append(dec(engs[0], b, None))
    ov = EncoderOverlap(engs, encoder_cus=32)
    got = ov.run(batches, enc, dec)
    ov.close()
    assert len(got) == len(seq)
    for a, b in zip(got, seq):
        assert np.array_equal(a[0], b[0])
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert not np.array_equal(seq[0][0], seq[1][0])          # the two kinds of batch do differ
    for e in engs:
        e.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16", "fp8"])
def test_slots_filled_in_groups_give_the_one_group_results(dtype):
    """tw_encode_at / tw_cross_kv_at (a serving pass assembled from a group of chunks that need a further seek iteration plus
    late arrivals): per slot the encoder output, the greedy ids and the token timestamps are bit-identical to encoding all
    clips in one call - in every context dtype (fp8: the e4m3 cross-K/V arenas and their scale bytes take the slot offset too)."""
    dims = dims_variant("micro", enc_layers=2, dec_layers=2)
    w = wo.make_weights(dims, 3)
    heads = [(1, 0), (1, 1)]
    T, B = 100, 5
    eng = make_engine(dims, w, T=T, max_batch=8, dtype=dtype, heads=heads, use_graph=True)
    try:
        pcm = clips(32000, ["speechlike", "noise", "sine", "speechlike", "noise"])
        mel = eng.logmel(torch.from_numpy(pcm).cuda())
        prompt = np.tile(np.array(PROMPT, dtype=np.int32), (B, 1))

        def decode():
            out = eng.generate_greedy(prompt, max_new_tokens=20, timestamps=True, want_alignment=True)
            ts = eng.token_timestamps(B, 3, out["sequences"].shape[1], [2 * T] * B)
            return out["sequences"], ts, eng.get_alignment(B, out["sequences"].shape[1] - 1)

        hidden = eng.encode(mel, return_hidden=True).cpu().numpy()
        eng.cross_kv(B)
        ids0, ts0, al0 = decode()
        # the same five clips in three groups: slots 0-1, 2, 3-4 (a different pass shape was in the context before)
        eng.encode(mel[:2])
        eng.cross_kv(2)
        eng.encode(mel[2:3], slot0=2)
        eng.cross_kv(1, slot0=2)
        eng.encode(mel[3:5], slot0=3)
        eng.cross_kv(2, slot0=3)
        ids1, ts1, al1 = decode()
        assert np.array_equal(ids0, ids1) and np.array_equal(ts0, ts1) and np.array_equal(al0, al1)
        # the first group alone reproduces its rows of the one-call encoder output (row results do not depend on the batch)
        assert np.array_equal(eng.encode(mel[:2], return_hidden=True).cpu().numpy(), hidden[:2])
        # misuse is refused: a gap before slot0, or slots beyond the context capacity
        with pytest.raises(RuntimeError):
            eng.encode(mel[:1], slot0=4)
        with pytest.raises(RuntimeError):
            eng.encode(mel[:5], slot0=4)
    finally:
        eng.close()


def test_c_abi_error_behaviour():
    """include/thewhisper.h: every entry point answers misuse with a negative TW_E* code and a message in tw_last_error - no
    exception crosses the boundary, nothing is launched - and the context stays usable: the valid call that follows gives the
    oracle's result."""
    import ctypes as C

    dims = wo.PRESETS["micro"]
    w = wo.make_weights(dims, 0)
    eng = make_engine(dims, w, T=100, max_batch=2, dtype="f32")
    lib, ctx, sp = eng.lib, eng.ctx, eng._sp()
    TW_EINVAL, TW_ESTATE = -1, -3

    def refused(rc, codes=(TW_EINVAL, TW_ESTATE)):
        assert rc in codes, (rc, lib.tw_last_error(ctx).decode())
        assert len(lib.tw_last_error(ctx)) > 0
    try:
        pcm = clips(32000, 2)
        # call order: decoding before any chunk was encoded
        ids = (C.c_int32 * 2)(50258, 50258)
        refused(lib.tw_cross_kv(ctx, 2, sp))
        refused(lib.tw_decoder_reset(ctx, 2, sp))
        refused(lib.tw_decode_step(ctx, 2, ids, None, sp))
        mel = eng.logmel(torch.from_numpy(pcm).cuda(), out_dtype=torch.float32)
        mp = C.c_void_p(mel.data_ptr())
        # batch outside 1 .. max_batch, null operands, unknown dtypes
        refused(lib.tw_encode(ctx, mp, 0, 0, None, 0, sp))
        refused(lib.tw_encode(ctx, mp, 0, 3, None, 0, sp))
        refused(lib.tw_encode(ctx, None, 0, 2, None, 0, sp))
        refused(lib.tw_encode(ctx, mp, 77, 2, None, 0, sp))
        hid = torch.empty((2, 100, dims.d_model), device="cuda")
        refused(lib.tw_encode(ctx, mp, 0, 2, C.c_void_p(hid.data_ptr()), 5, sp))                          # hidden states in an unknown element type
        refused(lib.tw_logmel(ctx, None, 32000, None, 2, 32000, mp, 0, sp))
        refused(lib.tw_logmel(ctx, C.c_void_p(mel.data_ptr()), 32000, None, 2, 32001, mp, 0, sp))     # not a whole number of hops
        eng.encode(mel)
        refused(lib.tw_cross_kv(ctx, 3, sp))
        refused(lib.tw_cross_kv_at(ctx, 1, 2, sp))
        eng.cross_kv(2)
        # greedy call: prompt that does not fit, null outputs, more streams than were encoded
        from thewhisper_amd import _cabi

        o = _cabi.tw_greedy_opts()
        o.eos_id = o.pad_id = 50257
        o.max_new_tokens, o.min_new_tokens, o.max_length = 4, 0, 448
        o.no_timestamps_id, o.max_initial_timestamp_index = 50364, 50
        prompt = np.tile(np.array(PROMPT, dtype=np.int32), (2, 1))
        pp = prompt.ctypes.data_as(C.POINTER(C.c_int32))
        out = np.zeros((2, 448), np.int32)
        op = out.ctypes.data_as(C.POINTER(C.c_int32))
        n = C.c_int32(0)
        refused(lib.tw_generate_greedy(ctx, 3, pp, 3, C.byref(o), op, C.byref(n), sp))
        refused(lib.tw_generate_greedy(ctx, 2, pp, 0, C.byref(o), op, C.byref(n), sp))
        refused(lib.tw_generate_greedy(ctx, 2, pp, 449, C.byref(o), op, C.byref(n), sp))
        refused(lib.tw_generate_greedy(ctx, 2, None, 3, C.byref(o), op, C.byref(n), sp))
        refused(lib.tw_generate_greedy(ctx, 2, pp, 3, None, op, C.byref(n), sp))
        refused(lib.tw_generate_greedy(ctx, 2, pp, 3, C.byref(o), None, C.byref(n), sp))
        # word timestamps without recorded alignment rows
        ts = np.zeros((2, 8), np.float32)
        refused(lib.tw_token_timestamps(ctx, 2, 3, 8, None, 0.02, ts.ctypes.data_as(C.POINTER(C.c_float)), sp))
        # unknown weight name / wrong shape after construction
        t = torch.zeros(4, device="cuda")
        shape = (C.c_int64 * 1)(4)
        assert lib.tw_load_weight(ctx, b"model.no.such.tensor", C.c_void_p(t.data_ptr()), 0, 1, shape, sp) < 0
        assert lib.tw_finalize_weights(ctx, sp) < 0                                                    # "call exactly once"
        # ... and the context still works
        got = eng.generate_greedy(prompt, max_new_tokens=12, timestamps=True)
        om = wo.OracleWhisper(dims, w, T=100)
        ref = wo.greedy_generate(om, om.encode(wo.log_mel(pcm, dims.n_mels)), prompt, wo.GreedyOptions(max_new_tokens=12, timestamps=True))
        assert np.array_equal(got["sequences"], ref["sequences"])
    finally:
        eng.close()
