"""CPU checks of the MXFP8 restatement used by the fp8 parity tests (oracle/whisper_oracle.py)."""
import numpy as np

from oracle import whisper_oracle as wo


def _e4m3_table():
    vals = []
    for e in range(16):
        for m in range(8):
            if e == 15 and m == 7:
                continue  # NaN
            vals.append(m * 2.0**-9 if e == 0 else (1 + m / 8) * 2.0 ** (e - 7))
    return np.array(sorted(set(vals)))


def test_e4m3_rne_matches_table_search():
    tab = _e4m3_table()
    assert tab.max() == 448.0 and len(tab) == 127
    rng = np.random.default_rng(0)
    v = np.concatenate([rng.uniform(-255, 255, 4000), rng.uniform(-0.05, 0.05, 2000), tab[tab < 256], (tab[:-1] + tab[1:])[tab[1:] < 256] / 2])
    got = wo.e4m3_rne(v)
    full = np.concatenate([-tab[::-1], tab])
    for x, g in zip(v, got):
        d = np.abs(full - x)
        best = full[d == d.min()]
        if len(best) == 1 or best[0] == -best[-1]:
            assert g == best[0] or (g == 0 and best[0] == 0)
        else:  # tie: even mantissa = the candidate whose index in the positive table is even
            idx = [int(np.where(tab == abs(b))[0][0]) for b in best]
            want = [b for b, i in zip(best, idx) if i % 2 == 0]
            assert g in want


def test_mx8_block_structure_and_error():
    rng = np.random.default_rng(1)
    x = wo.bf16_round(rng.standard_normal((3, 256)).astype(np.float32) * np.array([[1e-3], [1.0], [300.0]], dtype=np.float32))
    q = wo.mx8_quant_dequant(x)
    rel = np.abs(q - x) / np.maximum(np.abs(x), 1e-30)
    v = x.reshape(3, 2, 2, 2, 2, 2, 8)                         # [row, s, h, mm, u, kk, e]
    amax = np.abs(v).max(axis=(3, 5, 6), keepdims=True)
    big = (np.abs(v) > amax * 2.0**-6).reshape(x.shape)      # elements that stay normal numbers after scaling
    assert rel[big].max() <= 2.0**-4 + 1e-6                    # 3 mantissa bits
    assert np.array_equal(wo.mx8_quant_dequant(q), q)          # idempotent
    # scaling a block by a power of two scales its quantised values exactly (block scales are powers of two)
    assert np.array_equal(wo.mx8_quant_dequant(x * 8.0), q * 8.0)
    # changing one block leaves the others untouched: block (s=0, h=0, u=1) = k in {mm*32 + (2 + kk)*8 + e}
    y = x.copy()
    ks = [mm * 32 + (2 + kk) * 8 + e for mm in range(2) for kk in range(2) for e in range(8)]
    y[:, ks] *= 64.0
    q2 = wo.mx8_quant_dequant(y)
    other = np.ones(256, dtype=bool)
    other[ks] = False
    assert np.array_equal(q2[:, other], q[:, other])


def test_mxfp8_oracle_close_to_exact_oracle():
    dims = wo.PRESETS["micro"]
    w = wo.make_weights(dims, 0)
    a, b = wo.OracleWhisper(dims, w, T=100), wo.OracleWhisperMXFP8(dims, w, T=100)
    enc = a.encode(wo.log_mel(wo.synth_audio(16000 * 2, 0)[None], dims.n_mels))
    ca, cb = a.new_cache(enc), b.new_cache(enc)
    ids = np.array([[50258, 50259, 50360, 17]])
    la, lb = a.decode(ids, ca)[0], b.decode(ids, cb)[0]
    err = np.linalg.norm(la - lb) / np.linalg.norm(la)
    assert 1e-3 < err < 1e-1


def test_cross_query_ahead_is_the_same_projection_up_to_quantisation_noise():
    """The engine accumulates the cross-attention query before its LayerNorm (x0 W'^T + attn (W' Wo)^T + W' bo); the
    restatement of that order and of the plain order (LayerNorm of x1, then W') differ by fp8 noise only, and WITHOUT the
    quantisers the algebra is exact to fp32 rounding."""
    dims = wo.PRESETS["micro"]
    w = wo.make_weights(dims, 1)
    a, b = wo.OracleWhisperMXFP8(dims, w, T=100, cross_q_ahead=True), wo.OracleWhisperMXFP8(dims, w, T=100, cross_q_ahead=False)
    enc = wo.OracleWhisper(dims, w, T=100).encode(wo.log_mel(wo.synth_audio(16000 * 2, 1)[None], dims.n_mels))
    ids = np.array([[50258, 50259, 50360, 23]])
    la, lb = a.decode(ids, a.new_cache(enc))[0], b.decode(ids, b.new_cache(enc))[0]
    err = np.linalg.norm(la - lb) / np.linalg.norm(lb)
    assert 0 < err < 6e-2
    # the identity itself, in float64 on one layer's tensors
    rng = np.random.default_rng(0)
    d = dims.d_model
    x0, attn = rng.standard_normal((3, d)), rng.standard_normal((3, d))
    Wq, Wo, bo = rng.standard_normal((d, d)) / 8, rng.standard_normal((d, d)) / 8, rng.standard_normal(d)
    x1 = x0 + attn @ Wo.T + bo
    assert np.allclose(x1 @ Wq.T, x0 @ Wq.T + attn @ (Wq @ Wo).T + Wq @ bo, rtol=1e-10, atol=1e-10)


def test_kv8_per_key_scales():
    """fp8 cross-K/V restatement: one power-of-two scale per (key, head) row of 64 dims, e4m3 elements."""
    rng = np.random.default_rng(2)
    x = wo.bf16_round(rng.standard_normal((2, 3, 5, 64)).astype(np.float32) * rng.uniform(1e-3, 50.0, (2, 3, 5, 1)).astype(np.float32))
    q = wo.kv8_quant_dequant(x)
    amax = np.abs(x).max(axis=-1, keepdims=True)
    big = np.abs(x) > amax * 2.0**-6
    assert (np.abs(q - x)[big] / np.abs(x)[big]).max() <= 2.0**-4 + 1e-6
    assert np.array_equal(wo.kv8_quant_dequant(q), q)                                  # idempotent
    assert np.array_equal(wo.kv8_quant_dequant(x * 4.0), q * 4.0)                      # power-of-two scaling commutes
    y = x.copy()
    y[0, 1, 2] *= 1024.0                                                               # another key's scale is untouched
    q2 = wo.kv8_quant_dequant(y)
    mask = np.ones(x.shape[:3], bool)
    mask[0, 1, 2] = False
    assert np.array_equal(q2[mask], q[mask])
    assert np.array_equal(wo.kv8_quant_dequant(np.zeros((1, 1, 2, 64), np.float32)), np.zeros((1, 1, 2, 64), np.float32))
