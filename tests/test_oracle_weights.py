"""oracle.whisper_oracle.make_weights produces its tensors on several threads, each from a copy of ONE seeded generator advanced to
the tensor's place in the stream.  The values every golden file was generated with are those of the plain sequential loop -
`(rng.random(shape, float32) - 0.5) * (2 * amp)`, tensor after tensor - restated here; the two must agree bit for bit."""
import math

import numpy as np

from oracle import whisper_oracle as wo


def sequential_weights(dims, seed, scale, q_gain):
    rng = np.random.default_rng(seed)
    d, f, v = dims.d_model, dims.ffn, dims.vocab
    w = {}

    def uni(shape, amp):
        return ((rng.random(shape, dtype=np.float32) - 0.5) * (2.0 * amp)).astype(np.float32)

    def lin(name, out_f, in_f, bias=True):
        w[name + ".weight"] = uni((out_f, in_f), scale * 1.7 / math.sqrt(in_f))
        if bias:
            w[name + ".bias"] = uni((out_f,), 0.05)

    def ln(name):
        w[name + ".weight"] = (1.0 + uni((d,), 0.1)).astype(np.float32)
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
    w[e + ".embed_positions.weight"] = wo.sinusoids(dims.max_source_positions, d)
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
    if q_gain != 1.0:
        for k in w:
            if ".q_proj." in k:
                w[k] = (w[k] * np.float32(q_gain)).astype(np.float32)
    return w


def test_threaded_generation_is_the_sequential_stream():
    for preset, seed, scale, q_gain in (("micro", 0, 1.0, 1.0), ("micro", 7, 0.5, 8.0), ("micro80", 3, 0.5, 8.0)):
        if preset not in wo.PRESETS:
            continue
        dims = wo.PRESETS[preset]
        a = sequential_weights(dims, seed, scale, q_gain)
        b = wo.make_weights(dims, seed, scale=scale, q_gain=q_gain)
        assert list(a.keys()) == list(b.keys())
        for k in a:
            assert a[k].dtype == b[k].dtype == np.float32 and a[k].shape == b[k].shape, k
            assert np.array_equal(a[k], b[k]), k       # odd- and even-sized tensors: both halves of a 64-bit step occur as a start


def test_a_tensor_can_start_anywhere_in_the_stream():
    """Even and odd offsets (a tensor that starts on the high half of a 64-bit step), lengths that end on either half."""
    ref = np.random.default_rng(11).random(4001, dtype=np.float32)
    for off, n in ((0, 7), (1, 6), (2, 9), (333, 1000), (334, 1001), (3999, 2)):
        assert np.array_equal(wo.uniform_at(11, off, (n,)), ref[off : off + n]), (off, n)
    # ... and consecutive draws from ONE generator are that stream too (what the goldens' sequential loop did)
    g = np.random.default_rng(11)
    parts = [g.random(k, dtype=np.float32) for k in (3, 5, 1, 8)]
    assert np.array_equal(np.concatenate(parts), ref[:17])
