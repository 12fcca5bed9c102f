#!/usr/bin/env python3
"""Round 5 diagnostic (MI355X): "the same audio in another group of 16 streams gives the same result" at any size.

    python tools/dbg/group_symmetry.py ids <dec layers> <T> <graph 0|1> <new tokens>      [ENC_LAYERS=32]
        64 streams with different audio except that streams 17 and 63 repeat stream 0; free-running greedy ids of the three
    python tools/dbg/group_symmetry.py logits <dec layers>
        the same streams, teacher-forced logits: (a) 17 / 63 against 0 (must be bit-identical), (b) every stream against the same
        stream decoded in a 16-stream context (bf16-rounding-level differences are legitimate: fc2 splits K differently above 16 streams)

Run under TW_SK_CG_MODE / THEWHISPER_DECODE_CUS / TW_FUSE_EMBED settings (tools/ab.sh) to bisect; this is what located the failure of the
two-tiles-per-workgroup variant of the projection kernel in round 5 (profiles/r05_group_symmetry_bisect.txt)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch


def mode_ids(argv):
    import bench
    from tests.util import PROMPT, clips
    from thewhisper_amd.engine import WhisperEngine


    layers, T, graph, new = int(argv[0]), int(argv[1]), int(argv[2]), int(argv[3])
    dims = dict(bench.DIMS["large-v3"], enc_layers=int(os.environ.get("ENC_LAYERS", "2")), dec_layers=layers)
    heads = bench.alignment_heads(dims)
    eng = WhisperEngine(dims, T, max_batch=64, dtype="bf16", alignment_heads=heads, use_graph=bool(graph))
    eng.load_state_dict(bench.random_state_dict(dims, torch.device("cuda", 0), seed=0))
    kinds = ["speechlike", "noise", "sine", "speechlike"]
    pcm = torch.from_numpy(clips(T * 320, [kinds[i % 4] for i in range(64)])).cuda()
    pcm[17] = pcm[0]; pcm[63] = pcm[0]
    eng.encode(eng.logmel(pcm)); eng.cross_kv(64)
    prompt = np.tile(np.array(PROMPT, dtype=np.int32), (64, 1))
    out = eng.generate_greedy(prompt, max_new_tokens=new, timestamps=True, want_alignment=True)
    s = out["sequences"]
    def first_diff(a, b):
        d = np.nonzero(a != b)[0]
        return int(d[0]) if len(d) else None
    print(f"enc={dims['enc_layers']} mode={os.environ.get('TW_SK_CG_MODE','default')} cus={os.environ.get('THEWHISPER_DECODE_CUS','160')} layers={layers} T={T} graph={graph} new={new}: "
          f"first difference 17 vs 0: {first_diff(s[17], s[0])}, 63 vs 0: {first_diff(s[63], s[0])}, 17 vs 63: {first_diff(s[17], s[63])}", flush=True)
    eng.close()


def mode_logits(argv):
    from oracle import whisper_oracle as wo
    from tests.util import PROMPT, clips, dims_variant, make_engine


    B = 64
    layers = int(argv[0]) if argv else 1
    dims = dims_variant("large-v3", enc_layers=1, dec_layers=layers)
    w = wo.make_weights(dims, 2)
    T = 100
    kinds = ["speechlike", "noise", "sine", "speechlike"]
    pcm = clips(T * 320, [kinds[i % 4] for i in range(B)])
    pcm[17] = pcm[0]; pcm[63] = pcm[0]
    mel = wo.log_mel(pcm, dims.n_mels)
    ids = np.concatenate([np.tile(np.array(PROMPT), (B, 1)), np.random.default_rng(3).integers(0, 50000, size=(B, 4))], axis=1)
    ids[17] = ids[0]; ids[63] = ids[0]

    def run(eng, sel):
        n = len(sel)
        eng.encode(torch.from_numpy(mel[sel]).cuda()); eng.cross_kv(n); eng.decoder_reset(n)
        return np.stack([eng.decode_step(ids[sel, s].tolist()).cpu().numpy() for s in range(ids.shape[1])], axis=1)

    big = make_engine(dims, w, T=T, max_batch=B, dtype="bf16")
    got = run(big, np.arange(B)); big.close()
    small = make_engine(dims, w, T=T, max_batch=16, dtype="bf16")
    ref = np.concatenate([run(small, np.arange(lo, lo + 16)) for lo in (0, 16, 32, 48)]); small.close()
    m = os.environ.get("TW_SK_CG_MODE", "default")
    for s in range(ids.shape[1]):
        d17, d63 = np.abs(got[17, s] - got[0, s]).max(), np.abs(got[63, s] - got[0, s]).max()
        rel = np.linalg.norm(got[:, s] - ref[:, s], axis=1) / np.linalg.norm(ref[:, s], axis=1)
        worst = np.argsort(-rel)[:6]
        print(f"mode={m} layers={layers} step {s}: |17-0|={d17:.3e} |63-0|={d63:.3e}  rel-L2 vs 16-stream context: max {rel.max():.3e} median {np.median(rel):.3e} worst streams {worst.tolist()}"
              f" by group {[round(float(rel[g*16:(g+1)*16].max()),5) for g in range(4)]}", flush=True)


if __name__ == "__main__":
    (mode_ids if sys.argv[1] == "ids" else mode_logits)(sys.argv[2:])
