"""MI355X counterpart of the reference's examples/run_nvidia_asr.py (same flow, `amd` backend).

    python examples/run_amd_asr.py --audio-file speech.wav            # needs a checkpoint + librosa
    python examples/run_amd_asr.py --synthetic                        # seeded random weights + synthetic clip (no network)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from thewhisper_amd import ASRPipeline  # noqa: E402

parser = argparse.ArgumentParser()
parser.add_argument("--audio-file", type=str, default="example_speech.wav", help="Path to the audio file to transcribe")
parser.add_argument("--model", type=str, default="TheStageAI/thewhisper-large-v3-turbo")
parser.add_argument("--synthetic", action="store_true", help="random-weight micro model + synthetic audio (offline smoke run)")
args = parser.parse_args()

generate_kwargs = {"num_beams": 1, "do_sample": False, "use_cache": True, "language": "en"}
chunk_length_s = 10

if args.synthetic:
    import numpy as np
    from transformers import WhisperFeatureExtractor

    from thewhisper_amd import synthetic
    from thewhisper_amd.engine import WhisperEngine

    # no checkpoint, no tokenizer files: a model of the tiny.en SHAPE with random weights generated on the device
    dims = synthetic.DIMS["tiny.en"]
    heads = [tuple(h) for h in synthetic.default_alignment_heads(dims["dec_layers"], dims["heads"])]
    eng = WhisperEngine(dims, 50 * chunk_length_s, max_batch=4, dtype="bf16", alignment_heads=heads)
    eng.load_state_dict(synthetic.random_state_dict(dims, torch.device("cuda", 0), seed=0))
    pipe = ASRPipeline(synthetic.skeleton_model(dims, device="cuda:0", dtype=torch.bfloat16, alignment_heads=heads),
                       feature_extractor=WhisperFeatureExtractor(feature_size=dims["n_mels"], chunk_length=chunk_length_s),
                       tokenizer=synthetic.build_tokenizer(dims["vocab"]), chunk_length_s=chunk_length_s, batch_size=4,
                       device="cuda:0", torch_dtype=torch.bfloat16, engine=eng)
    audio = (np.random.default_rng(5).standard_normal(16000 * 25) * 0.1).astype(np.float32)
    generate_kwargs["max_new_tokens"] = 32
else:
    from librosa import load, resample

    pipe = ASRPipeline(args.model, chunk_length_s=chunk_length_s, model_size="S", batch_size=16, device="cuda",
                       torch_dtype=torch.bfloat16)
    audio, sr = load(args.audio_file)
    audio = resample(audio, orig_sr=sr, target_sr=16000)

output = pipe(audio, generate_kwargs=generate_kwargs, chunk_length_s=chunk_length_s - 1, return_timestamps="word")
print(output)
