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
    from oracle import hf_reference as hr  # test infrastructure, used here only to fabricate a model offline
    from oracle import whisper_oracle as wo

    dims = wo.PRESETS["micro"]
    pipe = ASRPipeline(hr.build_hf_model(dims, wo.make_weights(dims, 0)),
                       feature_extractor=hr.build_feature_extractor(dims, chunk_length_s), tokenizer=hr.build_tokenizer(dims),
                       chunk_length_s=chunk_length_s, batch_size=4, device="cuda", torch_dtype=torch.bfloat16)
    audio = wo.synth_audio(16000 * 25, 5, "speechlike")
    generate_kwargs["max_new_tokens"] = 32
else:
    from librosa import load, resample

    pipe = ASRPipeline(args.model, chunk_length_s=chunk_length_s, model_size="S", batch_size=16, device="cuda",
                       torch_dtype=torch.bfloat16)
    audio, sr = load(args.audio_file)
    audio = resample(audio, orig_sr=sr, target_sr=16000)

output = pipe(audio, generate_kwargs=generate_kwargs, chunk_length_s=chunk_length_s - 1, return_timestamps="word")
print(output)
