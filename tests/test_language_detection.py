"""The multilingual `language=None` path (HF `detect_language`, HF:models/whisper/generation_whisper.py:1610-1683, reached from
`_retrieve_init_tokens` :1455-1608): one teacher-forced forward pass on `<|startoftranscript|>` whose language-token logits pick
the prompt per row.  The drop-in answers that pass from the engine (`AMDWhisperForConditionalGeneration.forward`), the rest is
HF's own control flow.  CPU: the stand-in engine; GPU (`-m gpu`): the MI355X engine in strict f32.  Reference = the installed HF
model with the same seeded weights (what the reference's nvidia.ASRPipeline wraps)."""
import numpy as np
import pytest
import torch

from oracle import hf_reference as hr
from oracle import whisper_oracle as wo
from tests.oracle_engine import oracle_engine_factory

torch.set_grad_enabled(False)


def _both(device, engine_factory, chunk_s=10):
    from thewhisper_amd import ASRPipeline

    dims = wo.PRESETS["micro"]
    w = wo.make_weights(dims, 0)
    ref = hr.patch_chunk_length(hr.build_hf_model(dims, w), chunk_s)
    kw = {} if engine_factory is None else {"engine_factory": engine_factory}
    pipe = ASRPipeline(hr.build_hf_model(dims, w), feature_extractor=hr.build_feature_extractor(dims, chunk_s), tokenizer=hr.build_tokenizer(dims),
                       chunk_length_s=chunk_s, device=device, torch_dtype=torch.float32, batch_size=3, **kw)
    return dims, ref, pipe


def _check(device, engine_factory):
    dims, ref, pipe = _both(device, engine_factory)
    fe = hr.build_feature_extractor(dims, 10)
    clips = [wo.synth_audio(160000, s, k) for s, k in ((0, "speechlike"), (1, "noise"), (2, "sine"))]
    feats = fe(clips, sampling_rate=16000, return_tensors="pt", return_attention_mask=True)
    x, am = feats["input_features"], feats["attention_mask"]
    # (1) the detection itself: the language ids HF's detect_language derives from OUR forward pass == from the reference model's
    want_lang = ref.detect_language(input_features=x)
    got_lang = pipe.model.detect_language(input_features=x.to(pipe.model.device)).cpu()
    assert torch.equal(want_lang, got_lang)
    # (2) generate(language=None): detection -> per-row prompt -> greedy decode; ids identical to the reference model's
    gk = dict(language=None, task="transcribe", num_beams=1, do_sample=False, max_new_tokens=16, return_timestamps=True)
    want = ref.generate(input_features=x, attention_mask=am, **gk)
    got = pipe.model.generate(input_features=x.to(pipe.model.device), attention_mask=am.to(pipe.model.device), **gk).cpu()
    assert torch.equal(want, got), (want, got)
    assert len(set(want_lang.tolist())) > 1           # (the rows do get different prompts: the per-row path is exercised)
    # (3) `encoder_outputs` instead of `input_features` is refused, loudly (the engine keeps its encoder states internally)
    with pytest.raises(NotImplementedError, match="encoder_outputs"):
        pipe.model.forward(encoder_outputs=(torch.zeros(3, 500, dims.d_model),), decoder_input_ids=torch.full((3, 1), 50258))
    with pytest.raises(NotImplementedError, match="decoder_input_ids"):
        pipe.model.forward(input_features=x.to(pipe.model.device))


def test_language_detection_cpu():
    _check("cpu", oracle_engine_factory)


@pytest.mark.gpu
def test_language_detection_on_the_mi355x():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    _check("cuda", None)
