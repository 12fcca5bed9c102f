"""Voice-activity gate on the MI355X with silero-vad's calling contract.

The reference gates ``StreamingPipeline.add_new_chunk`` with silero-vad loaded through ``torch.hub`` - a network download -
and calls it as ``prob = self.vad_model(torch_frame_512, 16000).item()`` on consecutive 512-sample frames, the model keeping
state between calls (R:thestage_speechkit/streaming/streaming_pipeline.py:533-538, :589-622).  ``EnergyVAD`` honours that
contract with a different detector: an adaptive-noise-floor energy rule (stated in csrc/k_vad.hip, restated in
oracle/whisper_oracle.py::energy_vad) computed by ``tw_vad_energy``.  It is NOT silero and makes no claim of equal
decisions; it exists so that the default ``use_vad=True`` path of the reference scheduler can run offline and so that a
serving tick can gate all sessions in one launch (``BatchedVAD``).

    sp = StreamingPipeline(backend=..., use_vad=False)       # constructor must not reach torch.hub
    attach_vad(sp)                                           # sp.vad_model = EnergyVAD(); sp.use_vad = True
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _cabi

FRAME = 512


class BatchedVAD:
    """``n_streams`` independent detectors; ``probs(frames)`` runs the next frame(s) of every stream in ONE launch."""

    def __init__(self, n_streams: int = 1, device: int = 0):
        if not torch.cuda.is_available():
            raise RuntimeError("thewhisper_amd.vad needs an MI355X (no CPU fallback)")
        self.lib = _cabi.load_library()
        self.n = int(n_streams)
        self.device = torch.device("cuda", device)
        self.state = torch.zeros((self.n, 2), dtype=torch.float32, device=self.device)

    def reset_states(self, stream: Optional[int] = None):
        if stream is None:
            self.state.zero_()
        else:
            self.state[stream].zero_()

    def probs(self, pcm) -> torch.Tensor:
        """pcm: float32 [n_streams, k*512] (numpy or torch, host or device) -> device tensor [n_streams, k] of speech
        probabilities; the per-stream state advances by k frames."""
        x = torch.as_tensor(pcm, dtype=torch.float32)
        if x.dim() == 1:
            x = x[None]
        if x.shape[0] != self.n or x.shape[1] < FRAME or x.shape[1] % FRAME:
            raise ValueError(f"expected [{self.n}, k*{FRAME}] samples, got {tuple(x.shape)}")
        x = x.to(self.device).contiguous()
        k = x.shape[1] // FRAME
        out = torch.empty((self.n, k), dtype=torch.float32, device=self.device)
        st = torch.cuda.current_stream(self.device).cuda_stream
        rc = self.lib.tw_vad_energy(self.device.index or 0, C.c_void_p(x.data_ptr()), x.stride(0), self.n, k,
                                    C.c_void_p(self.state.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(int(st) if st else None))
        if rc != 0:
            raise RuntimeError(f"tw_vad_energy failed ({rc}): {self.lib.tw_last_error(None).decode()}")
        return out


class EnergyVAD:
    """Single-stream detector with silero's call signature: ``vad(frame_512, sampling_rate) -> 0-dim tensor`` (``.item()``)."""

    def __init__(self, device: int = 0):
        self._b = BatchedVAD(1, device)

    def reset_states(self):
        self._b.reset_states()

    def __call__(self, x, sampling_rate: int = 16000) -> torch.Tensor:
        if sampling_rate != 16000:
            raise ValueError("EnergyVAD expects 16 kHz audio")
        x = torch.as_tensor(x, dtype=torch.float32).reshape(-1)
        if x.numel() != FRAME:
            raise ValueError(f"EnergyVAD expects {FRAME}-sample frames (as silero-vad at 16 kHz), got {x.numel()}")
        return self._b.probs(x[None])[0, 0]


def attach_vad(streaming_pipeline, vad=None):
    """Switch the voice-activity gate of a reference ``StreamingPipeline`` (constructed with ``use_vad=False`` so that its
    constructor does not call torch.hub) to ``vad`` (default: a fresh ``EnergyVAD``)."""
    streaming_pipeline.vad_model = vad if vad is not None else EnergyVAD()
    streaming_pipeline.use_vad = True
    return streaming_pipeline
