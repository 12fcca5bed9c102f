"""CPU stand-in for ``thewhisper_amd.engine.WhisperEngine`` backed by the numpy oracle.

TESTS ONLY.  It lets the HF glue of the product package (``model.py``, ``asr_pipeline.py``, ``streaming.py``,
``feature_extraction.py``) be exercised on a machine without a GPU and compared with the reference's own
pipeline; the product never imports it - it is injected through the ``engine_factory`` test seam.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import torch

from oracle import whisper_oracle as wo


class OracleEngine:
    def __init__(self, dims: Dict[str, int], T: int, max_batch: int, dtype: str, alignment_heads, device_index: int = 0):
        self.dims = wo.WhisperDims(**{k: dims[k] for k in ("d_model", "enc_layers", "dec_layers", "heads", "ffn", "vocab", "n_mels")},
                                   max_target_positions=dims.get("max_target_positions", 448))
        self.T = T
        self.max_batch = max_batch
        self.n_mels = self.dims.n_mels
        self.vocab = self.dims.vocab
        self.d_model = self.dims.d_model
        self.alignment_heads = [tuple(h) for h in alignment_heads]
        self.model: Optional[wo.OracleWhisper] = None
        self.calls = {"logmel": 0, "encode": 0, "generate": 0, "dtw": 0}

    def load_state_dict(self, sd):
        w = {k: v.detach().float().cpu().numpy() for k, v in sd.items()}
        self.model = wo.OracleWhisper(self.dims, w, T=self.T)

    def logmel(self, pcm: torch.Tensor, n_valid=None, n_samples=None, out_dtype=None) -> torch.Tensor:
        self.calls["logmel"] += 1
        x = pcm.detach().cpu().numpy()
        if x.ndim == 1:
            x = x[None]
        if n_valid is not None:
            x = np.stack([np.where(np.arange(x.shape[1]) < nv, x[i], 0) for i, nv in enumerate(n_valid)])
        return torch.from_numpy(wo.log_mel(x, self.n_mels, n_samples))

    def encode(self, mel: torch.Tensor, return_hidden=False, hidden_dtype=torch.float32, slot0=0):
        self.calls["encode"] += 1
        m = mel.detach().float().cpu().numpy()
        if m.shape[1:] != (self.n_mels, 2 * self.T):
            raise ValueError(f"Whisper expects the mel input features to be of length {2 * self.T}, but found {m.shape[-1]}")
        enc = self.model.encode(m)
        if slot0:      # tw_encode_at: the slots before slot0 keep what this pass put there
            assert slot0 <= self._enc.shape[0] and slot0 + enc.shape[0] <= self.max_batch
            self._enc = np.concatenate([self._enc[:slot0], enc], axis=0)
        else:
            self._enc = enc
        return torch.from_numpy(self._enc) if return_hidden else None

    def cross_kv(self, B: int, slot0=0):
        assert self._enc.shape[0] >= slot0 + B

    device = None   # no HIP device: serving._Prefetcher then runs without a CU-masked stream

    def sibling(self, max_batch=None):
        """Same model, own state (WhisperEngine.sibling / tw_create_sibling)."""
        new = OracleEngine.__new__(OracleEngine)
        new.__dict__.update(self.__dict__)
        new.max_batch = int(max_batch or self.max_batch)
        new.calls = {"logmel": 0, "encode": 0, "generate": 0, "dtw": 0}
        new._enc = None
        return new

    def adopt_cross_kv(self, src, src_slot0: int, B: int, dst_slot0: int):
        """tw_adopt_cross_kv: slots of another context become this one's next slots."""
        part = src._enc[src_slot0 : src_slot0 + B]
        assert part.shape[0] == B and dst_slot0 + B <= self.max_batch
        prev = self._enc[:dst_slot0] if dst_slot0 else part[:0]
        assert prev.shape[0] == dst_slot0
        self._enc = np.concatenate([prev, part], axis=0)

    def close(self):
        pass

    def decoder_reset(self, B: int):
        self._cache = self.model.new_cache(self._enc[:B])

    def decode_step(self, ids: Sequence[int], want_logits: bool = True):
        lg, _ = self.model.decode(np.asarray(ids)[:, None], self._cache)
        return torch.from_numpy(lg[:, 0].astype(np.float32))

    def generate_greedy(self, prompt, max_new_tokens=128, min_new_tokens=0, max_length=448, eos_id=50257, pad_id=50257,
                        timestamps=False, no_timestamps_id=50364, max_initial_timestamp_index=50, begin_suppress=(220, 50257),
                        suppress=(), want_alignment=False, n_forced=0, n_draft=0):
        self.calls["generate"] += 1
        opt = wo.GreedyOptions(eos=eos_id, pad=pad_id, max_new_tokens=max_new_tokens, min_new_tokens=min_new_tokens,
                               max_length=max_length, begin_suppress=tuple(begin_suppress), suppress=tuple(suppress),
                               timestamps=timestamps, no_timestamps_id=no_timestamps_id,
                               max_initial_timestamp_index=max_initial_timestamp_index,
                               alignment_heads=self.alignment_heads if want_alignment else None)
        B = prompt.shape[0]
        prompt = np.asarray(prompt)
        draft = None
        if n_draft:     # tw_greedy_opts::n_draft, by its definition: the result is the call's result WITHOUT the guesses
            assert not n_forced
            draft, prompt = prompt[:, prompt.shape[1] - int(n_draft):], prompt[:, : prompt.shape[1] - int(n_draft)]
        res = wo.greedy_generate(self.model, self._enc[:B], prompt, opt,
                                 begin_index=(prompt.shape[1] - int(n_forced)) if n_forced else None)
        self._cross = res["cross"]
        out = {"sequences": res["sequences"], "length": int(res["sequences"].shape[1])}
        if draft is not None:   # confirmed = the common prefix of guess and result, the shortest over the streams (lock-step), per stream
            gen = res["sequences"][:, prompt.shape[1]:]
            n = min(gen.shape[1], draft.shape[1])
            first = [int(np.argmax(np.append(gen[b, :n] != draft[b, :n], True))) for b in range(B)]
            out["draft"] = {"offered": int(n_draft) * B, "accepted": min(first) * B, "launches": 0, "rounds": 0}
        return out

    def token_timestamps(self, B, n_prompt, seq_len, num_frames=None, time_precision=0.02):
        self.calls["dtw"] += 1
        cols = None
        if num_frames is not None:   # the C ABI's rule (include/thewhisper.h): ONE Python-slice crop `[: n // 2]` per row
            T = self._cross.shape[-1]
            cols = [min(T, int(n) // 2) if int(n) // 2 >= 0 else max(0, T + int(n) // 2) for n in num_frames]
        return wo.token_timestamps(self._cross[:B, :, : seq_len - 1], n_prompt, None, time_precision, columns=cols)


def oracle_engine_factory(dims, T, max_batch, dtype, alignment_heads, device_index):
    return OracleEngine(dims, T, max_batch, dtype, alignment_heads, device_index)
