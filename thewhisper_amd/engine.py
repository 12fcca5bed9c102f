"""``WhisperEngine`` - thin Python handle on a ``tw_ctx`` (include/thewhisper.h).

PyTorch is used only as plumbing here: it owns the caller-side device tensors (PCM, mel, logits)
and hands raw device pointers + the current HIP stream to the C ABI.  Every operation of the hot
path runs in libthewhisper_gfx950.so.
"""
from __future__ import annotations

import os

import ctypes as C
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _cabi

_TORCH2TW = {torch.float32: _cabi.TW_F32, torch.bfloat16: _cabi.TW_BF16, torch.float16: _cabi.TW_F16}


def _stream_ptr(device: torch.device) -> C.c_void_p:
    s = torch.cuda.current_stream(device).cuda_stream
    return C.c_void_p(int(s) if s else None)


# Which flavour `dtype="fp8"` means (BASELINE config 5): "fp8a16" = MXFP8 weights widened to bf16 in registers, activations not
# quantised (TW_BF16_W8A16: no activation-quantisation error, up to 64 streams); "fp8a8" = weights AND activations on the scaled fp8
# MFMA (TW_BF16_MXFP8, <= 16 streams).  THEWHISPER_FP8=a8|a16 overrides.
FP8_DEFAULT = {"a8": "fp8a8", "a16": "fp8a16"}.get(os.environ.get("THEWHISPER_FP8", "").lower(), "fp8a16")


class WhisperEngine:
    """One MI355X context: weights + workspace + KV arenas for up to ``max_batch`` concurrent streams
    of ``T`` encoder frames (= 50 x chunk seconds)."""

    #: optional raw ``hipStream_t`` (int) to enqueue on instead of torch's current stream, e.g. a stream created with
    #: ``hipExtStreamCreateWithCUMask`` to confine this context to a subset of the CUs
    raw_stream: Optional[int] = None

    def _sp(self) -> C.c_void_p:
        if self.raw_stream is not None:
            return C.c_void_p(self.raw_stream)
        return _stream_ptr(self.device)

    def _decode_stream(self) -> Optional[int]:
        """The stream the greedy loop runs on when the caller does not manage streams (``raw_stream`` is None): confined to 160 of
        the 256 compute units.  Alone on the chip the loop is FASTER there than on all of them - 16 streams 1.3415 vs 1.3937 ms per
        step (192 CUs: 1.3564; 128: 1.664), one stream 1.0615 vs 1.0724 (tools/dbg/decode_cu_mask.py,
        profiles/r04_decode_cu_mask.txt): its launches have 160 or 320 workgroups.  THEWHISPER_DECODE_CUS=0 turns it off."""
        d = self.__dict__
        if "_dec_stream" not in d:
            ds = None
            try:
                n = int(os.environ.get("THEWHISPER_DECODE_CUS", "160"))
                total = torch.cuda.get_device_properties(self.device).multi_processor_count
                if 0 < n < total:
                    from .overlap import masked_stream

                    ds = masked_stream(0, n, total, self.device.index or 0)
            except Exception:  # noqa: BLE001  (no CU masks on this runtime: the loop runs on the caller's stream)
                ds = None
            d["_dec_stream"] = ds
        return d["_dec_stream"]

    def _adopt(self, *tensors) -> None:
        """With ``raw_stream`` set the work runs on a stream torch's caching allocator knows nothing about: order that
        stream after torch's current stream (which produced the inputs: H2D copies, slicing, casts) and keep every tensor
        handed to the library alive until the stream has been synchronised (``_release_held``), so that a temporary dropped
        on return is not recycled while the launches still read or write it.  (``Tensor.record_stream`` is not used: the
        allocator would touch the foreign stream when the tensor is freed - possibly after the stream was destroyed.)"""
        if self.raw_stream is None:
            return
        ext = torch.cuda.ExternalStream(self.raw_stream, device=self.device)
        ext.wait_stream(torch.cuda.current_stream(self.device))
        held = self.__dict__.setdefault("_held", [])
        if len(held) > 256:      # only asynchronous calls for a long time: drain rather than grow without bound
            ext.synchronize()
            held.clear()
        held.extend(t for t in tensors if t is not None and t.is_cuda)

    def _release_held(self) -> None:
        """Called after an entry point that synchronised the stream in use (greedy decode, timestamps, weight upload)."""
        held = self.__dict__.get("_held")
        if held:
            held.clear()

    def _publish(self) -> None:
        """Counterpart of ``_adopt`` for tensors RETURNED to the caller: torch's current stream waits for the foreign stream,
        so that the caller may consume the result with ordinary torch ops."""
        if self.raw_stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(torch.cuda.ExternalStream(self.raw_stream, device=self.device))

    def _sync_used_stream(self) -> None:
        if self.raw_stream is not None:
            torch.cuda.ExternalStream(self.raw_stream, device=self.device).synchronize()
        else:
            torch.cuda.current_stream(self.device).synchronize()
        self._release_held()


    def __init__(
        self,
        dims: Dict[str, int],
        T: int,
        max_batch: int = 1,
        dtype: str = "bf16",
        alignment_heads: Optional[Sequence[Tuple[int, int]]] = None,
        device: int = 0,
        use_graph: bool = True,
    ):
        if not torch.cuda.is_available():
            raise RuntimeError("thewhisper_amd needs an MI355X (torch.cuda is not available); there is no CPU fallback")
        self.lib = _cabi.load_library()
        self.dims = dict(dims)
        self.T = int(T)
        self.max_batch = int(max_batch)
        self.device = torch.device("cuda", device)
        # "fp8" = bf16 activations and encoder, decoder projection weights quantised to MXFP8 at load (BASELINE config 5)
        # "fp8a16" = the same fp8 weights widened to bf16 in registers, activations NOT quantised (up to 64 streams); "fp8a8" =
        # weights and activations on the scaled fp8 MFMA (<= 16 streams); "fp8" = the flavour config 5 ships with (FP8_DEFAULT)
        if dtype == "fp8":
            dtype = FP8_DEFAULT
        self.dtype_name = dtype
        # "f16" = float16 context (round 4): the reference's streaming default dtype; same kernels instantiated for _Float16
        self.tw_dtype = {"bf16": _cabi.TW_BF16, "f16": _cabi.TW_F16, "f32": _cabi.TW_F32, "fp8a8": _cabi.TW_BF16_MXFP8,
                         "fp8a16": _cabi.TW_BF16_W8A16}[dtype]
        self.torch_dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "fp8a8": torch.bfloat16,
                            "fp8a16": torch.bfloat16}[dtype]
        self.alignment_heads = [tuple(map(int, x)) for x in (alignment_heads or [])]
        cfg = _cabi.tw_config()
        cfg.d_model = dims["d_model"]
        cfg.enc_layers = dims["enc_layers"]
        cfg.dec_layers = dims["dec_layers"]
        cfg.heads = dims["heads"]
        cfg.ffn = dims["ffn"]
        cfg.vocab = dims["vocab"]
        cfg.n_mels = dims["n_mels"]
        cfg.source_positions = self.T
        cfg.target_positions = dims.get("max_target_positions", 448)
        cfg.max_batch = self.max_batch
        cfg.dtype = self.tw_dtype
        cfg.n_align_heads = len(self.alignment_heads)
        for i, (l, h) in enumerate(self.alignment_heads):
            cfg.align_heads[2 * i] = l
            cfg.align_heads[2 * i + 1] = h
        cfg.device = device
        cfg.use_graph = 1 if use_graph else 0
        self.P = cfg.target_positions
        self.vocab = cfg.vocab
        self.n_mels = cfg.n_mels
        self.d_model = cfg.d_model
        ctx = C.c_void_p()
        rc = self.lib.tw_create(C.byref(cfg), C.byref(ctx))
        if rc != 0:
            raise RuntimeError(f"tw_create failed ({rc}): {self.lib.tw_last_error(None).decode()}")
        self.ctx = ctx
        self._finalized = False

    def sibling(self, max_batch: Optional[int] = None) -> "WhisperEngine":
        """A second context of the same model on the same device that SHARES this engine's (finalized) weights and has its own
        workspace and K/V arenas (tw_create_sibling): two stages of a serving pipeline can then run at the same time - a context
        is not thread-safe, two contexts are independent.  This engine must outlive the sibling."""
        if not self._finalized:
            raise RuntimeError("sibling() needs loaded weights")
        new = object.__new__(type(self))
        new.__dict__.update({k: v for k, v in self.__dict__.items() if k not in ("ctx", "_held", "raw_stream", "_owner", "_dec_stream")})
        new.max_batch = int(max_batch or self.max_batch)
        ctx = C.c_void_p()
        rc = self.lib.tw_create_sibling(self.ctx, new.max_batch, C.byref(ctx))
        if rc != 0:
            raise RuntimeError(f"tw_create_sibling failed ({rc}): {self.lib.tw_last_error(None).decode()}")
        new.ctx = ctx
        new._owner = self           # keeps the weights' owner alive
        new._siblings = []
        self.__dict__.setdefault("_siblings", []).append(new)
        return new

    # ---- lifetime ----------------------------------------------------------------------------
    def close(self):
        for sib in self.__dict__.get("_siblings", []):   # they read this context's weights: they go first
            sib.close()
        self.__dict__["_siblings"] = []
        ds = self.__dict__.pop("_dec_stream", None)
        if ds is not None:
            try:
                from .overlap import _hiplib

                _hiplib().stream_destroy(ds)     # synchronises first
            except Exception:  # noqa: BLE001
                pass
        if getattr(self, "ctx", None):
            self.lib.tw_destroy(self.ctx)
            self.ctx = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int, what: str):
        _cabi.check(self.lib, self.ctx, rc, what)

    # ---- weights -----------------------------------------------------------------------------
    def load_weight(self, name: str, tensor: torch.Tensor):
        t = tensor.detach()
        if t.device.type != "cuda":
            t = t.to(self.device, non_blocking=False)
        t = t.contiguous()
        if t.dtype not in _TORCH2TW:
            t = t.float()
        shape = (C.c_int64 * t.dim())(*t.shape)
        self._adopt(t)
        rc = self.lib.tw_load_weight(self.ctx, name.encode(), C.c_void_p(t.data_ptr()), _TORCH2TW[t.dtype], t.dim(), shape,
                                     self._sp())
        self._chk(rc, f"tw_load_weight({name})")
        self._sync_used_stream()  # `t` may be a temporary

    def finalize(self):
        self._chk(self.lib.tw_finalize_weights(self.ctx, self._sp()), "tw_finalize_weights")
        self._finalized = True

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        for k, v in sd.items():
            if k == "proj_out.weight":
                continue
            self.load_weight(k, v)
        self.finalize()

    @classmethod
    def from_numpy_weights(cls, dims, weights: Dict[str, np.ndarray], T: int, **kw) -> "WhisperEngine":
        eng = cls(dims, T, **kw)
        eng.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights.items()})
        return eng

    # ---- A1 ----------------------------------------------------------------------------------
    def logmel(self, pcm: torch.Tensor, n_valid: Optional[Sequence[int]] = None, n_samples: Optional[int] = None,
               out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        """pcm: float32 CUDA tensor [B, n] -> [B, n_mels, n_samples//160] (zero padded to ``n_samples``)."""
        if pcm.dim() == 1:
            pcm = pcm[None]
        pcm = pcm.to(self.device, torch.float32).contiguous()
        B, n = pcm.shape
        n_samples = int(n_samples or n)
        if n_valid is None and n < n_samples:
            n_valid = [n] * B
        nv = None
        if n_valid is not None:
            nv = (C.c_int32 * B)(*[int(min(v, n)) for v in n_valid])
        out_dtype = out_dtype or self.torch_dtype
        out = torch.empty((B, self.n_mels, n_samples // 160), dtype=out_dtype, device=self.device)
        self._adopt(pcm, out)
        rc = self.lib.tw_logmel(self.ctx, C.c_void_p(pcm.data_ptr()), pcm.stride(0), nv, B, n_samples,
                                C.c_void_p(out.data_ptr()), _TORCH2TW[out_dtype], self._sp())
        self._chk(rc, "tw_logmel")
        self._publish()
        return out

    # ---- A2-A5 -------------------------------------------------------------------------------
    def encode(self, mel: torch.Tensor, return_hidden: bool = False, hidden_dtype: torch.dtype = torch.float32, slot0: int = 0):
        """Encoder for the B clips of ``mel`` into slots ``slot0 .. slot0+B-1`` of the context (``slot0 > 0``: the slots
        before it keep what earlier calls of this pass put there - ``tw_encode_at``)."""
        mel = mel.to(self.device).contiguous()
        if mel.dtype not in _TORCH2TW:
            mel = mel.float()
        B = mel.shape[0]
        if tuple(mel.shape[1:]) != (self.n_mels, 2 * self.T):
            # same failure mode as HF:models/whisper/modeling_whisper.py:612-617
            raise ValueError(
                f"Whisper expects the mel input features to be of length {2 * self.T}, but found {mel.shape[-1]} "
                f"(shape {tuple(mel.shape)})")
        out = None
        if return_hidden:
            out = torch.empty((B, self.T, self.d_model), dtype=hidden_dtype, device=self.device)
        self._adopt(mel, out)
        if slot0:
            if return_hidden:
                raise ValueError("return_hidden is only available for slot0 = 0")
            rc = self.lib.tw_encode_at(self.ctx, C.c_void_p(mel.data_ptr()), _TORCH2TW[mel.dtype], B, int(slot0), self._sp())
            self._chk(rc, "tw_encode_at")
            return None
        rc = self.lib.tw_encode(self.ctx, C.c_void_p(mel.data_ptr()), _TORCH2TW[mel.dtype], B,
                                C.c_void_p(out.data_ptr()) if out is not None else None,
                                _TORCH2TW[hidden_dtype], self._sp())
        self._chk(rc, "tw_encode")
        if out is not None:
            self._publish()
        return out

    def cross_kv(self, B: int, slot0: int = 0):
        if slot0:
            self._chk(self.lib.tw_cross_kv_at(self.ctx, B, int(slot0), self._sp()), "tw_cross_kv_at")
        else:
            self._chk(self.lib.tw_cross_kv(self.ctx, B, self._sp()), "tw_cross_kv")

    def adopt_cross_kv(self, src: "WhisperEngine", src_slot0: int, B: int, dst_slot0: int):
        """Slots ``src_slot0 .. +B`` of ``src`` (another context of the same weights, e.g. one that encoded new arrivals on a
        CU-masked side stream while this one was decoding) become slots ``dst_slot0 .. +B`` of this context (tw_adopt_cross_kv)."""
        self._chk(self.lib.tw_adopt_cross_kv(self.ctx, int(dst_slot0), src.ctx, int(src_slot0), int(B), self._sp(), src._sp()),
                  "tw_adopt_cross_kv")

    # ---- A6-A8 (teacher-forced stepping, used by the parity tests) ---------------------------
    def decoder_reset(self, B: int):
        self._chk(self.lib.tw_decoder_reset(self.ctx, B, self._sp()), "tw_decoder_reset")

    def decode_step(self, ids: Sequence[int], want_logits: bool = True) -> Optional[torch.Tensor]:
        B = len(ids)
        arr = (C.c_int32 * B)(*[int(i) for i in ids])
        out = torch.empty((B, self.vocab), dtype=torch.float32, device=self.device) if want_logits else None
        self._adopt(out)
        rc = self.lib.tw_decode_step(self.ctx, B, arr, C.c_void_p(out.data_ptr()) if out is not None else None,
                                     self._sp())
        self._chk(rc, "tw_decode_step")
        if out is not None:
            self._publish()
        return out

    # ---- A9/A10 ------------------------------------------------------------------------------
    def generate_greedy(
        self,
        prompt: np.ndarray,
        max_new_tokens: int = 128,
        min_new_tokens: int = 0,
        max_length: int = 448,
        eos_id: int = 50257,
        pad_id: int = 50257,
        timestamps: bool = False,
        no_timestamps_id: int = 50364,
        max_initial_timestamp_index: Optional[int] = 50,
        begin_suppress: Iterable[int] = (220, 50257),
        suppress: Iterable[int] = (),
        want_alignment: bool = False,
        n_forced: int = 0,
        n_draft: int = 0,
    ) -> Dict[str, np.ndarray]:
        """``n_forced``: the last ``n_forced`` tokens of every prompt row are forced OUTPUT tokens (they count as generated; a
        batched prefill processes them - tw_greedy_opts::n_forced); the returned sequences contain them like generated ones.
        ``n_draft``: the last ``n_draft`` tokens of every prompt row are GUESSES of the output (tw_greedy_opts::n_draft): verified in
        batched launches, the call returns what it returns without them; ``draft`` in the result says how many were confirmed."""
        prompt = np.ascontiguousarray(prompt, dtype=np.int32)
        B, n0 = prompt.shape
        o = _cabi.tw_greedy_opts()
        o.eos_id, o.pad_id = int(eos_id), int(pad_id)
        o.max_new_tokens, o.min_new_tokens, o.max_length = int(max_new_tokens), int(min_new_tokens), int(max_length)
        o.timestamps = 1 if timestamps else 0
        o.no_timestamps_id = int(no_timestamps_id)
        o.max_initial_timestamp_index = -1 if max_initial_timestamp_index is None else int(max_initial_timestamp_index)
        bs = [int(x) for x in begin_suppress]
        sp = [int(x) for x in suppress]
        bs_arr = (C.c_int32 * max(1, len(bs)))(*bs)
        sp_arr = (C.c_int32 * max(1, len(sp)))(*sp)
        o.n_begin_suppress, o.begin_suppress = len(bs), bs_arr
        o.n_suppress, o.suppress = len(sp), sp_arr
        o.want_alignment = 1 if want_alignment else 0
        o.n_forced = int(n_forced)
        o.n_draft = int(n_draft)
        out = np.full((B, int(max_length)), pad_id, dtype=np.int32)
        out_len = C.c_int32(0)
        sp = self._sp()
        # (calls with a forced prefix stay on the caller's stream: their batched prefill - launches of up to 64 rows - wants the whole
        #  chip; measured on the reuse path's short calls: 16.1 ms per tick there, 19.9 on the 160-CU stream)
        if self.raw_stream is None and int(n_forced) == 0 and int(n_draft) == 0:
            ds = self._decode_stream()
            if ds is not None:      # behind everything enqueued so far (the encoder stage); the call synchronises it before returning
                torch.cuda.ExternalStream(ds, device=self.device).wait_stream(torch.cuda.current_stream(self.device))
                sp = C.c_void_p(ds)
        rc = self.lib.tw_generate_greedy(self.ctx, B, prompt.ctypes.data_as(C.POINTER(C.c_int32)), n0, C.byref(o),
                                         out.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(out_len),
                                         sp)
        self._chk(rc, "tw_generate_greedy")
        self._release_held()   # the call returns after synchronising its stream, which is ordered after the encoder stage
        L = int(out_len.value)
        res = {"sequences": out[:, :L].astype(np.int64), "length": L}
        if int(n_draft) > 0:
            v = [C.c_int32(0) for _ in range(4)]
            self._chk(self.lib.tw_last_draft(self.ctx, *[C.byref(x) for x in v]), "tw_last_draft")
            res["draft"] = {"offered": int(v[0].value), "accepted": int(v[1].value), "launches": int(v[2].value), "rounds": int(v[3].value)}
        return res

    # ---- A11 ---------------------------------------------------------------------------------
    def token_timestamps(self, B: int, n_prompt: int, seq_len: int, num_frames: Optional[Sequence[int]] = None,
                         time_precision: float = 0.02) -> np.ndarray:
        out = np.zeros((B, seq_len), dtype=np.float32)
        nf = None
        if num_frames is not None:
            nf = (C.c_int32 * B)(*[int(x) for x in num_frames])
        rc = self.lib.tw_token_timestamps(self.ctx, B, n_prompt, seq_len, nf, float(time_precision),
                                          out.ctypes.data_as(C.POINTER(C.c_float)), self._sp())
        self._chk(rc, "tw_token_timestamps")
        return out

    def get_alignment(self, B: int, n_rows: int) -> np.ndarray:
        out = np.zeros((B, len(self.alignment_heads), n_rows, self.T), dtype=np.float32)
        rc = self.lib.tw_get_alignment(self.ctx, B, n_rows, out.ctypes.data_as(C.POINTER(C.c_float)),
                                       self._sp())
        self._chk(rc, "tw_get_alignment")
        return out

    def last_timings(self) -> Dict[str, float]:
        ms = (C.c_float * 5)()
        steps = C.c_int32(0)
        self._chk(self.lib.tw_last_timings(self.ctx, ms, C.byref(steps)), "tw_last_timings")
        names = ["logmel_ms", "encode_ms", "cross_kv_ms", "greedy_ms", "token_timestamps_ms"]
        d = {n: float(ms[i]) for i, n in enumerate(names)}
        d["decode_steps"] = int(steps.value)
        return d
