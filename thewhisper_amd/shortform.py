"""Whisper's short-form control flow, restated for the one configuration the reference's backends use, so that the
unit of batching becomes one *seek pass of one chunk* instead of one ``generate()`` call.

What it replaces.  ``WhisperGenerationMixin.generate`` (HF:models/whisper/generation_whisper.py:383-968) loops per call:
cut the segment at ``seek`` (:1831-1850), run the greedy decoder, slice the ids at the timestamp tokens
(``_retrieve_segment``, :1977-2074), advance ``seek``, repeat until the chunk is consumed, pad the batch
(``_pad_to_max_length``, :126-237).  On a batch the later iterations run with ever fewer rows (:785-795), and a
weight-streaming decode step costs the same for 4 rows as for 16 - which is where the serving front end lost a quarter
of the engine's throughput (profiles/r02_SUMMARY.md).  Here the per-chunk state (``ChunkWork``) is explicit and
``run_pass`` advances ANY set of chunks by one pass, whichever call or session they came from:

* ``generate_shortform``  - same inputs / same return value as HF's ``generate`` for a batch (used by
  ``AMDWhisperForConditionalGeneration.generate`` when the call is eligible), and
* ``thewhisper_amd.serving.BatchingHub`` keeps a pool of works from all sessions and fills every pass.

What is NOT restated: how a call's options become init tokens and logits processors (language / task tokens, suppress
lists, timestamp grammar switches - HF:...:1455-1608, :1774-1812).  That is *learned*: the first call with a given
set of options runs HF's own ``generate``; ``_EngineGreedyMixin.generate`` (model.py) records the decoder prompt and the
engine options it was handed, and ``ShortFormPlan`` replays exactly those.  Calls that are not eligible (prompt ids,
temperature fallback, thresholds, language detection, long-form input ...) keep going through HF's code.

Results are identical to HF's control flow by construction of every step (same torch ops in the same order for the
floating-point bits: time offsets, token-timestamp offsets) and by test: tests/test_shortform.py compares against HF's
``generate`` on the CPU stand-in engine, and the byte-identical pipeline goldens (reference outputs) run through it.
"""
from __future__ import annotations

import dataclasses
import time
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


MAX_SEEK_PASSES = 64    # a 30 s chunk advances by >= 1 timestamp step (0.02 s) per useful pass; far more than any real transcript needs


@dataclasses.dataclass
class ShortFormPlan:
    """Everything one pass needs that does not depend on the audio (learned from one HF-driven call)."""
    init_tokens: Tuple[int, ...]            # decoder prompt of every row (sot, language, task[, notimestamps])
    greedy: Dict[str, Any]                  # keyword arguments of WhisperEngine.generate_greedy (without the prompt)
    eos: int
    pad: int
    timestamp_begin: int
    return_timestamps: bool
    return_token_timestamps: bool
    return_segments: bool
    result_is_dict: bool                    # what the inner generate returned: GenerateEncoderDecoderOutput or a tensor
    time_precision: float = 0.02
    time_precision_features: float = 0.01
    input_stride: int = 2

    @property
    def n_prompt(self) -> int:
        return len(self.init_tokens)


class ChunkWork:
    """Decoding state of one <= chunk_length_s piece of audio (one row of a ``generate`` batch)."""

    __slots__ = ("feats", "num_frames", "max_frames", "seek", "segments", "passes", "tag", "forced", "draft", "draft_result", "first_pass")

    def __init__(self, feats: torch.Tensor, num_frames: Optional[int], tag: Any = None):
        # SURVEY.md section 8f-3 (opt-in, streaming.py): output tokens of the FIRST seek iteration that are already known - they
        # are handed to the greedy loop as forced output (batched prefill) and the loop decodes only what follows
        self.forced: Optional[np.ndarray] = None
        # ... or GUESSED (tw_greedy_opts::n_draft): verified by the engine in batched launches, the iteration's result is what it is
        # without them; ``draft_result`` = the engine's report (tokens offered / confirmed, launches, rounds)
        self.draft: Optional[np.ndarray] = None
        self.draft_result: Optional[Dict[str, int]] = None
        self.first_pass: Optional[Tuple[np.ndarray, np.ndarray]] = None   # (ids after the prompt, their timestamps) of iteration 1
        self.feats = feats                      # [n_mels, frames] log-mel (device tensor); frames = 2 * T for short-form
        self.num_frames = num_frames            # frames of real audio (attention_mask.sum), None if no mask was given
        self.max_frames = int(feats.shape[-1])  # HF:...:1769 - short-form: the padded feature length, not the audio length
        self.seek = 0
        self.segments: List[Dict[str, Any]] = []
        self.passes = 0
        self.tag = tag

    @property
    def done(self) -> bool:
        return self.seek >= self.max_frames


def retrieve_segment(seek_sequence: torch.Tensor, result: Any, token_timestamps, time_offset: torch.Tensor, timestamp_begin: int,
                     seek_num_frames: int, plan: ShortFormPlan, idx_offset: int) -> Tuple[List[Dict[str, Any]], int]:
    """``WhisperGenerationMixin._retrieve_segment`` (HF:models/whisper/generation_whisper.py:1977-2074) for one row.
    ``time_offset``: 0-dim float64 tensor (as HF's ``time_offset[prev_idx]``); returns (segments, frames to advance)."""
    time_precision = plan.time_precision
    timestamp_tokens = seek_sequence.ge(timestamp_begin)
    single_timestamp_ending = timestamp_tokens[-2:].tolist() == [False, True]
    timestamp_segment_indices = torch.where(timestamp_tokens[:-1] & timestamp_tokens[1:])[0]
    timestamp_segment_indices = timestamp_segment_indices + 1
    rtt = plan.return_token_timestamps
    if len(timestamp_segment_indices) > 0:
        slices = timestamp_segment_indices.tolist()
        segments = []
        if single_timestamp_ending:
            slices.append(len(seek_sequence))
        else:
            slices[-1] += 1   # keep the last timestamp token in the last segment: it was no single ending
        last_slice = 0
        for i, current_slice in enumerate(slices):
            is_last_slice = i == len(slices) - 1
            sliced_tokens = seek_sequence[last_slice:current_slice]
            start_timestamp_pos = sliced_tokens[0] - timestamp_begin
            idx_sliced_tokens = -1 if not is_last_slice or single_timestamp_ending else -2
            end_timestamp_pos = sliced_tokens[idx_sliced_tokens] - timestamp_begin
            seg = {
                "start": time_offset + start_timestamp_pos.to(torch.float64) * time_precision,
                "end": time_offset + end_timestamp_pos.to(torch.float64) * time_precision,
                "tokens": sliced_tokens,
                "idxs": (idx_offset + last_slice, idx_offset + current_slice),
                "result": result,
            }
            if rtt:
                seg["token_timestamps"] = token_timestamps[idx_offset + last_slice : idx_offset + current_slice] + time_offset
            segments.append(seg)
            last_slice = current_slice
        if single_timestamp_ending:
            segment_offset = seek_num_frames                      # no speech after the last timestamp
        else:
            # the unfinished tail is thrown away: seek to the last predicted "end of segment"
            last_timestamp_pos = seek_sequence[last_slice - 2].item() - timestamp_begin
            segment_offset = last_timestamp_pos * plan.input_stride
    else:
        timestamps = seek_sequence[timestamp_tokens.nonzero().flatten()]
        last_timestamp_pos: Any = int(seek_num_frames * plan.time_precision_features / time_precision)
        if timestamps.numel() > 0 and timestamps[-1] != timestamp_begin:
            last_timestamp_pos = (timestamps[-1] - timestamp_begin).to(torch.float64)
        seg = {
            "start": time_offset,
            "end": time_offset + last_timestamp_pos * time_precision,
            "tokens": seek_sequence,
            "idxs": (idx_offset, idx_offset + len(seek_sequence)),
            "result": result,
        }
        if rtt:
            seg["token_timestamps"] = token_timestamps[idx_offset : idx_offset + len(seek_sequence)] + time_offset
        segments = [seg]
        segment_offset = seek_num_frames
    return segments, int(segment_offset)


def hf_kept_columns(num_frames, batch: int, t_cols: int) -> List[int]:
    """How many leading columns of the [tokens, frames] alignment matrix HF's ``_extract_token_timestamps`` keeps per row for a
    given ``num_frames`` argument (HF:models/whisper/generation_whisper.py:310-330, :357-359).  HF crops with Python slices
    ``[..., : n // 2]`` (floor division, a negative bound counts from the end) and WHICH slices it applies depends on the type
    and uniformity of the argument: an int - one slice for the batch; equal values in a list / array / tensor - that slice for
    the batch AND the per-row slice again (a no-op for n >= 0, a SECOND removal of |n // 2| columns for n < 0, which happens
    when a seek iteration starts past the end of a clip shorter than the chunk: ``num_frames - seek < 0``, :1152-1155);
    different values - the per-row slice only.  With no column left HF's DTW walks the token axis at time index -1.
    The engine (tw_token_timestamps) crops ONCE per row; this function gives it the bound that reproduces HF's result."""
    def crop(length: int, k: int) -> int:
        return min(length, k) if k >= 0 else max(0, length + k)

    if num_frames is None:
        return [t_cols] * batch
    if isinstance(num_frames, (int, np.integer)):
        return [crop(t_cols, int(num_frames) // 2)] * batch
    nf = [int(x) for x in (num_frames.tolist() if hasattr(num_frames, "tolist") else list(num_frames))]
    if len(nf) != batch:
        nf = [int(x) for x in np.repeat(nf, batch // len(nf))]
    if len(set(nf)) == 1:
        k = nf[0] // 2
        return [crop(crop(t_cols, k), k)] * batch
    return [crop(t_cols, n // 2) for n in nf]


def columns_as_num_frames(cols: Sequence[int]) -> List[int]:
    """The per-row ``num_frames`` of tw_token_timestamps that keep exactly ``cols[i]`` leading columns (its rule: ``[: n // 2]``)."""
    return [2 * int(c) for c in cols]


class Pass:
    """One seek iteration (HF:...:785-903) for up to ``engine.max_batch`` chunks, assembled in GROUPS: ``add(works)`` cuts the
    groups' segments and enqueues their encoder + cross-K/V stage at the next free slots (asynchronous launches), ``run()``
    decodes all slots in one greedy loop and advances every chunk.  A serving loop adds the chunks that need a further
    iteration first and whatever arrives while the GPU is still encoding those; ``run_pass`` is the one-group form."""

    def __init__(self, engine, plan: ShortFormPlan):
        self.engine, self.plan = engine, plan
        self.works: List[ChunkWork] = []
        self.snf: List[int] = []
        self._keep: List[torch.Tensor] = []       # segment tensors stay alive until the pass has run

    @property
    def free(self) -> int:
        return int(self.engine.max_batch) - len(self.works)

    def add(self, works: Sequence[ChunkWork]) -> None:
        n_new = len(works)
        if n_new == 0:
            return
        if n_new > self.free:
            raise ValueError(f"a pass takes at most {self.engine.max_batch} chunks")
        nsf = 2 * int(self.engine.T)                  # num_segment_frames = input_stride * max_source_positions (HF :652-653)
        rows = []
        for w in works:
            if w.done:
                raise ValueError("finished chunk handed to a pass")
            n = min(w.max_frames - w.seek, nsf)
            self.snf.append(n)
            s = w.feats[:, w.seek : w.seek + n]
            if n < nsf:
                s = F.pad(s, pad=(0, nsf - n))        # HF:...:1840-1844
            rows.append(s)
        segment_input = torch.stack(rows, dim=0)
        slot0 = len(self.works)
        if slot0:
            self.engine.encode(segment_input, slot0=slot0)
            self.engine.cross_kv(n_new, slot0=slot0)
        else:
            self.engine.encode(segment_input)
            self.engine.cross_kv(n_new)
        self._keep.append(segment_input)
        self.works.extend(works)

    def adopt(self, works: Sequence[ChunkWork], side_engine, side_slot0: int) -> None:
        """The next rows of the pass are chunks whose next seek iteration (``next_segment`` at their CURRENT seek) was encoded
        elsewhere - by ``side_engine``, a sibling context of the same weights, at its slots ``side_slot0 ..`` (serving.py: arrivals
        and the chunks that sit a pass out are encoded on a CU-masked stream while the running pass decodes): their cross K/V are
        copied into this pass's next slots (tw_adopt_cross_kv).  The caller guarantees the chunks have not run since."""
        n_new = len(works)
        if n_new == 0:
            return
        if n_new > self.free:
            raise ValueError(f"a pass takes at most {self.engine.max_batch} chunks")
        nsf = 2 * int(self.engine.T)
        for w in works:
            if w.done:
                raise ValueError("finished chunk handed to a pass")
            self.snf.append(min(w.max_frames - w.seek, nsf))
        self.engine.adopt_cross_kv(side_engine, side_slot0, n_new, len(self.works))
        self.works.extend(works)

    def run(self, kept_columns: Optional[Sequence[int]] = None) -> None:
        """``kept_columns``: alignment-matrix columns per row as HF would keep them for the BATCH these rows are part of
        (``generate_shortform``); None = every row is a ``generate`` call of its own (the hub: one request = one call of the
        reference backend's pipeline with batch size 1), i.e. HF's equal-values rule applies to each row by itself."""
        engine, plan, works, snf = self.engine, self.plan, self.works, self.snf
        B = len(works)
        if B < 1:
            raise ValueError("empty pass")
        n_prompt = plan.n_prompt
        prompt = np.tile(np.asarray(plan.init_tokens, dtype=np.int32), (B, 1))
        n_forced = 0
        if any(w.forced is not None and w.seek == 0 for w in works):
            # forced output prefixes (first iteration only): one length for the whole pass (the engine's loop is in lock-step)
            lens = {len(w.forced) if (w.forced is not None and w.seek == 0) else 0 for w in works}
            if len(lens) != 1:
                raise ValueError("a pass takes forced prefixes of ONE length")
            n_forced = lens.pop()
            prompt = np.concatenate([prompt, np.stack([np.asarray(w.forced, dtype=np.int32) for w in works])], axis=1)
        n_draft = 0
        if any(w.draft is not None and w.seek == 0 for w in works):
            if n_forced:
                raise ValueError("a pass takes forced prefixes OR drafts")
            lens = {len(w.draft) if (w.draft is not None and w.seek == 0) else 0 for w in works}
            if len(lens) != 1:
                raise ValueError("a pass takes drafts of ONE length")
            n_draft = lens.pop()
            prompt = np.concatenate([prompt, np.stack([np.asarray(w.draft, dtype=np.int32) for w in works])], axis=1)
        t0 = time.perf_counter()
        if n_forced:
            out = engine.generate_greedy(prompt, n_forced=n_forced, **plan.greedy)
        elif n_draft:
            out = engine.generate_greedy(prompt, n_draft=n_draft, **plan.greedy)
        else:
            out = engine.generate_greedy(prompt, **plan.greedy)
        self.greedy_s = time.perf_counter() - t0        # the engine call (blocks until the loop has finished); the rest of run() is host work
        self._keep.clear()
        seq = torch.from_numpy(np.ascontiguousarray(out["sequences"])).to(torch.long)
        L = int(seq.shape[1])
        ts = None
        if plan.return_token_timestamps:
            if L - 1 <= n_prompt:          # one generated token: no cross-attention rows after the prompt (HF :341-344)
                ts = torch.zeros((B, L), dtype=torch.float32)
            else:
                nf = None
                if kept_columns is not None:
                    nf = columns_as_num_frames(kept_columns)
                elif works[0].num_frames is not None:                          # HF:...:1152-1155: num_frames - seek
                    nf = columns_as_num_frames([hf_kept_columns([int(w.num_frames) - int(w.seek)], 1, int(engine.T))[0] for w in works])
                ts = torch.from_numpy(engine.token_timestamps(B, n_prompt, L, nf, plan.time_precision))
        for i, w in enumerate(works):
            if plan.result_is_dict:
                result: Any = {"sequences": seq[i]}
                if ts is not None:
                    result["token_timestamps"] = ts[i]
            else:
                result = seq[i]
            seek_sequence = seq[i, n_prompt:]
            # HF:...:1068-1076: drop the padding, keep one eos for the (unused here) log-prob statistics, then drop that eos too
            if seek_sequence.numel() > 0 and seek_sequence[-1] == plan.pad:
                num_paddings = int((seek_sequence == plan.pad).sum())
                if plan.pad == plan.eos:
                    num_paddings -= 1
                if num_paddings != 0:
                    seek_sequence = seek_sequence[:-num_paddings]
            if seek_sequence.numel() > 0 and seek_sequence[-1] == plan.eos:
                seek_sequence = seek_sequence[:-1]
            if w.seek == 0 and w.first_pass is None:
                n_tok = int(seek_sequence.numel())
                w.first_pass = (seek_sequence.numpy().copy(),
                                ts[i, n_prompt : n_prompt + n_tok].numpy().copy() if ts is not None else None)
            time_offset = torch.tensor(w.seek, dtype=torch.long).to(torch.float64) * plan.time_precision / plan.input_stride
            segments, offset = retrieve_segment(seek_sequence, result, ts[i] if ts is not None else [], time_offset,
                                                plan.timestamp_begin, snf[i], plan, n_prompt)
            w.seek += offset
            w.segments += segments
            w.passes += 1
            w.forced = None        # a prefix is forced ONCE: a chunk whose seek stays at 0 decodes its next iteration afresh
            if w.draft is not None:
                w.draft_result = out.get("draft")
                w.draft = None


def next_segment(work: ChunkWork, T: int) -> torch.Tensor:
    """The [n_mels, 2T] segment a chunk's NEXT seek iteration encodes (what ``Pass.add`` cuts at the chunk's current seek)."""
    nsf = 2 * int(T)
    n = min(work.max_frames - work.seek, nsf)
    s = work.feats[:, work.seek : work.seek + n]
    if n < nsf:
        s = F.pad(s, pad=(0, nsf - n))
    return s


def first_segment(work: ChunkWork, T: int) -> torch.Tensor:
    """``next_segment`` of a chunk that has not run yet (seek = 0)."""
    if work.seek != 0:
        raise ValueError("the chunk has run already")
    return next_segment(work, T)


def run_pass(engine, plan: ShortFormPlan, works: Sequence[ChunkWork], kept_columns: Optional[Sequence[int]] = None) -> None:
    """One seek iteration for every work in ``works`` (all unfinished, len <= engine.max_batch): segment cut-out, encoder +
    cross-K/V + greedy loop (+ token timestamps) on the engine, segment slicing, seek advance."""
    if len(works) < 1 or len(works) > engine.max_batch:
        raise ValueError(f"a pass takes 1..{engine.max_batch} chunks, got {len(works)}")
    p = Pass(engine, plan)
    p.add(works)
    p.run(kept_columns)


def work_tokens(plan: ShortFormPlan, w: ChunkWork) -> Tuple[torch.Tensor, Optional[torch.Tensor], Optional[torch.Tensor]]:
    """(ids of all segments, token timestamps WITHOUT the time offsets - what ``_pad_to_max_length`` concatenates -, token
    timestamps WITH the offsets - what the pipeline takes from ``segments``) of a finished chunk, unpadded."""
    if len(w.segments) > 0:
        sequence = torch.cat([d["tokens"] for d in w.segments], dim=-1)
    else:
        sequence = torch.tensor([])
    raw = seg = None
    if plan.return_token_timestamps:
        if len(w.segments) > 0:
            raw = torch.cat([d["result"]["token_timestamps"][d["idxs"][0] : d["idxs"][1]] for d in w.segments], dim=-1)
            seg = torch.cat([d["token_timestamps"] for d in w.segments])
        else:
            raw = torch.tensor([])
            seg = torch.tensor([])
    return sequence, raw, seg


def assemble(plan: ShortFormPlan, works: Sequence[ChunkWork], device) -> Any:
    """What HF's ``generate`` returns for the batch (HF:...:905-968): right-padded ids (+ token timestamps, + segments)."""
    seqs, tss = [], []
    for w in works:
        s, raw, _ = work_tokens(plan, w)
        seqs.append(s)
        tss.append(raw)
    max_total_length = max(len(s) for s in seqs)
    for i in range(len(seqs)):
        pad_length = max_total_length - len(seqs[i])
        seqs[i] = F.pad(seqs[i], pad=(0, pad_length), value=plan.pad)
        if plan.return_token_timestamps:
            tss[i] = F.pad(tss[i], pad=(0, pad_length), value=tss[i][-1] if len(tss[i]) > 0 else 0.0)
    sequences = torch.stack(seqs, dim=0).to(device)
    final_segments = [w.segments for w in works]
    if not plan.return_segments and not plan.return_token_timestamps:
        return sequences
    outputs: Dict[str, Any] = {"sequences": sequences}
    if plan.return_token_timestamps:
        outputs["token_timestamps"] = torch.stack(tss, dim=0).to(device)
    if plan.return_segments:
        outputs["segments"] = final_segments
    return outputs


def generate_shortform(engine, plan: ShortFormPlan, input_features: torch.Tensor, attention_mask: Optional[torch.Tensor]) -> Any:
    """Drop-in for ``WhisperGenerationMixin.generate`` on an eligible batch: identical return value, fewer Python layers."""
    B = int(input_features.shape[0])
    nf: List[Optional[int]] = [None] * B
    if plan.return_token_timestamps and attention_mask is not None:
        nf = [int(x) for x in attention_mask.sum(-1).cpu().tolist()]          # HF:...:1694-1695
    works = [ChunkWork(input_features[i], nf[i]) for i in range(B)]
    cap = int(engine.max_batch)
    while True:
        active = [w for w in works if not w.done]                              # HF:...:790-795 (the batch shrinks)
        if not active:
            break
        cols = None
        if plan.return_token_timestamps and active[0].num_frames is not None:
            # HF hands `(num_frames - seek)[batch_idx_map]` of the WHOLE active batch to `_extract_token_timestamps` (:1152-1155)
            cols = hf_kept_columns([int(w.num_frames) - int(w.seek) for w in active], len(active), int(engine.T))
        for i in range(0, len(active), cap):                                   # a call wider than the engine: several passes per iteration
            run_pass(engine, plan, active[i : i + cap], None if cols is None else cols[i : i + cap])
        if any(w.passes > MAX_SEEK_PASSES for w in active):
            # a decoder that keeps closing its segments at <|0.00|> never advances `seek`; HF's loop spins forever on such a row
            raise RuntimeError(f"a chunk needed more than {MAX_SEEK_PASSES} seek passes (the decoder keeps seeking to frame 0)")
    return assemble(plan, works, input_features.device)
