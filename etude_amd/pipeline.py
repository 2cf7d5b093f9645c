"""The per-song stage sequence of infer.py, for a BATCH of clips on one GPU (BASELINE configs[4]: 64 clips x 27 attribute tuples).

What ``InferencePipeline.run`` does for one song and one attribute tuple (/root/reference/infer.py:82-104, 165-207):

    stage 1  extractor.extract(origin.wav -> extract.json)                      infer.py:82-97
             analyze_volume(origin.wav -> volume.json)                           infer.py:99-104
    stage 2  beat detection -> tempo.json   (OUT OF SCOPE: Spleeter / Beat-Transformer / madmom; the caller supplies tempo.json)
    stage 3  TinyREMITokenizer(tempo.json).encode(extract.json) -> vocab.encode_sequence -> split_sequence_into_bars
             -> model.generate(bars, attributes) -> tokenizer.decode_to_notes(events, volume.json) -> note_to_midi   infer.py:180-207

here runs once per CLIP for stages 1 / tokenize and once per (clip, attribute tuple) JOB for generate .. notes, with the notes,
volume maps, bars and token ids handed from stage to stage in memory as arrays (no JSON round trip, no per-note Python objects).
Every step is the library's own entry point -- `AMTAPC_Extractor.extract_note_array`, `VolumeAnalyzer`, `TinyREMITokenizer`'s array
paths, `decoder.run_engines`, `decode_ids_to_note_array` -- so a job's result equals what the stage-by-stage calls of the reference's
surface give (tests/test_gpu_pipeline.py).  bench.py times exactly this.
"""
from __future__ import annotations

import threading
import time
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .decoder import ABI_ATTR_KEYS, PackedBars, run_engines
from .preprocess import VolumeAnalyzer
from .tokenizer import TinyREMITokenizer


def attr_grid(n: int = 27, overlap: int = 2) -> List[Dict[str, int]]:
    """The attribute tuples of configs[4]: (polyphony, rhythm_intensity, sustain) in {0,1,2}^3, pitch overlap fixed (infer.py:296);
    n == 1 -> the CLI defaults (1, 1, 1)."""
    if n == 1:
        return [dict(polyphony_bin=1, rhythm_intensity_bin=1, sustain_bin=1, pitch_overlap_bin=overlap)]
    g = [dict(polyphony_bin=p, rhythm_intensity_bin=r, sustain_bin=s, pitch_overlap_bin=overlap) for p in range(3) for r in range(3) for s in range(3)]
    return g[:n]


def synthetic_tempo(n_downbeats: int = 90, bpm: float = 120.0, start: float = 0.5) -> List[dict]:
    """tempo.json of SURVEY.md 8(d) config 1 (stage 2 is out of scope): one region, 4/4, `n_downbeats` downbeats from `start`."""
    return [{"start": start, "bpm": bpm, "time_sig": 4, "downbeats": [round(start + 4 * 60.0 / bpm * i, 6) for i in range(n_downbeats)]}]


@dataclass
class ClipConditions:
    """What stages 1-2 and the tokenizer leave behind for one clip."""
    notes: np.ndarray                       # NOTE_DTYPE: what extract.json lists
    volume: np.ndarray                      # volume.json
    bars: PackedBars                        # all_x_bars
    tokenizer: TinyREMITokenizer = field(repr=False, default=None)


class ClipBatchPipeline:
    """extract -> tokenize -> generate (every attribute tuple) -> notes, for a batch of device-resident clips.

    ``extractors``: one or more `AMTAPC_Extractor` (each transcribes different clips on its own stream + host thread);
    ``decoders``: the decoder engines (`EtudeDecoder` + `clone()`s) the jobs are dealt over; ``tempo``: tempo.json content shared
    by the clips (list of regions) or one per clip."""

    def __init__(self, extractors, decoders, vocab, tempo, attrs: Sequence[Dict[str, int]], sample_rate: int = 44100,
                 force_bar_tokens: int = 0, temperature: float = 0.0, post_workers: int = 8, **generate_kwargs):
        self.exs = list(extractors)
        self.decs = list(decoders)
        self.vocab = vocab
        self.tempo = tempo
        self.attrs = list(attrs)
        self.sr = int(sample_rate)
        self.force_bar_tokens = int(force_bar_tokens)
        self.temperature = float(temperature)
        self.generate_kwargs = generate_kwargs
        self.dev = self.exs[0].device
        self.ex_streams = [torch.cuda.Stream(device=self.dev) for _ in self.exs]
        self.vols = [VolumeAnalyzer(self.sr, device=self.dev) for _ in self.exs]
        self.lut = TinyREMITokenizer.id_lookup(vocab)
        self.table = TinyREMITokenizer.event_table(vocab)
        self.bos, self.eos = vocab.get_bar_bos_id(), vocab.get_bar_eos_id()
        self.min_duration = self.exs[0].config.infer.min_duration
        self.post_workers = max(1, int(post_workers))
        self._a4 = [np.asarray([a[k] for k in ABI_ATTR_KEYS], np.int32) for a in self.attrs]

    # ------------------------------------------------------------------ stage 1 (+ tokenizer): once per clip
    def _tempo_of(self, c: int):
        t = self.tempo
        return t[c] if t and isinstance(t[0], list) else t

    def conditions_of(self, wav: torch.Tensor, c: int = 0, engine: int = 0) -> ClipConditions:
        ex, vol = self.exs[engine], self.vols[engine]
        notes = ex.extract_note_array(wav, self.sr, self.min_duration)          # extractor.extract -> extract.json (infer.py:90-96)
        volume = vol(wav)                                                       # analyze_volume -> volume.json (infer.py:99-104)
        tk = TinyREMITokenizer.from_tempo_data(self._tempo_of(c))               # TinyREMITokenizer(tempo.json) (infer.py:180)
        ev = tk.encode_note_array_to_events(notes)                              # .encode(extract.json)
        ids = tk.events_to_ids(ev, self.lut)                                    # vocab.encode_sequence
        bi, bo = tk.split_ids_into_packed_bars(ids, self.bos, self.eos)         # split_sequence_into_bars
        return ClipConditions(notes, volume, PackedBars(bi, bo), tk)

    def extract_stage(self, wavs: Sequence[torch.Tensor]) -> List[ClipConditions]:
        out: List[Optional[ClipConditions]] = [None] * len(wavs)
        errs: list = []
        n = len(self.exs)

        def run(i):
            try:
                torch.cuda.set_device(self.dev)
                with torch.cuda.stream(self.ex_streams[i]):
                    for c in range(i, len(wavs), n):
                        out[c] = self.conditions_of(wavs[c], c, i)
                self.ex_streams[i].synchronize()
            except Exception as e:      # noqa: BLE001
                errs.append(e)
        if n == 1:
            run(0)
        else:
            th = [threading.Thread(target=run, args=(i,)) for i in range(n)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        if errs:
            raise errs[0]
        return out  # type: ignore[return-value]

    # ------------------------------------------------------------------ stage 3: once per (clip, attribute tuple)
    def jobs_of(self, conds: Sequence[ClipConditions], max_bars: int = 0):
        jobs = []
        for cd in conds:
            pb = cd.bars
            if max_bars and len(pb) > max_bars:
                pb = PackedBars(pb.ids[: pb.offsets[max_bars]], pb.offsets[: max_bars + 1])
            for a4 in self._a4:
                jobs.append((pb, np.tile(a4, (len(pb), 1))))
        return jobs

    def decode_stage(self, conds: Sequence[ClipConditions], max_bars: int = 0, one_at_a_time: bool = False):
        """-> (per job (flat ids, bar lengths), per-engine stats)"""
        jobs = self.jobs_of(conds, max_bars)
        join = run_engines(self.decs, jobs, self.vocab, one_at_a_time=one_at_a_time, force_bar_tokens=self.force_bar_tokens,
                           temperature=self.temperature, as_arrays=True, **self.generate_kwargs)
        return join()

    def notes_stage(self, conds: Sequence[ClipConditions], results) -> List[np.ndarray]:
        """decode_to_notes of every job (tokenizer.py:446-496 with the clip's volume map); note_to_midi is the caller's"""
        na = len(self._a4)

        def one(j):
            cd = conds[j // na]
            return cd.tokenizer.decode_ids_to_note_array(results[j][0], self.table, cd.volume, pad_id=self.vocab.get_pad_id())
        if self.post_workers == 1 or len(results) < 4:
            return [one(j) for j in range(len(results))]
        with ThreadPoolExecutor(self.post_workers) as pool:
            return list(pool.map(one, range(len(results))))

    def run(self, wavs: Sequence[torch.Tensor]) -> dict:
        t0 = time.perf_counter()
        conds = self.extract_stage(wavs)
        t1 = time.perf_counter()
        results, stats = self.decode_stage(conds)
        torch.cuda.synchronize(self.dev)
        t2 = time.perf_counter()
        notes = self.notes_stage(conds, results)
        t3 = time.perf_counter()
        return dict(conditions=conds, results=results, notes=notes, stats=stats, t_extract=t1 - t0, t_decode=t2 - t1, t_notes=t3 - t2,
                    tokens=sum(s["tokens"] for s in stats))

    def close(self):
        for v in self.vols:
            v.close()
