"""Clip-level data parallelism: one process per GPU, no collective on the data path.

The reference has no distributed code at all.  The path shards
embarrassingly by clip (and by (clip, attribute tuple) decode job): rank r of R takes items r::R,
runs extract + decode for them on its own GPU, and the only exchange is ONE final gather of the small
variable-length results (token ids, notes) over RCCL/xGMI (``torch.distributed`` backend "nccl" on
ROCm) -- latency-bound, a few MB at most.  With the gloo backend the same code runs on CPU tensors
(tests/test_parallel_gloo.py).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist


def shard(items: Sequence, rank: int, world: int) -> List:
    """Round-robin assignment: rank r takes items r, r+R, r+2R, ... (64 clips -> 8 per GPU at R=8)."""
    return list(items[rank::world])


def unshard(per_rank: Sequence[Sequence], n_items: int) -> List:
    """Inverse of ``shard`` applied on every rank: interleave back to the original order."""
    world = len(per_rank)
    out: List = [None] * n_items
    for r, lst in enumerate(per_rank):
        for j, v in enumerate(lst):
            out[r + j * world] = v
    return out


def digest_in_global_clip_order(per_rank_jobs: Sequence[Sequence[np.ndarray]], jobs_per_clip: int) -> str:
    """sha256 (first 16 hex digits) over every job's int32 ids with the CLIPS in global order: rank r holds the jobs of clips r, r + R, ... (``shard``), ``jobs_per_clip``
    consecutive jobs per clip.  The value does not depend on R -- what bench.py prints as ``tokens_sha256_all`` (a job's ids do not depend on what shares its launches)."""
    import hashlib
    per_rank_clips = []
    for lst in per_rank_jobs:
        assert len(lst) % max(1, jobs_per_clip) == 0, "every rank holds whole clips"
        per_rank_clips.append([np.concatenate([np.asarray(x, np.int32).reshape(-1) for x in lst[c * jobs_per_clip:(c + 1) * jobs_per_clip]]) if jobs_per_clip else np.zeros(0, np.int32)
                               for c in range(len(lst) // max(1, jobs_per_clip))])
    clips = unshard(per_rank_clips, sum(len(x) for x in per_rank_clips))
    flat = np.concatenate(clips).astype(np.int32) if clips else np.zeros(0, np.int32)
    return hashlib.sha256(flat.tobytes()).hexdigest()[:16]


def _pack(arrays: Sequence[np.ndarray]) -> np.ndarray:
    arrs = [np.ascontiguousarray(a, np.int32).reshape(-1) for a in arrays]
    head = np.asarray([len(arrs)] + [a.size for a in arrs], np.int32)
    return np.concatenate([head] + arrs) if arrs else head


def _unpack(buf: np.ndarray) -> List[np.ndarray]:
    n = int(buf[0])
    lens = buf[1:1 + n].astype(np.int64)
    out, p = [], 1 + n
    for l in lens:
        out.append(buf[p:p + l].copy())
        p += int(l)
    return out


def gather_int_arrays(local: Sequence[np.ndarray], device: Optional[torch.device] = None, group=None, force: bool = False) -> List[List[np.ndarray]]:
    """All-gather a list of variable-length int32 arrays from every rank -> per-rank lists (same on all ranks).

    One all_reduce(MAX) of the packed length + one all_gather of the padded buffers."""
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return [[np.asarray(a, np.int32).reshape(-1) for a in local]]
    world = dist.get_world_size(group)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    packed = _pack(local)
    n = torch.tensor([packed.size], dtype=torch.int64, device=device)
    dist.all_reduce(n, op=dist.ReduceOp.MAX, group=group)
    cap = int(n.item())
    mine = torch.zeros(cap, dtype=torch.int32, device=device)
    mine[: packed.size] = torch.from_numpy(packed).to(device)
    bufs = [torch.empty(cap, dtype=torch.int32, device=device) for _ in range(world)]
    dist.all_gather(bufs, mine, group=group)
    return [_unpack(b.cpu().numpy()) for b in bufs]


def notes_to_array(notes: Sequence[dict]) -> np.ndarray:
    """note dicts -> int32 view of [n, 6]: onset(f64 as 2x i32), offset(f64 as 2x i32), pitch, velocity (bit exact)."""
    if not notes:
        return np.zeros(0, np.int32)
    a = np.zeros((len(notes), 3), np.float64)
    a[:, 0] = [n["onset"] for n in notes]
    a[:, 1] = [n["offset"] for n in notes]
    pv = np.asarray([[n["pitch"], n["velocity"]] for n in notes], np.int32)
    return np.concatenate([a[:, :2].copy().view(np.int32).reshape(len(notes), 4), pv], axis=1).reshape(-1)


def array_to_notes(arr: np.ndarray) -> List[dict]:
    if arr.size == 0:
        return []
    a = np.ascontiguousarray(arr, np.int32).reshape(-1, 6)
    t = np.ascontiguousarray(a[:, :4]).view(np.float64).reshape(-1, 2)
    return [{"pitch": int(p), "onset": float(on), "offset": float(off), "velocity": int(v)} for (on, off), (p, v) in zip(t, a[:, 4:])]
