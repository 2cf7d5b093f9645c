"""N>1 path on CPU: world_size-2 gloo run of the clip sharding + final gather (etude_amd/parallel.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from etude_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_job_result(item: int):
    rng = np.random.default_rng(item)
    toks = rng.integers(0, 154, int(rng.integers(1, 50))).astype(np.int32)
    notes = [{"pitch": int(rng.integers(21, 109)), "onset": float(rng.random()), "offset": float(1 + rng.random()), "velocity": int(rng.integers(1, 128))}
             for _ in range(int(rng.integers(0, 6)))]
    return toks, notes


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = parallel.shard(list(range(n_items)), rank, world)
    toks, notes = [], []
    for it in mine:
        t, n = _fake_job_result(it)
        toks.append(t)
        notes.append(parallel.notes_to_array(n))
    g_t = parallel.gather_int_arrays(toks)
    g_n = parallel.gather_int_arrays(notes)
    all_t = parallel.unshard(g_t, n_items)
    all_n = parallel.unshard([[parallel.array_to_notes(a) for a in lst] for lst in g_n], n_items)
    ok = all(np.array_equal(all_t[i], _fake_job_result(i)[0]) and all_n[i] == _fake_job_result(i)[1] for i in range(n_items))
    dist.barrier()
    q.put((rank, ok, len(mine)))
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n_items = 7                      # ragged: rank 0 gets 4 items, rank 1 gets 3
    ps = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert res == [(0, True, 4), (1, True, 3)]


def test_shard_unshard_roundtrip_and_single_process_gather():
    items = list(range(64))
    per = [parallel.shard(items, r, 8) for r in range(8)]
    assert all(len(p) == 8 for p in per) and per[3][:2] == [3, 11]
    assert parallel.unshard(per, 64) == items
    out = parallel.gather_int_arrays([np.arange(5), np.zeros(0, np.int32)])
    assert len(out) == 1 and np.array_equal(out[0][0], np.arange(5)) and out[0][1].size == 0
    notes = [{"pitch": 60, "onset": 0.1234567890123, "offset": 1.5, "velocity": 99}]
    assert parallel.array_to_notes(parallel.notes_to_array(notes)) == notes


def test_global_clip_order_digest_is_the_same_at_every_world_size():
    """bench.py's `tokens_sha256_all`: every job's ids with the clips in global order (rank r holds clips r, r + R, ...).  The digest of a 64-clip x 3-tuple batch
    is the same whether 1, 2, 4 or 8 ranks held it -- what lets an N-rank run be compared with the 1-rank run."""
    rng = np.random.default_rng(3)
    n_clips, na = 64, 3
    jobs = [[rng.integers(0, 154, int(rng.integers(5, 40))).astype(np.int32) for _ in range(na)] for _ in range(n_clips)]      # clip-major, tuple-minor
    want = None
    for world in (1, 2, 4, 8):
        per_rank = []
        for r in range(world):
            mine = parallel.shard(list(range(n_clips)), r, world)
            per_rank.append([jobs[c][t] for c in mine for t in range(na)])
        d = parallel.digest_in_global_clip_order(per_rank, na)
        want = want or d
        assert d == want, world
    # and it is the plain digest of the batch in clip order
    import hashlib
    assert want == hashlib.sha256(np.concatenate([jobs[c][t] for c in range(n_clips) for t in range(na)]).astype(np.int32).tobytes()).hexdigest()[:16]
