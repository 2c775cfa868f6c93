"""N>1 path on CPU: pass sharding arithmetic and the cond-entry all-gather under gloo (world_size 2)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from det_sam2_amd import parallel as P


def test_pass_sharding_covers_all_passes_once():
    for world in (1, 2, 4, 8):
        seen = sorted(k for r in range(world) for k in P.passes_of_rank(33, world, r))
        assert seen == list(range(33))
    assert P.pass_window(0, 30, 60) == (0, 29)
    assert P.pass_window(1, 30, 60) == (0, 59)
    assert P.pass_window(2, 30, 60) == (30, 89)


def _entry(rank, B=3):
    g = torch.Generator().manual_seed(100 + rank)
    return {"maskmem_features": torch.randn(B, 4096, 64, generator=g).to(torch.bfloat16),
            "pred_masks": torch.randn(B, 1, 256, 256, generator=g),
            "obj_ptr": torch.randn(B, 256, generator=g),
            "object_score_logits": torch.randn(B, 1, generator=g), "maskmem_pos_enc": None}


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    got = P.allgather_cond_entries(_entry(rank))
    ok = len(got) == world
    for r, e in enumerate(got):
        ref = _entry(r)
        for k in P.ENTRY_FIELDS:
            ok &= bool(torch.equal(e[k], ref[k])) and e[k].dtype == ref[k].dtype
    q.put((rank, ok))
    dist.destroy_process_group()


def test_allgather_cond_entries_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_pack_unpack_roundtrip_single_process():
    e = _entry(0)
    back = P.unpack_entry(P.pack_entry(e), e)
    assert all(torch.equal(back[k], e[k]) for k in P.ENTRY_FIELDS)
    assert P.allgather_cond_entries(e)[0] is e


def _bworker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    for src in range(world):        # every rank is the owner of some pass
        payload = None
        if rank == src:
            payload = {"obj_ids": [3, 7, 9], "entries": {10 * src: _entry(src), 10 * src + 5: _entry(src + 10, B=2)}}
        got = P.broadcast_cond_entries(payload, src, "cpu")
        ok &= got["obj_ids"] == [3, 7, 9] and sorted(got["entries"]) == [10 * src, 10 * src + 5]
        for t, ref in ((10 * src, _entry(src)), (10 * src + 5, _entry(src + 10, B=2))):
            for k in P.ENTRY_FIELDS:
                ok &= bool(torch.equal(got["entries"][t][k], ref[k])) and got["entries"][t][k].dtype == ref[k].dtype
    q.put((rank, ok))
    dist.destroy_process_group()


def test_broadcast_cond_entries_gloo_world2():
    """The exchange of the pass-sharded stream (owner -> everyone, ragged object counts) under gloo."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_bworker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_merge_segments_rule():
    segs = [{t: {0: ("r0", t)} for t in range(90)}, {t: {0: ("r1", t)} for t in range(90)}]
    m = P.merge_segments(segs, 30, 60, 3, 2)
    assert m[0][0][0] == "r1" and m[29][0][0] == "r1"       # buffer 0: passes 0 (rank 0) and 1 (rank 1) -> pass 1
    assert m[30][0][0] == "r0" and m[59][0][0] == "r0"      # buffer 1: passes 1 and 2 -> pass 2 (rank 0)
    assert m[60][0][0] == "r0" and m[89][0][0] == "r0"      # buffer 2: only pass 2 so far
