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
