"""N>1 path on CPU: pass sharding arithmetic and the cond-entry all-gather under gloo (world_size 2)."""
import os
import socket

import pytest
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


def _sharded_worker(rank, world, port, q, n_frames, appear):
    """One rank of a 2-process gloo job: the real ShardedVideoProcessor + real predictor state machine over the CPU
    stand-in for the HIP stages, every collective of the round (fixed-size tensor all-gathers, batch_isend_irecv ring
    shift) through torch.distributed."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _fake_hip import fake_predictor
    from det_sam2_amd.synth import SyntheticDetector, synthetic_frame
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        vp = P.ShardedVideoProcessor(model_cfg="sam2.1_hiera_t", detector=SyntheticDetector(3, size=32, appear=appear),
                                     skip_classes=set(), predictor=fake_predictor(), frame_buffer_size=3, detect_interval=3,
                                     max_frame_num_to_track=6, max_inference_state_frames=6)
        assert (vp.rank, vp.world) == (rank, world) and isinstance(vp.comm, P.TorchDistComm)
        segs = vp.run(frames=[synthetic_frame(t, size=32) for t in range(n_frames)])
        q.put((rank, {t: s.to_dict() for t, s in segs.items()}, vp.owned_passes, vp.predictor.stats["encoder_runs"],
               [(r, op) for r, op, _ in vp.comm_log]))
    finally:
        dist.destroy_process_group()


def test_sharded_stream_through_gloo_world2_equals_sequential():
    """ADVICE/VERDICT round 1: the sharded driver had only run with an in-process mailbox.  Here two processes run one
    17-frame stream (6 passes, the last one partial; a new class appears in pass 2) through torch.distributed."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import numpy as np
    from _fake_hip import fake_predictor
    from det_sam2_amd.det_sam2_RT import VideoProcessor
    from det_sam2_amd.synth import SyntheticDetector, synthetic_frame
    n, appear = 17, {2: 6}
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q, n, appear)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=180) for _ in ps), key=lambda x: x[0])
    for p in ps:
        p.join(timeout=60)
    seq = VideoProcessor(model_cfg="sam2.1_hiera_t", detector=SyntheticDetector(3, size=32, appear=appear), skip_classes=set(),
                         predictor=fake_predictor(), frame_buffer_size=3, detect_interval=3, max_frame_num_to_track=6,
                         max_inference_state_frames=6)
    ref = seq.run(frames=[synthetic_frame(t, size=32) for t in range(n)])
    assert res[0][2] == [0, 2, 4] and res[1][2] == [1, 3, 5]
    merged = P.merge_segments([res[0][1], res[1][1]], 3, 6, 6, 2, n)
    assert sorted(merged) == sorted(ref) == list(range(n))
    for t in ref:
        assert sorted(merged[t]) == sorted(ref[t])
        for o in ref[t]:
            assert np.array_equal(merged[t][o], ref[t][o]), (t, o)
    assert n <= res[0][3] + res[1][3] <= n + 2                      # frames encoded once (hand-off), not once per pass
    assert {"ring_shift", "all_gather_dets", "all_gather_entries"} <= {op for _, op in res[0][4]}


def test_sharded_stream_through_gloo_world3_partial_final_round():
    """Three processes, 13 frames at buffer 3 = 5 passes: round 0 holds passes 0-2, the FINAL round only passes 3-4 (one
    of them a 1-frame buffer) - rank 2 owns no pass there but still takes part in every collective of the round."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import numpy as np
    from _fake_hip import fake_predictor
    from det_sam2_amd.det_sam2_RT import VideoProcessor
    from det_sam2_amd.synth import SyntheticDetector, synthetic_frame
    n, appear, world = 13, {1: 3}, 3
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_sharded_worker, args=(r, world, port, q, n, appear)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=240) for _ in ps), key=lambda x: x[0])
    for p in ps:
        p.join(timeout=60)
    seq = VideoProcessor(model_cfg="sam2.1_hiera_t", detector=SyntheticDetector(3, size=32, appear=appear), skip_classes=set(),
                         predictor=fake_predictor(), frame_buffer_size=3, detect_interval=3, max_frame_num_to_track=6,
                         max_inference_state_frames=6)
    ref = seq.run(frames=[synthetic_frame(t, size=32) for t in range(n)])
    assert [r[2] for r in res] == [[0, 3], [1, 4], [2]]
    merged = P.merge_segments([r[1] for r in res], 3, 6, 5, world, n)
    assert sorted(merged) == sorted(ref) == list(range(n))
    for t in ref:
        assert sorted(merged[t]) == sorted(ref[t])
        for o in ref[t]:
            assert np.array_equal(merged[t][o], ref[t][o]), (t, o)
    # every rank saw the same sequence of collectives (nobody skipped one in the partial round)
    assert res[0][4] == res[1][4] == res[2][4]


def test_sharded_class_is_one_object():
    """parallel.__getattr__ used to build a new class on every access (isinstance was always False)."""
    assert P.ShardedVideoProcessor is P.ShardedVideoProcessor
    from det_sam2_amd.det_sam2_RT import VideoProcessor
    assert issubclass(P.ShardedVideoProcessor, VideoProcessor)


def test_detection_and_meta_wire_formats_round_trip():
    import numpy as np
    dets = {"frame_30": [{"coordinates": np.array([1.5, 2.25, 300.125, 400.0], np.float32), "class": np.array([7.0], np.float32),
                          "confidence": np.array([0.8125], np.float32)},
                         {"coordinates": np.array([0.1, 0.2, 0.3, 0.4], np.float32), "class": np.array([11.0], np.float32),
                          "confidence": np.array([0.3], np.float32)}],
            "frame_45": []}
    assert P.dets_count(dets) == 2
    t = P.dets_to_tensor(dets, 5)
    assert tuple(t.shape) == (1 + 5, 7) and t.dtype == torch.float64
    back = P.dets_from_tensor(t)
    assert list(back) == ["frame_30"] and len(back["frame_30"]) == 2
    for a, b in zip(back["frame_30"], dets["frame_30"]):
        for k in ("coordinates", "class", "confidence"):
            assert a[k].dtype == np.float32 and np.array_equal(a[k], b[k])          # bit-exact fp32 through fp64
    assert P.dets_from_tensor(P.dets_to_tensor({}, 1)) == {}
    meta = {90: 16, 60: 17}
    assert P.meta_from_tensor(P.meta_to_tensor(meta, 2)) == meta
    with pytest.raises(RuntimeError):
        P.dets_to_tensor(dets, 1)


def test_merge_segments_rule():
    segs = [{t: {0: ("r0", t)} for t in range(90)}, {t: {0: ("r1", t)} for t in range(90)}]
    m = P.merge_segments(segs, 30, 60, 3, 2)
    assert P.merge_segments(segs, 30, 60, 3, 2, 90) == m
    assert m[0][0][0] == "r1" and m[29][0][0] == "r1"       # buffer 0: passes 0 (rank 0) and 1 (rank 1) -> pass 1
    assert m[30][0][0] == "r0" and m[59][0][0] == "r0"      # buffer 1: passes 1 and 2 -> pass 2 (rank 0)
    assert m[60][0][0] == "r0" and m[89][0][0] == "r0"      # buffer 2: only pass 2 so far
