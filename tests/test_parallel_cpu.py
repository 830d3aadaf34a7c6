"""The N>1 path without a GPU: two gloo processes exercise the flat-gradient all-reduce (`GradSync`), the gradient-ready bucket
plan and the rank-0 broadcast that `csmae_hip.parallel.DataParallel` performs with RCCL on the MI355X node."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "cross-scale-mae_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import models_mae
        from csmae_hip.parallel import FlatBuffers, GradSync, bucket_ranges
        micro = dict(dim_model=128, encoder_num_layers=6, encoder_num_heads=2, decoder_embed_dim=64, decoder_num_layers=2, decoder_num_heads=2)
        torch.manual_seed(0)
        m = models_mae.MAE_ViT_MsLdCeCd(**micro, input_size=64, predictor_hidden_size=128)
        slots, off = {}, 0
        names = []
        for n, p in m.named_parameters():
            slots[n] = (off, p.numel(), tuple(p.shape))
            names.append(n)
            off += (p.numel() + 7) // 8 * 8
        total = off
        ranges = bucket_ranges(slots, names, n_encoder=6, enc_per_bucket=4)
        # the plan tiles [0, total) exactly once, in gradient-ready order: tail first, stem last
        assert ranges[0][0] == "tail" and ranges[-1][0] == "stem" and ranges[0][2] == total and ranges[-1][1] == 0
        covered = sorted((lo, hi) for _, lo, hi in ranges)
        assert covered[0][0] == 0 and covered[-1][1] == total and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
        assert [r[0] for r in ranges[1:-1]] == [("enc", 2), ("enc", 0)]
        assert ranges[0][1] == slots["decoder.0.norm1.weight"][0] and ranges[-1][2] == slots["encoder.0.norm1.weight"][0]
        tap = bucket_ranges(slots, names, n_encoder=6, enc_per_bucket=4, taper=True)  # small messages last: groups 4, 1, 1 (12 layers: 4, 4, 2, 1, 1)
        assert [r[0] for r in tap[1:-1]] == [("enc", 2), ("enc", 1), ("enc", 0)]
        cov = sorted((lo, hi) for _, lo, hi in tap)
        assert cov[0][0] == 0 and cov[-1][1] == total and all(a[1] == b[0] for a, b in zip(cov, cov[1:]))
        # mean all-reduce of every range == mean of the per-rank gradients
        g = torch.Generator().manual_seed(100 + rank)
        flat = torch.randn(total, generator=g)
        mine = flat.clone()
        sync = GradSync(flat)
        for _, lo, hi in ranges:
            sync.reduce_range(lo, hi)
        sync.finish()
        others = [torch.randn(total, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
        want = torch.stack(others).mean(0)
        assert torch.allclose(flat, want, atol=1e-6), float((flat - want).abs().max())
        assert torch.equal(others[rank], mine)
        # per-bucket timing (bench.py's data_parallel diagnosis) is a GPU-side measurement: on CPU tensors it is inert and the exchange unchanged
        flat3 = mine.clone()
        s3 = GradSync(flat3)
        s3.timing = True
        for _, lo, hi in ranges:
            s3.reduce_range(lo, hi)
        s3.finish()
        assert s3.bucket_times() == [] and torch.allclose(flat3, want, atol=1e-6) and len(s3.issued) == len(ranges)
        # bf16 payload variant stays within bf16 rounding of the exact mean
        flat2 = mine.clone()
        s2 = GradSync(flat2, comm_dtype=torch.bfloat16)
        s2.reduce_range(0, total)
        assert torch.allclose(flat2, want, atol=3e-2, rtol=2e-2)
        # rank-0 broadcast of parameters + BatchNorm buffers (DDP constructor / broadcast_buffers semantics)
        p = torch.full((16,), float(rank))
        nbt = torch.tensor(rank, dtype=torch.long)
        sync.broadcast([p, nbt])
        assert float(p.sum()) == 0.0 and int(nbt) == 0
        # the per-forward BatchNorm-buffer broadcast is ONE message: running_mean, running_var (fp32) and num_batches_tracked (int64)
        # live in one byte buffer; in-place updates of the module's buffers land in it, a broadcast of it reaches the module
        bn = m.predictor[1]
        fb = FlatBuffers(m)
        assert fb.still_homed() and [n for n, _ in fb.items] == ["predictor.1.running_mean", "predictor.1.running_var", "predictor.1.num_batches_tracked"]
        bn.running_mean.fill_(1.0 + rank)
        bn.running_var.mul_(2.0 + rank)
        bn.num_batches_tracked.add_(5 + rank)
        sync.broadcast([fb.raw])
        assert float(bn.running_mean.mean()) == 1.0 and float(bn.running_var.mean()) == 2.0 and int(bn.num_batches_tracked) == 5
        m.load_state_dict(m.state_dict())      # copies in place: still homed
        assert fb.still_homed()
        # the update-gate slot behind the last parameter rides in the tail range
        assert bucket_ranges(slots, names, n_encoder=6, tail_extra=8)[0][2] == total + 8
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_sync_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results


def test_single_process_is_a_noop():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "cross-scale-mae_amd"))
    from csmae_hip.parallel import GradSync
    g = torch.arange(8.0)
    s = GradSync(g)
    s.reduce_range(0, 8)
    s.finish()
    assert torch.equal(g, torch.arange(8.0))
