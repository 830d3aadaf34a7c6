"""End-to-end on the MI355X: the reference's config-1 epoch (BASELINE.json configs[0]) through our engine_pretrain loop, the CLI driver
with the synthetic loader, checkpoint round trip, and the data-parallel wrapper at world size 1 over RCCL."""
import json
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module", autouse=True)
def need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def checksum(t):
    flat = t.double().reshape(-1).cpu()
    w = torch.arange(flat.numel(), dtype=torch.float64) % 97 + 1
    return [float(flat.sum()), float(flat.abs().sum()), float((flat * w).sum())]


def test_config1_epoch_matches_reference_engine():
    """MAE_ViT_Baseline ViT-B/16, 2x3x64x64, mask 0.75, one optimizer step: stats dict and updated weights vs the reference's
    engine_pretrain.train_one_epoch run on its CPU path (tests/golden/vitb_anchor.*)."""
    import models_mae
    from csmae_hip.optim import FusedAdamW, add_weight_decay
    from engine_pretrain import train_one_epoch
    from util.misc import NativeScalerWithGradNormCount
    a = json.load(open(os.path.join(G, "vitb_anchor.json")))
    d = np.load(os.path.join(G, "vitb_anchor.npz"))
    torch.manual_seed(0)
    m = models_mae.mae_vit_base(input_size=64, patch_size="16", loss="mse", device="cuda").cuda()
    m.compute_dtype = torch.float32  # parity mode (the loop itself runs under autocast like the reference's)
    opt = FusedAdamW(add_weight_decay(m, 0.05), lr=1e-3, betas=(0.9, 0.95))
    args = types.SimpleNamespace(accum_iter=1, lr=1e-3, min_lr=0.0, warmup_epochs=0, epochs=1, mask_ratio=0.75)
    m._test_draws = dict(noise=[torch.from_numpy(d["cfg1_noise"])], box=None)
    stats = train_one_epoch(m, [(torch.from_numpy(d["cfg1_imgs"]), None)], opt, torch.device("cuda"), 0, NativeScalerWithGradNormCount(), args=args)
    ref = a["cfg1_stats"]
    assert {"lr", "loss", "time_epoch", "time_step"} <= set(stats)
    assert stats["lr"] == ref["lr"]
    assert abs(stats["loss"] - ref["loss"]) < 1e-4 * abs(ref["loss"]), (stats["loss"], ref["loss"])
    sd = m.state_dict()
    for k, c in a["cfg1_after_checksums"].items():
        got = checksum(sd[k])
        for x, y in zip(got, c[:3]):
            assert abs(x - y) <= 2e-4 * abs(y) + 2e-3, (k, got, c[:3])


def test_cli_driver_synthetic_bf16_and_checkpoint(tmp_path):
    import main_pretrain
    argv = ["--model", "mae_vit_base_MsLdCeCd", "--dataset_type", "synthetic", "--input_size", "64", "--batch_size", "8", "--epochs", "2",
            "--warmup_epochs", "1", "--synthetic_len", "3", "--output_dir_base", str(tmp_path), "--output_dir", "run", "--blr", "1e-3"]
    args = main_pretrain.get_args_parser().parse_args(argv)
    main_pretrain.main(args)
    out = tmp_path / "run"
    logs = [json.loads(l) for l in open(out / "log.jsonl")]
    assert [l["epoch"] for l in logs] == [0, 1] and all(np.isfinite(l["train_loss"]) for l in logs)
    assert logs[1]["train_loss"] < logs[0]["train_loss"] + 0.5
    ck = torch.load(out / "checkpoint-1.pth", weights_only=False)
    assert set(ck) == {"model", "optimizer", "epoch", "scaler", "args"}
    assert "predictor.1.running_mean" in ck["model"] and len(ck["optimizer"]["state"]) > 200
    # resume: weights and AdamW moments come back
    args2 = main_pretrain.get_args_parser().parse_args(argv + ["--resume", str(out / "checkpoint-1.pth"), "--epochs", "3", "--honor_start_epoch"])
    main_pretrain.main(args2)
    logs = [json.loads(l) for l in open(out / "log.jsonl")]
    assert [l["epoch"] for l in logs] == [0, 1, 2]


def test_cli_driver_fmow_csv_through_gpu_input_step(tmp_path):
    """`--dataset_type fmow_rgb` reads the reference's CSV layout (label, path), decodes on the workers and runs the
    training transform on the GPU."""
    from PIL import Image
    import main_pretrain
    rng = np.random.default_rng(0)
    rows = ["label,image_path"]
    for k in range(12):
        h, w = 70 + 5 * k, 90 + 3 * k
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(tmp_path / f"im{k}.png")
        rows.append(f"{k % 3},im{k}.png")
    (tmp_path / "train.csv").write_text("\n".join(rows) + "\n")
    argv = ["--model", "mae_vit_base_MsLdCeCd", "--dataset_type", "fmow_rgb", "--train_path", str(tmp_path / "train.csv"), "--input_size", "64",
            "--batch_size", "4", "--epochs", "1", "--warmup_epochs", "0", "--num_workers", "0", "--output_dir_base", str(tmp_path), "--output_dir", "run"]
    main_pretrain.main(main_pretrain.get_args_parser().parse_args(argv))
    logs = [json.loads(l) for l in open(tmp_path / "run" / "log.jsonl")]
    assert len(logs) == 1 and np.isfinite(logs[0]["train_loss"])


def test_data_parallel_world1_rccl_matches_plain():
    import torch.distributed as dist
    import models_mae
    from csmae_hip.parallel import DataParallel
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        micro = dict(dim_model=128, encoder_num_layers=6, encoder_num_heads=2, decoder_embed_dim=64, decoder_num_layers=2, decoder_num_heads=2)
        x = torch.randn(4, 3, 64, 64, generator=torch.Generator().manual_seed(3)).cuda()
        g = torch.Generator().manual_seed(4)
        noise = [torch.rand(4, 16, generator=g), torch.rand(4, 16, generator=g)]
        grads = []
        for wrap in (False, True):
            torch.manual_seed(0)
            m = models_mae.MAE_ViT_MsLdCeCd(**micro, input_size=64, predictor_hidden_size=128).cuda()
            w = DataParallel(m) if wrap else m
            m._test_draws = dict(noise=noise, box=(7, 2, 45, 48))
            loss, _, _ = w(x)
            loss.backward()
            grads.append({n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
        assert grads[0].keys() == grads[1].keys()
        for n in grads[0]:
            torch.testing.assert_close(grads[0][n], grads[1][n], rtol=1e-4, atol=1e-6)
    finally:
        if created:
            dist.destroy_process_group()


def _dp_two_rank_worker(rank, world, port, out_dir):
    """Two processes on the ONE GPU of the test box, gloo as the transport (RCCL refuses two ranks on one device): exercises the
    world_size > 1 control flow of DataParallel + Engine.backward on real kernels — rank-0 broadcast, gradient-ready ranges,
    side-stream joins, BatchNorm buffer broadcast, no_sync."""
    import torch.distributed as dist
    import models_mae
    from csmae_hip.optim import FusedAdamW, add_weight_decay
    from csmae_hip.parallel import DataParallel
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        micro = dict(dim_model=128, encoder_num_layers=6, encoder_num_heads=2, decoder_embed_dim=64, decoder_num_layers=2, decoder_num_heads=2)
        torch.manual_seed(100 + rank)  # different initial weights per rank: the wrapper must broadcast rank 0's
        m = models_mae.MAE_ViT_MsLdCeCd(**micro, input_size=64, predictor_hidden_size=128).cuda().train()
        w = DataParallel(m)
        opt = FusedAdamW(add_weight_decay(m, 0.05), lr=1e-3, betas=(0.9, 0.95))
        x = torch.randn(4, 3, 64, 64, generator=torch.Generator().manual_seed(10 + rank)).cuda()
        g = torch.Generator().manual_seed(20 + rank)
        grads = None
        for step in range(3):
            m._test_draws = dict(noise=[torch.rand(4, 16, generator=g), torch.rand(4, 16, generator=g)], box=(7, 2, 45, 48))
            opt.zero_grad(set_to_none=True)
            if step == 1:  # accumulation micro-step: no exchange, gradients stay local
                with w.no_sync():
                    loss, _, _ = w(x)
                    loss.backward()
                m._test_draws = dict(noise=[torch.rand(4, 16, generator=g), torch.rand(4, 16, generator=g)], box=(7, 2, 45, 48))
            loss, _, _ = w(x)
            loss.backward()
            if step == 0:
                grads = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters() if p.grad is not None}
            opt.step()
        torch.cuda.synchronize()
        torch.save(dict(params={n: p.detach().cpu() for n, p in m.named_parameters()}, grads=grads, loss=float(loss.detach()),
                        bn=m.predictor[1].running_mean.detach().cpu()), os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_data_parallel_two_ranks_on_one_gpu_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = 29600 + os.getpid() % 200
    mp.spawn(_dp_two_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert np.isfinite(r0["loss"]) and np.isfinite(r1["loss"])
    # gradients after the exchange are the mean over ranks -> identical on both; so are the parameters after three AdamW steps
    for n in r0["grads"]:
        torch.testing.assert_close(r0["grads"][n], r1["grads"][n], rtol=0, atol=0)
    for n in r0["params"]:
        torch.testing.assert_close(r0["params"][n], r1["params"][n], rtol=0, atol=0)
    # ... and they are not simply rank 0's local gradients: the two ranks saw different data
    m = max(float((r0["grads"][n]).abs().max()) for n in r0["grads"])
    assert m > 0


def test_bench_two_ranks_plumbing_on_one_gpu():
    """bench.py's N > 1 path exactly as the driver launches it (torch.distributed.run, one rank per process, barrier + max-over-ranks
    timing, rank 0 prints ONE JSON line) — with gloo and both ranks on the one GPU of the test box, because RCCL refuses two ranks on a
    device.  The 8-GPU RCCL run itself is the driver's."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CSMAE_BENCH_ONE_GPU="1")
    port = 29700 + os.getpid() % 200
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--batch", "8", "--no-kernel-timing"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["config"]["global_batch"] == 16
    assert d["config"]["parallelism"] == "dp2" and d["value"] > 0 and np.isfinite(d["loss"]) and "cpu_baseline" not in d
    assert abs(d["value"] - 16 * 2 / (d["ms_per_step"] * 2 / 1e3)) < 1e-2 * d["value"]
