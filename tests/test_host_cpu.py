"""CPU-only checks of the host-side mirror of the reference interface: CLI surface, factories / state_dict manifest / seeded
init (vs the reference's own values in tests/golden/manifest.json), LR schedule, the epoch loop (with a stub model), meters,
the scaler contract, checkpoint layout."""
import json
import math
import os
import types

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")


def checksum(t):
    flat = t.double().reshape(-1)
    w = torch.arange(flat.numel(), dtype=torch.float64) % 97 + 1
    return [float(flat.sum()), float(flat.abs().sum()), float((flat * w).sum()), [float(x) for x in flat[:4]]]


def test_cli_matches_reference_flags():
    import main_pretrain
    a = main_pretrain.get_args_parser().parse_args([])
    expect = dict(batch_size=512, epochs=200, accum_iter=1, model="mae_vit_base", input_size=224, patch_size=16, print_level=1, mask_ratio=0.75,
                  attn_name="scaled_dot_product", use_xformers=False, ffn_name="MLP", spatial_mask=False, loss="mse", norm_pix_loss=False,
                  weight_decay=0.05, lr=None, blr=5e-5, min_lr=0.0, warmup_epochs=40, train_path="./train.csv", dataset_type="fmow_rgb",
                  masked_bands=None, dropped_bands=None, output_dir=None, output_dir_base="./out", val_img_path="./images/", device="cuda",
                  seed=0, resume=None, start_epoch=0, wandb_entity="utk-iccv23", wandb_project=None, wandb_id=None, num_workers=os.cpu_count(),
                  pin_mem=True, world_size=1, dist_on_itp=False, dist_url="env://")
    for k, v in expect.items():
        assert getattr(a, k) == v, k
    assert int(a.local_rank) == int(os.getenv("LOCAL_RANK", 0))
    b = main_pretrain.get_args_parser().parse_args(["--patch_size", "14", "--no_pin_mem", "--resume", "", "--masked_bands", "1", "2", "--loss", "ssim"])
    assert b.patch_size == "14" and b.pin_mem is False and b.resume is None and b.masked_bands == [1, 2]
    with pytest.raises(SystemExit):
        main_pretrain.get_args_parser().parse_args(["-h"])  # add_help=False in the reference


def test_factories_manifest_and_seeded_init_match_reference():
    import models_mae
    man = json.load(open(os.path.join(G, "manifest.json")))
    for f, m in man.items():
        size = 224 if "@224" in f else 64
        torch.manual_seed(0)
        mod = getattr(models_mae, f.split("@")[0])(input_size=size, patch_size="16", loss="mse", device="cpu")
        sd = mod.state_dict()
        assert list(sd.keys()) == m["keys"], f
        assert [list(v.shape) for v in sd.values()] == m["shapes"], f
        assert sum(p.numel() for p in mod.parameters() if p.requires_grad) == m["trainable"], f
        if "param_order" in m:
            assert [n for n, _ in mod.named_parameters()] == m["param_order"]
            assert [bool(p.requires_grad) for _, p in mod.named_parameters()] == m["requires_grad"]
        for k, c in m.get("checksums", {}).items():
            assert checksum(sd[k]) == c, (f, k)  # bit-identical seeded initialisation


def test_constructor_surface():
    import models_mae
    a = vars(__import__("main_pretrain").get_args_parser().parse_args(["--input_size", "64"]))
    a.pop("input_channels")
    m = models_mae.__dict__["mae_vit_base_MsLdCeCd"](**a)  # the whole namespace is splatted in (main_pretrain.py:398)
    assert (m.input_size, m.input_channels, m.patch_size, m.dim_model, m.decoder_embed_dim, m.num_patches) == (64, 3, 16, 768, 512, 16)
    assert m.patch_embed.patch_size == (16, 16) and m.patch_embed.num_patches == 16 and m.loss == "mse" and m.norm_pix_loss is False
    assert m.no_weight_decay() == {} and m.use_xformers is False and m.mask_ratio == 0.75
    x = torch.randn(2, 3, 64, 64)
    assert torch.equal(m.unpatchify(m.patchify(x, 16, 3), 16, 3), x)
    assert models_mae.mae_vit_base(input_size=64, loss="ssim").loss == "ssim"   # §8 f-4: reconstruction loss only ...
    with pytest.raises(ValueError, match="only serves the reconstruction loss"):  # ... the reference dies in forward (unpatchify(x, None, None))
        models_mae.mae_vit_base_MsLdCeCd(input_size=64, loss="mse_ssim")          # loss_cd defaults to `loss` (MAE_ViT_MsLdCeCd.py:18)
    assert models_mae.mae_vit_base_MsLdCeCd(input_size=64, loss="mse_ssim", loss_cd="mse").loss_cd == "mse"
    with pytest.raises(AttributeError, match="forward_loss_huber"):
        models_mae.mae_vit_base(input_size=64, loss="huber")
    with pytest.raises(NotImplementedError, match="xFormers"):
        models_mae.mae_vit_base(input_size=64, use_xformers=True)
    with pytest.raises(AssertionError):
        models_mae.mae_vit_base(input_size=64, attn_name="linformer")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(x)  # the product path never computes on the CPU


def test_lr_schedule_matches_reference_table():
    import util.lr_sched as lr_sched
    a = json.load(open(os.path.join(G, "vitb_anchor.json")))
    for lr, min_lr, wu, ep, e, want, g0, g1 in a["lr_table"]:
        opt = types.SimpleNamespace(param_groups=[dict(lr=0.0), dict(lr=0.0, lr_scale=0.5)])
        got = lr_sched.adjust_learning_rate(opt, e, types.SimpleNamespace(lr=lr, min_lr=min_lr, warmup_epochs=wu, epochs=ep))
        assert got == pytest.approx(want, rel=1e-12, abs=1e-18)
        assert opt.param_groups[0]["lr"] == pytest.approx(g0, rel=1e-12, abs=1e-18) and opt.param_groups[1]["lr"] == pytest.approx(g1, rel=1e-12, abs=1e-18)


class _Stub(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor([1.0, -2.0]))
        self.calls = []

    def forward(self, samples, mask_ratio=0.75):
        self.calls.append(mask_ratio)
        return (self.w * samples.mean()).pow(2).sum(), None, None


def test_train_one_epoch_contract_on_cpu():
    from engine_pretrain import train_one_epoch
    from util.misc import NativeScalerWithGradNormCount
    m = _Stub()
    opt = torch.optim.SGD(m.parameters(), lr=0.0)
    args = types.SimpleNamespace(accum_iter=2, lr=1e-2, min_lr=0.0, warmup_epochs=1, epochs=4, mask_ratio=0.6)
    data = [(torch.full((2, 1), float(i + 1)), None) for i in range(6)]
    stats = train_one_epoch(m, data, opt, torch.device("cpu"), 1, NativeScalerWithGradNormCount(), log_writer=None, args=args)
    assert set(stats) == {"lr", "loss", "time_epoch", "time_step"}
    assert m.calls == [0.6] * 6
    # lr of the last update step: iteration 4 of 6 in epoch 1 -> fractional epoch 1 + 4/6 (per-iteration schedule, accum aware)
    e = 1 + 4 / 6
    want = 0.5 * 1e-2 * (1 + math.cos(math.pi * (e - 1) / 3))
    assert opt.param_groups[0]["lr"] == pytest.approx(want)
    w0 = torch.tensor([1.0, -2.0])
    assert stats["loss"] > 0 and not torch.equal(m.w.detach(), w0)
    bad = [(torch.full((2, 1), float("nan")), None)]
    with pytest.raises(ValueError, match="stopping training"):
        train_one_epoch(_Stub(), bad, opt, torch.device("cpu"), 0, NativeScalerWithGradNormCount(), args=args)


def test_meters_scaler_and_checkpoint_layout(tmp_path):
    import util.misc as misc
    sv = misc.SmoothedValue(window_size=3)
    for v in (1.0, 2.0, 3.0, 10.0):
        sv.update(v)
    assert sv.median == 3.0 and sv.global_avg == 4.0 and sv.max == 10.0 and sv.value == 10.0 and sv.avg == pytest.approx(5.0)
    sc = misc.NativeScalerWithGradNormCount()
    assert sc.state_dict_key == "amp_scaler" and sc.state_dict() == {}
    m = _Stub()
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    loss = m(torch.ones(2, 1))[0]
    assert sc(loss, opt, parameters=m.parameters(), update_grad=False) is None and m.w.grad is not None
    norm = sc(m(torch.ones(2, 1))[0], opt, clip_grad=1.0, parameters=m.parameters())
    assert float(norm) > 0
    args = types.SimpleNamespace(output_dir=str(tmp_path), resume=None)
    misc.save_model(args, 3, m, m, opt, sc)
    ck = torch.load(tmp_path / "checkpoint-3.pth", weights_only=False)
    assert set(ck) == {"model", "optimizer", "epoch", "scaler", "args"} and ck["epoch"] == 3
    args.resume = str(tmp_path / "checkpoint-3.pth")
    m2 = _Stub()
    misc.load_model(args, m2, torch.optim.SGD(m2.parameters(), lr=0.1), sc)
    assert torch.equal(m2.w, m.w) and args.start_epoch == 4
    assert misc.all_reduce_mean(1.5) == 1.5 and misc.get_world_size() == 1 and misc.is_main_process()


def test_sincos_table_is_the_reference_table():
    from util.pos_embed import get_2d_sincos_pos_embed
    d = np.load(os.path.join(G, "sincos.npz"))
    for dim, grid in [(128, 4), (64, 4), (768, 4), (512, 4)]:
        assert np.array_equal(get_2d_sincos_pos_embed(dim, grid, cls_token=True), d[f"full_{dim}_{grid}"])
    for dim, grid in [(768, 14), (512, 14), (1024, 16), (1280, 16)]:
        assert np.array_equal(get_2d_sincos_pos_embed(dim, grid, cls_token=True).reshape(-1)[d[f"idx_{dim}_{grid}"]], d[f"val_{dim}_{grid}"])


def test_crop_box_sampler_consumes_the_cpu_rng_like_the_reference():
    from models_mae.MAE_ViT_MsLd import sample_crop_box
    d = np.load(os.path.join(G, "crop.npz"))
    torch.manual_seed(0)
    assert [list(sample_crop_box(224, (0.25, 0.75))) for _ in range(16)] == d["rrc_seed0_224"].tolist()
    torch.manual_seed(123)
    assert [list(sample_crop_box(64, (0.25, 0.75))) for _ in range(16)] == d["rrc_seed123_64"].tolist()


def test_checkpoint_key_mapping_to_vit_and_back():
    """the --transform_checkpoint_keys mapping of main_finetune.py:553-586 (timm path) and its inverse."""
    import json
    import os
    from util.checkpoint_keys import from_vit_keys, to_vit_keys
    man = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "manifest.json")))
    name = "mae_vit_base_MsLdCeCd" if "mae_vit_base_MsLdCeCd" in man else sorted(man)[0]
    entry = man[name]
    keys = list(entry["keys"]) if isinstance(entry, dict) and "keys" in entry else list(entry)
    keys = [k[0] if isinstance(k, (list, tuple)) else k for k in keys]
    sd = {k: i for i, k in enumerate(keys)}
    vit = to_vit_keys(sd)
    assert "pos_embed" in vit and "norm.weight" in vit and "norm.bias" in vit and "cls_token" in vit
    assert "patch_embed.proj.weight" in vit and "patch_embed.proj.bias" in vit
    assert vit["pos_embed"] == sd["encoder_pos_embed"] and vit["blocks.0.attn.qkv.weight"] == sd["encoder.0.attn.qkv.weight"]
    assert not any(k.startswith(("decoder", "mask_token", "predictor", "encoder")) for k in vit)
    n_enc = sum(1 for k in keys if k.startswith("encoder."))
    assert sum(1 for k in vit if k.startswith("blocks.")) == n_enc
    back = from_vit_keys(vit)
    assert back == {k: v for k, v in sd.items() if "encoder" in k or k in ("cls_token", "patch_embed.proj.weight", "patch_embed.proj.bias")}


def test_viz_and_metrics_host_helpers(tmp_path):
    """util/viz.py / util/metrics.py (eval side): file-name helper, image preparation (PIL path of the reference:
    RandomResizedCrop box drawn with the torchvision rule, bicubic resize, plot statistics), element-wise metrics, and the
    explicit error for what needs the GPU (ssim metrics)."""
    from PIL import Image
    from util import metrics, viz
    assert viz.title_to_fname("mae_vit_base - epoch 25 - img 1.jpg") == "mae_vit_base_epoch_25_img_1_jpg"
    rng = np.random.default_rng(0)
    arr = rng.integers(0, 256, (90, 120, 3), dtype=np.uint8)
    path = str(tmp_path / "a.png")
    Image.fromarray(arr).save(path)
    img = viz.prepare_image(path, 64)
    assert img.shape == (64, 64, 3) and img.dtype == np.float64
    back = img * viz.image_std + viz.image_mean
    assert 0.0 <= back.min() and back.max() <= 1.0
    a = viz.prepare_image(path, 64, random_crop=True, crop_seed=7)
    b = viz.prepare_image(path, 64, random_crop=True, crop_seed=7)
    c = viz.prepare_image(path, 64, random_crop=True, crop_seed=8)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    torch.manual_seed(7)
    from util.gpu_input import resized_crop_box
    i, j, h, w = resized_crop_box(90, 120, scale=(0.25, 1.0))
    want = np.array(Image.fromarray(arr).resize((64, 64), Image.BICUBIC, box=(j, i, j + w, i + h)).resize((64, 64), resample=None)) / 255.0
    np.testing.assert_allclose(a, (want - viz.image_mean) / viz.image_std)
    u, v = torch.rand(1, 8, 8, 3), torch.rand(1, 8, 8, 3)
    assert metrics.calc_metric(u, v, "mse") == pytest.approx(((u - v) ** 2).mean().item())
    assert metrics.calc_metric(u, v, "SSD") == pytest.approx(((u - v) ** 2).sum().item())
    assert metrics.calc_metric(u, v, "sad") == pytest.approx((u - v).abs().sum().item())
    assert metrics.calc_metric(u, v, "mae") == pytest.approx((u - v).abs().mean().item())
    assert set(metrics.METRICS_DICT) == {"mse", "mae", "l1", "l2", "ssim", "ms_ssim"}
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            metrics.calc_metric(u, v, "ssim")
    # the reference's matplotlib figures are plotting UI: not rebuilt
    assert not hasattr(viz, "plot_reconstruction") and not hasattr(viz, "plot_image")


def test_output_dir_is_never_overwritten(tmp_path):
    """main_pretrain.py:470-490: without --resume a single-process run moves to out_<name>+N, a distributed run refuses a directory
    that already holds checkpoints; with --resume the directory is kept."""
    import main_pretrain
    base = tmp_path / "out_m"
    assert main_pretrain.protect_output_dir(str(base), "m", resume=None, distributed=False) == str(base)
    base.mkdir()
    assert main_pretrain.protect_output_dir(str(base), "m", resume=None, distributed=False) == str(tmp_path / "out_m+1")
    (tmp_path / "out_m+1").mkdir()
    assert main_pretrain.protect_output_dir(str(base), "m", resume=None, distributed=False) == str(tmp_path / "out_m+2")
    assert main_pretrain.protect_output_dir(str(tmp_path / "out_m+1"), "m", resume=None, distributed=False) == str(tmp_path / "out_m+2")
    assert main_pretrain.protect_output_dir(str(base), "m", resume="x.pth", distributed=False) == str(base)
    assert main_pretrain.protect_output_dir(str(base), "m", resume=None, distributed=True) == str(base)   # exists, but no checkpoints
    (base / "checkpoint-0.pth").write_bytes(b"")
    with pytest.raises(ValueError, match="Checkpoints would be overwritten"):
        main_pretrain.protect_output_dir(str(base), "m", resume=None, distributed=True)
    assert main_pretrain.protect_output_dir(str(base), "m", resume="x.pth", distributed=True) == str(base)


def test_flat_params_version_stamp_sees_parameter_writes():
    """The bf16 weight mirror is recast when the stamp moves: every in-place write through torch has to move it (the flat
    buffer's own version counter does not move after `p.data = view`)."""
    import models_mae
    from csmae_hip.engine import FlatParams
    micro = dict(dim_model=64, encoder_num_layers=1, encoder_num_heads=2, decoder_embed_dim=32, decoder_num_layers=1, decoder_num_heads=2)
    m = models_mae.MAE_ViT_Baseline(**micro, input_size=32, patch_size="16")
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    flat = FlatParams(m, "cpu")
    assert flat.g.numel() == flat.total + 8 and flat.gate.data_ptr() == flat.g.data_ptr() + flat.total * 4
    assert m.cls_token.data_ptr() == flat.p.data_ptr() + flat.slots["cls_token"][0] * 4
    stamps = [flat.version_stamp()]
    m.load_state_dict(sd)
    stamps.append(flat.version_stamp())
    opt = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-3)
    for p in m.parameters():
        if p.requires_grad:
            p.grad = torch.ones_like(p)
    opt.step()
    stamps.append(flat.version_stamp())
    with torch.no_grad():
        m.decoder_pred.weight.mul_(0.5)
    stamps.append(flat.version_stamp())
    assert all(b > a for a, b in zip(stamps, stamps[1:])), stamps
    assert torch.equal(flat.P("decoder_pred.weight"), m.decoder_pred.weight)  # the write landed in the flat buffer
    flat.lp_stamp = flat.version_stamp()
    flat.mark_changed()
    assert flat.lp_stamp is None


def test_debug_gate_and_trace_are_inert_by_default(monkeypatch):
    """CSMAE_DEBUG="key[=value],..." is the one gate for settled A/B aids (csmae_hip.debug_opt); the roctx ranges of csmae_hip.trace are no-ops
    without a profiler and must nest / unwind on exceptions."""
    import csmae_hip
    from csmae_hip import trace
    monkeypatch.delenv("CSMAE_DEBUG", raising=False)
    assert csmae_hip.debug_opt("zero_main") is None and csmae_hip.debug_opt("dw_slots", "160") == "160"
    monkeypatch.setenv("CSMAE_DEBUG", "zero_main,dw_slots=96, bwd_main_cus=0:128")
    assert csmae_hip.debug_opt("zero_main") == "1" and csmae_hip.debug_opt("dw_slots") == "96" and csmae_hip.debug_opt("bwd_main_cus") == "0:128"
    assert csmae_hip.debug_opt("zero") is None      # keys match whole, not by prefix
    with trace.range_("csmae.test"):
        with trace.range_("csmae.test.inner"):
            pass
    with pytest.raises(RuntimeError):
        with trace.range_("csmae.test"):
            raise RuntimeError("unwinds")
