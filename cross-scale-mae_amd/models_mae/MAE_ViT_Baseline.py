"""MAE ViT (reference models_mae/MAE_ViT_Baseline.py): same constructor signature, attributes, parameter names,
registration order and seeded initialisation; forward/backward run on the MI355X through csmae_hip.Engine."""
from functools import partial

import torch
import torch.nn as nn

from util.pos_embed import get_2d_sincos_pos_embed

from ._holders import Block, PatchEmbed
from .MAE_ViT_Shared import SSIM_LOSSES, MAE_ViT_Shared


class _StepFn(torch.autograd.Function):
    """One coarse autograd node for the whole step: forward = HIP forward, backward = hand-written HIP reverse pass that
    writes straight into the flat gradient buffer (parameter grads are attached there, not returned).  The node's only
    differentiable input is a zero-dim anchor: handing it the ~250 parameters instead made autograd walk 250 AccumulateGrad nodes
    (device guards only, no work) after every backward — 0.7 ms per step with the GPU idle behind it."""

    @staticmethod
    def forward(ctx, model, engine, imgs, mask_ratio, noise, box, anchor, img1=None):
        ws = engine.forward(imgs, mask_ratio, noise, box, model.training, img1=img1, expect_backward=True)   # (the engine may start backward work that needs no upstream gradient)
        ctx.model, ctx.engine, ctx.gen = model, engine, engine.gen
        ctx.set_materialize_grads(False)
        outs = model._outputs(engine, ws, imgs.shape[0])
        ctx.mark_non_differentiable(*outs[2:])
        return outs

    @staticmethod
    def backward(ctx, gloss, gpred, *rest):
        if gpred is not None:
            raise NotImplementedError("gradients w.r.t. the returned prediction are not supported: only the loss is differentiable")
        if gloss is not None:
            model = ctx.model
            grads = [p.grad for p in model.parameters() if p.requires_grad]
            ctx.engine.backward(gloss, accumulate=any(g is not None for g in grads), gen=ctx.gen)
        return (None,) * 8


def _check_ssim_geometry(C):
    if C != 3:  # MAE_ViT_Shared.py:194-199: the mask is repeated for 3 channels and then un-patchified with input_channels
        raise ValueError(f"the ssim family of losses needs input_channels == 3 (got {C}): the reference builds its pixel mask for 3 channels "
                         "(MAE_ViT_Shared.py:194-196) and fails to reshape it otherwise")


class MAE_ViT_Baseline(MAE_ViT_Shared):
    """Masked Autoencoder with VisionTransformer backbone"""

    VARIANT = "Baseline"

    def __init__(self, input_size=128, input_channels=3, patch_size=16, mask_ratio=0.75, dim_model=1024,
                 encoder_num_layers=24, encoder_num_heads=16, decoder_embed_dim=512, decoder_num_layers=8, decoder_num_heads=16,
                 residual_norm_style="post", residual_dropout=0.0, ffn_name="MLP", ffn_activation="gelu", ffn_ratio=4, ffn_dropout=0.0,
                 attn_name="scaled_dot_product", attn_dropout=0.0, norm_layer=partial(nn.LayerNorm, eps=1e-6), use_xformers=False,
                 device=None, **kwargs):
        super().__init__(**kwargs)
        self.input_size = input_size
        self.input_channels = input_channels
        self.patch_size = int(patch_size)
        self.dim_model = dim_model
        self.decoder_embed_dim = decoder_embed_dim
        self.mask_ratio = mask_ratio
        self.use_xformers = use_xformers
        self.device = device
        assert input_size % self.patch_size == 0
        if use_xformers:
            raise NotImplementedError("use_xformers=True selects xFormers' post-norm block zoo (MAE_ViT_Baseline.py:94-157): a different "
                                      "network built on an un-vendored package, outside the MI355X hot-path scope (SURVEY.md §2 row 3)")
        assert attn_name == "scaled_dot_product", f"Attention {attn_name} not supported (timm path uses scaled_dot_product)"
        assert ffn_name == "MLP", f"Feedforward {ffn_name} not supported (timm path uses MLP)"
        assert ffn_activation == "gelu", f"Feedforward activation {ffn_activation} not supported (timm path uses gelu)"
        if residual_dropout or ffn_dropout or attn_dropout:
            raise NotImplementedError("dropout / drop-path > 0 is not implemented on the MI355X path (reference defaults are 0)")
        assert dim_model % encoder_num_heads == 0 and decoder_embed_dim % decoder_num_heads == 0
        self._geom = dict(He=encoder_num_heads, Ne=encoder_num_layers, Hd=decoder_num_heads, Nd=decoder_num_layers, ffn_ratio=ffn_ratio)
        if ffn_ratio != 4:
            raise NotImplementedError("ffn_ratio != 4 is not wired on the MI355X path")

        self.patch_embed = PatchEmbed(input_size, self.patch_size, input_channels, dim_model)
        self.num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim_model))
        self.encoder_pos_embed = nn.Parameter(torch.zeros(1, self.num_patches + 1, dim_model), requires_grad=False)
        self.decoder_embed = nn.Linear(dim_model, decoder_embed_dim, bias=True)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, decoder_embed_dim))
        self.decoder_pos_embed = nn.Parameter(torch.zeros(1, self.num_patches + 1, decoder_embed_dim), requires_grad=False)
        self.encoder = nn.ModuleList([Block(dim_model, encoder_num_heads, ffn_ratio, norm_layer) for _ in range(encoder_num_layers)])
        self.decoder = nn.ModuleList([Block(decoder_embed_dim, decoder_num_heads, ffn_ratio, norm_layer) for _ in range(decoder_num_layers)])
        self.decoder_pred = nn.Linear(decoder_embed_dim, self.patch_size ** 2 * input_channels, bias=True)
        self.decoder_norm = norm_layer(decoder_embed_dim)
        self.encoder_norm = norm_layer(dim_model)  # kept for checkpoint compatibility: its output is discarded (MAE_ViT_Baseline.py:264)
        self.initialize_weights()
        self.compute_dtype = None  # None: bf16 MFMA under torch autocast, exact fp32 otherwise; or force torch.bfloat16 / torch.float32
        self._flat = None
        self._engines = {}
        self._test_draws = None

    # ---- MAE_ViT_Baseline.py:201-241
    def initialize_weights(self):
        grid = int(self.patch_embed.num_patches ** 0.5)
        for pe in (self.encoder_pos_embed, self.decoder_pos_embed):
            pe.data.copy_(torch.from_numpy(get_2d_sincos_pos_embed(pe.shape[-1], grid, cls_token=True)).float().unsqueeze(0))
        w = self.patch_embed.proj.weight.data
        torch.nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
        torch.nn.init.normal_(self.cls_token, std=0.02)
        torch.nn.init.normal_(self.mask_token, std=0.02)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    # ---- engine plumbing
    def _cfg(self):
        g = self._geom
        p, C = self.patch_size, self.input_channels
        G = self.input_size // p
        return dict(S=self.input_size, C=C, p=p, G=G, L=G * G, P=p * p * C, D=self.dim_model, He=g["He"], Ne=g["Ne"],
                    Dd=self.decoder_embed_dim, Hd=g["Hd"], Nd=g["Nd"], Hp=getattr(self, "predictor_hidden_size", 0),
                    loss=self.loss, norm_pix=bool(self.norm_pix_loss), reduction=getattr(self, "ms_decoder_loss_reduction", "sum"),
                    loss_cd=getattr(self, "loss_cd", self.loss), loss_e=getattr(self, "loss_e", self.loss), variant=self.VARIANT)

    def _engine(self, imgs):
        from csmae_hip.engine import Engine, FlatParams
        if not imgs.is_cuda:
            raise RuntimeError("this model runs only on an MI355X: move the model and the batch to 'cuda' (there is no CPU fallback; "
                               "the CPU restatement lives in oracle/ and is test infrastructure)")
        first = next(self.parameters())
        if not first.is_cuda:
            raise RuntimeError("model parameters are on the CPU: call model.to('cuda') first")
        if self._flat is None or not self._flat.still_homed():
            self._flat = FlatParams(self, first.device)
            self._engines = {}
        dtype = self.compute_dtype
        if dtype is None:
            dtype = torch.bfloat16 if torch.is_autocast_enabled() else torch.float32
        if self.loss in SSIM_LOSSES:
            _check_ssim_geometry(self.input_channels)
        if dtype not in self._engines:
            self._engines[dtype] = Engine(self, self._flat, self._cfg(), dtype)
        return self._engines[dtype]

    def mark_parameters_changed(self):
        """Tell the engine that parameters were written behind torch's back (`p.data.mul_()`, raw pointers): in-place writes through
        torch (load_state_dict, torch.optim, `p.mul_()`) are noticed by themselves, see csmae_hip.engine.FlatParams.version_stamp."""
        if self._flat is not None:
            self._flat.mark_changed()

    def _draw(self, imgs, mask_ratio, mask_seed, consistent_mask=False):
        """The reference's RNG draws, in its order (MAE_ViT_Baseline.py:301-302,251 -> MAE_ViT_Shared.py:66)."""
        N, L = imgs.shape[0], self.num_patches
        hook, self._test_draws = self._test_draws, None
        if mask_seed is not None:
            torch.manual_seed(mask_seed)
        noise = hook["noise"][0].to(imgs.device) if hook else torch.rand(N, L, device=imgs.device)
        return noise, None

    def _outputs(self, eng, ws, N):
        c = eng.cfg
        pred = ws.pred.view(ws.B2, ws.Td, c["P"])
        lat = (ws.lat32 if ws.lat32 is not None else ws.enc["x"][c["Ne"]]).view(ws.B2, ws.Te, c["D"])   # fp32 (copy of a bf16 stream's latent)
        emb = ws.emb32.view(ws.B2, ws.Td, c["Dd"])
        return (ws.losses[0].clone(), pred[:N, 1:, :], ws.mask[:N], lat[:N], emb[:N])

    def _run(self, imgs, mask_ratio, noise, box, img1=None):
        if img1 is not None:
            if img1.shape != imgs.shape or img1.device != imgs.device:
                raise AssertionError(f"the two views must have the same shape and device: {tuple(imgs.shape)} vs {tuple(img1.shape)}")
            img1 = img1.contiguous().float()
        if imgs.dim() != 4 or imgs.shape[1] != self.input_channels or imgs.shape[2] != self.input_size or imgs.shape[3] != self.input_size:
            raise AssertionError(f"input {tuple(imgs.shape)} does not match (N, {self.input_channels}, {self.input_size}, {self.input_size})")
        imgs = imgs.contiguous().float()
        eng = self._engine(imgs)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            anchor = self.__dict__.get("_anchor")
            if anchor is None or anchor.device != imgs.device:
                anchor = self.__dict__["_anchor"] = torch.zeros((), device=imgs.device, requires_grad=True)
            return _StepFn.apply(self, eng, imgs, mask_ratio, noise, box, anchor, img1)
        ws = eng.forward(imgs, mask_ratio, noise, box, self.training, img1=img1)
        # Outside the training fast path the outputs are fresh tensors, as in the reference.  Under autograd (the path above) the
        # prediction / latents / embeddings are VIEWS of the engine's activation workspace: valid until the model's next forward,
        # which overwrites them — clone what must outlive it (the loss is always a fresh scalar).
        return tuple(o.clone() for o in self._outputs(eng, ws, imgs.shape[0]))

    # ---- public API (MAE_ViT_Baseline.py:299-320)
    def forward(self, imgs, mask_ratio=0.75, mask_seed=None, return_embeds=False):
        noise, box = self._draw(imgs, mask_ratio, mask_seed)
        loss, pred, mask, enc, dec = self._run(imgs, mask_ratio, noise, box)
        return (loss, pred, mask) if not return_embeds else (loss, pred, mask, enc, dec)

    def _engine_single(self, x):
        """Single-view inference engine over the same flat parameters (stand-alone forward_encoder / forward_decoder)."""
        from csmae_hip.engine import Engine
        eng = self._engine(x)
        dt_ = self.compute_dtype if self.compute_dtype is not None else (torch.bfloat16 if torch.is_autocast_enabled() else torch.float32)
        cache = self.__dict__.setdefault("_single_engines", {})
        key = (id(eng.flat), dt_)
        if key not in cache:
            cfg = dict(eng.cfg)
            cfg["variant"] = "Baseline"
            cache.clear()
            cache[key] = Engine(self, eng.flat, cfg, dt_)
        return cache[key]

    @torch.no_grad()
    def forward_encoder(self, x, mask_ratio):
        """(latent [N, keep+1, D], mask [N, L], ids_restore [N, L]) — MAE_ViT_Baseline.py:243-266.  Inference only (no autograd
        graph): training goes through `forward`, whose backward is one hand-written pass."""
        x = x.contiguous().float()
        noise = torch.rand(x.shape[0], self.num_patches, device=x.device)  # MAE_ViT_Shared.py:66
        return self._engine_single(x).encode(x, mask_ratio, noise)

    @torch.no_grad()
    def forward_decoder(self, x, ids_restore):
        """(pred [N, L, p*p*C], x_embed [N, L+1, Dd]) — MAE_ViT_Baseline.py:268-297.  Inference only."""
        return self._engine_single(x).decode(x, ids_restore)

    @torch.no_grad()
    def forward_loss(self, target, pred, mask=None, patch_embed_psize=None, input_channels=None):
        """Reconstruction loss of `--loss` (MAE_ViT_Shared.py:269-290), the reference's signature: `target` is a full image batch
        [N, C, H, W] when `patch_embed_psize` and `input_channels` are given (patchified, and pixel-normalised under norm_pix_loss, by
        `process_target` :97-111 — fused into the HIP loss kernel: the image is read once), otherwise an already patchified
        [N, L, p*p*C] target that is compared as it is (no norm_pix step, as in the reference).  Runs the same fused HIP loss kernels
        `forward` uses.  Inference only."""
        from csmae_hip import SSIM_KINDS, ops
        N, L, P = pred.shape
        kind, ssim = self.loss, SSIM_KINDS.get(self.loss)
        from_image = patch_embed_psize is not None and input_channels is not None
        if from_image:
            p, C = int(patch_embed_psize), int(input_channels)
            norm_pix = bool(self.norm_pix_loss)
            imgs = target.contiguous().float()
            if imgs.dim() != 4 or imgs.shape[1] != C or imgs.shape[2] != imgs.shape[3] or imgs.shape[2] % p:
                raise AssertionError(f"target {tuple(target.shape)} is not an (N, {C}, S, S) image batch with S % {p} == 0")  # patchify :31
            S = imgs.shape[2]
        else:
            if ssim is not None:  # forward_loss_ssim un-patchifies with kwargs['patch_embed_psize'] = None (MAE_ViT_Shared.py:181-185)
                raise TypeError(f"loss={self.loss!r} compares images: pass patch_embed_psize and input_channels (the reference fails in "
                                "unpatchify(x, None, None), MAE_ViT_Shared.py:181)")
            if target.shape != pred.shape:
                raise RuntimeError(f"patchified target {tuple(target.shape)} does not match pred {tuple(pred.shape)}")
            # compared as it is (no process_target, no pixel normalisation): every [P] row is handed to the fused kernel as a
            # one-patch "image" with P channels (a reshape, nothing is copied or permuted), so any feature size works
            norm_pix, p, C, S = False, 1, P, 1
            imgs = target.contiguous().float().reshape(N * L, P, 1, 1)
            pred = pred.reshape(N * L, 1, P)
            mask = None if mask is None else mask.reshape(N * L, 1)
            N, L = N * L, 1
        if (S // p) ** 2 != L or P != p * p * C:
            raise RuntimeError(f"pred {tuple(pred.shape)} does not match a {S}x{S} image with {p}-pixel patches and {C} channels")
        dev = imgs.device
        full = torch.zeros(N, L + 1, P, device=dev, dtype=torch.float32)  # the kernels index predictions with the cls row in place
        full[:, 1:, :] = pred
        full = full.view(N * (L + 1), P)
        rowloss = torch.empty(N * L, device=dev, dtype=torch.float32)
        mm = None
        m = torch.ones(N, L, device=dev, dtype=torch.float32) if mask is None else mask.to(torch.float32).contiguous()
        if ssim is not None:  # MAE_ViT_Shared.py:165-267 (no mask = every patch compared, :187-189)
            _check_ssim_geometry(C)
            kind = ssim[0]
            ws = torch.empty(ops.ssim_workspace_floats(N, C, S, p, ssim[1]), device=dev, dtype=torch.float32)
            terms = torch.empty(2, device=dev, dtype=torch.float32)
            ops.ssim_fwd(ssim[1], norm_pix, imgs, None, full, m, ws, terms, N, N, C, S, p)
        if kind == "bce":
            mm = torch.empty(2, device=dev, dtype=torch.float32)
            ops.target_minmax(imgs, None, torch.empty(N * L * 2, device=dev, dtype=torch.float32), mm, N, N, C, S, p, norm_pix)
        if kind == "none":
            rowloss.zero_()
        else:
            ops.recon_loss_fwd(kind, norm_pix, imgs, None, full, mm, rowloss, N, N, C, S, p)
        losses = torch.zeros(8, device=dev, dtype=torch.float32)
        ops.loss_finalize(N * L, 1, rowloss, m, 1.0, losses)
        if ssim is not None:
            ops.ssim_apply(kind == "none", 1, ssim[2], 1.0, terms, losses)
        return losses[1].clone()
