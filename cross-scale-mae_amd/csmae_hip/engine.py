"""Host-side step engine: sequences the HIP kernels of one Cross-Scale MAE pre-training step.

Mirrors (by behaviour, not by structure) `models_mae/MAE_ViT_Baseline.forward_encoder/_decoder/forward`
(MAE_ViT_Baseline.py:243-320), `MAE_ViT_MsLd.forward` (MAE_ViT_MsLd.py:37-77) and the loss heads of
`MAE_ViT_MsLd{Le,Cd,LeCd,CeCd}.py`, but MI355X-first:

* the two views (original + crop) are run as ONE batch of 2N samples through shared weights;
* patch-embed runs only on the kept 25 % of patches (same per-token arithmetic);
* all parameters / gradients / AdamW moments live in flat HBM buffers (one bf16 mirror for the MFMA GEMMs),
  so the optimizer and the RCCL all-reduce are single-buffer operations;
* autograd sees one coarse node: `backward()` below is the hand-written reverse pass;
* every per-step scalar (crop box, upstream gradient) is read by the kernels from device memory, so the
  whole sequence is hipGraph-capturable.

PyTorch is used for memory, streams and the RNG draws the reference makes (`torch.rand` on the device
generator for the masking noise, the CPU generator for the crop box — Appendix E5 of SURVEY.md).
"""
from __future__ import annotations

import contextlib
import math
import os
import weakref
from typing import Dict, Optional

import torch

from . import BF16, EPI_ATOMIC, EPI_DGELU, EPI_GELU, EPI_NONE, EPI_RESID, F32, SSIM_KINDS, debug_opt, ops
from . import trace

_FROZEN = ("encoder_pos_embed", "decoder_pos_embed")


def _round_up(x, m):
    return (x + m - 1) // m * m


class FlatParams:
    """All nn.Parameters of a model re-homed into one fp32 buffer (8-element aligned slots) + flat grad buffer."""

    def __init__(self, module: torch.nn.Module, device):
        self.names, self.slots = [], {}
        off = 0
        params = list(module.named_parameters())
        for name, p in params:
            n = p.numel()
            self.slots[name] = (off, n, tuple(p.shape))
            self.names.append(name)
            off += _round_up(n, 8)
        self.total = off
        self.p = torch.zeros(off, device=device, dtype=torch.float32)
        # gradient buffer + one tail slot: the update gate (the step's loss; see csmae_adamw).  It sits inside the last all-reduce
        # range, so with N > 1 every rank gates its optimizer step on the same rank-averaged value.
        self.g = torch.zeros(off + 8, device=device, dtype=torch.float32)
        self.gate = self.g[off:off + 1]
        self.w_lp: Optional[torch.Tensor] = None  # bf16 mirror (allocated on demand)
        self.lp_stamp = None   # sum of the parameters' version counters when the mirror was last made consistent (None: stale)
        self.raw_writes = 0    # writes to the masters that move no version counter (FusedAdamW's kernels, broadcasts): bumped by their authors
        self.params = dict(params)
        with torch.no_grad():
            for name, p in params:
                o, n, shape = self.slots[name]
                view = self.p[o:o + n].view(shape)
                view.copy_(p.data.to(device))
                p.data = view
        self.grad_views = {name: self.g[o:o + n].view(shape) for name, (o, n, shape) in self.slots.items()}
        self._by_id = {id(p): name for name, p in params}
        FlatParams._registry.append(weakref.ref(self))

    _registry = []

    @classmethod
    def owner_of(cls, p):
        """The live flat buffer a parameter is homed in (None if it is not)."""
        alive = []
        found = None
        for ref in cls._registry:
            f = ref()
            if f is None:
                continue
            alive.append(ref)
            name = f._by_id.get(id(p))
            if name is not None and p.is_cuda and p.data_ptr() == f.p.data_ptr() + f.slots[name][0] * 4:
                found = f
        cls._registry[:] = alive
        return found

    def slot_of(self, p):
        return self.slots[self._by_id[id(p)]]

    def version_stamp(self) -> int:
        """Moves whenever somebody writes a Parameter in place through torch (load_state_dict, torch.optim, `p.mul_()` ...).  The
        flat buffer's own counter does not: after `p.data = view` every Parameter keeps a version counter of its own."""
        return sum(p._version for p in self.params.values())

    def mark_changed(self):
        """Writes torch cannot see (`p.data.mul_()`, raw pointers) must call this so that the bf16 mirror is recast."""
        self.lp_stamp = None
        self.raw_writes += 1

    def still_homed(self) -> bool:
        first = self.params[self.names[0]]
        return first.data_ptr() == self.p.data_ptr() + self.slots[self.names[0]][0] * 4 and first.is_cuda

    def P(self, name):
        o, n, shape = self.slots[name]
        return self.p[o:o + n].view(shape)

    def G(self, name):
        return self.grad_views[name]

    def lp(self, name, rows=None):
        o, n, shape = self.slots[name]
        t = self.w_lp[o:o + n]
        return t.view(shape[0], -1) if len(shape) > 1 else t


class Workspace:
    """Per-(batch, keep) activation arena.  Everything the backward needs is kept (12 GB at ViT-B/N=128 — 4 % of HBM)."""

    def __init__(self, eng: "Engine", N: int, keep: int):
        c, dev, T = eng.cfg, eng.device, eng.act_dtype
        self.N, self.keep = N, keep
        V = eng.views
        B2 = V * N
        self.B2, self.Te, self.Td = B2, keep + 1, c["L"] + 1
        Me, Md = B2 * self.Te, B2 * self.Td
        self.Me, self.Md = Me, Md
        D, Dd, L, Pp = c["D"], c["Dd"], c["L"], eng.Pp
        f32 = dict(device=dev, dtype=torch.float32)
        lp = dict(device=dev, dtype=T)
        rs = dict(device=dev, dtype=eng.res_dtype)   # the residual stream x / x_mid (and its gradient): fp32, or bf16 in throughput mode
        E = torch.empty
        self.imgs_crop = E(N, c["C"], c["S"], c["S"], **f32) if V == 2 else None
        self.box = torch.zeros(4, device=dev, dtype=torch.int32)
        self.noise = E(B2, L, **f32)
        self.ids_restore = E(B2, L, device=dev, dtype=torch.long)
        self.mask = E(B2, L, **f32)
        self.ids_keep = E(B2, max(keep, 1), device=dev, dtype=torch.int32)
        self.a_pe = E(B2 * max(keep, 1), Pp, **lp)
        self.tok = E(B2 * max(keep, 1), D, **f32)

        def stack(nl, M, Dm, H):
            d = dict(x=E(nl + 1, M, Dm, **rs), xm=E(nl, M, Dm, **rs), y1=E(nl, M, Dm, **lp), y2=E(nl, M, Dm, **lp),
                     qkv=E(nl, M, 3 * Dm, **lp), o=E(nl, M, Dm, **lp), h=E(nl, M, 4 * Dm, **lp),
                     pre=E(nl, M, 4 * Dm, device=dev, dtype=torch.uint8 if eng.gp_q8 else T),   # gelu'(pre-activation): one byte per element in throughput mode
                     st=E(nl, 4, M, **f32), lse=E(nl, M * H, **f32))
            return d
        self.enc = stack(c["Ne"], Me, D, c["He"])
        self.dec = stack(c["Nd"], Md, Dd, c["Hd"])
        if eng.fp8 and eng.fp8_dw:   # fp8 weight gradients: the fp8 copies of y1 / o / y2 / h that the forward products read are KEPT per layer (they are the dW products' X operands)
            u8 = dict(device=dev, dtype=torch.uint8)
            for S, nl, M, Dm in ((self.enc, c["Ne"], Me, D), (self.dec, c["Nd"], Md, Dd)):
                S.update(y1_8=E(nl, M, Dm, **u8), o_8=E(nl, M, Dm, **u8), y2_8=E(nl, M, Dm, **u8), h_8=E(nl, M, 4 * Dm, **u8))
        self.lat_lp = E(Me, D, **lp) if (T != torch.float32 and eng.res_dtype == torch.float32) else None
        self.lat32 = E(Me, D, **f32) if eng.res_dtype != torch.float32 else None   # fp32 copy of the latent for the loss heads / outputs
        self.z = E(Me, Dd, **f32)
        self.emb_lp = E(Md, Dd, **lp)
        self.emb32 = E(Md, Dd, **f32)
        self.dn_st = E(2, Md, **f32)
        # decoder_pred's output: bf16 in throughput mode (what autocast hands the reference's loss; the reconstruction head reads it twice and the
        # product writes it once: 155 -> 77 MB each), fp32 in parity mode and for the ssim family (its kernels take fp32 planes)
        self.pred = E(Md, c["P"], **(lp if (T != torch.float32 and c["loss"] not in SSIM_KINDS and c["P"] % 8 == 0) else f32))
        self.rowloss = E(B2 * L, **f32)
        self.minmax = E(4, **f32)
        self.mm_scratch = E(B2 * L * 2, **f32) if c["loss"] == "bce" else None
        if c["loss"] in SSIM_KINDS:  # image planes per pyramid level + statistics, and the fp32 gradient share
            _, levels, _ = SSIM_KINDS[c["loss"]]
            self.ssim_ws = E(ops.ssim_workspace_floats(B2, c["C"], c["S"], c["p"], levels), **f32)
            self.ssim_terms = E(2, **f32)
            self.ssim_extra = E(B2 * L, c["P"], **f32)
        self.losses = torch.zeros(8, **f32)
        if eng.has_pred:
            Hp = c["Hp"]
            self.pin = E(N * L, Dd, **lp)
            self.u = E(N * L, Hp, **lp)
            self.r = E(N * L, Hp, **lp)
            self.v = E(N * L, Dd, **f32)
            self.bn_st = E(2, L, **f32)
            self.cd_partial = E(512, **f32)
            self.dv = E(N * L, Dd, **lp)
            self.dr = E(N * L, Hp, **lp)
            self.dpin = E(N * L, Dd, **lp)
            self.bn_tmp = torch.zeros(2, L, **f32)   # BatchNorm's dgamma / dbeta of the speculative (unit-gradient) backward chain
            self.one = torch.ones(1, **f32)
        if eng.has_ce:
            self.zc = E(B2, D, **f32)
            self.inv_norm = E(B2, **f32)
            self.E = E(B2, B2, **f32)
            self.neg = E(B2, **f32)
            self.ce_rowloss = E(B2, **f32)
            self.dpool = E(B2, D, **f32)
        if eng.has_le:
            self.e_partial = E(512, **f32)
        # backward scratch (shared by all layers)
        Mmax_e, Mmax_d = Me, Md
        self.gout = torch.ones(1, **f32)
        # Buffers that the weight-gradient stream reads rotate so that the main chain never waits for it (see Engine._block_bwd): a
        # block's ONE grouped launch reads its incoming residual gradient and the one after norm2's backward while the next block is
        # already writing its own two — five low-precision residual-gradient buffers in rotation, two each of dpre / dqkv.
        self.dres_e = E(Mmax_e, D, **f32)
        self.dres_e_lp = [E(Mmax_e, D, **lp) for _ in range(5)]
        self.dres_d = E(Mmax_d, Dd, **f32) if eng.res_dtype == torch.float32 else None
        self.dres_d_lp = [E(Mmax_d, Dd, **lp) for _ in range(5)]
        big = max(Me * 4 * D, Md * 4 * Dd)
        self.t4 = [E(big, **lp), E(big, **lp)]      # dpre
        self.t3 = [E(max(Me * 3 * D, Md * 3 * Dd), **lp) for _ in range(2)]  # dqkv
        self.t1 = E(max(Me * D, Md * Dd), **lp)          # dy / do
        self.dpred_lp = E(Md, Pp, **lp)
        self.demb = E(Md, Dd, **f32)
        self.dz_lp = E(Me, Dd, **lp)
        self.dtok_lp = E(B2 * max(keep, 1), D, **lp)
        # per-block dgamma / dbeta partial rows of every LayerNorm backward of the step (one slice each), folded by ONE launch per
        # block stack at the end instead of one small reduce per LayerNorm on the critical path
        # (<= 1024 rows per LayerNorm from csmae_layernorm_bwd, one per 128-row tile from csmae_gemm_ln_bwd)
        self.ln_part_e = E(2 * c["Ne"], max(1024, -(-Me // 128)) * 2 * D, **f32)
        self.ln_part_d = E(2 * c["Nd"] + 1, max(1024, -(-Md // 128)) * 2 * Dd, **f32)
        if eng.fp8:   # fp8 staging of the A operand (one buffer per forward stream) and the per-GEMM scale scalars of a step
            big8 = max(Me * 4 * D, Md * 4 * Dd)
            self.a8 = [E(big8, device=dev, dtype=torch.uint8) for _ in range(2)]
            nsite = 8 * (c["Ne"] + c["Nd"]) * 2 + 16
            self.fp8_amax = [torch.zeros(nsite, ops.FP8_SLOTS, **f32), torch.zeros(nsite, ops.FP8_SLOTS, **f32)]   # this step's / the previous step's max|x| per GEMM site (64 partial maxima each)
            self.q_b = [E(big8, device=dev, dtype=torch.uint8) for _ in range(2)]   # fp8 copies emitted by the fc1 / fc2-backward epilogues (h, dpre)
            self.q_a = [E(big8 // 4, device=dev, dtype=torch.uint8) for _ in range(2)]  # ... by LayerNorm forward (y1, y2) / backward (the residual gradient)
            self.fp8_dq = torch.ones(nsite, **f32)
            if eng.fp8_dw:   # fp8 twins of the rotating gradient buffers (dpre, dqkv, the five residual-gradient buffers per stack): the dW products' dY operands
                u8 = dict(device=dev, dtype=torch.uint8)
                self.t4_8 = [E(big8, **u8) for _ in range(2)]
                self.t3_8 = [E(max(Me * 3 * D, Md * 3 * Dd), **u8) for _ in range(2)]
                self.dres_e_8 = [E(Me, D, **u8) for _ in range(5)]
                self.dres_d_8 = [E(Md, Dd, **u8) for _ in range(5)]
            self.fp8_hist = False    # the previous step's amax exist (delayed scaling from the second step on)
            self.fp8_complete = False
        self.dw_ws = E(64 * 1024 * 1024, **f32)  # K-slice slabs of the weight-gradient launches (256 MiB)


class Engine:
    def __init__(self, module: torch.nn.Module, flat: FlatParams, cfg: dict, act_dtype=torch.bfloat16):
        self.module, self.cfg, self.device = module, cfg, flat.p.device
        # "fp8" (BASELINE.json configs[4]): the bf16 engine with the transformer blocks' forward and dX GEMMs on the fp8 MFMA path
        # (per-tensor scaled OCP fp8 operands: activations / weights e4m3, gradients e5m2; fp32 accumulation; weight gradients stay bf16)
        self.fp8 = act_dtype == "fp8"
        self._fp8_site = 0
        self.fp8_dw = self.fp8 and not debug_opt("fp8_bf16_dw")   # weight gradients on the fp8 MFMA path too (csmae_gemm_dw_group_fp8); A/B aid: bf16 weight gradients
        self._fp8_blk = {}      # (stack, block) -> the block's forward GEMM sites (y1, o, y2, h): shared by the two views' calls, read by the backward pass
        self._fp8_kept = {}     # (stack, block) -> which kept fp8 copies (y1_8 / o_8 / y2_8 / h_8) the last forward wrote
        self._fp8_cur = None    # (site, fp8 bytes or None) of the residual gradient the next block's fc2-backward product reads
        self._fp8_fuse_lnb = not debug_opt("fp8_no_fuse_lnb")
        self._fp8_fuse_attn = not debug_opt("fp8_no_fuse_attn")   # A/B aid: attention's fp8 copies by separate quantisation passes
        self._fp8_fuse = not debug_opt("fp8_no_fuse")   # A/B aid: every fp8 operand through the separate quantisation pass
        if self.fp8:
            act_dtype = torch.bfloat16
        if self.device.type != "cuda":
            raise RuntimeError("csmae_hip.Engine needs an MI355X (device 'cuda'): there is no CPU or eager fallback on the product path")
        self.act_dtype = act_dtype
        self.T = BF16 if act_dtype == torch.bfloat16 else F32
        # Residual stream (x, x_mid and the residual gradient): fp32 in parity mode.  In throughput mode it is bf16 as well: the
        # residual epilogues, LayerNorm forward / backward and the stack boundaries move half the bytes (they are HBM- / store-bound,
        # DESIGN §4).  CSMAE_RESID_FP32=1 keeps the fp32 stream under the bf16 GEMMs (what torch autocast does; A/B aid).
        self.res_dtype = torch.float32 if (self.T == F32 or os.environ.get("CSMAE_RESID_FP32")) else torch.bfloat16
        self.gp_q8 = self.T == BF16   # gelu' saved as 8-bit codes (csmae.h CSMAE_EPI_GELU_Q8)
        v = cfg["variant"]
        self.views = 1 if v == "Baseline" else 2
        self.has_pred = v in ("MsLdCd", "MsLdLeCd", "MsLdCeCd")
        self.has_ce = v == "MsLdCeCd"
        self.has_le = v in ("MsLdLe", "MsLdLeCd")
        self.Pp = _round_up(cfg["P"], 8) if self.T == BF16 else cfg["P"]
        self.flat = flat
        if self.T == BF16:
            if flat.w_lp is None:
                flat.w_lp = torch.zeros(flat.total, device=self.device, dtype=torch.bfloat16)
            if self.Pp != cfg["P"]:  # P = p*p*C not a multiple of 8 (ViT-H/14): zero-padded private copies of the two P-shaped weights
                self.w_pe_pad = torch.zeros(cfg["D"], self.Pp, device=self.device, dtype=torch.bfloat16)
                self.w_pred_pad = torch.zeros(self.Pp, cfg["Dd"], device=self.device, dtype=torch.bfloat16)
        self.ws: Optional[Workspace] = None
        self._saved = None
        self.gen = 0            # forward() counter: an autograd node may only run the backward of the forward it belongs to
        self._gen_done = -1
        self.side, self.main, self.aux = None, None, None
        self._fwd_streams = []
        self._so = {}            # raw stream handle -> torch stream object of the forward pass in flight (_opt_gate)
        self._events, self._ev_i, self._side_reads, self._tog = [], 0, {}, 0
        self._side_seq, self._side_waited = 0, 0   # weight-gradient launches issued / the youngest one the main stream has waited for
        self._dw_cache = {}
        for key in ("bwd_main_cus", "main_cus"):   # (a malformed experiment knob fails here, not in the middle of a backward pass)
            v = debug_opt(key)
            if v is not None and not (len(v.split(":")) == 2 and all(q.isdigit() for q in v.split(":")) and int(v.split(":")[0]) < int(v.split(":")[1])):
                raise ValueError(f"CSMAE_DEBUG {key}={v!r}: expected lo:hi (mask bits of the compute units)")
        if not str(debug_opt("dw_cus", "0")).isdigit():
            raise ValueError(f"CSMAE_DEBUG dw_cus={debug_opt('dw_cus')!r}: expected the number of CUs per XCD")
        # LayerNorm inside the GEMM epilogues of the stacks whose rows fit one workgroup (csmae_gemm_ln_fwd / _bwd: width <= 512, i.e. the decoders):
        # proj + norm2, fc2 + the next block's norm1, fc1-dX + norm2', qkv-dX + norm1'.  bf16 throughput mode with the bf16 residual stream only.
        self.ln_fuse = self.T == BF16 and not self.fp8 and self.res_dtype == torch.bfloat16 and not debug_opt("no_lnfuse")
        self.ln_fuse_fwd = self.ln_fuse and not debug_opt("no_lnfuse_fwd")   # (A/B aids: one direction only)
        self.ln_fuse_bwd = self.ln_fuse and not debug_opt("no_lnfuse_bwd")
        self._ln_fuse_mask = int(debug_opt("lnfuse_mask", "15"))   # A/B aid: bit 0 proj + norm2, 1 fc2 + next norm1, 2 fc1-dX + norm2', 3 qkv-dX + norm1'
        self.use_ks = not debug_opt("no_kslab")   # (CSMAE_DEBUG=no_kslab: forward products through csmae_gemm with the plain weight mirror; no K-slab mirror is kept)
        self._dw_slots = int(debug_opt("dw_slots", "160"))   # workgroups of a weight-gradient launch: ~5/8 of the CUs, the rest runs the main stream
        # ... per stack ("enc,dec") for the blocks' own launches: the decoder's long products (50 k tokens) run best on half the chip —
        # 128 workgroups = 4 / 8 whole K slices of its 32- and 16-tile launches (160 -> 128: -0.15 .. -0.3 ms per step; 64: +1.4 ms)
        se = os.environ.get("CSMAE_DW_SLOTS_ED", "" if debug_opt("dw_slots") else "160,128")
        self._dw_slots_ed = tuple(int(v) for v in se.split(",")) if se else None
        # weight gradients of a block: "half" = two grouped launches (fc2 + fc1 once dpre exists, proj + qkv after attention backward),
        # "block" = ONE launch when its tiles fit the slots (ViT-B: 108 encoder tiles, no K slice; 48 decoder tiles, 3 slices: 2.5 GB
        # less slab traffic per step, no fold kernel for the encoder — and measured 0.13 ms SLOWER per step on three boxes: the step is
        # bound by the main chain, which runs beside a 300-us launch worse than beside two shorter ones), "none" = one per product
        self._dw_mode = debug_opt("dw_group", "half")
        self._dw_rot = self._dw_mode != "half0"   # A/B aid: "half0" = the two-launch mode with its outgoing gradient written over the incoming one (a wait per buffer)
        if self._dw_mode == "half0":
            self._dw_mode = "half"
        if self._dw_mode not in ("block", "block_enc", "block_dec", "half", "none"):
            raise ValueError(f"CSMAE_DEBUG dw_group={self._dw_mode!r}: block, block_enc, block_dec, half or none")
        sl = flat.slots
        goff = lambda names: torch.tensor([[sl[n + ".weight"][0], sl[n + ".bias"][0]] for n in names], dtype=torch.long, device=self.device)
        self._goff_e = goff([f"encoder.{i}.norm{k}" for i in range(cfg["Ne"]) for k in (1, 2)])
        self._goff_d = goff([f"decoder.{i}.norm{k}" for i in range(cfg["Nd"]) for k in (1, 2)] + ["decoder_norm"])

    # ------------------------------------------------------------------ helpers
    def W(self, name):
        """GEMM-operand view of a weight: bf16 mirror in throughput mode, the fp32 master in parity mode."""
        if self.T == BF16:
            return self.flat.lp(name)
        t = self.flat.P(name)
        return t.view(t.shape[0], -1) if t.dim() > 1 else t

    def _refresh_lp(self, lazy_pads=False):
        """Bring the bf16 weight mirror up to date.  FusedAdamW writes the mirror in the same kernel that steps the fp32 master (through
        raw pointers: no version counter moves), so a full recast (0.7 GB of HBM traffic for ViT-B) is only needed when somebody else
        has written the parameters in place — torch.optim, load_state_dict, manual edits.  Those bump the *Parameters'* version
        counters (not the flat buffer's: `p.data = view` gave every Parameter its own), so the stamp is their sum; it lives on the
        FlatParams because every engine of a model shares the one mirror."""
        if self.T != BF16:
            return
        if not self.fp8:
            self.flat.fp8_active = False
        stamp = self.flat.version_stamp()
        if stamp != self.flat.lp_stamp:
            self._opt_gate()
            ops.cast_bf16(self.flat.p, self.flat.w_lp)
            self.flat.lp_stamp = stamp
        if self.Pp != self.cfg["P"] and not lazy_pads:
            self._opt_gate()
            self._refresh_pads(True, True)
        self._refresh_ks()

    def _refresh_pads(self, pe, pred):
        """Zero-padded bf16 copies of the two weights whose patch dimension is not a multiple of the GEMM's K step (patch 14: P = 588), from the masters."""
        P = self.cfg["P"]
        if pe:
            self.w_pe_pad[:, :P].copy_(self.flat.P("patch_embed.proj.weight").view(self.cfg["D"], P))
        if pred:
            self.w_pred_pad[:P].copy_(self.flat.P("decoder_pred.weight"))

    def _refresh_ks(self):
        """K-slab mirrors of the blocks' Linear weights (csmae.h csmae_gemm_ks: the forward products' B operand on the two-workgroups-per-CU
        kernel), re-made from the bf16 mirror whenever that one has moved: one launch over all 4 x (Ne + Nd) weights, ~0.35 GB of HBM traffic
        for ViT-B.  Lives on the FlatParams like the bf16 mirror (every engine of a model shares it)."""
        f = self.flat
        if self.T != BF16 or self.fp8 or not self.use_ks:
            return
        if getattr(f, "w_ks", None) is None:
            names = [n for n in self._fp8_names() if f.slots[n][2][1] % 64 == 0]
            f.w_ks = torch.zeros(f.total, device=self.device, dtype=torch.bfloat16)
            f.ks_names = set(names)
            f.ks_desc = torch.tensor([[f.slots[n][0], f.slots[n][2][0], f.slots[n][2][1]] for n in names], dtype=torch.long, device=self.device).reshape(-1, 3)
            f.ks_stamp = None
        stamp = (f.lp_stamp, f.raw_writes)
        if stamp != f.ks_stamp and f.ks_desc.shape[0]:
            self._opt_gate()
            # on the auxiliary stream, under the stem (crop, masking, patch embedding: ~0.2 ms before the first block needs a weight): the
            # 90-us launch is off the main chain; _ks_wait() orders the first consumer behind it
            cur = torch.cuda.current_stream()
            # (the event lives on the FlatParams like the stamp: every engine of the model — training, stand-alone halves, one per dtype —
            # must order its first consumer behind a refresh whichever engine launched it)
            if self.aux is not None and ops._timer is None:
                self.aux.wait_stream(cur)   # the optimizer's writes to the bf16 mirror
                ops.weights_kslab(f.ks_desc, f.w_lp, f.w_ks, st=self.aux.cuda_stream)
                f.ks_event = getattr(f, "ks_event", None) or torch.cuda.Event()
                f.ks_event.record(self.aux)
                f.ks_waited = set()      # streams that have been ordered behind this refresh
            else:
                ops.weights_kslab(f.ks_desc, f.w_lp, f.w_ks)
            f.ks_stamp = stamp

    def _opt_gate(self, names=None, so=None):
        """FusedAdamW(overlap=True) left its step on its own stream, one event per launch (optim.py): order stream `so` (default: the current one)
        behind the launch that steps `names` (None: behind all of them).  The launches are in parameter order on one stream, so a later one covers
        the earlier ones; a name the step did not touch (frozen) needs no launch of its own but rides on the first."""
        pend = getattr(self.flat, "opt_pending", None)
        if pend is None:
            return
        ev, last = pend["events"], len(pend["events"]) - 1
        k = last if names is None else max(pend["chunk_of"].get(n, 0) for n in names)
        so = so if so is not None else torch.cuda.current_stream()
        if pend["waited"].get(so.cuda_stream, -1) < k:
            so.wait_event(ev[k])
            pend["waited"][so.cuda_stream] = k

    def _ks_wait(self):
        f = self.flat
        waited = getattr(f, "ks_waited", None)
        if waited is not None:
            cur = torch.cuda.current_stream()
            if cur.cuda_stream not in waited:
                cur.wait_event(f.ks_event)
                waited.add(cur.cuda_stream)

    def _w_pe(self):
        if self.T == BF16 and self.Pp != self.cfg["P"]:
            return self.w_pe_pad
        return self.W("patch_embed.proj.weight")

    def _w_pred(self):
        if self.T == BF16 and self.Pp != self.cfg["P"]:
            return self.w_pred_pad
        return self.W("decoder_pred.weight")

    # ------------------------------------------------------------------ fp8 path
    def _fp8_names(self):
        c = self.cfg
        return [f"{stk}.{i}.{w}.weight" for stk, n in (("encoder", c["Ne"]), ("decoder", c["Nd"])) for i in range(n)
                for w in ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2")]

    def _refresh_fp8(self):
        """fp8 mirrors of the block weights, re-quantised from the fp32 masters every forward (per-tensor current scaling): W8 [out][in]
        for the forward products, W8T [in][out] for dX — so that every fp8 GEMM reads two K-contiguous operands."""
        f = self.flat
        if getattr(f, "w8", None) is None:
            f.w8 = torch.zeros(f.total, device=self.device, dtype=torch.uint8)
            f.w8t = torch.zeros(f.total, device=self.device, dtype=torch.uint8)
            f.w8_idx = {n: k for k, n in enumerate(self._fp8_names())}
            # [0]: the weights' maxima as last measured (what the mirrors' NEXT rewrite scales with), [1]: the slot FusedAdamW folds the new maxima into
            f.w8_amax = [torch.zeros(len(f.w8_idx), ops.FP8_SLOTS, device=self.device) for _ in range(2)]
            f.w8_dq = torch.ones(len(f.w8_idx), device=self.device)
            f.w8_desc = torch.tensor([[f.slots[n][0], f.slots[n][2][0], f.slots[n][2][1]] for n in f.w8_idx], dtype=torch.long, device=self.device)
            f.w8_stamp = None
        f.fp8_active = True    # (FusedAdamW: step the block weights through csmae_adamw_fp8, which re-writes both mirrors with delayed scaling)
        stamp = (f.version_stamp(), f.raw_writes)
        if f.w8_stamp == stamp and not debug_opt("fp8_requant"):
            return             # the mirrors were written by the optimizer step itself: nothing has touched the masters since
        self._opt_gate()
        f.w8_amax[0].zero_()
        ops.fp8_weights(f.w8_desc, f.p, f.w8, f.w8t, f.w8_amax[0], f.w8_dq)   # (one batched amax + two batched quantise launches instead of four per weight)
        f.w8_stamp = stamp

    def _fp8_begin(self):
        """Start of a step in fp8 mode: weight mirrors, site counter, amax pools.  Activations / gradients are scaled with the amax the
        same GEMM site saw in the previous step (one pass per tensor); the first step of a workspace measures its own (two passes)."""
        if self.fp8:
            ws = self.ws
            self._refresh_fp8()
            if ws.fp8_complete:            # the last pass through this workspace was a whole forward + backward
                ws.fp8_amax.reverse()      # what it recorded becomes "previous"
                ws.fp8_hist = True
            else:                          # first step, or the last pass was a different one (eval / stand-alone halves): start over
                ws.fp8_hist = False
            ws.fp8_complete = False
            ws.fp8_amax[0].zero_()
            self._fp8_site = 0
            self._fp8_blk, self._fp8_kept = {}, {}

    def _fp8_lean(self, Dm, T, H):
        """fp8 mode in its steady state: every consumer of y1 / y2 (the LayerNorm outputs), h (fc1's GELU output), dpre (fc2's dX x gelu') and dqkv reads the
        fp8 copy its producer emits — the next product, the weight-gradient products (csmae_gemm_dw_group_fp8), the bias gradients (column sums of the fp8
        bytes) — so the bf16 tensors are not written at all: 13 of a block's token x width matrices per step (ViT-H/14 at 256 per GPU: 49 GB of stores).
        Exactly the conditions under which _block_bwd takes the fp8 weight-gradient path (it raises if it finds otherwise)."""
        ws = self.ws
        return bool(self.fp8 and self.fp8_dw and ws.fp8_hist and self._fp8_fuse and self._fp8_fuse_lnb and self._fp8_fuse_attn and self._dw_mode == "half"
                    and self.res_dtype == torch.bfloat16 and Dm >= 256 and Dm % 16 == 0 and ops.attn_resident(ops.BF16, T, Dm // H) and not debug_opt("fp8_keep_bf16"))

    def _fp8_alloc(self):
        """Next GEMM-site index of the step (None outside fp8 mode).  The order of the calls is the same every step: that is what ties
        a site to the amax it recorded one step earlier."""
        if not self.fp8:
            return None
        k = self._fp8_site
        self._fp8_site += 1
        return k

    def _emit(self, site, buf, M, N, fmt):
        """Arguments that make a producer kernel write its [M, N] output as fp8 bytes into `buf` for GEMM site `site` (None when that is
        not possible: no fp8 mode, first step of a workspace — no previous amax —, fusion switched off)."""
        ws = self.ws
        if site is None or not self.fp8 or not ws.fp8_hist or not self._fp8_fuse:
            return None
        return (buf[: M * N].view(M, N), fmt, ws.fp8_amax[1][site], ws.fp8_amax[0][site], ws.fp8_dq[site:site + 1])

    def _emit_to(self, site, dst, fmt):
        """_emit with the destination given as a [M, N] uint8 view (a kept per-layer copy / a rotating twin) instead of a staging buffer."""
        ws = self.ws
        if site is None or not self.fp8 or not ws.fp8_hist or not self._fp8_fuse:
            return None
        return (dst, fmt, ws.fp8_amax[1][site], ws.fp8_amax[0][site], ws.fp8_dq[site:site + 1])

    def _mm(self, a, name, out, *, trans_b=False, bias=None, epilogue=EPI_NONE, aux=None, resid=None, st=None, lane=0, site=None, emit_site=None, a8=None, emit_dst=None,
            skip_out=False):
        """out = a W^T (forward) / a W (trans_b: dX) for a block weight `name`, through the bf16 / fp32 GEMM or, in fp8 mode, through
        quantise(a) + the fp8 GEMM.  `lane` picks the fp8 staging buffers (the two forward streams quantise concurrently).
        `emit_site`: this product's output is the A operand of GEMM site `emit_site` — with delayed scaling the epilogue writes its fp8
        copy (ws.q_b) itself; `site`: this product's A operand belongs to that pre-allocated site; `a8`: ... and its producer has
        already left it there as fp8 bytes (the `emit` of _emit())."""
        if not self.fp8:
            f = self.flat
            if not trans_b and self.T == BF16 and self.use_ks and name in f.ks_names:   # forward product: the weight's K-slab mirror (two workgroups per CU)
                o, cnt, _ = f.slots[name]
                return ops.gemm_ks(a, f.w_ks[o:o + cnt], self.W(name), out, bias=bias, epilogue=epilogue, aux=aux, resid=resid, st=st)
            return ops.gemm(a, self.W(name), out, trans_b=trans_b, bias=bias, epilogue=epilogue, aux=aux, resid=resid, st=st)
        f, ws = self.flat, self.ws
        M, K = a.shape
        o, cnt, shape = f.slots[name]
        b8 = f.w8t[o:o + cnt].view(shape[1], shape[0]) if trans_b else f.w8[o:o + cnt].view(shape)
        fmt = ops.FP8_E5M2 if trans_b else ops.FP8_E4M3
        prev, cur = ws.fp8_amax[1], ws.fp8_amax[0]
        if a8 is not None:
            k = site                                               # emitted by the producing kernel
        else:
            k = site if site is not None else self._fp8_alloc()
            a8 = ws.a8[lane][: M * K].view(M, K)
            if ws.fp8_hist:
                ops.fp8_quantize(a, a8, prev[k], ws.fp8_dq[k:k + 1], fmt=fmt, amax_next=cur[k], st=st)
            else:
                ops.fp8_quantize(a, a8, cur[k], ws.fp8_dq[k:k + 1], fmt=fmt, st=st)
        # (a forward product feeds a forward product, a dX product a dX product; emit_dst: the copy's home when it is kept for the weight gradients)
        emit = self._emit_to(emit_site, emit_dst, fmt) if emit_dst is not None else self._emit(emit_site, ws.q_b[lane], out.shape[0], out.shape[1], fmt)
        wi = f.w8_idx[name]
        return ops.gemm_fp8(a8, b8, out, ws.fp8_dq[k:k + 1], f.w8_dq[wi:wi + 1], a_fmt=fmt, bias=bias, epilogue=epilogue, aux=aux, resid=resid,
                            emit=emit, skip_out=skip_out and emit is not None, st=st)

    @staticmethod
    def _splitk(m_out, n_out, k_red, tile, ktile):
        tiles = math.ceil(m_out / tile) * math.ceil(n_out / tile)
        kt = math.ceil(k_red / ktile)
        want = max(1, 768 // tiles)
        return max(1, min(want, kt // 4 if kt >= 8 else 1))

    def _dw(self, dy, x, name):
        """dW[name] += dy^T x ; db[name] += colsum(dy)   (dy [tokens, >=out], x [tokens, >=in]; extra pad columns ignored)."""
        self._dw_group([(dy, x, name)])

    def _dw_group(self, items, slots=None, ready=None, items8=None):
        """Weight gradients of several Linear layers over the same tokens, [(dy, x, name)], in one launch (csmae_gemm_dw_group: the
        products share the chip, K slices are folded inside the kernel, the result goes straight into the gradient buffer).

        Weight gradients are leaves of the backward graph, so they run on a second HIP stream: their workgroups fill the CUs that the
        main chain's tails, small GEMMs, LayerNorm and attention kernels leave idle."""
        if items8 is not None:   # the same products from the fp8 copies of their operands: [(dy8, dq_y, x8, dq_x, name)] (csmae_gemm_dw_group_fp8)
            key = tuple((n, dy.data_ptr(), x.data_ptr(), qy.data_ptr(), qx.data_ptr()) for dy, qy, x, qx, n in items8)
            grp = self._dw_cache.get(key)
            if grp is None:
                prods = []
                for dy, qy, x, qx, name in items8:
                    gw = self.flat.G(name + ".weight")
                    gw2 = gw.view(gw.shape[0], -1)
                    prods.append((dy[:, : gw2.shape[0]], qy, x[:, : gw2.shape[1]], qx, gw2, self.flat.G(name + ".bias")))
                grp = self._dw_cache[key] = ops.DwGroup8(prods, self.ws.dw_ws)
        else:
            key = tuple((n, dy.data_ptr(), x.data_ptr()) for dy, x, n in items)
            grp = self._dw_cache.get(key)
        if grp is None:
            prods = []
            for dy, x, name in items:
                gw = self.flat.G(name + ".weight")
                gw2 = gw.view(gw.shape[0], -1)
                prods.append((dy[:, : gw2.shape[0]], x[:, : gw2.shape[1]], gw2, self.flat.G(name + ".bias")))
            grp = self._dw_cache[key] = ops.DwGroup(prods, self.ws.dw_ws)
        if ops._timer is not None or os.environ.get("CSMAE_DW_MAIN"):  # per-kernel HIP-event timing (bench.py) measures on the main stream:
            grp.launch(256, st=self.st)   # nothing runs beside the launch there, so it gets the whole chip like the other layouts' kernels
            return                        # (the 160-workgroup setting is a co-scheduling choice of the overlapped step, not a kernel property)
        side = self.side
        if ready is not None:   # the kernel that made the launch's last operand carries the event itself (ops.launch_done): no marker on the main stream
            ev = ready
        else:
            ev = self._event()
            ev.record(self.main)
        side.wait_event(ev)
        grp.launch(slots or self._dw_slots, st=side.cuda_stream)
        done = self._event()
        done.record(side)
        self._side_seq += 1
        for dy, _, _ in items:
            self._side_reads[dy.data_ptr()] = (self._side_seq, done)

    def _guard_write(self, *bufs):
        """Main stream is about to overwrite `bufs`: wait for the weight-gradient launches that still read them (if any).  The launches
        of the side stream finish in order, so ONE wait — for the youngest of them — covers all (every event wait that is enqueued
        before its event has fired costs the main queue a barrier packet, ~5-8 us of idle chip: DESIGN §5)."""
        last = None
        for b in bufs:
            e = self._side_reads.pop(b.data_ptr(), None)
            if e is not None and (last is None or e[0] > last[0]):
                last = e
        if last is not None:
            if self._side_waited < last[0]:
                self.main.wait_event(last[1])
                self._side_waited = last[0]

    def _event(self):
        if self._ev_i == len(self._events):
            e = torch.cuda.Event()
            e.record()   # (torch creates the HIP event at its first record: ops.launch_done needs the handle)
            self._events.append(e)
        self._ev_i += 1
        return self._events[self._ev_i - 1]

    def _new_side(self):
        """The weight-gradient stream.  CSMAE_DEBUG=dw_cus=n confines it to n compute units of every XCD (ops.cu_masked_stream) instead of letting its
        160-workgroup launches time-slice whole CUs with the main chain; the forward's second-view stream is then a stream of its own."""
        n = int(debug_opt("dw_cus", "0"))
        if n <= 0:
            return torch.cuda.Stream()
        self._side_masked = True
        return ops.cu_masked_stream(0, 8 * n)

    def _join_side(self):
        self.main.wait_stream(self.side)
        self._side_reads.clear()
        self._side_waited = self._side_seq

    # ------------------------------------------------------------------ transformer block
    def _block_fwd(self, S, i, pre, M, Dm, H, B2, T, b0=0, nb=None, st=None):
        """Samples [b0, b0 + nb) of the batch (default: all of it) on stream `st` (default: the engine's main stream)."""
        P = self.flat.P
        st = self.st if st is None else st
        nb = B2 if nb is None else nb
        r = slice(b0 * T, (b0 + nb) * T)
        x_in, x_mid, x_out = S["x"][i][r], S["xm"][i][r], S["x"][i + 1][r]
        stt = [a[r] for a in S["st"][i]]
        lse = S["lse"][i][b0 * H * T: (b0 + nb) * H * T]
        qkv, o, h, pre_a = S["qkv"][i][r], S["o"][i][r], S["h"][i][r], S["pre"][i][r]
        y1, y2 = S["y1"][i][r], S["y2"][i][r]
        ws_q_a = self.ws.q_a if self.fp8 else None
        ln = int(b0 > 0)
        Mr = y1.shape[0]
        k1 = ko = k2 = kh = None
        if getattr(self.flat, "opt_pending", None) is not None:   # an overlapped optimizer step: this block behind the launch that steps its weights
            self._opt_gate((pre + "attn.qkv.weight", pre + "mlp.fc2.weight"), None if st is None else (self._so.get(st) or torch.cuda.ExternalStream(st)))
        if self.fp8:
            # The block's four forward GEMM sites (A operands y1, o, y2, h) are shared by the two views' calls: one scale per tensor over the whole batch
            # — the weight gradients contract over both views' tokens at once.  (The step without amax history quantises with the tensor's own maximum,
            # which two concurrent calls cannot share: that step runs un-split, see _forward.)
            sites = self._fp8_blk.get((id(S), i))
            if sites is None:
                sites = self._fp8_blk[(id(S), i)] = tuple(self._fp8_alloc() for _ in range(4))
            k1, ko, k2, kh = sites
        keep8 = self.fp8 and self.fp8_dw   # the fp8 copies go to their per-layer homes (the dW products read them in the backward pass) instead of a staging buffer
        d8 = (lambda name, stage, n: S[name][i][r] if keep8 else stage[: Mr * n].view(Mr, n))
        e1 = self._emit_to(k1, d8("y1_8", ws_q_a[ln], Dm), 0) if self.fp8 else None
        lean = self._fp8_lean(Dm, T, H)
        fuse = self._ln_fused_fwd(Mr, Dm)
        if not (fuse and (self._ln_fuse_mask & 2) and i > 0):   # (fused: the previous block's fc2 epilogue has already left norm1(x_in) in y1 and its statistics)
            ops.layernorm_fwd(x_in, P(pre + "norm1.weight"), P(pre + "norm1.bias"), y1, stt[0], stt[1], emit=e1, skip_out=lean and e1 is not None, st=st)
        self._mm(y1, pre + "attn.qkv.weight", qkv, bias=P(pre + "attn.qkv.bias"), st=st, lane=ln, site=k1, a8=e1[0] if e1 else None)
        # (fp8 mode) attention leaves its output as fp8 bytes for attn.proj (q_a: y1 has been consumed by the qkv GEMM, y2 comes after proj)
        eo = self._emit_to(ko, d8("o_8", ws_q_a[ln], Dm), 0) if (self.fp8 and self._fp8_fuse_attn and ops.attn_resident(ops.BF16, T, Dm // H)) else None
        ops.attn_fwd(qkv, o, lse, nb, T, H, Dm // H, emit=eo, st=st)
        if fuse and (self._ln_fuse_mask & 1):   # x_mid = x_in + proj(o) and y2 = norm2(x_mid) in one kernel
            ops.gemm_ln_fwd(o, self._ks(pre + "attn.proj.weight"), P(pre + "attn.proj.bias"), x_in, x_mid, P(pre + "norm2.weight"), P(pre + "norm2.bias"),
                            y2, stt[2], stt[3], st=st)
            e2 = None
        else:
            self._mm(o, pre + "attn.proj.weight", x_mid, bias=P(pre + "attn.proj.bias"), epilogue=EPI_RESID, resid=x_in, st=st, lane=int(b0 > 0), site=ko,
                     a8=eo[0] if eo else None)
            e2 = self._emit_to(k2, d8("y2_8", ws_q_a[ln], Dm), 0) if self.fp8 else None
            ops.layernorm_fwd(x_mid, P(pre + "norm2.weight"), P(pre + "norm2.bias"), y2, stt[2], stt[3], emit=e2, skip_out=lean and e2 is not None, st=st)
        # (fp8 mode) kh = the site of fc2's A operand: h leaves the fc1 epilogue as bf16 AND as fp8 bytes
        h8 = d8("h_8", self.ws.q_b[ln], 4 * Dm) if self.fp8 else None
        eh = self._emit_to(kh, h8, 0) if self.fp8 else None
        self._mm(y2, pre + "mlp.fc1.weight", h, bias=P(pre + "mlp.fc1.bias"), epilogue=EPI_GELU, aux=pre_a, st=st, lane=ln, site=k2, a8=e2[0] if e2 else None, emit_site=kh,
                 emit_dst=h8, skip_out=lean)
        if keep8:   # which of the block's kept copies this pass really wrote (none without amax history; o only from the LDS-resident attention kernels)
            self._fp8_kept[(id(S), i)] = dict(y1=e1 is not None, o=eo is not None, y2=e2 is not None, h=eh is not None, lean=lean)
        if fuse and (self._ln_fuse_mask & 2) and i + 1 < S["xm"].shape[0]:   # x_out = x_mid + fc2(h) and the NEXT block's y1 = norm1(x_out) in one kernel
            nxt = pre[: pre.rstrip(".").rfind(".") + 1] + f"{i + 1}."
            sn = [a[r] for a in S["st"][i + 1][:2]]
            ops.gemm_ln_fwd(h, self._ks(pre + "mlp.fc2.weight"), P(pre + "mlp.fc2.bias"), x_mid, x_out, P(nxt + "norm1.weight"), P(nxt + "norm1.bias"),
                            S["y1"][i + 1][r], sn[0], sn[1], st=st)
        else:
            self._mm(h, pre + "mlp.fc2.weight", x_out, bias=P(pre + "mlp.fc2.bias"), epilogue=EPI_RESID, resid=x_mid, st=st, lane=ln, site=kh, a8=eh[0] if eh else None)

    def _ks(self, name):
        o, cnt, _ = self.flat.slots[name]
        return self.flat.w_ks[o:o + cnt]

    def _ln_fused_fwd(self, M, Dm):
        return self.ln_fuse_fwd and self.use_ks and ops.gemm_ln_supported(M, Dm, Dm)

    def _ln_fused_bwd(self, M, Dm):
        return self.ln_fuse_bwd and ops.gemm_ln_supported(M, Dm, 4 * Dm)

    def _block_bwd(self, S, i, pre, M, Dm, H, B2, T, dres, lps, k, part, lps8=None):
        """`lps` = the rotating low-precision copies of the residual gradient, lps[k] is current on entry; returns the index that is
        current on exit.  With a bf16 residual stream they ARE the residual gradient (`dres` is None); with the fp32 stream `dres` is
        updated in place.  `part` = partial-row slices of this block's two LayerNorms (norm1, norm2): their dgamma / dbeta are folded
        later (_ln_flush).

        Weight gradients: the block's four products (fc2, fc1, proj, qkv) leave as ONE grouped launch once dqkv exists ("block" mode) —
        together their tiles fill the launch's workgroups with one K slice (ViT-B encoder: 108 tiles) or three (decoder: 48 tiles)
        instead of 2-10, so the fp32 slab round trip and the fold kernel all but disappear.  The launch reads four gradient tensors
        that the main chain would overwrite within two kernels; the block's outgoing residual gradient therefore goes to a third
        buffer and the next block uses two more (five in rotation), and nothing ever waits for the weight-gradient stream unless it
        falls two blocks behind."""
        P, st, ws = self.flat.P, self.st, self.ws
        stt = S["st"][i]
        lse = S["lse"][i][: B2 * H * T]
        self._tog ^= 1
        dpre = ws.t4[self._tog][: M * 4 * Dm].view(M, 4 * Dm)
        dqkv = ws.t3[self._tog][: M * 3 * Dm].view(M, 3 * Dm)
        t1 = ws.t1[: M * Dm].view(M, Dm)
        mode = self._dw_mode
        y1, y2 = S["y1"][i], S["y2"][i]
        if mode in ("block_enc", "block_dec"):   # A/B aid: the one-launch form for one of the two stacks only
            mode = "block" if (mode == "block_enc") == (S is ws.enc) else "half"
        if mode == "block":   # (wide models: the block's tiles exceed the launch's workgroups — 192 at ViT-L, 300 at ViT-H — and two launches interleave better with the main chain)
            tiles = sum(-(-a // 256) * -(-b // 256) for a, b in ((Dm, 4 * Dm), (4 * Dm, Dm), (Dm, Dm), (3 * Dm, Dm)))
            if tiles > self._dw_slots or Dm < 256:
                mode = "half"
        n = len(lps)
        cur, nxt = lps[k], lps[(k + 1) % n]
        rot = mode == "block" or (mode == "half" and self._dw_rot)
        out = lps[(k + 2) % n] if rot else cur
        if rot:   # everything this block writes that an earlier block's launch may still read: one wait (normally for a launch two blocks back)
            self._guard_write(dpre, nxt, dqkv, out)
        if mode == "none":
            self._dw(cur, S["h"][i], pre + "mlp.fc2")
        self._guard_write(dpre)
        # (fp8 mode) A operands that their producer already left as fp8 bytes: `cur` (previous LayerNorm backward), dpre (the ×gelu′ epilogue)
        kc, c8 = self._fp8_cur if self._fp8_cur is not None else (None, None)
        self._fp8_cur = None
        kd = self._fp8_alloc()
        # fp8 weight gradients: the fp8 copies of the gradient tensors live in twins of the rotating bf16 buffers (same indices, same lifetimes: the guards
        # on the bf16 buffers cover them) instead of two staging buffers that the next kernel of the chain overwrites
        tw = self.fp8 and self.fp8_dw and lps8 is not None
        kept = self._fp8_kept.get((id(S), i), {}) if tw else {}
        lean = bool(kept.get("lean"))   # the forward pass left no bf16 y1 / y2 / h (_fp8_lean): this pass writes no bf16 dpre / dqkv either
        dpre8 = ws.t4_8[self._tog][: M * 4 * Dm].view(M, 4 * Dm) if tw else None
        ed = (self._emit_to(kd, dpre8, 1) if tw else self._emit(kd, ws.q_b[0], M, 4 * Dm, 1)) if self.fp8 else None
        if tw and kc is None and ws.fp8_hist and self._fp8_fuse:
            # the stack's first block: its incoming residual gradient comes from a kernel that emits no fp8 copy (latent_grad_finish).  The fc2-dX product
            # quantises it anyway (one pass, delayed scaling): into the buffer's fp8 twin instead of a staging buffer, so that the weight gradients of
            # fc2 can read it too.  (Same position in the site order as the allocation inside _mm that it replaces.)
            kc = self._fp8_alloc()
            c8 = lps8[k]
            ops.fp8_quantize(cur, c8, ws.fp8_amax[1][kc], ws.fp8_dq[kc:kc + 1], fmt=ops.FP8_E5M2, amax_next=ws.fp8_amax[0][kc], st=st)
        # the kernels whose outputs a weight-gradient launch waits for carry that launch's event themselves (an event recorded behind them
        # is a marker packet: ~5 us of idle main stream each, two per block)
        carried = ops._timer is None and not os.environ.get("CSMAE_DW_MAIN")
        ev1 = self._event() if (carried and mode == "half") else None
        with (ops.launch_done(ev1, st) if ev1 is not None else contextlib.nullcontext()):
            self._mm(cur, pre + "mlp.fc2.weight", dpre, trans_b=True, epilogue=EPI_DGELU, aux=S["pre"][i], st=st, site=kc, a8=c8, emit_site=kd, emit_dst=dpre8,
                     skip_out=lean)
        slots = self._dw_slots_ed[0 if S is ws.enc else 1] if self._dw_slots_ed else None
        fs = self._fp8_blk.get((id(S), i)) if tw else None   # the block's forward sites (y1, o, y2, h)
        dq = ws.fp8_dq if self.fp8 else None
        ok8 = tw and fs is not None and Dm >= 256 and Dm % 16 == 0
        if mode == "half":
            g8 = None
            if ok8 and c8 is not None and ed is not None and kept.get("h") and kept.get("y2"):
                g8 = [(c8, dq[kc:kc + 1], S["h_8"][i], dq[fs[3]:fs[3] + 1], pre + "mlp.fc2"), (ed[0], dq[kd:kd + 1], S["y2_8"][i], dq[fs[2]:fs[2] + 1], pre + "mlp.fc1")]
            if lean and g8 is None:
                raise RuntimeError("csmae_hip fp8: the forward pass kept no bf16 activations (Engine._fp8_lean) but the weight gradients of fc2 / fc1 cannot take the fp8 path")
            self._dw_group([(cur, S["h"][i], pre + "mlp.fc2"), (dpre, y2, pre + "mlp.fc1")], slots, ready=ev1, items8=g8)
        elif mode == "none":
            self._dw(dpre, y2, pre + "mlp.fc1")
        fuse = dres is None and self._ln_fused_bwd(M, Dm)
        if fuse:   # fc1's dX product, norm2's backward and the residual-gradient add in one kernel: the product never reaches HBM
            self._guard_write(nxt)
            ops.gemm_ln_bwd(dpre, self.W(pre + "mlp.fc1.weight"), S["xm"][i], stt[2], stt[3], P(pre + "norm2.weight"), cur, nxt, partial_ws=part[1], st=st)
            kn, en = None, None
        else:
            self._mm(dpre, pre + "mlp.fc1.weight", t1, trans_b=True, st=st, site=kd, a8=ed[0] if ed else None)
            self._guard_write(nxt)
            kn = self._fp8_alloc()
            en = None
            if self.fp8 and self._fp8_fuse_lnb:
                en = self._emit_to(kn, lps8[(k + 1) % n], 1) if tw else self._emit(kn, ws.q_a[0], M, Dm, 1)
        if fuse:
            pass
        elif dres is None:
            ops.layernorm_bwd(t1, S["xm"][i], stt[2], stt[3], P(pre + "norm2.weight"), nxt, None, None, dres_in=cur, partial_ws=part[1], emit=en, st=st)
        else:
            ops.layernorm_bwd(t1, S["xm"][i], stt[2], stt[3], P(pre + "norm2.weight"), dres, None, None, dres_in=dres, dx_lp=nxt, partial_ws=part[1], st=st)
            en = None
        if mode == "none":
            self._dw(nxt, S["o"][i], pre + "attn.proj")
        self._mm(nxt, pre + "attn.proj.weight", t1, trans_b=True, st=st, site=kn, a8=en[0] if en else None)
        self._guard_write(dqkv)
        ev2 = self._event() if (carried and mode in ("half", "block")) else None
        # (fp8 mode) ... and dqkv for attn.qkv's backward (q_b[0]: dpre's copy has been consumed by fc1's backward GEMM)
        kq = self._fp8_alloc()
        eq = None
        if self.fp8 and self._fp8_fuse_attn and ops.attn_resident(ops.BF16, T, Dm // H):
            eq = self._emit_to(kq, ws.t3_8[self._tog][: M * 3 * Dm].view(M, 3 * Dm), 1) if tw else self._emit(kq, ws.q_b[0], M, 3 * Dm, 1)
        with (ops.launch_done(ev2, st) if ev2 is not None else contextlib.nullcontext()):
            ops.attn_bwd(S["qkv"][i], S["o"][i], t1, lse, dqkv, B2, T, H, Dm // H, emit=eq, skip_out=lean and eq is not None, st=st)
        if mode == "block":
            self._dw_group([(cur, S["h"][i], pre + "mlp.fc2"), (dpre, y2, pre + "mlp.fc1"),
                            (nxt, S["o"][i], pre + "attn.proj"), (dqkv, y1, pre + "attn.qkv")], slots, ready=ev2)
        elif mode == "half":
            g8 = None
            if ok8 and en is not None and eq is not None and kept.get("o") and kept.get("y1"):
                g8 = [(en[0], dq[kn:kn + 1], S["o_8"][i], dq[fs[1]:fs[1] + 1], pre + "attn.proj"), (eq[0], dq[kq:kq + 1], S["y1_8"][i], dq[fs[0]:fs[0] + 1], pre + "attn.qkv")]
            if lean and g8 is None:
                raise RuntimeError("csmae_hip fp8: the forward pass kept no bf16 activations (Engine._fp8_lean) but the weight gradients of attn.proj / attn.qkv cannot take the fp8 path")
            self._dw_group([(nxt, S["o"][i], pre + "attn.proj"), (dqkv, y1, pre + "attn.qkv")], slots, ready=ev2, items8=g8)
        else:
            self._dw(dqkv, y1, pre + "attn.qkv")
        if fuse:   # qkv's dX product + norm1's backward + the residual-gradient add
            self._guard_write(out)
            ops.gemm_ln_bwd(dqkv, self.W(pre + "attn.qkv.weight"), S["x"][i], stt[0], stt[1], P(pre + "norm1.weight"), nxt, out, partial_ws=part[0], st=st)
            return (k + 2) % n if rot else k
        self._mm(dqkv, pre + "attn.qkv.weight", t1, trans_b=True, st=st, site=kq, a8=eq[0] if eq else None)
        self._guard_write(out)
        kx = self._fp8_alloc() if i > 0 else None     # the next block's fc2-backward reads `out`
        ex = None
        if self.fp8 and dres is None and self._fp8_fuse_lnb:
            ex = self._emit_to(kx, lps8[(k + 2) % n if rot else k], 1) if tw else self._emit(kx, ws.q_a[1], M, Dm, 1)
        if dres is None:
            ops.layernorm_bwd(t1, S["x"][i], stt[0], stt[1], P(pre + "norm1.weight"), out, None, None, dres_in=nxt, partial_ws=part[0], emit=ex, st=st)
        else:
            ops.layernorm_bwd(t1, S["x"][i], stt[0], stt[1], P(pre + "norm1.weight"), dres, None, None, dres_in=dres, dx_lp=out, partial_ws=part[0], st=st)
        if kx is not None:
            self._fp8_cur = (kx, ex[0] if ex else None)
        return (k + 2) % n if rot else k

    def _ln_flush(self, part, goff, lo, hi, M, Dm, fused=False):
        """dgamma / dbeta of LayerNorms [lo, hi) of a stack (rows of `part` / `goff`): one deterministic launch.  `fused`: their partial rows were
        left by the GEMM epilogues (csmae_gemm_ln_bwd: one row per 128-row tile)."""
        if hi > lo:
            if fused:
                ops.ln_param_reduce_rows(hi - lo, -(-M // 128), Dm, part[lo:hi], goff[lo:hi], self.flat.g, st=self.st)
            else:
                ops.ln_param_reduce(hi - lo, M, Dm, part[lo:hi], goff[lo:hi], self.flat.g, st=self.st)

    # ------------------------------------------------------------------ forward
    def forward(self, imgs: torch.Tensor, mask_ratio: float, noise: torch.Tensor, box_host: Optional[torch.Tensor], training: bool,
                img1: Optional[torch.Tensor] = None, expect_backward: bool = False):
        with trace.range_("csmae.forward"):
            return self._forward(imgs, mask_ratio, noise, box_host, training, img1, expect_backward)

    def _forward(self, imgs: torch.Tensor, mask_ratio: float, noise: torch.Tensor, box_host: Optional[torch.Tensor], training: bool,
                 img1: Optional[torch.Tensor] = None, expect_backward: bool = False):
        """`img1`: the second view given explicitly (MAE_ViT_MsLd_PAIRED, MAE_ViT_MsLd.py:79-146) instead of the random resized crop of `imgs`."""
        c = self.cfg
        N = imgs.shape[0]
        keep = int(c["L"] * (1 - mask_ratio))
        if keep < 1:
            raise ValueError(f"mask_ratio={mask_ratio} keeps no patch (L={c['L']})")
        if self.ws is None or self.ws.N != N or self.ws.keep != keep:
            self.ws = None
            self._dw_cache.clear()
            self.ws = Workspace(self, N, keep)
        ws, P = self.ws, self.flat.P
        self.st = st = ops.stream()
        B2, Te, Td, L, D, Dd = ws.B2, ws.Te, ws.Td, c["L"], c["D"], c["Dd"]
        # an overlapped optimizer step still in flight (FusedAdamW(overlap=True)): the bf16 engine orders every layer behind the launch that steps its
        # weights (_opt_gate) and lets the rest run under the stem and the first blocks; every other engine starts behind all of it
        fine = self.T == BF16 and not debug_opt("opt_gate_all")
        if not fine:
            self._opt_gate()
        self._so = {st: torch.cuda.current_stream()}
        self._refresh_lp(lazy_pads=fine)
        self._fp8_begin()
        pads = fine and self.Pp != c["P"]   # (patch 14: the padded copies of the stem's and the prediction head's weights are re-made below, each behind its own launch of the step)
        pred_pad_ev = None
        img0 = imgs
        two = self.views == 2 and ops._timer is None and not os.environ.get("CSMAE_FWD_ONE_STREAM")
        if self.fp8 and not ws.fp8_hist:
            two = False   # (fp8, no amax history yet: every tensor is scaled with its own maximum over BOTH views — one call per block, see _block_fwd)
        nch = 2   # sample chunks in flight on their own streams: one per view (4 / 8 chunks and an uneven split were measured and lost)
        # (The stem per view on the view's stream — the original's patches not waiting for the crop kernel — measured neutral: 21.87 vs 21.88 ms,
        # the original's patch-embed product queues behind the crop kernel's 10 k workgroups for CUs anyway.  One stem on the main stream.)
        if self.views != 2:
            img1 = None
        elif img1 is None:
            ws.box.copy_(box_host, non_blocking=True)
            ops.crop_resize(img0, ws.imgs_crop, ws.box, st=st)
            img1 = ws.imgs_crop
        ws.noise.copy_(noise)
        ops.mask_sort(ws.noise, keep, ws.ids_restore, ws.mask, ws.ids_keep, st=st)
        ops.patch_gather(img0, img1, ws.ids_keep, ws.a_pe, N, c["C"], c["S"], c["p"], keep, st=st)
        self._opt_gate(("patch_embed.proj.weight", "cls_token"))
        if pads:
            self._refresh_pads(True, False)
        ops.gemm(ws.a_pe, self._w_pe(), ws.tok, bias=P("patch_embed.proj.bias"), st=st)
        ops.embed_assemble(ws.tok, P("encoder_pos_embed").view(L + 1, D), P("cls_token").view(D), ws.ids_keep, ws.enc["x"][0], B2, keep, st=st)
        self._ks_wait()   # the K-slab weight mirrors (made under the stem on the auxiliary stream) before the first block and before the view streams fork
        latent = ws.enc["x"][c["Ne"]]
        lat_heads = ws.lat32 if ws.lat32 is not None else latent   # what the loss heads read (fp32)
        main = torch.cuda.current_stream()
        if self.side is None:
            self.side = self._new_side()
        if self.aux is None:
            self.aux = torch.cuda.Stream()
        ce_done = None
        if pads:   # the prediction head's padded weight on the auxiliary stream, behind the launch that steps decoder_pred (and behind the last backward pass, which read it)
            self.aux.wait_stream(main)
            self._opt_gate(("decoder_pred.weight",), self.aux)
            with torch.cuda.stream(self.aux):
                self._refresh_pads(False, True)
            pred_pad_ev = torch.cuda.Event()
            pred_pad_ev.record(self.aux)

        kind, npx = c["loss"], c["norm_pix"]
        ssim = SSIM_KINDS.get(kind)
        if ssim is not None:  # MAE_ViT_Shared.py:165-267: (per-patch kind, pyramid levels, weight); the term joins `losses` after finalize
            kind = ssim[0]
        # Loss heads per view, each on its view's stream: the reconstruction term of a view needs only that view's prediction, and the
        # cross-decoder predictor (MAE_ViT_MsLdCeCd.py:57: gather -> Linear -> BatchNorm/ReLU -> Linear) only the crop's decoder output —
        # they run behind their trunk instead of behind the join of both (where one stream idles until the other arrives).  Only when a
        # chunk IS a view and the per-patch kind needs no whole-tensor statistics (bce's min / max, the ssim family).
        view_heads = two and ssim is None and kind in ("mse", "l2", "mae", "l1")

        emb_ready, heads_done = [], []
        crop_heads_aux = view_heads and self.has_pred and ops._timer is None and not debug_opt("crop_heads_main")

        def predictor_fwd(st):
            bn = "predictor.1."
            mod = self.module.predictor[1]
            self._opt_gate(("predictor.0.weight", "predictor.3.weight"), self._so.get(st))
            ops.rows_gather(ws.emb32, ws.pin, L, Td, N * Td + 1, st=st)
            ops.gemm(ws.pin, self.W("predictor.0.weight"), ws.u, bias=P("predictor.0.bias"), st=st)
            ops.bnrelu_fwd(ws.u, P(bn + "weight"), P(bn + "bias"), ws.r, ws.bn_st[0], ws.bn_st[1], N, L, mod.running_mean, mod.running_var,
                           mod.num_batches_tracked, eps=mod.eps, momentum=mod.momentum, training=training, st=st)
            ops.gemm(ws.r, self.W("predictor.3.weight"), ws.v, bias=P("predictor.3.bias"), st=st)

        def trunk(b0, nb, st, stream_obj, evs):
            """Encoder -> decoder -> prediction for samples [b0, b0 + nb), as a generator that yields after every block so that the
            trunks of several streams are enqueued round-robin (a stream whose kernels are enqueued only after another stream's
            whole trunk starts late whenever the host is not far ahead of the GPU).  Appends the event after the encoder (contrastive
            branch) to `evs`."""
            for i in range(c["Ne"]):
                self._block_fwd(ws.enc, i, f"encoder.{i}.", ws.Me, D, c["He"], B2, Te, b0, nb, st)
                yield
            re_, rd_ = slice(b0 * Te, (b0 + nb) * Te), slice(b0 * Td, (b0 + nb) * Td)
            if ws.lat32 is not None:        # bf16 residual stream: the latent is a GEMM operand as it is; the heads read an fp32 copy
                ops.cast_f32(latent[re_], ws.lat32[re_], st=st)
                lat_op = latent[re_]
            elif self.T == BF16:
                ops.cast_bf16(latent[re_], ws.lat_lp[re_], st=st)
                lat_op = ws.lat_lp[re_]
            else:
                lat_op = latent[re_]
            ev = None
            if self.has_ce and ops._timer is None:
                ev = torch.cuda.Event()
                ev.record(stream_obj)
            evs.append(ev)
            self._opt_gate(("decoder_embed.weight", "mask_token"), stream_obj)
            ops.gemm(lat_op, self.W("decoder_embed.weight"), ws.z[re_], bias=P("decoder_embed.bias"), st=st)
            ops.unshuffle_fwd(ws.z[re_], P("mask_token").view(Dd), P("decoder_pos_embed").view(L + 1, Dd), ws.ids_restore[b0:b0 + nb],
                              ws.dec["x"][0][rd_], nb, L, keep, st=st)
            yield
            for i in range(c["Nd"]):
                self._block_fwd(ws.dec, i, f"decoder.{i}.", ws.Md, Dd, c["Hd"], B2, Td, b0, nb, st)
                yield
            ops.layernorm_fwd(ws.dec["x"][c["Nd"]][rd_], P("decoder_norm.weight"), P("decoder_norm.bias"), ws.emb_lp[rd_], ws.dn_st[0][rd_],
                              ws.dn_st[1][rd_], y32=ws.emb32[rd_], st=st)
            hst = st
            if view_heads and b0 > 0 and self.has_pred:   # the crop's decoder output exists: the predictor (on the main stream, below) may start
                emb_ready.append(torch.cuda.Event())
                emb_ready[0].record(stream_obj)
                if crop_heads_aux:
                    # ... and the crop's own reconstruction head (decoder_pred product + per-patch loss, ~75 us, nothing depends on it before the
                    # loss sum) moves to the auxiliary stream: on the main stream it stood between the crop's decoder_norm and the predictor chain,
                    # the longest tail of the forward pass
                    self.aux.wait_event(emb_ready[0])
                    hst = self.aux.cuda_stream
            self._opt_gate(("decoder_pred.weight",), stream_obj if hst is st else self.aux)
            if pred_pad_ev is not None:
                (stream_obj if hst is st else self.aux).wait_event(pred_pad_ev)
            ops.gemm(ws.emb_lp[rd_], self._w_pred()[: c["P"]], ws.pred[rd_], bias=P("decoder_pred.bias"), st=hst)
            if view_heads:   # (a chunk is a view here: samples [0, N) = the original, [N, 2N) = the crop)
                ops.recon_loss_fwd(kind, npx, img0 if b0 == 0 else img1, None, ws.pred[rd_], None, ws.rowloss[b0 * L:(b0 + nb) * L], nb, nb,
                                   c["C"], c["S"], c["p"], mask=ws.mask[b0:b0 + nb], st=hst)
            if hst is not st:
                heads_done.append(torch.cuda.Event())
                heads_done[0].record(self.aux)

        if two:
            # The two views are independent until the losses: view 1 runs on the second stream.  Its kernels fill the CUs that view
            # 0's partial waves, attention and LayerNorm kernels leave idle (same effect as the weight-gradient stream in backward).
            while len(self._fwd_streams) < nch - 1:
                self._fwd_streams.append(self.side if (not self._fwd_streams and not getattr(self, "_side_masked", False)) else torch.cuda.Stream())
            per = B2 // nch
            cuts = [k * per for k in range(nch)] + [B2]
            for so in self._fwd_streams[: nch - 1]:
                so.wait_stream(main)         # (the stem; and the previous step's readers of the workspace)
            if self.has_ce:
                self.aux.wait_stream(main)   # (workspace reuse: the previous step's backward read E / zc on the main stream) — BEFORE the trunks are
                                             # enqueued: the contrastive branch then starts when both encoders are done, under the decoders' GEMMs
            evs = []
            # which chunk the main stream takes: with the per-view heads the CROP's (the last chunk) — the main stream starts first and tends to
            # finish first, and the predictor that follows the crop's decoder is the longest tail of the forward pass (-0.09 ms against the original's)
            order = list(range(nch))
            if view_heads and self.has_pred:
                order = [nch - 1] + list(range(nch - 1))
            lanes = [(st, main)] + [(so.cuda_stream, so) for so in self._fwd_streams[: nch - 1]]
            self._so.update(lanes)
            gens = [trunk(cuts[k], cuts[k + 1] - cuts[k], lanes[i][0], lanes[i][1], evs) for i, k in enumerate(order)]

            def contrastive():
                for ev in evs:
                    self.aux.wait_event(ev)
                ops.ntxent_fwd(lat_heads, ws.zc, ws.inv_norm, ws.E, ws.neg, ws.ce_rowloss, N, Te, keep, st=self.aux.cuda_stream)
                done = torch.cuda.Event()
                done.record(self.aux)
                return done
            while gens:
                gens = [g for g in gens if next(g, StopIteration) is not StopIteration]
                if self.has_ce and ce_done is None and len(evs) == nch:   # both encoders are enqueued: the contrastive branch goes out now
                    ce_done = contrastive()
            if self.has_ce and ce_done is None:
                ce_done = contrastive()
            if view_heads and self.has_pred:
                # the predictor behind the main stream's own trunk, as soon as the crop's decoder_norm output exists: the second stream
                # still has its decoder_pred product and reconstruction term to do, and neither stream waits for the other's tail
                main.wait_event(emb_ready[0])
                predictor_fwd(st)
            for so in self._fwd_streams[: nch - 1]:
                main.wait_stream(so)
        else:
            evs = []
            for _ in trunk(0, B2, st, main, evs):
                pass
            ev0 = evs[0]
            if self.has_ce:
                if ev0 is None:
                    ops.ntxent_fwd(lat_heads, ws.zc, ws.inv_norm, ws.E, ws.neg, ws.ce_rowloss, N, Te, keep, st=st)
                else:
                    self.aux.wait_stream(main)
                    ops.ntxent_fwd(lat_heads, ws.zc, ws.inv_norm, ws.E, ws.neg, ws.ce_rowloss, N, Te, keep, st=self.aux.cuda_stream)
                    ce_done = torch.cuda.Event()
                    ce_done.record(self.aux)
        mm = None
        if ssim is not None:
            ops.ssim_fwd(ssim[1], npx, img0, img1, ws.pred, ws.mask, ws.ssim_ws, ws.ssim_terms, B2, N, c["C"], c["S"], c["p"], st=st)
        if kind == "bce":
            ops.target_minmax(img0, img1, ws.mm_scratch, ws.minmax, B2, N, c["C"], c["S"], c["p"], npx, st=st)
            mm = ws.minmax
        if kind == "none":
            ws.rowloss.zero_()
        elif not view_heads:
            ops.recon_loss_fwd(kind, npx, img0, img1, ws.pred, mm, ws.rowloss, B2, N, c["C"], c["S"], c["p"], mask=ws.mask, st=st)
        kw, kw_spec = {}, False
        if self.has_pred:
            kcd = c["loss_cd"]
            if not view_heads:
                predictor_fwd(st)
            ops.pair_loss_fwd(kcd, N * L, Dd, ws.v, (N * L, 0, 0), ws.emb32, (L, Td, 1), ws.cd_partial, st=st)
            kw.update(cd_partial=ws.cd_partial, cd_scale=self._pair_scale(kcd, N * L, Dd))
            # The head's BACKWARD chain (pair loss -> Linear -> BatchNorm / ReLU -> Linear: ~340 us, the longest chain of the forward / backward junction)
            # starts here, on the auxiliary stream, with unit upstream gradient: every gradient is linear in that scalar, and _backward applies the real
            # one (csmae_spec_fixup: a no-op for 1.0).  Only for a training forward that autograd records (`expect_backward`: the model's autograd node).
            spec = (training and expect_backward and view_heads and ops._timer is None and self.T == BF16 and not os.environ.get("CSMAE_DW_MAIN")
                    and not debug_opt("no_spec_junction"))
            if spec:
                bn, aux, ast = "predictor.1.", self.aux, self.aux.cuda_stream
                ev = torch.cuda.Event()
                ev.record(main)
                aux.wait_event(ev)
                self._opt_gate(("predictor.0.weight", "predictor.3.weight"), aux)
                with torch.cuda.stream(aux):
                    ws.bn_tmp.zero_()
                ops.pair_loss_bwd(kcd, N * L, Dd, ws.v, (N * L, 0, 0), ws.emb32, (L, Td, 1), ws.one, self._pair_scale(kcd, N * L, Dd), da_lp=ws.dv, st=ast)
                ops.gemm(ws.dv, self.W("predictor.3.weight"), ws.dr, trans_b=True, st=ast)
                ops.bnrelu_bwd(ws.u, ws.dr, P(bn + "weight"), P(bn + "bias"), ws.bn_st[0], ws.bn_st[1], ws.dr, ws.bn_tmp[0], ws.bn_tmp[1], N, L, st=ast)
                ops.gemm(ws.dr, self.W("predictor.0.weight"), ws.dpin, trans_b=True, st=ast)
            kw_spec = spec
        if self.has_le:
            ke = c["loss_e"]
            ops.pair_loss_fwd(ke, N * Te, D, lat_heads, (N * Te, 0, N * Te), lat_heads, (N * Te, 0, 0), ws.e_partial, st=st)
            kw.update(e_partial=ws.e_partial, e_scale=self._pair_scale(ke, N * Te, D))
        if self.has_ce:
            if ce_done is not None:
                torch.cuda.current_stream().wait_event(ce_done)
            kw.update(ce_rowloss=ws.ce_rowloss, ce_rows=B2)
        if heads_done:
            torch.cuda.current_stream().wait_event(heads_done[0])
        rscale = 0.5 if (self.views == 2 and c["reduction"] == "mean") else 1.0
        self._opt_gate()   # (whatever the caller does next on this stream is behind the whole optimizer step)
        ops.loss_finalize(N * L, self.views, ws.rowloss, ws.mask, rscale, ws.losses, st=st, **kw)
        if ssim is not None:
            ops.ssim_apply(kind == "none", self.views, ssim[2], rscale, ws.ssim_terms, ws.losses, st=st)
        self.gen += 1
        self._saved = dict(img0=img0, img1=img1, N=N, keep=keep, mm=mm, rscale=rscale, gen=self.gen, spec=bool(self.has_pred and kw_spec))
        return ws

    # ------------------------------------------------------------------ stand-alone halves (inference, one view)
    def encode(self, imgs: torch.Tensor, mask_ratio: float, noise: torch.Tensor):
        """`forward_encoder` of the reference (MAE_ViT_Baseline.py:243-266) for a single-view engine: patch-embed the kept patches,
        pos-embed, cls, encoder blocks (encoder_norm's output is discarded there).  -> latent [N, keep+1, D], mask, ids_restore."""
        assert self.views == 1, "encode()/decode() run on a single-view (Baseline-variant) engine"
        c = self.cfg
        N = imgs.shape[0]
        keep = int(c["L"] * (1 - mask_ratio))
        if keep < 1:
            raise ValueError(f"mask_ratio={mask_ratio} keeps no patch (L={c['L']})")
        if self.ws is None or self.ws.N != N or self.ws.keep != keep:
            self.ws = None
            self.ws = Workspace(self, N, keep)
        ws, P = self.ws, self.flat.P
        self.st = st = ops.stream()
        L, D = c["L"], c["D"]
        self._opt_gate()
        self._refresh_lp()
        self._ks_wait()
        self._fp8_begin()
        ws.noise.copy_(noise)
        ops.mask_sort(ws.noise, keep, ws.ids_restore, ws.mask, ws.ids_keep, st=st)
        ops.patch_gather(imgs, None, ws.ids_keep, ws.a_pe, N, c["C"], c["S"], c["p"], keep, st=st)
        ops.gemm(ws.a_pe, self._w_pe(), ws.tok, bias=P("patch_embed.proj.bias"), st=st)
        ops.embed_assemble(ws.tok, P("encoder_pos_embed").view(L + 1, D), P("cls_token").view(D), ws.ids_keep, ws.enc["x"][0], N, keep, st=st)
        for i in range(c["Ne"]):
            self._block_fwd(ws.enc, i, f"encoder.{i}.", ws.Me, D, c["He"], N, ws.Te)
        lat = ws.enc["x"][c["Ne"]]
        if ws.lat32 is not None:
            ops.cast_f32(lat, ws.lat32, st=st)
            lat = ws.lat32
        return lat.view(N, ws.Te, D).clone(), ws.mask.clone(), ws.ids_restore.clone()

    def decode(self, latent: torch.Tensor, ids_restore: torch.Tensor):
        """`forward_decoder` (MAE_ViT_Baseline.py:268-297): decoder_embed, mask-token fill + unshuffle + pos-embed, decoder blocks,
        decoder_norm, decoder_pred.  -> pred [N, L, P] (cls dropped), x_embed [N, L+1, Dd] (post-norm, cls included)."""
        assert self.views == 1
        c = self.cfg
        N, Te, D = latent.shape
        keep, L, Dd = Te - 1, c["L"], c["Dd"]
        if self.ws is None or self.ws.N != N or self.ws.keep != keep:
            self.ws = None
            self.ws = Workspace(self, N, keep)
        ws, P = self.ws, self.flat.P
        self.st = st = ops.stream()
        self._opt_gate()
        self._refresh_lp()
        self._ks_wait()
        self._fp8_begin()
        lat = latent.reshape(N * Te, D).to(torch.float32).contiguous()
        if self.T == BF16:
            lat_op = ws.lat_lp if ws.lat_lp is not None else ws.enc["x"][c["Ne"]]
            ops.cast_bf16(lat, lat_op, st=st)
        else:
            lat_op = lat
        ws.ids_restore.copy_(ids_restore)
        ops.gemm(lat_op, self.W("decoder_embed.weight"), ws.z, bias=P("decoder_embed.bias"), st=st)
        ops.unshuffle_fwd(ws.z, P("mask_token").view(Dd), P("decoder_pos_embed").view(L + 1, Dd), ws.ids_restore, ws.dec["x"][0], N, L, keep, st=st)
        for i in range(c["Nd"]):
            self._block_fwd(ws.dec, i, f"decoder.{i}.", ws.Md, Dd, c["Hd"], N, ws.Td)
        ops.layernorm_fwd(ws.dec["x"][c["Nd"]], P("decoder_norm.weight"), P("decoder_norm.bias"), ws.emb_lp, ws.dn_st[0], ws.dn_st[1], y32=ws.emb32, st=st)
        ops.gemm(ws.emb_lp, self._w_pred()[: c["P"]], ws.pred, bias=P("decoder_pred.bias"), st=st)
        pred = ws.pred
        if pred.dtype != torch.float32:   # (throughput mode writes bf16 predictions; the stand-alone decoder returns fp32 like its latent input)
            pred = torch.empty(ws.pred.shape, device=self.device, dtype=torch.float32)
            ops.cast_f32(ws.pred, pred, st=st)
        return pred.view(N, ws.Td, c["P"])[:, 1:, :].clone(), ws.emb32.view(N, ws.Td, Dd).clone()

    @staticmethod
    def _pair_scale(kind, rows, D):
        return 1.0 / (rows * D) if kind in ("mse", "mae") else 1.0 / rows

    # ------------------------------------------------------------------ backward
    def backward(self, gout: torch.Tensor, accumulate: bool, gen: Optional[int] = None):
        """Engine._backward, optionally with the main chain confined to a CU range for the length of the reverse pass (experiment aid,
        DESIGN §5 "CU partition": CSMAE_BWD_MAIN_CUS=lo:hi mask bits; the forward pass keeps the whole chip)."""
        rng = debug_opt("bwd_main_cus")
        if not rng:
            with trace.range_("csmae.backward"):
                return self._backward(gout, accumulate, gen)
        lo, hi = (int(v) for v in rng.split(":"))
        outer, inner = torch.cuda.current_stream(), ops.cu_masked_stream(lo, hi)
        inner.wait_stream(outer)
        with torch.cuda.stream(inner):
            self._backward(gout, accumulate, gen)
        outer.wait_stream(inner)

    def _backward(self, gout: torch.Tensor, accumulate: bool, gen: Optional[int] = None):
        """Reverse pass of the LAST forward (the activations live in the engine's one workspace).  `gen` = the forward this call
        belongs to (the autograd node passes it): a forward that has been overwritten by a later one, or whose backward already ran,
        raises instead of silently differentiating somebody else's activations."""
        c, ws, sv = self.cfg, self.ws, self._saved
        if sv is None:
            raise RuntimeError("backward() called without a preceding forward()")
        if gen is not None and gen != sv["gen"]:
            raise RuntimeError("csmae_hip: backward of a forward whose activations are gone — the model ran another forward (training, eval "
                               "or viz) in between; the engine keeps ONE activation workspace, so call loss.backward() before the next "
                               "model(...) (accumulate gradients across backward calls, not across forwards)")
        if sv["gen"] == self._gen_done:
            raise RuntimeError("csmae_hip: backward called twice for the same forward (retain_graph is not supported: the reverse pass "
                               "consumes the workspace)")
        self._gen_done = sv["gen"]
        P, G = self.flat.P, self.flat.G
        self.st = st = ops.stream()
        self._opt_gate()   # (the gradient buffer and the gate slot are the optimizer step's inputs)
        N, keep = sv["N"], sv["keep"]
        B2, Te, Td, L, D, Dd = ws.B2, ws.Te, ws.Td, c["L"], c["D"], c["Dd"]
        self.main = torch.cuda.current_stream()
        if self.side is None:
            self.side = self._new_side()
        self._ev_i, self._tog = 0, 0
        self._side_reads.clear()
        zeroed = None
        if not accumulate:
            # The gradient clear (455 MB for ViT-B) runs on the weight-gradient stream, which is idle between the forward and the backward
            # pass, beside the reconstruction head's backward on the main stream: every weight-gradient launch follows it in stream order,
            # the main stream's own writers into the buffer (BatchNorm / LayerNorm / token gradients) wait for `zeroed` below.
            if ops._timer is not None or os.environ.get("CSMAE_DW_MAIN") or debug_opt("zero_main"):   # (CSMAE_DEBUG=zero_main: A/B aid)
                self.flat.g[: self.flat.total].zero_()
            else:
                self.side.wait_stream(self.main)
                with torch.cuda.stream(self.side):
                    self.flat.g[: self.flat.total].zero_()
                zeroed = self._event()
                zeroed.record(self.side)
        ops.gate_accumulate(ws.losses, self.flat.gate, accumulate, st=st)   # (the gate slot sits behind `total`: written here, not cleared)
        ws.gout.copy_(gout.reshape(1).to(torch.float32))
        head_start = self._event()
        head_start.record(self.main)
        kind, npx = c["loss"], c["norm_pix"]
        trace.push("csmae.backward.junction")   # loss heads' backward up to the decoder's last LayerNorm
        # reconstruction head
        extra = None
        ssim = SSIM_KINDS.get(kind)
        if ssim is not None:
            kind, extra = ssim[0], ws.ssim_extra
            ops.ssim_bwd(ssim[1], ws.pred, ws.mask, ws.gout, sv["rscale"] * ssim[2], ws.ssim_ws, extra, B2, N, c["C"], c["S"], c["p"], st=st)
        ops.recon_loss_bwd(kind, npx, sv["img0"], sv["img1"], ws.pred, sv["mm"], ws.mask, ws.losses, ws.gout, sv["rscale"], ws.dpred_lp,
                           B2, N, c["C"], c["S"], c["p"], extra=extra, st=st)
        timed = ops._timer is not None or bool(os.environ.get("CSMAE_DW_MAIN"))
        heads_serial = timed or not self.has_pred
        if heads_serial:   # (otherwise: with the predictor's weight gradients, once the junction is over — below)
            self._dw(ws.dpred_lp, ws.emb_lp, "decoder_pred")
        ops.gemm(ws.dpred_lp, self._w_pred(), ws.demb, trans_b=True, st=st)
        if zeroed is not None:
            self.main.wait_event(zeroed)
        if self.has_pred:
            kcd = c["loss_cd"]
            bn = "predictor.1."
            if heads_serial:   # (one chain on the main stream: per-kernel timing, A/B aid)
                ops.pair_loss_bwd(kcd, N * L, Dd, ws.v, (N * L, 0, 0), ws.emb32, (L, Td, 1), ws.gout, self._pair_scale(kcd, N * L, Dd),
                                  da_lp=ws.dv, dt_acc=ws.demb, st=st)
                self._dw(ws.dv, ws.r, "predictor.3")
                ops.gemm(ws.dv, self.W("predictor.3.weight"), ws.dr, trans_b=True, st=st)
                ops.bnrelu_bwd(ws.u, ws.dr, P(bn + "weight"), P(bn + "bias"), ws.bn_st[0], ws.bn_st[1], ws.dr, G(bn + "weight"), G(bn + "bias"), N, L, st=st)
                self._dw(ws.dr, ws.pin, "predictor.0")
                ops.gemm(ws.dr, self.W("predictor.0.weight"), ws.dpin, trans_b=True, st=st)
                ops.rows_scatter_add(ws.dpin, ws.demb, L, Td, N * Td + 1, st=st)
            else:
                # The predictor's backward (pair loss -> Linear -> BatchNorm/ReLU -> Linear: four kernels, ~340 us) depends on the reconstruction
                # head's (recon_bwd -> decoder_pred dX, above) only through the buffer both add into.  It runs on the auxiliary stream beside
                # it; the two contributions to the decoder-embedding gradient — minus the loss gradient into the original's rows (the target is
                # not detached), the predictor's input gradient into the crop's — are added by ONE kernel once both chains are done.  The
                # weight-gradient launches wait for events recorded on the auxiliary stream.
                aux, ast = self.aux, self.aux.cuda_stream
                aux.wait_event(head_start)                   # gout, and everything the forward pass left on the main stream
                if sv.get("spec"):   # the chain ran with unit gradient behind the forward pass (_forward): apply the real upstream gradient
                    if zeroed is not None:
                        aux.wait_event(zeroed)
                    ops.spec_fixup(ws.gout, (ws.dv, ws.dr, ws.dpin), ws.bn_tmp, G(bn + "weight"), G(bn + "bias"), st=ast)
                else:
                    ops.pair_loss_bwd(kcd, N * L, Dd, ws.v, (N * L, 0, 0), ws.emb32, (L, Td, 1), ws.gout, self._pair_scale(kcd, N * L, Dd),
                                      da_lp=ws.dv, st=ast)
                    ops.gemm(ws.dv, self.W("predictor.3.weight"), ws.dr, trans_b=True, st=ast)
                    if zeroed is not None:   # BatchNorm's parameter gradients go straight into the flat buffer: the chain's first writer into it waits
                        aux.wait_event(zeroed)   # for the clear — not its head (the clear is a 117-us fill that starts with the backward pass)
                    ops.bnrelu_bwd(ws.u, ws.dr, P(bn + "weight"), P(bn + "bias"), ws.bn_st[0], ws.bn_st[1], ws.dr, G(bn + "weight"), G(bn + "bias"), N, L, st=ast)
                    ops.gemm(ws.dr, self.W("predictor.0.weight"), ws.dpin, trans_b=True, st=ast)
                self.main.wait_stream(aux)
                ops.rows_scatter_add2(ws.dv, -1.0, 1, ws.dpin, 1.0, N * Td + 1, ws.demb, L, Td, st=st)
                # the heads' weight-gradient launches wait for the junction to be over (an event behind the combining kernel): started as soon
                # as their operands exist, their 160 workgroups each took the CUs this chain — the longest of the junction — was running on
                # (22.19 vs 22.22 ms: within the noise, kept for the shorter junction)
                ej = self._event()
                ej.record(self.main)
                self._dw_group([(ws.dpred_lp, ws.emb_lp, "decoder_pred")], ready=ej)
                self._dw_group([(ws.dv, ws.r, "predictor.3"), (ws.dr, ws.pin, "predictor.0")], ready=ej)
        trace.pop()
        trace.push("csmae.backward.decoder")
        # decoder
        lp_stream = self.res_dtype != torch.float32   # bf16 residual-gradient stream: the ping-pong buffers are the stream itself
        pd, Nd2 = ws.ln_part_d, 2 * c["Nd"]
        self._fp8_cur = None
        if lp_stream:
            k0 = self._fp8_alloc()
            e0 = None
            if self.fp8 and self._fp8_fuse_lnb:
                e0 = self._emit_to(k0, ws.dres_d_8[0], 1) if self.fp8_dw else self._emit(k0, ws.q_a[1], ws.Md, Dd, 1)
            ops.layernorm_bwd(ws.demb, ws.dec["x"][c["Nd"]], ws.dn_st[0], ws.dn_st[1], P("decoder_norm.weight"), ws.dres_d_lp[0], None, None,
                              partial_ws=pd[Nd2], emit=e0, st=st)
            if k0 is not None:
                self._fp8_cur = (k0, e0[0] if e0 else None)
        else:
            ops.layernorm_bwd(ws.demb, ws.dec["x"][c["Nd"]], ws.dn_st[0], ws.dn_st[1], P("decoder_norm.weight"), ws.dres_d, None, None,
                              dx_lp=ws.dres_d_lp[0], partial_ws=pd[Nd2], st=st)
        kd_ = 0
        for i in reversed(range(c["Nd"])):
            kd_ = self._block_bwd(ws.dec, i, f"decoder.{i}.", ws.Md, Dd, c["Hd"], B2, Td, ws.dres_d, ws.dres_d_lp, kd_, (pd[2 * i], pd[2 * i + 1]),
                                  lps8=ws.dres_d_8 if (self.fp8 and self.fp8_dw and lp_stream) else None)
        if lp_stream and self._ln_fused_bwd(ws.Md, Dd):   # the blocks' LayerNorms left one partial row per GEMM tile, decoder_norm's kernel its own count
            self._ln_flush(pd, self._goff_d, 0, Nd2, ws.Md, Dd, fused=True)
            self._ln_flush(pd, self._goff_d, Nd2, Nd2 + 1, ws.Md, Dd)
        else:
            self._ln_flush(pd, self._goff_d, 0, Nd2 + 1, ws.Md, Dd)
        ops.unshuffle_bwd(ws.dres_d_lp[kd_] if lp_stream else ws.dres_d, ws.ids_restore, ws.dz_lp, G("mask_token").view(Dd), B2, L, keep, st=st)
        lat_op = ws.enc["x"][c["Ne"]] if (lp_stream or self.T != BF16) else ws.lat_lp
        self._dw(ws.dz_lp, lat_op, "decoder_embed")
        ops.gemm(ws.dz_lp, self.W("decoder_embed.weight"), ws.dres_e, trans_b=True, st=st)
        dp = getattr(self.module, "_dp", None)
        if dp is not None:   # decoder + heads are final: their all-reduce overlaps the encoder backward.  The exchange stream waits for
            # the weight-gradient stream itself (`also`): the main chain does not stop for the block's grouped launches to finish
            dp.grads_ready(self.flat, "tail", also=self.side if not (ops._timer is not None or os.environ.get("CSMAE_DW_MAIN")) else None)
        trace.pop()
        trace.push("csmae.backward.encoder")
        latent = ws.lat32 if ws.lat32 is not None else ws.enc["x"][c["Ne"]]
        if self.has_le:
            ke = c["loss_e"]
            ops.pair_loss_bwd(ke, N * Te, D, latent, (N * Te, 0, N * Te), latent, (N * Te, 0, 0), ws.gout, self._pair_scale(ke, N * Te, D),
                              da_acc=ws.dres_e, dt_acc=ws.dres_e, lp_dtype=self.T, st=st)
        dpool = None
        if self.has_ce:
            ops.ntxent_bwd(ws.zc, ws.inv_norm, ws.E, ws.neg, ws.gout, ws.dpool, N, st=st)
            dpool = ws.dpool
        ops.latent_grad_finish(ws.dres_e, dpool, 1.0 / keep, ws.dres_e_lp[0], B2, Te, st=st)
        self._fp8_cur = None
        pe, flushed = ws.ln_part_e, c["Ne"]
        fe = lp_stream and self._ln_fused_bwd(ws.Me, D)
        ke_ = 0
        for i in reversed(range(c["Ne"])):
            ke_ = self._block_bwd(ws.enc, i, f"encoder.{i}.", ws.Me, D, c["He"], B2, Te, None if lp_stream else ws.dres_e, ws.dres_e_lp, ke_, (pe[2 * i], pe[2 * i + 1]),
                                  lps8=ws.dres_e_8 if (self.fp8 and self.fp8_dw and lp_stream) else None)
            if dp is not None and dp.wants(("enc", i)):
                self._ln_flush(pe, self._goff_e, 2 * i, 2 * flushed, ws.Me, D, fused=fe)   # the bucket's LayerNorm gradients must be final before its exchange
                flushed = i
                dp.grads_ready(self.flat, ("enc", i), also=self.side if not (ops._timer is not None or os.environ.get("CSMAE_DW_MAIN")) else None)
        self._ln_flush(pe, self._goff_e, 0, 2 * flushed, ws.Me, D, fused=fe)
        ops.embed_assemble_bwd(ws.dres_e_lp[ke_] if lp_stream else ws.dres_e, ws.dtok_lp, G("cls_token").view(D), B2, keep, st=st)
        self._join_side()
        ops.DwGroup([(ws.dtok_lp, ws.a_pe[:, : c["P"]], G("patch_embed.proj.weight").view(D, c["P"]), G("patch_embed.proj.bias"))], ws.dw_ws).launch(256, st=st)
        if dp is not None:
            dp.grads_ready(self.flat, "stem")
            dp.backward_done(self.flat)
        trace.pop()
        if self.fp8:
            ws.fp8_complete = True
        # hand the gradient views to autograd's .grad slots (frozen and unused parameters keep None — encoder_norm: E1)
        for name, p in self.flat.params.items():
            if p.requires_grad and not name.startswith("encoder_norm."):
                p.grad = self.flat.grad_views[name]
            elif not (name in _FROZEN or name.startswith("encoder_norm.")):
                # a parameter the USER froze: the kernels write dW / db of every layer whether or not its Parameter takes part, so its slot is
                # cleared here (behind the last writer and the all-reduce on this stream) — the flat buffer then holds exactly what torch's
                # `.grad`s would, and global-norm clipping can norm it as a whole (util.misc.clip_grad_norm_; util/misc.py:310-318)
                self.flat.grad_views[name].zero_()

