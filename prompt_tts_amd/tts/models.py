"""TTSSingleSpeaker / TextEncoder on the HIP kernels (reference: tts/models.py:11-70,73-120,123-172).

Drop-in surface: `TTSSingleSpeaker(config)` with the reference's 1d_config keys, `forward(sample, timestep,
text_seq_ids, attention_mask, cross_attention_kwargs=None, return_dict=True)` -> object with `.sample`
(B, n_q, T), identical `state_dict` keys.  The module is constructible on the CPU (for checkpoints); compute
needs the model on an MI355X (`.to("cuda")`), there is no CPU compute path.

`forward` is differentiable through torch autograd as ONE node (loss.backward() works as in train.py:107-115):
its backward runs the hand-written backward kernels and deposits weight gradients straight into the flat grad
buffer that every `param.grad` is a view of.  `train_step` is the fused path used by train.py / bench.py.
"""
import math
from typing import Any, Dict, Optional, Union

import torch
from torch import nn

from .. import engine as E
from .. import ops
from .ldm.attention import BasicTransformerBlock
from .ldm.unet_1d_condition import Unet1DConditionModel, UNet1DConditionOutput


def positional_table(seq_len_cfg, S, d, inv_freq):
    """The (S, d) table tts/models.py:55-70 adds.  The reference permutes (B,S,d)->(B,d,S) BEFORE the 1-D
    encoding, so position runs over the feature index k and channel over the time index s:
    pos[s,k] = sin(k w_{s//2}) for even s, cos(k w_{s//2}) for odd s.  Constant: built once on the host."""
    ch = int(math.ceil(seq_len_cfg / 2) * 2)
    if S > ch:
        raise RuntimeError(f"text length {S} exceeds the positional channel count {ch} (cmu_seq_len)")
    inv = inv_freq.detach().float().cpu()
    ang = torch.arange(d, dtype=torch.float32)[:, None] * inv[None, :]
    tab = torch.stack((ang.sin(), ang.cos()), dim=-1).flatten(-2, -1)
    return tab[:, :S].transpose(0, 1).contiguous()


class PositionalEncoding1D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.org_channels = channels
        ch = int(math.ceil(channels / 2) * 2)
        self.channels = ch
        self.register_buffer("inv_freq", 1.0 / (10000 ** (torch.arange(0, ch, 2).float() / ch)))


class PositionalEncodingPermute1D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.penc = PositionalEncoding1D(channels)

    @property
    def org_channels(self):
        return self.penc.org_channels


class TextEncoder(nn.Module):
    """mask_mode: "ignored" (default) reproduces the pinned diffusers 0.15.x binding, where the additive mask the
    reference builds (tts/models.py:108-110) lands in `encoder_hidden_states` and is never used;
    "additive" masks padded keys (what diffusers >= 0.17 would do with the same call)."""

    def __init__(self, vocab_len, seq_len, dim, attention_head_dim, dropout=0.0, num_layers=1, mask_mode="ignored"):
        super().__init__()
        if dim % attention_head_dim != 0:
            raise ValueError("dim must be a multipliter of attention_head_dim")
        if mask_mode not in ("ignored", "additive"):
            raise ValueError("mask_mode must be 'ignored' or 'additive'")
        self.mask_mode, self.dim, self.vocab_len = mask_mode, dim, vocab_len
        self.word_embedding = nn.Embedding(vocab_len, dim)
        self.pos_embedding = PositionalEncodingPermute1D(seq_len)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(dim, dim // attention_head_dim, attention_head_dim, dropout=dropout)
             for _ in range(num_layers)])
        self._pos_cache = {}

    def _pos(self, S, device):
        key = (S, str(device))
        if key not in self._pos_cache:
            self._pos_cache[key] = positional_table(self.pos_embedding.org_channels, S, self.dim,
                                                    self.pos_embedding.penc.inv_freq).to(device)
        return self._pos_cache[key]

    def fwd(self, st, ids, mask, B, S):
        out = torch.empty(B * S, self.dim, dtype=st.dtype, device=ids.device)
        ops.embedding_fwd(ids.reshape(-1), st.w(self.word_embedding.weight), self._pos(S, ids.device), out, S)
        kv_len = None
        if self.mask_mode == "additive" and mask is not None:
            kv_len = mask.to(torch.int32).sum(dim=1).to(torch.int32).contiguous()   # collate masks are prefixes
        h, svs = out, []
        for blk in self.transformer_blocks:
            h, sv = blk.fwd(st, h, None, B, S, S, self_kv_len=kv_len)
            svs.append(sv)
        return h, (ids, svs)

    def bwd(self, st, saved, dh):
        ids, svs = saved
        for blk, sv in zip(reversed(self.transformer_blocks), reversed(svs)):
            dh, _ = blk.bwd(st, sv, dh)
        ops.embedding_bwd(ids.reshape(-1), dh, st.g(self.word_embedding.weight))


class _DenoiserFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, sample, timestep, ids, mask):
        out, tape = model._forward_impl(sample, timestep, ids, mask)
        ctx.model, ctx.tape = model, tape
        return out

    @staticmethod
    def backward(ctx, dout):
        ctx.model._backward_impl(ctx.tape, dout.contiguous())
        ctx.tape = None
        return None, None, None, None, None, None


class TTSSingleSpeaker(nn.Module):
    def __init__(self, config, dtype=torch.bfloat16, mask_mode="ignored", fp8=False):
        """fp8=True (bf16 only; BASELINE configs[4]): the feed-forward GEMMs that gain from it run with fp8 operands."""
        super().__init__()
        self.config = dict(config)
        self.compute_dtype = dtype
        if fp8 and dtype != torch.bfloat16:
            raise ValueError("fp8 GEMMs need dtype=torch.bfloat16")
        self.fp8 = bool(fp8)
        self.text_encoder = TextEncoder(
            vocab_len=config["cmu_vocab_len"], seq_len=config["cmu_seq_len"], dim=config["cross_attention_dim"],
            attention_head_dim=config["attention_head_dim"], dropout=config["text_encoder_dropout"],
            num_layers=config["text_encoder_layers"], mask_mode=mask_mode)
        self.unet = Unet1DConditionModel(
            sample_size=config["sample_size"], in_channels=config["in_channels"], out_channels=config["out_channels"],
            layers_per_block=config["layers_per_block"], block_out_channels=config["block_out_channels"],
            down_block_types=config["down_block_types"], mid_block_type=config["mid_block_type"],
            up_block_types=config["up_block_types"], cross_attention_dim=config["cross_attention_dim"])
        self._store = None
        self._anchor = None
        self._alphas_cumprod = None
        self.grad_ready_hook = None      # set by the data-parallel reducer
        # load_state_dict writes through the parameter views: the kernel-layout weight shadow must be repacked afterwards
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.refresh_weights())

    # ---- device state --------------------------------------------------------------------------------------
    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self._store = None               # parameters were re-allocated: rebuild the flat store lazily
        return out

    @property
    def store(self):
        p0 = next(self.parameters())
        if not p0.is_cuda:
            raise RuntimeError("TTSSingleSpeaker computes on an MI355X only: move the model with .to('cuda') first "
                               "(there is no CPU fallback)")
        if self._store is None or self._store.device != p0.device:
            self._store = E.ParamStore(self, p0.device, self.compute_dtype)
            self._store.enable_fp8(self.fp8)
            self._anchor = torch.zeros(1, device=p0.device, requires_grad=True)
        return self._store

    def refresh_weights(self):
        """Call after writing parameters by any route this module cannot see (direct edits of `p.data`); load_state_dict,
        `loss.backward()` followed by a torch optimizer, and the fused AdamW are tracked automatically."""
        if self._store is not None:
            self._store.mark_dirty()

    def alphas_cumprod(self, device):
        """DDPMScheduler(1000, linear 1e-4..0.02) cumulative alphas (train.py:32-36), f32."""
        if self._alphas_cumprod is None or self._alphas_cumprod.device != device:
            betas = torch.linspace(1e-4, 0.02, 1000, dtype=torch.float32)
            self._alphas_cumprod = torch.cumprod(1.0 - betas, dim=0).to(device)
        return self._alphas_cumprod

    # ---- reference-compatible forward ----------------------------------------------------------------------
    def forward(self, sample: torch.Tensor, timestep: Union[torch.Tensor, float, int], text_seq_ids: torch.Tensor,
                attention_mask: torch.Tensor, cross_attention_kwargs: Optional[Dict[str, Any]] = None,
                return_dict: bool = True):
        st = self.store
        sample = sample.to(device=st.device, dtype=torch.float32).contiguous()
        B = sample.shape[0]
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.float64 if isinstance(t, float) else torch.int64)
        elif t.dim() == 0:
            t = t[None]
        t = t.to(st.device).expand(B).to(torch.int64).contiguous()   # integer DDPM steps (train.py:92-95)
        ids = text_seq_ids.to(device=st.device, dtype=torch.int32).contiguous()
        mask = attention_mask.to(st.device) if attention_mask is not None else None
        if torch.is_grad_enabled():
            out = _DenoiserFn.apply(self._anchor, self, sample, t, ids, mask)
        else:
            out, _ = self._forward_impl(sample, t, ids, mask)
        if not return_dict:
            return (out,)
        return UNet1DConditionOutput(sample=out)

    def _check_inputs(self, sample, ids):
        cfg = self.config
        if sample.dim() != 3 or sample.shape[1] != cfg["in_channels"]:
            raise ValueError(f"sample must be (B, {cfg['in_channels']}, T), got {tuple(sample.shape)}")
        if ids.dim() != 2 or ids.shape[0] != sample.shape[0]:
            raise ValueError("text_seq_ids must be (B, S)")

    def _forward_tokens(self, st, xt, t, ids, mask, B, T, S):
        st.ensure_shadow_fresh()
        text_emb, sv_text = self.text_encoder.fwd(st, ids, mask, B, S)
        pred, tape = self.unet.fwd(st, xt, t, text_emb, B, T, S)
        return pred, (sv_text, tape, B, T, S)

    def _backward_tokens(self, st, tape, dpred):
        sv_text, utape, B, T, S = tape
        if next(p for p in st.params if not st.info[id(p)]["frozen"]).grad is None:
            st.zero_grad(); st.attach_grads()            # optimizer.zero_grad(set_to_none=True) dropped the views
        user_hook = self.grad_ready_hook
        hook = None
        if user_hook is not None:
            if getattr(getattr(user_hook, "__self__", None), "joins_side_stream", False):
                def hook(module):                 # the reducer orders its own stream after the wgrad stream: no stall here;
                    E.flush_wgrads(st.device)     # queued weight gradients of the announced module go out first
                    user_hook(module)
            else:
                def hook(module):                 # weight gradients are produced on the side stream: join it first
                    E.join_side_stream(st.device)
                    E.fold_grad_replicas(st.flat_g)
                    user_hook(module)
        dctx = self.unet.bwd(st, utape, dpred, hook)
        self.text_encoder.bwd(st, sv_text, dctx)
        if hook is not None:
            hook(self.text_encoder)
        E.join_side_stream(st.device)
        E.fold_grad_replicas(st.flat_g)

    def _forward_impl(self, sample, t, ids, mask):
        st = self.store
        self._check_inputs(sample, ids)
        B, n_q, T = sample.shape
        S = ids.shape[1]
        cpad = self.unet.cpad
        xt = torch.empty(B * T, cpad, dtype=st.dtype, device=st.device)
        ops.bct_to_tokens(sample, xt, B, n_q, T, cpad)
        pred, tape = self._forward_tokens(st, xt, t, ids, mask, B, T, S)
        out = torch.empty(B, n_q, T, dtype=torch.float32, device=st.device)
        ops.tokens_to_bct(pred, out, B, n_q, T, cpad)
        return out, tape

    def _backward_impl(self, tape, dout):
        st = self.store
        st.mark_dirty()                  # autograd path: a torch optimizer is about to step through the parameter views
        _, _, B, T, S = tape
        n_q, cpad = self.config["in_channels"], self.unet.cpad
        dpred = torch.empty(B * T, cpad, dtype=st.dtype, device=st.device)
        ops.bct_to_tokens(dout.to(torch.float32), dpred, B, n_q, T, cpad)
        self._backward_tokens(st, tape, dpred)

    # ---- fused training step (train.py:86-120 in one pass, no host sync) -------------------------------------
    def loss_and_backward(self, x0, noise, t, ids, mask, loss_out=None, grad_scale=1.0):
        """add_noise -> forward -> MSE(+dpred) -> backward.  x0/noise (B,n_q,T) f32, t int64 (B,).  Returns the f32
        device scalar loss (accumulated into `loss_out` if given).  Gradients accumulate in the flat buffer."""
        st = self.store
        self._check_inputs(x0, ids)
        B, n_q, T = x0.shape
        S = ids.shape[1]
        cpad = self.unet.cpad
        xt = torch.empty(B * T, cpad, dtype=st.dtype, device=st.device)
        ops.add_noise(x0, noise, t, self.alphas_cumprod(st.device), xt, n_q, T, cpad)
        if st.arena is None:
            st.arena = E.ZeroArena(st.device)
        st.arena.reset()                 # one memset for every GroupNorm statistic / workspace of this forward + backward
        st.arena_active = st.arena
        E.set_arena(st.arena)
        try:
            pred, tape = self._forward_tokens(st, xt, t, ids, mask, B, T, S)
            loss = loss_out if loss_out is not None else torch.zeros(1, dtype=torch.float32, device=st.device)
            dpred = torch.empty_like(pred)
            ops.mse_loss(pred, noise, loss, dpred, grad_scale, B, n_q, T, cpad)
            self._backward_tokens(st, tape, dpred)
        finally:
            st.arena_active = None
            E.set_arena(None)
        return loss

    def train_step(self, x0, noise, t, ids, mask, lr=1e-5, betas=(0.95, 0.999), eps=1e-8, weight_decay=1e-6,
                   max_grad_norm=1.0, reducer=None):
        """One optimizer step: zero grads, fused loss+backward, [DP all-reduce], clip, AdamW (train.py:79-120).
        Runs on a high-priority stream ordered after / before the caller's current stream (engine.main_stream)."""
        st = self.store
        hs = E.main_stream(st.device)
        if hs is None:
            return self._train_step(x0, noise, t, ids, mask, lr, betas, eps, weight_decay, max_grad_norm, reducer)
        cur = torch.cuda.current_stream(st.device)
        hs.wait_stream(cur)
        with torch.cuda.stream(hs):
            loss, gn = self._train_step(x0, noise, t, ids, mask, lr, betas, eps, weight_decay, max_grad_norm, reducer)
        cur.wait_stream(hs)
        loss.record_stream(cur); gn.record_stream(cur)
        return loss, gn

    def _train_step(self, x0, noise, t, ids, mask, lr, betas, eps, weight_decay, max_grad_norm, reducer):
        st = self.store
        st.zero_grad()
        if reducer is not None:
            reducer.begin()
        loss = self.loss_and_backward(x0, noise, t, ids, mask,
                                      grad_scale=reducer.grad_scale if reducer is not None else 1.0)
        if reducer is not None:
            reducer.finish()
        if reducer is not None and hasattr(reducer, "owned_ranges") and reducer.world > 1:
            gn = E.adamw_step_sharded(st, reducer, lr, betas, eps, weight_decay, max_grad_norm)
        else:
            gn = st.adamw_step(lr, betas, eps, weight_decay, max_grad_norm)
        return loss, gn
