"""Oracle restatement of the Encodec 24 kHz DECODE and ENCODE paths (decode_codec.py:8-16, data_preparation/
generate_code.py:45-51 -> encodec ^0.1.1).  TEST INFRA ONLY.

The `encodec` package is absent from /root/reference and from this image; pretrained weights are fetched by URL in
the reference and are unavailable offline.  This follows the published architecture (SEANet decoder: causal,
reflect-padded, weight-normed convs; 2-layer LSTM with skip; ratios 8,5,4,2; RVQ of 1024 x 128 codebooks, 8 used at
6 kbps) and is validated in tests/test_encodec_cpu.py against the locally installed
transformers.models.encodec.EncodecModel with seeded random weights (same architecture, independent code).

Weights are a flat dict of EFFECTIVE tensors (weight-norm already folded):
  codebooks (n_q,1024,128) | conv0.{w,b} (512,128,7) | lstm.{w_ih,w_hh,b_ih,b_hh}{0,1}
  up{i}.{w,b} ConvTranspose1d (Cin,Cout,2r) | res{i}.c3.{w,b} (C/2,C,3) | res{i}.c1.{w,b} (C,C/2,1) | res{i}.sc.{w,b} (C,C,1)
  final.{w,b} (1,32,7)
"""
import torch
import torch.nn.functional as F

RATIOS = (8, 5, 4, 2)


def causal_conv1d(x, w, b):
    """Conv1d stride 1 with reflect left padding of (k-1) (SConv1d causal=True, pad_mode='reflect')."""
    pad = w.shape[2] - 1
    if pad > 0:
        T = x.shape[-1]
        extra = 0
        if T <= pad:                                    # the reference pads zeros on the right before reflecting
            extra = pad - T + 1
            x = F.pad(x, (0, extra))
        x = F.pad(x, (pad, 0), mode="reflect")
        if extra:
            x = x[..., : x.shape[-1] - extra]
    return F.conv1d(x, w, b)


def causal_conv_transpose1d(x, w, b, stride):
    """ConvTranspose1d(k=2r, stride=r) then trim (k - r) samples on the right (causal, trim_right_ratio=1)."""
    y = F.conv_transpose1d(x, w, b, stride=stride)
    return y[..., : y.shape[-1] - (w.shape[2] - stride)]


def lstm2_skip(x, W):
    """x (B,C,T) -> 2-layer LSTM over time + skip."""
    h = x.permute(2, 0, 1)
    inp = h
    for l in range(2):
        w_ih, w_hh, b_ih, b_hh = (W[f"lstm.{n}{l}"] for n in ("w_ih", "w_hh", "b_ih", "b_hh"))
        H = w_hh.shape[1]
        hs = torch.zeros(inp.shape[1], H, dtype=x.dtype); cs = torch.zeros_like(hs)
        outs = []
        for t in range(inp.shape[0]):
            g = inp[t] @ w_ih.t() + b_ih + hs @ w_hh.t() + b_hh
            i, f, gg, o = g.chunk(4, dim=1)
            cs = torch.sigmoid(f) * cs + torch.sigmoid(i) * torch.tanh(gg)
            hs = torch.sigmoid(o) * torch.tanh(cs)
            outs.append(hs)
        inp = torch.stack(outs)
    return (inp + h).permute(1, 2, 0)


def rvq_decode(codes, codebooks):
    """codes (B,n_q,T) int64 -> (B,128,T): sum of codebook rows."""
    out = 0
    for q in range(codes.shape[1]):
        out = out + F.embedding(codes[:, q], codebooks[q])
    return out.permute(0, 2, 1)


def decode(codes, W):
    """codes (B, n_q, T) int64 in [0, 1023] -> wav (B, 1, 320*T) f32."""
    if codes.dim() != 3:
        raise BaseException("The encoded_frames must have the shape of [B, N_q, T]")
    x = rvq_decode(codes, W["codebooks"])
    x = causal_conv1d(x, W["conv0.w"], W["conv0.b"])
    x = lstm2_skip(x, W)
    for i, r in enumerate(RATIOS):
        x = causal_conv_transpose1d(F.elu(x), W[f"up{i}.w"], W[f"up{i}.b"], r)
        h = causal_conv1d(F.elu(x), W[f"res{i}.c3.w"], W[f"res{i}.c3.b"])
        h = causal_conv1d(F.elu(h), W[f"res{i}.c1.w"], W[f"res{i}.c1.b"])
        x = causal_conv1d(x, W[f"res{i}.sc.w"], W[f"res{i}.sc.b"]) + h
    return causal_conv1d(F.elu(x), W["final.w"], W["final.b"])


# ---------------------------------------------------------------------------------------------------------------------
# ENCODE path (data_preparation/generate_code.py:45-51 -> encodec ^0.1.1 `model.encode`): SEANet encoder (mirror of the
# decoder: conv k7 1->32; four stages {residual block, ELU, causal strided conv k=2r stride r, C -> 2C} with r = 2,4,5,8;
# 2-layer LSTM + skip; ELU; conv k7 512->128), then residual vector quantisation with the first n_q codebooks:
# idx = argmax_j -(|x|^2 - 2 x.e_j + |e_j|^2), residual -= e_idx.  Lengths that are multiples of 320 need no extra padding.
# Encoder weights (effective): enc.conv0.{w,b} (32,1,7) | enc.res{i}.{c3,c1,sc}.{w,b} | enc.down{i}.{w,b} (2C,C,2r) |
# enc.lstm.{w_ih,w_hh,b_ih,b_hh}{0,1} | enc.final.{w,b} (128,512,7)
# ---------------------------------------------------------------------------------------------------------------------
ENC_RATIOS = (2, 4, 5, 8)


def causal_strided_conv1d(x, w, b, stride):
    """Conv1d(k, stride) with reflect left padding of (k - stride); input length must be a multiple of the stride."""
    pad = w.shape[2] - stride
    if x.shape[-1] % stride != 0:
        raise ValueError("length must be a multiple of the stride (no extra right padding is modelled)")
    if pad > 0:
        x = F.pad(x, (pad, 0), mode="reflect")
    return F.conv1d(x, w, b, stride=stride)


def encoder_embeddings(wav, W):
    """wav (B, 1, L) f32, L % 320 == 0 -> embeddings (B, 128, L/320)."""
    if wav.dim() != 3 or wav.shape[1] != 1 or wav.shape[-1] % 320 != 0:
        raise ValueError("wav must be (B, 1, L) with L a multiple of 320")
    x = causal_conv1d(wav, W["enc.conv0.w"], W["enc.conv0.b"])
    for i, r in enumerate(ENC_RATIOS):
        h = causal_conv1d(F.elu(x), W[f"enc.res{i}.c3.w"], W[f"enc.res{i}.c3.b"])
        h = causal_conv1d(F.elu(h), W[f"enc.res{i}.c1.w"], W[f"enc.res{i}.c1.b"])
        x = causal_conv1d(x, W[f"enc.res{i}.sc.w"], W[f"enc.res{i}.sc.b"]) + h
        x = causal_strided_conv1d(F.elu(x), W[f"enc.down{i}.w"], W[f"enc.down{i}.b"], r)
    x = lstm2_skip(x, {k[4:]: v for k, v in W.items() if k.startswith("enc.lstm.")})
    return causal_conv1d(F.elu(x), W["enc.final.w"], W["enc.final.b"])


def rvq_encode(emb, codebooks, dtype=torch.float64):
    """emb (B,128,T) -> codes (B,n_q,T) int64.  Evaluated in `dtype` (f64: the reference answer a f32 search is judged by;
    returns also the gap between best and second-best score per search, for near-tie bookkeeping in the tests)."""
    B, D, T = emb.shape
    res = emb.permute(0, 2, 1).reshape(B * T, D).to(dtype)
    codes, gaps = [], []
    for q in range(codebooks.shape[0]):
        e = codebooks[q].to(dtype)
        dist = -(res.pow(2).sum(1, keepdim=True) - 2 * res @ e.t() + e.pow(2).sum(1)[None, :])
        top2 = dist.topk(2, dim=1)
        idx = dist.max(dim=1).indices
        codes.append(idx); gaps.append(top2.values[:, 0] - top2.values[:, 1])
        res = res - e[idx]
    codes = torch.stack(codes, 1).view(B, T, -1).permute(0, 2, 1).contiguous()
    gaps = torch.stack(gaps, 1).view(B, T, -1).permute(0, 2, 1).contiguous()
    return codes, gaps


def encode(wav, W, dtype=torch.float32):
    """wav (B,1,L) -> codes (B, n_q, L/320) int64 (generate_code.py:48 `torch.cat([e[0] for e in encoded_frames], -1)`)."""
    return rvq_encode(encoder_embeddings(wav, W), W["codebooks"], dtype)[0]


def random_encoder_weights(seed=1, n_q=8, scale=1.0):
    g = torch.Generator().manual_seed(seed)

    def t(*shape, fan=None):
        fan = fan or (shape[1] * (shape[2] if len(shape) > 2 else 1))
        return (torch.rand(shape, generator=g) * 2 - 1) * (3.0 / fan) ** 0.5 * scale

    W = {"codebooks": torch.randn(n_q, 1024, 128, generator=g) * 0.5,
         "enc.conv0.w": t(32, 1, 7), "enc.conv0.b": t(32, fan=100)}
    C = 32
    for i, r in enumerate(ENC_RATIOS):
        W[f"enc.res{i}.c3.w"] = t(C // 2, C, 3); W[f"enc.res{i}.c3.b"] = t(C // 2, fan=100)
        W[f"enc.res{i}.c1.w"] = t(C, C // 2, 1); W[f"enc.res{i}.c1.b"] = t(C, fan=100)
        W[f"enc.res{i}.sc.w"] = t(C, C, 1); W[f"enc.res{i}.sc.b"] = t(C, fan=100)
        W[f"enc.down{i}.w"] = t(2 * C, C, 2 * r); W[f"enc.down{i}.b"] = t(2 * C, fan=100)
        C *= 2
    for l in range(2):
        W[f"enc.lstm.w_ih{l}"] = t(2048, 512); W[f"enc.lstm.w_hh{l}"] = t(2048, 512)
        W[f"enc.lstm.b_ih{l}"] = t(2048, fan=100); W[f"enc.lstm.b_hh{l}"] = t(2048, fan=100)
    W["enc.final.w"] = t(128, 512, 7); W["enc.final.b"] = t(128, fan=100)
    return W


def encoder_weights_from_hf(model):
    """Effective ENCODER weights (+ codebooks) out of a transformers EncodecModel."""
    def eff(conv):
        return conv.weight.detach().clone(), conv.bias.detach().clone()
    L = model.encoder.layers
    W = {"codebooks": torch.stack([q.codebook.embed.detach().clone() for q in model.quantizer.layers])}
    W["enc.conv0.w"], W["enc.conv0.b"] = eff(L[0].conv)
    idx = 1
    for i in range(4):
        rb = L[idx]
        W[f"enc.res{i}.c3.w"], W[f"enc.res{i}.c3.b"] = eff(rb.block[1].conv)
        W[f"enc.res{i}.c1.w"], W[f"enc.res{i}.c1.b"] = eff(rb.block[3].conv)
        W[f"enc.res{i}.sc.w"], W[f"enc.res{i}.sc.b"] = eff(rb.shortcut.conv)
        W[f"enc.down{i}.w"], W[f"enc.down{i}.b"] = eff(L[idx + 2].conv)
        idx += 3
    lstm = L[idx].lstm
    for l in range(2):
        for n in ("w_ih", "w_hh", "b_ih", "b_hh"):
            W[f"enc.lstm.{n}{l}"] = getattr(lstm, f"{'weight' if n[0] == 'w' else 'bias'}_{n[2:]}_l{l}").detach().clone()
    W["enc.final.w"], W["enc.final.b"] = eff(L[idx + 2].conv)
    return W


def random_weights(seed=0, n_q=8, scale=1.0):
    """Seeded effective weights of the 24 kHz architecture (fan-in scaled, so activations stay O(1))."""
    g = torch.Generator().manual_seed(seed)

    def t(*shape, fan=None):
        fan = fan or (shape[1] * (shape[2] if len(shape) > 2 else 1))
        return (torch.rand(shape, generator=g) * 2 - 1) * (3.0 / fan) ** 0.5 * scale

    W = {"codebooks": torch.randn(n_q, 1024, 128, generator=g) * 0.5,
         "conv0.w": t(512, 128, 7), "conv0.b": t(512, fan=100)}
    for l in range(2):
        W[f"lstm.w_ih{l}"] = t(2048, 512); W[f"lstm.w_hh{l}"] = t(2048, 512)
        W[f"lstm.b_ih{l}"] = t(2048, fan=100); W[f"lstm.b_hh{l}"] = t(2048, fan=100)
    C = 512
    for i, r in enumerate(RATIOS):
        W[f"up{i}.w"] = t(C, C // 2, 2 * r, fan=2 * C); W[f"up{i}.b"] = t(C // 2, fan=100)
        C //= 2
        W[f"res{i}.c3.w"] = t(C // 2, C, 3); W[f"res{i}.c3.b"] = t(C // 2, fan=100)
        W[f"res{i}.c1.w"] = t(C, C // 2, 1); W[f"res{i}.c1.b"] = t(C, fan=100)
        W[f"res{i}.sc.w"] = t(C, C, 1); W[f"res{i}.sc.b"] = t(C, fan=100)
    W["final.w"] = t(1, 32, 7); W["final.b"] = t(1, fan=100)
    return W


def weights_from_hf(model):
    """Effective weights out of a transformers EncodecModel (weight-norm parametrisations folded)."""
    def eff(conv):
        return conv.weight.detach().clone(), conv.bias.detach().clone()
    L = model.decoder.layers
    W = {"codebooks": torch.stack([q.codebook.embed.detach().clone() for q in model.quantizer.layers])}
    W["conv0.w"], W["conv0.b"] = eff(L[0].conv)
    lstm = L[1].lstm
    for l in range(2):
        for n in ("w_ih", "w_hh", "b_ih", "b_hh"):
            W[f"lstm.{n}{l}"] = getattr(lstm, f"{'weight' if n[0] == 'w' else 'bias'}_{n[2:]}_l{l}").detach().clone()
    idx = 2
    for i in range(4):
        W[f"up{i}.w"], W[f"up{i}.b"] = eff(L[idx + 1].conv)
        rb = L[idx + 2]
        W[f"res{i}.c3.w"], W[f"res{i}.c3.b"] = eff(rb.block[1].conv)
        W[f"res{i}.c1.w"], W[f"res{i}.c1.b"] = eff(rb.block[3].conv)
        W[f"res{i}.sc.w"], W[f"res{i}.sc.b"] = eff(rb.shortcut.conv)
        idx += 3
    W["final.w"], W["final.b"] = eff(L[idx + 1].conv)
    return W
