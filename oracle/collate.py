"""Oracle restatement of the collate (tts/dataloader.py:12-15,52-83,123-188).  TEST INFRASTRUCTURE ONLY.

numpy only.  Integer outputs (ids, mask) are bit-exact targets; the code normalisation follows
the reference's dtype path exactly: float64(code)/1023 (dataloader.py:64) -> float32 (:169,
torch.FloatTensor) -> torchvision Normalize([0.5],[0.5]) = (x - 0.5) / 0.5 in float32 (:143).
"""
import numpy as np


def intersperse(lst, item):
    out = [item] * (len(lst) * 2 + 1)
    out[1::2] = lst
    return out


def pad_ids(seqs, max_length, pad_token_id=0):
    ids = np.full((len(seqs), max_length), pad_token_id, dtype=np.int32)
    mask = np.zeros((len(seqs), max_length), dtype=np.int32)
    for i, s in enumerate(seqs):
        n = min(len(s), max_length)
        ids[i, :n] = np.asarray(s[:n], dtype=np.int64)
        mask[i, :n] = 1
    return ids, mask


def normalise_codes(codes_int):
    x = (np.asarray(codes_int).astype(np.float64) / 1023).astype(np.float32)
    return (x - np.float32(0.5)) / np.float32(0.5)


def denormalise_to_codes(x):
    """Build-defined inverse (SURVEY 8a'): idx = clamp(round((x+1)/2*1023), 0, 1023) -> int64."""
    v = np.rint((np.asarray(x, dtype=np.float32) + np.float32(1.0)) * np.float32(0.5) * np.float32(1023.0))
    return np.clip(v, 0, 1023).astype(np.int64)


def collate(items, max_seq_length):
    ids, mask = pad_ids([it["cmu_sequence"] for it in items], max_seq_length)
    out = {
        "code": normalise_codes(np.stack([np.asarray(it["code_int"]) for it in items])),
        "text": [it["text"] for it in items],
        "code_length": [it["code_length"] for it in items],
        "cmu_sequence": [it["cmu_sequence"] for it in items],
        "cmu_sequence_id": ids,
        "attention_mask": mask,
    }
    if "text_norm" in items[0]:
        out["text_norm"] = [it["text_norm"] for it in items]
    return out


def sample_topk(logits, k=1, uniforms=None, temperature=1.0):
    """Build-defined oracle of greedy / top-k sampling (SURVEY 8a'): torch.topk order, softmax over the k, inverse-CDF
    draw with injected uniforms (first j with cdf_j > u * total)."""
    import torch
    logits = torch.as_tensor(logits, dtype=torch.float32)
    if k == 1:
        return torch.argmax(logits, dim=-1)
    v, i = torch.topk(logits, k, dim=-1)
    e = torch.exp((v - v[:, :1]) / temperature)
    cdf = torch.cumsum(e, dim=-1)
    u = torch.as_tensor(uniforms, dtype=torch.float32)[:, None] * cdf[:, -1:]
    pick = (cdf > u).float().argmax(dim=-1)
    return i.gather(1, pick[:, None])[:, 0]
