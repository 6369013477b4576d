"""Collate + dataset surface of the reference (tts/dataloader.py:12-15,18-90,123-198).

Hot-path row a-9 is the COLLATE: pad/truncate phoneme ids to max_seq_length with pad id 0, 1/0 mask, int32;
codes -> float32(float64(code)/1023) -> (x-0.5)/0.5.  It is CPU-side, integer/byte work on tiny arrays and
runs on the host exactly as in the reference (numpy), producing the same batch dictionary keys.

Phoneme ids come from the text front-end (prompt_tts_amd/tts/process_text: english_cleaners + CMUdict lookup, as
tts/dataloader.py:21-22,52-53) unless the tar ships precomputed `<utt>.cmu.npy` id arrays or the caller passes its own
`text_to_ids` callable; the CMU dictionary FILE is located by process_text.find_cmu_dictionary().
"""
import io
import tarfile

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

BLANK_ID = 148          # len(symbols) in tts/process_text/symbols.py: the interspersed blank (dataloader.py:52-55)


def _default_text_to_ids():
    from .process_text import default_text_to_ids
    return default_text_to_ids()


def intersperse(lst, item):
    result = [item] * (len(lst) * 2 + 1)
    result[1::2] = lst
    return result


def _collate_batch_helpler(examples, pad_token_id, max_length, return_mask=False):
    """(name kept from the reference, typo included) -> python lists, like the reference."""
    ids = np.full((len(examples), max_length), pad_token_id, dtype=np.int64)
    mask = np.zeros((len(examples), max_length), dtype=np.int64)
    for i, ex in enumerate(examples):
        n = min(len(ex), max_length)
        ids[i, :n] = np.asarray(ex[:n], dtype=np.int64)
        mask[i, :n] = 1
    if return_mask:
        return ids.tolist(), mask.tolist()
    return ids.tolist()


def normalise_codes(code_over_1023):
    """torch.FloatTensor(np.array(batch_code)) then torchvision Normalize([0.5],[0.5]) (dataloader.py:143,169)."""
    x = torch.from_numpy(np.asarray(code_over_1023, dtype=np.float64).astype(np.float32))
    return (x - 0.5) / 0.5


class TTS_SingleSpkr_Collate_Fn(object):
    def __init__(self, max_seq_length):
        self.max_seq_length = max_seq_length

    def __call__(self, batch):
        cmu = [it["cmu_sequence"] for it in batch]
        cmu_seq, cmu_mask = _collate_batch_helpler(cmu, 0, self.max_seq_length, return_mask=True)
        out = {
            "code": normalise_codes(np.array([it["code"] for it in batch])),
            "text": [it["text"] for it in batch],
            "code_length": [it["code_length"] for it in batch],
            "cmu_sequence": cmu,
            "cmu_sequence_id": torch.tensor(cmu_seq, dtype=torch.int32),
            "attention_mask": torch.tensor(cmu_mask, dtype=torch.int32),
        }
        if "text_norm" in batch[0]:
            out["text_norm"] = [it["text_norm"] for it in batch]
        return out


class SingleSpeakerDataset(Dataset):
    """Whole tar in RAM, as the reference: <utt>.npy (int64 [n_q,T]), <utt>.txt, [<utt>.normalized.txt], <utt>.len.txt."""

    def __init__(self, data_path, text_to_ids=None):
        super().__init__()
        self.item_list = []
        with tarfile.open(data_path, "r") as tf:
            names = {m.name for m in tf.getmembers()}
            for name in sorted(n for n in names if n.endswith(".npy") and not n.endswith(".cmu.npy")):
                stem = name[:-4]
                code = np.load(io.BytesIO(tf.extractfile(name).read()))
                text = tf.extractfile(stem + ".txt").read().decode()
                has_norm = stem + ".normalized.txt" in names
                text_norm = tf.extractfile(stem + ".normalized.txt").read().decode() if has_norm else text
                if stem + ".cmu.npy" in names:
                    ids = np.load(io.BytesIO(tf.extractfile(stem + ".cmu.npy").read())).tolist()
                else:
                    if text_to_ids is None:
                        text_to_ids = _default_text_to_ids()       # raises FileNotFoundError if no dictionary is found
                    ids = list(text_to_ids(text_norm))
                item = {"code": code / 1023, "text": text, "cmu_sequence": intersperse(ids, BLANK_ID),
                        "code_length": float(tf.extractfile(stem + ".len.txt").read().decode())}
                if has_norm:
                    item["text_norm"] = text_norm
                self.item_list.append(item)

    def __len__(self):
        return len(self.item_list)

    def __getitem__(self, idx):
        return self.item_list[idx]


class LazySingleSpeakerDataset(Dataset):
    """Same items as SingleSpeakerDataset without holding the tar in RAM: one pass over the tar HEADERS builds
    an index (member name -> data offset, size); __getitem__ seeks and reads only that utterance's members.  Random access
    (shuffling) stays cheap because tar data is stored verbatim; every DataLoader worker opens its own file handle.
    Tars written by generate_code.py / encode_codec.py keep all texts at the END of the archive, so a sequential stream would
    have to buffer everything anyway: an offset index is the streaming form that fits this layout."""

    def __init__(self, data_path, text_to_ids=None):
        super().__init__()
        self.data_path, self.text_to_ids = data_path, text_to_ids
        self._fh = None
        with tarfile.open(data_path, "r:") as tf:                       # uncompressed tar: offsets address the file itself
            self.index = {m.name: (m.offset_data, m.size) for m in tf.getmembers() if m.isfile()}
        self.stems = sorted(n[:-4] for n in self.index if n.endswith(".npy") and not n.endswith(".cmu.npy"))
        for stem in self.stems:
            for suffix in (".txt", ".len.txt"):
                if stem + suffix not in self.index:
                    raise KeyError(f"{stem}{suffix} missing from {data_path}")
            if stem + ".cmu.npy" not in self.index and self.text_to_ids is None:
                self.text_to_ids = _default_text_to_ids()

    def _read(self, name):
        if self._fh is None:                                            # one handle per process (DataLoader workers fork)
            self._fh = open(self.data_path, "rb")
        off, size = self.index[name]
        self._fh.seek(off)
        return self._fh.read(size)

    def __getstate__(self):
        st = dict(self.__dict__); st["_fh"] = None
        return st

    def __len__(self):
        return len(self.stems)

    def __getitem__(self, idx):
        stem = self.stems[idx]
        code = np.load(io.BytesIO(self._read(stem + ".npy")))
        text = self._read(stem + ".txt").decode()
        has_norm = stem + ".normalized.txt" in self.index
        text_norm = self._read(stem + ".normalized.txt").decode() if has_norm else text
        if stem + ".cmu.npy" in self.index:
            ids = np.load(io.BytesIO(self._read(stem + ".cmu.npy"))).tolist()
        else:
            ids = list(self.text_to_ids(text_norm))
        item = {"code": code / 1023, "text": text, "cmu_sequence": intersperse(ids, BLANK_ID),
                "code_length": float(self._read(stem + ".len.txt").decode())}
        if has_norm:
            item["text_norm"] = text_norm
        return item


class DeviceFeeder:
    """Keeps the GPU fed: a background thread pulls collated batches, pins their tensors and copies them to `device` on its
    own stream (`depth` batches ahead); the consumer gets dictionaries whose tensors are already resident and ordered after
    the copy on the current stream.  The reference moves each batch synchronously inside the step loop (train.py:86-90)."""

    def __init__(self, loader, device, depth=2):
        self.loader, self.device, self.depth = loader, torch.device(device), depth

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        import queue
        import threading
        q = queue.Queue(maxsize=self.depth)
        stream = torch.cuda.Stream(device=self.device)
        stop = threading.Event()

        def work():
            try:
                for batch in self.loader:
                    if stop.is_set():
                        return
                    out = {}
                    with torch.cuda.stream(stream):
                        for k, v in batch.items():
                            out[k] = v.pin_memory().to(self.device, non_blocking=True) if torch.is_tensor(v) else v
                        ev = torch.cuda.Event(); ev.record(stream)
                    q.put((out, ev))
                q.put(None)
            except BaseException as e:                                   # surface loader errors in the consumer
                q.put(e)
        th = threading.Thread(target=work, daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                batch, ev = item
                torch.cuda.current_stream(self.device).wait_event(ev)
                for v in batch.values():
                    if torch.is_tensor(v):
                        v.record_stream(torch.cuda.current_stream(self.device))
                yield batch
        finally:
            stop.set()
            while th.is_alive():                                         # unblock a producer waiting on a full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    th.join(timeout=0.05)


class SyntheticDataset(Dataset):
    """LJSpeech-shaped synthetic items: codes U{0..1023}, phoneme ids U{1..147} blank-interspersed."""

    def __init__(self, n_items, n_q, T, max_text=256, seed=1234):
        g = np.random.default_rng(seed)
        self.item_list = []
        for i in range(n_items):
            n_ph = int(g.integers(16, max(17, (max_text - 1) // 2 + 1)))
            self.item_list.append({
                "code": g.integers(0, 1024, (n_q, T)).astype(np.int64) / 1023, "text": f"synthetic {i}",
                "cmu_sequence": intersperse(g.integers(1, 148, n_ph).tolist(), BLANK_ID), "code_length": float(T)})

    def __len__(self):
        return len(self.item_list)

    def __getitem__(self, idx):
        return self.item_list[idx]


class ShardedBatchSampler(torch.utils.data.Sampler):
    """This rank's share of the batches of one epoch, INDEX lists only (nothing is collated for other ranks).

    Same policy as the batch-sampler shard accelerate wraps around the reference's DataLoader (train.py:67-69; no drop_last,
    even batches): the epoch's index order is cut into batches of `batch_size`, rank r takes batches r, r + W, r + 2W, ...;
    a short final batch is completed, and the batch count rounded up to a multiple of W, with indices taken again from the
    START of the epoch's order, so every rank runs the same number of full-size steps (no rank is left alone inside a
    collective at the end of an epoch).  Every rank must draw the same order: seed the sampler's generator identically."""

    def __init__(self, sampler, batch_size, rank=0, world=1):
        if not 0 <= rank < world:
            raise ValueError(f"rank {rank} outside world of {world}")
        self.sampler, self.batch_size, self.rank, self.world = sampler, int(batch_size), int(rank), int(world)

    def __len__(self):
        n_batches = -(-len(self.sampler) // self.batch_size)
        return -(-n_batches // self.world) if n_batches else 0

    def __iter__(self):
        order = list(self.sampler)
        if not order:
            return
        bs, W = self.batch_size, self.world
        n_batches = -(-len(order) // bs)
        n_even = -(-n_batches // W) * W
        need = n_even * bs - len(order)          # indices to borrow from the start of the order (cyclically if tiny)
        k = 0
        while need > 0:
            order.append(order[k]); k += 1; need -= 1
        for b in range(self.rank, n_even, W):
            yield order[b * bs:(b + 1) * bs]


def create_dataloader(data_file, batch_size, max_seq_length, shuffle=False, text_to_ids=None, dataset=None, lazy=False,
                      num_workers=0, rank=0, world=1):
    """Reference signature first (dataloader.py:191-198); `lazy=True` reads utterances on demand from an offset index instead of
    loading the whole tar, `num_workers` collates in background processes; `world` > 1 hands this rank only its own batches
    (ShardedBatchSampler) -- the sharding accelerate.prepare applies to the reference's loader."""
    if dataset is None:
        dataset = (LazySingleSpeakerDataset if lazy else SingleSpeakerDataset)(data_file, text_to_ids)
    collate = TTS_SingleSpkr_Collate_Fn(max_seq_length)
    if world > 1:
        sampler = torch.utils.data.RandomSampler(dataset) if shuffle else torch.utils.data.SequentialSampler(dataset)
        return DataLoader(dataset, batch_sampler=ShardedBatchSampler(sampler, batch_size, rank, world), collate_fn=collate,
                          num_workers=num_workers, persistent_workers=num_workers > 0)
    return DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, collate_fn=collate,
                      num_workers=num_workers, persistent_workers=num_workers > 0)
