"""Generate tests/golden/text_frontend.json and tests/golden/collate_ref.npz by running the reference's own text front-end and
collate function in the authoring container.

Run from the repo root:   python tests/golden/make_text_golden.py
Needs /root/reference (absent on the GPU box -> never imported by tests; only the fixtures travel).

The reference's files are imported as they lie (tts/process_text/*.py, tts/dataloader.py).  Three third-party packages they
import are not installed and are supplied as inert stand-ins:
  * `unidecode`  -> identity.  Every sentence below is ASCII, for which unidecode is the identity.
  * `inflect`    -> an engine whose number_to_words raises.  Every sentence below is digit-free, so it is never called.
  * `torchvision.transforms.Normalize` -> (x - mean) / std over the channel dimension, torchvision's documented formula
    (the reference uses it only as Normalize([0.5], [0.5]) at tts/dataloader.py:143).
So the fixtures pin the reference's symbol table, dictionary lookup, brace handling, token / space logic, blank interspersing,
padding, truncation, masks and dtypes; the number speller and non-ASCII transliteration are pinned by known-answer tests instead
(tests/test_text_frontend.py).  The dictionary excerpt holds the CMUdict lines of exactly the words the sentences look up
(data, CMU's BSD-style licence), so the tests do not need the 3.7 MB dictionary.
"""
import io
import json
import os
import sys
import tarfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

SENTENCES = [
    "Printing, in the only sense with which we are at present concerned, differs from most if not from all the arts.",
    "The quick brown fox jumps over the lazy dog!",
    "Mr. Smith and Dr. Jones met Mrs. Brown at St. James's church; it was raining.",
    "Turn left on {HH AW1 S T AH0 N} Street, then walk to the zzyzx gate.",
    "Hello?  Is   anybody (really) there: yes -- no.",
    "a",
    "",
]


def _shims():
    u = types.ModuleType("unidecode"); u.unidecode = lambda s: s; sys.modules["unidecode"] = u
    inf = types.ModuleType("inflect")

    class _Engine:
        def number_to_words(self, *a, **k):
            raise RuntimeError("inflect stand-in: the fixture sentences are digit-free")
    inf.engine = _Engine; sys.modules["inflect"] = inf
    tv = types.ModuleType("torchvision"); tvt = types.ModuleType("torchvision.transforms")

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = torch.tensor(mean, dtype=torch.float32), torch.tensor(std, dtype=torch.float32)

        def __call__(self, t):
            return (t - self.mean.view(-1, 1, 1)) / self.std.view(-1, 1, 1)
    tvt.Normalize = Normalize; tv.transforms = tvt
    sys.modules["torchvision"] = tv; sys.modules["torchvision.transforms"] = tvt


def main():
    _shims()
    sys.path.insert(0, REF)
    from tts.process_text import cmudict, sequence_to_text, text_to_sequence
    from tts.process_text.symbols import symbols
    from tts.process_text import cleaners
    import tts.dataloader as rd
    dict_path = os.path.join(REF, "tts", "process_text", "cmu_dictionary")
    cmu = cmudict.CMUDict(dict_path)
    out = {"n_symbols": len(symbols), "symbols": symbols, "n_dictionary_words": len(cmu), "cases": []}
    looked_up = set()
    plain_lookup = cmu.lookup

    def recording_lookup(word):
        looked_up.add(word.upper())
        return plain_lookup(word)
    cmu.lookup = recording_lookup
    for s in SENTENCES:
        ids_dict = text_to_sequence(s, ["english_cleaners"], cmu) if s else None   # the reference indexes [-1]: empty text raises
        ids_plain = text_to_sequence(s, ["english_cleaners"])
        cleaned = cleaners.english_cleaners(s) if "{" not in s else None      # ARPAbet in braces carries stress digits
        out["cases"].append({"text": s, "cleaned": cleaned, "ids_with_dictionary": ids_dict, "ids_without_dictionary": ids_plain,
                             "round_trip": sequence_to_text(ids_dict) if ids_dict is not None else None,
                             "cmu_sequence": rd.intersperse(ids_dict, len(symbols)) if ids_dict is not None else None})
    # dictionary excerpt: every line of the dictionary file whose head word (alternates included) was looked up
    excerpt = []
    with open(dict_path, encoding="latin-1") as fh:
        for line in fh:
            head = line.split("  ")[0]
            base = head.split("(")[0]
            if base and base in looked_up and ("A" <= line[0] <= "Z" or line[0] == "'"):
                excerpt.append(line.rstrip("\n"))
    out["dictionary_excerpt"] = excerpt
    with open(os.path.join(HERE, "text_frontend.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(f"text_frontend.json: {len(out['cases'])} cases, {len(excerpt)} dictionary lines, {len(cmu)} words in the full dictionary")

    # ---- collate fixture: the reference's TTS_SingleSpkr_Collate_Fn on a 3-utterance synthetic tar ----
    rng = np.random.default_rng(2024)
    texts = [SENTENCES[1], SENTENCES[0], SENTENCES[5]]                 # medium, long (truncated at 48), one symbol
    codes = [rng.integers(0, 1024, (8, 37)).astype(np.int64) for _ in texts]
    lens = [37.0, 31.0, 5.0]
    items = []
    for t, c, n in zip(texts, codes, lens):
        item = {"code": c / 1023, "text": t, "cmu_sequence": rd.intersperse(text_to_sequence(t, ["english_cleaners"], cmu), len(symbols)),
                "code_length": n}
        items.append(item)
    got = rd.TTS_SingleSpkr_Collate_Fn(48)(items)
    items_norm = [dict(it, text_norm=it["text"].upper()) for it in items]
    got_norm = rd.TTS_SingleSpkr_Collate_Fn(48)(items_norm)
    # the same three utterances as the tar the reference's data preparation writes (<utt>.npy, .txt, .len.txt)
    buf = io.BytesIO()
    with tarfile.open(fileobj=buf, mode="w") as tf:
        def add(name, data):
            ti = tarfile.TarInfo(name); ti.size = len(data); tf.addfile(ti, io.BytesIO(data))
        for i, (t, c, n) in enumerate(zip(texts, codes, lens)):
            b = io.BytesIO(); np.save(b, c); add(f"utt{i}.npy", b.getvalue())
            add(f"utt{i}.txt", t.encode()); add(f"utt{i}.len.txt", str(n).encode())
    np.savez_compressed(
        os.path.join(HERE, "collate_ref.npz"), tar=np.frombuffer(buf.getvalue(), dtype=np.uint8), max_seq_length=48,
        code=got["code"].numpy(), cmu_sequence_id=got["cmu_sequence_id"].numpy(), attention_mask=got["attention_mask"].numpy(),
        code_dtype=str(got["code"].dtype), id_dtype=str(got["cmu_sequence_id"].dtype), mask_dtype=str(got["attention_mask"].dtype),
        keys=json.dumps(sorted(got.keys())), keys_with_norm=json.dumps(sorted(got_norm.keys())),
        code_length=np.array(got["code_length"]), texts=json.dumps(got["text"]),
        cmu_sequence=json.dumps(got["cmu_sequence"]))
    print("collate_ref.npz:", {k: tuple(v.shape) for k, v in got.items() if torch.is_tensor(v)}, got["code"].dtype,
          got["cmu_sequence_id"].dtype)


if __name__ == "__main__":
    main()
