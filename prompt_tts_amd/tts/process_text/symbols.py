"""Symbol inventory of the text front-end (reference: tts/process_text/symbols.py:8-18, cmudict.py:6-16).

ids are positions in `symbols`: pad '_' (0), '-', the punctuation "!'(),.:;? ", A-Z, a-z, then the 84 ARPAbet phonemes
(stress-marked vowels included) prefixed with '@'.  len(symbols) == 148 is the blank id the dataset intersperses
(tts/dataloader.py:52-55) and cmu_vocab_len = 149 in the 1d_config.
"""
_VOWELS = ("AA", "AE", "AH", "AO", "AW", "AY", "EH", "ER", "EY", "IH", "IY", "OW", "OY", "UH", "UW")
_CONSONANTS = ("B", "CH", "D", "DH", "F", "G", "HH", "JH", "K", "L", "M", "N", "NG", "P", "R", "S", "SH", "T", "TH", "V", "W", "Y",
               "Z", "ZH")


def _arpabet():
    # alphabetical order of the base phoneme; a vowel is followed by its three stress variants
    out = []
    for base in sorted(_VOWELS + _CONSONANTS):
        out.append(base)
        if base in _VOWELS:
            out += [base + s for s in "012"]
    return out


valid_symbols = _arpabet()
PAD, SPECIAL, PUNCTUATION = "_", "-", "!'(),.:;? "
LETTERS = "ABCDEFGHIJKLMNOPQRSTUVWXYZ" + "abcdefghijklmnopqrstuvwxyz"
symbols = [PAD] + list(SPECIAL) + list(PUNCTUATION) + list(LETTERS) + ["@" + s for s in valid_symbols]
symbol_to_id = {s: i for i, s in enumerate(symbols)}
