"""Text -> phoneme/character id sequence (reference: tts/process_text/__init__.py:13-100), the front-end behind
SingleSpeakerDataset (tts/dataloader.py:52-55).

text_to_sequence(text, cleaner_names, dictionary): text inside {curly braces} is taken as ARPAbet; the rest is cleaned and,
when a CMU dictionary is given, split on single spaces -- a token found in the dictionary (punctuation attached to a word
makes it a miss, as in the reference) becomes its first pronunciation, any other token is spelled character by character --
with a space id after every token and the trailing one removed.  Characters outside the symbol table, '_' and '~' are
dropped.
"""
import os
import re

from . import cleaners
from .cmudict import CMUDict
from .symbols import symbol_to_id, symbols

_ID_TO_SYMBOL = dict(enumerate(symbols))
_BRACES = re.compile(r"(.*?)\{(.+?)\}(.*)")
_SPACE = symbol_to_id[" "]


def _chars(text):
    return [symbol_to_id[c] for c in text if c in symbol_to_id and c not in "_~"]


def _arpabet(text):
    return [symbol_to_id["@" + ph] for ph in text.split() if "@" + ph in symbol_to_id]


def _clean(text, cleaner_names):
    for name in cleaner_names:
        fn = getattr(cleaners, name, None)
        if fn is None:
            raise Exception("Unknown cleaner: %s" % name)
        text = fn(text)
    return text


def get_arpabet(word, dictionary):
    found = dictionary.lookup(word)
    return "{" + found[0] + "}" if found is not None else word


def text_to_sequence(text, cleaner_names, dictionary=None):
    seq = []
    while text:
        m = _BRACES.match(text)
        if m is None:
            cleaned = _clean(text, cleaner_names)
            if dictionary is None:
                seq += _chars(cleaned)
            else:
                for token in cleaned.split(" "):
                    token = get_arpabet(token, dictionary)
                    seq += _arpabet(token[1:-1]) if token.startswith("{") else _chars(token)
                    seq.append(_SPACE)
            break
        seq += _chars(_clean(m.group(1), cleaner_names))
        seq += _arpabet(m.group(2))
        text = m.group(3)
    if dictionary is not None and seq and seq[-1] == _SPACE:
        seq.pop()
    return seq


def sequence_to_text(sequence):
    out = ""
    for i in sequence:
        s = _ID_TO_SYMBOL.get(i)
        if s is not None:
            out += "{%s}" % s[1:] if len(s) > 1 and s[0] == "@" else s
    return out.replace("}{", " ")


# ---- locating the dictionary ---------------------------------------------------------------------------------------------
_DICT_CACHE = {}


def find_cmu_dictionary():
    """The reference opens <repo>/tts/process_text/cmu_dictionary (tts/dataloader.py:21-22).  Looked for, in order: the
    PT_CMUDICT environment variable, ./tts/process_text/cmu_dictionary under the working directory (a checkout of the
    reference), and a copy placed next to this file."""
    cands = [os.environ.get("PT_CMUDICT"), os.path.join(os.getcwd(), "tts", "process_text", "cmu_dictionary"),
             os.path.join(os.path.dirname(os.path.abspath(__file__)), "cmu_dictionary")]
    for c in cands:
        if c and os.path.isfile(c):
            return c
    raise FileNotFoundError(
        "CMU pronouncing dictionary not found: set PT_CMUDICT=/path/to/cmu_dictionary (the reference ships it as "
        "tts/process_text/cmu_dictionary), or run from a directory that holds tts/process_text/cmu_dictionary")


def default_text_to_ids(dictionary_path=None):
    """The callable SingleSpeakerDataset applies to each transcript: english_cleaners + CMUdict lookup (dataloader.py:52-53)."""
    path = dictionary_path or find_cmu_dictionary()
    d = _DICT_CACHE.get(path)
    if d is None:
        d = _DICT_CACHE[path] = CMUDict(path)
    return lambda text: text_to_sequence(text, ["english_cleaners"], d)
