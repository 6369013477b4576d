"""Text cleaners (reference: tts/process_text/cleaners.py:17-89): english_cleaners = ASCII transliteration -> lower case ->
numbers spelled out -> abbreviations expanded -> whitespace collapsed.

The reference transliterates with the third-party `unidecode` (not installed here).  `convert_to_ascii` below is exact for
ASCII input (identity, as unidecode is) and, for the rest, decomposes accented Latin letters (NFKD, combining marks dropped)
and maps the common ligatures / typographic punctuation the way unidecode does; characters it does not know are dropped.
"""
import re
import unicodedata

from .numbers import normalize_numbers

_WS = re.compile(r"\s+")
_ABBREVIATIONS = [(re.compile(r"\b%s\." % short, re.IGNORECASE), full) for short, full in (
    ("mrs", "misess"), ("mr", "mister"), ("dr", "doctor"), ("st", "saint"), ("co", "company"), ("jr", "junior"),
    ("maj", "major"), ("gen", "general"), ("drs", "doctors"), ("rev", "reverend"), ("lt", "lieutenant"),
    ("hon", "honorable"), ("sgt", "sergeant"), ("capt", "captain"), ("esq", "esquire"), ("ltd", "limited"),
    ("col", "colonel"), ("ft", "fort"))]
_TRANSLIT = {"ß": "ss", "æ": "ae", "Æ": "AE", "œ": "oe", "Œ": "OE", "ø": "o", "Ø": "O", "đ": "d", "Đ": "D", "ð": "d", "Ð": "D",
             "þ": "th", "Þ": "Th", "ł": "l", "Ł": "L", "ı": "i", "‘": "'", "’": "'", "‚": ",", "“": '"', "”": '"', "„": '"',
             "–": "-", "—": "--", "―": "--", "…": "...", "•": "*", "«": "<<", "»": ">>", "°": "deg", "×": "x", "÷": "/",
             " ": " ", "£": "£"}          # the pound sign must survive: the number pass spells "£5" as "five pounds"


def convert_to_ascii(text):
    if text.isascii():
        return text
    out = []
    for ch in text:
        if ord(ch) < 128:
            out.append(ch)
        elif ch in _TRANSLIT:
            out.append(_TRANSLIT[ch])
        else:
            out.append("".join(c for c in unicodedata.normalize("NFKD", ch) if ord(c) < 128))
    return "".join(out)


def lowercase(text):
    return text.lower()


def expand_numbers(text):
    return normalize_numbers(text)


def expand_abbreviations(text):
    for pattern, full in _ABBREVIATIONS:
        text = pattern.sub(full, text)
    return text


def collapse_whitespace(text):
    return _WS.sub(" ", text)


def basic_cleaners(text):
    return collapse_whitespace(lowercase(text))


def transliteration_cleaners(text):
    return collapse_whitespace(lowercase(convert_to_ascii(text)))


def english_cleaners(text):
    return collapse_whitespace(expand_abbreviations(expand_numbers(lowercase(convert_to_ascii(text)))))
