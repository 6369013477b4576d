"""CMU pronouncing dictionary reader (reference: tts/process_text/cmudict.py:19-64).

File format: one entry per line, `WORD  PH ON EM ES` (two spaces after the head word); alternate pronunciations are
`WORD(1)  ...`; comment lines start with `;;;`.  Only head words that begin with A-Z or an apostrophe are kept, and only
pronunciations made entirely of the 84 ARPAbet symbols.  `lookup` upper-cases the query and returns the list of
pronunciations (strings) or None, as the reference does.
"""
import re

from .symbols import valid_symbols

_VALID = frozenset(valid_symbols)
_ALT = re.compile(r"\([0-9]+\)")


def _entries(lines):
    table = {}
    for line in lines:
        if not line or not ("A" <= line[0] <= "Z" or line[0] == "'"):
            continue
        fields = line.split("  ")
        if len(fields) < 2:
            continue
        phones = fields[1].strip().split(" ")
        if any(ph not in _VALID for ph in phones):
            continue
        table.setdefault(_ALT.sub("", fields[0]), []).append(" ".join(phones))
    return table


class CMUDict:
    def __init__(self, file_or_path, keep_ambiguous=True):
        if isinstance(file_or_path, str):
            with open(file_or_path, encoding="latin-1") as fh:
                table = _entries(fh)
        else:
            table = _entries(file_or_path)
        if not keep_ambiguous:
            table = {w: p for w, p in table.items() if len(p) == 1}
        self._entries = table

    def __len__(self):
        return len(self._entries)

    def lookup(self, word):
        return self._entries.get(word.upper())
