"""Number normalisation of the English cleaner (reference: tts/process_text/numbers.py:8-71).

The reference spells numbers with the third-party `inflect` package (not vendored, not installed here): this module restates
the part of inflect.engine().number_to_words the reference calls --
  number_to_words(n, andword='')                      cardinals, "one hundred twenty-three", "one thousand, two hundred ..."
  number_to_words(n, andword='', zero='oh', group=2)  years read as digit pairs, "nineteen, oh five"
  number_to_words("243rd")                            ordinals with the default 'and': "two hundred and forty-third"
-- pinned by the expected strings of the keithito/tacotron number tests the reference's front-end descends from
(tests/test_text_frontend.py).  Everything else follows the reference's regular expressions and branches.
"""
import re

_UNITS = ["zero", "one", "two", "three", "four", "five", "six", "seven", "eight", "nine", "ten", "eleven", "twelve", "thirteen",
          "fourteen", "fifteen", "sixteen", "seventeen", "eighteen", "nineteen"]
_TENS = ["", "", "twenty", "thirty", "forty", "fifty", "sixty", "seventy", "eighty", "ninety"]
_SCALES = ["", "thousand", "million", "billion", "trillion", "quadrillion", "quintillion", "sextillion", "septillion",
           "octillion", "nonillion", "decillion"]
_ORDINAL_WORD = {"one": "first", "two": "second", "three": "third", "five": "fifth", "eight": "eighth", "nine": "ninth",
                 "twelve": "twelfth"}


def _below_100(n):
    if n < 20:
        return _UNITS[n]
    t, u = divmod(n, 10)
    return _TENS[t] + ("-" + _UNITS[u] if u else "")


def cardinal(n, andword="and"):
    """inflect's plain cardinal: groups of three digits joined by ', '; `andword` between hundreds and the rest, and in
    place of the last comma when the final group has no hundreds ("one thousand and one")."""
    n = int(n)
    if n == 0:
        return "zero"
    groups = []
    i = 0
    while n:
        n, g = divmod(n, 1000)
        if g:
            if i >= len(_SCALES):
                raise ValueError("number too large to spell")
            groups.append((g, i))
        i += 1
    groups.reverse()
    glue = f" {andword} " if andword else " "
    parts = []
    for g, idx in groups:
        h, r = divmod(g, 100)
        words = (f"{_UNITS[h]} hundred" + (glue + _below_100(r) if r else "")) if h else _below_100(r)
        parts.append(words + (" " + _SCALES[idx] if idx else ""))
    if len(parts) > 1 and groups[-1][1] == 0 and groups[-1][0] < 100:
        return ", ".join(parts[:-1]) + glue + parts[-1]
    return ", ".join(parts)


def pairs(n, zero="oh"):
    """inflect group=2: the digit string read in pairs from the left, ', ' between pairs, a leading 0 of a pair as `zero`."""
    digits = str(int(n))
    out = []
    for k in range(0, len(digits), 2):
        chunk = digits[k:k + 2]
        if len(chunk) == 1:
            out.append(_UNITS[int(chunk)] if chunk != "0" else zero)
        elif chunk[0] != "0":
            out.append(_below_100(int(chunk)))
        elif chunk[1] != "0":
            out.append(f"{zero} {_UNITS[int(chunk[1])]}")
        else:
            out.append(f"{zero} {zero}")
    return ", ".join(out)


def ordinal(n):
    words = cardinal(n, andword="and")
    head, sep, last = words.rpartition(" ")
    pre, hyphen, tail = last.rpartition("-")
    if tail in _ORDINAL_WORD:
        tail = _ORDINAL_WORD[tail]
    elif tail.endswith("y"):
        tail = tail[:-1] + "ieth"
    else:
        tail += "th"
    return head + sep + pre + hyphen + tail


_COMMA_NUMBER = re.compile(r"([0-9][0-9\,]+[0-9])")
_DECIMAL = re.compile(r"([0-9]+\.[0-9]+)")
_POUNDS = re.compile(r"£([0-9\,]*[0-9]+)")
_DOLLARS = re.compile(r"\$([0-9\.\,]*[0-9]+)")
_ORDINAL = re.compile(r"[0-9]+(st|nd|rd|th)")
_NUMBER = re.compile(r"[0-9]+")


def _dollars(m):
    text = m.group(1)
    fields = text.split(".")
    if len(fields) > 2:
        return text + " dollars"                      # unexpected format: left for the plain-number pass
    whole = int(fields[0]) if fields[0] else 0
    cents = int(fields[1]) if len(fields) > 1 and fields[1] else 0
    d = f"{whole} {'dollar' if whole == 1 else 'dollars'}"
    c = f"{cents} {'cent' if cents == 1 else 'cents'}"
    if whole and cents:
        return f"{d}, {c}"
    if whole:
        return d
    if cents:
        return c
    return "zero dollars"


def _number(m):
    n = int(m.group(0))
    if 1000 < n < 3000:                                # read like a year
        if n == 2000:
            return "two thousand"
        if 2000 < n < 2010:
            return "two thousand " + cardinal(n % 100)
        if n % 100 == 0:
            return cardinal(n // 100) + " hundred"
        return pairs(n, zero="oh").replace(", ", " ")
    return cardinal(n, andword="")


def normalize_numbers(text):
    text = _COMMA_NUMBER.sub(lambda m: m.group(1).replace(",", ""), text)
    text = _POUNDS.sub(r"\1 pounds", text)
    text = _DOLLARS.sub(_dollars, text)
    text = _DECIMAL.sub(lambda m: m.group(1).replace(".", " point "), text)
    text = _ORDINAL.sub(lambda m: ordinal(re.match(r"[0-9]+", m.group(0)).group(0)), text)
    text = _NUMBER.sub(_number, text)
    return text
