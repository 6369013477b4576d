"""CPU: the text front-end (tts/process_text/*) and the collate (8a-9) against fixtures produced by the
reference's OWN modules (tests/golden/make_text_golden.py: tts/process_text/__init__.py, cmudict.py, symbols.py, cleaners.py
and tts/dataloader.py imported as they lie), plus known answers for the two third-party pieces the reference leans on
(inflect's number speller: the expected strings of the keithito/tacotron number tests; unidecode: ASCII identity)."""
import io
import json
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def fixture():
    return json.load(open(os.path.join(GOLD, "text_frontend.json")))


@pytest.fixture(scope="module")
def dict_path(fixture, tmp_path_factory):
    p = tmp_path_factory.mktemp("cmu") / "cmu_dictionary"
    p.write_text(";;; excerpt of CMUdict 0.7b: the entries the fixture sentences look up\n" + "\n".join(fixture["dictionary_excerpt"]) + "\n",
                 encoding="latin-1")
    return str(p)


def test_symbol_table_is_the_references(fixture):
    from prompt_tts_amd.tts.process_text.symbols import symbols, valid_symbols
    from prompt_tts_amd.tts.dataloader import BLANK_ID
    assert symbols == fixture["symbols"] and len(symbols) == fixture["n_symbols"] == 148 == BLANK_ID and len(valid_symbols) == 84


def test_text_to_sequence_matches_reference_ids(fixture, dict_path):
    from prompt_tts_amd.tts.process_text import CMUDict, cleaners, sequence_to_text, text_to_sequence
    from prompt_tts_amd.tts.dataloader import intersperse
    cmu = CMUDict(dict_path)
    assert len(cmu) >= 30
    n = 0
    for case in fixture["cases"]:
        text = case["text"]
        assert text_to_sequence(text, ["english_cleaners"]) == case["ids_without_dictionary"]          # bit-exact id lists
        if case["cleaned"] is not None:
            assert cleaners.english_cleaners(text) == case["cleaned"]
        if case["ids_with_dictionary"] is None:
            assert text == ""          # the reference indexes sequence[-1] on an empty transcript; here it is an empty list
            assert text_to_sequence(text, ["english_cleaners"], cmu) == []
            continue
        ids = text_to_sequence(text, ["english_cleaners"], cmu)
        assert ids == case["ids_with_dictionary"]
        assert sequence_to_text(ids) == case["round_trip"]
        assert intersperse(ids, 148) == case["cmu_sequence"]
        n += len(ids)
    assert n > 300
    with pytest.raises(Exception):
        text_to_sequence("x", ["no_such_cleaner"])


def test_number_speller_known_answers():
    """Expected strings of the number tests of the Tacotron front-end this code descends from (inflect semantics)."""
    from prompt_tts_amd.tts.process_text.numbers import cardinal, normalize_numbers as nn, ordinal, pairs
    want = {"1": "one", "15": "fifteen", "24": "twenty-four", "100": "one hundred", "101": "one hundred one", "456": "four hundred fifty-six",
            "1000": "one thousand", "1800": "eighteen hundred", "2,000": "two thousand", "3000": "three thousand", "18000": "eighteen thousand",
            "24,000": "twenty-four thousand", "124,001": "one hundred twenty-four thousand one", "6.4 sec": "six point four sec",
            "1st": "first", "2nd": "second", "9th": "ninth", "243rd place": "two hundred and forty-third place",
            "1400": "fourteen hundred", "1901": "nineteen oh one", "1999": "nineteen ninety-nine", "2000": "two thousand",
            "2004": "two thousand four", "2010": "twenty ten", "2012": "twenty twelve", "2025": "twenty twenty-five",
            "September 11, 2001": "September eleven, two thousand one", "July 26, 1984.": "July twenty-six, nineteen eighty-four.",
            "$0.00": "zero dollars", "$1": "one dollar", "$10": "ten dollars", "$.01": "one cent", "$0.25": "twenty-five cents",
            "$5.00": "five dollars", "$5.01": "five dollars, one cent", "$135.99.": "one hundred thirty-five dollars, ninety-nine cents.",
            "$40,000": "forty thousand dollars", "for £2500!": "for twenty-five hundred pounds!"}
    assert {k: nn(k) for k in want} == want
    assert cardinal(1234) == "one thousand, two hundred and thirty-four" and cardinal(1001) == "one thousand and one"
    assert cardinal(1000000) == "one million" and cardinal(0) == "zero" and cardinal(1100, "") == "one thousand, one hundred"
    assert pairs(1905) == "nineteen, oh five" and pairs(1010) == "ten, ten"
    assert [ordinal(n) for n in (3, 5, 8, 12, 20, 21, 40, 100, 101, 1000)] == [
        "third", "fifth", "eighth", "twelfth", "twentieth", "twenty-first", "fortieth", "one hundredth", "one hundred and first",
        "one thousandth"]


def test_english_cleaner_pipeline():
    from prompt_tts_amd.tts.process_text import cleaners as c
    assert c.english_cleaners("Dr.  Smith paid $5.01 on the 3rd\tof May, 1999.") == \
        "doctor smith paid five dollars, one cent on the third of may, nineteen ninety-nine."
    assert c.convert_to_ascii("plain ASCII stays") == "plain ASCII stays"
    assert c.convert_to_ascii("café naïve – “quoted” façade") == 'cafe naive - "quoted" facade'
    assert c.transliteration_cleaners("Ærø  Straße") == "aero strasse" and c.basic_cleaners("A  B") == "a b"


def test_dataset_and_collate_match_the_reference_fixture(dict_path, tmp_path, monkeypatch):
    """The tar the reference's data preparation would write -> SingleSpeakerDataset with the built-in front-end (reference
    4-argument create_dataloader signature, no text_to_ids) -> collate: every tensor bit-exact against what the reference's
    TTS_SingleSpkr_Collate_Fn produced for the same three utterances (truncation at max_seq_length, pad id 0, int32 masks,
    float64 -> float32 code normalisation)."""
    from prompt_tts_amd.tts.dataloader import LazySingleSpeakerDataset, TTS_SingleSpkr_Collate_Fn, create_dataloader
    z = np.load(os.path.join(GOLD, "collate_ref.npz"))
    tar = tmp_path / "three.tar"
    tar.write_bytes(z["tar"].tobytes())
    monkeypatch.setenv("PT_CMUDICT", dict_path)
    L = int(z["max_seq_length"])
    for lazy in (False, True):
        dl = create_dataloader(str(tar), 3, L) if not lazy else create_dataloader(str(tar), 3, L, lazy=True)
        (batch,) = list(dl)
        assert sorted(batch.keys()) == json.loads(str(z["keys"]))
        assert str(batch["code"].dtype) == str(z["code_dtype"]) and str(batch["cmu_sequence_id"].dtype) == str(z["id_dtype"])
        assert str(batch["attention_mask"].dtype) == str(z["mask_dtype"])
        assert np.array_equal(batch["code"].numpy(), z["code"])                          # bit-exact f32
        assert np.array_equal(batch["cmu_sequence_id"].numpy(), z["cmu_sequence_id"])    # bit-exact ids, truncated / padded
        assert np.array_equal(batch["attention_mask"].numpy(), z["attention_mask"])
        assert batch["cmu_sequence"] == json.loads(str(z["cmu_sequence"])) and batch["text"] == json.loads(str(z["texts"]))
        assert np.array_equal(np.array(batch["code_length"]), z["code_length"])
    assert int(z["attention_mask"][1].sum()) == L and int(z["attention_mask"][2].sum()) == 3     # truncated row, 1-symbol row
    # without a dictionary anywhere the dataset says what is missing instead of guessing
    monkeypatch.delenv("PT_CMUDICT")
    monkeypatch.chdir(tmp_path)
    with pytest.raises(FileNotFoundError):
        create_dataloader(str(tar), 3, L)
