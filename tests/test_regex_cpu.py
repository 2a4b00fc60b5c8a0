"""The library's built-in regular-expression engine (fdb_regex_match; what `=~` / `!~` use when the host passes no matcher): RE2
syntax as Go's regexp compiles it (filter.go:105-124), matched unanchored like regexp.Regexp.Match (regexpfilter.go:84-166).

The reference holds regex vectors only for plain patterns (logictest/testdata/exec/filter/filter:165-191: those run through the
oracle in test_oracle_golden.py); what is checked here is the ENGINE, against an independent one: Python's `re` on the pattern
translated by oracle.go_regexp_to_python (spelling differences only). Go-specific behaviour that Python does not share is listed
explicitly with the expected answer and the place in Go's documentation it comes from."""
import re
import time

import pytest


@pytest.fixture(scope="module")
def pp():
    from frostdb_amd import physicalplan
    return physicalplan


CASES = [
    ("", [b"", b"a"]), ("abc", [b"abc", b"xabcx", b"ab", b""]), ("^abc$", [b"abc", b"abcd", b"xabc"]),
    ("a.c", [b"abc", b"a\nc", b"ac", "aéc".encode()]), ("(?s)a.c", [b"a\nc", b"abc"]), ("a|b|", [b"", b"c"]), ("a*", [b"", b"bbb"]),
    ("a+b?c{2,3}", [b"aacc", b"abccc", b"ac", b"abcccc", b"xaccx"]), ("[a-c]+[^a-c]", [b"abcd", b"abc", b"d"]), ("[]a]", [b"]", b"a", b"b"]),
    ("[^]a]", [b"]", b"b", b"a"]), (r"[a\]b]", [b"]", b"c"]), (r"\d+\.\d+", [b"3.14", b"314", b"a1.5b"]), (r"\w+@\w+\.com", [b"me@x.com", b"me@x.org"]),
    (r"\bfoo\b", [b"foo", b"a foo b", b"foobar", b"xfoo"]), (r"\Bfoo", [b"xfoo", b"foo"]), ("(?i)hello", [b"HeLLo", b"help"]),
    ("(?i:he)llo", [b"HEllo", b"HELLO"]), ("(?i)[a-c]x", [b"Bx", b"BX", b"dx"]), (r"(?P<year>\d{4})-(?P<m>\d\d)", [b"2024-05", b"24-05"]),
    (r"(?<year>\d{4})", [b"in 2024", b"24"]), ("(?:ab)+c", [b"ababc", b"ac"]), ("(a|b)*c", [b"ababc", b"c", b"ab"]), ("(a*)*b", [b"aaab", b"b", b"aaa"]),
    ("(a*)+$", [b"aaa", b"aab"]), ("x{0}", [b"", b"x"]), ("x{2}", [b"x", b"xx"]), ("x{2,}", [b"x", b"xxxxx"]), ("a{b", [b"a{b", b"ab"]),
    (r"\.\*\+\?\(\)\[\]\{\}\|\^\$\\", [b".*+?()[]{}|^$\\", b"x"]), ("(?m)^b$", [b"a\nb\nc", b"ab"]), ("^b$", [b"a\nb\nc", b"b"]),
    (r"\Aab\z", [b"ab", b"ab\n", b"xab"]), ("é+", ["éé".encode(), b"e"]), ("[à-ÿ]", ["é".encode(), b"e"]), ("/api/v[12]/.*(users|orders)$", [b"/api/v1/x/users", b"/api/v3/users", b"/api/v2/orders/1"]),
    ("^(GET|POST) ", [b"GET /", b"PUT /"]), ("(?i)^get$", [b"GeT", b"gets"]), ("a.*b.*c", [b"a123b456c", b"acb"]), ("(ab|a)(bc|c)?$", [b"abc", b"ac", b"abcc"]),
    ("[[:alpha:]]+[[:digit:]]", [b"abc1", b"1", b"abc"]), (r"[\d\s]+x", [b"1 2x", b"x"]), ("[a-]", [b"-", b"b"]), ("[-a]", [b"-"]), (r"\Qa.b*\E+", [b"a.b**", b"a.b", b"axb"]),
    ("a??b", [b"b", b"ab"]), ("a*?b", [b"aab"]), ("(?U)a+b", [b"aab", b"b"]), ("(a|ab)(c|bcd)(d*)", [b"abcd", b"ad"]), ("[^\\x00-\\x7f]", ["€".encode(), b"abc"]),
    ("(?i)STRASSE|strasse", [b"Strasse", b"street"]),
    # Perl classes and word boundaries are ASCII-only in RE2 whatever the text (the translation spells them out for Python)
    (r"^\d+$", [b"12", "٣".encode()]), (r"^\w+$", [b"ab_1", "é".encode()]), (r"\bfoo\b", ["éfoo".encode(), b"xfoo", b"a foo"]), (r"^[\D]+$", ["ab٣".encode(), b"a1"]),
    (r"^\S+$", [b"a b", "a\u00a0b".encode()]), (r"^[^\w]+$", ["é-".encode(), b"a-"]), (r"\Bfoo", ["éfoo".encode(), b"xfoo"]), (r"^\s$", [b"\x0b", b" ", b"\t"]),
    (r"^[\W\d]+$", ["é1".encode(), b"a1"]), (r"(?i)^\w+$", ["\u212a\u017f".encode(), "é".encode()]),
    # Unicode general categories (the independent side spells them out from unicodedata) and case folding beyond ASCII
    (r"^\p{Lu}\p{Ll}+$", ["Éclair".encode(), b"eclair", "ÉCLAIR".encode()]), (r"^[\p{Nd}x]+$", ["٣x4".encode(), b"3y"]), (r"^\PL+$", [b"123 ", b"12a"]),
    (r"\p{Sc}\d+", ["€5".encode(), b"$7", b"5"]), (r"^\p{^Zs}+$", [b"abc", "a\u00a0b".encode()]), (r"^[^\p{L}\p{N}]+$", [b"-+!", b"-a-"]), (r"^\pZ$", ["\u2003".encode(), b"x"]),
    (r"^\p{Any}{2}$", ["😀x".encode(), b"x"]), ("(?i)^σοφός$", ["ΣΟΦΌΣ".encode(), "σοφόσ".encode()]), ("(?i)^привет$", ["ПРИВЕТ".encode(), "ПРИВЕД".encode()]),
    ("(?i)^[à-þ]+$", ["ÀÉÎ".encode(), "àéî".encode(), b"aei"]), ("(?i)^k$", ["\u212a".encode(), b"K", b"c"]), (r"(?i)^\p{Lu}+$", [b"abc", b"ab1"]), ("node-[0-9]+\\.(eu|us)-(east|west)-[1-3]$", [b"node-12.eu-west-2", b"node-12.eu-west-4", b"node-.us-east-1"]),
]


def test_the_builtin_engine_agrees_with_an_independent_engine_on_re2_syntax(pp):
    from oracle import go_regexp_to_python
    n = 0
    for pat, values in CASES:
        rx = re.compile(go_regexp_to_python(pat.encode()).decode())
        for v in values:
            want = rx.search(v.decode("utf-8", "replace")) is not None
            assert pp.regex_match(pat, v) == want, (pat, v, want)
            n += 1
    assert n > 170


def test_go_specific_semantics(pp):
    # `$` without the m flag is the end of the TEXT, not "before a trailing newline" (regexp/syntax: "at end of text (like \z not \Z)")
    assert pp.regex_match("^abc$", b"abc\n") is False and pp.regex_match("(?m)^abc$", b"abc\n") is True
    # `.` never matches a newline unless s is set, and sees RUNES: one invalid byte is one U+FFFD (utf8.DecodeRune)
    assert pp.regex_match("^.$", b"\xff") is True and pp.regex_match("^.$", "é".encode()) is True and pp.regex_match("^.$", b"\xc3") is True
    assert pp.regex_match("^..$", "é".encode()) is False and pp.regex_match("^[^a]$", "€".encode()) is True
    # `{` that does not open a valid repetition is a literal; `{,n}` is not a repetition in RE2
    assert pp.regex_match("a{,2}", b"a{,2}") is True and pp.regex_match("^a{,2}$", b"aa") is False
    # flags apply to the rest of the enclosing group; a negated flag switches back
    assert pp.regex_match("^a(?i)b$", b"aB") is True and pp.regex_match("^a(?i)b$", b"AB") is False
    assert pp.regex_match("^(?i)a(?-i)b$", b"Ab") is True and pp.regex_match("^(?i)a(?-i)b$", b"AB") is False
    assert pp.regex_match("^((?i)a)b$", b"Ab") is True and pp.regex_match("^((?i)a)b$", b"AB") is False
    # octal and hex escapes
    assert pp.regex_match(r"^\x41\x{263A}\101\0\012$", "A☺A\x00\n".encode()) is True
    # the empty pattern matches everything (what regexpfilter.go:23-33 asks about a missing column)
    assert pp.regex_match("", b"") is True and pp.regex_match("x*", b"") is True and pp.regex_match("x+", b"") is False


@pytest.mark.parametrize("pat,why", [
    (r"\p{Greek}", "not scripts"), (r"\p{Cn}", "invalid character class range"), (r"\p{", "invalid character class range"), ("a**", "invalid nested repetition operator"), ("(a", "missing closing )"), ("a)", "unexpected )"), ("[a", "missing closing ]"),
    (r"\1", "invalid escape sequence"), ("(?=a)", "invalid or unsupported Perl syntax"), ("(?<!a)b", "invalid"), ("a{1001}", "invalid repeat count"),
    (r"\C", "invalid escape sequence"), (r"\q", "invalid escape sequence"), ("x{3,2}", "invalid repeat count"), ("*a", "missing argument to repetition operator"),
    ("(?i", "missing closing )"), ("[b-a]", "invalid character class range"), ("a\\", "trailing backslash"), ("(?P<n-1>a)", "invalid named capture"),
    ("((((a{1000}){1000}){1000}){1000})", "too large"),
])
def test_what_re2_rejects_is_rejected_with_go_style_wording(pp, pat, why):
    """No backreferences, no look-around, bounded repeat counts and program size: like RE2, so that no pattern can make matching
    super-linear. A pattern that does not compile is FDB_ERR_INVALID — at fdb_plan_create too (regexp.Compile at plan build)."""
    from frostdb_amd.logicalplan import Col
    with pytest.raises(pp.FdbError) as e:
        pp.regex_match(pat, b"x")
    assert e.value.code == pp.FDB_ERR_INVALID and "error parsing regexp" in str(e.value) and why in str(e.value), str(e.value)
    with pytest.raises(pp.FdbError) as e:
        pp.explain(Col("labels.x").RegexMatch(pat))
    assert e.value.code == pp.FDB_ERR_INVALID


def test_unicode_general_categories_against_unicodedata(pp):
    """\\p{..} general categories (what Go's regexp takes from unicode.Categories): every two-letter category and the one-letter
    groups, positive and negated, inside and outside brackets, checked rune by rune against this interpreter's unicodedata on a
    sample that covers every plane with assigned characters."""
    import random
    import unicodedata
    rng = random.Random(7)
    sample = list(range(0, 0x3000, 7)) + [rng.randrange(0x3000, 0x30000) for _ in range(3000)] + [0xE000, 0xF0000, 0x10FFFF, 0xE0001]
    sample = [c for c in sample if not 0xD800 <= c <= 0xDFFF]
    cats = sorted({unicodedata.category(chr(c)) for c in range(0x110000) if not 0xD800 <= c <= 0xDFFF} - {"Cn"})
    assert "Lu" in cats and "Zs" in cats and "Co" in cats
    for name in cats + ["L", "M", "N", "P", "S", "Z", "C"]:
        brace = "{%s}" % name
        for c in sample[::5] if len(name) == 2 else sample[::9]:
            cat = unicodedata.category(chr(c))
            want = cat == name if len(name) == 2 else (cat[0] == name and cat != "Cn")
            v = chr(c).encode()
            assert pp.regex_match("^\\p" + brace + "$", v) is want, (name, hex(c), cat)
            assert pp.regex_match("^\\P" + brace + "$", v) is (not want), (name, hex(c))
    for c in sample[::11]:
        cat, v = unicodedata.category(chr(c)), chr(c).encode()
        assert pp.regex_match(r"^\pL$", v) is (cat[0] == "L")
        assert pp.regex_match(r"^[\p{Nd}\p{Lu}_]$", v) is (cat in ("Nd", "Lu") or c == 0x5F)
        assert pp.regex_match(r"^[^\p{L}\p{N}]$", v) is (cat[0] not in "LN")
        assert pp.regex_match(r"^\p{^Zs}$", v) is (cat != "Zs")
        assert pp.regex_match(r"^\p{Any}$", v) is True


def test_case_folding_follows_orbits_not_ascii(pp):
    """(?i) folds the way regexp/syntax does with unicode.SimpleFold: a rune matches every rune of its case-folding orbit."""
    yes = [("(?i)^k$", "K"), ("(?i)^k$", "\u212a"), ("(?i)^\u212a$", "k"), ("(?i)^σ$", "ς"), ("(?i)^ς$", "Σ"), ("(?i)^straße$", "STRAẞE"),
           ("(?i)^ǆ$", "ǅ"), ("(?i)^µ$", "Μ"), ("(?i)^ÀÉÎ$", "àéî"), ("(?i)^[а-я]+$", "ПРИВЕТ"), ("(?i)^\\p{Lu}+$", "abc"), ("(?i)^[^a]$", "b"),
           ("(?i)^ſ$", "S")]
    no = [("(?i)^i$", "İ"), ("(?i)^I$", "ı"), ("^k$", "K"), ("(?i)^[^k]$", "\u212a"), ("(?i)^ß$", "ss"), ("(?i)^[^a]$", "A")]
    for pat, val in yes:
        assert pp.regex_match(pat, val.encode()) is True, (pat, val)
    for pat, val in no:
        assert pp.regex_match(pat, val.encode()) is False, (pat, val)
    # every cased letter of the BMP against its own lower / upper forms, where those are single runes with the same simple folding
    for c in range(0x80, 0x2000):
        ch = chr(c)
        for other in {ch.lower(), ch.upper()}:
            if len(other) == 1 and other != ch and other.casefold() == ch.casefold() and len(ch.casefold()) == 1:
                assert pp.regex_match("(?i)^" + re.escape(ch) + "$", other.encode()) is True, (hex(c), other)


def test_negated_named_classes_fold_before_they_negate(pp):
    """regexp/syntax folds the POSITIVE class and negates afterwards (parser.appendGroup, parseUnicodeClass; parse_test.go:
    `(?i)\\W` → cc{0x0-0x2f 0x3a-0x40 0x5b-0x5e 0x60 0x7b-0x17e 0x180-0x2129 0x212b-0x10ffff}, i.e. WITHOUT U+017F and U+212A):
    negating first would let the folding pull k / K / s / S back into the negated set."""
    from oracle import go_regexp_to_python
    no = [(r"(?i)^\W$", "k"), (r"(?i)^\W$", "K"), (r"(?i)^\W$", "s"), (r"(?i)^\W$", "S"), (r"(?i)^\W$", "\u212a"), (r"(?i)^\W$", "\u017f"),
          (r"(?i)^[[:^alpha:]]$", "k"), (r"(?i)^[[:^alpha:]]$", "\u212a"), (r"(?i)^[\W]$", "s"), (r"(?i)^[^\W]$", "-"), (r"(?i)^\P{Lu}$", "a"),
          (r"(?i)^\p{^Lu}$", "a"), (r"(?i)^[\P{Ll}]$", "A"), (r"(?i)^[[:^upper:]]$", "a"), (r"(?i)^[[:^lower:]x]$", "Q"), (r"^\W$", "k"),
          (r"(?i)^[^\W\d]$", "7")]
    yes = [(r"(?i)^\W$", "-"), (r"(?i)^\W$", "é"), (r"^\W$", "\u212a"), (r"^\W$", "\u017f"), (r"^[[:^alpha:]]$", "\u212a"), (r"(?i)^[^\W]$", "\u212a"),
           (r"(?i)^[^\W]$", "k"), (r"(?i)^\D$", "x"), (r"(?i)^\S$", "x"), (r"(?i)^\P{Lu}$", "1"), (r"(?i)^[[:^alpha:]x]$", "X"), (r"(?i)^[[:^alpha:]]$", "1"),
           (r"(?i)^[^\W\d]$", "\u017f")]
    for want, cases in ((False, no), (True, yes)):
        for pat, val in cases:
            assert pp.regex_match(pat, val.encode()) is want, (pat, val, want)
            rx = re.compile(go_regexp_to_python(pat.encode()).decode())  # the oracle's engine agrees (it had the same flaw)
            assert (rx.search(val) is not None) is want, ("oracle", pat, val, want)


def test_patterns_go_refuses(pp):
    """regexp.Compile errors the built-in engine used to let through: a capture name used twice, and pattern bytes that are not UTF-8
    (overlong forms, surrogates, runes past U+10FFFF)."""
    for pat in (b"(?P<n>a)(?P<n>b)", b"(?<n>a)|(?P<n>b)", b"a\xc0\xafb", b"\xed\xa0\x80", b"\xf4\x90\x80\x80", b"[\xe0\x80\xaf]"):
        with pytest.raises(pp.FdbError) as e:
            pp.regex_match(pat, b"x")
        assert e.value.code == pp.FDB_ERR_INVALID, pat
    assert pp.regex_match("(?P<a>x)(?P<b>x)", b"xx") is True


def test_matching_is_linear_in_the_value(pp):
    """The classic backtracking bombs: (a+)+$ / (a|aa)*$ / (.*)*x over a long run of a's end in milliseconds (a backtracking
    engine needs 2^n steps); a 1 MB value against a 60-state pattern in well under a second."""
    bomb = b"a" * 5000 + b"!"
    t0 = time.perf_counter()
    for pat in ("(a+)+$", "^(a|aa)*$", "(.*)*x", "(a*)*b", "^(a?){40}a{40}$"):
        assert pp.regex_match(pat, bomb) is False, pat
    assert pp.regex_match("(x+x+)+y", b"x" * 100000) is False
    assert pp.regex_match(r"[a-z]+\d{3}-(foo|bar|baz)+[^!]*!$", b"q" * 500000 + b"123-foobarbaz" + b"." * 500000 + b"!") is True
    assert time.perf_counter() - t0 < 20.0
