"""Differential fuzz of the built-in regular-expression engine (fdb_regex_match) against an independent one: random patterns from a
small RE2 grammar (literals incl. non-ASCII, `.`, bracket classes with ranges / negation / Perl / POSIX / \\p classes, groups,
alternation, greedy and lazy quantifiers, anchors, word boundaries, i / s / m flags, scoped flag groups) are matched against random
values by the library and by Python's `re` on the pattern translated with oracle.go_regexp_to_python (test infrastructure; spelling
differences only). Every disagreement is printed; exit status 1 if there was one.

    python tools/regex_diff_fuzz.py [patterns, default 20000] [seed]

Left out on purpose (documented differences between Python and Go that the translation does not hide): the Turkish dotted / dotless
i under (?i) (Python folds İ to i, Go does not), `{,n}` (a literal in RE2, a quantifier in Python), flags set in the middle of a
pattern together with `$` (the translation tracks the m flag for the whole pattern)."""
import os
import random
import re
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frostdb_amd import physicalplan as pp  # noqa: E402
from oracle import go_regexp_to_python  # noqa: E402

warnings.simplefilter("ignore")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
LIT = list("abcxyzABZ019_ -.") + ["é", "É", "σ", "ς", "Σ", "ß", "я", "Я", "K", "k", "K", "\n", "٣", "€"]
ESC = {".": r"\.", "-": r"\-", " ": " ", "\n": r"\n"}
CLASS_ITEMS = ["a-c", "x-z", "A-Z", "0-9", "à-ÿ", "а-я", r"\d", r"\w", r"\s", r"\D", r"\W", "[:alpha:]", "[:digit:]", "[:^space:]", r"\p{Lu}", r"\p{L}", r"\P{Nd}", r"\pN",
               "_", "é", "σ", "k", r"\-", r"\n"]


def lit():
    c = rng.choice(LIT)
    return ESC.get(c, c)


def cls():
    items = [rng.choice(CLASS_ITEMS) for _ in range(rng.randint(1, 3))]
    return "[" + ("^" if rng.random() < 0.3 else "") + "".join(items) + "]"


def atom(depth):
    r = rng.random()
    if r < 0.45:
        return lit()
    if r < 0.55:
        return "."
    if r < 0.70:
        return cls()
    if r < 0.78:
        return rng.choice([r"\d", r"\w", r"\s", r"\D", r"\W", r"\S", r"\p{Lu}", r"\pL", r"\P{L}", r"\p{Nd}"])
    if r < 0.84:
        return rng.choice([r"\b", r"\B", "^", "$", r"\A", r"\z"])
    if depth <= 0:
        return lit()
    kind = rng.random()
    body = expr(depth - 1)
    if kind < 0.5:
        return "(" + body + ")"
    if kind < 0.75:
        return "(?:" + body + ")"
    if kind < 0.9:
        return "(?i:" + body + ")"
    return "(?s:" + body + ")"


def piece(depth):
    a = atom(depth)
    if a in (r"\b", r"\B", "^", "$", r"\A", r"\z"):
        return a
    r = rng.random()
    if r < 0.6:
        return a
    q = rng.choice(["*", "+", "?", "{2}", "{1,3}", "{0,2}", "{2,}", "*?", "+?", "??"])
    return a + q


def expr(depth):
    alts = []
    for _ in range(1 if rng.random() < 0.7 else rng.randint(2, 3)):
        alts.append("".join(piece(depth) for _ in range(rng.randint(1, 4))))
    return "|".join(alts)


def value():
    return "".join(rng.choice(LIT) for _ in range(rng.randint(0, 8)))


bad = checked = compiled = 0
for it in range(n):
    pat = rng.choice(["", "", "(?i)", "(?s)", "(?m)", "(?is)"]) + expr(2)
    try:
        rx = re.compile(go_regexp_to_python(pat.encode()).decode())
    except (re.error, RecursionError, OverflowError):
        rx = None
    vals = [value() for _ in range(4)]
    try:
        got = [pp.regex_match(pat, v.encode()) for v in vals]
    except pp.FdbError:
        got = None
    if rx is None or got is None:
        if (rx is None) != (got is None):
            # one side refuses what the other takes: only report the library refusing (Python is stricter about a few things, e.g.
            # look-behind widths never arise here; quantified anchors do)
            if got is None:
                print("library refuses", repr(pat))
                bad += 1
        continue
    compiled += 1
    for v, g in zip(vals, got):
        w = rx.search(v) is not None
        checked += 1
        if g != w:
            bad += 1
            print("MISMATCH", repr(pat), repr(v), "library", g, "python", w)
print("patterns", n, "compiled by both", compiled, "matches compared", checked, "disagreements", bad)
sys.exit(1 if bad else 0)
