"""Short, seeded runs of the three host-side fuzzers (tools/desc_fuzz.py, tools/arrow_fuzz.py, tools/asan_parquet_run.py) against the
library as built — in child processes with a timeout, so that a crash or an endless loop is a failed test, not a dead test session.
The long campaigns (and the ASan / UBSan builds) are tools/asan_full.sh's job; this keeps the three bugs they found from coming back."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_tool(args, env=None, timeout=240):
    from frostdb_amd import build
    build.build()
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([sys.executable] + args, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, f"{args}: rc {p.returncode}\n{p.stdout[-2000:]}\n{p.stderr[-2000:]}"
    return p.stdout


@pytest.mark.timeout(300)
def test_random_plan_descriptors_never_crash_explain():
    out = run_tool([os.path.join(ROOT, "tools", "desc_fuzz.py"), "2500", "3"])
    assert "codes" in out


@pytest.mark.timeout(300)
def test_arrow_records_with_detectable_defects_come_back_as_error_codes():
    out = run_tool([os.path.join(ROOT, "tools", "arrow_fuzz.py"), "1200", "3"])
    assert "codes" in out


@pytest.mark.timeout(300)
def test_mutated_parquet_chunks_are_refused_or_parsed_never_fatal():
    lib = os.path.join(ROOT, "frostdb_amd", "libfrostdb_amd.so")
    out = run_tool([os.path.join(ROOT, "tools", "asan_parquet_run.py"), "25", "3"], env={"FDB_ASAN_LIB": lib})
    # every file variant of the tool (codecs, page versions, DELTA byte-array encodings, literal pages) × 25 mutations: the count
    # comes from the tool's own variant list, so adding a variant there cannot turn this test red
    m = re.search(r"variants (\d+) runs (\d+)", out)
    assert m, out[-500:]
    assert int(m.group(1)) >= 14 and int(m.group(2)) == 25 * int(m.group(1)), out[-500:]


@pytest.mark.timeout(300)
def test_random_regex_patterns_compile_or_are_refused_and_never_take_long():
    out = run_tool([os.path.join(ROOT, "tools", "regex_fuzz.py"), "6000", "5"])
    assert "runs 6000" in out and "slow 0" in out


@pytest.mark.timeout(300)
def test_random_re2_patterns_match_like_an_independent_engine():
    """tools/regex_diff_fuzz.py: random patterns from an RE2 grammar (Unicode categories, case folding, POSIX and Perl classes,
    lazy quantifiers, word boundaries, flags) against random values — the built-in engine and Python's `re` (on the translated
    pattern) must agree on every match."""
    out = run_tool([os.path.join(ROOT, "tools", "regex_diff_fuzz.py"), "2000", "4"])
    assert "patterns 2000" in out and "disagreements 0" in out
