"""Fuzzes the built-in regular-expression engine (fdb_regex_match) with random patterns made of RE2 syntax fragments and random
values: python tools/regex_fuzz.py [patterns, default 20000] [seed]. Every call must come back 0 (matched or not) or 1
(FDB_ERR_INVALID: does not compile) — never crash, never take long (matching is linear; compile is bounded by the program-size
limit). With FDB_ASAN_LIB set the ASan + UBSan build of the library is used (tools/asan_full.sh)."""
import ctypes, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = os.environ.get("FDB_ASAN_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "frostdb_amd", "libfrostdb_amd.so")
lib = ctypes.CDLL(path)
lib.fdb_regex_match.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_char_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int32)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
random.seed(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
frags = ["a", "b", "é", ".", "*", "+", "?", "|", "(", ")", "(?:", "(?i)", "(?i:", "(?P<n>", "(?s)", "(?-i)", "[", "]", "[^", "a-z", "\\d", "\\W", "\\b", "\\B", "\\A", "\\z", "^", "$",
         "{2}", "{1,3}", "{0,}", "{", "}", "{1000}", "{1001}", "\\Q", "\\E", "[:alpha:]", "[[:digit:]]", "\\x41", "\\x{1F600}", "\\101", "\\", "\\1", "\\p", "-", ",", "\xff", "\n", "(?", "(?<", "=", "!",
         "\\p{Lu}", "\\pL", "\\P{Nd}", "\\p{^Zs}", "\\p{Any}", "\\p{Greek}", "\\p{", "\\PZ", "σ", "ς", "\u212a", "ß", "Я"]
vals = [b"", b"a", b"ab" * 50, "éa".encode(), b"\xff\xfe", b"A1_b\n", b"(a)", "Σσς K\u212a ẞ я".encode()]
codes, slow = {}, 0
for it in range(n):
    pat = "".join(random.choice(frags) for _ in range(random.randint(1, 12))).encode("utf-8", "surrogateescape") if random.random() < 0.9 else bytes(random.randrange(256) for _ in range(random.randint(1, 10)))
    v = random.choice(vals)
    m = ctypes.c_int32()
    t0 = time.perf_counter()
    rc = lib.fdb_regex_match(pat, len(pat), v, len(v), ctypes.byref(m))
    if time.perf_counter() - t0 > 1.0:
        slow += 1
        print("slow", pat, flush=True)
    codes[rc] = codes.get(rc, 0) + 1
    assert rc in (0, 1), (rc, pat)
print("runs", n, "return codes", codes, "slow", slow)
