"""ctypes wrapper of the CPU oracle (oracle/oracle.cpp).

*** TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT. *** Only tests/, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this package; nothing under ``frostdb_amd/`` does — and this package imports
nothing from ``frostdb_amd/`` either (oracle/_bridge.py builds descriptors and exports Arrow records on its own).

``OraclePlan`` mirrors one query: N operator chains ``PredicateFilter → HashAggregate(final=false)``
fanned into a ``Synchronizer`` and a ``HashAggregate(final=true)`` (query/physicalplan/physicalplan.go:432-474).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Any, Dict, List, Optional, Sequence

import numpy as np
import pyarrow as pa

from ._bridge import Desc, Exported  # the oracle's own descriptor / Arrow plumbing: nothing is shared with frostdb_amd

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

T_I64, T_U64, T_F64, T_BOOL, T_STR, T_DICT = 1, 2, 3, 4, 5, 6


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, i32, i64, u64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64
        L.oracle_last_error.restype = ctypes.c_char_p
        L.oracle_metro_hash64.restype = u64
        L.oracle_metro_hash64.argtypes = [ctypes.c_char_p, i64, u64]
        L.oracle_hash_combine.restype = u64
        L.oracle_hash_combine.argtypes = [u64, u64]
        L.oracle_build_index_ranges.restype = i64
        L.oracle_build_index_ranges.argtypes = [ctypes.POINTER(ctypes.c_uint32), i64, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
        L.oracle_batch_import.argtypes = [vp, vp, ctypes.POINTER(vp)]
        L.oracle_batch_release.argtypes = [vp]
        L.oracle_batch_num_rows.restype = i64
        L.oracle_batch_num_rows.argtypes = [vp]
        L.oracle_batch_num_cols.restype = i32
        L.oracle_batch_num_cols.argtypes = [vp]
        L.oracle_batch_col_name.restype = ctypes.c_char_p
        L.oracle_batch_col_name.argtypes = [vp, i32]
        L.oracle_batch_col_type.restype = i32
        L.oracle_batch_col_type.argtypes = [vp, i32]
        L.oracle_batch_col_valid.argtypes = [vp, i32, vp]
        L.oracle_batch_col_i64.argtypes = [vp, i32, vp]
        L.oracle_batch_col_f64.argtypes = [vp, i32, vp]
        L.oracle_batch_col_str.argtypes = [vp, i32, i64, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(i64)]
        L.oracle_batch_col_codes.restype = i64
        L.oracle_batch_col_codes.argtypes = [vp, i32, vp]
        L.oracle_batch_col_code_value.argtypes = [vp, i32, i64, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(i64)]
        L.oracle_plan_create.argtypes = [vp, i32, u64, ctypes.POINTER(vp)]
        L.oracle_plan_close.argtypes = [vp]
        L.oracle_plan_last_error.restype = ctypes.c_char_p
        L.oracle_plan_last_error.argtypes = [vp]
        L.oracle_plan_push.argtypes = [vp, i32, vp]
        L.oracle_plan_filter.argtypes = [vp, vp, ctypes.POINTER(vp), ctypes.POINTER(i32), vp, ctypes.POINTER(i64)]
        L.oracle_plan_finish.argtypes = [vp, ctypes.POINTER(vp)]
        L.oracle_plan_finish_next.argtypes = [vp, ctypes.POINTER(vp)]
        L.oracle_plan_execute.argtypes = [vp, ctypes.POINTER(vp), i64, i32, ctypes.POINTER(vp)]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"oracle error {code}: {msg}")
        self.code = code


def metro_hash64(data: bytes, seed: int = 0) -> int:
    return lib().oracle_metro_hash64(data, len(data), seed)


def build_index_ranges(indices):
    """buildIndexRanges (filter.go:332-354): sorted row indices → [(start, end)) runs."""
    n = len(indices)
    arr = (ctypes.c_uint32 * max(n, 1))(*indices)
    st, en = (ctypes.c_uint32 * max(n, 1))(), (ctypes.c_uint32 * max(n, 1))()
    k = lib().oracle_build_index_ranges(arr, n, st, en)
    return [(st[i], en[i]) for i in range(k)]


class OracleBatch:
    def __init__(self, handle: int):
        self.handle = handle

    @classmethod
    def from_arrow(cls, batch: pa.RecordBatch) -> "OracleBatch":
        out = ctypes.c_void_p()
        with Exported(batch) as ex:
            rc = lib().oracle_batch_import(ex.array, ex.schema, ctypes.byref(out))
        if rc != 0:
            raise OracleError(rc, lib().oracle_last_error().decode())
        return cls(out.value)

    @property
    def num_rows(self) -> int:
        return lib().oracle_batch_num_rows(self.handle)

    def to_pydict(self) -> Dict[str, List[Any]]:
        """Column name → list of Python values (None for null; bytes for string-like columns)."""
        L = lib()
        n = self.num_rows
        out: Dict[str, List[Any]] = {}
        for c in range(L.oracle_batch_num_cols(self.handle)):
            name = L.oracle_batch_col_name(self.handle, c).decode()
            t = L.oracle_batch_col_type(self.handle, c)
            valid = np.zeros(n, dtype=np.uint8)
            if n:
                L.oracle_batch_col_valid(self.handle, c, valid.ctypes.data)
            if t in (T_I64, T_U64, T_BOOL):
                v = np.zeros(n, dtype=np.int64)
                if n:
                    L.oracle_batch_col_i64(self.handle, c, v.ctypes.data)
                if t == T_U64:
                    v = v.view(np.uint64)
                vals = [(bool(x) if t == T_BOOL else int(x)) if ok else None for x, ok in zip(v, valid)]
            elif t == T_F64:
                v = np.zeros(n, dtype=np.float64)
                if n:
                    L.oracle_batch_col_f64(self.handle, c, v.ctypes.data)
                vals = [float(x) if ok else None for x, ok in zip(v, valid)]
            else:
                vals = []
                p = ctypes.c_char_p()
                ln = ctypes.c_int64()
                for r in range(n):
                    if not valid[r]:
                        vals.append(None)
                        continue
                    L.oracle_batch_col_str(self.handle, c, r, ctypes.byref(p), ctypes.byref(ln))
                    vals.append(ctypes.string_at(p, ln.value))
            out[name] = vals
        return out

    def to_arrow(self) -> pa.RecordBatch:
        """The record as Arrow, built column-wise (big results: millions of groups): string-like columns become
        dictionary<uint32, binary> over the oracle's value table, int64 / uint64 / bool / float64 columns plain arrays."""
        L = lib()
        n = self.num_rows
        arrays, names = [], []
        for c in range(L.oracle_batch_num_cols(self.handle)):
            names.append(L.oracle_batch_col_name(self.handle, c).decode())
            t = L.oracle_batch_col_type(self.handle, c)
            valid = np.zeros(n, dtype=np.uint8)
            if n:
                L.oracle_batch_col_valid(self.handle, c, valid.ctypes.data)
            mask = valid == 0
            if t in (T_I64, T_U64, T_BOOL):
                v = np.zeros(n, dtype=np.int64)
                if n:
                    L.oracle_batch_col_i64(self.handle, c, v.ctypes.data)
                arr = pa.array(v.view(np.uint64) if t == T_U64 else v.astype(bool) if t == T_BOOL else v, mask=mask)
            elif t == T_F64:
                v = np.zeros(n, dtype=np.float64)
                if n:
                    L.oracle_batch_col_f64(self.handle, c, v.ctypes.data)
                arr = pa.array(v, mask=mask)
            else:
                codes = np.zeros(max(n, 1), dtype=np.uint32)
                k = L.oracle_batch_col_codes(self.handle, c, codes.ctypes.data)
                vals = []
                p, ln = ctypes.c_char_p(), ctypes.c_int64()
                for i in range(k):
                    L.oracle_batch_col_code_value(self.handle, c, i, ctypes.byref(p), ctypes.byref(ln))
                    vals.append(ctypes.string_at(p, ln.value))
                arr = pa.DictionaryArray.from_arrays(pa.array(codes[:n], mask=mask), pa.array(vals, type=pa.binary()))
            arrays.append(arr)
        return pa.RecordBatch.from_arrays(arrays, names=names)

    def close(self) -> None:
        if self.handle:
            lib().oracle_batch_release(self.handle)
            self.handle = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_UNICODE_CLASS_CACHE = {}
# RE2's Perl classes as class bodies: (the set, its complement over all runes); \s has no \v in RE2
_PERL_CLASS = {
    b"d": (b"0-9", b"\\x00-/:-\\U0010FFFF"),
    b"w": (b"0-9A-Za-z_", b"\\x00-/:-@\\[-\\^`{-\\U0010FFFF"),
    b"s": (b"\\t\\n\\f\\r ", b"\\x00-\\x08\\x0b\\x0e-\\x1f!-\\U0010FFFF"),
}
_PERL_CLASS.update({k.upper(): v for k, v in list(_PERL_CLASS.items())})


def _unicode_class_ranges(name: str, negated: bool):
    """The body of a Python character class (\\UXXXXXXXX-\\UXXXXXXXX runs) for a Unicode general category (two letters), a group of them
    (one letter) or Any — None for anything else (scripts). Built from this interpreter's unicodedata, rune by rune: independent of
    the product's generated tables."""
    import unicodedata
    key = (name, negated)
    if key in _UNICODE_CLASS_CACHE:
        return _UNICODE_CLASS_CACHE[key]
    if name == "Any":
        member = lambda cat: True  # noqa: E731
    elif len(name) == 2 and name != "Cn":
        member = lambda cat: cat == name  # noqa: E731
    elif len(name) == 1 and name in "LMNPSZC":
        member = lambda cat: cat[0] == name and cat != "Cn"  # noqa: E731
    else:
        return None
    runs, start, prev = [], None, None
    for c in range(0x110000):
        cat = "Cs" if 0xD800 <= c <= 0xDFFF else unicodedata.category(chr(c))
        hit = member(cat) != negated
        if 0xD800 <= c <= 0xDFFF:
            hit = False  # (a str pattern cannot hold lone surrogates; no UTF-8 text decodes to one either)
        if hit:
            if start is None:
                start = c
            prev = c
        elif start is not None:
            runs.append((start, prev)); start = None
    if start is not None:
        runs.append((start, prev))
    if not runs:
        return None
    body = "".join("\\U%08X" % a if a == b else "\\U%08X-\\U%08X" % (a, b) for a, b in runs).encode()
    _UNICODE_CLASS_CACHE[key] = body
    return body


def go_regexp_to_python(pattern: bytes) -> bytes:
    """RE2 (Go regexp) syntax → Python `re` syntax for the constructs whose SPELLING differs; their meaning is the same. `$` without
    the m flag is the end of the TEXT in Go (Python's would also match before a trailing newline) and `\\z` is Python's `\\Z`;
    `(?U)` only changes which match is preferred (irrelevant for match / no match); `\\Q…\\E` quotes; POSIX classes are spelled out;
    `(?<name>…)` is `(?P<name>…)`; `\\p{..}` general categories are spelled out as ranges from unicodedata. Script classes (\\p{Greek}) are left alone and
    fail to compile there too."""
    import re as _re
    posix = {b"alnum": b"0-9A-Za-z", b"alpha": b"A-Za-z", b"ascii": b"\\x00-\\x7f", b"blank": b"\\t ", b"cntrl": b"\\x00-\\x1f\\x7f", b"digit": b"0-9",
             b"graph": b"!-~", b"lower": b"a-z", b"print": b" -~", b"punct": b"!-/:-@\\[-`{-~", b"space": b"\\t\\n\\v\\f\\r ", b"upper": b"A-Z",
             b"word": b"0-9A-Za-z_", b"xdigit": b"0-9A-Fa-f"}
    out, i, n, in_class, multiline = bytearray(), 0, len(pattern), False, False
    # A bracket class is collected and emitted at its `]`: members that are NEGATED named classes (\W, \P{..}, [:^alpha:]) cannot
    # be spelled as their complement's ranges, because under (?i) regexp/syntax folds the POSITIVE class and negates afterwards
    # (parser.appendGroup): `(?i)[\W]` does not match "k" although U+212A, which folds to k, is in the complement of \w. They
    # become `[^…]` alternatives, which Python evaluates the same way (fold the rune, then test the negated set).
    cls_pos, cls_neg, cls_negated = bytearray(), [], False

    def close_class():
        if not cls_neg:
            return b"[" + (b"^" if cls_negated else b"") + bytes(cls_pos) + b"]"
        alts = ([b"[" + bytes(cls_pos) + b"]"] if cls_pos else []) + [b"[^" + b + b"]" for b in cls_neg]
        alt = b"(?:" + b"|".join(alts) + b")"
        return b"(?:(?!" + alt + b")[\s\S])" if cls_negated else alt
    while i < n:
        c = pattern[i:i + 1]
        if c == b"\\" and i + 1 < n:
            nx = pattern[i + 1:i + 2]
            if nx == b"Q" and not in_class:
                j = pattern.find(b"\\E", i + 2)
                lit = pattern[i + 2: j if j >= 0 else n]
                out += _re.escape(lit)
                i = (j + 2) if j >= 0 else n
                continue
            if nx == b"z" and not in_class:
                out += b"\\Z"; i += 2; continue
            if nx in _PERL_CLASS:  # \d \w \s are ASCII-only in RE2 (Python's are Unicode-aware on str patterns): spelled out
                pos, _comp = _PERL_CLASS[nx]
                if nx.isupper():
                    if in_class:
                        cls_neg.append(pos)
                    else:
                        out += b"[^" + pos + b"]"
                elif in_class:
                    cls_pos += pos
                else:
                    out += b"[" + pos + b"]"
                i += 2; continue
            if nx in (b"b", b"B") and not in_class:  # ASCII word boundary (Python's \b counts é as a word character, Go's does not)
                w = b"[0-9A-Za-z_]"
                if nx == b"b":  # ((?-i: the boundary is about ASCII word characters whatever the case folding in force: K U+212A is none)
                    out += b"(?-i:(?<=" + w + b")(?!" + w + b")|(?<!" + w + b")(?=" + w + b"))"
                else:
                    out += b"(?-i:(?<=" + w + b")(?=" + w + b")|(?<!" + w + b")(?!" + w + b"))"
                i += 2; continue
            if nx in (b"p", b"P"):  # \pL \p{Lu} \P{Nd} \p{^Zs} \p{Any}: spelled out as ranges from unicodedata (Python's re has no \p)
                neg = nx == b"P"
                if pattern[i + 2:i + 3] == b"{":
                    j = pattern.find(b"}", i + 3)
                    name = pattern[i + 3:j] if j >= 0 else None
                    end = j + 1
                else:
                    name, end = pattern[i + 2:i + 3], i + 3
                if name is not None and name.startswith(b"^"):
                    neg, name = not neg, name[1:]
                body = _unicode_class_ranges(name.decode("ascii", "replace"), False) if name else None
                if body is not None:
                    if neg and in_class:
                        cls_neg.append(body)
                    elif neg:
                        out += b"[^" + body + b"]"
                    elif in_class:
                        cls_pos += body
                    else:
                        out += b"[" + body + b"]"
                    i = end; continue
            if in_class:
                cls_pos += pattern[i:i + 2]
            else:
                out += pattern[i:i + 2]
            i += 2; continue
        if in_class:
            if c == b"[" and pattern[i + 1:i + 2] == b":":
                j = pattern.find(b":]", i + 2)
                name = pattern[i + 2:j] if j >= 0 else b""
                neg = name.startswith(b"^")
                if j >= 0 and name.lstrip(b"^") in posix:
                    if neg:
                        cls_neg.append(posix[name[1:]])
                    else:
                        cls_pos += posix[name]
                    i = j + 2; continue
            if c == b"]":
                in_class = False
                out += close_class()
                i += 1; continue
            cls_pos += c; i += 1; continue
        if c == b"[":
            in_class = True
            cls_pos, cls_neg, cls_negated = bytearray(), [], False
            i += 1
            if pattern[i:i + 1] == b"^":
                cls_negated = True; i += 1
            if pattern[i:i + 1] == b"]":
                cls_pos += b"\\]"; i += 1
            continue
        if c == b"(" and pattern[i + 1:i + 2] == b"?":
            m = _re.match(rb"\(\?([imsU]*)(-[imsU]+)?([:)])", pattern[i:])
            if m:
                on = m.group(1).replace(b"U", b"")
                off = (m.group(2) or b"").replace(b"U", b"")
                if b"m" in on:
                    multiline = True
                flags = on + (off if len(off) > 1 else b"")
                if m.group(3) == b")":
                    out += (b"(?" + flags + b")") if flags else b""
                else:
                    out += (b"(?" + flags + b":") if flags else b"(?:"
                i += m.end(); continue
            if pattern[i + 2:i + 3] == b"<" and pattern[i + 3:i + 4] not in (b"=", b"!"):
                out += b"(?P<"; i += 3; continue
        if c == b"$" and not multiline:
            out += b"\\Z"; i += 1; continue
        out += c; i += 1
    if in_class:  # (unterminated: left for Python to refuse, like Go does)
        out += b"[" + (b"^" if cls_negated else b"") + bytes(cls_pos)
    return bytes(out)


_REGEX_CB_TYPE = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64)


def _python_regex_callback():
    """fdb_regex_match_fn over Python's `re` (str patterns on UTF-8 text: `.` and classes see runes, like Go's), compiled once per pattern."""
    import re as _re
    cache = {}

    def cb(_user, pat_p, pat_n, val_p, val_n):
        try:
            pat = ctypes.string_at(pat_p, pat_n)
            rx = cache.get(pat)
            if rx is None:
                rx = cache[pat] = _re.compile(go_regexp_to_python(pat).decode("utf-8"))
            val = ctypes.string_at(val_p, val_n) if val_n else b""
            return 1 if rx.search(val.decode("utf-8", "replace")) else 0
        except Exception:  # noqa: BLE001  (does not compile)
            return -1
    return _REGEX_CB_TYPE(cb)


class OraclePlan:
    def __init__(self, filter_expr, aggs: Sequence = (), groups: Sequence = (), nchains: int = 1, seed: int = 0x5EED, go_regexp: bool = True):
        """`go_regexp` (default): `=~` / `!~` literals are RE2 syntax matched by Python's `re` on the translated pattern (Go's
        regexp semantics, an engine independent of the product's); False: the restatement's own std::regex (ECMAScript)."""
        self._regex_cb = _python_regex_callback() if go_regexp else None
        self._desc = Desc(filter_expr, list(aggs), list(groups), regex_fn=ctypes.cast(self._regex_cb, ctypes.c_void_p).value if self._regex_cb else 0)
        out = ctypes.c_void_p()
        rc = lib().oracle_plan_create(self._desc.address, nchains, seed, ctypes.byref(out))
        if rc != 0:
            raise OracleError(rc, lib().oracle_last_error().decode())
        self.handle = out.value
        self.nchains = nchains
        self._rr = 0

    def _check(self, rc: int) -> None:
        if rc != 0:
            raise OracleError(rc, lib().oracle_plan_last_error(self.handle).decode())

    def push(self, batch, chain: Optional[int] = None) -> None:
        """≙ Callback on one chain (round-robin like query.FakeTableReader, query/testing.go:24-43)."""
        own = None
        if isinstance(batch, pa.RecordBatch):
            own = batch = OracleBatch.from_arrow(batch)
        if chain is None:
            chain = self._rr % self.nchains
            self._rr += 1
        self._check(lib().oracle_plan_push(self.handle, chain, batch.handle))
        if own is not None:
            own.close()

    def filter(self, batch):
        """≙ filter(): (compacted OracleBatch or None, selected indices)."""
        own = None
        if isinstance(batch, pa.RecordBatch):
            own = batch = OracleBatch.from_arrow(batch)
        out = ctypes.c_void_p()
        empty = ctypes.c_int32()
        idx = np.zeros(max(batch.num_rows, 1), dtype=np.uint32)
        n = ctypes.c_int64()
        self._check(lib().oracle_plan_filter(self.handle, batch.handle, ctypes.byref(out), ctypes.byref(empty),
                                             idx.ctypes.data, ctypes.byref(n)))
        if own is not None:
            own.close()
        return (None if empty.value else OracleBatch(out.value)), idx[: n.value].copy()

    def finish(self) -> OracleBatch:
        out = ctypes.c_void_p()
        self._check(lib().oracle_plan_finish(self.handle, ctypes.byref(out)))
        return OracleBatch(out.value)

    def finish_next(self) -> Optional[OracleBatch]:
        """The records of a Finish after the first (a plain binary key builder that reached its size limit started a new aggregate,
        aggregate.go:426-468; $FDB_TEST_MAX_KEY_BYTES lowers math.MaxInt32 for tests); None when there is none left."""
        out = ctypes.c_void_p()
        self._check(lib().oracle_plan_finish_next(self.handle, ctypes.byref(out)))
        return OracleBatch(out.value) if out.value else None

    def execute(self, batches: Sequence[OracleBatch], nthreads: int) -> OracleBatch:
        """The CPU baseline: `nthreads` chains pulling from one queue, then Synchronizer + final stage."""
        arr = (ctypes.c_void_p * len(batches))(*[b.handle for b in batches])
        out = ctypes.c_void_p()
        self._check(lib().oracle_plan_execute(self.handle, arr, len(batches), nthreads, ctypes.byref(out)))
        return OracleBatch(out.value)

    def close(self) -> None:
        if self.handle:
            lib().oracle_plan_close(self.handle)
            self.handle = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
