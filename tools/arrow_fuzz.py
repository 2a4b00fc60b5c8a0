"""Arrow C data records with DETECTABLE defects — negative lengths / offsets / null counts, missing buffer tables, missing value or
offset buffers, wrong child counts, missing children / dictionaries, unknown or missing format strings — through the host-only
fdb_arrow_roundtrip (import + export, no device): an error code or a record, never a crash. (Defects the interface cannot express —
a length larger than a buffer really is — are the producer's to avoid: the C data interface carries no buffer sizes.)
    python tools/arrow_fuzz.py [records] [seed]"""
import os, sys, ctypes, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pyarrow as pa
from frostdb_amd import physicalplan as pp
from frostdb_amd.arrow_c import ArrowArray, ArrowSchema, ExportedBatch, _RELEASE_FN
lib = ctypes.CDLL(os.environ['FDB_FUZZ_LIB']) if os.environ.get('FDB_FUZZ_LIB') else pp.lib()  # (e.g. the ASan build of tools/asan_arrow.sh)
random.seed(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
rng = np.random.default_rng(3)

def record(n):
    cols = {
        "labels.d": pa.DictionaryArray.from_arrays(pa.array(rng.integers(0, 4, n).astype(np.uint32), mask=rng.random(n) < 0.2), pa.array([b"a", b"bb", b"", b"dddd"], type=pa.binary())),
        "labels.s": pa.array([None if i % 5 == 0 else "s%d" % (i % 7) for i in range(n)], type=pa.string()),
        "labels.lb": pa.array([b"x%d" % (i % 3) for i in range(n)], type=pa.large_binary()),
        "ts": pa.array(rng.integers(0, 10**6, n).astype(np.int64)),
        "value": pa.array(rng.random(n), mask=rng.random(n) < 0.1),
        "flag": pa.array(rng.random(n) < 0.5),
        "u": pa.array(rng.integers(0, 10, n).astype(np.uint64)),
    }
    names = random.sample(list(cols), random.randint(1, len(cols)))
    b = pa.RecordBatch.from_arrays([cols[k] for k in names], names=names)
    return b.slice(random.randint(0, 3), max(0, n - 7)) if random.random() < 0.3 else b

keep = []
log = []
def mutate_array(a, depth=0, real_children=None):
    r = random.random(); log.append(('array', depth, round(r, 3), a.length, a.n_buffers, a.n_children))
    if r < 0.10: a.length = random.choice([-1, -2**40])
    elif r < 0.18: a.offset = random.choice([-1, -7])
    elif r < 0.26: a.null_count = random.choice([-5, 2**40])
    elif r < 0.34: a.n_buffers = random.choice([0, 1, -1])
    elif r < 0.42: a.buffers = None
    elif r < 0.52 and a.buffers and a.n_buffers > 1: a.buffers[random.randrange(1, a.n_buffers)] = None
    elif r < 0.60: a.n_children = random.choice([0, a.n_children + 1 if a.children else 3, -1])
    elif r < 0.66: a.children = None
    elif r < 0.72: a.dictionary = None
    elif real_children and a.children and depth < 1:  # (only children that really exist: the counts may already be mutated)
        k = random.randrange(real_children)
        if a.children[k]: mutate_array(a.children[k].contents, depth + 1)

def mutate_schema(s, depth=0, real_children=None):
    r = random.random(); log.append(('schema', depth, round(r, 3), s.format, s.n_children))
    if r < 0.15:
        f = random.choice([None, b"", b"?", b"+l", b"tsu:", b"w:4", b"+m"]); keep.append(f); s.format = f  # (never a known type of another width: that is a lie about buffer sizes, which the interface cannot express)
    elif r < 0.25: s.n_children = random.choice([0, -1, s.n_children + 2])
    elif r < 0.32: s.children = None
    elif r < 0.40: s.dictionary = None
    elif r < 0.46: s.name = None
    elif real_children and s.children and depth < 1:
        k = random.randrange(real_children)
        if s.children[k]: mutate_schema(s.children[k].contents, depth + 1)

codes = {}
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3000):
    ex = ExportedBatch(record(random.choice([0, 1, 9, 64, 300])))
    # a private copy of the top-level structs so that pyarrow's own release still sees what it exported
    a, s = ArrowArray(), ArrowSchema()
    ctypes.memmove(ctypes.addressof(a), ctypes.addressof(ex.array), ctypes.sizeof(a)); ctypes.memmove(ctypes.addressof(s), ctypes.addressof(ex.schema), ctypes.sizeof(s))
    saved = []
    def snapshot(st):  # child structs are mutated in place: remember them
        saved.append((ctypes.addressof(st), ctypes.string_at(ctypes.addressof(st), ctypes.sizeof(st))))
        if st.n_children > 0 and st.children:
            for k in range(st.n_children): snapshot(st.children[k].contents)
        if st.dictionary: snapshot(st.dictionary.contents)
    snapshot(ex.array); snapshot(ex.schema)
    bufsaved = []
    def snapbuf(st):
        if st.buffers and st.n_buffers > 0: bufsaved.append((ctypes.cast(st.buffers, ctypes.c_void_p).value, st.n_buffers, ctypes.string_at(ctypes.cast(st.buffers, ctypes.c_void_p).value, 8 * st.n_buffers)))
        if st.n_children > 0 and st.children:
            for k in range(st.n_children): snapbuf(st.children[k].contents)
        if st.dictionary: snapbuf(st.dictionary.contents)
    snapbuf(ex.array)
    for _ in range(random.randint(1, 3)):
        which = random.random()
        if which < 0.3: mutate_array(a, 0, ex.array.n_children)
        elif which < 0.6 and ex.array.n_children > 0: mutate_array(ex.array.children[random.randrange(ex.array.n_children)].contents, 1)
        elif which < 0.8: mutate_schema(s, 0, ex.schema.n_children)
        elif ex.schema.n_children > 0: mutate_schema(ex.schema.children[random.randrange(ex.schema.n_children)].contents, 1)
    out, outs = ArrowArray(), ArrowSchema()
    open(os.path.join(os.environ.get("TMPDIR", "/tmp"), "fdb_arrow_case.txt"), "w").write(repr((it, log))); del log[:]
    rc = lib.fdb_arrow_roundtrip(ctypes.byref(a), ctypes.byref(s), ctypes.byref(out), ctypes.byref(outs))
    codes[rc] = codes.get(rc, 0) + 1
    if os.environ.get('FDB_FUZZ_TRACE'): print('returned', it, rc, flush=True)
    if rc == 0:
        for st in (out, outs):
            if st.release: _RELEASE_FN(st.release)(ctypes.addressof(st))
    for addr, raw in saved: ctypes.memmove(addr, raw, len(raw))
    for addr, nb, raw in bufsaved: ctypes.memmove(addr, raw, len(raw))
    ex.close()
    if os.environ.get('FDB_FUZZ_TRACE'): print('closed', it, flush=True)
print("codes", codes)
