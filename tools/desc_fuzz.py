"""Random plan descriptors — out-of-range child indices, cycles, missing arrays, negative counts, malformed projections, dynamic
aggregations, patterns that do not compile — through fdb_plan_explain (validation + Draw, no device): every call must come back with
an error code or a drawing, never crash or hang. python tools/desc_fuzz.py [descriptors] [seed]; the case in flight is kept in
$TMPDIR/fdb_desc_case.txt. (It found the one crash it was written for: a projection without nodes under a dynamic aggregation.)"""
import os, sys, ctypes, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frostdb_amd import physicalplan as pp
from frostdb_amd.logicalplan import CLiteral, CExpr, CAggregation, CGroupExpr, CProjNode, CProjection, CPlanDesc
lib = pp.lib()
if os.environ.get('FDB_FUZZ_LIB'):  # an instrumented build of the whole library (hipcc -fsanitize=address,undefined -fno-gpu-sanitize)
    lib = ctypes.CDLL(os.environ['FDB_FUZZ_LIB'])
    lib.fdb_plan_explain.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
random.seed(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
names = [b"labels.a", b"value", b"timestamp", b"", b"x" * 300, None, b"labels", b"sum(value)"]
pats = [b"a.*", b"(", b"[", b"\\", b"", b"^x$", b"(a|b)+" * 50]
def lit():
    l = CLiteral(); l.type = random.choice([0, 1, 2, 3, 4, 5, 6, 7, 99, -1])
    l.i64 = random.choice([0, 1, -1, 2**62]); l.u64 = 5; l.f64 = random.choice([0.0, float("nan"), 1e300])
    d = random.choice(pats + [None]); l.data = d; l.len = random.choice([0 if d is None else len(d), 0, -1 if d is None else len(d)])
    if d is None: l.len = 0
    return l
codes = {}
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2000):
    keep = []; pjlog = []
    d = CPlanDesc()
    nf = random.choice([0, 0, 1, 2, 3, 5, 9])
    if nf:
        ex = (CExpr * nf)()
        for i in range(nf):
            ex[i].op = random.choice(list(range(0, 16)) + [99, -3])
            ex[i].left = random.choice([-1, 0, 1, i - 1, i, i + 1, nf, 1000, -5])
            ex[i].right = random.choice([-1, 0, 1, i - 1, i, i + 1, nf, 1000, -5])
            ex[i].column = random.choice(names)
            ex[i].literal = lit()
        keep.append(ex); d.filter = ctypes.cast(ex, ctypes.POINTER(CExpr))
    d.n_filter = random.choice([nf, nf, nf, -1])
    if nf == 0 and d.n_filter > 0: d.n_filter = 0
    d.filter_root = random.choice([-1, 0, nf - 1, nf, 77])
    na = random.choice([0, 1, 2, 4, 9])
    if na:
        ag = (CAggregation * na)()
        for i in range(na):
            ag[i].func = random.choice([0, 1, 2, 3, 4, 5, 6, 7, 50, -1]); ag[i].dynamic = random.choice([0, 1, 7]); ag[i].column = random.choice(names)
        keep.append(ag); d.aggs = ctypes.cast(ag, ctypes.POINTER(CAggregation))
    d.n_aggs = na
    ng = random.choice([0, 1, 2, 70])
    if ng:
        gr = (CGroupExpr * ng)()
        for i in range(ng):
            gr[i].name = random.choice(names); gr[i].dynamic = random.choice([0, 1])
        keep.append(gr); d.groups = ctypes.cast(gr, ctypes.POINTER(CGroupExpr))
    d.n_groups = ng
    d.final_stage = random.choice([0, 0, 1]); d.ordered = random.choice([0, 0, 1, 5])
    npj = random.choice([0, 0, 1, 2])
    if npj:
        pj = (CProjection * npj)()
        for i in range(npj):
            nn = random.choice([1, 2, 3, 6])
            nodes = (CProjNode * nn)()
            for k in range(nn):
                nodes[k].kind = random.choice([0, 1, 2, 3, 4, 5, 6, 7, -1]); nodes[k].op = random.choice([0, 1, 5, 16, 17, 18, 19, 20, k - 1, 99])
                nodes[k].left = random.choice([-1, 0, k - 1, k, nn, 50]); nodes[k].right = random.choice([-1, 0, k - 1, k, nn, 50])
                nodes[k].column = random.choice(names); nodes[k].literal = lit()
            keep.append(nodes); pjlog.append(dict(nodes=[(nodes[k].kind, nodes[k].op, nodes[k].left, nodes[k].right, nodes[k].column, nodes[k].literal.type) for k in range(nn)]))
            pj[i].name = random.choice(names); pj[i].nodes = ctypes.cast(nodes, ctypes.POINTER(CProjNode)) if random.random() < 0.9 else None
            pj[i].n_nodes = random.choice([nn, nn, 0, -1]); pj[i].root = random.choice([nn - 1, 0, nn, -1]); pjlog[-1].update(name=pj[i].name, n_nodes=pj[i].n_nodes, root=pj[i].root, null_nodes=not bool(pj[i].nodes))
        keep.append(pj); d.projections = ctypes.cast(pj, ctypes.POINTER(CProjection))
    d.n_projections = npj
    buf = ctypes.create_string_buffer(4096); need = ctypes.c_int64(0)
    open(os.path.join(os.environ.get("TMPDIR", "/tmp"), "fdb_desc_case.txt"), "w").write(repr(dict(it=it, nf=nf, n_filter=d.n_filter, root=d.filter_root, na=na, ng=ng, npj=npj, final=d.final_stage, ordered=d.ordered, ex=[(ex[i].op, ex[i].left, ex[i].right, ex[i].column, ex[i].literal.type, ex[i].literal.data, ex[i].literal.len) for i in range(nf)] if nf else [], ag=[(ag[i].func, ag[i].dynamic, ag[i].column) for i in range(na)] if na else [], gr=[(gr[i].name, gr[i].dynamic) for i in range(ng)] if ng else [], pj=pjlog)))
    rc = lib.fdb_plan_explain(ctypes.addressof(d), buf, len(buf), ctypes.byref(need))
    codes[rc] = codes.get(rc, 0) + 1
print("codes", codes)
