"""Expression builders that mirror the reference's ``query/logicalplan`` vocabulary.

Only what the hot path's boundary needs: the *inputs* of ``physicalplan.Filter`` and
``physicalplan.Aggregate`` (query/physicalplan/filter.go:230-237, aggregate.go:23-79):

* ``Col(name)`` / ``DynCol(name)``            — logicalplan/expr.go:321-355, :540-566
* ``Col(x) == lit`` … ``.RegexMatch`` ``.Contains`` — BinaryExpr builders, expr.go:357-470
* ``And(a, b, …)`` / ``Or(a, b, …)``           — expr.go:472-520 (left-deep folding like the reference)
* ``Sum/Min/Max/Count(Col(x))``               — AggregationFunction, expr.go:752-800; ``Name()`` is
  ``"sum(value)"`` (expr.go:700-702) and is the output column name.

``to_desc()`` flattens them into the C structs of ``include/frostdb_amd.h`` (numeric enum values are
the reference's own ``Op`` / ``AggFunc`` iota values).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field
from typing import Any, Callable, List, Optional, Sequence, Union

# logicalplan.Op (expr.go:17-35)
OP_EQ, OP_NOT_EQ, OP_LT, OP_LT_EQ, OP_GT, OP_GT_EQ = 1, 2, 3, 4, 5, 6
OP_REGEX_MATCH, OP_REGEX_NOT_MATCH, OP_AND, OP_OR = 7, 8, 9, 10
OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_CONTAINS, OP_NOT_CONTAINS = 11, 12, 13, 14, 15, 16

_OP_STR = {
    OP_EQ: "==", OP_NOT_EQ: "!=", OP_LT: "<", OP_LT_EQ: "<=", OP_GT: ">", OP_GT_EQ: ">=",
    OP_REGEX_MATCH: "=~", OP_REGEX_NOT_MATCH: "!~", OP_AND: "&&", OP_OR: "||",
    OP_CONTAINS: "contains", OP_NOT_CONTAINS: "not contains",
    OP_ADD: "+", OP_SUB: "-", OP_MUL: "*", OP_DIV: "/",
}
_ARITH = (OP_ADD, OP_SUB, OP_MUL, OP_DIV)

# logicalplan.AggFunc (expr.go:718-729)
AGG_SUM, AGG_MIN, AGG_MAX, AGG_COUNT, AGG_AVG, AGG_UNIQUE, AGG_AND = 1, 2, 3, 4, 5, 6, 7
_AGG_STR = {AGG_SUM: "sum", AGG_MIN: "min", AGG_MAX: "max", AGG_COUNT: "count", AGG_AVG: "avg",
            AGG_UNIQUE: "unique", AGG_AND: "and"}

LIT_NULL, LIT_INT64, LIT_UINT64, LIT_FLOAT64, LIT_STRING, LIT_BINARY, LIT_BOOL = 0, 1, 2, 3, 4, 5, 6


class UInt64(int):
    """Marks a Python int as a uint64 literal (scalar.Uint64)."""


@dataclass(frozen=True)
class Literal:
    value: Any  # None | int | UInt64 | float | str | bytes | bool

    def lit_type(self) -> int:
        v = self.value
        if v is None:
            return LIT_NULL
        if isinstance(v, bool):
            return LIT_BOOL
        if isinstance(v, UInt64):
            return LIT_UINT64
        if isinstance(v, int):
            return LIT_INT64
        if isinstance(v, float):
            return LIT_FLOAT64
        if isinstance(v, str):
            return LIT_STRING
        if isinstance(v, (bytes, bytearray)):
            return LIT_BINARY
        raise TypeError(f"unsupported literal {v!r}")

    def __str__(self) -> str:
        return "null" if self.value is None else str(self.value)

    # a literal may be the LEFT operand of an arithmetic expression (`2 * 3`, `2 - 1` in logictest/testdata/exec/aggregate/math)
    def __add__(self, other: Any) -> "BinaryExpr":
        return BinaryExpr(self, OP_ADD, _operand(other))

    def __sub__(self, other: Any) -> "BinaryExpr":
        return BinaryExpr(self, OP_SUB, _operand(other))

    def __mul__(self, other: Any) -> "BinaryExpr":
        return BinaryExpr(self, OP_MUL, _operand(other))

    def __truediv__(self, other: Any) -> "BinaryExpr":
        return BinaryExpr(self, OP_DIV, _operand(other))


class Expr:
    def __and__(self, other: "Expr") -> "BinaryExpr":
        return BinaryExpr(self, OP_AND, other)

    def __or__(self, other: "Expr") -> "BinaryExpr":
        return BinaryExpr(self, OP_OR, other)

    # arithmetic builders (logicalplan/expr.go: Add/Sub/Mul/Div → BinaryExpr with OpAdd … OpDiv); the operand may be
    # another expression or a Python int / float (a literal)
    def __add__(self, other: Any) -> "BinaryExpr":
        return BinaryExpr(self, OP_ADD, _operand(other))

    def __sub__(self, other: Any) -> "BinaryExpr":
        return BinaryExpr(self, OP_SUB, _operand(other))

    def __mul__(self, other: Any) -> "BinaryExpr":
        return BinaryExpr(self, OP_MUL, _operand(other))

    def __truediv__(self, other: Any) -> "BinaryExpr":
        return BinaryExpr(self, OP_DIV, _operand(other))

    def Alias(self, alias: str) -> "AliasExpr":
        return AliasExpr(self, alias)


def _operand(v: Any):
    return v if isinstance(v, (Expr, Literal)) else Literal(v)


def _lit(v: Any) -> Literal:
    return v if isinstance(v, Literal) else Literal(v)


@dataclass(frozen=True, eq=False)
class Column(Expr):
    name: str
    dynamic: bool = False

    # comparison builders (expr.go:357-470)
    def __eq__(self, other: Any) -> "BinaryExpr":  # type: ignore[override]
        return BinaryExpr(self, OP_EQ, _lit(other))

    def __ne__(self, other: Any) -> "BinaryExpr":  # type: ignore[override]
        return BinaryExpr(self, OP_NOT_EQ, _lit(other))

    def __lt__(self, other: Any) -> "BinaryExpr":
        return BinaryExpr(self, OP_LT, _lit(other))

    def __le__(self, other: Any) -> "BinaryExpr":
        return BinaryExpr(self, OP_LT_EQ, _lit(other))

    def __gt__(self, other: Any) -> "BinaryExpr":
        return BinaryExpr(self, OP_GT, _lit(other))

    def __ge__(self, other: Any) -> "BinaryExpr":
        return BinaryExpr(self, OP_GT_EQ, _lit(other))

    def Eq(self, other: Any) -> "BinaryExpr":
        return BinaryExpr(self, OP_EQ, _lit(other))

    def NotEq(self, other: Any) -> "BinaryExpr":
        return BinaryExpr(self, OP_NOT_EQ, _lit(other))

    def RegexMatch(self, pattern: str) -> "BinaryExpr":
        return BinaryExpr(self, OP_REGEX_MATCH, _lit(pattern))

    def RegexNotMatch(self, pattern: str) -> "BinaryExpr":
        return BinaryExpr(self, OP_REGEX_NOT_MATCH, _lit(pattern))

    def Contains(self, needle: Union[str, bytes]) -> "BinaryExpr":
        return BinaryExpr(self, OP_CONTAINS, _lit(needle))

    def NotContains(self, needle: Union[str, bytes]) -> "BinaryExpr":
        return BinaryExpr(self, OP_NOT_CONTAINS, _lit(needle))

    def __hash__(self) -> int:
        return hash((self.name, self.dynamic))

    def __str__(self) -> str:
        return self.name


def Col(name: str) -> Column:
    return Column(name, False)


def DynCol(name: str) -> Column:
    return Column(name, True)


@dataclass(frozen=True, eq=False)
class BinaryExpr(Expr):
    left: Any
    op: int
    right: Any

    def __str__(self) -> str:
        if self.op in (OP_AND, OP_OR):
            return f"({self.left} {'AND' if self.op == OP_AND else 'OR'} {self.right})"
        return f"{self.left} {_OP_STR[self.op]} {self.right}"

    @property
    def name(self) -> str:
        """BinaryExpr.Name(): left.Name() + " " + op + " " + right.Name() (logicalplan/expr.go:181-183) — no parentheses."""
        return str(self)

    @property
    def dynamic(self) -> bool:
        return False


@dataclass(frozen=True, eq=False)
class AliasExpr(Expr):
    """logicalplan.AliasExpr: Name() is the alias (expr.go:1029-1031)."""
    expr: Any
    alias: str

    @property
    def name(self) -> str:
        return self.alias

    @property
    def dynamic(self) -> bool:
        return False

    def __str__(self) -> str:
        return f"{self.expr} as {self.alias}"


@dataclass(frozen=True, eq=False)
class ConvertExpr(Expr):
    """logicalplan.ConvertExpr (expr.go:250-300): only int64 → float64 exists (physicalplan/project.go:507-521)."""
    expr: Any
    to: str = "float64"

    @property
    def name(self) -> str:
        return f"convert({self.expr.name if hasattr(self.expr, 'name') else self.expr}, {self.to})"

    dynamic = False

    def __str__(self) -> str:
        return self.name


@dataclass(frozen=True, eq=False)
class IsNullExpr(Expr):
    """logicalplan.IsNullExpr: Name() = "isnull(<expr>)" (expr.go:862-864)."""
    expr: Any

    @property
    def name(self) -> str:
        return f"isnull({self.expr.name})"

    dynamic = False

    def __str__(self) -> str:
        return self.name


@dataclass(frozen=True, eq=False)
class IfExpr(Expr):
    """logicalplan.IfExpr: Name() = "if(<cond>) { <then> } else { <else>}" (expr.go:982-984, missing space included)."""
    cond: Any
    then: Any
    els: Any

    @property
    def name(self) -> str:
        n = lambda e: e.name if hasattr(e, "name") else str(e)  # noqa: E731
        return "if(" + n(self.cond) + ") { " + n(self.then) + " } else { " + n(self.els) + "}"

    dynamic = False

    def __str__(self) -> str:
        return self.name


def Convert(e, to: str = "float64") -> ConvertExpr:
    return ConvertExpr(e, to)


def IsNull(e) -> IsNullExpr:
    return IsNullExpr(e)


def If(cond, then, els) -> IfExpr:
    return IfExpr(cond, _lit(then) if not isinstance(then, Expr) else then, _lit(els) if not isinstance(els, Expr) else els)


def And(*exprs: Expr) -> Expr:
    """logicalplan.And: folds left-deep (expr.go:472-495)."""
    out = exprs[0]
    for e in exprs[1:]:
        out = BinaryExpr(out, OP_AND, e)
    return out


def Or(*exprs: Expr) -> Expr:
    out = exprs[0]
    for e in exprs[1:]:
        out = BinaryExpr(out, OP_OR, e)
    return out


@dataclass(frozen=True)
class AggregationFunction:
    func: int
    expr: Any  # Column, or an arithmetic BinaryExpr over columns and literals (sum(value * timestamp))

    def Name(self) -> str:
        return f"{_AGG_STR[self.func]}({self.expr.name})"

    def __str__(self) -> str:
        return self.Name()


def Sum(c: Column) -> AggregationFunction:
    return AggregationFunction(AGG_SUM, c)


def Min(c: Column) -> AggregationFunction:
    return AggregationFunction(AGG_MIN, c)


def Max(c: Column) -> AggregationFunction:
    return AggregationFunction(AGG_MAX, c)


def Count(c: Column) -> AggregationFunction:
    return AggregationFunction(AGG_COUNT, c)


def Unique(c: Column) -> AggregationFunction:
    """logicalplan.Unique (expr.go:780-785): the group's value if all its rows carry the same non-NULL int64, else NULL."""
    return AggregationFunction(AGG_UNIQUE, c)


def AndAgg(c: Column) -> AggregationFunction:
    """logicalplan.AndAgg (expr.go:787-792): logical AND over a bool column's valid values."""
    return AggregationFunction(AGG_AND, c)


# ---- C structs of include/frostdb_amd.h ---------------------------------------------------------

class CLiteral(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int32), ("_pad", ctypes.c_int32), ("i64", ctypes.c_int64),
                ("u64", ctypes.c_uint64), ("f64", ctypes.c_double), ("data", ctypes.c_char_p),
                ("len", ctypes.c_int64)]


class CExpr(ctypes.Structure):
    _fields_ = [("op", ctypes.c_int32), ("left", ctypes.c_int32), ("right", ctypes.c_int32),
                ("_pad", ctypes.c_int32), ("column", ctypes.c_char_p), ("literal", CLiteral)]


class CAggregation(ctypes.Structure):
    _fields_ = [("func", ctypes.c_int32), ("dynamic", ctypes.c_int32), ("column", ctypes.c_char_p)]


class CGroupExpr(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("dynamic", ctypes.c_int32), ("_pad", ctypes.c_int32)]


class CProjNode(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("op", ctypes.c_int32), ("left", ctypes.c_int32), ("right", ctypes.c_int32),
                ("column", ctypes.c_char_p), ("literal", CLiteral)]


class CProjection(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("nodes", ctypes.POINTER(CProjNode)), ("n_nodes", ctypes.c_int32), ("root", ctypes.c_int32)]


class CPlanDesc(ctypes.Structure):
    _fields_ = [("filter", ctypes.POINTER(CExpr)), ("n_filter", ctypes.c_int32), ("filter_root", ctypes.c_int32),
                ("aggs", ctypes.POINTER(CAggregation)), ("n_aggs", ctypes.c_int32), ("n_groups", ctypes.c_int32),
                ("groups", ctypes.POINTER(CGroupExpr)), ("final_stage", ctypes.c_int32), ("n_projections", ctypes.c_int32),
                ("projections", ctypes.POINTER(CProjection)), ("regex_match", ctypes.c_void_p), ("regex_user", ctypes.c_void_p),
                ("ordered", ctypes.c_int32), ("_pad2", ctypes.c_int32)]


# fdb_regex_match_fn: int32 (*)(void* user, const char* pattern, int64 pattern_len, const uint8* value, int64 value_len)
REGEX_MATCH_FN = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_char), ctypes.c_int64,
                                  ctypes.POINTER(ctypes.c_char), ctypes.c_int64)


def regex_matcher(search: Callable[[bytes, bytes], bool]):
    """Wraps `search(pattern, value) -> bool` (unanchored; raise to reject the pattern) as the descriptor's host regex engine —
    what the Go shim does with regexp.Regexp.Match. Keep the returned object alive as long as plans use it (to_desc does)."""
    def fn(_user, pat, pat_len, val, val_len):
        try:
            return 1 if search(ctypes.string_at(pat, pat_len), ctypes.string_at(val, val_len) if val_len else b"") else 0
        except Exception:  # noqa: BLE001 — a pattern the engine rejects
            return -1
    return REGEX_MATCH_FN(fn)


@dataclass
class PlanDescHolder:
    """Owns a CPlanDesc and everything it points to."""
    desc: CPlanDesc
    keepalive: List[Any] = field(default_factory=list)

    def ptr(self):
        return ctypes.byref(self.desc)


def to_desc(filter_expr: Optional[Expr], aggs: Sequence[AggregationFunction], groups: Sequence[Column],
            final_stage: bool = False, regex=None, ordered: bool = False) -> PlanDescHolder:
    """`regex`: a `regex_matcher(...)` object — the host application's regular-expression engine (fdb_plan_desc.regex_match)."""
    keep: List[Any] = []
    nodes: List[CExpr] = []

    def visit(e: Expr) -> int:
        if not isinstance(e, BinaryExpr):
            raise TypeError("unsupported boolean expression")  # ≙ ErrUnsupportedBooleanExpression (filter.go:217-228)
        if e.op in (OP_AND, OP_OR):
            l = visit(e.left)
            r = visit(e.right)
            n = CExpr(op=e.op, left=l, right=r, column=None)
            nodes.append(n)
            return len(nodes) - 1
        if not isinstance(e.left, Column):
            raise TypeError("left side of binary expression must be a column")  # filter.go:91-93
        lit = _lit(e.right)
        cl = CLiteral(type=lit.lit_type())
        v = lit.value
        if cl.type in (LIT_INT64, LIT_BOOL):
            cl.i64 = int(v)
        elif cl.type == LIT_UINT64:
            cl.u64 = int(v)
        elif cl.type == LIT_FLOAT64:
            cl.f64 = float(v)
        elif cl.type in (LIT_STRING, LIT_BINARY):
            b = v.encode() if isinstance(v, str) else bytes(v)
            buf = ctypes.create_string_buffer(b, len(b) + 1)
            keep.append(buf)
            cl.data = ctypes.cast(buf, ctypes.c_char_p)
            cl.len = len(b)
        name = e.left.name.encode()
        keep.append(name)
        n = CExpr(op=e.op, left=-1, right=-1, column=name, literal=cl)
        nodes.append(n)
        return len(nodes) - 1

    d = CPlanDesc()
    if filter_expr is not None:
        root = visit(filter_expr)
        arr = (CExpr * len(nodes))(*nodes)
        keep.append(arr)
        d.filter = ctypes.cast(arr, ctypes.POINTER(CExpr))
        d.n_filter = len(nodes)
        d.filter_root = root
    else:
        d.filter = None
        d.n_filter = 0
        d.filter_root = -1
    if aggs:
        ca = (CAggregation * len(aggs))()
        for i, a in enumerate(aggs):
            nm = a.expr.name.encode()
            keep.append(nm)
            ca[i].func = a.func
            ca[i].dynamic = 1 if isinstance(a.expr, Column) and a.expr.dynamic else 0  # ≙ the DynamicColumn visitor of Aggregate(), aggregate.go:38-46
            ca[i].column = nm
        keep.append(ca)
        d.aggs = ctypes.cast(ca, ctypes.POINTER(CAggregation))
    d.n_aggs = len(aggs)
    if groups:
        cg = (CGroupExpr * len(groups))()
        for i, g in enumerate(groups):
            nm = g.name.encode()
            keep.append(nm)
            cg[i].name = nm
            cg[i].dynamic = 1 if g.dynamic else 0
        keep.append(cg)
        d.groups = ctypes.cast(cg, ctypes.POINTER(CGroupExpr))
    d.n_groups = len(groups)
    d.final_stage = 1 if final_stage else 0
    d.ordered = 1 if ordered else 0
    # computed columns: every aggregated expression / group expression that is arithmetic (or an alias of one) becomes a
    # projection named like the reference names it
    projs: List[CProjection] = []
    seen = set()

    def flatten(e, nodes: List[CProjNode]) -> int:
        if isinstance(e, AliasExpr):
            return flatten(e.expr, nodes)
        if isinstance(e, Column):
            nm = e.name.encode()
            keep.append(nm)
            nodes.append(CProjNode(kind=0, op=0, left=-1, right=-1, column=nm))
        elif isinstance(e, Literal):
            cl = CLiteral(type=e.lit_type())
            if cl.type == LIT_INT64:
                cl.i64 = int(e.value)
            elif cl.type == LIT_FLOAT64:
                cl.f64 = float(e.value)
            elif cl.type == LIT_UINT64:  # uint64 arithmetic (project.go:138-150): the scalar must be a uint64 too
                cl.u64 = int(e.value)
            elif cl.type in (LIT_STRING, LIT_BINARY):  # the right side of a comparison inside a boolean projection (`labels.a == 'x'` as a key, project.go:409-470)
                b = e.value.encode() if isinstance(e.value, str) else bytes(e.value)
                buf = ctypes.create_string_buffer(b, len(b) + 1)
                keep.append(buf)
                cl.data = ctypes.cast(buf, ctypes.c_char_p)
                cl.len = len(b)
            elif cl.type == LIT_NULL:
                pass
            else:
                raise TypeError(f"unsupported literal in a projection: {e.value!r}")
            nodes.append(CProjNode(kind=1, op=0, left=-1, right=-1, column=None, literal=cl))
        elif isinstance(e, BinaryExpr) and e.op in _ARITH:
            l = flatten(e.left, nodes)
            r = flatten(e.right, nodes)
            nodes.append(CProjNode(kind=2, op=e.op, left=l, right=r, column=None))
        elif isinstance(e, BinaryExpr) and (OP_EQ <= e.op <= OP_GT_EQ or e.op in (OP_AND, OP_OR)):  # boolean projection: `value > 0` as a distinct / group key
            l = flatten(e.left, nodes)
            r = flatten(e.right, nodes)
            nodes.append(CProjNode(kind=3, op=e.op, left=l, right=r, column=None))
        elif isinstance(e, ConvertExpr):
            if e.to not in ("float64", "double", "float"):
                raise TypeError(f"unsupported conversion to {e.to}")
            l = flatten(e.expr, nodes)
            nodes.append(CProjNode(kind=4, op=0, left=l, right=-1, column=None))
        elif isinstance(e, IsNullExpr):
            l = flatten(e.expr, nodes)
            nodes.append(CProjNode(kind=5, op=0, left=l, right=-1, column=None))
        elif isinstance(e, IfExpr):
            c = flatten(e.cond, nodes)
            l = flatten(e.then, nodes)
            r = flatten(e.els, nodes)
            nodes.append(CProjNode(kind=6, op=c, left=l, right=r, column=None))
        else:
            raise TypeError(f"unsupported expression in projection: {e}")
        return len(nodes) - 1

    for e in [a.expr for a in aggs] + list(groups):
        inner = e.expr if isinstance(e, AliasExpr) else e
        computed = isinstance(inner, (ConvertExpr, IsNullExpr, IfExpr)) or (
            isinstance(inner, BinaryExpr) and (inner.op in _ARITH or OP_EQ <= inner.op <= OP_GT_EQ or inner.op in (OP_AND, OP_OR)))
        if not computed or e.name in seen:
            continue
        seen.add(e.name)
        nodes: List[CProjNode] = []
        root = flatten(inner, nodes)
        arr = (CProjNode * len(nodes))(*nodes)
        nm = e.name.encode()
        keep += [arr, nm]
        projs.append(CProjection(name=nm, nodes=ctypes.cast(arr, ctypes.POINTER(CProjNode)), n_nodes=len(nodes), root=root))
    if projs:
        pa_ = (CProjection * len(projs))(*projs)
        keep.append(pa_)
        d.projections = ctypes.cast(pa_, ctypes.POINTER(CProjection))
    d.n_projections = len(projs)
    if regex is not None:
        keep.append(regex)
        d.regex_match = ctypes.cast(regex, ctypes.c_void_p)
    return PlanDescHolder(d, keep)
