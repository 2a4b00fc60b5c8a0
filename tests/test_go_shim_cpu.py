"""What can be checked of the Go drop-in without a Go toolchain (there is none in this image).

integration/go/gpuplan/operator.go is source a FrostDB maintainer compiles; these tests remove the classes of error a compiler
would catch and that five rounds of blind edits could have introduced: an import cycle between physicalplan and gpuplan, cgo
identifiers / struct fields the header does not declare, reference identifiers that do not exist, unbalanced syntax, and a
Build patch that does not apply to the reference's physicalplan.go.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "integration", "go", "gpuplan", "operator.go")
PATCH = os.path.join(ROOT, "integration", "go", "patches", "physicalplan_operator_factory.diff")
HEADER = os.path.join(ROOT, "include", "frostdb_amd.h")
REF_PP = "/root/reference/query/physicalplan/physicalplan.go"


def go_code(text):
    """Go source with comments, the cgo preamble, string / rune / raw-string literals blanked (same length, newlines kept)."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if text.startswith("//", i):
            j = text.find("\n", i)
            j = n if j < 0 else j
            out.append(" " * (j - i)); i = j
        elif text.startswith("/*", i):
            j = text.index("*/", i) + 2
            out.append("".join(ch if ch == "\n" else " " for ch in text[i:j])); i = j
        elif c == '"':
            j = i + 1
            while text[j] != '"':
                j += 2 if text[j] == "\\" else 1
            out.append('"' + " " * (j - i - 1) + '"'); i = j + 1
        elif c == "`":
            j = text.index("`", i + 1)
            out.append("`" + "".join(ch if ch == "\n" else " " for ch in text[i + 1:j]) + "`"); i = j + 1
        elif c == "'":
            j = i + 1
            while text[j] != "'":
                j += 2 if text[j] == "\\" else 1
            out.append("'" + " " * (j - i - 1) + "'"); i = j + 1
        else:
            out.append(c); i += 1
    return "".join(out)


def header_text():
    return re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)


def header_structs():
    """struct name → set of field names, from the typedefs of include/frostdb_amd.h."""
    structs = {}
    for body, name in re.findall(r"typedef struct \w+ \{(.*?)\} (\w+);", header_text(), flags=re.S):
        fields = set()
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            m = re.search(r"(\w+)\s*(?:\[[^\]]*\])?$", decl)
            if m:
                fields.add(m.group(1))
        structs[name] = fields
    return structs


def test_shim_syntax_is_balanced():
    code = go_code(open(SHIM).read())
    stack, pairs = [], {")": "(", "]": "[", "}": "{"}
    for ln, line in enumerate(code.split("\n"), 1):
        for ch in line:
            if ch in "([{":
                stack.append((ch, ln))
            elif ch in ")]}":
                assert stack and stack[-1][0] == pairs[ch], f"unbalanced {ch!r} at line {ln}"
                stack.pop()
    assert not stack, stack[-3:]
    # every func / type at column 0 closes at column 0 (a cheap structural check of the hand-edited file)
    assert code.count("\nfunc ") == len(re.findall(r"^func .*\{\s*$|^func .*\}\s*$", code, flags=re.M)), "a func header spans lines or lacks its brace"


def test_every_cgo_identifier_exists_in_the_header():
    code = go_code(open(SHIM).read())
    hdr = header_text()
    declared_fns = set(re.findall(r"\b(fdb_[a-z_0-9]+)\s*\(", hdr))
    declared_types = set(re.findall(r"typedef (?:struct|enum) \w+(?: \{.*?\})? (\w+);", hdr, flags=re.S)) | {"fdb_plan", "fdb_batch", "fdb_comm"}
    declared_consts = set(re.findall(r"\b(FDB_[A-Z_0-9]+)\b", hdr))
    declared_types |= {"fdb_regex_match_fn"}
    preamble = {"fdbRegexMatchFn", "fdbRegexMatchC", "fdbRegexMatch"}
    stdlib = {"int", "int32_t", "int64_t", "uint8_t", "uint64_t", "double", "char", "size_t", "calloc", "malloc", "free", "CString", "GoString", "GoStringN",
              "struct_ArrowArray", "struct_ArrowSchema"}
    unknown = []
    for ident in sorted(set(re.findall(r"\bC\.(\w+)", code))):
        if ident in stdlib or ident in preamble:
            continue
        if ident in declared_fns or ident in declared_types or ident in declared_consts:
            continue
        unknown.append(ident)
    assert not unknown, unknown
    # called with the right NUMBER of arguments (the one arity check a regex can do: top-level commas of the call)
    protos = {m.group(1): m.group(2) for m in re.finditer(r"\b(fdb_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S)}
    for m in re.finditer(r"\bC\.(fdb_[a-z_0-9]+)\(", code):
        name, i, depth, commas, empty = m.group(1), m.end(), 1, 0, True
        while depth:
            ch = code[i]
            depth += ch in "([{"
            depth -= ch in ")]}"
            commas += ch == "," and depth == 1
            empty = empty and (ch.isspace() or depth == 0)
            i += 1
        got = 0 if empty else commas + 1
        params = protos[name].strip()
        want = 0 if params in ("", "void") else params.count(",") + 1
        assert got == want, f"C.{name}: {got} arguments, the header declares {want}"


def test_every_struct_field_of_a_c_literal_exists_in_the_header():
    code = go_code(open(SHIM).read())
    structs = header_structs()
    checked = 0
    for m in re.finditer(r"\bC\.(fdb_[a-z_]+)\{", code):
        name, i, depth = m.group(1), m.end(), 1
        j = i
        while depth:
            depth += code[j] == "{"
            depth -= code[j] == "}"
            j += 1
        body = code[i:j - 1]
        for key in re.findall(r"(?:^|[,{\s])(\w+):", body):
            field = key[1:] if key in ("_type", "_func") else key  # cgo spells C fields that are Go keywords with a leading underscore
            assert field in structs[name], f"C.{name} has no field {field!r}"
            checked += 1
    # plain field accesses on values of known struct type (the shim's own naming: dst = *fdb_literal, aggs[i], groups[i], n = fdb_expr / fdb_proj_node)
    for var, struct in (("dst", "fdb_literal"), (r"aggs\[i\]", "fdb_aggregation"), (r"groups\[i\]", "fdb_group_expr"), ("desc", "fdb_plan_desc")):
        for key in re.findall(r"\b" + var + r"\.(\w+)\b", code):
            field = key[1:] if key in ("_type", "_func") else key
            assert field in structs[struct], f"{struct} has no field {field!r}"
            checked += 1
    assert checked > 30


def test_no_import_cycle_between_physicalplan_and_gpuplan():
    src = open(SHIM).read()
    assert '"github.com/polarsignals/frostdb/query/physicalplan"' in src
    added = [ln[1:] for ln in open(PATCH).read().splitlines() if ln.startswith("+") and not ln.startswith("+++")]
    assert added, "empty patch"
    assert not [ln for ln in added if "gpuplan" in ln], "the Build patch must not know gpuplan: physicalplan → gpuplan → physicalplan is an import cycle"
    # INTEGRATION.md shows the same patch, not a Build that calls into gpuplan
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "gpuplan.New(" not in doc and "physicalplan.WithOperatorFactory(gpuplan.Factory(" in doc


def test_reference_identifiers_the_shim_uses_exist():
    """physicalplan.X / logicalplan.X used by the shim: defined by the reference (when its tree is here) or added by the patch."""
    code = go_code(open(SHIM).read())
    added = "\n".join(ln[1:] for ln in open(PATCH).read().splitlines() if ln.startswith("+") and not ln.startswith("+++"))
    from_patch = set(re.findall(r"^(?:type|func|var) (\w+)", added, flags=re.M))
    assert {"FusedStage", "OperatorFactory", "ErrOperatorNotFused", "WithOperatorFactory"} <= from_patch
    used_pp = set(re.findall(r"\bphysicalplan\.(\w+)", code))
    used_lp = set(re.findall(r"\blogicalplan\.(\w+)", code))
    # fields of FusedStage the factory reads
    stage_fields = set(re.findall(r"^\t(\w+)\s", re.search(r"type FusedStage struct \{(.*?)\n\}", added, flags=re.S).group(1), flags=re.M))
    for f in set(re.findall(r"\bst\.(\w+)", code)):
        assert f in stage_fields, f"FusedStage has no field {f}"
    if not os.path.isdir("/root/reference/query"):
        pytest.skip("reference tree not present on this host")

    def defined(pkg_dir):
        names = set()
        for fn in os.listdir(pkg_dir):
            if fn.endswith(".go") and not fn.endswith("_test.go"):
                text = open(os.path.join(pkg_dir, fn)).read()
                names |= set(re.findall(r"^(?:type|func|var|const) (\w+)", text, flags=re.M))
                for block in re.findall(r"^(?:const|var) \((.*?)^\)", text, flags=re.M | re.S):
                    names |= set(re.findall(r"^\t(\w+)", block, flags=re.M))
        return names
    pp = defined("/root/reference/query/physicalplan") | from_patch
    lp = defined("/root/reference/query/logicalplan")
    assert not (used_pp - pp), used_pp - pp
    assert not (used_lp - lp), used_lp - lp


@pytest.mark.skipif(not (os.path.exists(REF_PP) and shutil.which("patch")), reason="needs the reference tree and patch(1)")
def test_build_patch_applies_to_the_reference(tmp_path):
    work = tmp_path / "physicalplan.go"
    shutil.copy(REF_PP, work)
    r = subprocess.run(["patch", "-p3", "--fuzz=0", str(work)], stdin=open(PATCH), capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    text = open(work).read()
    code = go_code(text)
    assert code.count("{") == code.count("}") and code.count("(") == code.count(")")
    assert "gpuplan" not in text
    # the cases Build consults the factory in
    assert text.count("tryFactory(FusedStage{") == 2 and "flushPending()" in text
