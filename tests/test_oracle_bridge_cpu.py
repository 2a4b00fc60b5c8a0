"""oracle/_bridge.py lays the descriptor structs out by hand (struct.pack) so that it shares no code with the product's ctypes
mirrors; this checks its sizes and field offsets against what the C compiler makes of include/frostdb_amd.h, and that the oracle
package imports nothing from the product."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hand_packed_layouts_match_the_header():
    from oracle import _bridge as b
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "frostdb_amd.h"
    int main(void) {
      printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(fdb_literal), sizeof(fdb_expr), sizeof(fdb_aggregation), sizeof(fdb_group_expr),
             sizeof(fdb_proj_node), sizeof(fdb_projection), sizeof(fdb_plan_desc));
      printf("%zu %zu %zu %zu %zu %zu\n", offsetof(fdb_literal, data), offsetof(fdb_expr, literal), offsetof(fdb_proj_node, literal),
             offsetof(fdb_plan_desc, groups), offsetof(fdb_plan_desc, projections), offsetof(fdb_plan_desc, ordered));
      printf("%zu %zu %zu %zu\n", sizeof(struct ArrowArray), sizeof(struct ArrowSchema), offsetof(struct ArrowArray, release), offsetof(struct ArrowSchema, release));
      return 0;
    }
    '''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        lines = [[int(x) for x in ln.split()] for ln in subprocess.check_output([exe]).decode().splitlines()]
    assert lines[0] == [b.LITERAL.size, b.EXPR.size + b.LITERAL.size, b.AGG.size, b.GROUP.size, b.PROJ_NODE.size + b.LITERAL.size,
                        b.PROJECTION.size, b.PLAN_DESC.size]
    assert lines[1] == [32, b.EXPR.size, b.PROJ_NODE.size, 32, 48, 72]
    assert lines[2] == [80, 72, 64, 56]


def test_oracle_package_does_not_import_the_product():
    code = "import sys; import oracle; import oracle._bridge; bad = [m for m in sys.modules if m.startswith('frostdb_amd')]; assert not bad, bad"
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)
