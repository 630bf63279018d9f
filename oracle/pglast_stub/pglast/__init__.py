"""Import stub for the `pglast` SQL parser (pinned 1.17 by the reference's setup.py:35; absent from this image, no
network).  TEST INFRASTRUCTURE ONLY: it exists so that `import vinum` succeeds in the build container
(vinum/__init__.py:3-17 hard-requires the package; vinum/parser/parser.py:5-6 imports these names) when
tests/golden/gen_golden_planner.py drives the REAL planner / executor with hand-built Query ASTs
(vinum/parser/query.py:408-432).  Nothing is parsed: parse_sql raises.  SURVEY.md §8(c) item 3."""


class Node:  # pglast.Node: only referenced in type positions / isinstance checks of the parser
    pass


def parse_sql(sql):
    raise NotImplementedError("pglast is not installed: build Query objects directly (vinum/parser/query.py)")
