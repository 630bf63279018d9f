"""Seam B2: the reference's Python-level filter / projection dispatch, bound to the device kernels.

In the reference, filter and projection never cross into native code: the planner turns every SQL
expression into a tree of `VectorizedExpression` objects (vinum/core/base.py:89-125) whose `_function` comes from the
registry `EXPRESSION_FUNCTIONS: {SQLExpression: (callable, FunctionType)}` (vinum/core/expressions.py:12-52), and
`Operator.next()` evaluates that tree node by node with NumPy / pyarrow.compute before `FilterOperator._kernel` calls
`RecordBatch.filter(mask)` (vinum/core/algebra.py:108-123, vinum/arrow/record_batch.py:85-90).

This module is what a Vinum maintainer adds to run those operators on the GPU WITHOUT touching the planner:

  * `lower(tree, registry)` reads a VectorizedExpression-SHAPED tree -- anything with the attributes the reference's
    classes have (`arguments`, `_function`, `_is_binary_func`; `Column.get_column_name()`; `Literal.value`;
    `AggregateFunction.get_agg_func_name()` / `get_input_column_name()`) -- identifies each node's SQL operator by the
    identity of its callable in the registry, honours the left fold of binary operators (base.py:145-151) and returns the
    prefix form `vnm_project` / `vnm_filter_*` programs are compiled from (vinum_amd.ops.compile_expr).
  * `device_filter(batch, predicate)` is the `RecordBatch.filter` replacement: predicate tree -> ONE fused mask kernel ->
    one compaction pass over every column (or `vnm_filter_cmp` alone for `column <op> literal`), NULL mask entries and
    NaN comparisons as record_batch.py:85-90,112-118.
  * `GpuFilterOperator` / `GpuProjectOperator` take the reference's constructor arguments (a VectorizedExpression
    predicate; a list of Column / Literal / VectorizedExpression arguments + col_names + keep_input_table) and implement
    the reference's `next()` generator protocol over HBM-resident batches.
  * `install(vinum_pkg)` rebinds the operator names the planner instantiates (vinum/planner/planner.py:16-31) to the GPU
    operators of this package; `QueryPlanner.plan_query` itself stays untouched.

`SQLExpression`, `EXPRESSION_FUNCTIONS`, `Column`, `Literal`, `VectorizedExpression`, `AggregateFunction` below are this
package's own mirror of that contract (same member names, the same third-party callables) -- the reference does not
travel to the GPU box, so tests build trees with the mirror and, in the build container, with the reference's own
planner; both must lower to the same programs.
"""
import enum
from functools import partial
from typing import Any, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc

from . import ops
from .core.base import DeviceRecordBatch, Operator


# ---- mirror of the expression contract -----------------------------------------------------------------------
class SQLExpression(enum.Enum):   # vinum/parser/query.py:17-59 (the members the hot path uses)
    ADDITION = enum.auto(); SUBTRACTION = enum.auto(); MULTIPLICATION = enum.auto(); DIVISION = enum.auto()
    MODULUS = enum.auto(); NEGATION = enum.auto(); BINARY_NOT = enum.auto(); BINARY_AND = enum.auto()
    BINARY_OR = enum.auto(); BINARY_XOR = enum.auto()
    EQUALS = enum.auto(); NOT_EQUALS = enum.auto(); GREATER_THAN = enum.auto(); GREATER_THAN_OR_EQUAL = enum.auto()
    LESS_THAN = enum.auto(); LESS_THAN_OR_EQUAL = enum.auto()
    AND = enum.auto(); BETWEEN = enum.auto(); NOT_BETWEEN = enum.auto(); IN = enum.auto(); NOT_IN = enum.auto()
    NOT = enum.auto(); OR = enum.auto(); IS_NULL = enum.auto(); IS_NOT_NULL = enum.auto()
    FUNCTION = enum.auto()


class FunctionType(enum.Enum):    # vinum/core/functions.py:347-350
    ARROW = enum.auto()
    NUMPY = enum.auto()
    CLASS = enum.auto()


# the same third-party callables the reference registers (vinum/core/expressions.py:12-52): NumPy ufuncs, comparison
# lambdas, pyarrow.compute kernels
EXPRESSION_FUNCTIONS = {
    SQLExpression.NEGATION: (np.negative, FunctionType.NUMPY),
    SQLExpression.BINARY_NOT: (lambda x: ~x, FunctionType.NUMPY),
    SQLExpression.BINARY_AND: (np.bitwise_and, FunctionType.NUMPY),
    SQLExpression.BINARY_OR: (np.bitwise_or, FunctionType.NUMPY),
    SQLExpression.BINARY_XOR: (np.bitwise_xor, FunctionType.NUMPY),
    SQLExpression.ADDITION: (np.add, FunctionType.NUMPY),
    SQLExpression.SUBTRACTION: (np.subtract, FunctionType.NUMPY),
    SQLExpression.MULTIPLICATION: (np.multiply, FunctionType.NUMPY),
    SQLExpression.DIVISION: (np.divide, FunctionType.NUMPY),
    SQLExpression.MODULUS: (np.mod, FunctionType.NUMPY),
    SQLExpression.AND: (pc.and_, FunctionType.ARROW),
    SQLExpression.OR: (pc.or_, FunctionType.ARROW),
    SQLExpression.NOT: (pc.invert, FunctionType.ARROW),
    SQLExpression.EQUALS: (lambda x, y: x == y, FunctionType.NUMPY),
    SQLExpression.NOT_EQUALS: (lambda x, y: x != y, FunctionType.NUMPY),
    SQLExpression.GREATER_THAN: (lambda x, y: x > y, FunctionType.NUMPY),
    SQLExpression.GREATER_THAN_OR_EQUAL: (lambda x, y: x >= y, FunctionType.NUMPY),
    SQLExpression.LESS_THAN: (lambda x, y: x < y, FunctionType.NUMPY),
    SQLExpression.LESS_THAN_OR_EQUAL: (lambda x, y: x <= y, FunctionType.NUMPY),
    SQLExpression.IS_NULL: (pc.is_null, FunctionType.ARROW),
    SQLExpression.IS_NOT_NULL: (pc.is_valid, FunctionType.ARROW),
    SQLExpression.IN: (np.isin, FunctionType.NUMPY),
    SQLExpression.NOT_IN: (partial(np.isin, invert=True), FunctionType.NUMPY),
    SQLExpression.BETWEEN: (lambda x, low, high: np.logical_and(x >= low, x <= high), FunctionType.NUMPY),
    SQLExpression.NOT_BETWEEN: (lambda x, low, high: np.logical_or(x < low, x > high), FunctionType.NUMPY),
}
BINARY_EXPRESSIONS = {SQLExpression.ADDITION, SQLExpression.SUBTRACTION, SQLExpression.MULTIPLICATION, SQLExpression.DIVISION,
                      SQLExpression.MODULUS, SQLExpression.AND, SQLExpression.OR, SQLExpression.EQUALS, SQLExpression.NOT_EQUALS,
                      SQLExpression.GREATER_THAN, SQLExpression.GREATER_THAN_OR_EQUAL, SQLExpression.LESS_THAN,
                      SQLExpression.LESS_THAN_OR_EQUAL}       # vinum/core/expressions.py:55-69


class Column:
    def __init__(self, name: str):
        self._name = name

    def get_column_name(self) -> str:
        return self._name


class Literal:
    def __init__(self, value: Any):
        self._value = value

    @property
    def value(self):
        return self._value


class VectorizedExpression:
    """Shape of vinum/core/base.py:89-125: arguments + the registry callable + the two dispatch flags."""

    def __init__(self, arguments: Iterable, function=None, is_numpy_func: bool = False, is_binary_func: bool = False):
        self._arguments = tuple(arguments)
        self._function = function
        self._is_numpy_function = is_numpy_func
        self._is_binary_func = is_binary_func

    @property
    def arguments(self):
        return self._arguments


class AggregateFunction(VectorizedExpression):
    """Shape of vinum/core/aggregate.py:12-29."""

    def __init__(self, func: str, column: Optional[Column] = None):
        super().__init__([])
        self._func = func
        self._input_column_name = column.get_column_name() if column else ""

    def get_input_column_name(self) -> str:
        return self._input_column_name

    def get_agg_func_name(self) -> str:
        return self._func.upper()


_SPEC_TO_SQL = {"add": "ADDITION", "sub": "SUBTRACTION", "mul": "MULTIPLICATION", "div": "DIVISION", "mod": "MODULUS",
                "neg": "NEGATION", "bnot": "BINARY_NOT", "band": "BINARY_AND", "bor": "BINARY_OR", "bxor": "BINARY_XOR",
                "eq": "EQUALS", "ne": "NOT_EQUALS", "gt": "GREATER_THAN", "ge": "GREATER_THAN_OR_EQUAL", "lt": "LESS_THAN",
                "le": "LESS_THAN_OR_EQUAL", "and": "AND", "or": "OR", "not": "NOT", "is_null": "IS_NULL",
                "is_not_null": "IS_NOT_NULL", "in": "IN", "not_in": "NOT_IN", "between": "BETWEEN",
                "not_between": "NOT_BETWEEN"}
_SQL_TO_SPEC = {v: k for k, v in _SPEC_TO_SQL.items()}


def vectorize(expr, registry=None, classes=None):
    """Build the VectorizedExpression tree the planner builds for `expr` (QueryPlanner._process_expressions_tree,
    vinum/planner/planner.py:140-222) from the prefix form -- with this module's mirror classes by default, or with the
    reference's own (classes = (Column, Literal, VectorizedExpression, AggregateFunction, SQLExpression, FunctionType,
    BINARY_EXPRESSIONS), registry = its EXPRESSION_FUNCTIONS) in the build container."""
    registry = registry or EXPRESSION_FUNCTIONS
    C, Lt, VE, AF, SQL, FT, BIN = classes or (Column, Literal, VectorizedExpression, AggregateFunction, SQLExpression,
                                             FunctionType, BINARY_EXPRESSIONS)

    def build(e):
        if isinstance(e, str):
            return C(e)
        if isinstance(e, (int, float)):
            return Lt(e)
        op, args = e[0], e[1:]
        if op == "fn":
            arg = build(args[1]) if len(args) > 1 else None
            return AF(args[0], arg) if arg is not None else AF(args[0])
        member = SQL[_SPEC_TO_SQL[op]]
        fn, ftype = registry[member]
        if op in ("in", "not_in"):
            built = [build(args[0]), Lt(list(args[1]))]          # parser.py:151-160: ONE Literal holding the list
        else:
            built = [build(a) for a in args]
        return VE(built, function=fn, is_numpy_func=(ftype == FT.NUMPY), is_binary_func=(member in BIN))
    return build(expr)


# ---- the adapter ---------------------------------------------------------------------------------------------------
def registry_index(expression_functions) -> Dict[int, str]:
    """id(callable) -> SQL operator name, for any registry with the reference's layout."""
    return {id(fn): member.name for member, (fn, _ftype) in expression_functions.items()}


_DEFAULT_INDEX = registry_index(EXPRESSION_FUNCTIONS)


def lower(node, index: Optional[Dict[int, str]] = None):
    """VectorizedExpression-shaped tree -> prefix expression (str | number | tuple) for vinum_amd.ops."""
    index = index or _DEFAULT_INDEX
    if node is None:
        return None
    if isinstance(node, (bool, int, float, str)):
        return node
    if hasattr(node, "get_agg_func_name"):                     # AggregateFunction (vinum/core/aggregate.py:12-29)
        name = node.get_agg_func_name().lower()
        col = node.get_input_column_name()
        return ("fn", name, col) if col else ("fn", name)
    if hasattr(node, "arguments") and hasattr(node, "_function"):
        fn = node._function
        if fn is None or id(fn) not in index:
            raise NotImplementedError(f"no GPU lowering for {fn!r} (UDFs, string and datetime functions stay on the CPU path)")
        sql = index[id(fn)]
        op = _SQL_TO_SPEC[sql]
        args = [lower(a, index) for a in node.arguments]
        if op in ("in", "not_in"):
            vals = args[1]
            return (op, args[0], tuple(vals) if isinstance(vals, (list, tuple, np.ndarray)) else (vals,))
        if getattr(node, "_is_binary_func", False) and len(args) > 2:
            return (op,) + tuple(args)                         # left fold (base.py:145-151) == the emitter's n-ary chain
        return (op,) + tuple(args)
    if hasattr(node, "value"):                                 # Literal (vinum/parser/query.py:126-176)
        v = node.value
        return list(v) if isinstance(v, (list, tuple)) else v
    if hasattr(node, "get_column_name"):                       # Column (:179-233)
        return node.get_column_name()
    raise TypeError(f"Unsupported OperatorArgument type: {type(node)}")     # base.py:52-54


def device_filter(batch: DeviceRecordBatch, predicate) -> DeviceRecordBatch:
    """`RecordBatch.filter(mask)` (vinum/arrow/record_batch.py:85-90) with the mask still an expression: the predicate
    tree becomes ONE fused byte-mask kernel and every column is compacted in one more pass; `column <op> literal` is a
    single fused compare + compact kernel.  NULL operands compare as NaN (record_batch.py:112-118)."""
    from .core.algebra import FilterOperator
    from .planner import _simple_predicate, _t
    expr = _t(predicate) if isinstance(predicate, (tuple, list)) else lower(predicate)
    simple = _simple_predicate(expr)
    return FilterOperator(simple if simple is not None else expr, None)._kernel(batch)


class GpuFilterOperator(Operator):
    """FilterOperator(predicate: VectorizedExpression, parent_operator) -- vinum/core/algebra.py:108-123."""

    def __init__(self, predicate, parent_operator, index: Optional[Dict[int, str]] = None):
        super().__init__(parent_operator)
        self.predicate = lower(predicate, index)

    def _kernel(self, batch: DeviceRecordBatch) -> DeviceRecordBatch:
        return device_filter(batch, self.predicate)


class GpuProjectOperator(Operator):
    """ProjectOperator(arguments, parent_operator, col_names=None, keep_input_table=False) -- algebra.py:28-105.
    Arguments are Column / Literal / VectorizedExpression objects; every computed expression of the list goes into one
    fused kernel (vnm_project_multi)."""

    def __init__(self, arguments, parent_operator, col_names: Optional[Sequence[str]] = None, keep_input_table: bool = False,
                 index: Optional[Dict[int, str]] = None):
        super().__init__(parent_operator)
        from .core.algebra import ProjectOperator
        args = list(arguments)
        exprs = [lower(a, index) for a in args]
        if col_names is None:                                   # algebra.py:66-75: the arguments name their columns
            col_names = [a.get_column_name() if hasattr(a, "get_column_name") else f"expr_{i}" for i, a in enumerate(args)]
        self._inner = ProjectOperator(exprs, None, col_names=list(col_names), keep_input_table=keep_input_table)
        self.expressions = exprs

    def _kernel(self, batch: DeviceRecordBatch) -> DeviceRecordBatch:
        return self._inner._kernel(batch)


def install(vinum_pkg) -> None:
    """Rebind the operator classes the reference's planner instantiates (vinum/planner/planner.py:16-31 imports them
    into its namespace) to GPU operators.  After this, QueryPlanner.plan_query builds a plan of GPU operators for
    numeric hot-path queries; nothing in the planner, binder or executor is edited.  (sys.modules['vinum_lib'] =
    vinum_amd.vinum_lib covers the native seam B1, see INTEGRATION.md.)"""
    planner_mod = __import__(vinum_pkg.__name__ + ".planner.planner", fromlist=["x"])
    expr_mod = __import__(vinum_pkg.__name__ + ".core.expressions", fromlist=["x"])
    index = registry_index(expr_mod.EXPRESSION_FUNCTIONS)

    class _Filter(GpuFilterOperator):
        def __init__(self, predicate, parent_operator):
            super().__init__(predicate, parent_operator, index=index)

    class _Project(GpuProjectOperator):
        def __init__(self, arguments, parent_operator, col_names=None, keep_input_table=False):
            super().__init__(arguments, parent_operator, col_names=col_names, keep_input_table=keep_input_table, index=index)

    planner_mod.FilterOperator = _Filter
    planner_mod.ProjectOperator = _Project
    planner_mod._vinum_amd_index = index
