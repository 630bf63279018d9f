"""pglast.enums names imported by vinum/parser/parser.py:6 (never evaluated without a parse tree)."""
import enum


class A_Expr_Kind(enum.IntEnum):
    AEXPR_OP = 0
    AEXPR_OP_ANY = 1
    AEXPR_OP_ALL = 2
    AEXPR_DISTINCT = 3
    AEXPR_NOT_DISTINCT = 4
    AEXPR_NULLIF = 5
    AEXPR_OF = 6
    AEXPR_IN = 7
    AEXPR_LIKE = 8
    AEXPR_ILIKE = 9
    AEXPR_SIMILAR = 10
    AEXPR_BETWEEN = 11
    AEXPR_NOT_BETWEEN = 12
    AEXPR_BETWEEN_SYM = 13
    AEXPR_NOT_BETWEEN_SYM = 14
    AEXPR_PAREN = 15


class BoolExprType(enum.IntEnum):
    AND_EXPR = 0
    OR_EXPR = 1
    NOT_EXPR = 2
