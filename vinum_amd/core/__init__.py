"""Physical operators mirroring vinum/core/{base,algebra,aggregate}.py with HBM-resident batches."""
from .base import DeviceRecordBatch, Operator  # noqa: F401
from .algebra import (  # noqa: F401
    FileReaderOperator, FilterOperator, MaterializeTableOperator, ProjectOperator, SliceOperator, SortOperator,
    TableReaderOperator,
)
from .aggregate import AggregateFunction, AggregateOperator  # noqa: F401
