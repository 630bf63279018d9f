"""Operator protocol (mirror of vinum/core/base.py:226-260): a linear chain of pull-based operators, each a
generator over record batches -- here the batches live in HBM."""
from typing import Dict, Iterable, List

import pyarrow as pa

from ..device import DeviceColumn


class DeviceRecordBatch:
    """Named HBM-resident columns of equal length (device counterpart of vinum/arrow/record_batch.py)."""

    def __init__(self, columns: Dict[str, DeviceColumn], num_rows: int = None):
        self.columns = dict(columns)
        self.num_rows = num_rows if num_rows is not None else (next(iter(columns.values())).length if columns else 0)

    @staticmethod
    def from_arrow(batch: pa.RecordBatch, dictionaries: Dict = None) -> "DeviceRecordBatch":
        """dictionaries (optional, name -> KeyDictionary, filled on demand): non-numeric columns (strings, bools, decimals)
        are dictionary-encoded with the running dictionary of their name and staged as int32 codes."""
        from ..device import is_supported
        cols = {}
        for i, n in enumerate(batch.schema.names):
            arr = batch.column(i)
            if dictionaries is not None and not is_supported(arr.type):
                from ..vinum_lib import KeyDictionary
                if n not in dictionaries:
                    dictionaries[n] = KeyDictionary(arr.type)
                cols[n] = DeviceColumn.from_arrow(arr, dictionary=dictionaries[n])
            else:
                cols[n] = DeviceColumn.from_arrow(arr)
        return DeviceRecordBatch(cols, batch.num_rows)

    @property
    def column_names(self) -> List[str]:
        return list(self.columns)

    def column(self, name: str) -> DeviceColumn:
        if name not in self.columns:
            raise ValueError(f'Column "{name}" is not found.')   # record_batch.py:74-75
        return self.columns[name]

    def slice(self, offset: int, length: int) -> "DeviceRecordBatch":
        """Zero-copy row range (vinum/arrow/record_batch.py:92-95): every column becomes a view with a larger Arrow offset."""
        return DeviceRecordBatch({n: c.slice(offset, length) for n, c in self.columns.items()}, length)

    def to_arrow(self) -> pa.RecordBatch:
        return pa.RecordBatch.from_arrays([c.to_arrow() for c in self.columns.values()], names=list(self.columns))


class Operator:
    """next() yields batches; subclasses override _kernel (base.py:254-267) or next()."""

    def __init__(self, parent_operator: "Operator" = None):
        self._parent_operator = parent_operator

    def next(self) -> Iterable[DeviceRecordBatch]:
        for batch in self._parent_operator.next():
            yield self._kernel(batch)

    def _kernel(self, batch: DeviceRecordBatch) -> DeviceRecordBatch:
        raise NotImplementedError
