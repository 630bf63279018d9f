"""Runs LAST (file name): every route DESIGN.md section 4 names must have been taken at least once by the GPU tests of this process.

The library leaves a note per operator call that commits a batch to a path -- route name + the reason in numbers (route_note,
vnm_route_counts / vnm_route_last; VNM_AGG_TRACE=1 prints them).  A path nobody's test reaches any more (a threshold moved, a
fallback edge shadowed by a new rule) shows up here instead of in a later round's bug report (VERDICT r04 "next" #9)."""
import ctypes

import pytest

pytestmark = pytest.mark.gpu

# route -> where DESIGN.md describes it
REQUIRED = {
    "onegroup:register_scan": "4.1", "onegroup:lds_scan": "4.1",
    "scan:hot": "4.1", "scan:hot_segments": "4.2a", "scan:hot_generic": "4.1", "scan:hot_two_columns": "4.1", "scan:hot_nullable_value": "4.1",
    "scan:hotn": "4.2c", "scan:lds_generic": "4.1", "scan:wide_keys": "4.2b", "scan:spilled_entries": "4.2",
    "dense_scan:hot": "4.1", "dense_scan:generic": "4.1", "dense:two_levels": "4.2", "dense:32_partitions": "4.2", "dense:stream_segments": "4.2a", "dense:nullable_key": "4.2",
    "dense:split_final": "4.2", "dense:generic": "4.2", "dense:generic_split_final": "4.2", "dense:nullable_value": "4.2", "dense:two_values": "4.2c",
    "hash_partitions:hot": "4.2", "hash_partitions:rings": "4.2", "hash_partitions:rings_failed": "4.2", "hash_partitions:wide_entries": "4.2c", "hash_partitions:generic": "4.2",
    "keys:packed": "4.2b", "keys:packed_with_dictionary_fields": "4.2b", "keys:tuple_dictionary": "4.2b",
    "split_program:small_range_per_column": "4.2c", "split_program:dense_per_column": "4.2c", "split_program:many_columns": "4.2c",
    "split_program:few_groups_many_columns": "4.2c", "split_program:batch": "4.2c",
    "stream:segments_of_one_launch": "4.2a", "stream:batches_singly": "4.2a",
    "result:fused_columns": "4.2", "result:fused_columns_with_side_table": "4.2b'", "result:finish_then_finalize": "3",
    "dense:count_bytes": "4.2d", "dense:fixed_point": "4.2e", "dense:fixed_point_misfit": "4.2e", "dense:exact_adds": "4.2e", "dense:fixed_point_columns": "4.2f", "dense:fixed_point_columns_failed": "4.2f",
    "minmax:ordered_mode": "2", "minmax:compose_prefix_suffix": "2",
    "sort:topk_threshold": "4.3", "sort:sample_sort": "4.3", "sort:sample_sort_words": "4.3", "sort:already_sorted": "4.3", "sort:lsd_radix": "4.3", "sort:string_key_ranks": "4.3",
}


def _counts():
    from vinum_amd import _lib as L
    lib = L.lib()
    need = lib.vnm_route_counts(None, 0)
    buf = ctypes.create_string_buffer(int(need) + 16)
    lib.vnm_route_counts(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        k, _, v = line.rpartition("=")
        out[k] = int(v)
    return out


def test_every_route_of_design_section_4_was_taken(capsys):
    counts = _counts()
    total = sum(counts.values())
    with capsys.disabled():
        print(f"\n[route coverage] {len(counts)} routes, {total} notes: " + ", ".join(f"{k}={v}" for k, v in sorted(counts.items())))
    if total < 3000:
        pytest.skip(f"only {total} route notes in this process: the coverage check needs the whole GPU suite (python -m pytest tests -m gpu)")
    missing = sorted(r for r in REQUIRED if counts.get(r, 0) == 0)
    assert not missing, f"routes of DESIGN.md no GPU test took in this run: {missing}"


def test_last_route_names_the_reason():
    import numpy as np
    import pyarrow as pa
    from oracle import oracle as O
    from vinum_amd import _lib as L
    from tests.test_gpu_agg import gpu_aggregate
    t = pa.table({"k": pa.array(np.arange(1000, dtype=np.int64) % 7), "v": pa.array(np.ones(1000))})
    gpu_aggregate(O.SINGLE, ["k"], ["k"], [(O.SUM, "v", "s")], t.to_batches())
    buf = ctypes.create_string_buffer(600)
    L.lib().vnm_route_last(buf, len(buf))
    text = buf.value.decode()
    assert text.startswith("result:") and ":" in text, text
