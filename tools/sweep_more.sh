#!/bin/bash
# a look for cliffs beyond bench.py's shapes: the stream of 59 x 2^24-row batches (stream mode) and several input columns over G
echo "== stream (59 x 2^24 rows, stream mode), last line of tools/stream2.py G 2 1"
for g in 7 1e3 1e4 1e5 1e6 2e6 5e6 1e7 1e8; do echo -n "G=$g: "; python tools/stream2.py $g 2 1 2>&1 | tail -1 | cut -c1-200; done
echo "== several columns (5e8 rows): tools/manycol.py 5e8 G C, last line"
for c in 2 3 4; do for g in 1e3 1e4 1e5 1e6 2e6 5e6 1e7 1e8; do echo -n "C=$c G=$g: "; python tools/manycol.py 5e8 $g $c 2>&1 | tail -1 | cut -c1-200; done; done
