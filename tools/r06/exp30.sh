#!/bin/bash
timeout 900 python tools/r06/sortedkeys.py 1e9 1e8 2>&1 | tail -4
timeout 900 python tools/r06/sortedkeys.py 1e9 1e6 2>&1 | tail -4
