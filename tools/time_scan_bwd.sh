#!/bin/bash
# rocprofv3 per-kernel median/min durations of the backward scan at the headline shape (run on the GPU box)
# usage: tools/time_scan_bwd.sh <tag> [launches] [lib.so]
PROF_SCRIPT=tools/prof_scan_bwd.py exec bash "$(dirname "$0")/time_scan_fwd.sh" "$@"
