#!/bin/bash
set -u
for cfg in lego fox; do timeout 300 python tools/bench_march.py --config $cfg 2>/dev/null | tail -1 | cut -c1-330; done
NGP_MARCH_COUNT=coop timeout 300 python tools/bench_march.py --config fox 2>/dev/null | tail -1 | cut -c1-330
