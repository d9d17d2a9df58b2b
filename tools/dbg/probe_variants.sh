#!/bin/bash
for v in 0 1 2 3 4; do echo "== PL_DBG=$v"; tools/dbg/probe.sh "-DPL_TIMERS -DPL_DBG=$v" 130944 1 | grep -E "us per|wg   0"; done
