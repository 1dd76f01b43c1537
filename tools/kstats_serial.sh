#!/bin/bash
# kernel times without cross-stream contention: bench with every overlap switched off, under rocprofv3 --kernel-trace --stats
export EFTB_AP_OVERLAP=0 EFTB_PREP_OVERLAP=0
exec "$(dirname "$0")/kstats.sh" "$@"
