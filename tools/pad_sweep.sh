#!/bin/bash
# GPU box: placement of the engine's streams on the hardware queues (EFTB_STREAM_PAD = extra streams in front of main, side, look-ahead, back, copy)
cd "$GRAFT_REPO_ROOT"
( echo ""; for p in "1,0,0,0,0" "0,1,0,0,0" "0,0,1,0,0" "0,0,0,1,0" "2,0,0,0,0" "0,2,0,0,0" "0,0,2,0,0" "0,0,0,2,0" "1,1,0,0,0" "0,1,1,0,0" "0,0,1,1,0" "1,0,1,0,0" "0,1,0,1,0" "3,0,0,0,0" "0,3,0,0,0" "0,0,0,3,0" "1,1,1,1,0" "2,1,0,1,0"; do echo "EFTB_STREAM_PAD=$p"; done; echo "" ) | K=${K:-200} bash tools/env_sweep.sh | awk 'NR % 2 == 1 || 1'
