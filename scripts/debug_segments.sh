#!/bin/bash
cd "$(dirname "$0")/.."
export DD_SEG_DEBUG=1
echo "=== train disp_init"; timeout 300 python scripts/debug_segments.py train disp_init 2>&1 | grep -E "segments\]|step|OK|Fatal|Error|error" | tail -24
echo "=== eval disp_init";  timeout 300 python scripts/debug_segments.py eval disp_init 2>&1 | grep -E "segments\]|step|OK|Fatal|Error|error" | tail -8
echo "=== train disp_init, separate pose passes"; DD_STOCK_POSE_PASSES=1 timeout 300 python scripts/debug_segments.py train disp_init 2>&1 | grep -E "segments\]|step|OK|Fatal|Error|error" | tail -8
echo "=== train fine_tune"; timeout 300 python scripts/debug_segments.py train fine_tune 2>&1 | grep -E "segments\]|step|OK|Fatal|Error|error" | tail -24
echo "=== train fine_tune nchw"; timeout 300 python scripts/debug_segments.py train fine_tune --nchw 2>&1 | grep -E "segments\]|step|OK|Fatal|Error|error" | tail -8
echo "=== native backtrace of the eval case"
timeout 900 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGUSR1 nostop noprint" -ex run -ex bt --args python scripts/debug_segments.py eval disp_init 2>&1 | grep -v "^\[New Thread\|^\[Thread\|warning:" | tail -60
