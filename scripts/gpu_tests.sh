#!/bin/bash
# usage (GPU box): scripts/gpu_tests.sh <log-file> [pytest arguments ...]     (default: the whole GPU suite)
# The GPU tests with every piece of abort evidence switched on: output not captured (-s: ROCr's "Memory access fault by GPU ..."
# line and glibc's fatal messages go to stderr, which pytest's capture would swallow and lose when the process dies), the
# library's device allocations logged (NVCA_ALLOC_LOG: a fault names an address), the abort shim preloaded (C call stack of
# the raising thread and its start routine, tests/san/abrt_trace.c).  The log is complete; the summary lines are echoed.
LOG=$1; shift
mkdir -p "$(dirname "$LOG")"
[ -f tests/san/build/abrt_trace.so ] || { mkdir -p tests/san/build; gcc -shared -fPIC -O1 -g -o tests/san/build/abrt_trace.so tests/san/abrt_trace.c -ldl -lpthread; }
if [ $# -eq 0 ]; then set -- tests -m gpu; fi
HSA_ENABLE_VM_FAULT_MESSAGE=1 LIBC_FATAL_STDERR_=1 NVCA_ALLOC_LOG=1 LD_PRELOAD=$PWD/tests/san/build/abrt_trace.so python3 -m pytest -s -x -q "$@" > "$LOG" 2>&1
rc=$?
echo "rc=$rc" >> "$LOG"
grep -n "Memory access fault\|HW Exception\|abrt_trace: SIGABRT\|start routine\| passed\| failed\|rc=" "$LOG" | tail -8
exit $rc
