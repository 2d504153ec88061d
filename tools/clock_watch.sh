#!/bin/bash
# samples the shader clock and the socket power while a command runs: tools/clock_watch.sh out.txt -- cmd ...
out=$1; shift; shift
( while true; do /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.5; done ) > "$out" &
w=$!
"$@"
rc=$?
kill $w
exit $rc
