#!/bin/bash
# poll the shader clock and socket power while a command runs:  tools/clock_poll.sh <logfile> -- <command...>
log=$1; shift; shift
( while true; do echo "$(date +%s.%N) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | sed 's/.*: //' | tr '\n' ' ')"; sleep 0.25; done ) > $log &
poll=$!
"$@"
kill $poll
