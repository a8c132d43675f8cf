#!/bin/bash
# block until the queued gpurun call has taken its snapshot of /root/repo (in_flight with a positive elapsed time), so that
# further edits cannot race the push
for i in $(seq 1 400); do
  s=$(/usr/local/graft/bin/gpurun --status 2>/dev/null)
  if echo "$s" | grep -q '"in_flight": 1'; then
    e=$(echo "$s" | grep elapsed_s | sed 's/[^0-9.]//g')
    if [ -n "$e" ] && python3 -c "import sys; sys.exit(0 if float('$e') > 8 else 1)"; then echo "pushed (elapsed $e s)"; exit 0; fi
  fi
  sleep 5
done
echo timeout; exit 1
