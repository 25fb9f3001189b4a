#!/bin/bash
# run every test id matching $1 in its own process (a GPU fault in one does not hide the others): tools/run_isolated.sh 'split_k' [file]
cd "$(dirname "$0")/.."
F=${2:-tests/test_gpu_kernels.py}
python -m pytest $F -q --collect-only -k "$1" -p no:cacheprovider 2>/dev/null | grep "::" > /tmp/ids.txt
while read -r id; do
  timeout 120 python -m pytest "$id" -q -x -p no:cacheprovider > /tmp/one.log 2>&1
  rc=$?
  echo "rc=$rc $id"
  if [ $rc -ne 0 ]; then grep -i -m3 "fault\|error\|assert" /tmp/one.log | cut -c1-200; fi
done < /tmp/ids.txt
