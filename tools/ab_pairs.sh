#!/bin/bash
# stdin lines "variant|bench args": default against that variant on that workload
while IFS='|' read -r var spec; do
  [ -z "$var" ] && continue
  echo "== $var: $spec"
  timeout -k 10 900 python tools/tools_ab.py default $var -- $spec || exit 1
done
