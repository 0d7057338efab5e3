#!/bin/bash
# tools/ab_envs.sh "<variant names>" <env> [bench args] ... one A/B table per env spec given on stdin lines
while read -r spec; do
  [ -z "$spec" ] && continue
  echo "== $spec"
  timeout -k 10 600 python tools/tools_ab.py $VARIANTS -- $spec || exit 1
done
