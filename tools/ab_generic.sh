#!/bin/bash
# Runs on the GPU box: interleaved A/B runs.  usage: ab_generic.sh "<variants...>" "<bench args>" ["<bench args>" ...]
variants=$1; shift
for spec in "$@"; do
  echo "== $spec"
  timeout -k 10 900 python tools/tools_ab.py default $variants -- $spec || exit 1
done
