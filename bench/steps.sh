#!/bin/bash
# Runs GPU steps one after the other on the GPU box, each under its own time limit, and stops for good at the
# first step that ends in a time limit, an abort, a segmentation fault or a kill (nothing more may touch the GPU
# in the same call then); an ordinary failure (a failing test) is reported and the next step still runs.
#   bash bench/steps.sh <outdir> "<seconds> <name> <command...>" ...
out=$1; shift
mkdir -p "$out"
for spec in "$@"; do
  secs=${spec%% *}; rest=${spec#* }; name=${rest%% *}; cmd=${rest#* }
  echo "== $name (limit ${secs}s)"
  timeout -k 10 "$secs" bash -c "$cmd" > "$out/$name.log" 2>&1
  rc=$?
  echo "== $name rc=$rc"
  tail -n 6 "$out/$name.log"
  case $rc in
    124|137|134|139|143) echo "== stopping: $name ended with $rc"; exit $rc;;
  esac
done
exit 0
