#!/bin/bash
# A/B timing of variant builds of the library (tools/ab/*.so); usage: tools/ab_run.sh <out.txt> "<views w h n reps>" lib1 lib2 ...
out=$1; shift
cfg=$1; shift
mkdir -p $(dirname $out); : > $out
for lib in "$@"; do
  python tools/time_lib.py tools/ab/$lib $cfg >> $out 2>&1
done
grep "kernel ms" $out
