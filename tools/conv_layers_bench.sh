#!/bin/bash
# The U-Net's hot 3x3x3 layers through tools/conv_bench.py (fwd, dgrad; fp32 and bf16): the A/B harness of the forward kernel.
# usage: tools/conv_layers_bench.sh OUT.txt [--lib path/to/variant.so]
out=$1; shift
: > "$out"
for spec in "48 16" "16 16" "8 16" "16 32" "32 32" ; do
  set -- $spec "$@"
  ci=$1; co=$2; shift 2
  if [ "$ci" = "16" ] && [ "$co" = "32" ]; then dims="80 96 80"; elif [ "$ci" = "32" ]; then dims="80 96 80"; else dims="160 192 160"; fi
  python tools/conv_bench.py "$@" $ci $co $dims 2 10 fwd,dgrad >> "$out" 2>&1 || exit 1
  python tools/conv_bench.py "$@" $ci $co $dims 2 10 fwd,dgrad bf16 >> "$out" 2>&1 || exit 1
done
