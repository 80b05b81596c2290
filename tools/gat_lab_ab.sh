#!/bin/bash
# tools/gat_lab.py on the in-tree library and on every variant under tools/ab/ (tools/build_variant.sh)
echo "== in-tree"; timeout -k 10 120 python tools/gat_lab.py 2>&1 | grep "TB/s" || exit 1
for so in tools/ab/lib_*.so; do
  [ -f "$so" ] || continue
  echo "== $so"; TRACEVAE_HIP_LIB=$PWD/$so timeout -k 10 120 python tools/gat_lab.py 2>&1 | grep "TB/s" || exit 1
done
