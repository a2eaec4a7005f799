#!/bin/bash
# Round 5: the row-resident 1x1 kernel — bit identity, per-layer timing against the previous heuristic (alt library) and phase ablations, in-situ A/B.
O=gpurun_out/${1:-r05b}; mkdir -p $O
timeout 300 python tools/check_conv_variants.py rows > $O/check_rows.txt 2>&1; echo "check rc=$?" >> $O/check_rows.txt
tail -4 $O/check_rows.txt
for v in - norows rows_noload rows_nomfma rows_noepi; do
  lib=-; [ "$v" != "-" ] && lib=gpurun_alt/libdir_hip_$v.so
  [ "$v" != "-" ] && [ ! -f $lib ] && continue
  timeout 300 python tools/probe_conv_variants.py 256 0 $lib rows > $O/probe_$v.txt 2>&1
  echo "== $v"; grep "k1" $O/probe_$v.txt | cut -c1-95
done
if [ "${2:-ab}" = "ab" ]; then
  timeout 600 python tools/ab_two_libs.py gpurun_alt/libdir_hip_norows.so 2 12 > $O/ab_rows.txt 2>&1; cat $O/ab_rows.txt
fi
