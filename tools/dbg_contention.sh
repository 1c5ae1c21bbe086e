#!/bin/bash
# conv_in (k_conv3x3_fewcin) sharing the GPU with hand-over convs / the whole op mix: shipping form vs the round-5 form  (round 6; profiles/r6j)
mkdir -p gpurun_out
{
for form in "" 1; do
  for mode in "mix:handover" "mix:rec conv" "mix" "same:handover"; do
    echo "=== MDTILE_FEWCIN_FORM=$form, $mode"
    MDTILE_FEWCIN_FORM=$form timeout 300 python probes/contention_fewcin.py 4 100 "$mode" 2>&1 | grep -v "amdgpu\|failure\|wrong value\|^\[probes\]" | tail -1
  done
done
for form in "" 1; do MDTILE_FEWCIN_FORM=$form timeout 300 python probes/fewcin_time.py 2>&1 | grep "^form"; done
} > gpurun_out/contention_fix.log 2>&1
cat gpurun_out/contention_fix.log | cut -c1-300
