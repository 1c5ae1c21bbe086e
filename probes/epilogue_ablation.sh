# Store / residual ablations of the record conv's epilogue on 32 CUs (MDTILE_REC_GRID=32: no chip-level contention), by s_memtime stamps:
#   bash probes/epilogue_ablation.sh        (on the GPU box; profiles/r4x/epilogue_ablation_r4x.log)
# --dbg 16: the fp32 stores are skipped, 32: the record stores are skipped, 48: both (ConvRParams::dbg / EpiCtx::dbg)
export MDTILE_REC_GRID=32
for d in 0 16 32 48; do echo "== dbg=$d"; timeout 200 python probes/conv_item_timeline.py --dbg $d --forms "rec->both(+res),rec->f32+rec,rec->f32(+res)" 2>&1 | grep "128->128" | sed 's/\[cycles of s_memtime, 64 items stamped\]//' | cut -c1-330; done
