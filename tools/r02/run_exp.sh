# K-loop ablation of the small-P kernel: needs tools/experiments/r02_patch_resident_pixels.diff applied (it carries the
# SP_EXP_NO_DMA / SP_EXP_NO_FRAGS / SP_EXP_NO_ADDR switches) and the variant builds st_nofrags / st_nodma / st_noaddr / st_none of
# tools/build_variant.sh with -DSF_STAMP plus the respective switch.
for v in stamp st_nofrags st_nodma st_noaddr st_none; do
  for m in 1 0; do
    echo "== $v patch=$m"
    SF_SP_PATCH=$m SF_LIB_PATH=build_var/$v/libsfnative.so timeout 120 python tools/r02/stamps.py 1 50 50 2>/dev/null | grep -A1 "^launch  [039]:" | grep "consumer" 
  done
done
