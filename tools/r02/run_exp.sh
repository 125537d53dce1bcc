for v in stamp st_nofrags st_nodma st_noaddr st_none; do
  for m in 1 0; do
    echo "== $v patch=$m"
    SF_SP_PATCH=$m SF_LIB_PATH=build_r02/$v/libsfnative.so timeout 120 python tools/r02/stamps.py 1 50 50 2>/dev/null | grep -A1 "^launch  [039]:" | grep "consumer" 
  done
done
