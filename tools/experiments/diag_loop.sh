#!/bin/bash
# Timing probes of the staged conv main loop: build libsfnative variants with -DSF_DIAG=n (conv_igemm.hip) here,
# then on the GPU box: bash tools/experiments/diag_loop.sh run
cd "$(dirname "$0")/../.."
D=tools/experiments/diag
if [ "$1" = build ]; then
  mkdir -p $D
  for n in ${DIAGS:-1 2 4 5}; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -D${DNAME:-SF_DIAG}=$n -c streamingflow_amd/csrc/conv_igemm.hip -o $D/conv_igemm_${DNAME:-SF_DIAG}_$n.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libsfnative_${DNAME:-SF_DIAG}_$n.so $D/conv_igemm_${DNAME:-SF_DIAG}_$n.o $(ls streamingflow_amd/csrc/*.o | grep -v conv_igemm) ) &
  done
  wait
  ls -la $D
else
  out=gpurun_out/diag_loop.txt
  : > $out
  echo "== product" >> $out; python tools/modbench.py --bigconvs 2>/dev/null | grep conv >> $out
  for n in ${DIAGS:-1 2 4 5}; do
    echo "== ${DNAME:-SF_DIAG}=$n" >> $out
    SF_LIB_PATH=$PWD/$D/libsfnative_${DNAME:-SF_DIAG}_$n.so python tools/modbench.py --bigconvs 2>/dev/null | grep conv >> $out
  done
  cat $out
fi
