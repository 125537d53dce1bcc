#!/bin/bash
# LDS-DMA staging variants of the large-tile conv kernels (SF_GLDS / SF_GLDS_VAR, api.hip): parity, then timing
out=gpurun_out/sweep_glds.txt
: > $out
echo "== product" >> $out; python tools/modbench.py --bigconvs 2>/dev/null | grep conv >> $out
for v in ${VARS:-4 6 7}; do   # tile variants of the 64-cout family (launch_conv_glds); the buffer / issue-placement variants of the first sweeps are gone from the tree
  echo "== SF_GLDS=3 SF_GLDS_VAR=$v" >> $out
  SF_GLDS=3 SF_GLDS_VAR=$v python -m pytest tests/test_gpu_ops.py tests/test_gpu_conv_random.py tests/test_gpu_forward.py -x -q -m gpu -k "conv2d or random or config2" 2>&1 | tail -2 >> $out
  SF_GLDS=3 SF_GLDS_VAR=$v python tools/modbench.py --bigconvs 2>/dev/null | grep conv >> $out
done
cat $out
