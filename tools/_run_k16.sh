cd $GRAFT_REPO_ROOT
for l in "" blockcopy-video-processing-pytorch_amd/lib/dbg/libbc_w4dbg16.so; do
echo "=== lib $l"
if [ -n "$l" ]; then export BLOCKCOPY_HIP_LIB=$PWD/$l; fi
python tools/conv_stamps.py --cfg 4096 2>&1 | grep -A3 "layer1\|up1/4" | grep -v "^--"
done
