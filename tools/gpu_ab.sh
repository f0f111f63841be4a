cd $GRAFT_REPO_ROOT
for lib in base new; do
  if [ $lib = base ]; then export STREAMYOLO_HIP_LIB=$PWD/tools/probes/_build/libsy_base.so; else unset STREAMYOLO_HIP_LIB; fi
  echo "== $lib"
  python tools/conv_probe.py --mode stats --shapes 9+5+7+2 --tiles 121+122 --chain 8 2>&1 | grep -v amdgpu
  python tools/conv_probe.py --mode stats --shapes 10+6+11+3 --tiles 117+118 --chain 8 2>&1 | grep -v amdgpu
done
