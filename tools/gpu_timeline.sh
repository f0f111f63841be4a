cd $GRAFT_REPO_ROOT; export STREAMYOLO_HIP_LIB=$PWD/tools/probes/_build/libstreamyolo_probe.so; mkdir -p gpurun_out/$1
( for args in "--kind bnred --shape 10" "--kind bnred --shape 6" "--kind conv --shape 9 --tile 121 --mode stats" "--kind conv --shape 10 --tile 117 --mode stats" "--kind wgrad --shape 10 --tile 59 --half 0"; do
  python tools/kernel_timeline.py $args 2>&1 | grep -v amdgpu.ids; echo; done ) > gpurun_out/$1/timelines.txt 2>&1
cat gpurun_out/$1/timelines.txt
