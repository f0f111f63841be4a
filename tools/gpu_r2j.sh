#!/bin/bash
# round-2 stage j: whole-K 1x1 tile kernel (121..123) probes + A/B bench; SQ counter passes on the halo kernels
mkdir -p gpurun_out/j
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/j
(timeout 900 python -m pytest tests/test_kernels_conv.py -m gpu -q -x 2>&1 | grep -vE 'RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids' | tail -5) > $O/pytest_conv_kernels.log 2>&1
tail -2 $O/pytest_conv_kernels.log
S1=2,4,5,7,9,16,17,18,19,20
(timeout 300 python tools/conv_probe.py --mode stats --shapes $S1 --tiles 86,102,83,120,121,122,123,377,633,889 --reps 7 2>&1 | grep -v amdgpu.ids) > $O/conv_probe_1x1_stats.txt 2>&1
(timeout 300 python tools/conv_probe.py --mode dgrad --shapes $S1 --tiles 86,102,83,120,121,122,123 --reps 7 2>&1 | grep -v amdgpu.ids) > $O/conv_probe_1x1_dgrad.txt 2>&1
(timeout 300 python tools/conv_probe.py --mode fwd --shapes $S1 --tiles 86,102,83,121,122,123 --reps 7 2>&1 | grep -v amdgpu.ids) > $O/conv_probe_1x1_eval.txt 2>&1
cat $O/conv_probe_1x1_stats.txt $O/conv_probe_1x1_dgrad.txt $O/conv_probe_1x1_eval.txt
(timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_l_new.json 2>&1
(STREAMYOLO_HALO_TILES=112,113,114,115,116 STREAMYOLO_TILE_1X1K= timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_l_old.json 2>&1
(timeout 300 python bench.py --workload infer --model l --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_infer_l.json 2>&1
(STREAMYOLO_HALO_TILES=112,113,114,115,116 STREAMYOLO_TILE_1X1K= timeout 300 python bench.py --workload infer --model l --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_infer_l_old.json 2>&1
(timeout 600 python tools/profile_train.py 2>&1 | grep -v amdgpu.ids) > $O/train_l_layer_profile.txt 2>&1
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['value'],1), round(d['ms_per_step'],3), d.get('step_ms'), round(d['roofline']['frac'],4), d['roofline'].get('per_kind_ms'))" 2>&1 | cut -c1-600; done
head -75 $O/train_l_layer_profile.txt
# SQ counters: halo 115 vs 117 vs 118 (+ no-load ablations of 117) on d4.m.c2 and head0, and the 1x1 kernels on d4.m.c1
PA="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
PB="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
PD="SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL"
i=0
for P in "$PA" "$PB" "$PD"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_h$i -- python $GRAFT_REPO_ROOT/tools/conv_probe.py --mode stats --shapes 10,13 --tiles 115,117,118,885 --reps 3 > /dev/null 2>&1)
  echo "# pass $i: $P (3x3 stats: d4.m.c2 N16 38x60 256->256, head0 N8 75x120 256->256; tiles 115,117,118,117-noXW)" >> $O/pmc_halo.txt
  python tools/pmc_summary.py $O/pmc_h$i conv3x3_halo >> $O/pmc_halo.txt 2>&1
  (cd /tmp && timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_k$i -- python $GRAFT_REPO_ROOT/tools/conv_probe.py --mode stats --shapes 9,5 --tiles 86,121,123 --reps 3 > /dev/null 2>&1)
  echo "# pass $i: $P (1x1 stats: d4.m.c1 N16 38x60 256->256, d3.m.c1 N16 75x120 128->128; tiles 86,121,123)" >> $O/pmc_1x1.txt
  python tools/pmc_summary.py $O/pmc_k$i conv >> $O/pmc_1x1.txt 2>&1
  rm -rf $O/pmc_h$i $O/pmc_k$i
done
cat $O/pmc_halo.txt $O/pmc_1x1.txt
