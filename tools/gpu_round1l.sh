#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
(timeout 600 python tools/profile_train.py 2>&1 | tail -75) > gpurun_out/profile_train_l.log 2>&1
cat gpurun_out/profile_train_l.log
