#!/bin/bash
cd /root/repo; out=/root/repo/gpurun_out/r4r; mkdir -p $out
PYTHONUNBUFFERED=1 timeout 600 python -u -m pytest tests/test_adam.py -v -x -m gpu -p no:cacheprovider > $out/adam.log 2>&1 < /dev/null; echo "rc $?" >> $out/adam.log; tail -40 $out/adam.log
PYTHONUNBUFFERED=1 timeout 900 python -u -m pytest tests/test_networks.py tests/test_trainer_gpu.py tests/test_train_loop_gpu.py -q -x -m gpu -p no:cacheprovider > $out/rest.log 2>&1 < /dev/null; echo "rc $?" >> $out/rest.log; tail -30 $out/rest.log
