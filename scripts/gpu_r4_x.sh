#!/bin/bash
cd /root/repo; bash scripts/pmc_adam.sh r04 2>&1 | tail -8; cat gpurun_out/pmc_adam_r04/*.log | grep -i "error\|Traceback" | head
