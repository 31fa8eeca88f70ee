#!/bin/bash
cd /root/repo; bash scripts/pmc_small_convs.sh r04 2>&1 | tail -40; grep -i "error\|Traceback" gpurun_out/pmc_convs_r04/*.log | head -5
