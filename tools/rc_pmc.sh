#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU -d /tmp/q1 --output-format csv -- python3 $R/tools/time_raycast.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d /tmp/q2 --output-format csv -- python3 $R/tools/time_raycast.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/q3 --output-format csv -- python3 $R/tools/time_raycast.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/q4 --output-format csv -- python3 $R/tools/time_raycast.py > /dev/null 2>&1
cd $R && python3 tools/pmc_summary.py /tmp/q1 /tmp/q2 /tmp/q3 /tmp/q4 --match ray_ > $O/rc_tiles_pmc.json
rocprofv3 --kernel-trace --stats -d /tmp/q5 --output-format csv -- python3 $R/tools/time_raycast.py > /dev/null 2>&1
python3 tools/profile_summary.py trace $(ls /tmp/q5/*/*kernel_trace.csv | head -1) | grep -i "ray_\|kernel |\|---" > $O/rc_tiles_bygrid.md
python3 tools/time_raycast.py 2>&1 | grep ms_per > $O/rc_tiles_time.txt
