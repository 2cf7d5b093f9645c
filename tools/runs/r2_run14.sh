export GPU_MAX_HW_QUEUES=8
ETD_SCHED_STATS=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>&1 | grep "sched\]" | tail -8
