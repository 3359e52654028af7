# throughput of the streamed C3 bench by depth (slabs in flight) and by the number of hardware queues the runtime multiplexes streams onto
for q in 8 16 24; do for d in 6 8 10 12; do echo -n "HWQ=$q depth=$d: "; GPU_MAX_HW_QUEUES=$q python bench.py --steps 48 --warmup 12 --depth $d --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['chunks_per_s'], d['ms_per_step'])"; done; done
